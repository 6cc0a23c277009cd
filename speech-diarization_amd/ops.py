"""Thin torch-tensor wrappers over the layer / cosine operators of libsd_hip.so.

Inputs and outputs are CUDA tensors; every call is asynchronous on the current
torch stream.  These are the operators `sd_ecapa_forward_f32` is built from, exposed
so that each can be parity-tested on its own, plus the cosine helpers used by the
clustering-side callers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as N

_ACT = {None: N.SD_ACT_NONE, "none": N.SD_ACT_NONE, "relu": N.SD_ACT_RELU, "tanh": N.SD_ACT_TANH, "sigmoid": N.SD_ACT_SIGMOID}


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("libsd_hip operators take GPU tensors; there is no CPU fallback")


def _ptr(t):
    return None if t is None else t.data_ptr()


def pack_weight(w: torch.Tensor | np.ndarray, device, dtype=torch.float32) -> torch.Tensor:
    """[cout, cin, k] -> packed [cout, k, cin_pad] on `device` (f32: cin_pad % 32 == 0, f16: % 64)."""
    from .engine import pack_conv_weight
    w = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
    npdt = np.float16 if dtype == torch.float16 else np.float32
    return torch.from_numpy(pack_conv_weight(w.astype(np.float32), npdt)).to(device)


def _dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return N.SD_DT_F32
    if t == torch.float16:
        return N.SD_DT_F16
    raise TypeError(f"unsupported dtype {t}")


def conv1d_cl(x: torch.Tensor, w_packed: torch.Tensor, T: int, *, cin: int, dil: int = 1, bias=None, bias_per_seg=False,
              act=None, scale=None, shift=None, act2=None, a_col0: int = 0, out: torch.Tensor | None = None,
              o_col0: int = 0, tee: torch.Tensor | None = None, tee_lo: int = 0, tee_hi: int = 0,
              tee_add: torch.Tensor | None = None, ta_col0: int = 0, out_dtype: torch.dtype | None = None,
              colstat: torch.Tensor | None = None) -> torch.Tensor:
    """Channel-last conv1d ("same", reflect) with the fused TDNN epilogue. x: [M, lda], returns [M, ldo].
    f32 weights -> exact-f32 operator; f16 weights -> f16-operand / f32-accumulate operator (x f32 or
    f16, output `out_dtype`).  `colstat` (f32, `colstat_floats(M, cout)` elements) receives the epilogue's
    per-tile column sums for `colstat_finish`."""
    _need_cuda(x, w_packed, bias, scale, shift, out, tee, tee_add, colstat)
    lib = N.load()
    cout, taps, cin_pad = w_packed.shape
    M = x.shape[0]
    half = w_packed.dtype == torch.float16
    if out is None:
        out = torch.empty((M, cout), dtype=out_dtype or (torch.float16 if half else torch.float32), device=x.device)
    a = N.sd_conv_args()
    a.x, a.lda, a.a_col0 = x.data_ptr(), x.stride(0), a_col0
    a.w, a.w_dtype = w_packed.data_ptr(), _dt(w_packed.dtype)
    a.x_dtype, a.y_dtype = _dt(x.dtype), _dt(out.dtype)
    for extra in (tee, tee_add):
        if extra is not None and extra.dtype != out.dtype:
            raise TypeError("tee / tee_add must have the output dtype")
    a.y, a.ldo, a.o_col0 = out.data_ptr(), out.stride(0), o_col0
    a.M, a.T = M, T
    a.cin, a.cin_pad, a.cout, a.taps, a.dil = cin, cin_pad, cout, taps, dil
    a.bias, a.bias_per_seg = _ptr(bias), int(bool(bias_per_seg))
    a.act, a.act2 = _ACT[act], _ACT[act2]
    a.scale, a.shift = _ptr(scale), _ptr(shift)
    if tee is not None:
        a.tee, a.ldt, a.tee_lo, a.tee_hi = tee.data_ptr(), tee.stride(0), tee_lo, tee_hi
        if tee_add is not None:
            a.tee_add, a.ld_ta, a.ta_col0 = tee_add.data_ptr(), tee_add.stride(0), ta_col0
    if colstat is not None:
        if colstat.dtype != torch.float32 or colstat.numel() < colstat_floats(M, cout):
            raise ValueError("colstat must be f32 with colstat_floats(M, cout) elements")
        a.colstat = colstat.data_ptr()
    with torch.cuda.device(x.device):
        fn, name = (lib.sd_conv1d_cl_f16, "sd_conv1d_cl_f16") if half else (lib.sd_conv1d_cl_f32, "sd_conv1d_cl_f32")
        N.check(fn(C.byref(a), _stream(x)), name)
    return out


def seg_gemm(x: torch.Tensor, w_packed: torch.Tensor, *, cin: int, bias=None, act=None, scale=None, shift=None, act2=None,
             out: torch.Tensor | None = None, scratch: torch.Tensor | None = None) -> torch.Tensor:
    """A per-segment layer (one row per segment: SE squeeze FC, global-context bias, final FC) through `sd_seg_gemm_f32`: with `scratch`
    (f32, `seg_gemm_scratch_bytes` bytes; allocated here when None) K is split over the grid for M <= 256 rows and cin_pad >= 512."""
    _need_cuda(x, w_packed, bias, scale, shift, out, scratch)
    lib = N.load()
    cout, taps, cin_pad = w_packed.shape
    M = x.shape[0]
    if taps != 1 or w_packed.dtype != torch.float32 or x.dtype != torch.float32:
        raise TypeError("seg_gemm: f32 activations and one-tap f32 weights")
    if out is None:
        out = torch.empty((M, cout), dtype=torch.float32, device=x.device)
    need = int(lib.sd_seg_gemm_scratch_bytes(M, cin_pad, cout))
    if scratch is None and need:
        scratch = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    a = N.sd_conv_args()
    a.x, a.lda, a.a_col0 = x.data_ptr(), x.stride(0), 0
    a.w, a.w_dtype = w_packed.data_ptr(), N.SD_DT_F32
    a.x_dtype, a.y_dtype = N.SD_DT_F32, N.SD_DT_F32
    a.y, a.ldo, a.o_col0 = out.data_ptr(), out.stride(0), 0
    a.M, a.T = M, 1
    a.cin, a.cin_pad, a.cout, a.taps, a.dil = cin, cin_pad, cout, 1, 1
    a.bias, a.bias_per_seg = _ptr(bias), 0
    a.act, a.act2 = _ACT[act], _ACT[act2]
    a.scale, a.shift = _ptr(scale), _ptr(shift)
    with torch.cuda.device(x.device):
        N.check(lib.sd_seg_gemm_f32(C.byref(a), scratch.data_ptr() if scratch is not None else None,
                                    scratch.numel() * 4 if scratch is not None else 0, _stream(x)), "sd_seg_gemm_f32")
    return out


def split16_pack(x: torch.Tensor, a_col0: int = 0, cin: int | None = None, mul: float = 1.0) -> torch.Tensor:
    """f32 [M, ld] columns [a_col0, a_col0 + cin) -> SD_DT_SPLIT16 rows as an f16 tensor [M, 2 * cin_pad32]
    (per 32 values: [hi x 32 | lo x 32], hi = f16(v), lo = f16(v - hi); padding values zero)."""
    _need_cuda(x)
    if x.dtype != torch.float32 or x.stride(1) != 1:
        raise TypeError("x must be f32 with contiguous channels")
    M = x.shape[0]
    cin = x.shape[1] - a_col0 if cin is None else cin
    cp = (cin + 31) // 32 * 32
    out = torch.empty((M, 2 * cp), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_split16_pack_f32(x.data_ptr(), x.stride(0), a_col0, M, cin, C.c_float(mul), out.data_ptr(), cp, _stream(x)), "sd_split16_pack_f32")
    return out


def pack_weight_split16(w: torch.Tensor | np.ndarray, device):
    """[cout, cin, k] -> (SD_DT_SPLIT16 weights as an f16 tensor [cout, k, cin_pad32 / 32, 64] on `device`, s):
    scaled by 2^s, hi / lo halves interleaved per 32 input channels (`engine.pack_conv_weight_split16`)."""
    from .engine import pack_conv_weight_split16
    w = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
    packed, s = pack_conv_weight_split16(w.astype(np.float32))
    return torch.from_numpy(packed).to(device), s


def conv1d_cl_split16(x: torch.Tensor, w_split: torch.Tensor, w_shift: int, T: int, *, cin: int, dil: int = 1, bias=None, bias_per_seg=False,
                      act=None, scale=None, shift=None, act2=None, a_col0: int = 0, out: torch.Tensor | None = None, o_col0: int = 0,
                      tee: torch.Tensor | None = None, tee_lo: int = 0, tee_hi: int = 0, tee_add: torch.Tensor | None = None, ta_col0: int = 0,
                      colstat: torch.Tensor | None = None, narrow: bool = False, out_split: torch.Tensor | None = None) -> torch.Tensor:
    """The "f32-split16x3" conv: f32 x [M, lda] -> f32 y [M, ldo] at f32-level accuracy on the f16 matrix cores.
    `w_split`, `w_shift` from `pack_weight_split16`; bias / scale / shift are the layer's ordinary f32 vectors.
    Default (wide outputs): `sd_split16_pack_f32` + the 256x256 kernel, the 2^s of the weight scaling folded into bias and scale here.
    `narrow=True`: the 128x128 kernel that splits the f32 activations while staging them (no pack pass; tee_add, per-segment bias and
    act2 allowed, no colstat), the 2^-s passed as `w_scale_inv`; `out_split`: its y as SD_DT_SPLIT16 rows instead of f32."""
    _need_cuda(x, w_split, bias, scale, shift, out, tee, tee_add, colstat)
    lib = N.load()
    cout, taps, chunks, _ = w_split.shape
    cp = chunks * 32
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, cout), dtype=torch.float32, device=x.device)
    f = float(2.0 ** w_shift)
    a = N.sd_conv_args()
    if narrow:
        if x.dtype != torch.float32 or x.stride(1) != 1:
            raise TypeError("x must be f32 with contiguous channels")
        keep = (x,)
        a.x, a.lda, a.a_col0, a.x_dtype = x.data_ptr(), x.stride(0), a_col0, N.SD_DT_F32
        a.bias, a.scale = _ptr(bias), _ptr(scale)
        a.w_scale_inv = 1.0 / f
    else:
        if bias_per_seg or act2 is not None or tee_add is not None:
            raise ValueError("per-segment bias / act2 / tee_add belong to the narrow kernel (narrow=True)")
        xs = split16_pack(x, a_col0, cin)
        bias_s = None if bias is None else bias * f
        scale_s = (torch.full((cout,), 1.0 / f, dtype=torch.float32, device=x.device) if scale is None else scale / f)
        keep = (xs, bias_s, scale_s)
        a.x, a.lda, a.a_col0, a.x_dtype = xs.data_ptr(), cp, 0, N.SD_DT_SPLIT16
        a.bias, a.scale = _ptr(bias_s), _ptr(scale_s)
    a.w, a.w_dtype = w_split.data_ptr(), N.SD_DT_SPLIT16
    a.y_dtype = N.SD_DT_F32
    a.y, a.ldo, a.o_col0 = out.data_ptr(), out.stride(0), o_col0
    if out_split is not None:
        # y leaves as SD_DT_SPLIT16 rows (f16 [M, 2 * ld], ld VALUE columns, a multiple of 32) instead of f32 --
        # bit for bit what split16_pack would make of the f32 result; `out` is then not written
        if out_split.dtype != torch.float16 or out_split.stride(1) != 1 or out_split.shape[1] % 64 or colstat is not None:
            raise TypeError("out_split: f16 [M, 2 * ld] with ld % 32 == 0, no colstat")
        _need_cuda(out_split)
        a.y_dtype = N.SD_DT_SPLIT16
        a.y, a.ldo = out_split.data_ptr(), out_split.shape[1] // 2
    a.M, a.T = M, T
    a.cin, a.cin_pad, a.cout, a.taps, a.dil = cin, cp, cout, taps, dil
    a.bias_per_seg = int(bool(bias_per_seg))
    a.act, a.act2 = _ACT[act], _ACT[act2]
    a.shift = _ptr(shift)
    if tee is not None:
        a.tee, a.ldt, a.tee_lo, a.tee_hi = tee.data_ptr(), tee.stride(0), tee_lo, tee_hi
        if tee_add is not None:
            a.tee_add, a.ld_ta, a.ta_col0 = tee_add.data_ptr(), tee_add.stride(0), ta_col0
    if colstat is not None:
        a.colstat = colstat.data_ptr()
    with torch.cuda.device(x.device):
        N.check(lib.sd_conv1d_cl_split16(C.byref(a), _stream(x)), "sd_conv1d_cl_split16")
    del keep
    return out


def res2net_chain_supported(T: int, chunk: int = 128, n: int = 7, taps: int = 3, dil: int = 2) -> bool:
    return bool(N.load().sd_res2net_chain_supported(T, chunk, n, taps, dil))


def res2net_chain(r: torch.Tensor, T: int, layers: list[dict]) -> torch.Tensor:
    """The Res2Net chain of one SE-Res2Net block in one kernel, IN PLACE on the tdnn1 output r [B*T, ld] (f16):
    y_1 = TDNN_1(c_1), y_j = TDNN_j(c_j + y_{j-1}); c_j = columns [128 j, 128 j + 128), y_j overwrites c_j.
    layers[j]: dict(w=packed f16 [128, 3, 128], bias, scale, shift (f32 [128] or None), dil)."""
    _need_cuda(r)
    if r.dtype != torch.float16 or r.stride(1) != 1:
        raise TypeError("r must be f16 with contiguous channels")
    lib = N.load()
    arr = (N.sd_layer * len(layers))()
    for L, d in zip(arr, layers):
        _need_cuda(d["w"], d.get("bias"), d.get("scale"), d.get("shift"))
        cout, taps, cin_pad = d["w"].shape
        L.w, L.bias, L.scale, L.shift = d["w"].data_ptr(), _ptr(d.get("bias")), _ptr(d.get("scale")), _ptr(d.get("shift"))
        L.cin, L.cin_pad, L.cout, L.taps, L.dil, L.w_dtype = d.get("cin", cin_pad), cin_pad, cout, taps, d["dil"], _dt(d["w"].dtype)
    B = r.shape[0] // T
    with torch.cuda.device(r.device):
        ws_bytes = max(256, int(lib.sd_res2net_chain_workspace_bytes(len(layers))))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=r.device)
        N.check(lib.sd_res2net_chain_f16(r.data_ptr(), r.stride(0), B, T, arr, len(layers), ws.data_ptr(), ws_bytes, _stream(r)), "sd_res2net_chain_f16")
    return r


def colstat_floats(M: int, cout: int) -> int:
    return int(N.load().sd_colstat_floats(M, cout))


def colstat_finish(colstat: torch.Tensor, y: torch.Tensor, B: int, T: int, *, pivot: torch.Tensor | None = None, want_std: bool = False,
                   eps: float = 1e-12, y_col0: int = 0, C_: int | None = None) -> torch.Tensor:
    """Per-segment mean (and std) of a conv output from the column sums its epilogue left in `colstat`
    (`pivot` = the conv's `shift`, or None).  -> f32 [B, C] or [B, 2C] = [mean | std]."""
    _need_cuda(colstat, y, pivot)
    C_ = y.shape[1] - y_col0 if C_ is None else C_
    out = torch.empty((B, (2 if want_std else 1) * C_), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        N.check(N.load().sd_colstat_finish_dt(colstat.data_ptr(), _ptr(pivot), y.data_ptr(), _dt(y.dtype), y.stride(0), y_col0, B, T, C_,
                                              int(want_std), C.c_float(eps), out.data_ptr(), _stream(y)), "sd_colstat_finish_dt")
    return out


def seg_mean(x: torch.Tensor, B: int, T: int, col0: int = 0, C_: int | None = None) -> torch.Tensor:
    _need_cuda(x)
    C_ = x.shape[1] - col0 if C_ is None else C_
    out = torch.empty((B, C_), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_seg_mean_f32(x.data_ptr(), x.stride(0), col0, B, T, C_, out.data_ptr(), _stream(x)), "sd_seg_mean_f32")
    return out


def seg_mean_std(x: torch.Tensor, B: int, T: int, eps: float = 1e-12) -> torch.Tensor:
    _need_cuda(x)
    C_ = x.shape[1]
    out = torch.empty((B, 2 * C_), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_seg_mean_std_f32(x.data_ptr(), x.stride(0), 0, B, T, C_, C.c_float(eps), out.data_ptr(), _stream(x)),
                "sd_seg_mean_std_f32")
    return out


def se_scale_residual(x: torch.Tensor, gate: torch.Tensor, res: torch.Tensor, B: int, T: int) -> torch.Tensor:
    _need_cuda(x, gate, res)
    C_ = x.shape[1]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_se_scale_residual_f32(x.data_ptr(), x.stride(0), gate.data_ptr(), res.data_ptr(), res.stride(0), 0,
                                                  y.data_ptr(), y.stride(0), 0, B, T, C_, _stream(x)), "sd_se_scale_residual_f32")
    return y


def asp_pool(logit: torch.Tensor, h: torch.Tensor, B: int, T: int, eps: float = 1e-12) -> torch.Tensor:
    _need_cuda(logit, h)
    C_ = h.shape[1]
    out = torch.empty((B, 2 * C_), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        N.check(N.load().sd_asp_pool_f32(logit.data_ptr(), logit.stride(0), h.data_ptr(), h.stride(0), B, T, C_, C.c_float(eps),
                                         out.data_ptr(), _stream(h)), "sd_asp_pool_f32")
    return out


def asp_attend_pool(a1: torch.Tensor, wc_packed: torch.Tensor, h: torch.Tensor, B: int, T: int, eps: float = 1e-12, split16: bool = False) -> torch.Tensor:
    """softmax_T(a1 @ wc^T) weighted mean / std of h in one kernel (`asp.conv` + pooling of speechbrain's
    AttentiveStatisticsPooling): a1 [B*T, att], wc_packed from `pack_weight`, h [B*T, C] -> f32 [B, 2C].
    Raises for geometries the fused kernel does not cover (see `asp_attend_pool_supported`)."""
    _need_cuda(a1, h, wc_packed)
    lib = N.load()
    C_, att = h.shape[1], a1.shape[1]
    if a1.dtype != h.dtype or wc_packed.dtype != h.dtype:
        raise ValueError("a1, wc and h must share a dtype")
    dt = N.SD_DT_F16 if h.dtype == torch.float16 else N.SD_DT_F32
    if split16:       # f32 tensors, the logits product as three f16 MFMA products per value pair (the f32-split16x3 mode)
        if h.dtype != torch.float32:
            raise ValueError("split16 takes f32 tensors")
        dt = N.SD_DT_SPLIT16
    a1 = a1.contiguous()
    out = torch.empty((B, 2 * C_), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        N.check(lib.sd_asp_attend_pool_dt(a1.data_ptr(), wc_packed.data_ptr(), h.data_ptr(), dt, h.stride(0), B, T, C_, att,
                                          C.c_float(eps), out.data_ptr(), _stream(h)), "sd_asp_attend_pool_dt")
    return out


def asp_attend_pool_supported(dtype: torch.dtype, T: int, C_: int, att: int) -> bool:
    return bool(N.load().sd_asp_attend_pool_supported(N.SD_DT_F16 if dtype == torch.float16 else N.SD_DT_F32, T, C_, att))


def l2norm_rows(x: torch.Tensor, eps_add: float = 0.0, sklearn_zero_guard: bool = False) -> torch.Tensor:
    _need_cuda(x)
    x = x.contiguous().float()
    n, d = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_l2norm_rows_f32(x.data_ptr(), d, n, d, C.c_float(eps_add), int(sklearn_zero_guard), out.data_ptr(), d,
                                            _stream(x)), "sd_l2norm_rows_f32")
    return out


def cosine_affinity(x: torch.Tensor, out: torch.Tensor | None = None, rows: tuple[int, int] | None = None,
                    split16: bool = False) -> torch.Tensor:
    """sklearn `cosine_similarity(X)` semantics on the GPU: f32 [N, D] -> f32 [N, N]
    (or rows [lo, hi) of it -> [hi-lo, N] when `rows` is given).  `split16`: same result to ~3e-7 through
    the f16 matrix cores (rows split hi + lo), for very large N."""
    _need_cuda(x)
    lib = N.load()
    x = x.contiguous().float()
    n, d = x.shape
    lo, hi = (0, n) if rows is None else rows
    if out is None:
        out = torch.empty((hi - lo, n), dtype=torch.float32, device=x.device)
    if n == 0 or hi == lo:
        return out
    with torch.cuda.device(x.device):
        size_fn, fn, name = ((lib.sd_cosine_split16_workspace_bytes, lib.sd_cosine_affinity_rows_split16, "sd_cosine_affinity_rows_split16")
                             if split16 else (lib.sd_cosine_workspace_bytes, lib.sd_cosine_affinity_rows_f32, "sd_cosine_affinity_rows_f32"))
        ws_bytes = int(size_fn(n, d))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
        N.check(fn(x.data_ptr(), n, d, lo, hi, out.data_ptr(), out.stride(0), ws.data_ptr(), ws_bytes, _stream(x)), name)
    return out


def adjacent_cosine(x: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    _need_cuda(x)
    x = x.contiguous().float()
    n, d = x.shape
    out = torch.empty((max(n - 1, 0),), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_adjacent_cosine_f32(x.data_ptr(), d, n, d, C.c_float(eps), out.data_ptr(), _stream(x)),
                "sd_adjacent_cosine_f32")
    return out


def sim_argmax(w: torch.Tensor, c: torch.Tensor):
    _need_cuda(w, c)
    w, c = w.contiguous().float(), c.contiguous().float()
    n, d = w.shape
    k = c.shape[0]
    best = torch.empty((n,), dtype=torch.int32, device=w.device)
    score = torch.empty((n,), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        N.check(N.load().sd_sim_argmax_f32(w.data_ptr(), d, n, d, c.data_ptr(), d, k, best.data_ptr(), score.data_ptr(), _stream(w)),
                "sd_sim_argmax_f32")
    return best, score


def topk_mean_std(x: torch.Tensor, k: int) -> torch.Tensor:
    """[rows, n] f32 -> [rows, 2] = mean and population std of each row's k largest values (k clipped to n).
    Precondition: finite values (a NaN would be selected into the top-k and poison the statistics)."""
    _need_cuda(x)
    x = x.float()
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, n = x.shape
    out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.load().sd_topk_mean_std_f32(x.data_ptr(), x.stride(0), rows, n, int(k), out.data_ptr(), _stream(x)), "sd_topk_mean_std_f32")
    return out


def asnorm_scores(query: torch.Tensor, centers: torch.Tensor, cohort: torch.Tensor, topk: int = 200) -> torch.Tensor:
    """`asnorm_scores` [REF diar_diag.py:196-208] on the device: rows scaled by 1 / (norm + 1e-9), the three cosine
    products on the f32 matrix cores (the operator behind the 1x1 convs, T = 1), top-k cohort statistics per row,
    z-normalisation against the query's and the centre's cohort scores, averaged.  f32 [nq, nr].  Precondition: finite inputs."""
    _need_cuda(query, centers, cohort)
    lib = N.load()
    qn, rn, cn = (l2norm_rows(t, eps_add=1e-9) for t in (query, centers, cohort))
    d = qn.shape[1]

    def product(a, b):                                   # a @ b.T with b as the packed "weights" [rows, 1, d_pad]
        wp = pack_weight(b.unsqueeze(2), b.device, torch.float32) if d % 32 else b.reshape(b.shape[0], 1, d).contiguous()
        return conv1d_cl(a, wp, 1, cin=d)

    raw = product(qn, rn)
    k = min(int(topk), cn.shape[0])
    qstat = topk_mean_std(product(qn, cn), k)
    rstat = topk_mean_std(product(rn, cn), k)
    out = torch.empty_like(raw)
    with torch.cuda.device(raw.device):
        N.check(lib.sd_asnorm_combine_f32(raw.data_ptr(), raw.stride(0), raw.shape[0], raw.shape[1], qstat.data_ptr(), rstat.data_ptr(),
                                          out.data_ptr(), out.stride(0), _stream(raw)), "sd_asnorm_combine_f32")
    return out


def viterbi(scores: torch.Tensor, alpha: float = 0.995) -> torch.Tensor:
    """`viterbi_hmm` [REF diar_diag.py:231-247] on the device: scores f32 [T, K <= 64] -> int32 path [T]; the two
    transition log-probabilities are formed as the reference forms them (float64 log, rounded to f32).  Precondition: finite
    scores (a NaN candidate is never selected here, np.argmax would return it); K == 1 uses log_move = 0 where the reference
    divides by K - 1 = 0."""
    _need_cuda(scores)
    scores = scores.contiguous().float()
    T, K = scores.shape
    path = torch.empty((T,), dtype=torch.int32, device=scores.device)
    if T == 0:
        return path
    eps = 1e-8
    log_move = float(np.float32(np.log((1 - alpha) / (K - 1) + eps))) if K > 1 else 0.0
    log_stay = float(np.float32(np.log(alpha + eps)))
    lib = N.load()
    with torch.cuda.device(scores.device):
        nbytes = int(lib.sd_viterbi_workspace_bytes(T, K))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=scores.device)
        N.check(lib.sd_viterbi_f32(scores.data_ptr(), scores.stride(0), T, K, C.c_float(log_stay), C.c_float(log_move), ws.data_ptr(), nbytes,
                                   path.data_ptr(), _stream(scores)), "sd_viterbi_f32")
    return path
