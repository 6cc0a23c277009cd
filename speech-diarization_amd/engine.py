"""Device-resident embedding engine: waveform [B, n] on the GPU -> 192-d embeddings on the GPU.

This is the host side of the hot path: it owns the packed weights, the fbank plan
and one workspace in HBM, and issues `sd_fbank_f32` + `sd_ecapa_forward_f32` on the
current torch stream.  PyTorch is used for device memory and streams only; all
arithmetic is in libsd_hip.so.

The reference re-creates its feature transform and crosses host<->device three times
per batch [REF speech_encode.py:17-38,76-77]; here tables and weights are uploaded
once and a batch costs one H2D (waveforms) and one D2H (embeddings) at the API edge.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _native as N
from .features import FbankPlan
from .synth import EcapaConfig, config_from_state_dict

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, as used by speechbrain's BatchNorm1d
K_ALIGN = 32   # K step of the conv/GEMM kernel


def _np(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def _pad_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


K_ALIGN_F16 = 64  # K step of the f16 conv/GEMM kernel


def pack_conv_weight(w: np.ndarray, dtype=np.float32) -> np.ndarray:
    """[cout, cin, k] (torch Conv1d) -> [cout, k, cin_pad], zero padded (kernel layout);
    f32 with cin_pad % 32 == 0, or f16 with cin_pad % 64 == 0."""
    cout, cin, k = w.shape
    cin_pad = _pad_to(cin, K_ALIGN if dtype == np.float32 else K_ALIGN_F16)
    out = np.zeros((cout, k, cin_pad), dtype=dtype)
    out[:, :, :cin] = np.transpose(w, (0, 2, 1)).astype(dtype)
    return out


def split16_exponent(w: np.ndarray) -> int:
    """s with max |w| 2^s in [512, 1024) (0 for an all-zero tensor)."""
    wmax = float(np.abs(w).max())
    if not wmax > 0:
        return 0
    s = int(np.floor(np.log2(1024.0 / wmax)))
    return s - 1 if wmax * 2.0 ** s >= 1024.0 else s


def pack_conv_weight_split16(w: np.ndarray):
    """[cout, cin, k] f32 -> (SD_DT_SPLIT16 [cout, k, cin_pad / 32, 64] f16, s): the weights scaled by 2^s (max |w| 2^s in
    [512, 1024): the low halves of small weights stay clear of the f16 subnormals, nothing overflows), every value split
    hi = f16(v), lo = f16(v - hi), interleaved per 32 input channels [hi x 32 | lo x 32] (sd_hip.h: SD_DT_SPLIT16)."""
    cout, cin, k = w.shape
    cp = _pad_to(cin, 32)
    s = split16_exponent(w)
    v = np.zeros((cout, k, cp), dtype=np.float32)
    v[:, :, :cin] = np.transpose(w, (0, 2, 1)) * np.float32(2.0 ** s)          # exact: power of two
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    out = np.empty((cout, k, cp // 32, 64), dtype=np.float16)
    out[..., :32] = hi.reshape(cout, k, cp // 32, 32)
    out[..., 32:] = lo.reshape(cout, k, cp // 32, 32)
    return out, s


def bn_affine(sd: dict, prefix: str):
    g, b = _np(sd[f"{prefix}.weight"]), _np(sd[f"{prefix}.bias"])
    rm, rv = _np(sd[f"{prefix}.running_mean"]), _np(sd[f"{prefix}.running_var"])
    scale = (g.astype(np.float64) / np.sqrt(rv.astype(np.float64) + BN_EPS))
    shift = b.astype(np.float64) - rm.astype(np.float64) * scale
    return scale.astype(np.float32), shift.astype(np.float32)


class EcapaWeights:
    """ECAPA-TDNN weights packed for the HIP kernels and resident on one device."""

    def __init__(self, state_dict: dict, device: torch.device, precision: str = "f32"):
        if precision not in ("f32", "f16", "f32s", "f32ns"):
            raise ValueError(f"precision must be 'f32', 'f16', 'f32s' (f32-split16x3) or 'f32ns' (exact-f32 wide layers, split16x3 narrow ones), got {precision!r}")
        self.device = device
        self.precision = precision
        self.split_narrow = True      # f32s: also the narrow convs on the split kernel (False: wide layers only, as first built)
        self.cfg: EcapaConfig = config_from_state_dict(state_dict)
        cfg = self.cfg
        if len(set(cfg.channels[:-1])) != 1 or cfg.channels[-1] != cfg.n_blocks * cfg.channels[0]:
            raise ValueError(f"unsupported ECAPA geometry {cfg.channels}: blocks must share one width and MFA = n_blocks * width")
        self._keep: list[torch.Tensor] = []
        sd = state_dict
        W = N.sd_ecapa_weights()
        W.w_dtype = N.SD_DT_F16 if precision == "f16" else N.SD_DT_F32
        W.split16 = {"f32s": 1, "f32ns": 2}.get(precision, 0)      # 2: only the narrow layers carry the split packing
        W.n_mels = cfg.input_size
        W.channels = cfg.channels[0]
        W.n_blocks = cfg.n_blocks
        W.res2_scale = cfg.res2net_scale
        W.mfa_channels = cfg.channels[-1]
        W.att_channels = cfg.attention_channels
        W.emb_dim = cfg.lin_neurons
        W.asp_eps = 1e-12
        self._fill(W.block0, _np(sd["blocks.0.conv.conv.weight"]), _np(sd["blocks.0.conv.conv.bias"]),
                   bn_affine(sd, "blocks.0.norm.norm"), cfg.dilations[0])
        for i in range(1, cfg.n_blocks + 1):
            blk = W.blocks[i - 1]
            p = f"blocks.{i}"
            d = cfg.dilations[i]
            self._fill(blk.tdnn1, _np(sd[f"{p}.tdnn1.conv.conv.weight"]), _np(sd[f"{p}.tdnn1.conv.conv.bias"]),
                       bn_affine(sd, f"{p}.tdnn1.norm.norm"), 1)
            for j in range(cfg.res2net_scale - 1):
                q = f"{p}.res2net_block.blocks.{j}"
                self._fill(blk.res2[j], _np(sd[f"{q}.conv.conv.weight"]), _np(sd[f"{q}.conv.conv.bias"]),
                           bn_affine(sd, f"{q}.norm.norm"), d)
            self._fill(blk.tdnn2, _np(sd[f"{p}.tdnn2.conv.conv.weight"]), _np(sd[f"{p}.tdnn2.conv.conv.bias"]),
                       bn_affine(sd, f"{p}.tdnn2.norm.norm"), 1)
            self._fill(blk.se1, _np(sd[f"{p}.se_block.conv1.conv.weight"]), _np(sd[f"{p}.se_block.conv1.conv.bias"]), None, 1, per_segment=True)
            self._fill(blk.se2, _np(sd[f"{p}.se_block.conv2.conv.weight"]), _np(sd[f"{p}.se_block.conv2.conv.bias"]), None, 1, per_segment=True)
        self._fill(W.mfa, _np(sd["mfa.conv.conv.weight"]), _np(sd["mfa.conv.conv.bias"]), bn_affine(sd, "mfa.norm.norm"), 1)
        cm = cfg.channels[-1]
        wa = _np(sd["asp.tdnn.conv.conv.weight"])  # [att, 3*cm, 1] acting on cat(h, mean, std)
        self._fill(W.asp_tdnn_h, wa[:, :cm], None, bn_affine(sd, "asp.tdnn.norm.norm"), 1)
        self._fill(W.asp_tdnn_g, wa[:, cm:], _np(sd["asp.tdnn.conv.conv.bias"]), None, 1, per_segment=True)
        self._fill(W.asp_conv, _np(sd["asp.conv.conv.weight"]), _np(sd["asp.conv.conv.bias"]), None, 1)
        if precision in ("f32s", "f32ns"):      # the fused pooling kernel splits these f32 weights in registers: it needs their 2^s
            W.asp_conv.split_scale_inv = float(2.0 ** -split16_exponent(_np(sd["asp.conv.conv.weight"])))
        # asp_bn is an affine map in front of a linear layer: fold it into fc (float64 on the host)
        s, t = bn_affine(sd, "asp_bn.norm")
        wf = _np(sd["fc.conv.weight"]).astype(np.float64)[:, :, 0]
        bf = _np(sd["fc.conv.bias"]).astype(np.float64)
        w_fold = (wf * s.astype(np.float64)[None, :]).astype(np.float32)[:, :, None]
        b_fold = (bf + wf @ t.astype(np.float64)).astype(np.float32)
        self._fill(W.fc, w_fold, b_fold, None, 1, per_segment=True)
        self.struct = W
        self.n_params = sum(int(np.prod(v.shape)) for k, v in sd.items() if "running" not in k and "num_batches" not in k)

    def _dev(self, a: np.ndarray, dtype=np.float32) -> int:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def _fill(self, L, w: np.ndarray, bias, affine, dil: int, per_segment: bool = False) -> None:
        """Frame-level layers (M = B*T rows) follow the engine precision; per-segment layers
        (M = B rows: SE gate, global-context bias, FC) always stay f32."""
        cout, cin, k = w.shape
        half = self.precision == "f16" and not per_segment
        wdt = np.float16 if half else np.float32
        packed = pack_conv_weight(w, wdt)
        L.w = self._dev(packed, wdt)
        L.w_dtype = N.SD_DT_F16 if half else N.SD_DT_F32
        L.bias = self._dev(bias) if bias is not None else None
        if affine is not None:
            L.scale, L.shift = self._dev(affine[0]), self._dev(affine[1])
        else:
            L.scale, L.shift = None, None
        L.cin, L.cin_pad, L.cout, L.taps, L.dil = cin, packed.shape[2], cout, k, dil
        # "f32s": the frame-level conv layers get a second, split-f16 packing.  Wide outputs (stem, tdnn1, tdnn2, MFA: 86 % of
        # the flops; 256x256 kernel, SD_DT_SPLIT16 activations) carry the weight scale 2^s in bias * 2^s / scale * 2^-s; narrow
        # ones (Res2Net convs, attention TDNN; 128x128 kernel that splits f32 activations while staging) pass 2^-s to the kernel
        L.w_split, L.bias_split, L.scale_split, L.split_scale_inv = None, None, None, 0.0
        if self.precision in ("f32s", "f32ns") and not per_segment and cin % 4 == 0:
            if self.precision == "f32s" and cout >= 1024 and affine is not None and bias is not None:
                ws, s = pack_conv_weight_split16(w)
                L.w_split = self._dev(ws, np.float16)
                L.bias_split = self._dev(bias * np.float32(2.0 ** s))
                L.scale_split = self._dev(affine[0] * np.float32(2.0 ** -s))
            elif cout <= 256 and self.split_narrow:
                ws, s = pack_conv_weight_split16(w)
                L.w_split = self._dev(ws, np.float16)
                L.split_scale_inv = float(2.0 ** -s)


class EmbeddingEngine:
    """fbank + ECAPA-TDNN on one GPU. Thread-safe (one forward at a time per engine)."""

    def __init__(self, state_dict: dict, device="cuda", max_batch: int = 512, precision: str = "f32"):
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError(f"EmbeddingEngine runs on the GPU only (got device {self.device}); there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = N.load()
        self.max_batch = int(max_batch)
        self._lock = threading.Lock()
        with torch.cuda.device(self.device):
            self.weights = EcapaWeights(state_dict, self.device, precision)
            self.plan = FbankPlan("speechbrain", n_mels=self.weights.cfg.input_size)
        self.dim = self.weights.cfg.lin_neurons
        self.precision = precision
        # "f32s" = f32-split16x3: the f32 schedule (f32 activations) whose wide layers run three f16 MFMA products per value
        # pair at f32-level accuracy; a property of the packed weights (sd_ecapa_weights.split16), same entry point
        self._forward = self._lib.sd_ecapa_forward_f16 if precision == "f16" else self._lib.sd_ecapa_forward_f32
        self._ws = None
        self._ws_frozen = False

    # -- workspace: feats + fbank scratch + ECAPA activations, three flat buffers that only ever grow (keyed by
    # capacity, not by the exact (B, n): the reference's callers send batches whose padded length changes every call
    # [REF anti_stick_diarize.py:163-166], and re-allocating ~10 MB per segment per call was a measurable part of it)
    def _workspace(self, B: int, n: int):
        T = FbankPlan.num_frames(n)
        n_mels = self.weights.cfg.input_size
        need = (B * T * n_mels * 4, max(self.plan.workspace_bytes(B, n), 256),
                int(self._lib.sd_ecapa_workspace_bytes(C.byref(self.weights.struct), B, T)))
        if self._ws is None:
            self._ws = [None, None, None]
        for i, nbytes in enumerate(need):
            if self._ws[i] is None or self._ws[i].numel() < nbytes:
                if self._ws_frozen:
                    raise RuntimeError("this engine's workspace is referenced by a captured hipGraph (StreamingEmbedder); "
                                       "use a separate EmbeddingEngine for larger batches")
                self._ws[i] = None                     # release before growing
                self._ws[i] = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        return self._ws

    def sibling(self) -> "EmbeddingEngine":
        """A second engine over the SAME packed weights and fbank tables (read-only on the device) with a workspace and a lock of
        its own: two siblings can run forwards on two streams at once (`HipEcapaEncoder.encode_batches`)."""
        other = object.__new__(EmbeddingEngine)
        other.__dict__.update(self.__dict__)
        other._lock = threading.Lock()
        other._ws = None
        other._ws_frozen = False
        return other

    def freeze_workspace(self) -> None:
        """After a hipGraph capture of `embed`: the captured launches hold the workspace pointers, so it may no longer move."""
        self._ws_frozen = True

    def embed(self, wav: torch.Tensor) -> torch.Tensor:
        """wav: f32 [B, n] on this engine's device -> f32 [B, dim] on the device (async on the current stream)."""
        if wav.dim() != 2:
            raise AssertionError("wav must be [B, n]")
        if wav.device != self.device:
            raise ValueError(f"wav is on {wav.device}, engine on {self.device}")
        wav = wav.contiguous().float()
        B, n = wav.shape
        out = torch.empty((B, self.dim), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        if n < 5 * 160:
            raise ValueError(f"segment of {n} samples is too short: ECAPA's reflect padding needs at least 5 frames (800 samples)")
        with self._lock, torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            mb = min(self.max_batch, B)
            feats, fb_ws, ec_ws = self._workspace(mb, n)
            T = FbankPlan.num_frames(n)
            W = C.byref(self.weights.struct)
            for lo in range(0, B, mb):
                nb = min(mb, B - lo)
                x = wav[lo:lo + nb]
                N.check(self._lib.sd_fbank_f32(self.plan.handle, x.data_ptr(), nb, n, 1, feats.data_ptr(), self.weights.cfg.input_size,
                                               fb_ws.data_ptr(), fb_ws.numel(), stream), "sd_fbank_f32")
                N.check(self._forward(W, feats.data_ptr(), nb, T, out[lo:lo + nb].data_ptr(),
                                      ec_ws.data_ptr(), ec_ws.numel(), stream), f"sd_ecapa_forward_{self.precision}")
        return out

    def embed_windows(self, signal: torch.Tensor, starts: torch.Tensor, n: int) -> torch.Tensor:
        """Embeddings of the B windows `signal[starts[b] : starts[b] + n]` of ONE recording (zeros where a window hangs
        over an end of the signal), without gathering them: `sd_fbank_windows_f32` reads the resident signal at
        `starts[b]`, so the recording crosses PCIe once and no [B, n] matrix is written to and re-read from HBM.
        Bitwise `embed(gathered windows)`.  signal: f32 [n_total] on the device; starts: int64 [B] (any device);
        -> f32 [B, dim] on the device.  [REF anti_stick_diarize.py:82-100, 396-430]: the callers' window loops."""
        if signal.dim() != 1:
            raise AssertionError("signal must be [n_total]")
        if signal.device != self.device:
            raise ValueError(f"signal is on {signal.device}, engine on {self.device}")
        signal = signal.contiguous().float()
        starts = starts.to(self.device, dtype=torch.int64).contiguous()
        B, n = int(starts.numel()), int(n)
        out = torch.empty((B, self.dim), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        if n < 5 * 160:
            raise ValueError(f"window of {n} samples is too short: ECAPA's reflect padding needs at least 5 frames (800 samples)")
        if signal.numel() < 1:
            raise ValueError("empty signal")
        with self._lock, torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            mb = min(self.max_batch, B)
            feats, fb_ws, ec_ws = self._workspace(mb, n)
            T = FbankPlan.num_frames(n)
            W = C.byref(self.weights.struct)
            for lo in range(0, B, mb):
                nb = min(mb, B - lo)
                N.check(self._lib.sd_fbank_windows_f32(self.plan.handle, signal.data_ptr(), signal.numel(), starts[lo:lo + nb].data_ptr(),
                                                       nb, n, 1, feats.data_ptr(), self.weights.cfg.input_size,
                                                       fb_ws.data_ptr(), fb_ws.numel(), stream), "sd_fbank_windows_f32")
                N.check(self._forward(W, feats.data_ptr(), nb, T, out[lo:lo + nb].data_ptr(),
                                      ec_ws.data_ptr(), ec_ws.numel(), stream), f"sd_ecapa_forward_{self.precision}")
        return out

    def features(self, wav: torch.Tensor) -> torch.Tensor:
        """The mean-normalised fbank the network consumes, [B, T, n_mels] (diagnostics / tests)."""
        wav = wav.contiguous().float()
        B, n = wav.shape
        return fbank_device(wav, self.plan, mean_norm=True)


def fbank_windows_device(signal: torch.Tensor, starts: torch.Tensor, n: int, plan: FbankPlan, mean_norm: bool = True) -> torch.Tensor:
    """HIP fbank of the windows signal[starts[b] : starts[b] + n] of one device-resident recording -> [B, T, n_mels]."""
    if signal.dim() != 1 or signal.device.type != "cuda":
        raise RuntimeError("fbank_windows_device needs a 1-d GPU tensor; there is no CPU fallback")
    lib = N.load()
    signal = signal.contiguous().float()
    starts = starts.to(signal.device, dtype=torch.int64).contiguous()
    B = int(starts.numel())
    T = plan.frames(int(n))
    out = torch.empty((B, T, plan.n_mels), dtype=torch.float32, device=signal.device)
    if B == 0:
        return out
    with torch.cuda.device(signal.device):
        ws = torch.empty((max(plan.workspace_bytes(B, n), 256),), dtype=torch.uint8, device=signal.device)
        stream = C.c_void_p(torch.cuda.current_stream(signal.device).cuda_stream)
        N.check(lib.sd_fbank_windows_f32(plan.handle, signal.data_ptr(), signal.numel(), starts.data_ptr(), B, int(n), int(bool(mean_norm)),
                                         out.data_ptr(), plan.n_mels, ws.data_ptr(), ws.numel(), stream), "sd_fbank_windows_f32")
    return out


def fbank_device(wav: torch.Tensor, plan: FbankPlan, mean_norm: bool = True) -> torch.Tensor:
    """Run the HIP fbank on a device tensor [B, n] -> [B, T, n_mels]."""
    if wav.dim() != 2:
        raise AssertionError("wav must be [B, n]")
    if wav.device.type != "cuda":
        raise RuntimeError("fbank_device needs a GPU tensor; there is no CPU fallback")
    lib = N.load()
    wav = wav.contiguous().float()
    B, n = wav.shape
    T = plan.frames(n)
    out = torch.empty((B, T, plan.n_mels), dtype=torch.float32, device=wav.device)
    if B == 0:
        return out
    with torch.cuda.device(wav.device):
        ws = torch.empty((max(plan.workspace_bytes(B, n), 256),), dtype=torch.uint8, device=wav.device)
        stream = C.c_void_p(torch.cuda.current_stream(wav.device).cuda_stream)
        N.check(lib.sd_fbank_f32(plan.handle, wav.data_ptr(), B, n, int(bool(mean_norm)), out.data_ptr(), plan.n_mels,
                                 ws.data_ptr(), ws.numel(), stream), "sd_fbank_f32")
    return out
