"""ctypes binding of libsd_hip.so (C ABI: include/sd_hip.h).

The HIP library is the product path.  There is no CPU fallback: if the shared
object is missing, or exports fewer symbols than the header declares, loading
raises instead of degrading silently.
"""
from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libsd_hip.so"

SD_OK = 0
SD_PAD_ZERO, SD_PAD_REFLECT = 0, 1
SD_LOG_LN_EPS, SD_LOG_DB_TOPDB = 0, 1
SD_ACT_NONE, SD_ACT_RELU, SD_ACT_TANH, SD_ACT_SIGMOID = 0, 1, 2, 3
SD_DT_F32, SD_DT_F16, SD_DT_SPLIT16 = 0, 1, 2
SD_TUNE_SKINNY_TILES = 1
SD_TUNE_WIDE_TILES = 2
SD_TUNE_F16_NARROW_TILES = 3
SD_TUNE_S64_TILES = 4
SD_TUNE_HALF_TILES = 5
SD_TUNE_TILE_ROWS = 6
SD_TUNE_T256_LOCKSTEP_TILES = 7
SD_MAX_RES2 = 15
SD_MAX_BLOCKS = 8
SD_ABI_VERSION = 10
SD_PROF_CONV_GEMM, SD_PROF_FBANK, SD_PROF_CONV_WIDE, SD_PROF_SEG_SPLITK = 0, 1, 2, 3


def profile_enable(on: bool) -> None:
    check(load().sd_profile_enable(int(on)), "sd_profile_enable")


def profile_read(kind: int):
    """(total_ms, launches, work) of one kernel family since profile_enable(True)."""
    ms, n, w = C.c_double(), C.c_longlong(), C.c_double()
    check(load().sd_profile_read(kind, C.byref(ms), C.byref(n), C.byref(w)), "sd_profile_read")
    return ms.value, n.value, w.value


class SdError(RuntimeError):
    """A libsd_hip.so entry point returned a non-zero status."""


class sd_conv_args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("lda", C.c_int), ("a_col0", C.c_int),
        ("w", C.c_void_p), ("w_dtype", C.c_int),
        ("y", C.c_void_p), ("ldo", C.c_int), ("o_col0", C.c_int),
        ("M", C.c_int), ("T", C.c_int),
        ("cin", C.c_int), ("cin_pad", C.c_int), ("cout", C.c_int), ("taps", C.c_int), ("dil", C.c_int),
        ("bias", C.c_void_p), ("bias_per_seg", C.c_int),
        ("act", C.c_int),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("act2", C.c_int),
        ("tee", C.c_void_p), ("ldt", C.c_int), ("tee_lo", C.c_int), ("tee_hi", C.c_int),
        ("tee_add", C.c_void_p), ("ld_ta", C.c_int), ("ta_col0", C.c_int),
        ("x_dtype", C.c_int), ("y_dtype", C.c_int),
        ("colstat", C.c_void_p),
        ("w_scale_inv", C.c_float),
    ]


class sd_layer(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("cin", C.c_int), ("cin_pad", C.c_int), ("cout", C.c_int), ("taps", C.c_int), ("dil", C.c_int),
        ("w_dtype", C.c_int),
        ("w_split", C.c_void_p), ("bias_split", C.c_void_p), ("scale_split", C.c_void_p), ("split_scale_inv", C.c_float),
    ]


class sd_se_res2_block(C.Structure):
    _fields_ = [
        ("tdnn1", sd_layer),
        ("res2", sd_layer * SD_MAX_RES2),
        ("tdnn2", sd_layer),
        ("se1", sd_layer), ("se2", sd_layer),
    ]


class sd_ecapa_weights(C.Structure):
    _fields_ = [
        ("w_dtype", C.c_int), ("n_mels", C.c_int), ("channels", C.c_int), ("n_blocks", C.c_int),
        ("res2_scale", C.c_int), ("mfa_channels", C.c_int), ("att_channels", C.c_int), ("emb_dim", C.c_int),
        ("asp_eps", C.c_float),
        ("split16", C.c_int),
        ("block0", sd_layer),
        ("blocks", sd_se_res2_block * SD_MAX_BLOCKS),
        ("mfa", sd_layer), ("asp_tdnn_h", sd_layer), ("asp_tdnn_g", sd_layer), ("asp_conv", sd_layer), ("fc", sd_layer),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t

# symbol -> (restype, argtypes); must list every function sd_hip.h declares
PROTOTYPES = {
    "sd_abi_version": (_I, []),
    "sd_sizeof": (_Z, [_I]),
    "sd_last_error": (C.c_char_p, []),
    "sd_device_count": (_I, []),
    "sd_profile_enable": (_I, [_I]),
    "sd_profile_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "sd_fbank_plan_create": (_P, [_P, _I, _I, _P, _I, _I, _I, _F, _F]),
    "sd_fbank_plan_destroy": (None, [_P]),
    "sd_fbank_num_frames": (_I, [_P, _I]),
    "sd_fbank_workspace_bytes": (_Z, [_P, _I, _I]),
    "sd_fbank_f32": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "sd_fbank_windows_f32": (_I, [_P, _P, C.c_longlong, _P, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "sd_conv1d_cl_f32": (_I, [C.POINTER(sd_conv_args), _P]),
    "sd_seg_gemm_f32": (_I, [C.POINTER(sd_conv_args), _P, _Z, _P]),
    "sd_seg_gemm_scratch_bytes": (_Z, [_I, _I, _I]),
    "sd_conv1d_cl_f16": (_I, [C.POINTER(sd_conv_args), _P]),
    "sd_conv1d_cl_split16": (_I, [C.POINTER(sd_conv_args), _P]),
    "sd_split16_pack_f32": (_I, [_P, _I, _I, _I, _I, _F, _P, _I, _P]),
    "sd_set_tuning": (_I, [_I, C.c_long]),
    "sd_colstat_floats": (_Z, [_I, _I]),
    "sd_colstat_finish_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "sd_seg_mean_std_dt": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "sd_se_scale_residual_dt": (_I, [_P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sd_asp_pool_dt": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "sd_asp_attend_pool_supported": (_I, [_I, _I, _I, _I]),
    "sd_asp_attend_pool_dt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "sd_res2net_chain_supported": (_I, [_I, _I, _I, _I, _I]),
    "sd_res2net_chain_workspace_bytes": (C.c_size_t, [_I]),
    "sd_res2net_chain_f16": (_I, [_P, _I, _I, _I, C.POINTER(sd_layer), _I, _P, C.c_size_t, _P]),
    "sd_ecapa_forward_f16": (_I, [C.POINTER(sd_ecapa_weights), _P, _I, _I, _P, _P, _Z, _P]),
    "sd_seg_mean_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sd_seg_mean_std_f32": (_I, [_P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "sd_se_scale_residual_f32": (_I, [_P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "sd_asp_pool_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    "sd_ecapa_workspace_bytes": (_Z, [C.POINTER(sd_ecapa_weights), _I, _I]),
    "sd_ecapa_forward_f32": (_I, [C.POINTER(sd_ecapa_weights), _P, _I, _I, _P, _P, _Z, _P]),
    "sd_l2norm_rows_f32": (_I, [_P, _I, _I, _I, _F, _I, _P, _I, _P]),
    "sd_cosine_workspace_bytes": (_Z, [_I, _I]),
    "sd_cosine_affinity_f32": (_I, [_P, _I, _I, _P, _I, _P, _Z, _P]),
    "sd_cosine_affinity_rows_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "sd_cosine_split16_workspace_bytes": (_Z, [_I, _I]),
    "sd_cosine_affinity_rows_split16": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "sd_adjacent_cosine_f32": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "sd_sim_argmax_f32": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "sd_topk_mean_std_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "sd_asnorm_combine_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "sd_viterbi_workspace_bytes": (_Z, [_I, _I]),
    "sd_viterbi_f32": (_I, [_P, _I, _I, _I, _F, _F, _P, _Z, _P, _P]),
}

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libsd_hip.so once per process; raises if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import os
        lib_path = LIB_PATH
        if os.environ.get("SD_HIP_LIB") and os.environ.get("SD_EXPERIMENT") == "1":   # A/B builds (build_native.py --variant); never in product runs
            lib_path = Path(os.environ["SD_HIP_LIB"])
            if not lib_path.exists():
                raise RuntimeError(f"SD_HIP_LIB={lib_path} does not exist")
        if not lib_path.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP hot path is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is deliberately no CPU fallback."
            )
        lib = C.CDLL(str(lib_path))
        missing = [name for name in PROTOTYPES if not hasattr(lib, name)]
        if missing:
            raise RuntimeError(f"{lib_path} lacks symbols declared in include/sd_hip.h: {missing}")
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.sd_abi_version() != SD_ABI_VERSION:
            raise RuntimeError(f"libsd_hip.so ABI {lib.sd_abi_version()} != binding ABI {SD_ABI_VERSION}; rebuild")
        for which, st in enumerate((sd_conv_args, sd_layer, sd_se_res2_block, sd_ecapa_weights)):
            if lib.sd_sizeof(which) != C.sizeof(st):
                raise RuntimeError(f"{st.__name__}: binding layout is {C.sizeof(st)} bytes, the library's {lib.sd_sizeof(which)}; rebuild")
        _lib = lib
    return _lib


def last_error() -> str:
    msg = load().sd_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, what: str) -> None:
    if status != SD_OK:
        raise SdError(f"{what} failed (status {status}): {last_error()}")
