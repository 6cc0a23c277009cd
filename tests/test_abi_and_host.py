"""No-GPU checks: the C-ABI library loads and exports every symbol include/sd_hip.h declares, the
binding table covers the header, argument validation that needs no device works, and the product
path refuses to run without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from speech_diarization_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "sd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sd_hip.h but not exported"
    assert set(names) == set(_native.PROTOTYPES), set(names) ^ set(_native.PROTOTYPES)
    assert lib.sd_abi_version() == _native.SD_ABI_VERSION


def test_struct_layouts_match_the_header_sizes():
    # plain C layout: pointers 8, ints 4; sizes are what the .so was compiled with (checked on the GPU by use)
    assert C.sizeof(_native.sd_layer) == 4 * 8 + 6 * 4 + 3 * 8 + 8          # (+ split_scale_inv and tail padding)
    lib = _native.load()
    for which, st in enumerate((_native.sd_conv_args, _native.sd_layer, _native.sd_se_res2_block, _native.sd_ecapa_weights)):
        assert lib.sd_sizeof(which) == C.sizeof(st), st.__name__          # the compiled C layout, not a guess
    assert lib.sd_sizeof(99) == 0
    assert C.sizeof(_native.sd_se_res2_block) == (2 + _native.SD_MAX_RES2 + 2) * C.sizeof(_native.sd_layer)
    assert C.sizeof(_native.sd_conv_args) % 8 == 0


def test_host_side_argument_validation_without_a_gpu():
    lib = _native.load()
    win = np.ones(400, np.float32)
    mel = np.zeros((201, 80), np.float32)
    h = lib.sd_fbank_plan_create(win.ctypes.data_as(C.c_void_p), 4, 2, mel.ctypes.data_as(C.c_void_p), 80, 0, 0, 1e-6, -1.0)
    assert not h and "8 <= n_fft" in _native.last_error()              # (any other framing is accepted: sd_fbank_generic.hip)
    h = lib.sd_fbank_plan_create(win.ctypes.data_as(C.c_void_p), 400, 401, mel.ctypes.data_as(C.c_void_p), 80, 0, 0, 1e-6, -1.0)
    assert not h and "hop <= n_fft" in _native.last_error()
    win[3] = 0.5                                       # breaks w[k] == w[400-k]
    h = lib.sd_fbank_plan_create(win.ctypes.data_as(C.c_void_p), 400, 160, mel.ctypes.data_as(C.c_void_p), 80, 0, 0, 1e-6, -1.0)
    assert not h and "symmetric" in _native.last_error()
    a = _native.sd_conv_args()
    assert lib.sd_conv1d_cl_f32(C.byref(a), None) == -1 and "null" in _native.last_error()
    assert lib.sd_cosine_workspace_bytes(10, 192) == 10 * 192 * 4 + (-(10 * 192 * 4)) % 256
    with pytest.raises(_native.SdError):
        _native.check(-1, "x")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_a_gpu():
    from speech_diarization_amd import ops, speech_encode
    from speech_diarization_amd.engine import EmbeddingEngine, fbank_device
    from speech_diarization_amd.features import FbankPlan
    from speech_diarization_amd import synth
    with pytest.raises(RuntimeError):
        speech_encode.fbank_batch(np.zeros((1, 1600), np.float32))
    with pytest.raises(RuntimeError):
        speech_encode.using_ecapa_encoder()
    with pytest.raises(AssertionError):
        speech_encode.fbank_batch(np.zeros(1600, np.float32))          # ndim check comes first, as in the reference
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EmbeddingEngine(synth.make_ecapa_state_dict(1, synth.EcapaConfig.small(64)), "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cosine_affinity(torch.zeros(4, 192))
    with pytest.raises(FileNotFoundError):
        speech_encode.eres2netv2_encode_batch(np.zeros((1, 1600), np.float32))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "speech-diarization_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_weight_packing_and_bn_folding():
    from speech_diarization_amd.engine import bn_affine, pack_conv_weight
    w = np.arange(2 * 5 * 3, dtype=np.float32).reshape(2, 5, 3)
    p = pack_conv_weight(w)
    assert p.shape == (2, 3, 32)
    assert np.array_equal(p[1, 2, :5], w[1, :, 2]) and np.all(p[:, :, 5:] == 0)
    sd = {"n.weight": np.array([2.0], np.float32), "n.bias": np.array([0.5], np.float32),
          "n.running_mean": np.array([1.0], np.float32), "n.running_var": np.array([3.0], np.float32)}
    s, t = bn_affine(sd, "n")
    x = 4.2
    assert np.isclose(x * s[0] + t[0], (x - 1.0) / np.sqrt(3.0 + 1e-5) * 2.0 + 0.5)
