"""The ctypes stub of INTEGRATION.md section 3, executed: a maintainer of the reference binding libsd_hip.so
directly (no package imports beyond locating the library and the window / mel tables) gets the same numbers as
the package's own operators and as the float64 oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_direct_ctypes_binding_matches_package_and_oracle(dev):
    from oracle import fbank_ref
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import features
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "speech-diarization_amd", "libsd_hip.so"))
    lib.sd_last_error.restype = C.c_char_p
    lib.sd_fbank_plan_create.restype = C.c_void_p
    lib.sd_fbank_plan_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    lib.sd_fbank_plan_destroy.argtypes = [C.c_void_p]
    lib.sd_fbank_workspace_bytes.restype = C.c_size_t
    lib.sd_fbank_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.sd_fbank_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sd_cosine_affinity_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sd_cosine_workspace_bytes.restype = C.c_size_t
    lib.sd_cosine_workspace_bytes.argtypes = [C.c_int, C.c_int]

    # replaces MelSpectrogram(...).to('cuda') + log + mean-norm   [REF speech_encode.py:17-36]
    window = np.ascontiguousarray(features.periodic_window("hann"), dtype=np.float32)
    mel_fb = np.ascontiguousarray(features.mel_filters_torchaudio(), dtype=np.float32)
    fe = features.FRONT_ENDS["torchaudio"]
    plan = lib.sd_fbank_plan_create(window.ctypes.data, 400, 160, mel_fb.ctypes.data, 80, fe.pad_mode, fe.log_mode,
                                    C.c_float(fe.log_eps), C.c_float(fe.top_db))
    assert plan, lib.sd_last_error()
    rng = np.random.default_rng(0)
    wavs = (0.1 * rng.standard_normal((5, 24000))).astype(np.float32)
    wav = torch.from_numpy(wavs).to(dev)
    T = 1 + wav.shape[1] // 160
    out = torch.empty(wav.shape[0], T, 80, device=dev)
    wsz = lib.sd_fbank_workspace_bytes(plan, wav.shape[0], wav.shape[1])
    ws = torch.empty(max(256, wsz), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.sd_fbank_f32(plan, wav.data_ptr(), wav.shape[0], wav.shape[1], 1, out.data_ptr(), 80, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, lib.sd_last_error()
    torch.cuda.synchronize()
    lib.sd_fbank_plan_destroy(plan)
    assert np.abs(out.cpu().numpy() - fbank_ref.fbank_batch_ref(wavs.astype(np.float64))).max() < 5e-4

    # replaces sklearn cosine_similarity(embs)                   [REF anti_stick_diarize.py:177]
    emb = torch.from_numpy(rng.standard_normal((257, 192)).astype(np.float32)).to(dev)
    n, d = emb.shape
    K = torch.empty(n, n, device=dev)
    wsz = lib.sd_cosine_workspace_bytes(n, d)
    ws2 = torch.empty(wsz, dtype=torch.uint8, device=dev)
    assert lib.sd_cosine_affinity_f32(emb.data_ptr(), n, d, K.data_ptr(), n, ws2.data_ptr(), wsz, stream) == 0, lib.sd_last_error()
    torch.cuda.synchronize()
    assert np.abs(K.cpu().numpy() - cosine_similarity(emb.cpu().numpy().astype(np.float64))).max() < 2e-6
