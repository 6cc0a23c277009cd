#!/usr/bin/env python3
"""Diagnostic (build_native.py --stamp): cycle shares of fbank_logmel_kernel, wave 0 of each workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import _native
from speech_diarization_amd.engine import fbank_device
from speech_diarization_amd.features import FbankPlan
dev = torch.device("cuda", 0)
wav = torch.randn(2048, 32000, device=dev) * 0.1
plan = FbankPlan("speechbrain")
for _ in range(3):
    fbank_device(wav, plan)
torch.cuda.synchronize()
lib = _native.load(); n = 8192 * 8; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_fbank_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_fbank_stamps(buf, n) == 0
nb = min(8192, (2048 * 201 + 127) // 128)
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)[:nb, :6]
tot = st.sum(1)
for i, nm in enumerate(["bookkeeping + sample staging", "operand reads + MFMA", "basis stage + barrier", "|X|^2 -> mel", "log + max", "store"]):
    print(f"  {nm:30s} {np.median(st[:, i]):9.0f} cycles ({np.median(st[:, i] / tot) * 100:5.1f} %)")
print(f"  total {np.median(tot):9.0f} cycles per workgroup (128 frames); ideal MFMA time {7 * 6 * 34 * 64} cycles")
