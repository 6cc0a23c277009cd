#!/usr/bin/env python3
"""Diagnostic (a --variant built with -DSD_STAMP): where MFMA wave 0 of each res2net_chain_f16_kernel workgroup spends its cycles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import ops, _native
B = int(os.environ.get("SEGS", "4096")); T = int(os.environ.get("T", "201")); dil = 2
dev = torch.device("cuda", 0)
r = (torch.randn(B * T, 1024, device=dev) * 0.7).half()
layers = [dict(w=ops.pack_weight(torch.randn(128, 128, 3) / 384 ** 0.5, dev, torch.float16), bias=torch.randn(128, device=dev) * 0.1,
               scale=torch.rand(128, device=dev) + 0.5, shift=torch.randn(128, device=dev) * 0.1, dil=dil) for _ in range(7)]
for _ in range(int(os.environ.get("REPS", "5"))):
    ops.res2net_chain(r, T, layers)
torch.cuda.synchronize()
lib = _native.load(); n = 4096 * 32; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_res2_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_res2_stamps(buf, n) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.float64)[:min(B, 4096)]
st = raw[:, 20:26]
t0 = raw[:, 11:12]
arr = raw[:, :10] - t0
print("  conv 3: arrival at barrier A relative to wave 0 entering the K loop, median cycles: " + " ".join(f"w{i}={np.median(arr[:, i]):.0f}" for i in range(10)))
print(f"  DMA wave after its vmcnt(0): {np.median(raw[:, 18] - raw[:, 11]):.0f}; barrier A released (wave 0): {np.median(raw[:, 10] - raw[:, 11]):.0f}")
names = ["K loops (7)", "barrier A (7)", "epilogues (7)", "barrier B (7)", "prologue wait", "total"]
print(f"B={B} T={T}: MFMA wave 0, median cycles per workgroup: " + "  ".join(f"{nm} {np.median(st[:, i]):.0f}" for i, nm in enumerate(names)))
print("  (matrix pipe needs 7 convs x 24 steps x 7 tiles x 32 cycles = 37632 cycles per SIMD at one MFMA per 32 cycles)")
