#!/usr/bin/env python3
"""Diagnostic (build_native.py --stamp): where wave 0 of each conv_gemm_f16_t256_kernel workgroup spends its time:
K loop segments, prologue, epilogue (s_memtime ticks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SD_F16_KERNEL"] = "t256"
import numpy as np, torch
from speech_diarization_amd import ops, _native
cin, cout = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0); T = 201; M = 1024 * T
x = torch.randn(M, cin, device=dev).half(); w = torch.randn(cout, cin, 1) / cin ** 0.5
bias = torch.randn(cout, device=dev); scale = torch.rand(cout, device=dev) + 0.5; shift = torch.randn(cout, device=dev)
wp = ops.pack_weight(w, dev, torch.float16); out = torch.empty(M, cout, device=dev, dtype=torch.float16)
for _ in range(3):
    ops.conv1d_cl(x, wp, T, cin=cin, bias=bias, act="relu", scale=scale, shift=shift, out=out)
torch.cuda.synchronize()
lib = _native.load(); n = 8192 * 8; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
nb = min(8192, ((M + 255) // 256) * ((cout + 255) // 256)); st = st[:nb]
kb = int(os.environ.get("SD_T256_K", "64")); nk = cin // kb
tot = st[:, :4].sum(1)
print(f"cin={cin} cout={cout} blocks={nb} ksteps={nk} (K step {kb}): cycles per K step (wave 0, median over workgroups)")
for i, nm in enumerate(["DMA wait (vmcnt)", "barrier", "DMA issue", "LDS reads + MFMA"]):
    print(f"  {nm:20s} {np.median(st[:, i]) / nk:8.0f} cycles  ({np.median(st[:, i] / tot) * 100:5.1f} %)")
print(f"  K loop total         {np.median(tot) / nk:8.0f} cycles per K step (s_memtime ticks, 100 MHz)")
print(f"  per workgroup: prologue {np.median(st[:, 4]):.0f}  K loop {np.median(tot):.0f}  epilogue {np.median(st[:, 5]):.0f}  total {np.median(st[:, 6]):.0f} ticks")
t0 = st[:, 7]
print(f"  launch span {(t0.max() - t0.min() + np.median(st[:, 6])):.0f} ticks for {nb} workgroups on 256 CUs ({nb / 256:.1f} waves of workgroups)")
