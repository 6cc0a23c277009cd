#!/usr/bin/env python3
"""Diagnostic (build_native.py --stamp or a --variant built with -DSD_STAMP): where wave 0 of each
conv_gemm_f16_t256_kernel workgroup spends its cycles (prologue / K loop / epilogue) and the clock the chip holds
inside the kernel (s_memtime cycles per s_memrealtime 100 MHz tick).  Never time such a build.

    SD_HIP_LIB=speech-diarization_amd/variants/libsd_hip_stamp.so REPS=60 SEGS=5000 python tools/stamp_t256.py 3072 3072 [split]
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SD_EXPERIMENT"] = "1"; os.environ["SD_F16_KERNEL"] = "t256"
REPS = int(os.environ.get("REPS", "3"))
import numpy as np, torch
from speech_diarization_amd import ops, _native
cin, cout = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0); T = 201; M = int(os.environ.get("SEGS", "1024")) * T
split = len(sys.argv) > 3 and sys.argv[3] == "split"
w = torch.randn(cout, cin, 1) / cin ** 0.5
bias = torch.randn(cout, device=dev); scale = torch.rand(cout, device=dev) + 0.5; shift = torch.randn(cout, device=dev)
if split:        # the f32-split16x3 form of the same kernel (f32 output: twice the epilogue bytes, 1.5x the MFMAs per step)
    x = torch.randn(M, cin, device=dev); ws, sh = ops.pack_weight_split16(w, dev); out = torch.empty(M, cout, device=dev)
    for _ in range(REPS):
        ops.conv1d_cl_split16(x, ws, sh, T, cin=cin, bias=bias, act="relu", scale=scale, shift=shift, out=out)
else:
    x = torch.randn(M, cin, device=dev).half()
    wp = ops.pack_weight(w, dev, torch.float16); out = torch.empty(M, cout, device=dev, dtype=torch.float16)
    for _ in range(REPS):
        ops.conv1d_cl(x, wp, T, cin=cin, bias=bias, act="relu", scale=scale, shift=shift, out=out)
torch.cuda.synchronize()
lib = _native.load(); n = 8192 * 8; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
nb = min(8192, ((M + 255) // 256) * ((cout + 255) // 256)); st = st[:nb]
nk = cin // 32 if split else cin // 64
pro, loop, epi, tot, real = (np.median(st[:, i]) for i in range(5))
clk = st[:, 3] / np.maximum(st[:, 4], 1) * 100.0
print(f"cin={cin} cout={cout} blocks={nb} ksteps={nk}: per workgroup (wave 0, median): prologue {pro:.0f}  K loop {loop:.0f} "
      f"({loop / nk:.0f} per step; the matrix pipe needs {3072 if split else 2048})  epilogue {epi:.0f}  total {tot:.0f} cycles")
print(f"  in-kernel clock: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f})")
