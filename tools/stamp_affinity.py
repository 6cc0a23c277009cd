#!/usr/bin/env python3
"""Diagnostic (a --variant built with -DSD_STAMP): per-workgroup cycles of the split16x3 affinity's 128 x 128 tiles
(affinity_sym_kernel, sd_affinity.hip): prologue / K loop / store issue / store drain, and the in-kernel clock.

    python speech-diarization_amd/build_native.py --variant stamp "-DSD_STAMP"
    SD_EXPERIMENT=1 SD_HIP_LIB=speech-diarization_amd/variants/libsd_hip_stamp.so python tools/stamp_affinity.py [N]
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import ops, _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda", 0)
x = torch.randn(n, 192, device=dev); K = torch.empty(n, n, device=dev)
for _ in range(5):
    ops.cosine_affinity(x, out=K, split16=True)
torch.cuda.synchronize()
lib = _native.load(); m = 8192 * 8; buf = (C.c_ulonglong * m)()
lib.sd_debug_read_affinity_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_affinity_stamps(buf, m) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
st = st[(st[:, 4] > 0) & (st[:, 6] == 1)]                      # stamped off-diagonal tiles
pro, loop, issue, drain, tot = (np.median(st[:, i]) for i in range(5))
clk = st[:, 4] / np.maximum(st[:, 5], 1) * 100.0
print(f"N={n}: {len(st)} stamped off-diagonal tiles (every 16th workgroup), wave 0, medians: prologue {pro:.0f}  K loop {loop:.0f} (6 steps; the wave's "
      f"MFMAs are 6 x 768)  epilogue shuffles + store issue {issue:.0f}  drain {drain:.0f}  total {tot:.0f} cycles = {tot / np.median(clk):.1f} us at {np.median(clk):.0f} MHz; "
      f"p10 / p90 of the total {np.percentile(st[:, 4], 10):.0f} / {np.percentile(st[:, 4], 90):.0f}")
