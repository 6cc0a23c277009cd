#!/usr/bin/env python3
"""Diagnostic (variant build with -DSD_STAMP, SD_EXPERIMENT=1 SD_HIP_LIB=<variant>): where the time of a ring-kernel launch goes.
One Res2Net-shaped conv (128 -> 128, k = 3, tee + tee_add) at SEGS segments; median cycles of wave 0 per workgroup.

    SEGS=32 SD_EXPERIMENT=1 SD_HIP_LIB=speech-diarization_amd/variants/libsd_hip_stamp.so python tools/stamp_s64.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import ops, _native
dev = torch.device("cuda", 0); T = 201; B = int(os.environ.get("SEGS", "32")); M = B * T
cin = cout = 128
x = torch.randn(M, cin, device=dev); w = torch.randn(cout, cin, 3) / (cin * 3) ** 0.5
wp = ops.pack_weight(w, dev); out = torch.empty(M, 1024, device=dev); tee = torch.empty(M, cout, device=dev)
kw = dict(cin=cin, dil=2, act="relu", bias=torch.randn(cout, device=dev), scale=torch.rand(cout, device=dev) + 0.5, shift=torch.randn(cout, device=dev),
          out=out, o_col0=128, tee=tee, tee_lo=0, tee_hi=cout, tee_add=out, ta_col0=256)
for _ in range(5):
    ops.conv1d_cl(x, wp, T, **kw)
torch.cuda.synchronize()
lib = _native.load(); n = 8192 * 10; buf = (C.c_ulonglong * n)()
lib.sd_debug_read_c32_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_c32_stamps(buf, n) == 0
g = -(-M // (32 if -(-M // 64) * 2 < 128 else 64)) * 2
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 10).astype(np.float64)[:g]
names = ["arguments + addresses + first requests", "first stage landed", "K loop (12 steps)", "accumulators -> C tile", "params + tee_add + store issue", "stores retired"]
print(f"{B} segments, {g} workgroups; s_memtime cycles (shader clock, ~2.1-2.4 GHz), median over workgroups")
for i, nm in enumerate(names):
    print(f"  {nm:40s} {np.median(st[:, i]):8.0f}")
print(f"  total                                    {np.median(st[:, :6].sum(1)):8.0f}")
t0 = st[:, 9]
print(f"  workgroup start spread: {t0.max() - t0.min():.0f} cycles; last end - first start: {(t0 + st[:, :6].sum(1)).max() - t0.min():.0f} cycles")
