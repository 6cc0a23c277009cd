#!/usr/bin/env python3
"""Time the conv operator on the shapes a small launch (the reference's batches of 16 / 32 / 128 segments) runs on the 32x32 split-K
kernel: the narrow Res2Net conv and the per-segment layers.  HIP events around 50 back-to-back launches.

    python tools/probe_small_shapes.py [B ...]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops, _native as N

dev = torch.device("cuda", 0)
lib = N.load()


def time_conv(M, T, cin, cout, taps, dil, reps=50):
    x = torch.randn(M, cin, device=dev)
    w = torch.randn(cout, cin, taps) / (cin * taps) ** 0.5
    wp = ops.pack_weight(w, dev)
    bias = torch.randn(cout, device=dev)
    out = torch.empty(M, cout, device=dev)
    for _ in range(5):
        ops.conv1d_cl(x, wp, T, cin=cin, dil=dil, bias=bias, act="relu", out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv1d_cl(x, wp, T, cin=cin, dil=dil, bias=bias, act="relu", out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in [int(v) for v in sys.argv[1:]] or [16, 32, 128]:
    shapes = [("res2net 128->128 k3", B * 201, 201, 128, 128, 3, 2), ("se fc1 1024->128", B, 1, 1024, 128, 1, 1), ("se fc2 128->1024", B, 1, 128, 1024, 1, 1),
              ("gbias 6144->128", B, 1, 6144, 128, 1, 1), ("fc 6144->192", B, 1, 6144, 192, 1, 1), ("att tdnn 3072->128", B * 201, 201, 3072, 128, 1, 1)]
    for name, M, T, cin, cout, taps, dil in shapes:
        res = []
        for label, s64, val in (("default", -1, -1), ("64x64 ring", 1 << 30, -1), ("32x32 split-K", 0, -1), ("128x128", 0, 0)):
            N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, s64), "tune")
            N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, val), "tune")
            res.append(f"{label} {time_conv(M, T, cin, cout, taps, dil):7.1f} us")
        N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, -1), "tune")
        N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, -1), "tune")
        print(f"B={B:4d} {name:22s} M={M:6d}: " + "   ".join(res), flush=True)
