#!/usr/bin/env python3
"""How long does a 128x128 tile of a C -> C layer take when it is alone on its CU, and when two share the CU?  1024 -> 1024 pointwise conv
over M = 128 x (tiles per column) rows, 8 column tiles: 32 row tiles = 256 tiles (one per CU), 64 = 512 (two per CU), 51 = 408 (the
32-segment launch), 26 = 208 (the 16-segment hop).  HIP events around 50 launches.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops, _native as N

dev = torch.device("cuda", 0)
lib = N.load()
N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, 0), "tune")
N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, 0), "tune")
for cin, cout in ((1024, 1024), (3072, 3072)):
    w = torch.randn(cout, cin, 1) / cin ** 0.5
    wp = ops.pack_weight(w, dev)
    bias = torch.randn(cout, device=dev)
    for rows in (16, 26, 32, 51, 64, 96, 128):
        M = rows * 128
        x = torch.randn(M, cin, device=dev)
        out = torch.empty(M, cout, device=dev)
        for _ in range(5):
            ops.conv1d_cl(x, wp, 128, cin=cin, bias=bias, act="relu", out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.conv1d_cl(x, wp, 128, cin=cin, bias=bias, act="relu", out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tiles = rows * (cout // 128)
        print(f"{cin}->{cout}  {rows:4d} row tiles = {tiles:5d} tiles ({tiles / 256:5.2f} per CU): {us:8.1f} us  {2e-6 * M * cin * cout / us:7.1f} TFLOP/s", flush=True)
