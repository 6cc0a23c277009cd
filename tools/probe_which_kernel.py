#!/usr/bin/env python3
"""Under rocprofv3 --kernel-trace: one conv launch per shape, to see which kernel the f32 operator picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops
dev = torch.device("cuda", 0)
for (B, cin, cout, taps) in ((16, 1024, 1024, 1), (16, 80, 1024, 5), (32, 1024, 1024, 1)):
    M = B * 201
    x = torch.randn(M, cin, device=dev)
    wp = ops.pack_weight(torch.randn(cout, cin, taps) / 30, dev)
    for _ in range(3):
        ops.conv1d_cl(x, wp, 201, cin=cin, act="relu")
torch.cuda.synchronize()
