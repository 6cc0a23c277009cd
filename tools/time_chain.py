#!/usr/bin/env python3
"""Time the fused Res2Net chain at bench size (5000 segments, T = 201)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops
B = int(os.environ.get("SEGS", "5000")); T = int(os.environ.get("T", "201")); dev = torch.device("cuda", 0)
r = (torch.randn(B * T, 1024, device=dev) * 0.7).half()
layers = [dict(w=ops.pack_weight(torch.randn(128, 128, 3) / 384 ** 0.5, dev, torch.float16), bias=torch.randn(128, device=dev) * 0.1,
               scale=torch.rand(128, device=dev) * 0.5 + 0.25, shift=torch.randn(128, device=dev) * 0.1, dil=2) for _ in range(7)]
ops.res2net_chain(r, T, layers); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.res2net_chain(r, T, layers); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"res2net chain f16, {B} segments of T = {T}: best {min(ts):.3f} ms ({2.0 * B * T * 128 * 384 * 7 / min(ts) / 1e9:.0f} TFLOP/s, {B * T * 256 * 2 * 7 / min(ts) / 1e9:.2f} TB/s of chunk traffic)")
