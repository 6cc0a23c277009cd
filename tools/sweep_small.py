#!/usr/bin/env python3
"""Resident throughput at small batch sizes (the reference's caller batches: 16 streaming, 32, 128).
One warm engine for all sizes: a fresh engine per size measures the clock ramp after the idle time spent
packing weights, not the kernels.

    python tools/sweep_small.py f16|f32 [n_samples] [B,B,...]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import synth
from speech_diarization_amd.engine import EmbeddingEngine
dev = torch.device("cuda", 0); sd = synth.make_ecapa_state_dict(1234)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
sizes = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [16, 32, 64, 128, 256]
eng = EmbeddingEngine(sd, dev, max_batch=max(sizes), precision=prec)
for B in sizes:
    wav = (torch.randn(B, n, device=dev) * 0.1).clamp_(-1, 1)
    for _ in range(10): eng.embed(wav)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eng.embed(wav)
    t1 = time.perf_counter(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{prec} n={n} B={B}: {dt * 1e3:.3f} ms per batch ({(t1 - t0) / 50 * 1e3:.3f} ms of host enqueue), {B / dt:.0f} segments/s")
