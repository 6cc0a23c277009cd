#!/usr/bin/env python3
"""Where a call of the reference-shaped API (numpy in -> numpy out) spends its time, per batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import speech_encode, synth

def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

enc = speech_encode.using_ecapa_encoder()
dev = enc.device
for batch, n in ((32, 32000), (128, 16000), (128, 32000)):
    wavs = synth.synthetic_segments(5, batch, n)
    x_dev = torch.from_numpy(wavs).to(dev)
    pinned = torch.empty((batch, n), dtype=torch.float32).pin_memory()
    api = t(lambda: speech_encode.ecapa_encode_batch(wavs))
    h2d = t(lambda: torch.from_numpy(wavs).to(dev))
    h2d_pin = t(lambda: (pinned.copy_(torch.from_numpy(wavs)), pinned.to(dev, non_blocking=True)))
    comp = t(lambda: enc.engine.embed(x_dev))
    def comp_sync():
        enc.engine.embed(x_dev).cpu()
    comp_d2h = t(comp_sync)
    print(f"batch {batch} n {n}: api {api:.2f} ms ({batch / api * 1e3:.0f} seg/s) | pageable H2D {h2d:.2f} | via pinned {h2d_pin:.2f} | resident embed {comp:.2f} "
          f"({batch / comp * 1e3:.0f} seg/s) | embed + D2H + sync {comp_d2h:.2f}", flush=True)
