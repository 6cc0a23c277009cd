#!/usr/bin/env python3
"""The reference's batch loops (32-segment batches of embed_segments, 128-window batches of frame_reassign) one call at a time
against `ecapa_encode_batches` with 1 / 2 / 3 batches in flight; checks the results bit for bit.
    python tools/time_batches.py [--precision f32] [--seconds 2.0]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_diarization_amd import speech_encode, synth

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f32")
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--ragged", action="store_true", help="batches of different padded lengths (1.0 .. 3.0 s), as embed_segments produces them")
a = ap.parse_args()
speech_encode.set_precision(a.precision)
enc = speech_encode.using_ecapa_encoder()
n = int(a.seconds * 16000)
for batch, count in ((32, 64), (128, 24), (16, 64)):
    rng = np.random.default_rng(batch)
    lens = [int(16000 * rng.uniform(1.0, 3.0)) // 160 * 160 for _ in range(count)] if a.ragged else [n] * count
    batches = [synth.synthetic_segments(100 + i, batch, m) for i, m in enumerate(lens)]
    for _ in range(2):
        ref = [speech_encode.ecapa_encode_batch(b) for b in batches]
    t0 = time.perf_counter()
    ref = [speech_encode.ecapa_encode_batch(b) for b in batches]
    t_seq = time.perf_counter() - t0
    line = f"{a.precision} batch {batch} x {count}{' ragged' if a.ragged else ''}: one at a time {batch * count / t_seq:8.0f} seg/s ({t_seq / count * 1e3:.3f} ms per batch)"
    for lanes in (1, 2, 3):
        enc.encode_batches(batches, lanes=lanes)
        t0 = time.perf_counter()
        got = enc.encode_batches(batches, lanes=lanes)
        dt = time.perf_counter() - t0
        same = all(np.array_equal(g, r) for g, r in zip(got, ref))
        line += f" | {lanes} in flight {batch * count / dt:8.0f} ({dt / count * 1e3:.3f} ms){'' if same else ' RESULTS DIFFER'}"
    print(line, flush=True)
