#!/usr/bin/env python3
"""Run the resident embedding forward in a loop at one batch size (for rocprofv3 kernel traces of small batches).
    python tools/loop_embed.py --batch 128 --n 32000 --reps 50 [--precision f16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import synth
from speech_diarization_amd.engine import EmbeddingEngine
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128); ap.add_argument("--n", type=int, default=32000)
ap.add_argument("--reps", type=int, default=50); ap.add_argument("--precision", default="f32")
ap.add_argument("--skinny", type=int, default=-1, help="SD_TUNE_SKINNY_TILES override")
ap.add_argument("--wide", type=int, default=-1, help="SD_TUNE_WIDE_TILES override (256x256 tiles from which the f32 ring kernel takes the wide layers)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.skinny >= 0:
    from speech_diarization_amd import _native
    _native.check(_native.load().sd_set_tuning(_native.SD_TUNE_SKINNY_TILES, a.skinny), "sd_set_tuning")
if a.wide >= 0:
    from speech_diarization_amd import _native
    _native.check(_native.load().sd_set_tuning(_native.SD_TUNE_WIDE_TILES, a.wide), "sd_set_tuning")
eng = EmbeddingEngine(synth.make_ecapa_state_dict(1234), dev, max_batch=max(a.batch, 16), precision=a.precision)
x = torch.from_numpy(synth.synthetic_segments(5, a.batch, a.n)).to(dev)
for _ in range(20): eng.embed(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.reps): eng.embed(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
print(f"batch {a.batch} n {a.n} {a.precision}: {dt * 1e3:.3f} ms per forward, {a.batch / dt:.0f} segments/s")
