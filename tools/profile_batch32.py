#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the reference's own call, `ecapa_encode_batch(numpy [32, 32000])`, 200 times.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b32 -o b32 -- python3 tools/profile_batch32.py [f32|f32s|f16] [batch]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from speech_diarization_amd import speech_encode, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
speech_encode.set_precision(prec)
wavs = synth.synthetic_segments(5, batch, 32000)
for _ in range(100):
    speech_encode.ecapa_encode_batch(wavs)
t0 = time.perf_counter()
for _ in range(200):
    speech_encode.ecapa_encode_batch(wavs)
dt = (time.perf_counter() - t0) / 200
print(f"{prec} batch {batch}: {dt * 1e3:.3f} ms per call, {batch / dt:.0f} segments/s")
