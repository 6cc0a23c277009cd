#!/usr/bin/env python3
"""Time the fbank kernel pair on B synthetic segments (HIP events, best of rounds) and report the algorithmic HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import synth
from speech_diarization_amd.engine import fbank_device
from speech_diarization_amd.features import FbankPlan
B = int(os.environ.get("SEGS", "5000")); n = 32000
dev = torch.device("cuda", 0)
wav = synth.synthetic_segments_device(0, B, n, dev)
for kind in ("speechbrain", "torchaudio"):
    plan = FbankPlan(kind, n_mels=80)
    fbank_device(wav, plan); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fbank_device(wav, plan); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print(f"{kind}: {B} segments: best {t:.3f} ms (incl. finalize + output alloc), {B * 192320 / t / 1e6:.0f} GB/s algorithmic")
