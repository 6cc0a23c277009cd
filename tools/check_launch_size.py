#!/usr/bin/env python3
"""bench.py runs a step as ONE forward of 10 000 segments since round 3: the same embeddings as two forwards of 5 000 up to the
summation order of the SE means (tile boundaries move against segment boundaries): measured max cosine distance 2.4e-12 (f32),
2.4e-12 (f32-split16x3), 1.4e-9 (f16) over 10 000 segments.

    python tools/check_launch_size.py
"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from speech_diarization_amd import synth
from speech_diarization_amd.engine import EmbeddingEngine
dev = torch.device("cuda", 0)
sd = synth.make_ecapa_state_dict(1234)
wav = synth.synthetic_segments_device(0, 10000, 32000, dev, std=0.1)
for p in ("f32", "f32s", "f16"):
    a = EmbeddingEngine(sd, dev, max_batch=10000, precision=p).embed(wav)
    torch.cuda.empty_cache()
    b = EmbeddingEngine(sd, dev, max_batch=5000, precision=p).embed(wav)
    cd = 1.0 - torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)
    print(p, "finite", bool(torch.isfinite(a).all()), "max cos dist 10000-vs-2x5000", float(cd.max()), "bitwise", bool(torch.equal(a, b)), "last rows equal", bool(torch.equal(a[-100:], b[-100:])))
    del a, b
    torch.cuda.empty_cache()
