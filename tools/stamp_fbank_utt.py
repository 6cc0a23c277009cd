#!/usr/bin/env python3
"""Diagnostic (a --variant built with -DSD_STAMP, selected with SD_EXPERIMENT=1 SD_HIP_LIB=...): where the waves of fbank_utt_kernel
spend their cycles, per phase (median over the first 1024 workgroups)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_diarization_amd import _native, synth
from speech_diarization_amd.engine import fbank_device
from speech_diarization_amd.features import FbankPlan
B = int(os.environ.get("SEGS", "10000")); n = int(os.environ.get("N", "32000"))
dev = torch.device("cuda", 0)
wav = synth.synthetic_segments_device(0, B, n, dev)
plan = FbankPlan("speechbrain", n_mels=80)
for _ in range(int(os.environ.get("REPS", "5"))):
    fbank_device(wav, plan)
torch.cuda.synchronize()
lib = _native.load(); cnt = 1024 * 4 * 16; buf = (C.c_ulonglong * cnt)()
lib.sd_debug_read_utt_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_utt_stamps(buf, cnt) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 16).astype(np.float64)[:min(B, 1024)]
names = {1: "loads + edges + peak", 2: "(barrier) scale + split + image", 3: "barrier", 4: "r0 stage 1 (kk 0-2)", 5: "r0 pairs (6 problems)", 6: "r0 stage 1 (kk 3-4)",
         7: "r0 barrier + pairs (3 problems)", 8: "r0 log", 9: "r1 stage 1", 10: "r1 pairs (6)", 11: "r1 stage 1", 12: "r1 pairs (3)", 13: "r1 log",
         14: "final barrier", 15: "floor + mean + store"}
for w in range(4):
    t = raw[:, w, :]
    prev = t[:, 0]
    out = []
    for i in range(1, 16):
        if w == 3 and i in (9, 10, 11, 12, 13) and n == 32000:
            continue
        d = t[:, i] - prev
        out.append(f"{names[i]} {np.median(d):.0f}")
        prev = t[:, i]
    print(f"wave {w}: total {np.median(t[:, 15] - t[:, 0]):.0f} cycles: " + "; ".join(out))
