#!/usr/bin/env python3
"""Diagnostic (a --variant built with -DSD_STAMP, selected with SD_EXPERIMENT=1 SD_HIP_LIB=...): where the waves of fbank_utt16_kernel
spend their cycles, per phase (median over workgroups 4096..5119)."""
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
lib = _native.load(); cnt = 1024 * 8 * 16; buf = (C.c_ulonglong * cnt)()
lib.sd_debug_read_utt16_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sd_debug_read_utt16_stamps(buf, cnt) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 16).astype(np.float64)
names = {1: "loads + edges + peak", 2: "(barrier) scale + split + image", 3: "barrier", 4: "r0 stage 1", 5: "r0 barrier + transposes", 6: "r0 stage 2", 7: "r0 mel + log",
         8: "r1 stage 1", 9: "r1 transposes", 10: "r1 stage 2", 11: "r1 mel + log", 14: "final barrier", 15: "floor + mean + store"}
ntiles = (1 + n // 160 + 15) // 16
for w in range(8):
    t = raw[:, w, :]
    prev = t[:, 0]
    out = []
    for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 14, 15):
        if i in (8, 9, 10, 11) and w + 8 >= ntiles:
            continue
        if i in (5, 6, 7) and w >= ntiles:
            continue
        d = t[:, i] - prev
        out.append(f"{names[i]} {np.median(d):.0f}")
        prev = t[:, i]
    print(f"wave {w}: total {np.median(t[:, 15] - t[:, 0]):.0f} cycles: " + "; ".join(out))
