#!/usr/bin/env python3
"""Randomised consistency run over the paths that were new in round 5 (GPU box; a few minutes):
  1. `fbank_batch` at random sample rates / lengths / batch sizes / mel counts against the float64 oracle (generic path: < 5e-5);
  2. `ecapa_encode_batches` with random batch shapes and 2 / 3 lanes against the one-at-a-time calls, bit for bit;
  3. the 256x256 f16 / split16x3 ring kernel's super-tile walk against the hardware dispatch on random shapes, bit for bit.
    python tools/stress_round5.py [--seed 0] [--rounds 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import fbank_ref
from speech_diarization_amd import _native, ops, speech_encode, synth

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
bad = 0

worst = 0.0
for it in range(a.rounds):
    sr = int(rng.choice([4000, 8000, 11025, 12000, 22050, 24000, 32000, 44100, 48000, 96000, int(rng.integers(3000, 60000))]))
    n_fft = int(sr * 0.025)
    n = int(rng.integers(n_fft // 2 + 2, 3 * sr))
    B = int(rng.integers(1, 6))
    n_mels = int(rng.choice([24, 40, 64, 80, 96, 128]))
    mean_nor = bool(rng.integers(0, 2))
    wav = synth.synthetic_segments(1000 + it, B, n, std=float(rng.choice([0.3, 0.05, 0.001])))
    got = speech_encode.fbank_batch(wav, sr=sr, n_mels=n_mels, mean_nor=mean_nor)
    ref = fbank_ref.fbank_batch_ref(wav, sr=sr, n_mels=n_mels, mean_nor=mean_nor)
    err = float(np.abs(got - ref).max()) if got.shape == ref.shape else float("inf")
    worst = max(worst, err) if sr != 16000 else worst
    if not err < (5e-5 if sr != 16000 else 2e-4):
        bad += 1
        print(f"fbank_batch sr={sr} n={n} B={B} n_mels={n_mels} mean_nor={mean_nor}: max err {err:.3e} shapes {got.shape} {ref.shape}", flush=True)
print(f"1. fbank_batch at {a.rounds} random (sr, n, B, n_mels): worst error off 16 kHz {worst:.2e}", flush=True)

enc = speech_encode.using_ecapa_encoder()
for it in range(3):
    shapes = [(int(rng.integers(1, 40)), int(rng.integers(5, 300)) * 160 + int(rng.integers(0, 160))) for _ in range(int(rng.integers(3, 12)))]
    batches = [synth.synthetic_segments(2000 + 50 * it + i, b, n) for i, (b, n) in enumerate(shapes)]
    want = [speech_encode.ecapa_encode_batch(b) for b in batches]
    for lanes in (2, 3):
        got = enc.encode_batches(batches, lanes=lanes)
        same = all(np.array_equal(g, w) for g, w in zip(got, want))
        bad += not same
        print(f"2. encode_batches, {len(batches)} batches {shapes[:3]}..., {lanes} lanes: {'bitwise equal' if same else 'DIFFERS'}", flush=True)

lib = _native.load()
g = torch.Generator().manual_seed(a.seed)
for it in range(10):
    T = 201
    B = int(rng.integers(1, 420))
    cin = int(rng.choice([1024, 2048, 3072])); cout = int(rng.choice([1024, 2048, 3072]))
    M = B * T
    mode = "f16" if it % 2 == 0 else "split16"
    w = torch.randn(cout, cin, 1, generator=g) / cin ** 0.5
    bias, scale, shift = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    x = torch.randn(M, cin, generator=g).to(dev)
    outs = []
    for thr in (1 << 40, 0):
        _native.check(lib.sd_set_tuning(_native.SD_TUNE_T256_LOCKSTEP_TILES, thr), "sd_set_tuning")
        _native.check(lib.sd_set_tuning(_native.SD_TUNE_F16_NARROW_TILES, 0), "sd_set_tuning")
        if mode == "f16":
            y = torch.empty(M, cout, device=dev, dtype=torch.float16)
            ops.conv1d_cl(x.half(), ops.pack_weight(w, dev, torch.float16), T, cin=cin, bias=bias, act="relu", scale=scale, shift=shift, out=y)
        else:
            ws, s = ops.pack_weight_split16(w, dev)
            y = ops.conv1d_cl_split16(x, ws, s, T, cin=cin, bias=bias, act="relu", scale=scale, shift=shift)
        outs.append(y)
    same = bool(torch.equal(outs[0], outs[1]))
    bad += not same
    print(f"3. {mode} ring kernel, {B} segments {cin} -> {cout}: super-tile walk {'== dispatch, bitwise' if same else 'DIFFERS'}", flush=True)
    del x, outs
_native.check(lib.sd_set_tuning(_native.SD_TUNE_T256_LOCKSTEP_TILES, -1), "sd_set_tuning")
_native.check(lib.sd_set_tuning(_native.SD_TUNE_F16_NARROW_TILES, -1), "sd_set_tuning")
print("stress run:", "OK" if not bad else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
