#!/usr/bin/env python3
"""Where the split-f16 fbank kernels' error against float64 comes from, in the terms of DESIGN.md's bound.

Model (oracle/fbank_ref.py: log_mel_error_unit): |err(t, m)| <= r0 + k 2^-22 A_t sqrt(S_m / (mel_tm + eps)).  This tool measures the k each
input class needs for several r0, for both HIP kernels (one launch up to 201 frames, folded beyond) and for torch.stft in f32 on the CPU.
    python tools/fbank_error_model.py [--seeds 32]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import fbank_ref
from speech_diarization_amd import synth
from speech_diarization_amd.engine import fbank_device
from speech_diarization_amd.features import FbankPlan


def inputs(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    white = synth.synthetic_segments(seed, 1, n, std=0.1)[0]
    yield "white", white
    tone = 0.5 * np.sin(2 * np.pi * (300.0 + 3000.0 * rng.random()) * t)
    yield "tone + broadband 60 dB below", (tone + 0.5e-3 * rng.standard_normal(n)).astype(np.float32)
    yield "tone + broadband 40 dB below", (tone + 0.5e-2 * rng.standard_normal(n)).astype(np.float32)
    conv = synth.synthetic_conversation(max(2.0, n / 16000.0), 2, seed=seed)
    yield "synthetic voices", conv.wav[:n].astype(np.float32)
    q = white.copy(); q[: n // 2] *= 1e-3
    yield "60 dB level step", q


R0S = (1e-5, 2e-5, 3e-5, 5e-5)


def measure(kind, wav, dev, plan, torch_f32=False):
    """-> (max |err| in ln units, [k needed for each r0 in R0S]) over all (frame, mel) of the [B, n] batch, mean_norm off, for the model
        |err(t, m)| <= r0 + k unit(t, m)          (oracle/fbank_ref.py: log_mel_error_unit)
    torch_f32: the same for the f32 torch.stft formulation on the CPU (the arithmetic class of the reference's own path), for scale."""
    if torch_f32:
        got = (fbank_ref.fbank_batch_torch(torch.from_numpy(wav), mean_nor=False) if kind == "torchaudio"
               else fbank_ref.speechbrain_fbank_torch(torch.from_numpy(wav), mean_norm=False)).numpy().astype(np.float64)
    else:
        got = fbank_device(torch.from_numpy(wav).to(dev), plan, mean_norm=False).cpu().numpy().astype(np.float64)
    ref, unit, live = fbank_ref.log_mel_error_unit(wav, kind)
    to_ln = 1.0 if kind == "torchaudio" else np.log(10.0) / 10.0
    err = np.abs(got - ref) * live * to_ln
    unit = unit * to_ln
    ks = [float((np.maximum(err - r0, 0.0) / np.maximum(unit, 1e-300)).max()) for r0 in R0S]
    return float(err.max()), ks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else None
    print("model: |err(t, m)| <= r0 + k 2^-22 A_t sqrt(S_m / (mel_tm + eps));  k needed for r0 = " + ", ".join(f"{r:.0e}" for r in R0S))
    for who in (("hip",) if dev is not None else ()) + ("torch-f32",):
        for kind in ("torchaudio", "speechbrain"):
            plan = FbankPlan(kind) if who == "hip" else None
            for n in (32000, 48000) if who == "hip" else (32000,):      # one-launch kernel / folded kernel
                worst = {}
                for seed in range(a.seeds if who == "hip" else min(a.seeds, 8)):
                    for name, w in inputs(seed, n):
                        e, ks = measure(kind, w[None, :], dev, plan, torch_f32=who != "hip")
                        we, wk = worst.get(name, (0.0, [0.0] * len(R0S)))
                        worst[name] = (max(we, e), [max(p, q) for p, q in zip(wk, ks)])
                for name, (e, ks) in worst.items():
                    print(f"{who:9s} {kind:11s} n={n}  {name:30s} max |err| {e:.3e}   k: " + "  ".join(f"{k:8.2f}" for k in ks), flush=True)


if __name__ == "__main__":
    main()
