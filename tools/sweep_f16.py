#!/usr/bin/env python3
"""Time the f16 conv operator over a list of shapes in one process (interleaved rounds, best and median per shape).

    python tools/sweep_f16.py [--B 5000] [--rounds 5] K,N[,taps[,dil]] ...
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=5000)
    ap.add_argument("--T", type=int, default=201)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--colstat", action="store_true")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (DVFS check: same cycles, less switching energy)")
    ap.add_argument("--lockstep", type=int, default=-1, help="SD_TUNE_T256_LOCKSTEP_TILES: 0 = always the persistent lock-step walk, 1 = never (a huge threshold), -1 = the default rule")
    ap.add_argument("shapes", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.lockstep >= 0:
        from speech_diarization_amd import _native
        _native.check(_native.load().sd_set_tuning(_native.SD_TUNE_T256_LOCKSTEP_TILES, 0 if a.lockstep == 0 else 1 << 40), "sd_set_tuning")
    M = a.B * a.T
    cases = []
    for s in a.shapes:
        f = [int(v) for v in s.split(",")]
        K, N = f[0], f[1]
        taps = f[2] if len(f) > 2 else 1
        dil = f[3] if len(f) > 3 else 1
        x = torch.zeros(M, K, device=dev, dtype=torch.float16) if a.zeros else (torch.randn(M, K, device=dev) * 0.5).half()
        w = torch.zeros(N, K, taps) if a.zeros else torch.randn(N, K, taps) / (K * taps) ** 0.5
        wp = ops.pack_weight(w, dev, torch.float16)
        bias = torch.randn(N, device=dev); scale = torch.rand(N, device=dev) + 0.5; shift = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        cs = torch.empty(ops.colstat_floats(M, N), device=dev) if a.colstat else None
        cases.append((K, N, taps, dil, x, wp, bias, scale, shift, out, cs))
    def run(c):
        K, N, taps, dil, x, wp, bias, scale, shift, out, cs = c
        ops.conv1d_cl(x, wp, a.T, cin=K, dil=dil, bias=bias, act="relu", scale=scale, shift=shift, out=out, colstat=cs)
    for c in cases:
        run(c)
    torch.cuda.synchronize()
    times = [[] for _ in cases]
    for _ in range(a.rounds):
        for i, c in enumerate(cases):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(c); e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1))
    for c, t in zip(cases, times):
        K, N, taps, dil = c[:4]
        fl = 2.0 * M * N * K * taps
        t = sorted(t)
        print(f"M={M} K={K} N={N} taps={taps} dil={dil}: best {t[0]:.3f} ms ({fl / t[0] / 1e9:.0f} TF)  median {t[len(t)//2]:.3f} ms ({fl / t[len(t)//2] / 1e9:.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
