#!/usr/bin/env python3
"""Time (or profile under rocprofv3 --pmc) one shape of the implicit-GEMM conv operator.

    python tools/probe_conv.py --B 1024 --T 201 --cin 1024 --cout 1024 --taps 1 --reps 5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--T", type=int, default=201)
    ap.add_argument("--cin", type=int, default=1024)
    ap.add_argument("--cout", type=int, default=1024)
    ap.add_argument("--taps", type=int, default=1)
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--f16", action="store_true")
    a = ap.parse_args()
    from speech_diarization_amd import ops
    dev = torch.device("cuda", 0)
    M = a.B * a.T
    x = torch.zeros(M, a.cin, device=dev) if a.zeros else torch.randn(M, a.cin, device=dev)
    w = torch.randn(a.cout, a.cin, a.taps) / (a.cin * a.taps) ** 0.5
    if a.f16:
        x = x.half()
    wp = ops.pack_weight(w, dev, torch.float16 if a.f16 else torch.float32)
    bias = torch.randn(a.cout, device=dev)
    scale = torch.rand(a.cout, device=dev) + 0.5
    shift = torch.randn(a.cout, device=dev)
    out = torch.empty(M, a.cout, device=dev, dtype=torch.float16 if a.f16 else torch.float32)
    ops.conv1d_cl(x, wp, a.T, cin=a.cin, dil=a.dil, bias=bias, act="relu", scale=scale, shift=shift, out=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
    ev[0].record()
    for i in range(a.reps):
        ops.conv1d_cl(x, wp, a.T, cin=a.cin, dil=a.dil, bias=bias, act="relu", scale=scale, shift=shift, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps)]
    flops = 2.0 * M * a.cout * a.cin * a.taps
    best = min(ms)
    print(f"M={M} cin={a.cin} cout={a.cout} taps={a.taps}: min {best:.3f} ms  median {sorted(ms)[len(ms)//2]:.3f} ms  "
          f"{flops / best / 1e9:.1f} TFLOP/s (best)")


if __name__ == "__main__":
    main()
