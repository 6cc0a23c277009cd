#!/usr/bin/env python3
"""Time the f32-split16x3 conv (pack + conv, and the conv alone) next to the exact-f32 and f16 kernels on one shape.

    python tools/probe_split16.py --B 5000 --cin 1024 --cout 1024
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ms[0], ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=5000)
    ap.add_argument("--T", type=int, default=201)
    ap.add_argument("--cin", type=int, default=1024)
    ap.add_argument("--cout", type=int, default=1024)
    ap.add_argument("--taps", type=int, default=1)
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    from speech_diarization_amd import _native as N, ops
    dev = torch.device("cuda", 0)
    M = a.B * a.T
    x = torch.randn(M, a.cin, device=dev)
    w = torch.randn(a.cout, a.cin, a.taps) / (a.cin * a.taps) ** 0.5
    bias, scale, shift = torch.randn(a.cout, device=dev), torch.rand(a.cout, device=dev) + 0.5, torch.randn(a.cout, device=dev)
    flops = 2.0 * M * a.cout * a.cin * a.taps
    kw = dict(cin=a.cin, bias=bias, act="relu", scale=scale, shift=shift)
    out = torch.empty(M, a.cout, device=dev)
    w32 = ops.pack_weight(w, dev)
    b32, m32 = timeit(lambda: ops.conv1d_cl(x, w32, a.T, out=out, **kw), a.reps)
    ws, s = ops.pack_weight_split16(w, dev)
    bs, msp = timeit(lambda: ops.conv1d_cl_split16(x, ws, s, a.T, out=out, **kw), a.reps)
    bp, mp = timeit(lambda: ops.split16_pack(x, 0, a.cin), a.reps)
    bn, mn = timeit(lambda: ops.conv1d_cl_split16(x, ws, s, a.T, out=out, narrow=True, **kw), a.reps)
    xh = x.half(); w16 = ops.pack_weight(w, dev, torch.float16); o16 = torch.empty(M, a.cout, device=dev, dtype=torch.float16)
    b16, m16 = timeit(lambda: ops.conv1d_cl(xh, w16, a.T, out=o16, **kw), a.reps)
    print(f"M={M} {a.cin}->{a.cout} k{a.taps}: exact f32 {m32:.3f} ms ({flops / m32 / 1e9:.0f} TFLOP/s) | split16x3 pack+conv {msp:.3f} ms "
          f"({flops / msp / 1e9:.0f} f32-equivalent TFLOP/s), pack alone {mp:.3f} ms ({M * a.cin * 8 / mp / 1e6:.0f} GB/s), conv alone {msp - mp:.3f} ms "
          f"({flops / (msp - mp) / 1e9:.0f}; {3 * flops / (msp - mp) / 1e9:.0f} f16 TFLOP/s issued) | narrow split kernel (no pack) {mn:.3f} ms "
          f"({flops / mn / 1e9:.0f}) | f16 {m16:.3f} ms ({flops / m16 / 1e9:.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
