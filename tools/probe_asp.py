#!/usr/bin/env python3
"""Time the fused attention-pooling operator (and the conv + pooling pair it replaces) at bench size."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops, _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=5000); ap.add_argument("--T", type=int, default=201)
ap.add_argument("--C", type=int, default=3072); ap.add_argument("--f16", action="store_true"); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0); dt = torch.float16 if a.f16 else torch.float32
a1 = torch.tanh(torch.randn(a.B * a.T, 128, device=dev)).to(dt)
wp = ops.pack_weight(torch.randn(a.C, 128, 1) / 4, dev, dt)
h = torch.randn(a.B * a.T, a.C, device=dev, dtype=dt)

def timed(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)

t_f = timed(lambda: ops.asp_attend_pool(a1, wp, h, a.B, a.T))
e = torch.empty(a.B * a.T, a.C, device=dev, dtype=dt); out = torch.empty(a.B, 2 * a.C, device=dev)
def two():
    ops.conv1d_cl(a1, wp, a.T, cin=128, out=e)
    N.check(N.load().sd_asp_pool_dt(e.data_ptr(), a.C, h.data_ptr(), N.SD_DT_F16 if a.f16 else N.SD_DT_F32, a.C, a.B, a.T, a.C,
                                    C.c_float(1e-12), out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "asp")
t_2 = timed(two)
fl = 2.0 * a.B * a.T * a.C * 128
print(f"{'f16' if a.f16 else 'f32'} B={a.B} T={a.T} C={a.C}: fused {t_f:.3f} ms ({fl / t_f / 1e9:.1f} TFLOP/s on the logits GEMM), conv + pool {t_2:.3f} ms")
