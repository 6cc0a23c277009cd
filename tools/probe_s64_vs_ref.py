#!/usr/bin/env python3
"""The 64x64 / 32x64 ring kernels and the 32x32 split-K kernel against a float64 reference on the slice / tee / tee_add case of
tests/test_gpu_split16.py::test_conv1d_cl_split16_narrow_epilogues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from speech_diarization_amd import ops, _native as N
dev = torch.device("cuda", 0)
lib = N.load()
g = torch.Generator().manual_seed(5)
B, T, C, hid = 3, 57, 256, 64
xbig = torch.randn(B * T, C, generator=g)
w = torch.randn(hid, hid, 3, generator=g) / 14
x_d = xbig.to(dev)
xt = xbig[:, 64:128].double().view(B, T, hid).transpose(1, 2)
ref = F.conv1d(F.pad(xt, (2, 2), mode="reflect"), w.double(), None, dilation=2).transpose(1, 2).reshape(B * T, hid).relu()
ws, s = ops.pack_weight_split16(w, dev)
for label, s64, sk in (("ring", -1, -1), ("split32", 0, -1), ("128x128", 0, 0)):
    N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, s64), "t"); N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, sk), "t")
    out = torch.zeros(B * T, C, device=dev); tee = torch.zeros(B * T, hid, device=dev)
    ops.conv1d_cl(x_d, ops.pack_weight(w, dev), T, cin=hid, dil=2, act="relu", a_col0=64, out=out, o_col0=64, tee=tee, tee_lo=0, tee_hi=hid, tee_add=x_d, ta_col0=128)
    e = (out[:, 64:128].cpu().double() - ref).abs().max().item()
    et = (tee.cpu().double() - (ref + xbig[:, 128:192].double())).abs().max().item()
    print(f"{label:8s} max err y {e:.3e} tee {et:.3e} (max |y| {ref.abs().max():.3f})  zeros outside: {out[:, :64].abs().max().item() == 0 and out[:, 128:].abs().max().item() == 0}")
out = torch.zeros(B * T, C, device=dev); tee = torch.zeros(B * T, hid, device=dev)
ops.conv1d_cl_split16(x_d, ws, s, T, narrow=True, cin=hid, dil=2, act="relu", a_col0=64, out=out, o_col0=64, tee=tee, tee_lo=0, tee_hi=hid, tee_add=x_d, ta_col0=128)
print(f"split16  max err y {(out[:, 64:128].cpu().double() - ref).abs().max().item():.3e}")
