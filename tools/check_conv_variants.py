#!/usr/bin/env python3
"""Digest of the f16 conv operator's outputs over a set of edge shapes (ragged M, taps, dilation, colstat, f32 output):
run under different kernel selections (SD_T256_V=old|new, SD_F16_KERNEL=reg) and diff the lines; also checks each
against a float64 torch reference."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_diarization_amd import ops

dev = torch.device("cuda", 0)
cases = [  # B, T, K, N, taps, dil, colstat, out f32
    (40, 201, 1024, 1024, 1, 1, False, False),
    (37, 201, 1024, 1024, 1, 1, True, False),
    (16, 201, 3072, 3072, 1, 1, True, False),
    (9, 101, 192, 1024, 1, 1, False, False),      # K tail (cin_pad 192), short segments
    (5, 626, 256, 1024, 3, 2, False, False),
    (7, 201, 128, 1280, 5, 1, False, True),
    (3, 61, 64, 1024, 3, 3, False, False),
    (1, 300, 1024, 2048, 1, 1, False, False),
]
g = torch.Generator(device="cpu").manual_seed(5)
for B, T, K, N, taps, dil, cs, f32o in cases:
    M = B * T
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(N, K, taps, generator=g) / (K * taps) ** 0.5).half().float()
    bias = torch.randn(N, generator=g).to(dev); scale = (torch.rand(N, generator=g) + 0.5).to(dev); shift = torch.randn(N, generator=g).to(dev)
    wp = ops.pack_weight(w, dev, torch.float16)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32o else torch.float16)
    csb = torch.zeros(ops.colstat_floats(M, N), device=dev) if cs else None
    ops.conv1d_cl(x, wp, T, cin=K, dil=dil, bias=bias, act="relu", scale=scale, shift=shift, out=out, colstat=csb)
    torch.cuda.synchronize()
    # float64 reference: reflect-padded dilated conv per segment
    xs = x.double().view(B, T, K).permute(0, 2, 1)
    pad = (taps // 2) * dil
    xp = torch.nn.functional.pad(xs, (pad, pad), mode="reflect") if pad else xs
    ref = torch.nn.functional.conv1d(xp, w.double().to(dev), dilation=dil)
    ref = (torch.relu(ref + bias.double()[None, :, None]) * scale.double()[None, :, None] + shift.double()[None, :, None]).permute(0, 2, 1).reshape(M, N)
    err = (out.double() - ref).abs().max().item()
    tol = 2e-3 * ref.abs().max().item() if not f32o else 1e-4 * ref.abs().max().item()
    dig = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    extra = ""
    if cs:
        mean = ops.colstat_finish(csb, out, B, T, pivot=shift, want_std=True)
        mref = torch.cat([ref.view(B, T, N).mean(1), ref.view(B, T, N).var(1, unbiased=False).clamp_min(1e-12).sqrt()], 1)
        extra = f" stat_err={(mean.double() - mref).abs().max().item():.2e} stat={hashlib.sha256(mean.cpu().numpy().tobytes()).hexdigest()[:12]}"
    print(f"B={B} T={T} K={K} N={N} taps={taps} dil={dil} cs={int(cs)} f32={int(f32o)}: {dig} err={err:.3e} {'OK' if err <= tol and not torch.isnan(out).any() else 'FAIL'}{extra}", flush=True)
