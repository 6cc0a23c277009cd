"""Deterministic synthetic weights and audio (there is no network for checkpoints or datasets).

Everything derives from a counter-based integer hash (splitmix64) so the build
container and the GPU box regenerate bit-identical integer streams from a seed
instead of shipping 83 MB of weights; floats come from those integers by exact
arithmetic plus Box-Muller in float64.

ECAPA-TDNN geometry and state-dict key names follow speechbrain's
`ECAPA_TDNN` as used by `EncoderClassifier.encode_batch`
[REF speech_encode.py:66-69,77], so a real
`embedding_model.ckpt` can be loaded through the same code path.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, tag: str, n: int, lane: int) -> np.ndarray:
    """n uniform doubles in (0, 1) for (seed, tag, lane)."""
    key = (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32)) ^ np.uint64(zlib.crc32(tag.encode()) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(lane)) & _MASK
        bits = _splitmix64(_splitmix64(ctr ^ key) + key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def uniform(seed: int, tag: str, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = _stream(seed, tag, n, 0)
    return (lo + (hi - lo) * u).reshape(shape).astype(np.float32)


def normal(seed: int, tag: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    u1 = _stream(seed, tag, n, 0)
    u2 = _stream(seed, tag, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (mean + std * z).reshape(shape).astype(np.float32)


@dataclass(frozen=True)
class EcapaConfig:
    """speechbrain ECAPA_TDNN hyper-parameters (spkrec-ecapa-* geometry by default)."""
    input_size: int = 80
    channels: tuple = (1024, 1024, 1024, 1024, 3072)
    kernel_sizes: tuple = (5, 3, 3, 3, 1)
    dilations: tuple = (1, 2, 3, 4, 1)
    attention_channels: int = 128
    lin_neurons: int = 192
    res2net_scale: int = 8
    se_channels: int = 128

    @property
    def n_blocks(self) -> int:
        return len(self.channels) - 2

    @staticmethod
    def small(width: int = 128, lin: int = 192) -> "EcapaConfig":
        """Reduced-width geometry for CPU-sized parity tests (same topology)."""
        return EcapaConfig(channels=(width, width, width, width, 3 * width), attention_channels=32,
                           lin_neurons=lin, res2net_scale=8, se_channels=32)


def _conv(sd: dict, seed: int, name: str, cout: int, cin: int, k: int) -> None:
    fan_in = cin * k
    sd[f"{name}.conv.weight"] = normal(seed, f"{name}.w", (cout, cin, k), std=1.0 / math.sqrt(fan_in))
    sd[f"{name}.conv.bias"] = normal(seed, f"{name}.b", (cout,), std=0.1)


def _bn(sd: dict, seed: int, name: str, c: int) -> None:
    # non-trivial running statistics so eval-BatchNorm handling is actually exercised
    sd[f"{name}.norm.weight"] = uniform(seed, f"{name}.g", (c,), 0.5, 1.5)
    sd[f"{name}.norm.bias"] = normal(seed, f"{name}.beta", (c,), std=0.1)
    sd[f"{name}.norm.running_mean"] = normal(seed, f"{name}.rm", (c,), std=0.1)
    sd[f"{name}.norm.running_var"] = uniform(seed, f"{name}.rv", (c,), 0.5, 1.5)


def make_ecapa_state_dict(seed: int = 1234, cfg: EcapaConfig = EcapaConfig()) -> dict:
    """Random-init weights of the ECAPA-TDNN architecture, speechbrain key names, numpy f32."""
    sd: dict = {}
    ch = cfg.channels
    _conv(sd, seed, "blocks.0.conv", ch[0], cfg.input_size, cfg.kernel_sizes[0])
    _bn(sd, seed, "blocks.0.norm", ch[0])
    for i in range(1, len(ch) - 1):
        c = ch[i]
        hid = c // cfg.res2net_scale
        p = f"blocks.{i}"
        _conv(sd, seed, f"{p}.tdnn1.conv", c, ch[i - 1], 1)
        _bn(sd, seed, f"{p}.tdnn1.norm", c)
        for j in range(cfg.res2net_scale - 1):
            _conv(sd, seed, f"{p}.res2net_block.blocks.{j}.conv", hid, hid, cfg.kernel_sizes[i])
            _bn(sd, seed, f"{p}.res2net_block.blocks.{j}.norm", hid)
        _conv(sd, seed, f"{p}.tdnn2.conv", c, c, 1)
        _bn(sd, seed, f"{p}.tdnn2.norm", c)
        _conv(sd, seed, f"{p}.se_block.conv1", cfg.se_channels, c, 1)
        _conv(sd, seed, f"{p}.se_block.conv2", c, cfg.se_channels, 1)
    cm = ch[-1]
    _conv(sd, seed, "mfa.conv", cm, cm, cfg.kernel_sizes[-1])
    _bn(sd, seed, "mfa.norm", cm)
    _conv(sd, seed, "asp.tdnn.conv", cfg.attention_channels, 3 * cm, 1)
    _bn(sd, seed, "asp.tdnn.norm", cfg.attention_channels)
    _conv(sd, seed, "asp.conv", cm, cfg.attention_channels, 1)
    _bn(sd, seed, "asp_bn", 2 * cm)
    _conv(sd, seed, "fc", cfg.lin_neurons, 2 * cm, 1)
    return sd


def config_from_state_dict(sd: dict) -> EcapaConfig:
    """Recover the geometry from a speechbrain-style state dict (synthetic or real)."""
    def shape(k):
        return tuple(sd[k].shape)
    c0, n_mels, k0 = shape("blocks.0.conv.conv.weight")
    channels, kernels = [c0], [k0]
    i = 1
    while f"blocks.{i}.tdnn1.conv.conv.weight" in sd:
        channels.append(shape(f"blocks.{i}.tdnn1.conv.conv.weight")[0])
        kernels.append(shape(f"blocks.{i}.res2net_block.blocks.0.conv.conv.weight")[2])
        i += 1
    n_blocks = i - 1
    hid = shape("blocks.1.res2net_block.blocks.0.conv.conv.weight")[0]
    scale = channels[1] // hid
    cm = shape("mfa.conv.conv.weight")[0]
    channels.append(cm)
    kernels.append(shape("mfa.conv.conv.weight")[2])
    return EcapaConfig(
        input_size=n_mels, channels=tuple(channels), kernel_sizes=tuple(kernels),
        dilations=tuple([1] + list(range(2, 2 + n_blocks)) + [1]),
        attention_channels=shape("asp.tdnn.conv.conv.weight")[0],
        lin_neurons=shape("fc.conv.weight")[0], res2net_scale=scale,
        se_channels=shape("blocks.1.se_block.conv1.conv.weight")[0],
    )


def synthetic_segments(seed: int, batch: int, n_samples: int, std: float = 0.1) -> np.ndarray:
    """Benchmark input: N(0, std^2) clipped to [-1, 1], f32 [batch, n_samples]."""
    x = normal(seed, f"seg.{n_samples}", (batch, n_samples), std=std)
    return np.clip(x, -1.0, 1.0)


def synthetic_segments_device(seed: int, batch: int, n_samples: int, device, std: float = 0.1, first_row: int = 0, chunk_rows: int = 500,
                              row_stride: int = 1):
    """`synthetic_segments` generated ON the device (counter-based generator, seed 0,
    N(0, std^2) clipped to [-1, 1]): the same splitmix64 counter stream in int64 torch arithmetic, Box-Muller in
    float64.  Rows first_row + k * row_stride (k < batch) of the stream, so ranks can own disjoint rows of one data set
    (row_stride = world size, first_row = rank: the round-robin shard of `dist.shard_indices`).
    Equal to the numpy generator up to the device's float64 log / cos rounding (<= 1 f32 ulp)."""
    import torch
    key = ((seed & 0xFFFFFFFF) << 32) ^ (zlib.crc32(f"seg.{n_samples}".encode()) & 0xFFFFFFFF)

    def i64(v: int) -> int:                      # two's-complement view of a 64-bit constant
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k: int):                          # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = x + i64(0x9E3779B97F4A7C15)
        z = (x ^ lsr(x, 30)) * i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)

    def unit(ctr):
        bits = mix(mix(ctr ^ i64(key)) + i64(key))
        return (lsr(bits, 11).to(torch.float64) + 0.5) * (1.0 / (1 << 53))

    out = torch.empty((batch, n_samples), dtype=torch.float32, device=device)
    for lo in range(0, batch, chunk_rows):
        hi = min(batch, lo + chunk_rows)
        rows = first_row + row_stride * torch.arange(lo, hi, dtype=torch.int64, device=device)
        idx = (rows[:, None] * n_samples + torch.arange(n_samples, dtype=torch.int64, device=device)[None, :]).reshape(-1)
        z = torch.sqrt(-2.0 * torch.log(unit(idx * 2))) * torch.cos(2.0 * math.pi * unit(idx * 2 + 1))
        out[lo:hi] = (std * z).to(torch.float32).clamp_(-1.0, 1.0).view(hi - lo, n_samples)
    return out


# ---------------------------------------------------------------- synthetic conversations

@dataclass
class Voice:
    f0: float
    formants: tuple
    bandwidths: tuple = (90.0, 110.0, 160.0)


DEFAULT_VOICES = (
    Voice(110.0, (730.0, 1090.0, 2440.0)),
    Voice(210.0, (270.0, 2290.0, 3010.0)),
    Voice(150.0, (530.0, 1840.0, 2480.0)),
    Voice(260.0, (660.0, 1720.0, 2410.0)),
    Voice(95.0, (440.0, 1020.0, 2240.0)),
    Voice(185.0, (300.0, 870.0, 2240.0)),
    Voice(130.0, (640.0, 1190.0, 2390.0)),
    Voice(235.0, (490.0, 1350.0, 1690.0)),
)


def _voice_signal(voice: Voice, n: int, sr: int, seed: int, tag: str) -> np.ndarray:
    """Harmonic source (30 harmonics, 1/h roll-off, slow 3 % vibrato) shaped by 3 formant resonances."""
    t = np.arange(n, dtype=np.float64) / sr
    vib = 1.0 + 0.03 * np.sin(2.0 * math.pi * 5.0 * t + 2.0 * math.pi * float(_stream(seed, tag + ".ph", 1, 0)[0]))
    phase = 2.0 * math.pi * np.cumsum(voice.f0 * vib) / sr
    y = np.zeros(n, dtype=np.float64)
    for hnum in range(1, 31):
        fh = voice.f0 * hnum
        if fh >= sr / 2:
            break
        gain = 0.0
        for fc, bw in zip(voice.formants, voice.bandwidths):
            gain += 1.0 / (1.0 + ((fh - fc) / bw) ** 2)
        y += (gain + 0.02) / hnum * np.sin(hnum * phase)
    return y


@dataclass
class Conversation:
    wav: np.ndarray                 # f32 [n]
    sr: int
    turns: list = field(default_factory=list)   # (start_s, end_s, speaker_index)


def synthetic_conversation(duration_s: float = 60.0, n_speakers: int = 2, sr: int = 16000, seed: int = 0,
                           turn_s=(3.0, 6.0), gap_s=(0.3, 0.8)) -> Conversation:
    """Alternating turns of synthetic voices separated by silence, -40 dB noise floor, peak 0.5."""
    n = int(round(duration_s * sr))
    wav = np.zeros(n, dtype=np.float64)
    u = _stream(seed, "conv.turns", 4096, 0)
    ui = 0
    t = float(gap_s[0])
    spk = 0
    turns = []
    while True:
        dur = turn_s[0] + (turn_s[1] - turn_s[0]) * u[ui]; ui += 1
        gap = gap_s[0] + (gap_s[1] - gap_s[0]) * u[ui]; ui += 1
        if t + dur > duration_s - 0.1:
            break
        s, e = int(round(t * sr)), int(round((t + dur) * sr))
        seg = _voice_signal(DEFAULT_VOICES[spk % len(DEFAULT_VOICES)], e - s, sr, seed, f"turn{len(turns)}")
        ramp = int(0.02 * sr)
        env = np.ones(e - s)
        env[:ramp] = np.linspace(0.0, 1.0, ramp)
        env[-ramp:] = np.linspace(1.0, 0.0, ramp)
        wav[s:e] += seg * env
        turns.append((s / sr, e / sr, spk))
        spk = (spk + 1) % n_speakers
        t = t + dur + gap
    peak = np.max(np.abs(wav)) or 1.0
    wav = 0.5 * wav / peak
    wav += 0.5 * 10 ** (-40 / 20) * normal(seed, "conv.noise", (n,)).astype(np.float64)
    return Conversation(wav.astype(np.float32), sr, turns)
