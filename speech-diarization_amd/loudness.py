"""ITU-R BS.1770-4 integrated loudness, for `anti_stick_diarize.loudness_normalize` [REF anti_stick_diarize.py:53-61].

The reference calls `pyloudnorm.Meter(sr).integrated_loudness(y)` and `pyloudnorm.normalize.loudness(y, l, target)`
unconditionally.  pyloudnorm is not in this image, and skipping the step would hand VAD, SCD and the encoder a differently
scaled, unclipped signal, so the meter is restated here from pyloudnorm's published algorithm [UPSTREAM-RECALLED: parity
unpinned -- no fixture of the reference holds a loudness value]: the "K-weighting" filter pair as two biquads (high shelf
+4 dB at 1500 Hz, Q = 1/sqrt(2); high pass at 38 Hz, Q = 0.5; coefficients from the audio-EQ-cookbook forms at the signal's
own rate), 400 ms blocks with 75 % overlap, the absolute gate at -70 LUFS and the relative gate 10 LU below the loudness
of the blocks that pass the absolute one.  When pyloudnorm IS importable the caller uses it instead of this module.
Host numpy / scipy: one pass over the recording before anything reaches the GPU.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import lfilter

_CHANNEL_GAINS = (1.0, 1.0, 1.0, 1.41, 1.41)     # L, R, C, Ls, Rs


def _biquad(kind: str, gain_db: float, q: float, fc: float, rate: float):
    a_lin = 10.0 ** (gain_db / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    alpha = np.sin(w0) / (2.0 * q)
    c = np.cos(w0)
    if kind == "high_shelf":
        r = 2.0 * np.sqrt(a_lin) * alpha
        b = np.array([a_lin * ((a_lin + 1) + (a_lin - 1) * c + r), -2 * a_lin * ((a_lin - 1) + (a_lin + 1) * c),
                      a_lin * ((a_lin + 1) + (a_lin - 1) * c - r)])
        a = np.array([(a_lin + 1) - (a_lin - 1) * c + r, 2 * ((a_lin - 1) - (a_lin + 1) * c), (a_lin + 1) - (a_lin - 1) * c - r])
    elif kind == "high_pass":
        b = np.array([(1 + c) / 2, -(1 + c), (1 + c) / 2])
        a = np.array([1 + alpha, -2 * c, 1 - alpha])
    else:
        raise ValueError(kind)
    return b / a[0], a / a[0]


class Meter:
    """`Meter(rate).integrated_loudness(data)` with pyloudnorm's defaults (K-weighting, block 0.400 s)."""

    def __init__(self, rate: int, block_size: float = 0.400):
        self.rate = rate
        self.block_size = block_size
        self._filters = [_biquad("high_shelf", 4.0, 1.0 / np.sqrt(2.0), 1500.0, rate), _biquad("high_pass", 0.0, 0.5, 38.0, rate)]

    def integrated_loudness(self, data: np.ndarray) -> float:
        x = np.asarray(data, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        if x.ndim != 2 or x.shape[1] > 5:
            raise ValueError("Audio must be [samples] or [samples, channels <= 5]")
        n, ch = x.shape
        if n < self.block_size * self.rate:
            raise ValueError("Audio must have length greater than the block size")
        for b, a in self._filters:
            x = lfilter(b, a, x, axis=0)
        t_g, gamma_a, step = self.block_size, -70.0, 0.25
        n_blocks = int(np.round((n / self.rate - t_g) / (t_g * step)) + 1)
        j = np.arange(n_blocks)
        lo = (t_g * (j * step) * self.rate).astype(np.int64)
        hi = (t_g * (j * step + 1) * self.rate).astype(np.int64)
        csum = np.concatenate((np.zeros((1, ch)), np.cumsum(x * x, axis=0)), axis=0)
        z = (csum[np.minimum(hi, n)] - csum[np.minimum(lo, n)]).T / (t_g * self.rate)       # [ch, blocks]
        g = np.asarray(_CHANNEL_GAINS[:ch])[:, None]
        with np.errstate(divide="ignore"):
            l_j = -0.691 + 10.0 * np.log10((g * z).sum(axis=0))
            keep = l_j >= gamma_a
            z_avg = z[:, keep].mean(axis=1) if keep.any() else np.full(ch, np.nan)
            gamma_r = -0.691 + 10.0 * np.log10((g[:, 0] * z_avg).sum()) - 10.0
            keep = (l_j > gamma_r) & (l_j > gamma_a)
            z_avg = np.nan_to_num(z[:, keep].mean(axis=1)) if keep.any() else np.zeros(ch)
            return float(-0.691 + 10.0 * np.log10((g[:, 0] * z_avg).sum()))


def normalize_loudness(data: np.ndarray, input_loudness: float, target_loudness: float) -> np.ndarray:
    """pyloudnorm.normalize.loudness: one gain of 10^((target - measured) / 20)."""
    return np.asarray(data) * (10.0 ** ((target_loudness - input_loudness) / 20.0))
