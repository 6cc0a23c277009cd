"""Host-side tables for the HIP fbank kernel: analysis windows and mel filter matrices.

Two front ends exist on the reference's path and differ only in these tables and
three switches (padding, log law, top_db floor):

* "torchaudio"  — `fbank_batch` [REF speech_encode.py:17-36]: MelSpectrogram(n_fft=400,
  hop=160, f_min=20, f_max=sr/2-100, power=2) with torchaudio defaults (periodic Hann,
  center/reflect, HTK mel, norm=None), then ln(x + 1e-6).  SURVEY.md Appendix A.1.
* "speechbrain" — the Fbank inside `EncoderClassifier.encode_batch`
  [REF speech_encode.py:77]: periodic Hamming, center/zero pad, 80 triangular filters
  0..8000 Hz built from the left bandwidth, 10*log10(max(x, 1e-10)), top_db = 80.
  SURVEY.md Appendix A.2.

The tables are computed once in float64 and handed to `sd_fbank_plan_create`.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _native as N

N_FFT = 400
HOP = 160
N_FREQ = N_FFT // 2 + 1


def _hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def _mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def periodic_window(kind: str, n: int = N_FFT) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    if kind == "hann":
        w = 0.5 - 0.5 * np.cos(2.0 * math.pi * k / n)
    elif kind == "hamming":
        w = 0.54 - 0.46 * np.cos(2.0 * math.pi * k / n)
    else:
        raise ValueError(f"unknown window {kind!r}")
    w = w.astype(np.float32)
    # enforce exact symmetry w[k] == w[n-k] after rounding (the kernel folds the DFT on it)
    w[n // 2 + 1:] = w[1:(n + 1) // 2][::-1]
    return w


def mel_filters_torchaudio(n_mels: int = 80, sr: int = 16000, f_min: float = 20.0, f_max: float | None = None,
                           n_freq: int = N_FREQ) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(mel_scale="htk", norm=None) -> [n_freq, n_mels]."""
    f_max = sr / 2 - 100 if f_max is None else f_max
    freqs = np.linspace(0.0, sr // 2, n_freq)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2))
    diff = pts[1:] - pts[:-1]
    slopes = pts[None, :] - freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


def mel_filters_speechbrain(n_mels: int = 80, sr: int = 16000, f_min: float = 0.0, f_max: float = 8000.0) -> np.ndarray:
    """speechbrain Filterbank (triangular, left-bandwidth symmetric in Hz) -> [n_freq, n_mels]."""
    hz = _mel_to_hz(np.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2))
    band = (hz[1:] - hz[:-1])[:-1]
    centre = hz[1:-1]
    freqs = np.linspace(0.0, sr // 2, N_FREQ)
    slope = (freqs[:, None] - centre[None, :]) / band[None, :]
    return np.maximum(0.0, np.minimum(slope + 1.0, -slope + 1.0)).astype(np.float32)


@dataclass(frozen=True)
class FrontEnd:
    window: str
    pad_mode: int
    log_mode: int
    log_eps: float
    top_db: float
    filters: str


FRONT_ENDS = {
    "torchaudio": FrontEnd("hann", N.SD_PAD_REFLECT, N.SD_LOG_LN_EPS, 1e-6, -1.0, "torchaudio"),
    "speechbrain": FrontEnd("hamming", N.SD_PAD_ZERO, N.SD_LOG_DB_TOPDB, 1e-10, 80.0, "speechbrain"),
}


class FbankPlan:
    """Owns an `sd_fbank_plan` (device-side DFT basis + sparse mel table)."""

    def __init__(self, kind: str = "speechbrain", n_mels: int = 80, sr: int = 16000):
        if kind not in FRONT_ENDS:
            raise ValueError(f"unknown front end {kind!r}")
        win_length, hop = int(sr * 0.025), int(sr * 0.010)  # [REF speech_encode.py:14-15]
        if kind == "speechbrain" and (win_length, hop) != (N_FFT, HOP):
            raise ValueError(f"the ECAPA encoder's own front end is 16 kHz (n_fft=400, hop=160); sr={sr} gives {win_length}/{hop}")
        if win_length < 8 or hop < 1:
            raise ValueError(f"sr={sr} gives a window of {win_length} samples and a hop of {hop}")
        fe = FRONT_ENDS[kind]
        self.kind, self.n_mels, self.sr = kind, n_mels, sr
        self.n_fft, self.hop = win_length, hop
        # 16 kHz: the split-f16 kernels (sd_fbank_utt16.hip / sd_fbank.hip); any other rate: a float64 DFT kernel + the mel product
        # on the exact-f32 conv operator (sd_fbank_generic.hip); the library decides from (n_fft, hop)
        self.window = periodic_window(fe.window, win_length)
        n_freq = win_length // 2 + 1
        self.mel = (mel_filters_torchaudio(n_mels, sr, n_freq=n_freq) if fe.filters == "torchaudio" else mel_filters_speechbrain(n_mels, sr))
        self._lib = N.load()
        win = np.ascontiguousarray(self.window)
        mel = np.ascontiguousarray(self.mel)
        self._h = self._lib.sd_fbank_plan_create(win.ctypes.data_as(C.c_void_p), win_length, hop, mel.ctypes.data_as(C.c_void_p), n_mels,
                                                 fe.pad_mode, fe.log_mode, C.c_float(fe.log_eps), C.c_float(fe.top_db))
        if not self._h:
            raise N.SdError(f"sd_fbank_plan_create failed: {N.last_error()}")

    @property
    def handle(self):
        return self._h

    @staticmethod
    def num_frames(n: int) -> int:
        """Frames of an n-sample utterance at the 16 kHz framing (the encoder's front end)."""
        return 1 + n // HOP

    def frames(self, n: int) -> int:
        """Frames at THIS plan's framing: torch.stft(center=True) gives 1 + (n + 2 (n_fft // 2) - n_fft) // hop."""
        return 1 + (n + 2 * (self.n_fft // 2) - self.n_fft) // self.hop

    def workspace_bytes(self, B: int, n: int) -> int:
        return int(self._lib.sd_fbank_workspace_bytes(self._h, B, n))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.sd_fbank_plan_destroy(h)
            except Exception:
                pass
