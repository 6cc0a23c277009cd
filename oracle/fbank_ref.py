"""CPU restatement of the two log-mel front ends on the reference's path.  TEST INFRASTRUCTURE.

PARITY UNPINNED: torchaudio / speechbrain are absent (see oracle/__init__.py); this file
restates their published algorithms around the reference's
own parameters:

* `fbank_batch_ref`   follows [REF speech_encode.py:10-38]: MelSpectrogram(sample_rate=sr,
  n_mels, n_fft=win_length=int(sr*0.025), hop_length=int(sr*0.010), f_min=20,
  f_max=sr/2-100, power=2) -> log(feat + 1e-6) -> transpose -> minus mean over T.
* `speechbrain_fbank_ref` is the feature stage of `encoder.encode_batch(x)`
  [REF speech_encode.py:77]: STFT(hamming, center, constant pad) -> power -> triangular
  filterbank 0..8000 Hz -> 10 log10(clamp 1e-10) -> top_db 80 -> sentence mean-norm.

Two formulations each: an explicit float64 framed DFT (numpy) and a float32 `torch.stft`
one (also the timed CPU baseline); tests check they agree.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz(m):
    return 700.0 * (np.power(10.0, np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'): [n_freqs, n_mels]."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
    f_pts = mel_to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.clip(np.minimum(down, up), 0.0, None)


def speechbrain_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, f_max: float = 8000.0) -> np.ndarray:
    """speechbrain.processing.features.Filterbank triangular filters: [n_freqs, n_mels]."""
    mel = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
    hz = mel_to_hz(mel)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    slope = (all_freqs[:, None] - f_central[None, :]) / band[None, :]
    left, right = slope + 1.0, -slope + 1.0
    return np.clip(np.minimum(left, right), 0.0, None)


def _window(kind: str, n: int) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    if kind == "hann":
        return 0.5 - 0.5 * np.cos(2 * math.pi * k / n)       # torch.hann_window(n, periodic=True)
    return 0.54 - 0.46 * np.cos(2 * math.pi * k / n)           # torch.hamming_window(n, periodic=True)


def _power_spectrogram_f64(wavs: np.ndarray, n_fft: int, hop: int, window: np.ndarray, pad_mode: str) -> np.ndarray:
    """center=True STFT power, [B, T, n_fft//2+1], float64, by explicit framing + real DFT."""
    x = np.asarray(wavs, dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect" if pad_mode == "reflect" else "constant")
    T = 1 + (xp.shape[1] - n_fft) // hop          # torch.stft: = 1 + n // hop for an even n_fft, 1 + (n - 1) // hop for an odd one
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = xp[:, idx] * window[None, None, :]
    spec = np.fft.rfft(frames, n=n_fft, axis=-1)
    return spec.real ** 2 + spec.imag ** 2


def fbank_batch_ref(wavs: np.ndarray, sr: int = 16000, n_mels: int = 80, mean_nor: bool = True) -> np.ndarray:
    """float64 restatement of fbank_batch [REF speech_encode.py:10-38] -> [B, T, n_mels]."""
    assert wavs.ndim == 2
    win_length = int(sr * 0.025)
    hop_length = int(sr * 0.010)
    power = _power_spectrogram_f64(wavs, win_length, hop_length, _window("hann", win_length), "reflect")
    fb = melscale_fbanks_htk(win_length // 2 + 1, 20.0, sr / 2 - 100, n_mels, sr)
    feat = np.log(power @ fb + 1e-6)
    if mean_nor:
        feat = feat - feat.mean(axis=1, keepdims=True)
    return feat


def speechbrain_fbank_ref(wavs: np.ndarray, sr: int = 16000, n_mels: int = 80, mean_norm: bool = True) -> np.ndarray:
    """float64 restatement of speechbrain Fbank + InputNormalization(sentence, std_norm=False)."""
    assert wavs.ndim == 2
    n_fft = 400
    win_length = int(round(sr / 1000.0 * 25))
    hop_length = int(round(sr / 1000.0 * 10))
    power = _power_spectrogram_f64(wavs, n_fft, hop_length, _window("hamming", win_length), "constant")
    fb = speechbrain_filterbank(n_fft // 2 + 1, n_mels, sr)
    x_db = 10.0 * np.log10(np.clip(power @ fb, 1e-10, None))
    floor = x_db.max(axis=(1, 2), keepdims=True) - 80.0
    x_db = np.maximum(x_db, floor)
    if mean_norm:
        x_db = x_db - x_db.mean(axis=1, keepdims=True)
    return x_db


# ----------------------------------------------------------- float32 torch formulations

def fbank_batch_torch(wavs: torch.Tensor, sr: int = 16000, n_mels: int = 80, mean_nor: bool = True) -> torch.Tensor:
    n_fft, hop = int(sr * 0.025), int(sr * 0.010)
    spec = torch.stft(wavs.float(), n_fft, hop, n_fft, window=torch.hann_window(n_fft), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                      # [B, F, T]
    fb = torch.from_numpy(melscale_fbanks_htk(n_fft // 2 + 1, 20.0, sr / 2 - 100, n_mels, sr)).float()
    feat = torch.log(torch.matmul(power.transpose(1, 2), fb) + 1e-6)
    if mean_nor:
        feat = feat - feat.mean(1, keepdim=True)
    return feat


def speechbrain_fbank_torch(wavs: torch.Tensor, sr: int = 16000, n_mels: int = 80, mean_norm: bool = True) -> torch.Tensor:
    n_fft, hop = 400, 160
    spec = torch.stft(wavs.float(), n_fft, hop, n_fft, window=torch.hamming_window(n_fft), center=True,
                      pad_mode="constant", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = torch.from_numpy(speechbrain_filterbank(n_fft // 2 + 1, n_mels, sr)).float()
    x_db = 10.0 * torch.log10(torch.clamp(torch.matmul(power.transpose(1, 2), fb), min=1e-10))
    floor = x_db.amax(dim=(-2, -1), keepdim=True) - 80.0
    x_db = torch.maximum(x_db, floor)
    if mean_norm:
        x_db = x_db - x_db.mean(1, keepdim=True)
    return x_db


# ----------------------------------------------------------- error model of an f32-class DFT (tests/test_gpu_fbank_ecapa.py, tools/fbank_error_model.py)

def log_mel_error_unit(wavs: np.ndarray, kind: str = "torchaudio"):
    """What a log-mel value may differ by from this float64 restatement when the DFT is computed in f32-class arithmetic.

    A frame's DFT comes out with an absolute error delta = k 2^-22 A_t in re and im, where A_t^2 = sum_p (w_p x_p)^2 is the frame's windowed
    energy (the level the bins of a white frame sit at; rounding errors of a length-N transform scale with the LARGEST component, not with the
    bin).  A mel value mel_m = sum_k fb_mk |X_k|^2 then errs by at most 2 delta sqrt(S_m mel_m) (S_m = sum_k fb_mk, Cauchy-Schwarz, delta^2
    dropped), its logarithm by that over (mel_m + eps), i.e. by at most

        k * unit(t, m),     unit = 2^-22 A_t sqrt(S_m / (mel_tm + eps))        (natural-log units),

    plus a level-independent r0 for the mel product and the log.  Quiet bins of loud frames are where it shows: 2 k 2^-22 10^(DR / 20) for a
    bin DR dB below the frame's level.  Measured (tools/fbank_error_model.py, 32 seeds x 5 input classes): torch.stft in f32 -- the
    arithmetic class of the reference's own path -- needs k <= 5.5 at r0 = 3e-5; the HIP kernels k <= 8.9 (one launch) / 4.2 (folded).

    -> (ref [B, T, n_mels]: the raw log-mel values (no mean removal; top_db floor applied for "speechbrain"),
        unit [B, T, n_mels] in the units of ref (ln, or dB for "speechbrain"),
        live [B, T, n_mels] bool: False where the top_db floor replaced the value (no DFT error left in it))."""
    x = np.asarray(wavs, dtype=np.float64)
    if kind == "torchaudio":
        win, pad_mode, eps = _window("hann", 400), "reflect", 1e-6
        fb = melscale_fbanks_htk(201, 20.0, 7900.0, 80, 16000)
    else:
        win, pad_mode, eps = _window("hamming", 400), "constant", 1e-10
        fb = speechbrain_filterbank(201, 80, 16000)
    mel = _power_spectrogram_f64(x, 400, 160, win, pad_mode) @ fb
    xp = np.pad(x, ((0, 0), (200, 200)), mode="reflect" if pad_mode == "reflect" else "constant")
    idx = np.arange(mel.shape[1])[:, None] * 160 + np.arange(400)[None, :]
    amp = np.sqrt(((xp[:, idx] * win) ** 2).sum(-1))                     # A_t, [B, T]
    s = fb.sum(0)
    if kind == "torchaudio":
        ref = np.log(mel + eps)
        unit = 2.0 ** -22 * amp[:, :, None] * np.sqrt(s[None, None, :] / (mel + eps))
        return ref, unit, np.ones(ref.shape, dtype=bool)
    db = 10.0 * np.log10(np.clip(mel, eps, None))
    floor = db.max(axis=(1, 2), keepdims=True) - 80.0
    unit = (10.0 / np.log(10.0)) * 2.0 ** -22 * amp[:, :, None] * np.sqrt(s[None, None, :] / np.maximum(mel, eps))
    return np.maximum(db, floor), unit, db > floor + 1e-3
