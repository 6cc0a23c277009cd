"""Minimal audio I/O for the entry points: WAV through scipy, FLAC through `flac.py` (torchaudio, librosa and soundfile — the
reference's readers and writer [REF anti_stick_diarize.py:33] [REF diarization_baseline.py:65,101] — are not installed, nor is any
FLAC codec: the stems' container and subframe coding are written out in numpy from the format specification)."""
from __future__ import annotations

from math import gcd
from pathlib import Path

import numpy as np
from scipy.io import wavfile
from scipy.signal import resample_poly


def to_float32(x: np.ndarray) -> np.ndarray:
    if x.dtype == np.int16:
        return (x.astype(np.float32) / 32768.0)
    if x.dtype == np.int32:
        return (x.astype(np.float32) / 2147483648.0)
    if x.dtype == np.uint8:
        return ((x.astype(np.float32) - 128.0) / 128.0)
    return x.astype(np.float32)


def read_audio(path, sr: int = 16000, mono: bool = True):
    """-> (float32 [n] (mono) or [channels, n], sr)."""
    path = Path(path)
    if path.suffix.lower() == ".flac":
        from . import flac
        pcm, file_sr, bps = flac.read_flac(path)
        y = (pcm.astype(np.float64) / float(1 << (bps - 1))).astype(np.float32)          # [channels, n]
    elif path.suffix.lower() == ".wav":
        file_sr, data = wavfile.read(str(path))
        y = to_float32(np.asarray(data))
        y = y[None, :] if y.ndim == 1 else y.T                  # [channels, n]
    else:
        raise NotImplementedError(f"{path.suffix} decoding needs a codec that is not installed; convert to WAV or FLAC")
    if file_sr != sr:
        g = gcd(int(file_sr), int(sr))
        y = resample_poly(y, sr // g, file_sr // g, axis=1).astype(np.float32)
    if mono:
        y = y.mean(axis=0)
    return np.ascontiguousarray(y, dtype=np.float32), sr


def write_flac16(path, y: np.ndarray, sr: int) -> None:
    """y: [n] or [channels, n] float -> 16-bit FLAC (`torchaudio.save(..., format="flac", bits_per_sample=16)` in the reference)."""
    from . import flac
    flac.write_flac16(path, y, sr)


def write_wav16(path, y: np.ndarray, sr: int) -> None:
    """y: [n] or [channels, n] float -> 16-bit PCM WAV."""
    y = np.asarray(y, dtype=np.float32)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(str(path), sr, pcm.T if pcm.ndim == 2 else pcm)
