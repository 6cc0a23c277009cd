"""Minimal audio I/O for the entry points: WAV in / WAV out with scipy (torchaudio, librosa and
soundfile — the reference's readers [REF anti_stick_diarize.py:33] [REF diarization_baseline.py:65] —
are not installed; FLAC needs a codec that is absent, so stems are written as 16-bit WAV)."""
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
    if path.suffix.lower() != ".wav":
        raise NotImplementedError(f"{path.suffix} decoding needs a codec that is not installed; convert to WAV")
    file_sr, data = wavfile.read(str(path))
    y = to_float32(np.asarray(data))
    y = y[None, :] if y.ndim == 1 else y.T                      # [channels, n]
    if file_sr != sr:
        g = gcd(int(file_sr), int(sr))
        y = resample_poly(y, sr // g, file_sr // g, axis=1).astype(np.float32)
    if mono:
        y = y.mean(axis=0)
    return np.ascontiguousarray(y, dtype=np.float32), sr


def write_wav16(path, y: np.ndarray, sr: int) -> None:
    """y: [n] or [channels, n] float -> 16-bit PCM WAV."""
    y = np.asarray(y, dtype=np.float32)
    pcm = np.clip(np.round(y * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(str(path), sr, pcm.T if pcm.ndim == 2 else pcm)
