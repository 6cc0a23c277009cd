"""Framing and VAD post-processing that produce the windows the embedding path consumes.

Same public names and signatures as the reference's `vad.py`
(`frame_audio`, `SileroVAD.probs`, `using_silero_vad`, `hysteresis_binarize`,
`morph_open_close`, `mask_to_segments`, `silero_vad_segments`
[REF vad.py:9,19-50,53-55,59-74,77-87,90-163,167-187]).  These are tiny sequential
host computations (360 k frames per hour of audio); they stay on the host, written
here as whole-array numpy instead of per-frame loops.  The Silero network itself is a
remote `torch.hub` download in the reference and out of scope: `SileroVAD` accepts any
frame scorer and ships an energy-based stand-in for synthetic audio.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Callable

import numpy as np
import torch
from scipy import ndimage


def frame_audio(y: np.ndarray, sr: int, win_ms: float = 30.0, hop_ms: float = 10.0) -> np.ndarray:
    """Strided view [n_frames, win] with n_frames = 1 + (len(y) - win) // hop, no padding
    (librosa.util.frame(...).T in the reference, [REF vad.py:9-16])."""
    win = int(round(win_ms / 1000.0 * sr))
    hop = int(round(hop_ms / 1000.0 * sr))
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError(f"frame_audio expects a 1-D signal, got shape {y.shape}")
    if hop < 1:
        raise ValueError(f"Invalid hop_length: {hop}")
    if y.shape[0] < win:
        raise ValueError(f"Input is too short (n={y.shape[0]}) for frame_length={win}")
    return np.lib.stride_tricks.sliding_window_view(y, win)[::hop]


class EnergyScorer:
    """Stand-in frame scorer: logistic of the frame's RMS level relative to a threshold (dBFS)."""

    def __init__(self, threshold_db: float = -35.0, slope: float = 0.6):
        self.threshold_db, self.slope = threshold_db, slope

    def __call__(self, frames: torch.Tensor, sr: int) -> torch.Tensor:
        rms = frames.float().pow(2).mean(dim=1).clamp_min(1e-12).sqrt()
        db = 20.0 * torch.log10(rms)
        return torch.sigmoid(self.slope * (db - self.threshold_db))


class SileroVAD:
    """Batched frame scoring with the reference's loop shape [REF vad.py:31-50].

    `model(frames[B, win], sr) -> probs[B]`.  With `model=None` the reference would fetch
    snakers4/silero-vad through torch.hub; there is no network here, so a scorer must be
    supplied (e.g. `EnergyScorer()`), otherwise construction fails loudly.
    """

    def __init__(self, device: str = "cpu", model: Callable | None = None):
        if model is None:
            raise RuntimeError(
                "Silero VAD weights are a remote torch.hub download [REF vad.py:21-27] and cannot be fetched offline; "
                "pass model=<callable(frames, sr) -> probs> (e.g. vad.EnergyScorer())")
        self.model = model
        if hasattr(model, "to"):
            self.model = model.to(device).eval()
        self.device = device

    @torch.inference_mode()
    def probs(self, y: np.ndarray, sr: int = 16000, win_ms: float = 30.0, hop_ms: float = 10.0,
              batch_size: int = 1024) -> np.ndarray:
        frames = frame_audio(y, sr, win_ms, hop_ms)
        out = np.zeros(frames.shape[0], dtype=np.float32)
        for lo in range(0, frames.shape[0], batch_size):
            fb = torch.from_numpy(np.ascontiguousarray(frames[lo:lo + batch_size])).to(self.device)
            pb = self.model(fb, sr)
            out[lo:lo + len(pb)] = pb.cpu().numpy()
        return out


@lru_cache(maxsize=1)
def using_silero_vad():
    return SileroVAD(model=EnergyScorer())


def hysteresis_binarize(probs: np.ndarray, on: float = 0.6, off: float = 0.4) -> np.ndarray:
    """Two-threshold scan: speech starts at p >= on, ends at p < off [REF vad.py:59-74]."""
    probs = np.asarray(probs)
    n = probs.shape[0]
    if n == 0:
        return np.zeros(probs.shape, dtype=np.bool_)
    if not on > off:
        # degenerate thresholds: a frame can be both a start and a stop event; keep the scan literal
        mask = np.zeros(n, dtype=np.bool_)
        state = False
        for i, p in enumerate(probs):
            if not state and p >= on:
                state = True
            elif state and p < off:
                state = False
            mask[i] = state
        return mask
    # the state after frame i is decided by the most recent start/stop event at or before i
    event = np.where(probs >= on, 1, np.where(probs < off, -1, 0)).astype(np.int8)
    last = np.maximum.accumulate(np.where(event != 0, np.arange(n), -1))
    return (last >= 0) & (event[np.maximum(last, 0)] == 1)


def _line(ms: float, hop_ms: float) -> np.ndarray:
    return np.ones(max(1, int(round(ms / hop_ms))), dtype=bool)


def morph_open_close(mask: np.ndarray, hop_ms: float, open_ms: float = 80.0, close_ms: float = 40.0) -> np.ndarray:
    """1-D binary opening (drops blips) then closing (fills pinholes) [REF vad.py:77-87]."""
    out = mask.copy()
    if open_ms > 0:
        out = ndimage.binary_opening(out, structure=_line(open_ms, hop_ms))
    if close_ms > 0:
        out = ndimage.binary_closing(out, structure=_line(close_ms, hop_ms))
    return out


def mask_to_segments(mask: np.ndarray, hop_ms: float, min_speech_ms: float = 250.0, min_gap_ms: float = 100.0,
                     speech_pad_ms: float = 80.0) -> list[tuple[float, float]]:
    """Boolean frame mask -> [(start_s, end_s)]: drop short runs, THEN bridge short gaps, pad,
    clamp to the signal and round to 3 decimals [REF vad.py:90-163]."""
    mask = np.asarray(mask, dtype=bool)
    if not mask.any():
        return []
    min_run = round(min_speech_ms / hop_ms)
    max_gap = round(min_gap_ms / hop_ms)
    pad = round(speech_pad_ms / hop_ms)
    hop_s = hop_ms / 1000.0
    total = mask.shape[0]

    edges = np.diff(np.concatenate(([0], mask.astype(np.int8), [0])))
    starts = np.flatnonzero(edges == 1)
    ends = np.flatnonzero(edges == -1)
    keep = (ends - starts) >= min_run
    starts, ends = starts[keep], ends[keep]
    if starts.size == 0:
        return []
    # runs are disjoint and ordered, so a run opens a new segment iff its gap to the previous run is too long
    opens = np.concatenate(([True], (starts[1:] - ends[:-1]) > max_gap))
    seg_start = starts[opens]
    seg_end = ends[np.concatenate((np.flatnonzero(opens)[1:] - 1, [starts.size - 1]))]
    # Two rounding regimes [REF vad.py:157-160]: frame indices are numpy integers there, so `index * hop_s` is a numpy
    # float and round() is numpy's scale / rint / unscale -- except for an end that the clamp replaces by len(mask), a
    # Python int, whose product is a Python float rounded correctly to 3 decimals.  The two differ on ties of the
    # scaled value (hop 12.5 ms: 361 * 0.0125 -> 4.513 by Python's round, 4.512 by numpy's).
    s = np.round(np.maximum(seg_start - pad, 0) * hop_s, 3)
    e_pad = seg_end + pad
    e = np.round(np.minimum(e_pad, total) * hop_s, 3)
    if e_pad[-1] > total:
        e[e_pad > total] = round(total * hop_s, 3)
    return [(float(a), float(b)) for a, b in zip(s, e)]


def silero_vad_segments(y: np.ndarray, sr: int = 16000, on_threshold: float = 0.6, off_threshold: float = 0.4,
                        min_speech_ms: float = 250.0, min_silence_ms: float = 100.0, speech_pad_ms: float = 40,
                        win_ms: float = 30.0, hop_ms: float = 10.0, morph_open_ms: float = 80.0,
                        morph_close_ms: float = 40.0, batch_size: int = 512):
    """probabilities -> hysteresis -> open/close -> segments [REF vad.py:167-187]."""
    scorer = using_silero_vad()
    probs = scorer.probs(y, sr, win_ms=win_ms, hop_ms=hop_ms, batch_size=batch_size)
    mask = hysteresis_binarize(probs, on=on_threshold, off=off_threshold)
    mask = morph_open_close(mask, hop_ms, open_ms=morph_open_ms, close_ms=morph_close_ms)
    return mask_to_segments(mask, hop_ms, min_speech_ms=min_speech_ms, min_gap_ms=min_silence_ms,
                            speech_pad_ms=speech_pad_ms)
