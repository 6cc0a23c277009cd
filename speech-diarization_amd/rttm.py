"""RTTM output schema and a frame-level DER scorer.

The reference writes RTTM through pyannote's `Annotation.write_rttm`
[REF diarization_baseline.py:263-265] — one line per turn:

    SPEAKER <uri> 1 <start:.3f> <duration:.3f> <NA> <NA> <label> <NA> <NA>

with labels `SPEAKER_00`, `SPEAKER_01`, ... and segment tuples `(start_s, end_s, speaker)`
[REF diarization_baseline.py:259-261].  The reference has no DER code; `der()` exists so the
GPU path's RTTM can be scored against the CPU path's (expected 0.0 when assignments agree).
"""
from __future__ import annotations

import itertools
from pathlib import Path

import numpy as np


def speaker_label(index: int) -> str:
    return f"SPEAKER_{int(index):02d}"


def write_rttm(segments, uri: str, f) -> None:
    """segments: iterable of (start_s, end_s, speaker) -> RTTM lines on file object `f`."""
    for start, end, spk in segments:
        label = spk if isinstance(spk, str) else speaker_label(spk)
        f.write(f"SPEAKER {uri} 1 {float(start):.3f} {float(end) - float(start):.3f} <NA> <NA> {label} <NA> <NA>\n")


def read_rttm(path) -> list[tuple[float, float, str]]:
    out = []
    for line in Path(path).read_text().splitlines():
        parts = line.split()
        if len(parts) < 8 or parts[0] != "SPEAKER":
            continue
        start, dur = float(parts[3]), float(parts[4])
        out.append((start, start + dur, parts[7]))
    return out


def _frame_labels(segments, names: list, n_frames: int, step: float) -> np.ndarray:
    """[n_frames, n_speakers] bool activity matrix."""
    act = np.zeros((n_frames, len(names)), dtype=bool)
    idx = {n: i for i, n in enumerate(names)}
    for s, e, spk in segments:
        a, b = int(round(s / step)), int(round(e / step))
        act[max(a, 0):min(b, n_frames), idx[spk]] = True
    return act


def der(reference, hypothesis, step: float = 0.01, max_permute: int = 8) -> float:
    """Diarization error rate = (miss + false alarm + confusion) / reference speech time, no collar,
    optimal speaker mapping (exhaustive for <= max_permute speakers, greedy beyond)."""
    if not reference:
        return 0.0 if not hypothesis else float("inf")
    end = max([e for _, e, _ in reference] + [e for _, e, _ in hypothesis])
    n = int(np.ceil(end / step)) + 1
    rn = sorted({s for _, _, s in reference}, key=str)
    hn = sorted({s for _, _, s in hypothesis}, key=str)
    R = _frame_labels(reference, rn, n, step)
    H = _frame_labels(hypothesis, hn, n, step) if hn else np.zeros((n, 0), dtype=bool)
    overlap = R.T.astype(np.int64) @ H.astype(np.int64) if hn else np.zeros((len(rn), 0), dtype=np.int64)
    k = max(len(rn), len(hn))
    cost = np.zeros((k, k), dtype=np.int64)
    cost[: len(rn), : len(hn)] = overlap
    if k <= max_permute:
        best = max(sum(cost[i, p[i]] for i in range(k)) for p in itertools.permutations(range(k)))
    else:
        from scipy.optimize import linear_sum_assignment
        r, c = linear_sum_assignment(-cost)
        best = int(cost[r, c].sum())
    n_ref = R.sum(1)
    n_hyp = H.sum(1) if hn else np.zeros(n, dtype=np.int64)
    total = int(n_ref.sum())
    miss = int(np.maximum(n_ref - n_hyp, 0).sum())
    fa = int(np.maximum(n_hyp - n_ref, 0).sum())
    correct = best
    confusion = int(np.minimum(n_ref, n_hyp).sum()) - correct
    return (miss + fa + confusion) / max(total, 1)
