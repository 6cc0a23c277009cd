"""Score normalisation, smoothing and export helpers of the reference's diagnostic tool
(`diar_diag.py`), the consumers of the affinity / centroid scores.

Same function names and semantics as [REF diar_diag.py:187-208,213-229,231-247,252-272]; the
diagnostic CLI, plots and the tool's private VAD copy are out of scope.  The heavy inputs (window
embeddings, the cohort similarity matrices of AS-norm) come from the GPU path; these reductions are
tiny and stay on the host.
"""
from __future__ import annotations

import csv
import json

import numpy as np

from .cluster import ahc_cosine, hdbscan_precomputed, whiten_l2  # noqa: F401  (whiten_l2 re-exported)


def _l2n(x: np.ndarray) -> np.ndarray:
    return x / (np.linalg.norm(x, axis=-1, keepdims=True) + 1e-9)


def asnorm_scores(query_embs: np.ndarray, ref_centers: np.ndarray, cohort_embs: np.ndarray, topk: int = 200, device=None) -> np.ndarray:
    """Adaptive symmetric score normalisation: cosine scores z-normalised against the top-k cohort
    scores of the query and of the reference, averaged [REF diar_diag.py:196-208].
    `device="cuda"`: the cohort products, the top-k statistics and the combination run on the GPU
    (`ops.asnorm_scores`, f32); default: the reference's numpy arithmetic in the input dtype."""
    if device is not None:
        import torch
        from . import ops
        dev = torch.device(device)
        t = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in (query_embs, ref_centers, cohort_embs)]
        return ops.asnorm_scores(t[0], t[1], t[2], topk).cpu().numpy()
    Q, R, Cc = _l2n(np.asarray(query_embs)), _l2n(np.asarray(ref_centers)), _l2n(np.asarray(cohort_embs))
    raw = Q @ R.T
    k = min(topk, Cc.shape[0])
    qc = np.sort(Q @ Cc.T, axis=1)[:, -k:]
    rc = np.sort(R @ Cc.T, axis=1)[:, -k:]
    zq = (raw - qc.mean(axis=1, keepdims=True)) / (qc.std(axis=1, keepdims=True) + 1e-6)
    zr = (raw - rc.mean(axis=1, keepdims=True).T) / (rc.std(axis=1, keepdims=True).T + 1e-6)
    return 0.5 * (zq + zr)


def cluster_embeddings(embs: np.ndarray, method: str = "hdbscan", cos_thr: float = 0.68, affinity=None, clusterer_factory=None) -> np.ndarray:
    """"agglo": average-linkage AHC on 1 - cosine cut at 1 - cos_thr; "hdbscan": HDBSCAN(min_cluster_size=6, min_samples=3,
    metric="precomputed") on 1 - cosine, `allow_single_cluster` left at the clusterer's default [REF diar_diag.py:213-229]
    (`cluster.default_hdbscan_factory` unless `clusterer_factory` is given).  The N x N cosine runs on the GPU
    (`ops.cosine_affinity`) unless `affinity(embs) -> K` is injected (CPU tests)."""
    if method not in ("hdbscan", "agglo"):
        raise ValueError("method must be 'hdbscan' or 'agglo'")
    if affinity is None:
        import torch
        from . import ops
        K = ops.cosine_affinity(torch.from_numpy(np.ascontiguousarray(embs, dtype=np.float32)).cuda()).cpu().numpy()
    else:
        K = np.asarray(affinity(embs))
    if method == "agglo":
        return ahc_cosine(K, cos_thr)
    return hdbscan_precomputed(K, min_cluster_size=6, clusterer_factory=clusterer_factory, min_samples=3, allow_single_cluster=None)


def viterbi_hmm(scores: np.ndarray, alpha: float = 0.995, device=None) -> np.ndarray:
    """Most likely speaker path through per-window log-scores [T, K] under a sticky transition matrix
    (stay alpha, move (1-alpha)/(K-1)), float32 arithmetic as in [REF diar_diag.py:231-247].
    `device="cuda"`: the same recurrence in one wave on the GPU (`ops.viterbi`, K <= 64, f32 scores)."""
    if device is not None:
        import torch
        from . import ops
        sc = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32)).to(torch.device(device))
        return ops.viterbi(sc, alpha).cpu().numpy()
    scores = np.asarray(scores)
    T, K = scores.shape
    eps = 1e-8
    logA = np.full((K, K), np.log((1 - alpha) / (K - 1) + eps) if K > 1 else 0.0, dtype=np.float32)
    np.fill_diagonal(logA, np.log(alpha + eps))
    dp = np.full((T, K), -1e9, dtype=np.float32)
    back = np.zeros((T, K), dtype=np.int32)
    dp[0] = scores[0]
    cols = np.arange(K)
    for t in range(1, T):
        cand = dp[t - 1][:, None] + logA
        back[t] = np.argmax(cand, axis=0)
        dp[t] = cand[back[t], cols] + scores[t]
    path = np.zeros(T, dtype=np.int32)
    path[-1] = int(np.argmax(dp[-1]))
    for t in range(T - 2, -1, -1):
        path[t] = back[t + 1, path[t + 1]]
    return path


def save_json(out_path: str, segments: list[dict], speakers: list[str]) -> None:
    with open(out_path, "w", encoding="utf-8") as f:
        json.dump({"segments": segments, "speakers": speakers}, f, ensure_ascii=False, indent=2)


def _srt_time(ts: float) -> str:
    h = int(ts // 3600)
    ts -= h * 3600
    m = int(ts // 60)
    ts -= m * 60
    s = int(ts)
    return f"{h:02d}:{m:02d}:{s:02d},{int(round((ts - s) * 1000)):03d}"


def save_srt(out_path: str, segments: list[dict]) -> None:
    with open(out_path, "w", encoding="utf-8") as f:
        for i, seg in enumerate(segments, 1):
            f.write(f"{i}\n{_srt_time(seg['start'])} --> {_srt_time(seg['end'])}\n{seg['speaker']}\n\n")


def save_csv(out_path: str, segments: list[dict]) -> None:
    with open(out_path, "w", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=["start", "end", "speaker"])
        w.writeheader()
        for seg in segments:
            w.writerow({k: seg[k] for k in ("start", "end", "speaker")})
