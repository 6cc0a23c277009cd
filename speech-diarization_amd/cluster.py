"""Host-side consumers of the cosine affinity: clustering into speakers.

The affinity itself is computed on the GPU (`ops.cosine_affinity`, sklearn
`cosine_similarity` semantics); clustering is global, sequential and tiny next to the
embedding work, and stays on the host with scikit-learn exactly as the reference does:

* `ahc_cosine`   — AgglomerativeClustering(n_clusters=None, linkage="average",
                   metric="precomputed", distance_threshold=1 - cos_thr) on 1 - K
                   [REF diar_diag.py:218-226]; threshold default 0.70 is the reference's
                   clustering threshold [REF diarization_baseline.py:37,248].
* `spectral`     — SpectralClustering(affinity="precomputed") on max(K, 0)
                   (BASELINE.json configs[0] asks for spectral clustering).
* `hdbscan_*`    — the reference's HDBSCAN variants [REF anti_stick_diarize.py:175-270]
                   need the `hdbscan` package, which is not installed here: gated.
"""
from __future__ import annotations

import numpy as np


def center(embs: np.ndarray) -> np.ndarray:
    """Remove the recording-level mean embedding (the first step of the reference's `whiten_l2`)."""
    embs = np.asarray(embs)
    return embs - embs.mean(axis=0, keepdims=True) if embs.shape[0] else embs


def whiten_l2(embs: np.ndarray) -> np.ndarray:
    """Centre, whiten with the inverse square root of the covariance (+1e-6), L2-normalise rows
    [REF diar_diag.py:187-194]."""
    X = center(np.asarray(embs, dtype=np.float64))
    S, U = np.linalg.eigh(np.atleast_2d(np.cov(X.T)))
    Xw = X @ ((U / np.sqrt(np.clip(S, 0.0, None) + 1e-6)) @ U.T)
    return Xw / (np.linalg.norm(Xw, axis=1, keepdims=True) + 1e-9)


def _as_f64_affinity(K) -> np.ndarray:
    K = np.asarray(K.detach().cpu().numpy() if hasattr(K, "detach") else K, dtype=np.float64)
    if K.ndim != 2 or K.shape[0] != K.shape[1]:
        raise ValueError(f"affinity must be square, got {K.shape}")
    return K


def ahc_cosine(K, cos_thr: float = 0.70) -> np.ndarray:
    """Average-linkage agglomerative clustering on D = 1 - K, cut at 1 - cos_thr."""
    from sklearn.cluster import AgglomerativeClustering
    K = _as_f64_affinity(K)
    n = K.shape[0]
    if n == 0:
        return np.zeros(0, dtype=int)
    if n == 1:
        return np.zeros(1, dtype=int)
    D = 1.0 - K
    D = 0.5 * (D + D.T)                       # exact symmetry for the linkage routine
    np.fill_diagonal(D, 0.0)
    np.clip(D, 0.0, None, out=D)
    model = AgglomerativeClustering(n_clusters=None, linkage="average", metric="precomputed",
                                    distance_threshold=1.0 - cos_thr)
    return model.fit_predict(D)


def spectral(K, n_speakers: int, random_state: int = 0) -> np.ndarray:
    from sklearn.cluster import SpectralClustering
    K = _as_f64_affinity(K)
    n = K.shape[0]
    if n == 0:
        return np.zeros(0, dtype=int)
    if n_speakers <= 1 or n <= n_speakers:
        return np.zeros(n, dtype=int) if n_speakers <= 1 else np.arange(n)
    A = np.clip(0.5 * (K + K.T), 0.0, None)
    np.fill_diagonal(A, 1.0)
    model = SpectralClustering(n_clusters=n_speakers, affinity="precomputed", assign_labels="kmeans",
                               random_state=random_state)
    return model.fit_predict(A)


def estimate_num_speakers(K, min_speakers: int, max_speakers: int) -> int:
    """Eigengap of the normalised Laplacian of max(K, 0), clamped to [min, max]."""
    K = _as_f64_affinity(K)
    n = K.shape[0]
    lo, hi = max(1, min_speakers), max(1, min(max_speakers, n))
    if hi <= lo:
        return min(lo, hi) if n >= lo else max(1, n)
    A = np.clip(0.5 * (K + K.T), 0.0, None)
    d = A.sum(1)
    d[d <= 0] = 1.0
    L = np.eye(n) - A / np.sqrt(d[:, None] * d[None, :])
    ev = np.sort(np.linalg.eigvalsh(L))[: hi + 1]
    gaps = np.diff(ev)
    k = int(np.argmax(gaps[lo - 1: hi]) + lo)
    return k


def relabel_by_first_appearance(labels: np.ndarray) -> np.ndarray:
    """Canonical label names: speakers numbered in order of first appearance (noise -1 kept)."""
    labels = np.asarray(labels)
    out = np.full(labels.shape, -1, dtype=int)
    mapping: dict[int, int] = {}
    for i, lab in enumerate(labels.tolist()):
        if lab < 0:
            continue
        if lab not in mapping:
            mapping[lab] = len(mapping)
        out[i] = mapping[lab]
    return out


def _hdbscan_cls():
    try:
        from hdbscan import HDBSCAN
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ImportError("the `hdbscan` package is not installed; use cluster.ahc_cosine / cluster.spectral") from e
    return HDBSCAN


def hdbscan_precomputed(K, min_cluster_size: int = 2) -> np.ndarray:
    """[REF anti_stick_diarize.py:175-186]: HDBSCAN(metric='precomputed') on 1 - K."""
    HDBSCAN = _hdbscan_cls()
    D = 1.0 - _as_f64_affinity(K)
    return HDBSCAN(min_cluster_size=min_cluster_size, min_samples=None, allow_single_cluster=True,
                   metric="precomputed").fit_predict(D)
