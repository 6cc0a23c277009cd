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
* `cluster_hdbscan`, `cluster_hdbscan_two_stage` — the reference's own glue around a density
                   clusterer [REF anti_stick_diarize.py:175-270], restated over an injectable
                   `clusterer_factory`; the default is the `hdbscan` package when installed, else
                   scikit-learn's HDBSCAN.  Golden-pinned against the reference function itself
                   (tests/golden/cluster_two_stage.json).
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


def default_hdbscan_factory(**kwargs):
    """The density clusterer behind the reference's HDBSCAN call sites: the `hdbscan` package the
    reference imports [REF anti_stick_diarize.py:14] when it is installed, else scikit-learn's
    `sklearn.cluster.HDBSCAN` (the in-tree port of the same algorithm; same keyword names for the
    four arguments the reference passes).  Returns an object with `fit_predict(X) -> labels`."""
    try:
        from hdbscan import HDBSCAN
    except ImportError:
        from sklearn.cluster import HDBSCAN
    return HDBSCAN(**kwargs)


class AhcClusterer:
    """Average-linkage AHC as an injectable stand-in for an HDBSCAN object: `fit_predict` accepts what
    the reference hands its clusterers, i.e. L2-normalised rows (metric "euclidean") or a precomputed
    cosine-distance matrix (metric "precomputed"), and cuts at cosine `cos_thr`
    [REF diar_diag.py:218-226].  Never selected implicitly: pass `clusterer_factory=AhcClusterer.factory(thr)`."""

    def __init__(self, cos_thr: float = 0.70, metric: str = "euclidean", **_ignored):
        self.cos_thr = float(cos_thr)
        self.metric = metric

    @classmethod
    def factory(cls, cos_thr: float = 0.70):
        return lambda **kw: cls(cos_thr, metric=kw.get("metric", "euclidean"))

    def fit_predict(self, X) -> np.ndarray:
        X = np.asarray(X, dtype=np.float64)
        if self.metric == "precomputed":
            return ahc_cosine(1.0 - X, self.cos_thr)
        n = np.linalg.norm(X, axis=1, keepdims=True)
        Xn = X / np.where(n > 0, n, 1.0)
        return ahc_cosine(Xn @ Xn.T, self.cos_thr)


def cluster_hdbscan(embs: np.ndarray, min_cluster_size: int = 2, clusterer_factory=None, affinity=None) -> np.ndarray:
    """[REF anti_stick_diarize.py:175-186]: rows scaled by 1 / (norm + 1e-8), D = 1 - cosine_similarity,
    HDBSCAN(min_cluster_size, min_samples=None, allow_single_cluster=True, metric="precomputed").fit_predict(D).
    `affinity(X) -> K` lets the caller take the N x N product on the GPU (`ops.cosine_affinity`)."""
    embs = np.asarray(embs)
    if embs.shape[0] < 2:            # a density clusterer needs two points; one segment is one speaker
        return np.zeros(embs.shape[0], dtype=int)
    embs_norm = embs / (np.linalg.norm(embs, axis=1, keepdims=True) + 1e-8)
    if affinity is None:
        from sklearn.metrics.pairwise import cosine_similarity as affinity
    D = 1 - np.asarray(affinity(embs_norm))
    factory = clusterer_factory or default_hdbscan_factory
    clu = factory(min_cluster_size=min_cluster_size, min_samples=None, allow_single_cluster=True, metric="precomputed")
    return np.asarray(clu.fit_predict(D))


def hdbscan_precomputed(K, min_cluster_size: int = 2, clusterer_factory=None, min_samples: int | None = None,
                        allow_single_cluster: bool | None = True) -> np.ndarray:
    """HDBSCAN(metric='precomputed') on 1 - K for an affinity that already exists.  `allow_single_cluster=None` leaves the
    clusterer's own default (False), which is what [REF diar_diag.py:214-217] does: HDBSCAN(min_cluster_size=6,
    min_samples=3, metric='precomputed').  Fewer rows than the clusterer can take -> all noise (-1)."""
    D = 1.0 - _as_f64_affinity(K)
    n = D.shape[0]
    if n < 2 or n < (min_samples or min_cluster_size):
        return np.full(n, 0 if allow_single_cluster else -1, dtype=int)
    factory = clusterer_factory or default_hdbscan_factory
    kw = dict(min_cluster_size=min_cluster_size, min_samples=min_samples, metric="precomputed")
    if allow_single_cluster is not None:
        kw["allow_single_cluster"] = allow_single_cluster
    return np.asarray(factory(**kw).fit_predict(D))


def cluster_hdbscan_two_stage(embs: np.ndarray, min_cluster_size: int = 2, clusterer_factory=None) -> np.ndarray:
    """The reference's two-stage clustering glue [REF anti_stick_diarize.py:189-270] over an injectable
    clusterer (`clusterer_factory(**kwargs).fit_predict(X)`; default `default_hdbscan_factory`):

    1. over-cluster the rows scaled by 1 / (norm + 1e-8) (euclidean) into micro-clusters [REF :199-214];
       none found -> everything is speaker 0 [REF :216-218] (an empty float array for zero rows);
    2. centroid = mean of the UN-normalised member rows of each micro-cluster, in label order [REF :220-236];
    3. fewer centroids than `min_cluster_size` -> one speaker, else re-cluster the normalised centroids
       with a second clusterer of the same settings [REF :240-255];
    4. members inherit their centroid's label; members of a noise centroid and stage-1 noise stay -1
       [REF :257-266].
    """
    embs = np.asarray(embs)
    num_segments = embs.shape[0]
    if num_segments < 2:             # nothing to cluster: the reference's "no micro-clusters -> speaker 0" outcome [REF :216-218]
        return np.zeros(num_segments, dtype=int) if num_segments > 0 else np.array([])
    factory = clusterer_factory or default_hdbscan_factory
    settings = dict(min_cluster_size=min_cluster_size, min_samples=None, metric="euclidean", allow_single_cluster=True)
    embs_norm = embs / (np.linalg.norm(embs, axis=1, keepdims=True) + 1e-8)
    stage1 = np.asarray(factory(**settings).fit_predict(embs_norm))
    n_micro = int(np.max(stage1)) + 1
    if n_micro < 1:
        return np.zeros(num_segments, dtype=int) if num_segments > 0 else np.array([])
    member_rows = [np.flatnonzero(stage1 == i) for i in range(n_micro)]
    kept = [(i, rows) for i, rows in enumerate(member_rows) if rows.size]     # labels that actually occur
    if not kept:
        return np.zeros(num_segments, dtype=int)
    centroids = np.array([np.mean(embs[rows], axis=0) for _, rows in kept])
    if len(centroids) < min_cluster_size:
        stage2 = np.zeros(len(centroids), dtype=int)
    else:
        cents_norm = centroids / (np.linalg.norm(centroids, axis=1, keepdims=True) + 1e-8)
        stage2 = np.asarray(factory(**settings).fit_predict(cents_norm))
    final = np.full(num_segments, -1, dtype=int)
    for (_, rows), lab in zip(kept, stage2):
        if lab != -1:
            final[rows] = lab
    return final
