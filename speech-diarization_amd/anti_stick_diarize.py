"""The callers either side of the embedding path: batching policy, cosine sites, segment glue.

Function names, signatures and results follow the reference's `anti_stick_diarize.py`
(`Segment`, `scd_split_segments` [REF :78-127], `embed_segments` [REF :130-172],
`conservative_merge` [REF :273-330], `speaker_centroids` [REF :333-349],
`cluster_hdbscan` / `cluster_hdbscan_two_stage` [REF :175-270], `_get_speech_windows` [REF :352-367], `frame_reassign` [REF :390-460], `merge_adjacent`
[REF :464-475], `diar_read_audio` / `loudness_normalize` [REF :29-61], `diarize` [REF :493-560], `main` [REF :563-604]).
What changes is where the work runs:

* every function that embeds takes an optional `encode` callable (`wavs[B, n] -> [B, 192]`);
  the default is the HIP encoder (`speech_encode.ecapa_encode_batch`).  Tests inject a
  deterministic stand-in to pin the batching policy against the reference itself.
* fixed-length windows (SCD, reassignment) are embedded in large batches rather than per
  segment / per 128 — rows are independent, so the embeddings are the same;
  variable-length segments keep the reference's batches of 32 padded to the batch max,
  because zero padding counts as signal and batch composition therefore changes the result.
* the cosine sites (adjacent-pair, windows x centroids + argmax) run on the GPU when the
  default encoder is in use.

Known defects of the reference that this module does NOT reproduce by default
(SURVEY.md Appendix B): labels passed where embeddings are expected (B-1, available as
`compat_reference_bugs=True`), `np.array(dict.keys())` (B-3), duplicate label->segment pass (B-5).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import Callable

import numpy as np
from scipy.signal import find_peaks

from .vad import frame_audio, silero_vad_segments

Encoder = Callable[[np.ndarray], np.ndarray]
EMB_DIM = 192


@dataclass
class Segment:
    start: float
    end: float
    spk: int | None = None
    score: float | None = None


# --------------------------------------------------------------------------- loading + conditioning

def loudness_normalize(y: np.ndarray, sr: int, target_lufs: float = -18.0) -> np.ndarray:
    """Integrated-loudness normalisation to `target_lufs`, clipped to +-0.99 [REF anti_stick_diarize.py:53-61].  The meter
    is the reference's own dependency when it is installed (`pyloudnorm`, ITU-R BS.1770) and the restatement of its
    algorithm in `loudness.py` otherwise -- the step is never skipped: the reference applies it unconditionally, and
    VAD thresholds, SCD and the encoder all see the gain and the clip.  Only the absence of the `pyloudnorm` package
    itself selects the restatement; an ImportError from inside it (say a missing scipy) propagates."""
    try:
        import pyloudnorm as pyln
    except ModuleNotFoundError as e:
        if e.name != "pyloudnorm":
            raise
        from . import loudness
        y = loudness.normalize_loudness(y, loudness.Meter(sr).integrated_loudness(y), target_lufs)
    else:
        y = pyln.normalize.loudness(y, pyln.Meter(sr).integrated_loudness(y), target_lufs)
    return np.clip(y, -0.99, 0.99)


def diar_read_audio(path_wav, sr: int = 16000, lufs: float | None = -18.0):
    """-> (conditioned mono float32 signal, sr) [REF anti_stick_diarize.py:29-50]: a path is read and resampled
    (`audio_io.read_audio`: WAV; polyphase resampling where the reference uses librosa's kaiser_fast), an
    `(array, orig_sr)` tuple is transposed when it is [n, <= 2], resampled and mixed down; then loudness normalisation
    when `lufs` is not None (always: `loudness_normalize` carries its own BS.1770 meter for hosts without pyloudnorm;
    pass `lufs=None` to opt out, as in the reference), DC removal, and pre-emphasis 0.97 with librosa's initial state (`zi = 2 x[0] - x[1]` handed to `lfilter`
    as it is [UPSTREAM-RECALLED librosa.effects.preemphasis], i.e. y[0] = 3 x[0] - x[1], y[n] = x[n] - 0.97 x[n-1])."""
    from . import audio_io
    if isinstance(path_wav, (str, Path)):
        wav, sr = audio_io.read_audio(path_wav, sr=sr, mono=True)
    else:
        wav, orig_sr = path_wav
        wav = np.asarray(wav, dtype=np.float32)
        if wav.ndim == 2 and wav.shape[1] <= 2:
            wav = wav.T
        if orig_sr != sr:
            from scipy.signal import resample_poly
            g = math.gcd(int(orig_sr), int(sr))
            wav = resample_poly(wav, sr // g, int(orig_sr) // g, axis=-1).astype(np.float32)
        if wav.ndim == 2:
            wav = wav.mean(axis=0)
    if lufs is not None:
        wav = loudness_normalize(wav, sr, target_lufs=lufs)
    wav = np.asarray(wav, dtype=np.float32)
    if wav.size == 0:
        return wav, sr
    wav = wav - np.mean(wav)
    out = np.empty_like(wav)
    out[1:] = wav[1:] - np.float32(0.97) * wav[:-1]
    out[0] = wav[0] + (2.0 * wav[0] - (wav[1] if wav.size > 1 else wav[0]))
    return out.astype(np.float32), sr


def _default_encode() -> Encoder:
    from .speech_encode import ecapa_encode_batch
    return ecapa_encode_batch


def _on_gpu(encode: Encoder | None) -> bool:
    return encode is None


# --------------------------------------------------------------------------- cosine sites

def adjacent_cosine(embs: np.ndarray, use_gpu: bool) -> np.ndarray:
    """<e_i, e_{i+1}> / (|e_i| |e_{i+1}| + 1e-8) [REF anti_stick_diarize.py:102-104]."""
    if use_gpu:
        import torch
        from . import ops
        return ops.adjacent_cosine(torch.from_numpy(np.ascontiguousarray(embs, dtype=np.float32)).cuda()).cpu().numpy()
    a, b = embs[:-1], embs[1:]
    return np.einsum("id,id->i", a, b) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-8)


def _encode_rows(snippets: np.ndarray, encode: Encoder, rows_per_call: int) -> np.ndarray:
    if snippets.shape[0] == 0:
        return np.empty((0, EMB_DIM), dtype=np.float32)
    parts = [encode(np.ascontiguousarray(snippets[lo:lo + rows_per_call]))
             for lo in range(0, snippets.shape[0], rows_per_call)]
    return np.concatenate(parts, axis=0)


# --------------------------------------------------------------------------- SCD

def scd_split_segments(y: np.ndarray, sr: int, segments: list[Segment], win_ms: float = 1000.0, hop_ms: float = 200.0,
                       thr: float = 1.25, min_speech_ms: float = 1000.0, encode: Encoder | None = None,
                       rows_per_call: int = 2048) -> list[Segment]:
    """Split long VAD segments where the speaker embedding changes: sliding-window embeddings,
    adjacent cosine distance, z-score, peaks above `thr`; keep pieces >= min_speech_ms."""
    assert y.ndim == 1 and y.dtype == np.float32
    use_gpu = _on_gpu(encode)
    encode = encode or _default_encode()
    min_speech_s = min_speech_ms / 1000.0

    # pass 1: window every segment; all windows share one length, so they embed in common batches
    views, counts, starts = [], [], []
    win = int(round(win_ms / 1000.0 * sr))
    hop = int(round(hop_ms / 1000.0 * sr))
    for seg in segments:
        a = int(seg.start * sr)
        sub = y[a: int(seg.end * sr)]
        snips = frame_audio(sub, sr, win_ms=win_ms, hop_ms=hop_ms) if sub.shape[0] >= win else np.empty((0, win), np.float32)
        if len(snips) < 3:
            counts.append(0)
            continue
        views.append(snips)
        counts.append(len(snips))
        starts.append(a + hop * np.arange(len(snips), dtype=np.int64))      # where frame_audio's rows sit in y
    if not views:
        embs_all = None
    elif use_gpu:       # the signal goes up once and the windows are read in place (no 5x overlapping host gather)
        from .speech_encode import using_ecapa_encoder
        embs_all = using_ecapa_encoder().encode_windows(y, np.concatenate(starts), win, rows_per_call=rows_per_call)
    else:
        embs_all = _encode_rows(np.concatenate(views, axis=0), encode, rows_per_call)

    # pass 2: per-segment change detection
    out: list[Segment] = []
    off = 0
    for seg, n in zip(segments, counts):
        if n == 0:
            out.append(seg)
            continue
        embs = embs_all[off: off + n]
        off += n
        dists = 1 - adjacent_cosine(embs, use_gpu)
        z = (dists - dists.mean()) / dists.std() if np.std(dists) > 1e-6 else dists
        peaks, _ = find_peaks(z, height=thr)
        if peaks.size == 0:
            out.append(seg)
            continue
        cuts = sorted(set(seg.start + (peaks + 0.5) * hop_ms / 1000.0))
        last = seg.start
        for cut in cuts:
            if cut - last >= min_speech_s:
                out.append(Segment(last, cut))
                last = cut
        if seg.end - last >= min_speech_s:
            out.append(Segment(last, seg.end))
    return out


# --------------------------------------------------------------------------- segment embeddings

def embed_segments(y: np.ndarray, sr: int, segs: list[Segment], batch_size: int = 32, min_duration_ms: float = 500.0,
                   pad_duration_ms: float = 150.0, encode: Encoder | None = None) -> np.ndarray:
    """One embedding per segment -> (num_segments, 192).  Batches of `batch_size` consecutive
    segments, zero-padded to the batch's longest; segments shorter than `min_duration_ms` are
    widened by `pad_duration_ms` on both sides first."""
    if len(segs) == 0:
        return np.empty((0, EMB_DIM), dtype=np.float32)
    min_len = int(min_duration_ms / 1000.0 * sr)
    widen = int(pad_duration_ms / 1000.0 * sr)

    def snippet(seg: Segment) -> np.ndarray:
        s, e = int(seg.start * sr), int(seg.end * sr)
        piece = y[s:e]
        if piece.shape[0] < min_len:
            piece = y[max(0, s - widen): min(len(y), e + widen)]
        return piece

    batches = []
    for lo in range(0, len(segs), batch_size):
        pieces = [snippet(seg) for seg in segs[lo: lo + batch_size]]
        batch = np.zeros((len(pieces), max(len(p) for p in pieces)), dtype=np.float32)
        for row, piece in zip(batch, pieces):
            row[: len(piece)] = piece
        batches.append(batch)
    if encode is None:
        # the batches are independent and each keeps its own padded length [REF :163-166]: two of them in flight on the card
        # (copies under compute, one launch's tail under the other's kernels), every batch bit for bit what `ecapa_encode_batch` returns
        from .speech_encode import ecapa_encode_batches
        chunks = ecapa_encode_batches(batches)
    else:
        chunks = [encode(b) for b in batches]
    return np.concatenate(chunks, axis=0)


# --------------------------------------------------------------------------- merging

def _unit(v: np.ndarray) -> np.ndarray:
    return v / (np.linalg.norm(v) + 1e-8)


def conservative_merge(segs: list[Segment], embs: np.ndarray, max_gap_s: float = 0.5, max_turn_s: float = 30.0,
                       min_cos: float = 0.80) -> list[Segment]:
    """Chain-merge time-ordered neighbours of one speaker when the gap, the merged length and the
    cosine of their embeddings allow; a merged turn carries the normalised sum of its parts."""
    if not segs:
        return []
    order = sorted(range(len(segs)), key=lambda i: (segs[i].start, segs[i].end))
    kept: list[tuple[Segment, np.ndarray]] = []
    for i in order:
        seg, emb = segs[i], embs[i]
        if kept:
            prev, prev_emb = kept[-1]
            eligible = (seg.spk == prev.spk and seg.start - prev.end <= max_gap_s and seg.end - prev.start <= max_turn_s)
            if eligible and np.dot(_unit(prev_emb), _unit(emb)) >= min_cos:
                prev.end = seg.end
                kept[-1] = (prev, _unit(prev_emb + emb))
                continue
        kept.append((seg, emb))
    return [s for s, _ in kept]


def speaker_centroids(segs: list[Segment], embs: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(speaker ids [K], unit-norm centroid matrix [K, 192]) over segments with spk >= 0."""
    labels = np.array([(-1 if s.spk is None else s.spk) for s in segs])
    ids = np.array(sorted({int(v) for v in labels if v >= 0}), dtype=int)
    if ids.size == 0:
        return np.empty(0, dtype=int), np.empty((0, EMB_DIM), dtype=np.float32)
    cents = np.stack([_unit(np.mean(embs[labels == sid], axis=0)) for sid in ids])
    return ids, cents


def _get_speech_windows(y: np.ndarray, sr: int, speech_mask: list[Segment], win_samples: int, step_samples: int):
    """Start samples of every sliding window, and the indices of those whose CENTRE falls in speech
    (10 ms speech raster) [REF anti_stick_diarize.py:352-367]."""
    hop_s = 0.01
    n_frames = math.ceil(len(y) / sr / hop_s)
    raster = np.zeros(n_frames, dtype=bool)
    for seg in speech_mask:
        raster[int(seg.start / hop_s): int(seg.end / hop_s)] = True
    starts = np.arange(0, len(y) - win_samples, step_samples)
    centres = np.clip((((starts + win_samples / 2) / sr) / hop_s).astype(int), 0, n_frames - 1)
    return starts, np.where(raster[centres])[0]


def _labels_to_segments(window_starts: np.ndarray, valid_indices: np.ndarray, window_labels: np.ndarray, sr: int,
                        max_t: float) -> list[Segment]:
    """Run-length encode per-window labels (-1 = non-speech) into Segments; a run ends where the next begins."""
    timeline = np.full(len(window_starts), -1, dtype=int)
    timeline[valid_indices] = window_labels
    if timeline.size == 0:
        return []
    bounds = np.flatnonzero(np.concatenate(([True], timeline[1:] != timeline[:-1])))
    ends = np.concatenate((bounds[1:], [len(timeline)]))
    out = []
    for a, b in zip(bounds, ends):
        spk = int(timeline[a])
        if spk == -1:
            continue
        t0 = window_starts[a] / sr
        t1 = window_starts[b] / sr if b < len(window_starts) else max_t
        out.append(Segment(t0, t1, spk))
    return out


def merge_adjacent(segments: list[Segment], gap: float = 0.05) -> list[Segment]:
    """Fuse consecutive segments of one speaker separated by at most `gap` seconds."""
    merged: list[Segment] = []
    for seg in segments:
        if merged and merged[-1].spk == seg.spk and (seg.start - merged[-1].end) <= gap:
            merged[-1] = Segment(merged[-1].start, seg.end, seg.spk)
        else:
            merged.append(seg)
    return merged


# --------------------------------------------------------------------------- frame-level reassignment

def frame_reassign(y: np.ndarray, sr: int, speech_mask: list[Segment], segs: list[Segment], embs: np.ndarray,
                   smooth_step: float = 0.1, win: float = 1.0, batch_size: int = 128,
                   encode: Encoder | None = None) -> list[Segment]:
    """Re-label speech with 1 s windows every `smooth_step` s: each window goes to the speaker whose
    centroid is closest in cosine, runs of equal labels become segments [REF anti_stick_diarize.py:390-460].

    With the default (HIP) encoder the signal is uploaded once, windows are gathered, embedded,
    normalised and matched against the centroids on the GPU; only the labels come back.
    """
    if not segs or embs.size == 0:
        return []
    spk_ids, c_matrix = speaker_centroids(segs, embs)
    if c_matrix.size == 0:
        return segs
    win_samples = int(win * sr)
    step_samples = int(smooth_step * sr)
    window_starts, valid = _get_speech_windows(y, sr, speech_mask, win_samples, step_samples)
    if valid.size == 0:
        return segs

    if _on_gpu(encode):
        best = _assign_windows_gpu(y, window_starts[valid], win_samples, c_matrix)
    else:
        gather = window_starts[valid][:, None] + np.arange(win_samples)[None, :]
        w = _encode_rows(y[gather], encode, max(batch_size, 1))
        w = w / (np.linalg.norm(w, axis=1, keepdims=True) + 1e-8)
        best = np.argmax(np.dot(w, c_matrix.T), axis=1)
    window_labels = spk_ids[best]
    refined = _labels_to_segments(window_starts, valid, window_labels, sr, len(y) / sr)
    return merge_adjacent(refined, gap=0.05)


def _assign_windows_gpu(y: np.ndarray, starts: np.ndarray, win_samples: int, c_matrix: np.ndarray,
                        rows_per_call: int = 4096) -> np.ndarray:
    import torch
    from . import ops
    from .speech_encode import using_ecapa_encoder
    enc = using_ecapa_encoder()
    dev = enc.device
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).to(dev)
    cd = torch.from_numpy(np.ascontiguousarray(c_matrix, dtype=np.float32)).to(dev)
    sd = torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int64)).to(dev)
    best = []
    with torch.inference_mode():
        for lo in range(0, sd.numel(), rows_per_call):
            e = enc.engine.embed_windows(yd, sd[lo: lo + rows_per_call], win_samples)     # windows read in place: no gathered copy
            idx, _ = ops.sim_argmax(ops.l2norm_rows(e, eps_add=1e-8), cd)
            best.append(idx)
    return torch.cat(best).cpu().numpy().astype(np.int64)


# --------------------------------------------------------------------------- orchestration

def diarize(wav_path, sr: int = 16000, target_lufs: float = -18.0, vad_on_thr: float = 0.6, vad_off_thr: float = 0.4,
            min_speech_ms: float = 250, min_silence_ms: float = 100, speech_pad_ms: float = 70.0, morph_open_ms: float = 80.0,
            morph_close_ms: float = 40.0, scd_win_ms: float = 1000.0, scd_hop_ms: float = 200, scd_thr: float = 1.50,
            merge_max_gap_s: float = 0.5, merge_max_speech_s: float = 30.0, merge_mincos: float = 0.8, reseg: int = 1, *,
            cluster_cos: float = 0.70, encode: Encoder | None = None, vad_segments: Callable | None = None,
            compat_reference_bugs: bool = False, clusterer: str | Callable = "hdbscan_two_stage") -> list[Segment]:
    """VAD -> SCD split -> embed -> cluster -> conservative merge -> re-embed -> frame reassignment ->
    merge_adjacent: the stage order and the positional parameters of [REF anti_stick_diarize.py:493-511] (the 17
    parameters up to `reseg`, same order and defaults; what this build adds is keyword-only).

    `wav_path`: a path or an `(array, sample_rate)` tuple, loaded and conditioned by `diar_read_audio(wav_path, sr,
    lufs=target_lufs)` as in the reference [REF :512]; additionally a bare 1-d numpy array, taken as an already loaded and
    conditioned mono signal at `sr` (no loudness / DC / pre-emphasis step).

    `clusterer` selects what sits inside the reference's `cluster_hdbscan_two_stage(embs, min_cluster_size=2)`
    [REF :536]: "hdbscan_two_stage" (default: the two-stage glue over `cluster.default_hdbscan_factory`),
    "ahc" (the same glue with average-linkage AHC cut at `cluster_cos` injected as the clusterer), "ahc_affinity"
    (single-stage AHC on the GPU cosine affinity), or a `clusterer_factory(**kwargs)` callable."""
    from . import cluster
    if isinstance(wav_path, np.ndarray):
        y = np.ascontiguousarray(wav_path, dtype=np.float32)
    else:
        y, sr = diar_read_audio(wav_path, sr, lufs=target_lufs)
        y = np.ascontiguousarray(y, dtype=np.float32)
    vad_fn = vad_segments or silero_vad_segments
    speech_t = vad_fn(y, sr, on_threshold=vad_on_thr, off_threshold=vad_off_thr, min_speech_ms=min_speech_ms,
                      min_silence_ms=min_silence_ms, speech_pad_ms=speech_pad_ms, morph_open_ms=morph_open_ms,
                      morph_close_ms=morph_close_ms)
    if not speech_t:
        return []
    speech = [Segment(s, e) for s, e in speech_t]
    speech2 = scd_split_segments(y, sr, speech, win_ms=scd_win_ms, hop_ms=scd_hop_ms, thr=scd_thr, encode=encode)
    embs = embed_segments(y, sr, speech2, encode=encode)
    if clusterer == "ahc_affinity":
        raw = cluster.ahc_cosine(cosine_affinity(embs, _on_gpu(encode)), cluster_cos)
    else:
        factory = (None if clusterer == "hdbscan_two_stage" else
                   cluster.AhcClusterer.factory(cluster_cos) if clusterer == "ahc" else clusterer)
        if factory is not None and not callable(factory):
            raise ValueError(f"clusterer must be 'hdbscan_two_stage', 'ahc', 'ahc_affinity' or a factory, got {clusterer!r}")
        raw = cluster_hdbscan_two_stage(embs, min_cluster_size=2, clusterer_factory=factory)
    labels = cluster.relabel_by_first_appearance(raw)
    for s, lab in zip(speech2, labels):
        s.spk = int(lab)
    merge_input = labels if compat_reference_bugs else embs
    speech3 = conservative_merge(speech2, merge_input, max_gap_s=merge_max_gap_s, max_turn_s=merge_max_speech_s,
                                 min_cos=merge_mincos)
    embs3 = embed_segments(y, sr, speech3, encode=encode)
    speech4 = frame_reassign(y, sr, speech, speech3, embs3, smooth_step=0.10, win=1.0, encode=encode) if reseg else speech3
    return merge_adjacent(speech4, gap=merge_max_gap_s)


def main(wav_path: str, sr: int = 16000, target_lufs: float = -18.0, vad_thr: float = 0.55, min_speech: float = 0.15,
         min_silence: float = 0.10, speech_pad: float = 0.04, morph_bridge_ms: float = 80.0, scd_win: float = 0.8,
         scd_step: float = 0.2, scd_thr: float = 1.2, cluster_cos: float = 0.65, merge_gap: float = 0.25,
         merge_maxturn: float = 10.0, merge_mincos: float = 0.7, reseg: int = 1):
    """The CLI entry of [REF anti_stick_diarize.py:563-604], parameter for parameter.  The reference forwards its 16
    arguments POSITIONALLY into `diarize`, whose parameter order does not match (`min_speech` seconds land in
    `vad_off_thr`, `cluster_cos` in `scd_hop_ms`, ...: SURVEY.md Appendix B-2); the forwarding is kept as it is, so
    the same command line gives the same call."""
    final = diarize(wav_path, sr, target_lufs, vad_thr, min_speech, min_silence, speech_pad, morph_bridge_ms, scd_win,
                    scd_step, scd_thr, cluster_cos, merge_gap, merge_maxturn, merge_mincos, reseg)
    print(f"Segments:{len(final)}; Speakers:{len(set(s.spk for s in final))}")
    for i, s in enumerate(final[:10], 1):
        print(f"{i:02d}  {s.start:.2f}-{s.end:.2f}  SPK_{s.spk}")
    return final


def cluster_hdbscan(embs: np.ndarray, min_cluster_size: int = 2, clusterer_factory=None, use_gpu: bool = False) -> np.ndarray:
    """[REF anti_stick_diarize.py:175-186]; the N x N cosine on the GPU when `use_gpu`."""
    from . import cluster
    aff = (lambda x: cosine_affinity(x, True)) if use_gpu else None
    return cluster.cluster_hdbscan(embs, min_cluster_size, clusterer_factory, affinity=aff)


def cluster_hdbscan_two_stage(embs: np.ndarray, min_cluster_size: int = 2, clusterer_factory=None) -> np.ndarray:
    """[REF anti_stick_diarize.py:189-270] (restated in `cluster.cluster_hdbscan_two_stage`)."""
    from . import cluster
    return cluster.cluster_hdbscan_two_stage(embs, min_cluster_size, clusterer_factory)


def cosine_affinity(embs: np.ndarray, use_gpu: bool) -> np.ndarray:
    """N x N `cosine_similarity(embs)` [REF anti_stick_diarize.py:177]; HIP kernel unless a CPU encoder was injected."""
    if use_gpu:
        import torch
        from . import ops
        return ops.cosine_affinity(torch.from_numpy(np.ascontiguousarray(embs, dtype=np.float32)).cuda()).cpu().numpy()
    from sklearn.metrics.pairwise import cosine_similarity
    return cosine_similarity(embs)
