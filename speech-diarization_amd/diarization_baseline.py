"""The named entry point: audio file(s) -> (start, end, speaker) segments, RTTM, speaker stems.

Same names, signatures, defaults and output schema as the reference's
`diarization_baseline.py` (`DiarizationParameters` [REF :27-39], `extract_speaker_stems`
[REF :42-160], `merge_same_speaker` [REF :188-213], `adjust_segment_boundaries` [REF :216-233],
`diarize_audio` [REF :236-266], `expand_audios` [REF :273-280], `Diarizer` [REF :283-346],
`main` [REF :349-376]).

`diarize_audio` in the reference drives pyannote's gated `speaker-diarization-3.1` pipeline
(segmentation model + embedding + agglomerative clustering), which cannot be fetched and is
out of scope.  Here the same contract is met by the hot path this repository is about:
VAD -> fixed 2 s windows inside speech -> HIP fbank + ECAPA-TDNN embeddings (one large batch)
-> GPU cosine affinity -> spectral (or AHC) clustering on the host -> per-frame labels ->
turns.  `encoder=` lets BASELINE.json's configs[0] ("PyTorch-CPU ECAPA, plumbing, no GPU")
run the identical plumbing with a CPU encoder; it is never selected implicitly.
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Callable

import numpy as np

from . import audio_io, cluster, rttm, vad
from .anti_stick_diarize import cosine_affinity

Encoder = Callable[[np.ndarray], np.ndarray]


@dataclass(frozen=True)
class DiarizationParameters:
    min_stem_s: float = 3.
    max_segment_s: float = 20.0          # longest exported stem
    same_speaker_gap_s: float = 1.2      # neighbouring turns of one speaker merge up to this gap
    max_gap_s: float = 1.5               # silence kept between two pieces of one stem
    fade_ms: float = 20.0                # linear fade at both ends of every piece
    min_speech_duration_s: float = 0.35  # shorter speech is dropped
    min_silence_duration_s: float = 0.1  # shorter silence is bridged
    clustering_threshold: float = 0.7    # cosine similarity threshold for AHC
    min_speakers: int = 2
    max_speakers: int = 6


# ----------------------------------------------------------------------------- segment glue

def merge_same_speaker(segments: list[tuple[float, float, int | str]], max_gap_s: float, max_segment_s: float):
    """Extend a turn with the next one when it is the same speaker, the gap is at most `max_gap_s`
    and the turn built so far is still shorter than `max_segment_s`."""
    merged: list[tuple[float, float, int | str]] = []
    for start, end, spk in segments:
        if merged:
            c_start, c_end, c_spk = merged[-1]
            if c_end - c_start < max_segment_s and spk == c_spk and start - c_end <= max_gap_s:
                merged[-1] = (c_start, max(c_end, end), c_spk)
                continue
        merged.append((start, end, spk))
    return merged


def adjust_segment_boundaries(segments, padding: float) -> list[tuple[float, float, int | str]]:
    """Where two consecutive segments are at least `padding` apart, grow the first's end and pull
    the second's start (not below 0) by `padding` — room for the stems' fades."""
    if len(segments) < 2:
        return segments
    out = list(segments)
    for i in range(len(out) - 1):
        s0, e0, k0 = out[i]
        s1, e1, k1 = out[i + 1]
        if s1 - e0 >= padding:
            out[i] = (s0, e0 + padding, k0)
            out[i + 1] = (max(0, s1 - padding), e1, k1)
    return out


# ----------------------------------------------------------------------------- embedding windows

def speech_windows(speech: list[tuple[float, float]], n_samples: int, sr: int, win_s: float, hop_s: float,
                   margin_s: float = 0.15):
    """Fixed-length windows covering the speech: every `hop_s` inside a region; a region shorter
    than the window gets one window centred on it (shifted to stay inside the signal).
    Returns (start_samples [W], centre_seconds [W], region_index [W])."""
    win = int(round(win_s * sr))
    hop = int(round(hop_s * sr))
    margin = int(round(margin_s * sr))
    starts, regions = [], []
    for r, (s, e) in enumerate(speech):
        a, b = int(round(s * sr)), min(int(round(e * sr)), n_samples)
        if b - a >= win + 2 * margin:      # keep onsets / decays (and the VAD padding) out of the windows
            a, b = a + margin, b - margin
        if b - a >= win:
            st = list(range(a, b - win + 1, hop))
            if st[-1] + win < b:           # cover the tail
                st.append(b - win)
        else:
            mid = (a + b) // 2
            st = [min(max(0, mid - win // 2), max(0, n_samples - win))]
        starts.extend(st)
        regions.extend([r] * len(st))
    starts = np.asarray(starts, dtype=np.int64)
    return starts, (starts + win / 2.0) / sr, np.asarray(regions, dtype=np.int64)


def gather_windows(y: np.ndarray, starts: np.ndarray, win: int) -> np.ndarray:
    """[W, win] float32, zero padded where the signal is shorter than a window."""
    out = np.zeros((len(starts), win), dtype=np.float32)
    for i, s in enumerate(starts):
        piece = y[s: s + win]
        out[i, : len(piece)] = piece
    return out


def labels_to_turns(speech, centres: np.ndarray, regions: np.ndarray, labels: np.ndarray, frame_s: float = 0.01):
    """Inside each speech region every 10 ms frame takes the label of the nearest window centre;
    runs of equal labels become (start, end, label) turns."""
    turns = []
    for r, (s, e) in enumerate(speech):
        sel = np.flatnonzero(regions == r)
        if sel.size == 0:
            continue
        n = max(1, int(round((e - s) / frame_s)))
        t = s + (np.arange(n) + 0.5) * frame_s
        nearest = sel[np.argmin(np.abs(t[:, None] - centres[sel][None, :]), axis=1)]
        lab = labels[nearest]
        cut = np.flatnonzero(np.concatenate(([True], lab[1:] != lab[:-1])))
        ends = np.concatenate((cut[1:], [n]))
        for a, b in zip(cut, ends):
            turns.append((round(s + a * frame_s, 3), round(min(e, s + b * frame_s), 3), int(lab[a])))
    return turns


# ----------------------------------------------------------------------------- the pipeline

def _load(audio) -> tuple[np.ndarray, int, str]:
    if isinstance(audio, dict):
        w = np.asarray(audio["waveform"].cpu().numpy() if hasattr(audio["waveform"], "cpu") else audio["waveform"], dtype=np.float32)
        w = w.mean(axis=0) if w.ndim == 2 else w
        sr = int(audio["sample_rate"])
        if sr != 16000:
            from scipy.signal import resample_poly
            from math import gcd
            g = gcd(sr, 16000)
            w = resample_poly(w, 16000 // g, sr // g).astype(np.float32)
        return np.ascontiguousarray(w), 16000, str(audio.get("uri", "audio"))
    y, sr = audio_io.read_audio(audio, sr=16000, mono=True)
    return y, sr, Path(audio).stem


def diarize_audio(audio_filepath: str | Path | dict, min_speech_duration_s: float, min_silence_duration_s: float,
                  min_speakers: int, max_speakers: int, rttm_filepath: str | Path | None = None, *,
                  encoder: Encoder | None = None, clustering: str = "spectral", clustering_threshold: float = 0.70,
                  window_s: float = 2.0, hop_s: float = 0.25, vad_scorer=None, center_embeddings: bool = True,
                  return_details: bool = False, world: str | None = None) -> list[tuple[float, float, str | int]]:
    """-> [(start_s, end_s, "SPEAKER_xx")], RTTM written to `rttm_filepath` when given.

    `world="dist"` (under torchrun, after `dist.init_from_env()`): BASELINE.json configs[2].  Every rank runs the
    host glue on the same audio, embeds only the windows `i mod W == rank` (`dist.shard_indices`), ONE
    `all_gather_into_tensor` returns the full [N, 192] in window order to every rank, every rank clusters
    (deterministic, N x 192 is tiny), rank 0 writes the RTTM.  The result is identical to `world=None`."""
    y, sr, uri = _load(audio_filepath)
    use_gpu = encoder is None
    if encoder is None:
        from .speech_encode import ecapa_encode_batch
        encoder = ecapa_encode_batch

    scorer = vad.SileroVAD(model=vad_scorer or vad.EnergyScorer())
    probs = scorer.probs(y, sr)
    mask = vad.morph_open_close(vad.hysteresis_binarize(probs, 0.6, 0.4), 10.0)
    speech = vad.mask_to_segments(mask, 10.0, min_speech_ms=min_speech_duration_s * 1000.0,
                                  min_gap_ms=min_silence_duration_s * 1000.0, speech_pad_ms=40.0)
    segments: list[tuple[float, float, str | int]] = []
    details = {"speech": speech, "labels": np.zeros(0, dtype=int), "embeddings": np.zeros((0, 192), np.float32)}
    if speech:
        win = int(round(window_s * sr))
        starts, centres, regions = speech_windows(speech, len(y), sr, window_s, hop_s)
        if world is None:
            embs = _embed_windows(encoder, y, starts, win, use_gpu)
        elif world == "dist":
            embs = _embed_sharded(encoder, y, starts, win, use_gpu)
        else:
            raise ValueError(f"world must be None or 'dist', got {world!r}")
        # recording-level mean removal (first step of the reference's whiten_l2, [REF diar_diag.py:187-188]):
        # untrained / mismatched encoders put a large common component into every embedding
        K = cosine_affinity(cluster.center(embs) if center_embeddings else embs, use_gpu)
        if clustering == "spectral":
            k = cluster.estimate_num_speakers(K, min_speakers, max_speakers)
            labels = cluster.spectral(K, k)
        elif clustering == "ahc":
            labels = cluster.ahc_cosine(K, clustering_threshold)
        else:
            raise ValueError(f"unknown clustering {clustering!r}")
        labels = cluster.relabel_by_first_appearance(labels)
        segments = [(s, e, rttm.speaker_label(k)) for s, e, k in labels_to_turns(speech, centres, regions, labels)]
        details.update(labels=labels, embeddings=embs, affinity=K, window_starts=starts)
    if rttm_filepath and (world is None or _rank() == 0):
        with open(rttm_filepath, "w") as f:
            rttm.write_rttm(segments, uri, f)
    return (segments, details) if return_details else segments


def _rank() -> int:
    from . import dist
    return dist.world()[0]


def _embed_windows(encoder: Encoder, y: np.ndarray, starts: np.ndarray, win: int, use_gpu: bool, to_host: bool = True):
    """[len(starts), 192] embeddings of the windows y[s : s + win] (zero padded past the end).  HIP encoder: the signal is
    uploaded ONCE and the fbank kernel reads the windows in place (`sd_fbank_windows_f32`; a 1 h meeting at 2 s / 0.25 s is
    230 MB of signal instead of 1.8 GB of gathered, 8x overlapping windows).  Injected encoder (configs[0], tests): the
    literal host gather the reference's callers do."""
    if use_gpu:
        from .speech_encode import using_ecapa_encoder
        return using_ecapa_encoder().encode_windows(y, starts, win, to_host=to_host)
    return encoder(gather_windows(y, starts, win)) if len(starts) else np.zeros((0, 192), np.float32)


def _embed_sharded(encoder: Encoder, y: np.ndarray, starts: np.ndarray, win: int, use_gpu: bool) -> np.ndarray:
    """Embed this rank's round-robin shard of the windows, all-gather the 192-d rows (RCCL when the process
    group is "nccl": device tensors; gloo: host tensors), return all N rows in window order.  With the HIP encoder
    the embeddings stay on the device from the forward through the collective: one D2H of [N, 192] at the end."""
    import torch
    from . import dist
    rank, w = dist.world()
    n = len(starts)
    mine = dist.shard_indices(n, rank, w)
    on_device = use_gpu and (not torch.distributed.is_initialized() or torch.distributed.get_backend() == "nccl")
    local = _embed_windows(encoder, y, starts[mine], win, use_gpu, to_host=not on_device)
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
    return dist.all_gather_embeddings(t, n).cpu().numpy()


def expand_audios(root: Path):
    if root.is_file():
        root = root.resolve()
        return [root], root.parent
    exts = {".wav", ".flac", ".mp3", ".m4a", ".ogg", ".opus", ".aac"}
    return [p for p in root.rglob("*.*") if p.is_file() and p.suffix.lower() in exts], root


# ----------------------------------------------------------------------------- stems

def plan_speaker_tracks(segments, max_segment_s: float, max_gap_s: float):
    """Per speaker: tracks = lists of ("silence", seconds) / ("speech", start_s, end_s) items, a new
    track starting whenever adding the next turn (plus its leading silence, capped at max_gap_s)
    would exceed max_segment_s [REF diarization_baseline.py:106-153]."""
    per_spk = defaultdict(list)
    for start, end, spk in segments:
        per_spk[spk].append((start, end))
    plans = {}
    for spk, segs in per_spk.items():
        segs.sort()
        tracks, cur, cur_dur, last_end = [], [], 0.0, 0.0
        for i, (s, e) in enumerate(segs):
            sil = min(s - last_end, max_gap_s) if i > 0 else 0.0
            if cur_dur + sil + (e - s) > max_segment_s:
                tracks.append(cur)
                cur, cur_dur, sil = [], 0.0, 0.0
            if sil > 0:
                cur.append(("silence", sil))
                cur_dur += sil
            cur.append(("speech", s, e))
            cur_dur += e - s
            last_end = e
        tracks.append(cur)
        plans[spk] = tracks
    return plans


def extract_speaker_stems(audio: str | Path | dict, segments: list[tuple[float, float, str | int]], root: str | Path,
                          max_segment_s: float, max_gap_s: float, fade_ms: float, min_stem_s: float) -> dict[str | int, list[str]]:
    """One or more stems per speaker under `<root>/<speaker>/<stem>-NNN.flac` (16-bit FLAC, as the reference writes them
    [REF diarization_baseline.py:95-103]; `flac.py`), pieces faded in/out linearly, silences between pieces capped at `max_gap_s`,
    stems shorter than `min_stem_s` skipped."""
    if isinstance(audio, (str, Path)):
        y, sr = audio_io.read_audio(audio, sr=16000, mono=False)
        stem_name = Path(audio).stem
    else:
        y = np.asarray(audio["waveform"], dtype=np.float32)
        y = y[None, :] if y.ndim == 1 else y
        sr = int(audio["sample_rate"])
        stem_name = Path(root).stem
    root = Path(root)
    fade = int(round(fade_ms / 1000.0 * sr))
    ramp = np.linspace(0.0, 1.0, fade, dtype=np.float32) if fade > 0 else None
    out: dict = defaultdict(list)
    for spk, tracks in plan_speaker_tracks(segments, max_segment_s, max_gap_s).items():
        for items in tracks:
            chunks = []
            for it in items:
                if it[0] == "silence":
                    chunks.append(np.zeros((y.shape[0], int(it[1] * sr)), dtype=np.float32))
                else:
                    piece = y[:, int(it[1] * sr): int(it[2] * sr)].copy()
                    if fade > 0 and piece.shape[1] >= 2 * fade:
                        piece[:, :fade] *= ramp
                        piece[:, -fade:] *= ramp[::-1]
                    chunks.append(piece)
            if not chunks:
                continue
            wave = np.concatenate(chunks, axis=1)
            if wave.shape[1] / sr < min_stem_s:
                continue
            path = root / f"{spk}/{stem_name}-{len(out[spk]):03d}.flac"
            path.parent.mkdir(parents=True, exist_ok=True)
            audio_io.write_flac16(path, wave, sr)
            out[spk].append(str(path.absolute()))
    return dict(out)


class Diarizer:
    def __init__(self, hparams: DiarizationParameters, encoder: Encoder | None = None, clustering: str = "spectral"):
        self.hparams = hparams
        self.encoder = encoder
        self.clustering = clustering

    def diarize(self, apath: str | Path | dict, rttm_filepath: str | Path | None):
        hp = self.hparams
        segments = diarize_audio(apath, hp.min_speech_duration_s, min_silence_duration_s=hp.min_silence_duration_s,
                                 min_speakers=hp.min_speakers, max_speakers=hp.max_speakers, rttm_filepath=rttm_filepath,
                                 encoder=self.encoder, clustering=self.clustering,
                                 clustering_threshold=hp.clustering_threshold)
        segments = [s for s in segments if s[1] - s[0] >= hp.min_speech_duration_s]
        return sorted(segments)

    def merge_segments(self, segments):
        return merge_same_speaker(segments, self.hparams.same_speaker_gap_s, self.hparams.max_segment_s)

    def pad_segment(self, segments):
        return adjust_segment_boundaries(segments, padding=self.hparams.fade_ms * 2 / 1000)

    def extract_speaker(self, segments, audio_path: str | Path | dict, root: str | Path) -> dict:
        hp = self.hparams
        return extract_speaker_stems(audio_path, segments, root, hp.max_segment_s, hp.max_gap_s, hp.fade_ms, hp.min_stem_s)

    def __call__(self, audio_path: str | Path, root: str | Path, with_rttm: bool = False):
        rttm_filepath = Path(audio_path).with_suffix(".rttm") if with_rttm else None
        segments = self.diarize(audio_path, rttm_filepath)
        segments = self.merge_segments(segments)
        segments = self.pad_segment(segments)
        info = self.extract_speaker(segments, audio_path, root)
        return segments, info


def main(root: str, min_speakers: int = 2, max_speakers: int = 6, max_segment_s: float = 20.0,
         min_silence_duration_s: float = 0.1, min_speech_duration_s: float = 0.35, same_speaker_gap_s: float = 1.,
         max_gap_s: float = 1.5, fade_ms: float = 30.0):
    args = dict(locals())
    args.pop("root")
    unknown = set(args) - {f.name for f in fields(DiarizationParameters)}
    if unknown:                                 # dacite strict=True in the reference
        raise TypeError(f"unknown parameters {sorted(unknown)}")
    hparams = DiarizationParameters(**args)
    diarizer = Diarizer(hparams)
    audios, aroot = expand_audios(Path(root))
    print(aroot, len(audios))
    for apath in audios:
        if apath.with_suffix(".rttm").exists():   # the reference's only resume mechanism
            continue
        troot = apath.with_name(f"{apath.stem}-speakers")
        segments, info = diarizer(apath, troot, True)
        print(apath, len(segments))


if __name__ == "__main__":
    import argparse
    import inspect
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for name, prm in inspect.signature(main).parameters.items():
        if prm.default is inspect.Parameter.empty:
            ap.add_argument(name)
        else:
            ap.add_argument(f"--{name}", type=type(prm.default), default=prm.default)
    main(**vars(ap.parse_args()))
