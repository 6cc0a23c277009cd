#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own glue functions.

    python -B tests/golden/make_golden.py          # only works where /root/reference exists

The reference's flat modules import third-party packages that are not installed
(numba, librosa, onnxruntime, pyloudnorm, hdbscan, torchaudio, speechbrain, dacite,
pyannote).  They are replaced by INERT stubs (no behaviour: anything called on them raises),
which is enough to import the modules and execute the pure-numpy / pure-python functions the
reference itself authored.  Two call-throughs are substituted with test doubles, and the
fixtures say so: `ecapa_encode_batch` (a deterministic, row-independent stand-in encoder
defined below — the real one needs downloaded weights) and `frame_audio` (librosa.util.frame
is absent; the framing n = 1 + (len - win) // hop is restated in our `vad.frame_audio`).
What is pinned is therefore the reference's batching / padding / merging / thresholding logic.

Only inputs and outputs (small JSON) are written to tests/golden/; no reference source is copied.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        def _raise(*a, **k):
            raise RuntimeError(f"stub {self.__name__}.{name} called: this dependency is absent")
        return _raise


def install_stubs():
    import torch
    names = ["numba", "librosa", "librosa.util", "librosa.effects", "onnxruntime", "pyloudnorm", "hdbscan", "torchaudio",
             "torchaudio.transforms", "speechbrain", "speechbrain.inference", "speechbrain.inference.classifiers",
             "dacite", "pyannote", "pyannote.audio", "pyannote.core", "pyannote.audio.core", "pyannote.audio.core.model",
             "pyannote.audio.pipelines", "pyannote.audio.pipelines.utils", "pyannote.audio.pipelines.utils.hook",
             "jsonargparse", "soundfile"]
    for n in names:
        sys.modules[n] = _Inert(n)
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    sys.modules["pyannote.audio.core.model"].Model = type("Model", (torch.nn.Module,), {})
    for mod, attrs in {"torchaudio.transforms": ["MelSpectrogram"], "speechbrain.inference.classifiers": ["EncoderClassifier"],
                       "hdbscan": ["HDBSCAN"], "dacite": ["from_dict", "Config"], "pyannote.audio": ["Pipeline"],
                       "pyannote.core": ["Annotation"], "pyannote.audio.pipelines.utils.hook": ["ProgressHook"]}.items():
        for a in attrs:
            setattr(sys.modules[mod], a, type(a, (), {}))


def fake_encode(wavs: np.ndarray) -> np.ndarray:
    """Deterministic, row-independent stand-in for ecapa_encode_batch: 192 band energies of the
    row (zero padding lowers them, as it would change a real embedding)."""
    wavs = np.asarray(wavs, dtype=np.float64)
    n = wavs.shape[1]
    edges = np.linspace(0, n, 193).astype(int)
    out = np.stack([np.abs(wavs[:, edges[d]:max(edges[d + 1], edges[d] + 1)]).mean(axis=1) * (1.0 + 0.01 * d)
                    for d in range(192)], axis=1)
    return (out + 0.05 * np.sin(np.arange(192))[None, :]).astype(np.float32)


def test_signal(seed: int, seconds: float, sr: int = 16000) -> np.ndarray:
    """Piecewise-stationary signal: the carrier and envelope change every few seconds."""
    from speech_diarization_amd import synth
    n = int(seconds * sr)
    t = np.arange(n) / sr
    u = synth.uniform(seed, "golden.sig", (64,))
    y = np.zeros(n)
    pos, k = 0.0, 0
    while pos < seconds:
        dur = 1.5 + 3.0 * float(u[k % 64]); f = 100.0 + 700.0 * float(u[(k + 7) % 64]); amp = 0.1 + 0.5 * float(u[(k + 13) % 64])
        a, b = int(pos * sr), min(n, int((pos + dur) * sr))
        y[a:b] = amp * np.sin(2 * np.pi * f * t[a:b]) * (1.0 + 0.5 * np.sin(2 * np.pi * (2 + k % 3) * t[a:b]))
        pos += dur; k += 1
    return y.astype(np.float32)


def jsonable(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    return x


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import vad as rvad
    import anti_stick_diarize as rasd
    import diarization_baseline as rdb
    from speech_diarization_amd import synth
    from speech_diarization_amd import vad as myvad

    out = {}

    # ---- VAD post-processing on synthetic probability tracks (masks stored as '0'/'1' strings)
    def bits(m):
        return "".join("1" if v else "0" for v in np.asarray(m).astype(bool))

    tracks = []
    inputs = []
    for seed, n in [(1, 94), (2, 400), (3, 1000), (4, 37), (5, 2000), (6, 1)]:
        u = synth.uniform(seed, "golden.vad", (n,))
        t = np.arange(n)
        probs = np.clip(0.5 + 0.45 * np.sin(t / (5.0 + seed)) * np.sign(np.sin(t / (31.0 + 3 * seed))) + 0.25 * (u - 0.5), 0, 1)
        inputs.append(np.round(probs, 4).astype(np.float32))
    inputs.append(np.zeros(50, np.float32))
    inputs.append(np.ones(50, np.float32))
    for probs in inputs:
        variants = []
        for on, off in [(0.6, 0.4), (0.5, 0.5), (0.7, 0.2)]:
            mask = rvad.hysteresis_binarize(probs, on, off)
            morphs = []
            for open_ms, close_ms in [(80.0, 40.0), (0.0, 40.0), (30.0, 0.0)]:
                m2 = rvad.morph_open_close(mask, 10.0, open_ms, close_ms)
                segs = []
                for min_speech, min_gap, pad in [(250.0, 100.0, 80.0), (250.0, 100.0, 40.0), (150.0, 250.0, 0.0), (25.0, 5.0, 15.0)]:
                    segs.append(dict(min_speech_ms=min_speech, min_gap_ms=min_gap, speech_pad_ms=pad,
                                     segments=rvad.mask_to_segments(m2, 10.0, min_speech, min_gap, pad)))
                morphs.append(dict(open_ms=open_ms, close_ms=close_ms, mask=bits(m2), segments=segs))
            variants.append(dict(on=on, off=off, hyst=bits(mask), morphs=morphs))
        tracks.append(dict(probs=[float(v) for v in probs], variants=variants))
    out["vad"] = tracks

    # ---- the same functions off the default hop: the reference rounds an end that its clamp replaced by len(mask)
    # (a Python int) with Python's round and everything else with numpy's [REF vad.py:157-160]; the two differ on ties,
    # which hop 10 ms never produces.  Masks are random runs; `tail` forces the last run to reach (or stop short of) the end.
    hops = []
    rng = np.random.default_rng(20250)
    for hop_ms in [2.5, 7.5, 10.0, 12.5, 16.0, 20.0]:
        cases = []
        for k in range(60):
            n = int(rng.integers(20, 900))
            m = np.zeros(n, bool)
            pos = int(rng.integers(0, 30))
            while pos < n:
                run = int(rng.integers(1, 120))
                m[pos:pos + run] = True
                pos += run + int(rng.integers(1, 60))
            tail = k % 3
            if tail == 0:
                m[n - int(rng.integers(1, 40)):] = True        # the last run touches the end: any pad is clamped
            elif tail == 1:
                m[n - int(rng.integers(1, 12)):] = False       # ends a few frames short: clamped only by the larger pads
            min_speech, min_gap = [(250.0, 100.0), (25.0, 5.0), (150.0, 250.0)][k % 3]
            m2 = rvad.morph_open_close(m, hop_ms, 80.0, 40.0) if k % 2 else m
            segs = [dict(speech_pad_ms=pad, segments=rvad.mask_to_segments(m2, hop_ms, min_speech, min_gap, pad))
                    for pad in [0.0, 12.5, 40.0, 80.0]]
            cases.append(dict(mask=bits(m), morphed=bool(k % 2), mask2=bits(m2) if k % 2 else None, min_speech_ms=min_speech, min_gap_ms=min_gap, pads=segs))
        hops.append(dict(hop_ms=hop_ms, cases=cases))
    out["vad_hops"] = hops

    # ---- diarization_baseline glue
    db = []
    for seed in range(12):
        u = synth.uniform(seed, "golden.db", (40,))
        t, segs = 0.0, []
        for i in range(13):
            t += float(u[3 * i]) * 2.0
            d = 0.2 + float(u[3 * i + 1]) * 9.0
            spk = ["A", "B", "C"][int(u[3 * i + 2] * 3) % 3] if seed % 2 else int(u[3 * i + 2] * 3) % 3
            segs.append((round(t, 3), round(t + d, 3), spk))
            t += d - (0.3 if i % 5 == 4 else 0.0)   # occasional overlap
        for gap, mx in [(1.2, 20.0), (1.0, 20.0), (0.3, 5.0)]:
            merged = rdb.merge_same_speaker(list(segs), gap, mx)
            for padv in [0.04, 0.06, 0.5]:
                db.append(dict(segments=segs, max_gap_s=gap, max_segment_s=mx, merged=merged, padding=padv,
                               adjusted=rdb.adjust_segment_boundaries(list(merged), padv)))
    db.append(dict(segments=[], max_gap_s=1.2, max_segment_s=20.0, merged=rdb.merge_same_speaker([], 1.2, 20.0), padding=0.04,
                   adjusted=rdb.adjust_segment_boundaries([], 0.04)))
    out["diarization_baseline"] = db
    p = rdb.DiarizationParameters()
    out["diarization_parameters"] = {k: getattr(p, k) for k in p.__dataclass_fields__}

    # ---- anti_stick_diarize glue with the stand-in encoder
    rasd.ecapa_encode_batch = fake_encode
    rasd.frame_audio = myvad.frame_audio
    rasd.track = lambda it, **k: it          # silence rich progress bars
    asd = []
    for seed, seconds in [(11, 24.0), (12, 40.0)]:
        y = test_signal(seed, seconds)
        sr = 16000
        u = synth.uniform(seed, "golden.asd", (64,))
        t, segs = 0.2, []
        k = 0
        while t < seconds - 1.0:
            d = [0.3, 0.45, 1.2, 2.5, 5.0, 7.5][int(u[k % 64] * 6) % 6]
            e = min(seconds - 0.05, t + d)
            segs.append((round(t, 3), round(e, 3)))
            t = e + 0.05 + float(u[(k + 1) % 64]) * 0.8
            k += 2
        calls = []

        def logging_encode(w, _calls=calls):
            _calls.append([int(w.shape[0]), int(w.shape[1]), float(np.abs(w).sum())])
            return fake_encode(w)

        rasd.ecapa_encode_batch = logging_encode
        embs = rasd.embed_segments(y, sr, [rasd.Segment(s, e) for s, e in segs])
        embed_calls = list(calls)
        calls.clear()
        long_segs = [rasd.Segment(s, e) for s, e in segs if e - s >= 1.0]
        scd = {}
        for thr in (0.5, 1.25):
            res = rasd.scd_split_segments(y, sr, [rasd.Segment(s.start, s.end) for s in long_segs], thr=thr)
            scd[str(thr)] = [(s.start, s.end) for s in res]
        rasd.ecapa_encode_batch = fake_encode
        labels = [int(u[(3 * i) % 64] * 3) % 3 for i in range(len(segs))]
        labelled = [rasd.Segment(s, e, lab) for (s, e), lab in zip(segs, labels)]
        merged = rasd.conservative_merge([rasd.Segment(s.start, s.end, s.spk) for s in labelled], embs, 0.5, 30.0, 0.80)
        merged_loose = rasd.conservative_merge([rasd.Segment(s.start, s.end, s.spk) for s in labelled], embs, 1.0, 10.0, 0.0)
        merged_bug = rasd.conservative_merge([rasd.Segment(s.start, s.end, s.spk) for s in labelled], np.asarray(labels), 0.5, 30.0, 0.80)
        _, cents = rasd.speaker_centroids(labelled, embs)
        starts, valid = rasd._get_speech_windows(y, sr, [rasd.Segment(s, e) for s, e in segs], 16000, 1600)
        wl = np.asarray([int(u[i % 64] * 3) % 3 for i in range(len(valid))])
        l2s = rasd._labels_to_segments(starts, valid, wl, sr, len(y) / sr)
        adj = rasd.merge_adjacent([rasd.Segment(s.start, s.end, s.spk) for s in l2s], gap=0.05)
        asd.append(dict(seed=seed, seconds=seconds, segments=segs, labels=labels, embed_calls=embed_calls,
                        embs_sum=float(np.abs(embs).sum()), embs_first=embs[0][:8], embs_last=embs[-1][:8], scd=scd,
                        merged=[(s.start, s.end, s.spk) for s in merged], merged_loose=[(s.start, s.end, s.spk) for s in merged_loose],
                        merged_labels_as_embs=[(s.start, s.end, s.spk) for s in merged_bug], centroids_head=cents[:, :6],
                        n_windows=len(starts), valid=valid, window_labels=wl, labels_to_segments=[(s.start, s.end, s.spk) for s in l2s],
                        merge_adjacent=[(s.start, s.end, s.spk) for s in adj]))
    out["anti_stick_diarize"] = asd

    # ---- diar_diag score helpers (pure numpy in the reference)
    import tempfile
    import diar_diag as rdd
    dd = []
    for seed in (21, 22):
        e = synth.normal(seed, "golden.dd.e", (40, 24)).astype(np.float64)
        e[:20] += 2.0 * synth.normal(seed, "golden.dd.c0", (1, 24))
        e[20:] += 2.0 * synth.normal(seed, "golden.dd.c1", (1, 24))
        cents = np.stack([e[:20].mean(0), e[20:].mean(0)])
        cohort = synth.normal(seed, "golden.dd.coh", (64, 24)).astype(np.float64)
        scores = synth.normal(seed, "golden.dd.s", (60, 3)).astype(np.float32)
        scores[:25, 0] += 1.5; scores[25:45, 2] += 1.5; scores[45:, 1] += 1.5
        segs = [{"start": 0.0, "end": 3661.2345, "speaker": "S0"}, {"start": 59.9996, "end": 61.5, "speaker": "S1"}]
        with tempfile.TemporaryDirectory() as td:
            rdd.save_srt(os.path.join(td, "a.srt"), segs)
            rdd.save_csv(os.path.join(td, "a.csv"), segs)
            rdd.save_json(os.path.join(td, "a.json"), segs, ["S0", "S1"])
            texts = {k: open(os.path.join(td, f"a.{k}"), encoding="utf-8").read() for k in ("srt", "csv", "json")}
        dd.append(dict(seed=seed, embs=e, centers=cents, cohort=cohort, whiten=rdd.whiten_l2(e), asnorm=rdd.asnorm_scores(e, cents, cohort, topk=20),
                       scores=scores, viterbi=rdd.viterbi_hmm(scores, alpha=0.9), viterbi_sticky=rdd.viterbi_hmm(scores), segments=segs, texts=texts))
    out["diar_diag"] = dd

    # ---- cluster_hdbscan / cluster_hdbscan_two_stage: the reference's glue around a pluggable clusterer.
    # `hdbscan` is absent, so the module-level HDBSCAN name is replaced by stand-ins: (a) a scripted class
    # that returns prepared label vectors and records what it was constructed with and handed (this pins the
    # normalisation, the centroid arithmetic, the degenerate returns and the map-back), (b) scikit-learn's
    # HDBSCAN behind the same constructor (a real density clusterer through the whole function).
    class Scripted:
        script: list = []
        log: list = []

        def __init__(self, **kw):
            self.kw = kw

        def fit_predict(self, X):
            X = np.asarray(X)
            Scripted.log.append(dict(kwargs={k: (v if v is None or isinstance(v, (int, float, str, bool)) else str(v)) for k, v in self.kw.items()},
                                     shape=list(X.shape), X=X.astype(np.float64)))
            return np.asarray(Scripted.script.pop(0))

    def embs_for(seed, sizes, dim=16, spread=0.15):
        rows = []
        for k, n in enumerate(sizes):
            c = synth.normal(seed, f"golden.cl.c{k}", (1, dim)).astype(np.float64)
            rows.append(3.0 * (1 + 0.3 * k) * c + spread * synth.normal(seed, f"golden.cl.n{k}", (n, dim)).astype(np.float64))
        return np.concatenate(rows, axis=0)

    cl = []
    scripted_cases = [
        # name, embs, min_cluster_size, scripted fit_predict outputs (stage 1[, stage 2])
        ("no_micro_clusters", embs_for(31, [5]), 2, [[-1] * 5]),                                    # [REF :216-218]
        ("one_centroid_lt_min_cluster_size", embs_for(32, [4, 3]), 2, [[0, 0, -1, 0, 0, -1, 0]]),   # [REF :243-245]
        ("two_centroids_lt_mcs3", embs_for(33, [3, 3]), 3, [[0, 0, 0, 1, 1, -1]]),
        ("stage2_noise_centroid", embs_for(34, [3, 3, 3]), 2, [[0, 0, 0, 1, 1, 1, 2, 2, -1], [0, -1, 0]]),   # [REF :261-266]
        ("label_gap_and_merge", embs_for(35, [2, 2, 3, 2]), 2, [[0, 0, 3, 3, 1, -1, 1, 3, 3], [1, 1, 0]]),   # label 2 never occurs [REF :229]
        ("all_merge_to_one", embs_for(36, [4, 4]), 2, [[0, 1, 0, 1, 2, 2, 3, 3], [0, 0, 0, 0]]),
        ("zero_row_embedding", np.concatenate([embs_for(37, [3, 2]), np.zeros((1, 16))]), 2, [[0, 0, 0, 1, 1, 1], [0, 1]]),
    ]
    rasd.HDBSCAN = Scripted
    for name, e, mcs, script in scripted_cases:
        Scripted.script, Scripted.log = [list(v) for v in script], []
        labels = rasd.cluster_hdbscan_two_stage(e, min_cluster_size=mcs)
        cl.append(dict(name=name, kind="scripted", embs=e, min_cluster_size=mcs, script=script, calls=list(Scripted.log),
                       labels=np.asarray(labels), labels_dtype=str(np.asarray(labels).dtype)))
    # single-stage variant [REF :175-186]
    e = embs_for(38, [4, 3])
    Scripted.script, Scripted.log = [[0, 0, 0, 0, 1, 1, -1]], []
    labels = rasd.cluster_hdbscan(e, min_cluster_size=2)
    cl.append(dict(name="single_stage", kind="scripted_single", embs=e, min_cluster_size=2, script=[[0, 0, 0, 0, 1, 1, -1]],
                   calls=list(Scripted.log), labels=np.asarray(labels), labels_dtype=str(np.asarray(labels).dtype)))
    from sklearn.cluster import HDBSCAN as SkHDBSCAN
    rasd.HDBSCAN = SkHDBSCAN
    for seed, sizes, mcs in [(41, [12, 9, 7], 2), (42, [20, 15, 10, 5], 2), (43, [6, 6], 3), (44, [30], 2)]:
        e = embs_for(seed, sizes, dim=24, spread=0.6)
        labels = rasd.cluster_hdbscan_two_stage(e, min_cluster_size=mcs)
        cl.append(dict(name=f"sklearn_hdbscan_{seed}", kind="sklearn", embs=e, min_cluster_size=mcs, labels=np.asarray(labels),
                       labels_dtype=str(np.asarray(labels).dtype)))
    out["cluster_two_stage"] = cl

    # ---- diar_diag.cluster_embeddings [REF diar_diag.py:213-229]: the same two stand-ins in the `hdbscan.HDBSCAN` slot
    # (scripted: pins the constructor arguments and the distance matrix; scikit-learn: labels through a real clusterer),
    # plus the "agglo" branch, which runs on the scikit-learn that is installed.
    ce = []
    e = embs_for(51, [8, 7, 6], dim=24, spread=0.5)
    rdd.hdbscan.HDBSCAN = Scripted
    Scripted.script, Scripted.log = [[0] * 8 + [1] * 7 + [-1] * 6], []
    labels = rdd.cluster_embeddings(e, method="hdbscan")
    ce.append(dict(name="scripted", method="hdbscan", embs=e, calls=list(Scripted.log), labels=np.asarray(labels)))
    rdd.hdbscan.HDBSCAN = SkHDBSCAN
    for seed, sizes in [(52, [14, 12, 9]), (53, [30]), (54, [7, 7, 3])]:
        e = embs_for(seed, sizes, dim=24, spread=0.6)
        ce.append(dict(name=f"sklearn_{seed}", method="hdbscan", embs=e, labels=np.asarray(rdd.cluster_embeddings(e, method="hdbscan"))))
    for seed, sizes, thr in [(55, [10, 8, 6], 0.68), (56, [9, 9], 0.3)]:
        e = embs_for(seed, sizes, dim=24, spread=0.8)
        ce.append(dict(name=f"agglo_{seed}", method="agglo", cos_thr=thr, embs=e, labels=np.asarray(rdd.cluster_embeddings(e, method="agglo", cos_thr=thr))))
    out["cluster_embeddings"] = ce

    for name, payload in out.items():
        with open(os.path.join(HERE, f"{name}.json"), "w") as f:
            json.dump(jsonable(payload), f)
        print(name, os.path.getsize(os.path.join(HERE, f"{name}.json")), "bytes")


if __name__ == "__main__":
    main()
