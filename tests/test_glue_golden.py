"""Host-side glue against golden vectors produced by the REFERENCE's own functions
(tests/golden/make_golden.py imports /root/reference with inert stubs and records outputs)."""
import json
import os

import numpy as np
import pytest

from speech_diarization_amd import anti_stick_diarize as asd
from speech_diarization_amd import diarization_baseline as db
from speech_diarization_amd import vad


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, f"{name}.json")) as f:
        return json.load(f)


def _bits(s):
    return np.frombuffer(s.encode(), dtype=np.uint8) == ord("1")


def _pairs(x):
    return [tuple(v) for v in x]


def test_vad_postprocessing_matches_reference(golden_dir):
    tracks = _load(golden_dir, "vad")
    n_cases = 0
    for tr in tracks:
        probs = np.asarray(tr["probs"], dtype=np.float32)
        for var in tr["variants"]:
            mask = vad.hysteresis_binarize(probs, var["on"], var["off"])
            assert mask.dtype == np.bool_
            assert np.array_equal(mask, _bits(var["hyst"])), (var["on"], var["off"])
            for mo in var["morphs"]:
                m2 = vad.morph_open_close(mask, 10.0, mo["open_ms"], mo["close_ms"])
                assert np.array_equal(m2, _bits(mo["mask"]))
                for sg in mo["segments"]:
                    got = vad.mask_to_segments(m2, 10.0, sg["min_speech_ms"], sg["min_gap_ms"], sg["speech_pad_ms"])
                    assert got == _pairs(sg["segments"])
                    assert all(isinstance(a, float) and isinstance(b, float) for a, b in got)
                    n_cases += 1
    assert n_cases >= 200


def test_vad_edge_cases():
    assert vad.mask_to_segments(np.zeros(10, bool), 10.0) == []
    assert vad.mask_to_segments(np.zeros(0, bool), 10.0) == []
    assert vad.hysteresis_binarize(np.zeros(0, np.float32)).shape == (0,)
    # banker's rounding of the frame counts: round(25 / 10) == 2, so a 2-frame run survives min_speech_ms=25
    m = np.zeros(20, bool); m[5:7] = True
    assert vad.mask_to_segments(m, 10.0, min_speech_ms=25.0, min_gap_ms=0.0, speech_pad_ms=0.0) == [(0.05, 0.07)]
    # filter-then-merge: two short runs separated by a tiny gap are both dropped, not merged
    m = np.zeros(100, bool); m[10:20] = True; m[22:32] = True
    assert vad.mask_to_segments(m, 10.0, min_speech_ms=250.0, min_gap_ms=100.0, speech_pad_ms=0.0) == []


def test_frame_audio_shape_and_errors():
    y = np.arange(16000, dtype=np.float32)
    f = vad.frame_audio(y, 16000, 30.0, 10.0)
    assert f.shape == (1 + (16000 - 480) // 160, 480)
    assert np.array_equal(f[3], y[480:960])
    assert vad.frame_audio(y, 16000, 1000.0, 200.0).shape == (1, 16000)
    with pytest.raises(ValueError):
        vad.frame_audio(y[:100], 16000, 30.0, 10.0)


def test_silero_needs_a_scorer_offline():
    with pytest.raises(RuntimeError, match="torch.hub"):
        vad.SileroVAD()
    p = vad.SileroVAD(model=vad.EnergyScorer()).probs(np.zeros(16000, np.float32), batch_size=7)
    assert p.shape == (98,) and p.dtype == np.float32 and p.max() < 0.5


def test_merge_same_speaker_and_boundaries_match_reference(golden_dir):
    for case in _load(golden_dir, "diarization_baseline"):
        segs = _pairs(case["segments"])
        merged = db.merge_same_speaker(list(segs), case["max_gap_s"], case["max_segment_s"])
        assert merged == _pairs(case["merged"])
        assert db.adjust_segment_boundaries(list(merged), case["padding"]) == _pairs(case["adjusted"])


def test_diarization_parameters_defaults_match_reference(golden_dir):
    ref = _load(golden_dir, "diarization_parameters")
    p = db.DiarizationParameters()
    assert {k: getattr(p, k) for k in p.__dataclass_fields__} == ref
    with pytest.raises(Exception):
        p.min_speakers = 3          # frozen, like the reference's dataclass


def _fake_encode(wavs):
    """The stand-in encoder of tests/golden/make_golden.py (kept in sync by this test's fixtures)."""
    wavs = np.asarray(wavs, dtype=np.float64)
    n = wavs.shape[1]
    edges = np.linspace(0, n, 193).astype(int)
    out = np.stack([np.abs(wavs[:, edges[d]:max(edges[d + 1], edges[d] + 1)]).mean(axis=1) * (1.0 + 0.01 * d)
                    for d in range(192)], axis=1)
    return (out + 0.05 * np.sin(np.arange(192))[None, :]).astype(np.float32)


def _signal(seed, seconds):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.test_signal(seed, seconds)


def _triples(segs):
    return [(s.start, s.end, s.spk) for s in segs]


def test_embedding_callers_match_reference(golden_dir):
    for case in _load(golden_dir, "anti_stick_diarize"):
        y = _signal(case["seed"], case["seconds"])
        sr = 16000
        segs = _pairs(case["segments"])
        calls = []

        def enc(w):
            calls.append([int(w.shape[0]), int(w.shape[1]), float(np.abs(w).sum())])
            return _fake_encode(w)

        embs = asd.embed_segments(y, sr, [asd.Segment(s, e) for s, e in segs], encode=enc)
        # identical batch shapes AND contents (batches of 32, zero-padded to the batch max, short ones widened)
        assert [c[:2] for c in calls] == [c[:2] for c in case["embed_calls"]]
        assert np.allclose([c[2] for c in calls], [c[2] for c in case["embed_calls"]], rtol=1e-6)
        assert embs.shape == (len(segs), 192) and embs.dtype == np.float32
        assert np.isclose(np.abs(embs).sum(), case["embs_sum"], rtol=1e-6)
        assert np.allclose(embs[0][:8], case["embs_first"], rtol=1e-6) and np.allclose(embs[-1][:8], case["embs_last"], rtol=1e-6)

        long_segs = [asd.Segment(s, e) for s, e in segs if e - s >= 1.0]
        for thr, expect in case["scd"].items():
            res = asd.scd_split_segments(y, sr, [asd.Segment(s.start, s.end) for s in long_segs], thr=float(thr), encode=_fake_encode)
            assert [(s.start, s.end) for s in res] == _pairs(expect)

        labelled = [asd.Segment(s, e, lab) for (s, e), lab in zip(segs, case["labels"])]
        fresh = lambda: [asd.Segment(s.start, s.end, s.spk) for s in labelled]   # noqa: E731
        assert _triples(asd.conservative_merge(fresh(), embs, 0.5, 30.0, 0.80)) == _pairs(case["merged"])
        assert _triples(asd.conservative_merge(fresh(), embs, 1.0, 10.0, 0.0)) == _pairs(case["merged_loose"])
        assert _triples(asd.conservative_merge(fresh(), np.asarray(case["labels"]), 0.5, 30.0, 0.80)) == _pairs(case["merged_labels_as_embs"])
        ids, cents = asd.speaker_centroids(labelled, embs)
        assert ids.tolist() == sorted(set(case["labels"]))
        assert np.allclose(cents[:, :6], case["centroids_head"], rtol=1e-5)

        starts, valid = asd._get_speech_windows(y, sr, [asd.Segment(s, e) for s, e in segs], 16000, 1600)
        assert len(starts) == case["n_windows"] and valid.tolist() == case["valid"]
        l2s = asd._labels_to_segments(starts, valid, np.asarray(case["window_labels"]), sr, len(y) / sr)
        assert _triples(l2s) == _pairs(case["labels_to_segments"])
        assert _triples(asd.merge_adjacent(l2s, gap=0.05)) == _pairs(case["merge_adjacent"])


def test_embedding_callers_edge_cases():
    y = np.zeros(16000, np.float32)
    assert asd.embed_segments(y, 16000, [], encode=_fake_encode).shape == (0, 192)
    assert asd.conservative_merge([], np.zeros((0, 192))) == []
    assert asd.merge_adjacent([]) == []
    ids, c = asd.speaker_centroids([asd.Segment(0, 1, -1)], np.ones((1, 192), np.float32))
    assert ids.shape == (0,) and c.shape == (0, 192)
    # fewer than 3 SCD windows -> the segment is kept whole [REF anti_stick_diarize.py:96-98]
    seg = [asd.Segment(0.0, 1.3)]
    assert asd.scd_split_segments(np.zeros(32000, np.float32), 16000, seg, encode=_fake_encode) == seg
    assert asd.frame_reassign(y, 16000, [], [], np.zeros((0, 192)), encode=_fake_encode) == []


def test_diar_diag_score_helpers_match_reference(golden_dir, tmp_path):
    from speech_diarization_amd import diar_diag as dd
    for case in _load(golden_dir, "diar_diag"):
        e, cents, cohort = (np.asarray(case[k], dtype=np.float64) for k in ("embs", "centers", "cohort"))
        w = dd.whiten_l2(e)
        # whitening is defined up to the eigenvector basis of equal eigenvalues; the Gram matrix is basis-free
        assert np.allclose(w @ w.T, np.asarray(case["whiten"]) @ np.asarray(case["whiten"]).T, atol=1e-6)
        assert np.allclose(dd.asnorm_scores(e, cents, cohort, topk=20), case["asnorm"], atol=1e-9)
        scores = np.asarray(case["scores"], dtype=np.float32)
        assert dd.viterbi_hmm(scores, alpha=0.9).tolist() == case["viterbi"]
        assert dd.viterbi_hmm(scores).tolist() == case["viterbi_sticky"]
        dd.save_srt(tmp_path / "a.srt", case["segments"])
        dd.save_csv(tmp_path / "a.csv", case["segments"])
        dd.save_json(tmp_path / "a.json", case["segments"], ["S0", "S1"])
        for k in ("srt", "csv", "json"):
            assert (tmp_path / f"a.{k}").read_text(encoding="utf-8") == case["texts"][k]


# ---- cluster_hdbscan / cluster_hdbscan_two_stage [REF anti_stick_diarize.py:175-270]

class _Scripted:
    """Same stand-in the golden generator put in the reference's HDBSCAN slot: returns the prepared
    label vectors and records constructor arguments and inputs."""

    def __init__(self, script, log, **kw):
        self.script, self.log, self.kw = script, log, kw

    def fit_predict(self, X):
        self.log.append(dict(kwargs=dict(self.kw), X=np.asarray(X, dtype=np.float64)))
        return np.asarray(self.script.pop(0))


def test_cluster_two_stage_matches_reference(golden_dir):
    from speech_diarization_amd import cluster
    cases = _load(golden_dir, "cluster_two_stage")
    seen = set()
    for c in cases:
        embs = np.asarray(c["embs"], dtype=np.float64)
        want = np.asarray(c["labels"], dtype=np.int64)
        if c["kind"] == "sklearn":
            # a real density clusterer through the whole function (scikit-learn's HDBSCAN on both sides)
            from sklearn.cluster import HDBSCAN
            got = asd.cluster_hdbscan_two_stage(embs, c["min_cluster_size"], clusterer_factory=lambda **kw: HDBSCAN(**kw))
            assert np.array_equal(got, want), c["name"]
            assert np.array_equal(asd.cluster_hdbscan_two_stage(embs, c["min_cluster_size"]), want)   # default factory
            continue
        script, log = [list(v) for v in c["script"]], []
        factory = lambda **kw: _Scripted(script, log, **kw)   # noqa: E731
        fn = cluster.cluster_hdbscan if c["kind"] == "scripted_single" else cluster.cluster_hdbscan_two_stage
        got = fn(embs, c["min_cluster_size"], clusterer_factory=factory)
        assert np.array_equal(got, want), c["name"]
        assert str(np.asarray(got).dtype) == c["labels_dtype"]
        assert not script, f"{c['name']}: the reference made {len(c['calls'])} clusterer calls"
        assert len(log) == len(c["calls"])
        for mine, ref in zip(log, c["calls"]):
            assert mine["kwargs"] == ref["kwargs"]
            np.testing.assert_allclose(mine["X"], np.asarray(ref["X"]), rtol=0, atol=1e-12)   # normalisation / centroids / 1 - cos
        seen.add(c["name"])
    # the degenerate branches the reference spells out
    assert {"no_micro_clusters", "one_centroid_lt_min_cluster_size", "stage2_noise_centroid", "label_gap_and_merge"} <= seen


def test_cluster_two_stage_with_injected_ahc_separates_speakers():
    from speech_diarization_amd import cluster, synth
    rows = []
    for k in range(3):
        c = synth.normal(7, f"ahc.c{k}", (1, 32)).astype(np.float64)
        rows.append(4.0 * c + 0.2 * synth.normal(7, f"ahc.n{k}", (10, 32)))
    embs = np.concatenate(rows)
    labels = cluster.cluster_hdbscan_two_stage(embs, 2, clusterer_factory=cluster.AhcClusterer.factory(0.7))
    assert len(set(labels.tolist())) == 3 and all(len(set(labels[10 * k: 10 * k + 10].tolist())) == 1 for k in range(3))


def test_cluster_embeddings_matches_reference(golden_dir):
    """`diar_diag.cluster_embeddings` [REF diar_diag.py:213-229]: constructor arguments (min_cluster_size=6, min_samples=3,
    allow_single_cluster NOT passed), the 1 - cosine matrix, and labels through scikit-learn's HDBSCAN / AHC."""
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import diar_diag
    for c in _load(golden_dir, "cluster_embeddings"):
        embs = np.asarray(c["embs"], dtype=np.float64)
        want = np.asarray(c["labels"], dtype=np.int64)
        if c["name"] == "scripted":
            script, log = [list(want)], []
            got = diar_diag.cluster_embeddings(embs, "hdbscan", affinity=cosine_similarity,
                                               clusterer_factory=lambda **kw: _Scripted(script, log, **kw))
            assert np.array_equal(got, want) and len(log) == 1 == len(c["calls"])
            assert log[0]["kwargs"] == c["calls"][0]["kwargs"] == dict(min_cluster_size=6, min_samples=3, metric="precomputed")
            np.testing.assert_allclose(log[0]["X"], np.asarray(c["calls"][0]["X"]), rtol=0, atol=1e-12)
        elif c["method"] == "hdbscan":
            got = diar_diag.cluster_embeddings(embs, "hdbscan", affinity=cosine_similarity, clusterer_factory=lambda **kw: HDBSCAN(**kw))
            assert np.array_equal(got, want), c["name"]
            assert np.array_equal(diar_diag.cluster_embeddings(embs, "hdbscan", affinity=cosine_similarity), want)   # default factory
        else:
            got = diar_diag.cluster_embeddings(embs, "agglo", c["cos_thr"], affinity=cosine_similarity)
            assert np.array_equal(got, want), c["name"]
    with pytest.raises(ValueError):
        diar_diag.cluster_embeddings(np.zeros((3, 4)), "kmeans", affinity=cosine_similarity)


def test_clustering_small_inputs_do_not_raise():
    """One post-SCD segment (or none) is one speaker, not a ValueError from the density clusterer."""
    from speech_diarization_amd import cluster
    e1 = np.ones((1, 8))
    assert cluster.cluster_hdbscan_two_stage(e1).tolist() == [0]
    assert cluster.cluster_hdbscan_two_stage(np.zeros((0, 8))).size == 0
    assert cluster.cluster_hdbscan(e1).tolist() == [0]
    assert cluster.cluster_hdbscan(np.zeros((0, 8))).size == 0
    assert cluster.hdbscan_precomputed(np.ones((1, 1))).tolist() == [0]
    assert cluster.hdbscan_precomputed(np.ones((4, 4)), min_cluster_size=6, min_samples=3, allow_single_cluster=None).tolist() == [-1] * 4
    two = np.array([[1.0, 0.0], [0.0, 1.0]])
    assert len(cluster.cluster_hdbscan_two_stage(two)) == 2


def test_diarize_with_a_single_segment_returns_one_speaker():
    """diarize() on a recording whose VAD gives ONE short segment must not crash in the clusterer."""
    rng = np.random.default_rng(0)
    y = (0.1 * rng.standard_normal(16000 * 3)).astype(np.float32)

    def encode(w):
        w = np.asarray(w, dtype=np.float32)
        return np.stack([np.abs(np.fft.rfft(r, 382))[:192] for r in w]).astype(np.float32)

    segs = asd.diarize(y, vad_segments=lambda *a, **k: [(0.5, 1.4)], encode=encode, reseg=0)
    assert len(segs) == 1 and segs[0].spk == 0 and (segs[0].start, segs[0].end) == (0.5, 1.4)
    segs = asd.diarize(y, vad_segments=lambda *a, **k: [(0.5, 1.4)], encode=encode, reseg=1)
    assert all(s.spk == 0 for s in segs)


def test_diarize_accepts_the_reference_call_forms(tmp_path):
    """[REF anti_stick_diarize.py:493-512]: first argument a path or an (array, sr) tuple, `target_lufs` third; the
    loaded signal is conditioned by `diar_read_audio` (loudness normalisation unless `lufs=None`, DC removal, 0.97
    pre-emphasis).  A bare array is taken as already conditioned."""
    from scipy.signal import lfilter
    from speech_diarization_amd import audio_io
    rng = np.random.default_rng(1)
    x48 = (0.1 * rng.standard_normal((48000 * 2, 2)) + 0.02).astype(np.float32)        # [n, 2] at 48 kHz with a DC offset
    y, sr = asd.diar_read_audio((x48, 48000), 16000, lufs=None)
    assert sr == 16000 and y.dtype == np.float32 and y.shape == (32000,)
    from scipy.signal import resample_poly
    m0 = resample_poly(x48.T, 1, 3, axis=-1).astype(np.float32).mean(axis=0)
    m = m0 - m0.mean()
    want = lfilter([1.0, -0.97], [1.0], m.astype(np.float64), zi=[2.0 * m[0] - m[1]])[0]      # librosa's preemphasis
    assert np.abs(y - want).max() < 1e-6
    # lufs given: the gain and the +-0.99 clip are applied BEFORE DC removal and pre-emphasis [REF :44-49], never skipped
    y18, _ = asd.diar_read_audio((x48, 48000), 16000, lufs=-18.0)
    g = asd.loudness_normalize(m0, 16000, -18.0)
    assert np.abs(g).max() <= 0.99 and not np.allclose(g, m0)
    g = g.astype(np.float32) - g.astype(np.float32).mean()
    assert np.abs(y18[1:] - (g[1:] - np.float32(0.97) * g[:-1])).max() < 1e-6
    assert asd.diar_read_audio((np.zeros(0, np.float32), 16000), lufs=None)[0].size == 0
    wav = tmp_path / "a.wav"
    audio_io.write_wav16(wav, m, 16000)
    got = {}

    def vad_segments(sig, sr_, **kw):
        got["y"], got["kw"] = sig, kw
        return []

    assert asd.diarize(str(wav), 16000, None, 0.7, 0.3, vad_segments=vad_segments) == []
    assert got["kw"]["on_threshold"] == 0.7 and got["kw"]["off_threshold"] == 0.3          # positional slots 4 and 5, as in the reference
    assert got["y"].shape == (32000,) and abs(float(got["y"][5:].mean())) < 1e-3
    p = np.round(m * 32767.0).clip(-32768, 32767) / 32768.0
    p = p - p.mean()
    assert np.abs(got["y"][1:] - (p[1:] - 0.97 * p[:-1])).max() < 1e-5                     # read from the file, then conditioned
    asd.diarize(str(wav), 16000, -18.0, 0.7, 0.3, vad_segments=vad_segments)               # target_lufs third: a different signal
    assert np.abs(got["y"][1:] - (p[1:] - 0.97 * p[:-1])).max() > 1e-3
    asd.diarize((m, 16000), vad_segments=vad_segments, target_lufs=None)
    assert np.abs(got["y"][1:] - (m[1:] - np.float32(0.97) * m[:-1])).max() < 1e-6
    asd.diarize(m, vad_segments=vad_segments)                                              # bare array: untouched
    assert np.array_equal(got["y"], m)


def test_loudness_normalize_meter(monkeypatch):
    """[REF anti_stick_diarize.py:53-61] BS.1770 integrated loudness -> one gain -> clip to +-0.99.  Known answers of the
    standard: a full-scale 997 Hz sine reads -3.01 LKFS per channel (0.0 as identical stereo), loudness follows the gain
    dB for dB, silence in front of the programme is gated away.  The meter is pyloudnorm when importable, `loudness.py`
    otherwise; an ImportError from INSIDE pyloudnorm is not taken for its absence."""
    import builtins
    from speech_diarization_amd import loudness
    sr = 16000
    tone = np.sin(2 * np.pi * 997.0 * np.arange(5 * sr) / sr)
    m = loudness.Meter(sr)
    assert abs(m.integrated_loudness(tone) + 3.01) < 0.1
    assert abs(m.integrated_loudness(0.1 * tone) - (m.integrated_loudness(tone) - 20.0)) < 1e-9
    assert abs(m.integrated_loudness(np.stack([tone, tone], axis=1))) < 0.1
    assert abs(loudness.Meter(48000).integrated_loudness(np.sin(2 * np.pi * 997.0 * np.arange(5 * 48000) / 48000)) + 3.01) < 0.05
    gated = np.concatenate([np.zeros(5 * sr), 0.1 * tone])
    assert abs(m.integrated_loudness(gated) - m.integrated_loudness(0.1 * tone)) < 0.3
    with pytest.raises(ValueError):
        m.integrated_loudness(tone[:1000])                                                 # shorter than one 400 ms block
    out = asd.loudness_normalize(0.01 * tone, sr, -18.0)
    assert abs(m.integrated_loudness(out) + 18.0) < 1e-6 and np.abs(out).max() <= 0.99
    assert np.abs(asd.loudness_normalize(tone, sr, 0.0)).max() == 0.99                     # +3 dB on a full-scale tone: clipped
    real_import = builtins.__import__

    def broken(name, *a, **k):
        if name == "pyloudnorm":
            raise ModuleNotFoundError("No module named 'scipy.somewhere'", name="scipy.somewhere")
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", broken)
    with pytest.raises(ModuleNotFoundError):
        asd.loudness_normalize(tone, sr)


def test_vad_off_the_default_hop_matches_reference(golden_dir):
    """hop_ms in {2.5, 7.5, 10, 12.5, 16, 20} x pads {0, 12.5, 40, 80} x ends that are / are not clamped to len(mask):
    the reference rounds a clamped end with Python's round and every other time with numpy's [REF vad.py:157-160]."""
    n_cases = n_clamped = 0
    for h in _load(golden_dir, "vad_hops"):
        hop_ms = h["hop_ms"]
        for c in h["cases"]:
            m = _bits(c["mask"])
            m2 = m
            if c["morphed"]:
                m2 = vad.morph_open_close(m, hop_ms, 80.0, 40.0)
                assert np.array_equal(m2, _bits(c["mask2"])), hop_ms
            for p in c["pads"]:
                got = vad.mask_to_segments(m2, hop_ms, c["min_speech_ms"], c["min_gap_ms"], p["speech_pad_ms"])
                assert got == _pairs(p["segments"]), (hop_ms, p["speech_pad_ms"], c["mask"])
                n_cases += 1
                n_clamped += bool(got) and got[-1][1] == round(len(m2) * hop_ms / 1000.0, 3)
    assert n_cases >= 1400 and n_clamped >= 300


def test_mask_to_segments_tie_cases_the_two_roundings_disagree_on():
    # 361 frames of 12.5 ms: 361 * 0.0125 = 4.5125 exactly-ish; Python's round gives 4.513, numpy's 4.512.  The reference
    # uses the former only when the end was clamped to len(mask) [REF vad.py:158-160].
    m = np.zeros(361, bool)
    m[300:361] = True
    assert vad.mask_to_segments(m, 12.5, 25.0, 5.0, 40.0)[-1][1] == 4.513       # clamped: Python round
    m = np.zeros(400, bool)
    m[300:361] = True
    assert vad.mask_to_segments(m, 12.5, 25.0, 5.0, 0.0)[-1][1] == 4.512        # not clamped: numpy round
    m = np.zeros(361, bool)
    m[300:361] = True
    assert vad.mask_to_segments(m, 12.5, 25.0, 5.0, 0.0)[-1][1] == 4.512        # e + pad == total: min() keeps the numpy int
