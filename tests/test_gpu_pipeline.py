"""GPU twins of the host-level tests: the reference-named API on the HIP path, and the pipeline
producing the same RTTM / cluster assignments as the CPU-oracle path on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cos_dist(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope="module")
def small_encoder(dev):
    """A reduced-width HIP encoder installed as the process-wide singleton, plus its CPU twin."""
    from oracle.ecapa_ref import EcapaRef
    from oracle.pipeline_ref import encode_batch_ref
    from speech_diarization_amd import speech_encode, synth
    sd = synth.make_ecapa_state_dict(1234, synth.EcapaConfig.small(128))
    enc = speech_encode.HipEcapaEncoder(sd, dev)
    speech_encode.using_ecapa_encoder.cache_clear()
    orig = speech_encode.using_ecapa_encoder
    from speech_diarization_amd import ecapa_annote
    speech_encode.using_ecapa_encoder = ecapa_annote.using_ecapa_encoder = lambda device="cuda": enc
    net = EcapaRef(sd, torch.float32)
    cpu = lambda wavs: encode_batch_ref(sd, np.asarray(wavs, np.float32), torch.float32, net).astype(np.float32)  # noqa: E731
    yield enc, cpu
    speech_encode.using_ecapa_encoder = ecapa_annote.using_ecapa_encoder = orig


def test_speech_encode_api(dev, small_encoder):
    from oracle import fbank_ref
    from speech_diarization_amd import ecapa_annote, speech_encode, synth
    enc, cpu = small_encoder
    wav = synth.synthetic_segments(2, 5, 16000)
    keep = wav.copy()
    f = speech_encode.fbank_batch(wav)
    assert isinstance(f, np.ndarray) and f.dtype == np.float32 and f.shape == (5, 101, 80)
    assert np.array_equal(wav, keep)                                   # caller data is not mutated
    assert np.abs(f - fbank_ref.fbank_batch_ref(wav)).max() < 2e-4
    f40 = speech_encode.fbank_batch(wav, n_mels=40, mean_nor=False)
    assert np.abs(f40 - fbank_ref.fbank_batch_ref(wav, n_mels=40, mean_nor=False)).max() < 2e-4
    with pytest.raises(AssertionError):
        speech_encode.fbank_batch(wav[0])
    with pytest.raises(ValueError):
        speech_encode.fbank_batch(wav, sr=300)                         # a 7-sample window
    e = speech_encode.ecapa_encode_batch(wav.astype(np.float64))       # any float dtype, cast like .float()
    assert e.shape == (5, 192) and e.dtype == np.float32
    assert _cos_dist(e, cpu(wav)).max() < 1e-5
    t = enc.encode_batch(torch.from_numpy(wav))
    assert t.shape == (5, 1, 192) and t.device == dev
    model = ecapa_annote.ECAPAEncoder(0)
    assert model.dimension == 192
    y2 = model(torch.from_numpy(wav).to(dev))
    y3 = model(torch.from_numpy(wav)[:, None, :])
    assert y2.shape == (5, 192) and torch.equal(y2, y3) and torch.equal(y2, t.squeeze(1))
    with pytest.raises(FileNotFoundError):
        ecapa_annote.ERes2NetV2Encoder()
    with pytest.raises(ValueError, match="too short"):
        speech_encode.ecapa_encode_batch(np.zeros((1, 300), np.float32))
    assert speech_encode.ecapa_encode_batch(np.zeros((0, 16000), np.float32)).shape == (0, 192)


def test_encoder_is_callable_from_another_thread(small_encoder):
    """The gradio UI calls the pipeline from a worker thread [REF diarize-webui.py:142-160]."""
    import threading
    from speech_diarization_amd import speech_encode, synth
    wav = synth.synthetic_segments(3, 4, 16000)
    main = speech_encode.ecapa_encode_batch(wav)
    box = {}
    th = threading.Thread(target=lambda: box.update(e=speech_encode.ecapa_encode_batch(wav)))
    th.start(); th.join()
    assert np.array_equal(box["e"], main)


def test_frame_reassign_gpu_matches_cpu_path(small_encoder):
    """Windows gathered / embedded / matched on the GPU give the labels of the literal host path."""
    from speech_diarization_amd import anti_stick_diarize as asd, synth
    enc, cpu = small_encoder
    conv = synth.synthetic_conversation(24.0, 2, seed=3)
    y = conv.wav
    speech = [asd.Segment(s, e) for s, e, _ in conv.turns]
    segs = [asd.Segment(s, e, k) for s, e, k in conv.turns]
    embs_gpu = asd.embed_segments(y, 16000, segs)
    embs_cpu = asd.embed_segments(y, 16000, segs, encode=cpu)
    assert _cos_dist(embs_gpu, embs_cpu).max() < 1e-5
    a = asd.frame_reassign(y, 16000, speech, segs, embs_gpu)
    b = asd.frame_reassign(y, 16000, speech, segs, embs_cpu, encode=cpu)
    assert [(s.start, s.end, s.spk) for s in a] == [(s.start, s.end, s.spk) for s in b]
    assert len({s.spk for s in a}) == 2
    c = asd.scd_split_segments(y, 16000, [asd.Segment(0.3, 20.0)], thr=1.25)
    d = asd.scd_split_segments(y, 16000, [asd.Segment(0.3, 20.0)], thr=1.25, encode=cpu)
    assert [(s.start, s.end) for s in c] == [(s.start, s.end) for s in d]


def test_config0_on_gpu_gives_the_cpu_rttm(small_encoder, tmp_path):
    """diarization_baseline on the 60 s 2-speaker WAV: HIP path vs PyTorch-CPU path on the same inputs ->
    identical cluster assignments, identical RTTM, DER 0."""
    from speech_diarization_amd import audio_io, diarization_baseline as db, rttm, synth
    enc, cpu = small_encoder
    conv = synth.synthetic_conversation(60.0, 2, seed=0)
    wav = tmp_path / "meeting.wav"
    audio_io.write_wav16(wav, conv.wav, conv.sr)
    seg_g, det_g = db.diarize_audio(wav, 0.35, 0.1, 2, 6, rttm_filepath=tmp_path / "gpu.rttm", return_details=True)
    seg_c, det_c = db.diarize_audio(wav, 0.35, 0.1, 2, 6, rttm_filepath=tmp_path / "cpu.rttm", encoder=cpu, return_details=True)
    assert _cos_dist(det_g["embeddings"], det_c["embeddings"]).max() < 1e-3          # north_star tolerance
    assert np.array_equal(det_g["labels"], det_c["labels"])                          # identical cluster assignments
    assert (tmp_path / "gpu.rttm").read_text() == (tmp_path / "cpu.rttm").read_text()
    assert rttm.der(seg_c, seg_g) == 0.0
    assert rttm.der([(s, e, f"T{k}") for s, e, k in conv.turns], seg_g) < 0.10
    assert np.abs(det_g["affinity"] - det_c["affinity"]).max() < 1e-3


def test_full_diarize_pipeline_runs_on_gpu(small_encoder):
    from speech_diarization_amd import anti_stick_diarize as asd, synth
    conv = synth.synthetic_conversation(30.0, 2, seed=5)
    for clusterer in ("hdbscan_two_stage", "ahc", "ahc_affinity"):      # the reference's glue over HDBSCAN / injected AHC; one-stage AHC
        out = asd.diarize(conv.wav, 16000, scd_thr=3.0, cluster_cos=0.2, clusterer=clusterer)
        assert out and all(isinstance(s, asd.Segment) and s.spk is not None and s.end > s.start for s in out), clusterer
    assert asd.diarize(np.zeros(32000, np.float32), 16000) == []                     # no speech -> []


def test_first_encoder_call_from_racing_worker_threads(tmp_path):
    """A fresh process whose FIRST launches of every kernel come from two worker threads at once (the reference's web UI
    calls the pipeline from a worker thread [REF diarize-webui.py:142-160]): the per-device one-time kernel attributes
    (dynamic LDS > 64 KB) must be set exactly once, under a lock, before either launch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, threading
sys.path.insert(0, {root!r})
import numpy as np, torch
from speech_diarization_amd import synth
from speech_diarization_amd.engine import EmbeddingEngine
dev = torch.device("cuda", 0)
sd = synth.make_ecapa_state_dict(1234, synth.EcapaConfig.small(256))
engines = [EmbeddingEngine(sd, dev, precision=p) for p in ("f32", "f16")]
wav = torch.from_numpy(synth.synthetic_segments(3, 8, 32000)).to(dev)
out, err = {{}}, []
def work(i):
    try:
        with torch.cuda.device(dev):
            out[i] = engines[i].embed(wav).cpu().numpy()
    except Exception as e:
        err.append(repr(e))
ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not err, err
again = [e.embed(wav).cpu().numpy() for e in engines]
assert np.array_equal(out[0], again[0]) and np.array_equal(out[1], again[1])
print("ok")
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stderr[-2000:]


def test_diarize_on_gpu_equals_diarize_with_the_cpu_encoder(small_encoder):
    """`anti_stick_diarize.diarize` end to end [REF anti_stick_diarize.py:493-560]: VAD -> SCD -> embed -> two-stage
    clustering -> merge -> re-embed -> frame reassignment, HIP path (windows read in place, cosine sites on the device)
    against the same function with the CPU oracle encoder injected: the same segments, one for one."""
    from speech_diarization_amd import anti_stick_diarize as asd, synth
    enc, cpu = small_encoder
    for seed, seconds, spk in ((5, 40.0, 2), (9, 50.0, 3)):
        conv = synth.synthetic_conversation(seconds, spk, seed=seed)
        for kw in (dict(clusterer="ahc", cluster_cos=0.2, scd_thr=3.0), dict(clusterer="hdbscan_two_stage", scd_thr=1.5)):
            g = asd.diarize(conv.wav, 16000, **kw)
            c = asd.diarize(conv.wav, 16000, encode=cpu, **kw)
            assert g and [(s.start, s.end, s.spk) for s in g] == [(s.start, s.end, s.spk) for s in c], (seed, kw)


def test_batches_in_flight_give_the_bits_of_one_call_at_a_time(small_encoder):
    """`ecapa_encode_batches` / `HipEcapaEncoder.encode_batches`: the batch loop of `embed_segments` [REF anti_stick_diarize.py:150-171] with
    two or three batches in flight on separate streams and workspaces.  Ragged batch sizes and padded lengths (also past the one-launch
    fbank kernel's 201 frames), an empty batch, more lanes than batches; every batch bit for bit what `ecapa_encode_batch` returns, fresh
    arrays, inputs untouched, and `embed_segments` on the GPU equal to its loop with the one-at-a-time encoder."""
    from speech_diarization_amd import anti_stick_diarize as asd, speech_encode, synth
    enc, _ = small_encoder
    shapes = [(32, 16000), (7, 40000), (32, 20800), (1, 8000), (0, 16000), (32, 16000), (13, 33600), (32, 12000), (5, 16160)]
    batches = [synth.synthetic_segments(40 + i, b, n) for i, (b, n) in enumerate(shapes)]
    keep = [b.copy() for b in batches]
    want = [speech_encode.ecapa_encode_batch(b) if len(b) else np.empty((0, 192), np.float32) for b in batches]
    for lanes in (1, 2, 3, 16):
        got = speech_encode.ecapa_encode_batches(batches, lanes=lanes)
        assert len(got) == len(want)
        for g, w, (b, n) in zip(got, want, shapes):
            assert isinstance(g, np.ndarray) and g.dtype == np.float32 and g.shape == (b, 192) and g.flags.owndata
            assert np.array_equal(g, w), (lanes, b, n)
    assert all(np.array_equal(a, b) for a, b in zip(batches, keep))
    assert speech_encode.ecapa_encode_batches([]) == []
    assert np.array_equal(enc.encode_batches(batches[:1], lanes=2)[0], want[0])             # fewer batches than lanes
    with pytest.raises(ValueError, match="too short"):                                       # a refused batch in the middle: the error of the single call,
        speech_encode.ecapa_encode_batches([batches[0], np.zeros((2, 300), np.float32), batches[2]])
    again = speech_encode.ecapa_encode_batches(batches[:3])                                  # and the lanes are clean afterwards
    assert all(np.array_equal(a, w) for a, w in zip(again, want[:3]))
    conv = synth.synthetic_conversation(40.0, 2, seed=5)
    y = conv.wav.astype(np.float32)
    segs = [asd.Segment(0.3 * k, 0.3 * k + 0.2 + 0.05 * (k % 17)) for k in range(100)]      # 100 segments -> 4 batches of ragged lengths
    one_at_a_time = asd.embed_segments(y, conv.sr, segs, encode=speech_encode.ecapa_encode_batch)
    assert np.array_equal(asd.embed_segments(y, conv.sr, segs), one_at_a_time)
