"""The real-checkpoint path: what replaces `EncoderClassifier.from_hparams(source="LanceaKing/spkrec-ecapa-cnceleb")`
[REF speech_encode.py:66-69] for a user who has the weights on disk."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_checkpoint_through_env_var_and_singleton_gives_the_in_memory_engine_bitwise(dev, tmp_path, monkeypatch):
    """A speechbrain-style state dict on disk, found through SD_ECAPA_CKPT, loaded with weights_only=True by the
    `using_ecapa_encoder()` singleton and used by `ecapa_encode_batch`: bitwise the embeddings of an engine built
    from the same arrays in memory."""
    from speech_diarization_amd import speech_encode, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(77, synth.EcapaConfig.small(128))
    ckpt = tmp_path / "embedding_model.ckpt"
    tensors = {k: torch.from_numpy(v) for k, v in sd.items()}
    tensors["blocks.0.norm.norm.num_batches_tracked"] = torch.tensor(123)          # real checkpoints carry integer buffers too
    torch.save(tensors, ckpt)
    monkeypatch.setenv("SD_ECAPA_CKPT", str(ckpt))
    assert hasattr(speech_encode.using_ecapa_encoder, "cache_clear"), "the lru_cache singleton of the reference must be in place"
    speech_encode.using_ecapa_encoder.cache_clear()
    try:
        loaded = speech_encode.load_ecapa_state_dict()
        assert set(loaded) == set(sd) and all(np.array_equal(loaded[k], sd[k]) for k in sd)
        enc = speech_encode.using_ecapa_encoder()
        assert speech_encode.using_ecapa_encoder() is enc                            # process-wide singleton [REF speech_encode.py:64]
        wav = synth.synthetic_segments(9, 6, 24000)
        got = speech_encode.ecapa_encode_batch(wav)
        ref = EmbeddingEngine(sd, dev).embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
        assert got.shape == (6, 192) and got.dtype == np.float32 and np.array_equal(got, ref)
        assert np.array_equal(enc.encode_batch(torch.from_numpy(wav)).squeeze(1).cpu().numpy(), ref)
        monkeypatch.setenv("SD_ECAPA_CKPT", str(tmp_path / "missing.ckpt"))
        with pytest.raises(FileNotFoundError):
            speech_encode.load_ecapa_state_dict()
    finally:
        speech_encode.using_ecapa_encoder.cache_clear()


def _cos_dist(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def test_precision_switch_reaches_the_drop_in_api(dev, tmp_path):
    """BASELINE configs[4] ("fp16 ECAPA") through the reference's names: `set_precision("f16")` (or the environment
    variable SD_ECAPA_PRECISION) makes `using_ecapa_encoder()` / `ecapa_encode_batch()` / `diarize_audio()` run the f16
    engine — the process-wide switch the reference has for its own precision knob [REF diarization_baseline.py:20-21]."""
    import os
    import subprocess
    import sys
    from speech_diarization_amd import audio_io, diarization_baseline as db, speech_encode, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    wav = synth.synthetic_segments(3, 6, 32000)
    assert speech_encode.get_precision() == "f32"
    speech_encode.using_ecapa_encoder.cache_clear()
    try:
        with pytest.warns(RuntimeWarning):
            e32 = speech_encode.ecapa_encode_batch(wav)
        assert speech_encode.using_ecapa_encoder().engine.precision == "f32"
        speech_encode.set_precision("f16")
        with pytest.warns(RuntimeWarning):
            enc = speech_encode.using_ecapa_encoder()
        assert enc.precision == "f16" and enc.engine.precision == "f16" and speech_encode.using_ecapa_encoder() is enc
        e16 = speech_encode.ecapa_encode_batch(wav)
        sd = synth.make_ecapa_state_dict(speech_encode.SYNTHETIC_SEED)      # what the singleton loads when no checkpoint is present
        direct = EmbeddingEngine(sd, dev, precision="f16").embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
        assert np.array_equal(e16, direct)                       # it IS the f16 engine
        assert not np.array_equal(e16, e32) and _cos_dist(e16, e32).max() < 1e-3
        conv = synth.synthetic_conversation(30.0, 2, seed=0)
        p = tmp_path / "m.wav"
        audio_io.write_wav16(p, conv.wav, conv.sr)
        _, det = db.diarize_audio(p, 0.35, 0.1, 2, 6, return_details=True)
        ref = enc.encode_windows(audio_io.read_audio(p, sr=16000, mono=True)[0], det["window_starts"], 32000)
        assert np.array_equal(det["embeddings"], ref)
        speech_encode.set_precision("f32s")                      # f32-split16x3 through the same names
        with pytest.warns(RuntimeWarning):
            es = speech_encode.ecapa_encode_batch(wav)
        assert speech_encode.using_ecapa_encoder().engine.precision == "f32s"
        assert not np.array_equal(es, e32) and _cos_dist(es, e32).max() < 1e-9
        speech_encode.set_precision("f32ns")                     # exact-f32 wide layers, split16x3 narrow ones
        with pytest.warns(RuntimeWarning):
            en = speech_encode.ecapa_encode_batch(wav)
        assert speech_encode.using_ecapa_encoder().engine.precision == "f32ns"
        assert not np.array_equal(en, e32) and not np.array_equal(en, es) and _cos_dist(en, e32).max() < 1e-9
        with pytest.raises(ValueError):
            speech_encode.set_precision("bf16")
    finally:
        speech_encode.set_precision("f32")
        speech_encode.using_ecapa_encoder.cache_clear()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, warnings; sys.path.insert(0, %r); warnings.simplefilter('ignore'); from speech_diarization_amd import speech_encode as s; "
            "print(s.get_precision(), s.using_ecapa_encoder().engine.precision)") % root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, SD_ECAPA_PRECISION="f16"))
    assert res.returncode == 0 and res.stdout.split() == ["f16", "f16"], res.stderr[-1500:]
