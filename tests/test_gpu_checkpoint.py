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
