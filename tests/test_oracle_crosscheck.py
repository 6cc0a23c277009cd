"""The oracle against INDEPENDENT third-party implementations that are installed in the image.

torchaudio and speechbrain (the libraries the reference calls) are absent, so the oracle for fbank
and ECAPA-TDNN is a restatement (oracle/__init__.py: "parity unpinned").  Two packages that ARE here
carry their own, separately written code for the same published algorithms:

* `transformers.audio_utils` — `mel_filter_bank` / `spectrogram` / `window_function`, documented
  as matching torchaudio's `melscale_fbanks` / `Spectrogram`;
* `transformers.models.qwen2_5_omni` — `ECAPA_TimeDelayNet`, a speechbrain-derived ECAPA-TDNN
  (same blocks, reflect-"same" convolutions, Res2Net chain, SE, attentive statistics pooling) with
  the BatchNorms removed.

Agreement does not pin the oracle to the reference's exact library versions; it does rule out a
structural misreading of either algorithm.
"""
import types

import numpy as np
import pytest
import torch

from oracle import ecapa_ref, fbank_ref
from speech_diarization_amd import features, synth

au = pytest.importorskip("transformers.audio_utils")


def test_htk_mel_filters_match_transformers():
    ours = fbank_ref.melscale_fbanks_htk(201, 20.0, 7900.0, 80, 16000)
    theirs = au.mel_filter_bank(num_frequency_bins=201, num_mel_filters=80, min_frequency=20.0,
                                max_frequency=7900.0, sampling_rate=16000, norm=None, mel_scale="htk")
    assert ours.shape == theirs.shape == (201, 80)
    assert np.abs(ours - theirs).max() < 1e-9
    # the product path's table (f32) is the same matrix
    assert np.abs(features.mel_filters_torchaudio() - theirs).max() < 1e-6


def test_torchaudio_front_end_matches_transformers_spectrogram():
    rng = np.random.default_rng(3)
    wavs = (0.1 * rng.standard_normal((3, 4000))).astype(np.float64)
    wavs[1] = 0.3 * np.sin(2 * np.pi * 1234.0 * np.arange(4000) / 16000.0)
    window = au.window_function(400, "hann", periodic=True)
    fb = au.mel_filter_bank(201, 80, 20.0, 7900.0, 16000, norm=None, mel_scale="htk")
    ours = fbank_ref.fbank_batch_ref(wavs, mean_nor=False)
    for b in range(wavs.shape[0]):
        mel = au.spectrogram(wavs[b], window, frame_length=400, hop_length=160, fft_length=400, power=2.0,
                             center=True, pad_mode="reflect", onesided=True, mel_filters=fb, mel_floor=0.0,
                             dtype=np.float64)
        theirs = np.log(mel.T + 1e-6)                     # [REF speech_encode.py:32-33]
        assert theirs.shape == ours[b].shape == (26, 80)
        assert np.abs(ours[b] - theirs).max() < 1e-6         # their FFT buffer is complex64


def _hf_ecapa(cfg: synth.EcapaConfig):
    mod = pytest.importorskip("transformers.models.qwen2_5_omni.modeling_qwen2_5_omni")
    hf_cfg = types.SimpleNamespace(mel_dim=cfg.input_size, enc_channels=list(cfg.channels),
                                   enc_kernel_sizes=list(cfg.kernel_sizes), enc_dilations=list(cfg.dilations),
                                   enc_res2net_scale=cfg.res2net_scale, enc_se_channels=cfg.se_channels,
                                   enc_attention_channels=cfg.attention_channels, enc_dim=cfg.lin_neurons)
    return mod.ECAPA_TimeDelayNet(hf_cfg).double().eval()


def test_ecapa_topology_matches_transformers_ecapa():
    cfg = synth.EcapaConfig.small(64, lin=24)
    sd = synth.make_ecapa_state_dict(seed=7, cfg=cfg)
    # that implementation has no BatchNorm: make ours the identity ((x - 0) / sqrt(var + eps) * 1 + 0)
    for k in list(sd):
        if k.endswith(".norm.weight"):
            sd[k] = np.ones_like(sd[k], dtype=np.float64)
        elif k.endswith(".norm.bias") or k.endswith(".norm.running_mean"):
            sd[k] = np.zeros_like(sd[k], dtype=np.float64)
        elif k.endswith(".norm.running_var"):
            sd[k] = np.full(sd[k].shape, 1.0 - ecapa_ref.BN_EPS, dtype=np.float64)

    net = _hf_ecapa(cfg)
    target = net.state_dict()
    loaded = set()
    for k, v in sd.items():
        if ".norm." in k:
            continue
        # speechbrain wraps every Conv1d once more than that implementation does
        cands = [k.replace(".conv.conv.", ".conv."), k.replace(".conv.", ".", 1), k.replace(".conv.conv.", ".")]
        name = next(c for c in cands if c in target and c not in loaded)
        assert tuple(target[name].shape) == tuple(v.shape), (k, name)
        target[name] = torch.as_tensor(np.asarray(v, dtype=np.float64))
        loaded.add(name)
    assert loaded == set(target), sorted(set(target) - loaded)
    net.load_state_dict(target)

    rng = np.random.default_rng(11)
    feats = torch.from_numpy(rng.standard_normal((3, 57, 80)))
    with torch.no_grad():
        theirs = net(feats).numpy()
    ours = ecapa_ref.EcapaRef(sd, dtype=torch.float64).forward_features(feats).numpy().reshape(3, -1)
    assert theirs.shape == ours.shape == (3, 24)
    assert np.abs(ours - theirs).max() < 1e-9 * max(1.0, np.abs(theirs).max())
    # and the naive numpy formulation agrees with both
    naive = ecapa_ref.ecapa_forward_numpy(sd, feats.numpy()).reshape(3, -1)
    assert np.abs(naive - theirs).max() < 1e-9 * max(1.0, np.abs(theirs).max())
