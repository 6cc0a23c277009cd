"""GPU parity of the fused fbank kernel and the ECAPA-TDNN forward against the CPU oracle.

The oracle is float64 (ground truth); the HIP path is exact f32.  north_star's bar is
1e-3 cosine on embeddings; the bounds asserted here are much tighter and written per test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cos_dist(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
@pytest.mark.parametrize("B,n", [(1, 32000), (5, 32000), (3, 16000), (4, 4999), (7, 3200), (2, 48123)])
def test_fbank_matches_oracle(dev, kind, B, n):
    from oracle import fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    wav = synth.synthetic_segments(B * 7 + n, B, n)
    wav[0, : n // 3] *= 0.01                      # a quiet stretch so the top_db floor is exercised
    plan = FbankPlan(kind)
    got = fbank_device(torch.from_numpy(wav).to(dev), plan, mean_norm=True).cpu().numpy()
    ref = fbank_ref.fbank_batch_ref(wav) if kind == "torchaudio" else fbank_ref.speechbrain_fbank_ref(wav)
    assert got.shape == ref.shape == (B, 1 + n // 160, 80)
    # ln / dB of a 200-term f32 power sum: absolute error bound 2e-4 (dB scale: 1e-3)
    tol = 2e-4 if kind == "torchaudio" else 1e-3
    assert np.abs(got - ref).max() < tol, np.abs(got - ref).max()


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
@pytest.mark.parametrize("level", [1.0, 1e-2, 3e-5, 1e-6])
def test_fbank_keeps_its_accuracy_at_every_signal_level(dev, kind, level):
    """The DFT runs on the f16 matrix cores with every operand split hi + lo; the samples are scaled by 2^10 first so that
    the low halves stay out of the f16 subnormals.  Full scale (|x| up to 1: folded sums up to 2048, no f16 overflow), a
    quiet stretch, one 16-bit LSB (3e-5) and far below: the error against float64 does not depend on the level."""
    from oracle import fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    wav = synth.synthetic_segments(17, 3, 16000, std=0.3).astype(np.float64)
    wav = np.clip(wav / np.abs(wav).max(), -1, 1) * level            # peak exactly `level`
    wav[1, 5000:5400] = level                                          # a clipped plateau: x[k] + x[400 - k] = 2 * level
    wav = wav.astype(np.float32)
    plan = FbankPlan(kind)
    got = fbank_device(torch.from_numpy(wav).to(dev), plan, mean_norm=False).cpu().numpy()
    ref = fbank_ref.fbank_batch_ref(wav, mean_nor=False) if kind == "torchaudio" else fbank_ref.speechbrain_fbank_ref(wav, mean_norm=False)
    assert np.isfinite(got).all()
    tol = 2e-4 if kind == "torchaudio" else 1e-3
    assert np.abs(got - ref).max() < tol, (level, np.abs(got - ref).max())


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
@pytest.mark.parametrize("n", [801, 3203, 16001, 32100, 32102, 35003])
def test_fbank_both_kernels_at_their_length_switch(dev, kind, n):
    """Utterances whose padded signal fits the CU's LDS beside the table ring (n <= 32 100 samples: 201 frames) take the one-launch kernel (sd_fbank_utt16.hip: one workgroup per utterance,
    factored DFT, floor + mean in LDS), longer ones the folded-DFT kernel + finalize pass (sd_fbank.hip): both sides of the switch,
    lengths that are not multiples of 4 (the one-launch kernel loads groups of four samples) and one that ends inside a tile, against
    float64 with a 60 dB quieter stretch (the one-launch kernel scales by the utterance's PEAK: the quiet part must keep its accuracy)."""
    from oracle import fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    wav = synth.synthetic_segments(n, 3, n, std=0.2)
    wav[1, n // 3: 2 * n // 3] *= 1e-3
    wav[2] *= 1e-2
    plan = FbankPlan(kind)
    for mean_norm in (True, False):
        got = fbank_device(torch.from_numpy(wav).to(dev), plan, mean_norm=mean_norm).cpu().numpy()
        ref = (fbank_ref.fbank_batch_ref(wav, mean_nor=mean_norm) if kind == "torchaudio" else fbank_ref.speechbrain_fbank_ref(wav, mean_norm=mean_norm))
        assert got.shape == ref.shape == (3, 1 + n // 160, 80)
        tol = 2e-4 if kind == "torchaudio" else 1e-3
        assert np.abs(got - ref).max() < tol, (n, mean_norm, np.abs(got - ref).max())


def test_fbank_known_answers(dev):
    """all-zero waveform -> every bin log(eps) -> exactly 0 after mean-norm; pure tone -> peak at its mel bin."""
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan, mel_filters_torchaudio
    plan = FbankPlan("torchaudio")
    z = fbank_device(torch.zeros(2, 8000, device=dev), plan, mean_norm=True)
    assert z.abs().max() < 4e-6            # f32 sum/T of a constant is exact to ~1 ulp of ln(1e-6) = -13.8
    raw = fbank_device(torch.zeros(1, 8000, device=dev), plan, mean_norm=False)
    assert torch.allclose(raw, torch.full_like(raw, float(np.log(1e-6))), rtol=0, atol=1e-5)
    t = np.arange(16000) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)[None]
    f = fbank_device(torch.from_numpy(tone).to(dev), plan, mean_norm=False).cpu().numpy()[0]
    expect = int(np.argmax(mel_filters_torchaudio()[25]))          # 1000 Hz = DFT bin 25
    assert abs(int(np.argmax(f[50])) - expect) <= 1


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
def test_fbank_full_size_properties(dev, kind):
    """10 000 segments of 2 s (BASELINE configs[1] size): finite, per-utterance mean removed, rows independent
    (duplicates planted far apart come out bitwise equal), gain behaves as the front end says (ln: a gain g adds
    2 ln g before mean removal, i.e. nothing after it where x >> eps; dB with top_db: invariant), and a strided
    sample of rows matches the float64 oracle."""
    from oracle import fbank_ref
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    plan = FbankPlan(kind)
    n = 10000
    g = torch.Generator(device=dev).manual_seed(5)
    wav = (torch.randn((n, 32000), generator=g, device=dev) * 0.1).clamp_(-1.0, 1.0)
    wav[9000] = wav[3]
    wav[4321] = 0.5 * wav[10]
    f = fbank_device(wav, plan, mean_norm=True)
    assert f.shape == (n, 201, 80) and bool(torch.isfinite(f).all())
    assert float(f.mean(dim=1).abs().max()) < 2e-5
    assert torch.equal(f[9000], f[3])
    assert float((f[4321] - f[10]).abs().max()) < 2e-2          # ln(x + 1e-6): exact only where x >> eps
    idx = [0, 3, 2500, 4321, 9999]
    ref_fn = fbank_ref.fbank_batch_ref if kind == "torchaudio" else fbank_ref.speechbrain_fbank_ref
    ref = ref_fn(wav[idx].cpu().numpy().astype(np.float64))
    tol = 5e-4 if kind == "torchaudio" else 2e-3          # ln vs dB units
    assert np.abs(f[idx].cpu().numpy() - ref).max() < tol


def test_fbank_short_segments_and_errors(dev):
    from oracle import fbank_ref
    from speech_diarization_amd import synth, _native
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    plan = FbankPlan("speechbrain")
    wav = synth.synthetic_segments(1, 9, 1600)             # T = 11 < 32: one tile row per utterance
    got = fbank_device(torch.from_numpy(wav).to(dev), plan).cpu().numpy()
    assert np.abs(got - fbank_ref.speechbrain_fbank_ref(wav)).max() < 1e-3
    assert fbank_device(torch.zeros(0, 1600, device=dev), plan).shape == (0, 11, 80)
    with pytest.raises(_native.SdError, match="reflect"):
        fbank_device(torch.zeros(1, 150, device=dev), FbankPlan("torchaudio"))
    with pytest.raises(AssertionError):
        fbank_device(torch.zeros(1600, device=dev), plan)


@pytest.fixture(params=["auto", "split32", "tiles128", "tiles64", "rows80", "rows96", "rows112", "wide256"])
def conv_kernel(request):
    """"tiles128" pins the 128x128 f32 conv kernel; "auto" lets small launches take the 32x32 split-K kernel."""
    from speech_diarization_amd import _native as N
    lib = N.load()
    # "auto": small launches take the 64x64 ring kernel (time-axis convs) or the 32x32 split-K kernel (per-segment layers);
    # "split32": the 32x32 split-K kernel for both
    N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, 0 if request.param != "auto" else -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, 0 if request.param not in ("auto", "split32") else -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_WIDE_TILES, 0 if request.param == "wide256" else -1), "sd_set_tuning")   # cout >= 1024 layers: the 256x256 ring kernel
    # 128x64 tiles: by the rule ("auto"), always ("tiles64"), never (the pinned kernels)
    N.check(lib.sd_set_tuning(N.SD_TUNE_HALF_TILES, {"auto": -1, "split32": -1, "tiles64": 1}.get(request.param, 0)), "sd_set_tuning")
    # tiles of 80 / 96 / 112 rows: by the rule ("auto"), that height wherever the layer allows ("rowsNN"), never (the pinned kernels)
    N.check(lib.sd_set_tuning(N.SD_TUNE_TILE_ROWS, int(request.param[4:]) if request.param.startswith("rows") else (-1 if request.param in ("auto", "split32") else 0)), "sd_set_tuning")
    yield request.param
    N.check(lib.sd_set_tuning(N.SD_TUNE_HALF_TILES, -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_TILE_ROWS, -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_S64_TILES, -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_SKINNY_TILES, -1), "sd_set_tuning")
    N.check(lib.sd_set_tuning(N.SD_TUNE_WIDE_TILES, -1), "sd_set_tuning")


@pytest.mark.parametrize("width,B,n", [(64, 6, 16000), (128, 3, 32000)])
def test_ecapa_small_geometry_matches_oracle(dev, conv_kernel, width, B, n):
    from oracle import ecapa_ref, fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(21, synth.EcapaConfig.small(width))
    wav = synth.synthetic_segments(3, B, n)
    eng = EmbeddingEngine(sd, dev, max_batch=4)         # max_batch < B: exercises micro-batching
    got = eng.embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    feats = fbank_ref.speechbrain_fbank_ref(wav)
    ref = ecapa_ref.EcapaRef(sd, torch.float64).forward_features(torch.from_numpy(feats)).numpy()
    assert got.shape == ref.shape == (B, 192)
    assert _cos_dist(got, ref).max() < 1e-6
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("B,n", [(4, 32000), (5, 16000), (2, 100000), (3, 9600)])
def test_ecapa_full_geometry_matches_oracle(dev, conv_kernel, B, n):
    """The spkrec-ecapa geometry (C=1024, 20.8 M parameters): the bench's 2 s segments, the reference's 1 s SCD /
    reassignment windows (tiles spanning three segments), a long VAD segment as embed_segments pads them
    (T = 626: attentive pooling falls back to conv + streaming pooling) and 0.6 s windows (T = 61: statistics from
    the separate kernels)."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    wav = synth.synthetic_segments(0, B, n)
    eng = EmbeddingEngine(sd, dev)
    got = eng.embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    ref = pipeline_ref.encode_batch_ref(sd, wav, torch.float64)
    cd = _cos_dist(got, ref)
    assert cd.max() < 1e-5, cd           # north_star bar: 1e-3
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()


def test_ecapa_zero_padding_counts_as_signal(dev):
    """Zero-padded tails participate in mean-norm / SE / ASP exactly as in the reference
    (no lengths are passed, [REF anti_stick_diarize.py:163-168])."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(21, synth.EcapaConfig.small(64))
    wav = synth.synthetic_segments(9, 3, 24000)
    wav[1, 9000:] = 0.0
    wav[2, 20000:] = 0.0
    got = EmbeddingEngine(sd, dev).embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    ref = pipeline_ref.encode_batch_ref(sd, wav, torch.float64)
    assert _cos_dist(got, ref).max() < 1e-6


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
def test_fbank_windows_in_place_equals_gathered_bitwise(dev, kind):
    """`sd_fbank_windows_f32` (rows read at starts[b] inside ONE resident signal, zeros past either end) against
    `sd_fbank_f32` on the gathered, zero-padded [B, n] matrix the reference's callers build on the host
    [REF anti_stick_diarize.py:82-100, 163-168, 396-430]: the same bits, for overlapping windows, windows that hang
    over the end of the signal, a window that starts before it, duplicates, and 1 s / 2 s / odd lengths."""
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device, fbank_windows_device
    from speech_diarization_amd.features import FbankPlan
    plan = FbankPlan(kind)
    y = synth.synthetic_segments(23, 1, 16000 * 30)[0]
    yd = torch.from_numpy(y).to(dev)
    for n, hop in ((32000, 4000), (16000, 1600), (4999, 777)):
        starts = np.concatenate([np.arange(0, len(y) - n // 2, hop), [len(y) - 5, len(y) - n + 1, 0, 0, -300]]).astype(np.int64)
        rows = np.zeros((len(starts), n), np.float32)
        for i, s in enumerate(starts):
            lo, hi = max(s, 0), min(s + n, len(y))
            rows[i, lo - s: hi - s] = y[lo:hi]
        want = fbank_device(torch.from_numpy(rows).to(dev), plan, mean_norm=True)
        got = fbank_windows_device(yd, torch.from_numpy(starts), n, plan, mean_norm=True)
        assert got.shape == want.shape and torch.equal(got, want), (kind, n)


def test_embed_windows_equals_embed_of_gathered_rows(dev):
    """The engine entry the pipeline uses (`EmbeddingEngine.embed_windows`): bitwise `embed(gathered)`, full geometry,
    more windows than one micro-batch, both precisions."""
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    y = synth.synthetic_segments(29, 1, 16000 * 20)[0]
    starts = np.concatenate([np.arange(0, len(y) - 16000, 4000), [len(y) - 100]]).astype(np.int64)
    rows = np.zeros((len(starts), 32000), np.float32)
    for i, s in enumerate(starts):
        piece = y[s:s + 32000]
        rows[i, : len(piece)] = piece
    for precision in ("f32", "f16"):
        eng = EmbeddingEngine(sd, dev, max_batch=32, precision=precision)
        a = eng.embed_windows(torch.from_numpy(y).to(dev), torch.from_numpy(starts), 32000)
        b = eng.embed(torch.from_numpy(rows).to(dev))
        assert a.shape == (len(starts), 192) and torch.equal(a, b), precision
    assert eng.embed_windows(torch.from_numpy(y).to(dev), torch.zeros(0, dtype=torch.int64), 32000).shape == (0, 192)
    with pytest.raises(ValueError, match="too short"):
        eng.embed_windows(torch.from_numpy(y).to(dev), torch.zeros(1, dtype=torch.int64), 300)


@pytest.mark.parametrize("kind", ["torchaudio", "speechbrain"])
def test_fbank_out_of_range_samples_are_clipped_not_nan(dev, kind):
    """sd_hip.h: |x| <= 16 is exact; larger samples are clipped to +-16 (un-normalised int16-scale floats used to turn a
    whole segment into NaNs: the folded sums are scaled by 2^10 before the f16 split).  Non-finite samples only affect their row."""
    from oracle import fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    plan = FbankPlan(kind)
    ref_fn = (lambda w: fbank_ref.fbank_batch_ref(w, mean_nor=False)) if kind == "torchaudio" else (lambda w: fbank_ref.speechbrain_fbank_ref(w, mean_norm=False))
    wav = synth.synthetic_segments(31, 3, 16000, std=0.3)
    big = wav.copy()
    big[0] *= 16.0 / np.abs(big[0]).max()              # peak exactly 16: still exact
    big[1] *= 32768.0                                   # int16-scale floats
    big[2, 3000] = 1e30
    got = fbank_device(torch.from_numpy(big).to(dev), plan, mean_norm=False).cpu().numpy()
    assert np.isfinite(got).all()
    tol = 2e-4 if kind == "torchaudio" else 1e-3
    assert np.abs(got[0] - ref_fn(big[:1])[0]).max() < tol
    clipped = np.clip(big, -16.0, 16.0)
    assert np.abs(got - ref_fn(clipped)).max() < tol
    nan = wav.copy(); nan[1, 777] = np.nan; nan[2, 9000] = np.inf         # a NaN sample: NaN features for ITS row (sd_hip.h), an infinite
    g1 = fbank_device(torch.from_numpy(wav).to(dev), plan, mean_norm=False).cpu().numpy()   # one saturates; the neighbours are untouched
    for mean_norm in (False, True):
        g2 = fbank_device(torch.from_numpy(nan).to(dev), plan, mean_norm=mean_norm).cpu().numpy()
        assert np.isnan(g2[1]).all() and np.isfinite(g2[0]).all() and np.isfinite(g2[2]).all()
        if not mean_norm:
            assert np.array_equal(g2[0], g1[0])


@pytest.mark.parametrize("n", [16000, 48000])
def test_a_nan_sample_gives_a_nan_embedding_for_its_row_only(dev, n):
    """the clamp in front of the f16 sample image must not launder a NaN into a plausible embedding.  Both fbank kernels
    (one launch up to 201 frames, the folded one beyond), first / middle / last sample, windows read in place included."""
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234, synth.EcapaConfig.small(128))
    eng = EmbeddingEngine(sd, dev)
    wav = synth.synthetic_segments(3, 5, n, std=0.2)
    clean = eng.embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    bad = wav.copy()
    bad[0, 0] = np.nan; bad[2, n // 2] = np.nan; bad[4, n - 1] = np.nan
    got = eng.embed(torch.from_numpy(bad).to(dev)).cpu().numpy()
    assert np.isnan(got[[0, 2, 4]]).all()
    assert np.array_equal(got[[1, 3]], clean[[1, 3]])


def _fbank_error_inputs(seed, n):
    from speech_diarization_amd import synth
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    white = synth.synthetic_segments(seed, 1, n, std=0.1)[0]
    tone = 0.5 * np.sin(2 * np.pi * (300.0 + 3000.0 * rng.random()) * t)
    voices = synth.synthetic_conversation(max(2.0, n / 16000.0), 2, seed=seed).wav[:n].astype(np.float32)
    step = white.copy()
    step[: n // 2] *= 1e-3
    return {"white": white, "tone + broadband 60 dB below": (tone + 0.5e-3 * rng.standard_normal(n)).astype(np.float32),
            "tone + broadband 40 dB below": (tone + 0.5e-2 * rng.standard_normal(n)).astype(np.float32), "synthetic voices": voices,
            "60 dB level step": step}


FBANK_R0, FBANK_K_BAR = 3e-5, 18.0


@pytest.mark.parametrize("n", [32000, 48000])
def test_fbank_error_stays_inside_the_f32_dft_model(dev, n, capsys):
    """the split-f16 DFT's error is not a flat number, it scales with the in-frame dynamic range, as ANY f32-class DFT's
    does (oracle/fbank_ref.py `log_mel_error_unit` has the derivation): a bin DR dB below its frame's level errs by ~2 k 2^-22 10^(DR/20) in ln
    units.  Bar: |err(t, m)| <= r0 + K unit(t, m) with r0 = 3e-5 (mel product + log) and K = 18; torch.stft in f32, the arithmetic class of the
    reference's own path, needs k = 5.5 on the same inputs (profiles/r05_fbank_error_model.txt).  32 seeds x {white noise, a loud tone over broadband
    noise 60 / 40 dB below it (the adversarial case), synthetic voices, a 60 dB level step} x both front ends; n = 32 000: the one-launch kernel,
    n = 48 000: the folded one.  The achieved k must leave 2x headroom."""
    from oracle import fbank_ref
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    worst = {}
    for kind in ("torchaudio", "speechbrain"):
        plan = FbankPlan(kind)
        to_ln = 1.0 if kind == "torchaudio" else np.log(10.0) / 10.0
        for seed in range(32):
            for name, w in _fbank_error_inputs(seed, n).items():
                got = fbank_device(torch.from_numpy(w[None, :]).to(dev), plan, mean_norm=False).cpu().numpy().astype(np.float64)
                ref, unit, live = fbank_ref.log_mel_error_unit(w[None, :], kind)
                err = np.abs(got - ref) * live * to_ln
                k = float((np.maximum(err - FBANK_R0, 0.0) / np.maximum(unit * to_ln, 1e-300)).max())
                assert k <= FBANK_K_BAR, (kind, name, seed, k, float(err.max()))
                worst[(kind, name)] = max(worst.get((kind, name), (0.0, 0.0)), (k, float(err.max())))
    kmax = max(k for k, _ in worst.values())
    with capsys.disabled():
        print(f"\nfbank error model, n = {n}: achieved k <= {kmax:.2f} against the bar K = {FBANK_K_BAR} (r0 = {FBANK_R0}); per class (k, max |err| ln): "
              + "; ".join(f"{kind[:5]} {name}: {k:.2f}, {e:.1e}" for (kind, name), (k, e) in sorted(worst.items())))
    assert kmax <= FBANK_K_BAR / 2.0


def test_fbank_achieved_error_is_recorded(dev, capsys):
    """The flat figure at the bench shape (white noise, 24 segments of 2 s), next to its tolerance, so that drift is visible in the log.  What
    bounds it is `test_fbank_error_stays_inside_the_f32_dft_model`: the largest errors sit in the bins ~50 dB below their frame's level."""
    from oracle import fbank_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import fbank_device
    from speech_diarization_amd.features import FbankPlan
    wav = synth.synthetic_segments(41, 24, 32000)
    out = {}
    for kind in ("torchaudio", "speechbrain"):
        got = fbank_device(torch.from_numpy(wav).to(dev), FbankPlan(kind), mean_norm=True).cpu().numpy().astype(np.float64)
        ref = fbank_ref.fbank_batch_ref(wav) if kind == "torchaudio" else fbank_ref.speechbrain_fbank_ref(wav)
        out[kind] = float(np.abs(got - ref).max())
    with capsys.disabled():
        print(f"\nfbank max abs error vs float64: torchaudio (ln) {out['torchaudio']:.3e}, speechbrain (dB) {out['speechbrain']:.3e}")
    assert out["torchaudio"] < 2e-4 and out["speechbrain"] < 1e-3


@pytest.mark.parametrize("sr,n,B", [(8000, 16000, 5), (22050, 44100, 3), (22050, 22000, 2), (44100, 30000, 2), (48000, 96000, 3), (48000, 700, 1)])
def test_fbank_batch_at_other_sample_rates(dev, sr, n, B):
    """`fbank_batch(wavs, sr)` for sr != 16 kHz [REF speech_encode.py:14-24]: win_length = n_fft = int(sr * 0.025), hop = int(sr * 0.010),
    f_max = sr / 2 - 100 (200 / 80, 551 / 220 -- an odd n_fft --, 1102 / 441, 1200 / 480).  The generic path (sd_fbank_generic.hip: the DFT in
    float64, the mel product on the exact-f32 conv operator) against the float64 oracle, with and without mean removal, 40 and 80 bins."""
    from oracle import fbank_ref
    from speech_diarization_amd import speech_encode, synth
    wav = synth.synthetic_segments(11, B, n, std=0.2)
    for n_mels, mean_nor in ((80, True), (40, False)):
        got = speech_encode.fbank_batch(wav, sr=sr, n_mels=n_mels, mean_nor=mean_nor)
        ref = fbank_ref.fbank_batch_ref(wav, sr=sr, n_mels=n_mels, mean_nor=mean_nor)
        assert got.shape == ref.shape and got.dtype == np.float32
        assert np.abs(got - ref).max() < 5e-5, (sr, n_mels, np.abs(got - ref).max())


def test_generic_fbank_plan_modes_windows_and_nan(dev):
    """The generic framing under the other front end's switches (zero padding, dB law, top_db floor) -- n_mels = 96 sends the 16 kHz framing
    down the generic path too, where it must agree with the split-f16 kernels --, windows read in place from one signal, several workspace
    chunks, and a NaN sample (NaN features for its utterance, neighbours bitwise untouched)."""
    import ctypes as C
    from oracle import fbank_ref
    from speech_diarization_amd import _native as N, synth
    from speech_diarization_amd.engine import fbank_device, fbank_windows_device
    from speech_diarization_amd.features import FbankPlan, FRONT_ENDS, periodic_window
    lib = N.load()

    class Plan(FbankPlan):                         # a plan from explicit tables: (n_fft, hop) and the front end's switches
        def __init__(self, n_fft, hop, mel, kind):
            fe = FRONT_ENDS[kind]
            self.kind, self.n_mels, self.sr, self.n_fft, self.hop = kind, mel.shape[1], 16000, n_fft, hop
            self._lib = lib
            win = np.ascontiguousarray(periodic_window(fe.window, n_fft))
            mel = np.ascontiguousarray(mel, dtype=np.float32)
            self._h = lib.sd_fbank_plan_create(win.ctypes.data_as(C.c_void_p), n_fft, hop, mel.ctypes.data_as(C.c_void_p), mel.shape[1],
                                               fe.pad_mode, fe.log_mode, C.c_float(fe.log_eps), C.c_float(fe.top_db))
            assert self._h, N.last_error()

    wav = synth.synthetic_segments(4, 6, 16000, std=0.2)
    x = torch.from_numpy(wav).to(dev)
    # speechbrain's switches at 16 kHz with 96 bins (> 80: generic path) against the oracle's formulation
    mel96 = fbank_ref.speechbrain_filterbank(201, 96, 16000)
    got = fbank_device(x, Plan(400, 160, mel96, "speechbrain"), mean_norm=True).cpu().numpy()
    power = fbank_ref._power_spectrogram_f64(wav, 400, 160, fbank_ref._window("hamming", 400), "constant")
    db = 10.0 * np.log10(np.clip(power @ mel96, 1e-10, None))
    db = np.maximum(db, db.max(axis=(1, 2), keepdims=True) - 80.0)
    assert np.abs(got - (db - db.mean(axis=1, keepdims=True))).max() < 2e-4
    # the same tables as the 80-bin product kernels: generic (as 96 bins, the extra 16 all-zero filters) == split-f16 kernels to their tolerance
    mel80 = np.concatenate([fbank_ref.melscale_fbanks_htk(201, 20.0, 7900.0, 80, 16000), np.zeros((201, 16))], axis=1)
    g96 = fbank_device(x, Plan(400, 160, mel80, "torchaudio"), mean_norm=False).cpu().numpy()
    fast = fbank_device(x, FbankPlan("torchaudio"), mean_norm=False).cpu().numpy()
    ref80 = fbank_ref.fbank_batch_ref(wav, mean_nor=False)
    print(f"generic fbank (float64 DFT) vs float64: {np.abs(g96[:, :, :80] - ref80).max():.3e}; split-f16 kernels vs float64: {np.abs(fast - ref80).max():.3e}")
    assert np.abs(g96[:, :, :80] - ref80).max() < 5e-5 and np.abs(fast - ref80).max() < 2e-4
    assert np.allclose(g96[:, :, 80:], np.log(1e-6), atol=1e-6)
    # windows of one signal, read in place (zeros where a window hangs over an end): bitwise the gathered rows
    plan8 = FbankPlan("torchaudio", sr=8000)
    sig = torch.from_numpy(synth.synthetic_segments(9, 1, 40000, std=0.2)[0]).to(dev)
    starts = torch.tensor([0, 777, 39000, 20000, -300], dtype=torch.int64)
    rows = torch.zeros((5, 4000), device=dev)
    for i, s in enumerate(starts.tolist()):
        lo, hi = max(s, 0), min(s + 4000, 40000)
        rows[i, lo - s: hi - s] = sig[lo:hi]
    assert torch.equal(fbank_windows_device(sig, starts, 4000, plan8), fbank_device(rows, plan8))
    # more utterances than one workspace chunk holds (48 kHz, 2 s: ~2 MB per utterance, chunks of ~128): chunk boundaries change nothing
    plan48 = FbankPlan("torchaudio", sr=48000)
    big = torch.from_numpy(synth.synthetic_segments(12, 150, 48000, std=0.2)).to(dev)
    assert torch.equal(fbank_device(big, plan48)[140:], fbank_device(big[140:], plan48))
    # NaN sample
    bad = wav.copy()
    bad[2, 5000] = np.nan
    for mean_norm in (False, True):
        g = fbank_device(torch.from_numpy(bad).to(dev), plan8, mean_norm=mean_norm).cpu().numpy()
        c = fbank_device(x, plan8, mean_norm=mean_norm).cpu().numpy()
        assert np.isnan(g[2]).any() and (not mean_norm or np.isnan(g[2]).all())
        assert np.array_equal(np.delete(g, 2, axis=0), np.delete(c, 2, axis=0))
