"""The "f32-split16x3" mode: f32 VALUES carried as two f16 halves, three f16 MFMA products per value pair, f32
accumulation — f32-level accuracy on the f16 matrix cores for the wide layers (stem, tdnn1, tdnn2, MFA).

It must pass the exact-f32 path's bars: embeddings within 1e-5 cosine / 1e-3 * max absolute of the float64 oracle at the
full geometry, identical cluster assignments; per operator the error against float64 is compared with what the exact-f32
kernel achieves on the same inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["wide256", "auto"], autouse=True)
def split_kernel_choice(request):
    """"wide256": the C-wide layers of the f32-split16x3 schedule always on the 256x256 kernel (at these test sizes "auto" sends them
    to the 128x128 split kernel: launches of at most 128 big tiles)."""
    from speech_diarization_amd import _native as N
    lib = N.load()
    N.check(lib.sd_set_tuning(N.SD_TUNE_F16_NARROW_TILES, 0 if request.param == "wide256" else -1), "sd_set_tuning")
    yield request.param
    N.check(lib.sd_set_tuning(N.SD_TUNE_F16_NARROW_TILES, -1), "sd_set_tuning")


def _cos_dist(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def _ref_conv_cl(x, w, b, T, dil):
    M, cin = x.shape
    xt = x.view(M // T, T, cin).transpose(1, 2)
    pad = dil * (w.shape[2] - 1) // 2
    if pad:
        xt = F.pad(xt, (pad, pad), mode="reflect")
    return F.conv1d(xt, w, b, dilation=dil).transpose(1, 2).reshape(M, -1)


def test_split16_pack_layout_and_accuracy(dev):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(37, 200, generator=g) * torch.logspace(-6, 3, 200)[None, :]      # magnitudes 1e-6 .. 1e3
    x[3, 10] = 1e6                                                                      # beyond the f16 range: clamped
    x[5, 11] = 0.0
    p = ops.split16_pack(x.to(dev), a_col0=8, cin=150).cpu()                         # 150 -> 160 value columns
    assert p.shape == (37, 320) and p.dtype == torch.float16
    v = p.view(37, 5, 2, 32)
    hi, lo = v[:, :, 0, :].reshape(37, 160).double(), v[:, :, 1, :].reshape(37, 160).double()
    want = x[:, 8:158].double().clamp(-65504.0, 65504.0)
    assert torch.equal(hi[:, :150], want.float().half().double())                     # hi = f16(v)
    rec = hi + lo
    assert (rec[:, 150:] == 0).all()
    # v = hi + lo to 2^-22 relative, or 2^-25 absolute where lo is an f16 subnormal
    assert ((rec[:, :150] - want).abs() <= np.maximum(2.0 ** -22 * want.abs(), 2.0 ** -25)).all()


@pytest.mark.parametrize("B,T,cin,cout,k,dil", [
    (3, 201, 1024, 1024, 1, 1),      # tdnn1 / tdnn2
    (2, 201, 80, 1024, 5, 1),        # stem: k = 5 reflect gather, cin padded 80 -> 96
    (2, 150, 3072, 3072, 1, 1),      # MFA, tiles spanning two segments
    (5, 101, 256, 1280, 1, 1),       # 1 s windows, cout not a multiple of the tile
    (1, 40, 96, 256, 3, 2),          # one partial tile, dilated
])
def test_conv1d_cl_split16_is_as_accurate_as_exact_f32(dev, B, T, cin, cout, k, dil):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(B * 131 + T + cout)
    x = torch.randn(B * T, cin, generator=g) * 3.0
    x[:, ::7] *= 1e-3                                       # small and large channels side by side
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    ref = torch.relu(_ref_conv_cl(x.double(), w.double(), b.double(), T, dil)) * scale.double() + shift.double()
    ws, s = ops.pack_weight_split16(w, dev)
    got = ops.conv1d_cl_split16(x.to(dev), ws, s, T, cin=cin, dil=dil, bias=b.to(dev), act="relu", scale=scale.to(dev), shift=shift.to(dev))
    f32 = ops.conv1d_cl(x.to(dev), ops.pack_weight(w, dev), T, cin=cin, dil=dil, bias=b.to(dev), act="relu", scale=scale.to(dev), shift=shift.to(dev))
    torch.cuda.synchronize()
    assert got.dtype == torch.float32 and got.shape == ref.shape
    e_split = (got.cpu().double() - ref).abs().max().item()
    e_f32 = (f32.cpu().double() - ref).abs().max().item()
    top = ref.abs().max().item()
    print(f"\n[{B}x{T} {cin}->{cout} k{k}] max abs error vs float64: split16x3 {e_split:.3e}, exact f32 {e_f32:.3e} (output max {top:.2f})")
    assert e_split < 2e-6 * max(1.0, top)                  # f32-level: the exact-f32 kernel's own bar
    assert e_split < 8.0 * e_f32 + 1e-7 * top


def test_conv1d_cl_split16_tee_and_colstat(dev):
    """The two epilogues the wide layers use: tdnn1's store-only tee (LDS-staged epilogue) and tdnn2 / MFA's column
    statistics (register epilogue), against the exact-f32 kernel's outputs."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(33)
    B, T, cin, cout, chunk = 3, 201, 256, 1024, 128
    x = torch.randn(B * T, cin, generator=g)
    w = torch.randn(cout, cin, 1, generator=g) / 16
    b, scale, shift = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ws, s = ops.pack_weight_split16(w, dev)
    kw = dict(cin=cin, bias=b.to(dev), act="relu", scale=scale.to(dev), shift=shift.to(dev))
    tee = torch.zeros(B * T, chunk, device=dev)
    y = ops.conv1d_cl_split16(x.to(dev), ws, s, T, tee=tee, tee_lo=chunk, tee_hi=2 * chunk, **kw)
    y32 = ops.conv1d_cl(x.to(dev), ops.pack_weight(w, dev), T, **kw)
    assert (y - y32).abs().max() < 2e-6 * y32.abs().max() and torch.equal(tee, y[:, chunk:2 * chunk])
    cs = torch.zeros(ops.colstat_floats(B * T, cout), device=dev)
    y2 = ops.conv1d_cl_split16(x.to(dev), ws, s, T, colstat=cs, **kw)
    assert torch.equal(y2, y)
    st = ops.colstat_finish(cs, y2, B, T, pivot=shift.to(dev), want_std=True)
    yd = y2.double().view(B, T, cout)
    mean, std = yd.mean(1), yd.var(1, unbiased=False).clamp_min(1e-12).sqrt()
    assert (st[:, :cout].double() - mean).abs().max() < 1e-5 and (st[:, cout:].double() - std).abs().max() < 1e-5


@pytest.mark.parametrize("B,T,cin,cout,k,dil", [
    (3, 201, 128, 128, 3, 2),        # Res2Net conv (block 1)
    (5, 101, 128, 128, 3, 4),        # 1 s windows, dilation 4: reflect gathers across tile edges
    (2, 201, 3072, 128, 1, 1),       # attention TDNN
    (1, 40, 96, 200, 3, 1),          # partial tiles both ways, cin padded 96 -> 96, cout not a multiple of 128
    (2, 77, 36, 64, 5, 1),           # cin % 32 != 0 (zero-filled weight padding), k = 5
])
def test_conv1d_cl_split16_narrow_is_as_accurate_as_exact_f32(dev, B, T, cin, cout, k, dil):
    """The 128x128 kernel that splits its f32 activations while staging them (Res2Net convs, attention TDNN of the f32-split16x3
    mode), against float64 and next to the exact-f32 kernel on the same inputs."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(B * 17 + T + cin)
    x = torch.randn(B * T, cin, generator=g) * 2.0
    x[:, ::5] *= 1e-3
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    ref = torch.relu(_ref_conv_cl(x.double(), w.double(), b.double(), T, dil)) * scale.double() + shift.double()
    ws, s = ops.pack_weight_split16(w, dev)
    kw = dict(cin=cin, dil=dil, bias=b.to(dev), act="relu", scale=scale.to(dev), shift=shift.to(dev))
    got = ops.conv1d_cl_split16(x.to(dev), ws, s, T, narrow=True, **kw)
    f32 = ops.conv1d_cl(x.to(dev), ops.pack_weight(w, dev), T, **kw)
    torch.cuda.synchronize()
    e_split = (got.cpu().double() - ref).abs().max().item()
    e_f32 = (f32.cpu().double() - ref).abs().max().item()
    top = ref.abs().max().item()
    print(f"\n[narrow {B}x{T} {cin}->{cout} k{k} d{dil}] max abs error vs float64: split16x3 {e_split:.3e}, exact f32 {e_f32:.3e} (output max {top:.2f})")
    assert e_split < 2e-6 * max(1.0, top) and e_split < 8.0 * e_f32 + 1e-7 * top


def test_conv1d_cl_split16_narrow_epilogues(dev):
    """What the Res2Net chain and the attention TDNN ask of the narrow kernel: an input slice of a wider buffer, the output into a
    slice, tee + tee_add (the c_{j+1} + y_j add), per-segment bias and the tanh after the affine — against the exact-f32 kernel."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, C, hid = 3, 57, 256, 64
    xbig = torch.randn(B * T, C, generator=g)
    w = torch.randn(hid, hid, 3, generator=g) / 14
    x_d = xbig.to(dev)
    ws, s = ops.pack_weight_split16(w, dev)
    outs = []
    for split in (True, False):
        out = torch.zeros(B * T, C, device=dev)
        tee = torch.zeros(B * T, hid, device=dev)
        kw = dict(cin=hid, dil=2, act="relu", a_col0=64, out=out, o_col0=64, tee=tee, tee_lo=0, tee_hi=hid, tee_add=x_d, ta_col0=128)
        if split:
            ops.conv1d_cl_split16(x_d, ws, s, T, narrow=True, **kw)
        else:
            ops.conv1d_cl(x_d, ops.pack_weight(w, dev), T, **kw)
        outs.append((out, tee))
    (o1, t1), (o2, t2) = outs
    assert (o1 - o2).abs().max() < 2e-6 * o2.abs().max() and (t1 - t2).abs().max() < 2e-6 * t2.abs().max()
    assert o1[:, :64].abs().max() == 0 and o1[:, 128:].abs().max() == 0
    # attention TDNN form: per-segment bias, relu -> affine -> tanh
    cin, cout = 512, 128
    x = torch.randn(B * T, cin, generator=g).to(dev)
    w2 = torch.randn(cout, cin, 1, generator=g) / np.sqrt(cin)
    gb = torch.randn(B, cout, generator=g).to(dev)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    ws2, s2 = ops.pack_weight_split16(w2, dev)
    kw = dict(cin=cin, bias=gb, bias_per_seg=True, act="relu", scale=scale, shift=shift, act2="tanh")
    ys = ops.conv1d_cl_split16(x, ws2, s2, T, narrow=True, **kw)
    y32 = ops.conv1d_cl(x, ops.pack_weight(w2, dev), T, **kw)
    # both against float64 (two f32-level results differ by the sum of their errors, which depends on the exact kernel's summation order)
    ref = torch.tanh(torch.relu(x.cpu().double() @ w2[:, :, 0].double().T + gb.cpu().double().repeat_interleave(T, 0)) * scale.cpu().double() + shift.cpu().double())
    assert (ys.cpu().double() - ref).abs().max() < 2e-6 and (y32.cpu().double() - ref).abs().max() < 2e-6
    assert (ys - y32).abs().max() < 4e-6


def test_narrow_split16_conv_writes_split_output(dev):
    """The narrow kernel with y as SD_DT_SPLIT16 rows (what tdnn2 of the f32-split16x3 schedule reads): bit for bit the pack of its f32
    result, written into a channel slice of a wider row; the tee (+ tee_add) copy stays f32; rows past the slice untouched."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, hid, C = 3, 201, 128, 384
    x = torch.randn(B * T, hid, generator=g).to(dev)
    nxt = torch.randn(B * T, C, generator=g).to(dev)
    w = torch.randn(hid, hid, 3, generator=g) / 14
    bias, scale, shift = torch.randn(hid, generator=g).to(dev), (torch.rand(hid, generator=g) + 0.5).to(dev), torch.randn(hid, generator=g).to(dev)
    ws, s = ops.pack_weight_split16(w, dev)
    kw = dict(cin=hid, dil=3, bias=bias, act="relu", scale=scale, shift=shift, o_col0=128, tee_lo=0, tee_hi=hid, tee_add=nxt, ta_col0=256)
    y32 = torch.zeros(B * T, C, device=dev)
    tee32 = torch.zeros(B * T, hid, device=dev)
    ops.conv1d_cl_split16(x, ws, s, T, narrow=True, out=y32, tee=tee32, **kw)
    ysp = torch.full((B * T, 2 * C), 7.0, device=dev, dtype=torch.float16)
    tee2 = torch.zeros(B * T, hid, device=dev)
    dummy = torch.zeros(B * T, C, device=dev)
    ops.conv1d_cl_split16(x, ws, s, T, narrow=True, out=dummy, out_split=ysp, tee=tee2, **kw)
    want = ops.split16_pack(y32, 128, hid)                   # [M, 2 * 128]: the four groups of the slice
    assert torch.equal(ysp[:, 256:512], want)
    assert bool((ysp[:, :256] == 7.0).all()) and bool((ysp[:, 512:] == 7.0).all()) and bool((dummy == 0).all())
    assert torch.equal(tee2, tee32)


@pytest.mark.parametrize("cin,cout,taps,B,T", [(80, 1024, 5, 3, 201), (256, 512, 1, 2, 150)])
def test_wide_split16_conv_writes_split_output(dev, cin, cout, taps, B, T):
    """The 256x256 split kernel with y as SD_DT_SPLIT16 rows (the stem of the f32-split16x3 schedule): bit for bit the pack of its f32
    result (which comes from the register epilogue, the split form from the LDS-staged one), incl. a tile that hangs over row M."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B * T, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, taps, generator=g) / np.sqrt(cin * taps)
    bias, scale, shift = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    ws, s = ops.pack_weight_split16(w, dev)
    kw = dict(cin=cin, bias=bias, act="relu", scale=scale, shift=shift)
    y32 = ops.conv1d_cl_split16(x, ws, s, T, **kw)
    ysp = torch.zeros((B * T, 2 * cout), device=dev, dtype=torch.float16)
    ops.conv1d_cl_split16(x, ws, s, T, out=torch.zeros_like(y32), out_split=ysp, **kw)
    assert torch.equal(ysp, ops.split16_pack(y32, 0, cout))


@pytest.mark.parametrize("B,n", [(4, 32000), (5, 16000), (2, 100000), (3, 9600)])
def test_ecapa_split16_full_geometry_matches_oracle(dev, B, n):
    """The exact-f32 path's full-geometry test (tests/test_gpu_fbank_ecapa.py) with the SAME bars, on the split16x3 engine;
    also reports how far it sits from the exact-f32 engine."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    wav = synth.synthetic_segments(0, B, n)
    got = EmbeddingEngine(sd, dev, precision="f32s").embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    f32 = EmbeddingEngine(sd, dev, precision="f32").embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    ref = pipeline_ref.encode_batch_ref(sd, wav, torch.float64)
    cd, cd32 = _cos_dist(got, ref), _cos_dist(f32, ref)
    print(f"\nsplit16x3 vs float64: cos-dist {cd.max():.2e}, abs {np.abs(got - ref).max() / np.abs(ref).max():.2e} of max; "
          f"exact f32 vs float64: {cd32.max():.2e}, {np.abs(f32 - ref).max() / np.abs(ref).max():.2e}")
    assert cd.max() < 1e-5, cd
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()
    assert not np.array_equal(got, f32)                      # it IS another kernel


@pytest.mark.parametrize("B,n", [(4, 32000), (3, 9600)])
def test_ecapa_narrow_split_mode_matches_oracle(dev, B, n):
    """precision "f32ns": the wide layers on the exact-f32 kernel, only the narrow convs and the attention logits as split16x3 products
    (VERDICT r2 item 4's alternative).  The exact-f32 path's bars; another set of kernels than either "f32" or "f32s"; reachable
    through the drop-in switch."""
    from oracle import pipeline_ref
    from speech_diarization_amd import speech_encode, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    wav = synth.synthetic_segments(0, B, n)
    wd = torch.from_numpy(wav).to(dev)
    got = EmbeddingEngine(sd, dev, precision="f32ns").embed(wd).cpu().numpy()
    f32 = EmbeddingEngine(sd, dev, precision="f32").embed(wd).cpu().numpy()
    f32s = EmbeddingEngine(sd, dev, precision="f32s").embed(wd).cpu().numpy()
    ref = pipeline_ref.encode_batch_ref(sd, wav, torch.float64)
    cd = _cos_dist(got, ref)
    print(f"\nf32 wide + split16x3 narrow vs float64: cos-dist {cd.max():.2e}, abs {np.abs(got - ref).max() / np.abs(ref).max():.2e} of max")
    assert cd.max() < 1e-5, cd
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()
    assert not np.array_equal(got, f32) and not np.array_equal(got, f32s)
    assert "f32ns" in speech_encode._PRECISIONS
    with pytest.raises(ValueError):
        EmbeddingEngine(sd, dev, precision="f32x")


def test_ecapa_split16_exchange_path_matches_oracle(dev):
    """44 segments = 8844 rows: just past the small-launch routing (<= 8192 rows send the C-wide layers to the 128x128 kernels), so the
    256x256 split kernels run and the layers exchange SD_DT_SPLIT16 tensors instead of f32 ones (the stem's output, the Res2Net output
    r, the block outputs: no f32 copy, no pack pass).  Same bars against the float64 oracle as the exact-f32 path; a batch of 4
    (under the default routing: everything on the narrow kernels, f32 tensors) must agree with it to f32 rounding."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    wav = synth.synthetic_segments(3, 44, 32000)
    eng = EmbeddingEngine(sd, dev, precision="f32s")
    big = eng.embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
    small = eng.embed(torch.from_numpy(wav[:4]).to(dev)).cpu().numpy()
    ref = pipeline_ref.encode_batch_ref(sd, wav[:6], torch.float64)
    cd = _cos_dist(big[:6], ref)
    assert cd.max() < 1e-5 and np.abs(big[:6] - ref).max() < 1e-3 * np.abs(ref).max()
    assert _cos_dist(big[:4], small).max() < 1e-9          # (another schedule under the default routing, the same one when the fixture forces the big tiles)


def test_split16_identical_clusters_and_properties_at_full_size(dev):
    """north_star's 'identical cluster assignments' for the split16x3 engine (vs the exact-f32 engine and the CPU oracle,
    C = 1024), and the configs[1] batch (5000 segments through the 256x256 kernel at full occupancy): finite, run-to-run
    bitwise, within 1e-6 cosine of the exact-f32 engine."""
    from oracle import pipeline_ref
    from speech_diarization_amd import cluster, ops, synth
    from speech_diarization_amd.diarization_baseline import gather_windows, speech_windows
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    conv = synth.synthetic_conversation(30.0, 3, seed=2)
    starts, _, _ = speech_windows([(s, e) for s, e, _ in conv.turns], len(conv.wav), 16000, 2.0, 0.5)
    wav = gather_windows(conv.wav, starts, 32000)
    eng = EmbeddingEngine(sd, dev, max_batch=2500, precision="f32s")
    embs = [eng.embed(torch.from_numpy(wav).to(dev)).cpu().numpy(),
            EmbeddingEngine(sd, dev, precision="f32").embed(torch.from_numpy(wav).to(dev)).cpu().numpy(),
            pipeline_ref.encode_batch_ref(sd, wav, torch.float32)]
    labels = []
    for emb in embs:
        K = ops.cosine_affinity(torch.from_numpy(cluster.center(emb).astype(np.float32)).to(dev)).cpu().numpy()
        labels.append((cluster.relabel_by_first_appearance(cluster.spectral(K, 3)), cluster.relabel_by_first_appearance(cluster.ahc_cosine(K, 0.3))))
    for lab in labels[1:]:
        assert np.array_equal(lab[0], labels[0][0]) and np.array_equal(lab[1], labels[0][1])
    assert _cos_dist(embs[0], embs[2]).max() < 1e-5
    g = torch.Generator(device=dev).manual_seed(11)
    big = (torch.randn((5000, 32000), generator=g, device=dev) * 0.1).clamp_(-1.0, 1.0)
    a = eng.embed(big)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, eng.embed(big))
    del eng
    torch.cuda.empty_cache()
    b = EmbeddingEngine(sd, dev, max_batch=2500, precision="f32").embed(big)
    cd = 1.0 - torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)
    print(f"\n5000 segments: split16x3 vs exact f32 max cosine distance {float(cd.max()):.2e}")
    assert float(cd.max()) < 1e-6


@pytest.mark.parametrize("B,T,Cc", [(3, 201, 512), (2, 101, 256), (4, 33, 256), (2, 256, 256), (5, 1, 256), (1, 129, 3072), (2, 208, 1024), (2, 17, 256)])
def test_fused_attention_pooling_split16_matches_f64(dev, B, T, Cc):
    """asp.conv + softmax over T + weighted mean / std with the logits product on split operands (three f16 MFMA products per value
    pair): the exact-f32 kernel's bars against float64, and next to the exact-f32 kernel on the same inputs."""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    att = 128
    a1 = torch.tanh(torch.randn(B * T, att, generator=g))
    wc = torch.randn(Cc, att, 1, generator=g) / 4
    h = torch.randn(B * T, Cc, generator=g) * 1.5 + 0.3
    wp = ops.pack_weight(wc, dev)
    got = ops.asp_attend_pool(a1.to(dev), wp, h.to(dev), B, T, split16=True).cpu().double()
    f32 = ops.asp_attend_pool(a1.to(dev), wp, h.to(dev), B, T).cpu().double()
    a = torch.softmax((a1.double() @ wc[:, :, 0].double().T).view(B, T, Cc), dim=1)
    hr = h.double().view(B, T, Cc)
    mu = (a * hr).sum(1)
    sd = torch.sqrt(((a * (hr - mu[:, None]) ** 2).sum(1)).clamp_min(1e-12))
    assert (got[:, :Cc] - mu).abs().max() < 2e-5
    if T > 1:
        assert (got[:, Cc:] - sd).abs().max() < 2e-5
    assert (got[:, :Cc] - mu).abs().max() < 4.0 * (f32[:, :Cc] - mu).abs().max() + 1e-6


@pytest.mark.parametrize("width,att,B,n", [(64, 32, 3, 16000), (128, 32, 2, 9600), (256, 32, 4, 32000), (256, 128, 41, 32000), (512, 128, 3, 32000)])
def test_ecapa_split16_small_geometries_match_oracle(dev, width, att, B, n):
    """with C <= 256 the wide-ROLE layers (stem, tdnn1, tdnn2) carry the NARROW split packing (weights scaled by 2^s, no
    folded bias / BatchNorm scale); the schedule must run them on the 128x128 split kernel with `w_scale_inv` (or exact f32), never on
    the folded-scale form.  256 < C < 1024 carries no split packing on the wide layers at all.  41 segments = 8241 rows: past the
    small-launch routing, column statistics on.  The exact-f32 bars against the float64 oracle, for "f32s" and "f32ns"."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    cfg = synth.EcapaConfig(channels=(width, width, width, width, 3 * width), attention_channels=att, lin_neurons=192, res2net_scale=8, se_channels=32)
    sd = synth.make_ecapa_state_dict(1234, cfg)
    wav = synth.synthetic_segments(5, B, n)
    wd = torch.from_numpy(wav).to(dev)
    ref = pipeline_ref.encode_batch_ref(sd, wav[:6], torch.float64)
    for precision in ("f32s", "f32ns", "f32"):
        got = EmbeddingEngine(sd, dev, precision=precision).embed(wd).cpu().numpy()[:6]
        cd = _cos_dist(got, ref)
        assert cd.max() < 1e-5, (precision, cd)
        assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max(), precision


def test_split_sites_keep_nan(dev):
    """every f32 -> (hi, lo) split site clamps to the f16 range AND keeps NaN (`sd_split16_clamp`): the pack kernel, the
    narrow kernel's staging, its SD_DT_SPLIT16 output and the wide kernel's.  A NaN activation reaches the matrix cores as a NaN (it used
    to become -65504, a finite value) and leaves the other rows' bits alone; +-inf and out-of-range values clamp to +-65504.  The affected
    rows come out NaN, in the f32 result and in its split twin.  (Round 5: the epilogue's lower clamp keeps a NaN too,
    `sd_max_keep_nan`; until then `fmaxf(acc, -inf)` turned the matrix cores' NaN into -inf here and a ReLU turned it into 0 -- the
    "-inf from the f16 MFMA" this test used to describe was that clamp, not the instruction.)"""
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, cin, cout = 2, 150, 128, 128
    x = torch.randn(B * T, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, 3, generator=g) / np.sqrt(3 * cin)
    ws, s = ops.pack_weight_split16(w, dev)
    clean = ops.conv1d_cl_split16(x, ws, s, T, cin=cin, dil=2, act=None, narrow=True)
    xn = x.clone()
    xn[200, 5] = float("nan")                          # row 200 = frame 50 of segment 1; taps reach frames 48, 50, 52
    hit = [T + 48, T + 50, T + 52]
    y = ops.conv1d_cl_split16(xn, ws, s, T, cin=cin, dil=2, act=None, narrow=True)
    assert not bool(torch.isfinite(y[hit]).any())
    keep = torch.ones(B * T, dtype=torch.bool, device=dev)
    keep[hit] = False
    assert torch.equal(y[keep], clean[keep])
    ysp = torch.zeros((B * T, 2 * cout), device=dev, dtype=torch.float16)
    ops.conv1d_cl_split16(xn, ws, s, T, cin=cin, dil=2, act=None, narrow=True, out=torch.zeros_like(y), out_split=ysp)
    assert torch.equal(ysp.view(torch.int16), ops.split16_pack(y, 0, cout).view(torch.int16))       # NaN rows included, bit for bit
    assert bool(torch.isnan(y[hit]).any(dim=1).all()) and bool(torch.isnan(ysp[hit].float()).any(dim=1).all())     # NaN, not a clamped -inf
    # the pack pass: NaN stays NaN, +-inf / out-of-range clamp
    v = torch.tensor([[float("nan"), float("inf"), -float("inf"), 1e6, -1e6, 1.5, 0.0, -2.25] * 4], device=dev)
    p = ops.split16_pack(v).float()
    hi, lo = p[0, :8], p[0, 32:40]
    assert bool(torch.isnan(hi[0])) and bool(torch.isnan(lo[0]))
    assert hi[1:].tolist() == [65504.0, -65504.0, 65504.0, -65504.0, 1.5, 0.0, -2.25] and lo[1:].abs().max() == 0
    # wide kernel (256x256), split output from the LDS-staged epilogue
    cinw, coutw = 256, 512
    xw = torch.randn(B * T, cinw, generator=g).to(dev)
    xw[77, 3] = float("nan")
    ww = torch.randn(coutw, cinw, 1, generator=g) / np.sqrt(cinw)
    wws, sw = ops.pack_weight_split16(ww, dev)
    bias, scale, shift = torch.randn(coutw, generator=g).to(dev), (torch.rand(coutw, generator=g) + 0.5).to(dev), torch.randn(coutw, generator=g).to(dev)
    kw = dict(cin=cinw, bias=bias, act=None, scale=scale, shift=shift)
    y32 = ops.conv1d_cl_split16(xw, wws, sw, T, **kw)
    assert not bool(torch.isfinite(y32[77]).any()) and int((~torch.isfinite(y32)).any(dim=1).sum()) == 1
    yws = torch.zeros((B * T, 2 * coutw), device=dev, dtype=torch.float16)
    ops.conv1d_cl_split16(xw, wws, sw, T, out=torch.zeros_like(y32), out_split=yws, **kw)
    assert torch.equal(yws.view(torch.int16), ops.split16_pack(y32, 0, coutw).view(torch.int16))


def test_split16_attention_logits_with_large_weights(dev):
    """the fused pooling kernel's SPLIT logits product scaled its f32 weights by a fixed 2^8 before the f16 cast: |w| >= 256
    overflowed to inf and the logits to NaN.  The forward now passes the layer's own 2^s (max |w| 2^s in [512, 1024), from the host,
    as for every other split weight); the stand-alone entry keeps 2^8 and clamps.  An ECAPA whose attention conv has |w| up to ~600:
    f32s and f32ns against the float64 oracle at the exact-f32 bars."""
    from oracle import pipeline_ref
    from speech_diarization_amd import ops, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    cfg = synth.EcapaConfig(channels=(256, 256, 256, 256, 768), attention_channels=128, lin_neurons=192, res2net_scale=8, se_channels=32)
    sd = dict(synth.make_ecapa_state_dict(1234, cfg))
    w = np.array(sd["asp.conv.conv.weight"], dtype=np.float32)
    # large, but few: one strong tap per output channel keeps the softmax from collapsing onto a single frame everywhere
    big = np.zeros_like(w)
    big[np.arange(w.shape[0]), np.arange(w.shape[0]) % w.shape[1], 0] = 600.0 * np.sign(w[np.arange(w.shape[0]), np.arange(w.shape[0]) % w.shape[1], 0])
    sd["asp.conv.conv.weight"] = (w + big).astype(np.float32)
    wav = synth.synthetic_segments(7, 3, 32000)
    ref = pipeline_ref.encode_batch_ref(sd, wav, torch.float64)
    for precision in ("f32", "f32s", "f32ns"):
        got = EmbeddingEngine(sd, dev, precision=precision).embed(torch.from_numpy(wav).to(dev)).cpu().numpy()
        assert np.isfinite(got).all(), precision
        assert _cos_dist(got, ref).max() < 1e-5, (precision, _cos_dist(got, ref))
    # the stand-alone entry (fixed 2^8): finite for any weight, exact while |w| 2^8 stays inside the f16 range
    g = torch.Generator().manual_seed(3)
    B, T, Cc = 2, 101, 256
    a1 = torch.tanh(torch.randn(B * T, 128, generator=g)).to(dev)
    h = (torch.randn(B * T, Cc, generator=g) * 1.5).to(dev)
    wc = torch.randn(Cc, 128, 1, generator=g) * 100.0
    out = ops.asp_attend_pool(a1, ops.pack_weight(wc, dev), h, B, T, split16=True)
    assert bool(torch.isfinite(out).all())
