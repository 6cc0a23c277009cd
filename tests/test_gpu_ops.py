"""GPU parity of the individual HIP operators against the CPU oracle / plain torch f64.

All calls go through the C ABI (ctypes -> libsd_hip.so).  Tolerances are stated per test:
the operators are exact-f32 (fma chains on the f32 matrix cores), the references are
float64, so the bound is f32 rounding of a K-term sum.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv_cl(x, w, b, T, dil):
    """x [M, cin] f64 channel-last, w [cout, cin, k]; 'same' reflect conv per segment."""
    M, cin = x.shape
    B = M // T
    xt = x.view(B, T, cin).transpose(1, 2)
    pad = dil * (w.shape[2] - 1) // 2
    if pad:
        xt = F.pad(xt, (pad, pad), mode="reflect")
    y = F.conv1d(xt, w, b, dilation=dil)
    return y.transpose(1, 2).reshape(M, -1)


@pytest.fixture(params=["auto", "split32", "tiles128", "tiles64", "rows80", "rows96", "rows112", "wide256"])
def conv_kernel(request):
    """Small launches pick the 32x32 split-K kernel by themselves; "tiles128" pins the 128x128 kernel so that
    both implementations of the operator see every case."""
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


@pytest.mark.parametrize("B,T,cin,cout,k,dil", [
    (3, 201, 1024, 1024, 1, 1),     # pointwise, tile-aligned N
    (5, 101, 128, 128, 3, 3),       # Res2Net conv, M not a tile multiple, segments straddle tiles
    (2, 201, 80, 256, 5, 1),        # stem: cin not a multiple of the K step
    (7, 1, 1024, 128, 1, 1),        # SE squeeze: T = 1, tiny M
    (4, 33, 256, 192, 1, 1),        # cout not a multiple of the N tile
    (2, 150, 96, 1280, 3, 2),       # wide output, 5 column tiles of 256, dilated taps, M = 300 (second row tile almost empty)
    (1, 300, 64, 1100, 1, 1),       # wide output that cannot take 16-byte stores (cout % 8 != 0): scalar epilogue
    (3, 201, 80, 1024, 5, 1),       # the stem at full width: cin not a multiple of the K step, five taps
])
def test_conv1d_cl_matches_torch(dev, conv_kernel, B, T, cin, cout, k, dil):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B * T, cin, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, k, generator=g, dtype=torch.float64) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    scale = torch.rand(cout, generator=g, dtype=torch.float64) + 0.5
    shift = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = torch.relu(_ref_conv_cl(x, w, b, T, dil)) * scale + shift
    got = ops.conv1d_cl(x.float().to(dev), ops.pack_weight(w.float(), dev), T, cin=cin, dil=dil, bias=b.float().to(dev),
                        act="relu", scale=scale.float().to(dev), shift=shift.float().to(dev))
    torch.cuda.synchronize()
    err = (got.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_conv1d_cl_slices_tee_and_per_segment_bias(dev, conv_kernel):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, C, hid = 3, 57, 256, 32
    xbig = torch.randn(B * T, C, generator=g, dtype=torch.float64)
    w = torch.randn(hid, hid, 3, generator=g, dtype=torch.float64) / 10
    segb = torch.randn(B, hid, generator=g, dtype=torch.float64)
    x_d = xbig.float().to(dev)
    out = torch.zeros(B * T, C, device=dev)
    tee = torch.zeros(B * T, hid, device=dev)
    # input = column slice [64, 96), output into column slice [32, 64), tee = y + xbig[:, 96:128]
    ops.conv1d_cl(x_d, ops.pack_weight(w.float(), dev), T, cin=hid, dil=2, bias=segb.float().to(dev), bias_per_seg=True,
                  act="relu", act2="tanh", a_col0=64, out=out, o_col0=32, tee=tee, tee_lo=0, tee_hi=hid, tee_add=x_d, ta_col0=96)
    torch.cuda.synchronize()
    y = _ref_conv_cl(xbig[:, 64:96].contiguous(), w, None, T, 2) + segb.repeat_interleave(T, dim=0)
    y = torch.tanh(torch.relu(y))
    assert (out[:, 32:64].cpu().double() - y).abs().max() < 1e-5
    assert out[:, :32].abs().max() == 0 and out[:, 64:].abs().max() == 0
    assert (tee.cpu().double() - (y + xbig[:, 96:128])).abs().max() < 1e-5


@pytest.mark.parametrize("M,cin,cout", [(32, 6144, 128),      # global-context bias at the reference's batch of 32: 24 splits of 256
                                        (16, 6144, 192),      # final FC, 6 column tiles
                                        (128, 1024, 128),     # SE squeeze FC: 8 splits of 128, 4 row tiles
                                        (7, 1000, 36),        # cin not a multiple of the K step, a ragged last split, cout % 32 != 0
                                        (200, 2080, 64),      # a last split of 32 values
                                        (300, 1024, 128),     # more than 256 rows: the plain operator
                                        (5, 256, 64)])        # K too short to split: the plain operator
def test_seg_gemm_matches_torch(dev, M, cin, cout):
    """The per-segment layers with K split over the grid (sd_seg_gemm_f32) against float64, bitwise repeatable, and with no scratch
    exactly the plain operator."""
    from speech_diarization_amd import ops, _native as N
    g = torch.Generator().manual_seed(M + cin)
    x = torch.randn(M, cin, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 1, generator=g, dtype=torch.float64) / np.sqrt(cin)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    scale = torch.rand(cout, generator=g, dtype=torch.float64) + 0.5
    shift = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = torch.sigmoid(torch.relu(x @ w[:, :, 0].T + b) * scale + shift)
    xd, wp = x.float().to(dev), ops.pack_weight(w.float(), dev)
    kw = dict(cin=cin, bias=b.float().to(dev), act="relu", scale=scale.float().to(dev), shift=shift.float().to(dev), act2="sigmoid")
    got = ops.seg_gemm(xd, wp, **kw)
    again = ops.seg_gemm(xd, wp, **kw)
    plain = ops.conv1d_cl(xd, wp, 1, **kw)
    torch.cuda.synchronize()
    assert (got.cpu().double() - ref).abs().max() < 2e-6
    assert torch.equal(got, again)
    assert (got - plain).abs().max() < 2e-6
    need = N.load().sd_seg_gemm_scratch_bytes(M, wp.shape[2], cout)
    assert (need > 0) == (M <= 256 and wp.shape[2] >= 512)
    if need:      # a scratch that is too small falls back to the plain operator, bit for bit
        small = torch.empty(need // 4 - 1, device=dev)
        assert torch.equal(ops.seg_gemm(xd, wp, scratch=small, **kw), plain)


def test_conv1d_cl_rejects_bad_arguments(dev):
    from speech_diarization_amd import ops, _native
    x = torch.zeros(10, 6, device=dev)
    w = torch.zeros(4, 1, 32, device=dev)
    with pytest.raises(_native.SdError, match="multiple of 4"):
        ops.conv1d_cl(x, w, 5, cin=6)
    x = torch.zeros(8, 8, device=dev)
    w3 = torch.zeros(4, 3, 32, device=dev)
    with pytest.raises(_native.SdError, match="reflect padding"):
        ops.conv1d_cl(x, w3, 2, cin=8, dil=2)   # pad 2 >= T 2


def test_segment_reductions_and_se(dev):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(11)
    B, T, C = 5, 201, 1024
    x = torch.randn(B * T, C, generator=g, dtype=torch.float64) * 3 + 1
    xd = x.float().to(dev)
    m = ops.seg_mean(xd, B, T)
    ms = ops.seg_mean_std(xd, B, T)
    xr = x.view(B, T, C)
    assert (m.cpu().double() - xr.mean(1)).abs().max() < 1e-5
    assert (ms[:, :C].cpu().double() - xr.mean(1)).abs().max() < 1e-5
    assert (ms[:, C:].cpu().double() - xr.std(1, unbiased=False)).abs().max() < 1e-5
    gate = torch.rand(B, C, generator=g, dtype=torch.float64)
    res = torch.randn(B * T, C, generator=g, dtype=torch.float64)
    y = ops.se_scale_residual(xd, gate.float().to(dev), res.float().to(dev), B, T)
    ref = x * gate.repeat_interleave(T, 0) + res
    assert (y.cpu().double() - ref).abs().max() < 1e-5


def test_std_clamp_on_constant_input(dev):
    """constant-in-time features -> variance 0 -> sqrt(clamp(1e-12)) = 1e-6 exactly (SURVEY 8c KAT)."""
    from speech_diarization_amd import ops
    B, T, C = 2, 50, 64
    x = torch.full((B * T, C), 0.75, device=dev)
    ms = ops.seg_mean_std(x, B, T, eps=1e-12)
    assert torch.all(ms[:, :C] == 0.75)
    assert torch.allclose(ms[:, C:], torch.full((B, C), 1e-6, device=dev), rtol=1e-6, atol=0)


@pytest.mark.parametrize("B,T,C", [(4, 201, 768),      # LDS-resident kernel (one HBM pass)
                                   (2, 400, 256),      # T too long for the LDS budget -> streaming kernel
                                   (3, 50, 100)])      # C not a multiple of 32 -> streaming kernel
def test_asp_pool(dev, B, T, C):
    from speech_diarization_amd import ops
    g = torch.Generator().manual_seed(3)
    logit = torch.randn(B * T, C, generator=g, dtype=torch.float64) * 4
    h = torch.randn(B * T, C, generator=g, dtype=torch.float64)
    out = ops.asp_pool(logit.float().to(dev), h.float().to(dev), B, T)
    a = torch.softmax(logit.view(B, T, C), dim=1)
    hv = h.view(B, T, C)
    mu = (a * hv).sum(1)
    sd = torch.sqrt((a * (hv - mu[:, None]) ** 2).sum(1).clamp(min=1e-12))
    assert (out[:, :C].cpu().double() - mu).abs().max() < 1e-5
    assert (out[:, C:].cpu().double() - sd).abs().max() < 1e-5


@pytest.mark.parametrize("n,d", [(1, 192), (37, 192), (300, 192), (129, 50)])
def test_cosine_affinity_matches_sklearn(dev, n, d):
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if n > 5:
        x[3] = 0.0          # zero row -> zero similarity row / column, diagonal 0 (sklearn semantics)
        x[5] = 2.5 * x[4]   # colinear rows -> 1
    got = ops.cosine_affinity(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = cosine_similarity(x)
    assert got.dtype == np.float32 and got.shape == (n, n)
    assert np.abs(got - ref).max() < 2e-6
    if n > 5:
        assert np.all(got[3] == 0) and np.all(got[:, 3] == 0)
        assert abs(got[4, 5] - 1.0) < 1e-6


@pytest.mark.parametrize("n,d", [(300, 192), (1500, 192), (129, 50), (2048, 192), (132, 50), (260, 32), (516, 256), (1024, 100), (8, 7)])
def test_cosine_affinity_split16_matches_sklearn(dev, n, d):
    """hi + lo split through the f16 matrix cores: same bar as the exact-f32 kernel."""
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import ops
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.01, 30.0, size=(n, 1)).astype(np.float32)
    x[7] = 0.0
    ref = cosine_similarity(x.astype(np.float64))
    xd = torch.from_numpy(x).to(dev)
    got = ops.cosine_affinity(xd, split16=True).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6
    assert np.all(got[7] == 0.0) and np.all(got[:, 7] == 0.0)
    if n % 4 == 0:                                            # the triangle + mirror kernel (any D: 1 .. 8 packed groups of 32 values)
        assert np.array_equal(got, got.T)
    lo, hi = n // 3, min(n // 3 + 37, n)
    blk = ops.cosine_affinity(xd, rows=(lo, hi), split16=True).cpu().numpy()
    assert np.abs(blk - ref[lo:hi]).max() < 2e-6


@pytest.mark.parametrize("n,pad", [(4, 0), (128, 0), (132, 4), (260, 0), (1500, 0), (1504, 0), (2052, 0), (2560, 0), (3000, 8), (3001, 0), (3004, 4),
                                   (4104, 0), (8200, 8)])
def test_cosine_affinity_split16_triangle_and_mirror(dev, n, pad):
    """The split16x3 affinity of a whole matrix (`sd_affinity.hip`): 128 x 128 tiles on and above the diagonal in 8 x 8 super-tiles, every
    off-diagonal tile stored twice from the same accumulators (as it lies and quad-transposed), diagonal tiles element by element.
    Every element must be there (partial edge tiles and partial super-tiles included), K exactly symmetric, within the exact-f32
    kernel's bar of sklearn, nothing written past a row; a size that is not a multiple of 4 takes the every-tile path (same
    products, another summation order)."""
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 192)).astype(np.float32) * rng.uniform(0.1, 10.0, size=(n, 1)).astype(np.float32)
    if n > 11:
        x[11] = 0.0
    xd = torch.from_numpy(x).to(dev)
    buf = torch.full((n, n + pad), float("nan"), device=dev)
    k = ops.cosine_affinity(xd, out=buf[:, :n], split16=True)
    assert not torch.isnan(k).any()
    if pad:
        assert torch.isnan(buf[:, n:]).all()
    if n % 4 == 0:
        assert torch.equal(k, k.T)
    ref = cosine_similarity(x.astype(np.float64))
    assert np.abs(k.cpu().numpy() - ref).max() < 2e-6
    if n > 11:
        assert bool(torch.all(k[11] == 0)) and bool(torch.all(k[:, 11] == 0))
    for lo, hi in ((0, 130), (n // 2 - 64, n // 2 + 200), (n - 131, n)):
        lo, hi = max(lo, 0), min(hi, n)
        blk = ops.cosine_affinity(xd, rows=(lo, hi), split16=True)
        assert (blk - k[lo:hi]).abs().max() < 1e-6


@pytest.mark.parametrize("n,pad", [(4, 0), (132, 4), (1500, 0), (2500, 0), (3000, 4), (3001, 0), (3001, 3), (3002, 2), (4100, 0), (8200, 0)])
def test_cosine_affinity_triangle_and_mirror(dev, n, pad):
    """The full matrix is computed on and above the diagonal and mirrored: every element must be there and K must be exactly
    symmetric.  N % 4 == 0 with aligned rows: `sd_affinity.hip`'s exact-f32 form (16x16x4 MFMA, 128 x 128 tiles in 8 x 8 super-tiles,
    both copies of a tile from the same accumulators) -- within 1e-6 of the row-block entry point, which sums in another order.
    Otherwise the conv kernel's triangle (bands of 8 tile rows, scalar tails) -- the row-block entry point's bits.  Sizes: 12 .. 65
    tiles per side (partial last bands / super-tiles), a padded row stride (16-byte stores stay aligned)."""
    from sklearn.metrics.pairwise import cosine_similarity
    from speech_diarization_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 192)).astype(np.float32) * rng.uniform(0.1, 10.0, size=(n, 1)).astype(np.float32)
    if n > 11:
        x[11] = 0.0
    xd = torch.from_numpy(x).to(dev)
    buf = torch.full((n, n + pad), float("nan"), device=dev)
    k = ops.cosine_affinity(xd, out=buf[:, :n])
    assert not torch.isnan(k).any()
    if pad:
        assert torch.isnan(buf[:, n:]).all()          # nothing written past a row
    assert torch.equal(k, k.T)
    every = ops.cosine_affinity(xd, rows=(0, n - 1))                               # every tile computed by the conv kernel
    if n % 4 == 0 and pad % 4 == 0:
        assert (every - k[: n - 1]).abs().max() < 1e-6
    else:
        assert torch.equal(every, k[: n - 1])                                      # the same kernel
    for lo, hi in ((0, 130), (n // 2 - 64, n // 2 + 200), (n - 131, n)):          # (few tiles: the split-K kernel sums in another order)
        lo, hi = max(lo, 0), min(hi, n)
        blk = ops.cosine_affinity(xd, rows=(lo, hi))
        assert (blk - k[lo:hi]).abs().max() < 1e-6
    if n <= 3002:
        ref = cosine_similarity(x.astype(np.float64))
        assert np.abs(k.cpu().numpy() - ref).max() < 2e-6
    if n > 11:
        assert torch.all(k[11] == 0) and torch.all(k[:, 11] == 0)


def test_cosine_affinity_identity_and_empty(dev):
    from speech_diarization_amd import ops
    eye = torch.eye(192, device=dev)
    k = ops.cosine_affinity(eye)
    assert torch.equal(k, eye)
    assert ops.cosine_affinity(torch.zeros(0, 192, device=dev)).shape == (0, 0)


def test_adjacent_cosine_and_argmax(dev):
    from oracle.pipeline_ref import adjacent_cosine_ref, assign_windows_ref
    from speech_diarization_amd import ops
    rng = np.random.default_rng(0)
    e = rng.standard_normal((41, 192)).astype(np.float32)
    got = ops.adjacent_cosine(torch.from_numpy(e).to(dev)).cpu().numpy()
    assert np.abs(got - adjacent_cosine_ref(e.astype(np.float64))).max() < 1e-6
    c = rng.standard_normal((5, 192)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    wn = ops.l2norm_rows(torch.from_numpy(e).to(dev), eps_add=1e-8)
    best, score = ops.sim_argmax(wn, torch.from_numpy(c).to(dev))
    ref_best, ref_sim = assign_windows_ref(e.astype(np.float64), c.astype(np.float64))
    assert np.array_equal(best.cpu().numpy(), ref_best)
    assert np.abs(score.cpu().numpy() - ref_sim.max(1)).max() < 1e-6


# ---- N3: AS-norm and Viterbi on the device [REF diar_diag.py:196-208, 231-247]

def test_topk_mean_std_matches_numpy(dev):
    from speech_diarization_amd import ops
    rng = np.random.default_rng(3)
    for rows, n, k in [(7, 64, 20), (3, 1000, 200), (5, 10000, 200), (2, 50, 200), (4, 257, 1), (1, 300, 300)]:
        x = rng.standard_normal((rows, n)).astype(np.float32)
        x[0, : n // 3] = x[0, 0]                       # many ties, some of them at the k-th value
        if rows > 1:
            x[1] = 0.25                                # a constant row: std exactly 0
        got = ops.topk_mean_std(torch.from_numpy(x).to(dev), k).cpu().numpy()
        kk = min(k, n)
        top = np.sort(x.astype(np.float64), axis=1)[:, -kk:]
        assert np.allclose(got[:, 0], top.mean(1), atol=2e-6), (rows, n, k)
        assert np.allclose(got[:, 1], top.std(1), atol=2e-6), (rows, n, k)
    xs = torch.randn(6, 300, device=dev)
    padded = torch.full((6, 512), 99.0, device=dev)    # row stride > n: the padding must not be read
    padded[:, :300] = xs
    assert torch.equal(ops.topk_mean_std(padded[:, :300], 50), ops.topk_mean_std(xs, 50))


def test_asnorm_scores_gpu_matches_reference_goldens_and_host(dev, golden_dir):
    import json, os
    from speech_diarization_amd import diar_diag as dd
    with open(os.path.join(golden_dir, "diar_diag.json")) as f:
        cases = json.load(f)
    for case in cases:
        e, cents, cohort = (np.asarray(case[k], dtype=np.float64) for k in ("embs", "centers", "cohort"))
        got = dd.asnorm_scores(e, cents, cohort, topk=20, device=dev)
        assert got.dtype == np.float32
        assert np.abs(got - np.asarray(case["asnorm"])).max() < 2e-4          # reference output (float64) vs the f32 device path
    rng = np.random.default_rng(9)
    q, c, coh = (rng.standard_normal((n, 192)).astype(np.float32) for n in (500, 6, 3000))
    host = dd.asnorm_scores(q.astype(np.float64), c.astype(np.float64), coh.astype(np.float64), topk=200)
    assert np.abs(dd.asnorm_scores(q, c, coh, topk=200, device=dev) - host).max() < 5e-4
    assert dd.asnorm_scores(q, c, coh[:50], topk=200, device=dev).shape == (500, 6)   # cohort smaller than top-k


def test_viterbi_gpu_path_equals_the_reference_path(dev, golden_dir):
    import json, os
    from speech_diarization_amd import diar_diag as dd
    with open(os.path.join(golden_dir, "diar_diag.json")) as f:
        cases = json.load(f)
    for case in cases:
        scores = np.asarray(case["scores"], dtype=np.float32)
        assert dd.viterbi_hmm(scores, alpha=0.9, device=dev).tolist() == case["viterbi"]
        assert dd.viterbi_hmm(scores, device=dev).tolist() == case["viterbi_sticky"]
    rng = np.random.default_rng(4)
    for T, K, alpha in [(1, 3, 0.9), (2, 2, 0.995), (129, 8, 0.9), (1000, 16, 0.995), (36000, 8, 0.995), (300, 1, 0.9), (257, 64, 0.9)]:
        sc = rng.standard_normal((T, K)).astype(np.float32)
        sc[::7] = np.round(sc[::7])                    # exact ties between states: first maximum must win
        got = dd.viterbi_hmm(sc, alpha, device=dev)
        assert got.dtype == np.int32 and got.tolist() == dd.viterbi_hmm(sc, alpha).tolist(), (T, K, alpha)
