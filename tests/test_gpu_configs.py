"""BASELINE.json configs[2], [3], [4] as parity-test cases (GPU).

configs[2]: long multi-speaker meeting, VAD segments -> windows sharded round-robin -> gathered -> clustered;
configs[3]: streaming 16-channel feed, hipGraph-captured fbank+ECAPA per hop, online agglomeration;
configs[4]: batch job — f16 ECAPA (covered by tests/test_gpu_f16.py) and the 50k x 50k tiled cosine affinity,
            checked at full size through size-independent properties.
"""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_sd(width=128):
    from speech_diarization_amd import synth
    return synth.make_ecapa_state_dict(1234, synth.EcapaConfig.small(width))


def test_config1_full_size_properties(dev):
    """BASELINE configs[1] at full size (10 000 synthetic 2 s segments, full spkrec-ecapa geometry, micro-batch
    5000): properties that need no oracle at this size, plus the float64 oracle on a handful of rows.
    Rows are independent: a permuted batch gives the permuted embeddings and a small batch (other kernels) the
    same ones, both to f32 rounding; the same batch twice is bitwise equal; the f16 path stays inside the 1e-3
    cosine bar."""
    from oracle import pipeline_ref
    from speech_diarization_amd import ops, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    n = 10000
    g = torch.Generator(device=dev).manual_seed(11)
    wav = (torch.randn((n, 32000), generator=g, device=dev) * 0.1).clamp_(-1.0, 1.0)
    wav[77] = 0.0                                            # a silent segment must not poison its neighbours
    eng = EmbeddingEngine(sd, dev, max_batch=5000)
    emb = eng.embed(wav)
    assert emb.shape == (n, 192) and bool(torch.isfinite(emb).all())
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(3)).to(dev)
    emb_p = eng.embed(wav[perm])
    # not bitwise: the SE / global statistics are summed per 128-row tile, so a segment's partial sums are
    # grouped by where it sits in the batch (f32 rounding of a 201-term mean)
    cosp = 1.0 - torch.nn.functional.cosine_similarity(emb_p.double(), emb[perm].double(), dim=1)
    assert float(cosp.max()) < 1e-10 and float((emb_p - emb[perm]).abs().max()) < 1e-5 * float(emb.abs().max())
    assert torch.equal(eng.embed(wav), emb)                  # run-to-run deterministic
    idx = torch.tensor([0, 77, 4999, 5000, 9999], device=dev)
    small = eng.embed(wav[idx])                              # 5 segments: the small-launch kernels
    cosd = 1.0 - torch.nn.functional.cosine_similarity(small.double(), emb[idx].double(), dim=1)
    assert float(cosd.max()) < 1e-9
    ref = pipeline_ref.encode_batch_ref(sd, wav[idx].cpu().numpy(), torch.float64)
    e = emb[idx].cpu().numpy().astype(np.float64)
    cd = 1.0 - (e * ref).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref, axis=1))
    assert cd.max() < 1e-5, cd
    K = ops.cosine_affinity(emb)
    assert (torch.diagonal(K) - 1.0).abs().max() < 1e-5 and float(K.abs().max()) <= 1.0 + 1e-5
    assert (K - K.T).abs().max() < 2e-7
    del eng, K
    torch.cuda.empty_cache()
    e16 = EmbeddingEngine(sd, dev, max_batch=5000, precision="f16").embed(wav)
    cos16 = 1.0 - torch.nn.functional.cosine_similarity(e16.double(), emb.double(), dim=1)
    assert float(cos16.max()) < 1e-3
    print(f"config 1 full size: f16 vs f32 max cosine distance {float(cos16.max()):.2e}, vs float64 oracle {cd.max():.2e}")


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_reference_batch_sizes_agree_with_a_small_batch_and_the_oracle(dev, precision):
    """The reference's own call sizes (16-segment hop, batches of 32 [REF anti_stick_diarize.py:134] and 128 [REF anti_stick_diarize.py:398]) and
    the sizes between them send the layers to different kernels (32x64 / 64x64 ring tiles, 80 / 96 / 112-row and 128x64 tiles, grid split-K for
    the per-segment layers, 256 / 512 pooling channels per workgroup): a segment's embedding must not depend on the batch it sits in beyond f32
    rounding, the same call twice is bitwise equal, and six of the rows are held against the float64 oracle."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    wav_np = synth.synthetic_segments(3, 300, 32000)
    wav = torch.from_numpy(wav_np).to(dev)
    eng = EmbeddingEngine(sd, dev, max_batch=512, precision=precision)
    pick = [0, 1, 7, 15, 150, 299]
    small = eng.embed(wav[pick])
    tol = 1e-9 if precision == "f32" else 2e-4
    from speech_diarization_amd import _native
    for B in (16, 24, 32, 48, 64, 100, 128, 224, 256, 257, 300):
        emb = eng.embed(wav[:B])
        _native.profile_enable(True)
        again = eng.embed(wav[:B])
        _, splitk_layers, _ = _native.profile_read(_native.SD_PROF_SEG_SPLITK)
        _native.profile_enable(False)
        assert torch.equal(emb, again), B
        # the five per-segment layers with K >= 512 (3 x SE squeeze FC, global-context bias, final FC) take the grid split-K pair up to
        # 256 rows: the forward's scratch is sized for the largest of them (4 MB were 0.7 MB short for the final FC at 225..256)
        assert splitk_layers == (5 if B <= 256 else 0), (B, splitk_layers)
        rows = [i for i, r in enumerate(pick) if r < B]
        cos = 1.0 - torch.nn.functional.cosine_similarity(emb[[pick[i] for i in rows]].double(), small[rows].double(), dim=1)
        assert float(cos.max()) < tol, (B, float(cos.max()))
    ref = pipeline_ref.encode_batch_ref(sd, wav_np[pick], torch.float64)
    e = small.cpu().numpy().astype(np.float64)
    cd = 1.0 - (e * ref).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref, axis=1))
    assert cd.max() < (1e-5 if precision == "f32" else 1e-3), cd


@pytest.mark.parametrize("precision", ["f32", "f32s"])
def test_config1_at_the_bench_launch_shape(dev, precision):
    """`bench.py`'s step since round 3: ONE forward of 10 000 segments (max_batch = micro-batch = 10 000; ~105 GB of
    workspace at f32, ~150 GB with the split copies of f32-split16x3), on the bench's own input stream.  Five rows
    against the float64 oracle (< 1e-5 cosine distance), the same launch twice bitwise equal, and the rows the launch
    shares with two launches of 5 000 equal to f32 rounding."""
    from oracle import pipeline_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = synth.make_ecapa_state_dict(1234)
    n = 10000
    wav = synth.synthetic_segments_device(0, n, 32000, dev, std=0.1)
    eng = EmbeddingEngine(sd, dev, max_batch=n, precision=precision)
    emb = eng.embed(wav)
    assert emb.shape == (n, 192) and bool(torch.isfinite(emb).all())
    assert torch.equal(eng.embed(wav), emb)                  # run-to-run bitwise at this launch size
    idx = torch.tensor([0, 4999, 5000, 7777, 9999], device=dev)
    ref = pipeline_ref.encode_batch_ref(sd, wav[idx].cpu().numpy(), torch.float64)
    e = emb[idx].cpu().numpy().astype(np.float64)
    cd = 1.0 - (e * ref).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref, axis=1))
    assert cd.max() < 1e-5, cd
    del eng
    torch.cuda.empty_cache()
    half = EmbeddingEngine(sd, dev, max_batch=5000, precision=precision)
    two = torch.cat([half.embed(wav[:5000]), half.embed(wav[5000:])])
    cos2 = 1.0 - torch.nn.functional.cosine_similarity(two.double(), emb.double(), dim=1)
    assert float(cos2.max()) < 1e-9
    print(f"bench launch shape, {precision}: vs float64 oracle {cd.max():.2e}, vs 2 x 5000 {float(cos2.max()):.2e}")


def test_config2_sharded_meeting_matches_unsharded_and_cpu(dev):
    """10 min, 8 voices: windows are embedded (a) in one piece, (b) as 8 round-robin shards that are
    gathered and de-interleaved (the W = 8 layout, executed rank by rank on this one GPU).  Rows are
    independent in every kernel; the batch size only selects between kernels that sum K in a different order
    (128x128 tiles vs the 32x32 split-K kernel of small launches), so (b) equals (a) to f32 rounding, the
    same batch twice is BITWISE equal, and cluster labels equal the CPU path's."""
    from oracle.ecapa_ref import EcapaRef
    from oracle.pipeline_ref import encode_batch_ref
    from speech_diarization_amd import cluster, dist as sdist, ops, synth, vad
    from speech_diarization_amd.diarization_baseline import gather_windows, speech_windows
    from speech_diarization_amd.engine import EmbeddingEngine
    sd = _small_sd()
    conv = synth.synthetic_conversation(600.0, 8, seed=0)
    scorer = vad.SileroVAD(model=vad.EnergyScorer())
    mask = vad.morph_open_close(vad.hysteresis_binarize(scorer.probs(conv.wav), 0.6, 0.4), 10.0)
    speech = vad.mask_to_segments(mask, 10.0, 350.0, 100.0, 40.0)
    starts, _, _ = speech_windows(speech, len(conv.wav), 16000, 2.0, 0.25)
    wav = torch.from_numpy(gather_windows(conv.wav, starts, 32000)).to(dev)
    n = wav.shape[0]
    assert n > 1000
    eng = EmbeddingEngine(sd, dev, max_batch=256)
    whole = eng.embed(wav)
    W = 8
    rows = sdist.shard_rows(n, W)
    gathered = torch.zeros((W, rows, 192), device=dev)
    for r in range(W):
        idx = torch.from_numpy(sdist.shard_indices(n, r, W)).to(dev)
        gathered[r, : idx.numel()] = eng.embed(wav[idx])
    merged = sdist.deinterleave(gathered, n, W)
    cosd = 1.0 - torch.nn.functional.cosine_similarity(merged.double(), whole.double(), dim=1)
    assert float(cosd.max()) < 1e-9 and float((merged - whole).abs().max()) < 1e-4 * float(whole.abs().max())
    idx0 = torch.from_numpy(sdist.shard_indices(n, 0, W)).to(dev)
    assert torch.equal(eng.embed(wav[idx0]), gathered[0, : idx0.numel()])       # run-to-run deterministic
    # clustering on the gathered embeddings vs the CPU path (subsampled: every 4th window keeps the CPU leg short)
    sub = np.arange(0, n, 4)
    e_gpu = whole[sub].cpu().numpy()
    net = EcapaRef(sd, torch.float32)
    e_cpu = encode_batch_ref(sd, wav[sub].cpu().numpy(), torch.float32, net)
    labs = []
    for e in (e_gpu, e_cpu):
        K = ops.cosine_affinity(torch.from_numpy(cluster.center(e).astype(np.float32)).to(dev)).cpu().numpy()
        labs.append(cluster.relabel_by_first_appearance(cluster.spectral(K, 8)))
    assert np.array_equal(labs[0], labs[1])
    assert len(set(labs[0].tolist())) == 8


def test_config3_streaming_graph_replay_matches_eager(dev):
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    from speech_diarization_amd.streaming import OnlineClusterer, StreamingEmbedder
    eng = EmbeddingEngine(_small_sd(), dev, max_batch=16)
    st = StreamingEmbedder(eng, channels=16, window_s=2.0, hop_s=0.25)
    eager = StreamingEmbedder(EmbeddingEngine(_small_sd(), dev, max_batch=16), channels=16, use_graph=False)
    feed = synth.synthetic_segments(5, 16, 4000 * 12)
    lat = []
    for h in range(12):
        chunk = torch.from_numpy(feed[:, h * 4000:(h + 1) * 4000]).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = st.push(chunk).clone()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        b = eager.push(chunk)
        assert a.shape == (16, 192)
        assert torch.equal(a, b)                        # same kernels, same order: bitwise equal
        want = np.zeros((16, 32000), np.float32)        # the ring holds the last 2 s of every channel, oldest first
        have = feed[:, max(0, (h + 1) * 4000 - 32000):(h + 1) * 4000]
        want[:, 32000 - have.shape[1]:] = have
        assert np.array_equal(st.ring.cpu().numpy(), want)
    print(f"streaming hop latency (small geometry): p50 {np.median(lat) * 1e3:.3f} ms")
    with pytest.raises(ValueError):
        st.push(torch.zeros(16, 100))
    oc = OnlineClusterer(threshold=0.5)
    x = np.eye(192)[:3]
    labels = oc.assign(np.concatenate([x, x + 0.01, x[:1]]))
    assert labels.tolist() == [0, 1, 2, 0, 1, 2, 0]


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("f32s", 1e-5), ("f16", 1e-3)])
def test_config3_streaming_full_geometry_matches_the_oracle(dev, precision, tol):
    """BASELINE configs[3] at the real geometry (C = 1024 spkrec-ecapa state dict, 16 channels x 2 s windows, 250 ms hop,
    hipGraph-captured fbank + ECAPA per hop), 11 hops: the graph-replayed embeddings are (a) bitwise the eager embeddings of
    the same ring, (b) bitwise `engine.embed` of the ring contents, and (c) the float64 oracle's embeddings of those
    contents (`oracle.pipeline_ref.encode_batch_ref`, the call being replaced is [REF speech_encode.py:73-78]): hop 2 (ring
    still mostly zeros) on 4 channels, hop 10 (ring wrapped) on all 16."""
    from oracle.pipeline_ref import encode_batch_ref
    from speech_diarization_amd import synth
    from speech_diarization_amd.engine import EmbeddingEngine
    from speech_diarization_amd.streaming import StreamingEmbedder
    sd = synth.make_ecapa_state_dict(1234)
    st = StreamingEmbedder(EmbeddingEngine(sd, dev, max_batch=16, precision=precision), channels=16, window_s=2.0, hop_s=0.25)
    eager = StreamingEmbedder(EmbeddingEngine(sd, dev, max_batch=16, precision=precision), channels=16, use_graph=False)
    plain = EmbeddingEngine(sd, dev, max_batch=16, precision=precision)
    feed = synth.synthetic_segments(51, 16, 4000 * 11)
    worst = 0.0
    for h in range(11):
        chunk = torch.from_numpy(feed[:, h * 4000:(h + 1) * 4000]).to(dev)
        a = st.push(chunk).clone()
        assert torch.equal(a, eager.push(chunk))
        ring = st.ring
        assert torch.equal(a, plain.embed(ring))
        if h in (2, 10):
            rows = slice(0, 4) if h == 2 else slice(0, 16)
            ref = encode_batch_ref(sd, ring[rows].cpu().numpy(), torch.float64)
            e = a[rows].cpu().numpy().astype(np.float64)
            cd = 1.0 - (e * ref).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(ref, axis=1))
            worst = max(worst, float(cd.max()))
            assert cd.max() < tol, (precision, h, cd)
    print(f"configs[3] full geometry, {precision}: max cosine distance to the float64 oracle {worst:.2e}")


def test_config2_one_hour_meeting_through_the_product_entry(tmp_path):
    """BASELINE configs[2] through `diarize_audio(world="dist")` [REF diarization_baseline.py:236-266] at full geometry on the
    GPU: a synthetic 1 h, 8-voice meeting; the sharded entry under RCCL (process group "nccl" in a world of one — the box
    has one card — with the all-gather forced) writes the RTTM of the unsharded call byte for byte, the embeddings never
    leave the device before the collective, and cluster labels computed from the GPU embeddings equal those computed from
    the CPU oracle's embeddings on a subsample of the windows.  Runs in a fresh process (it owns a process group)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import os, sys, time, warnings
sys.path.insert(0, {root!r})
warnings.simplefilter("ignore")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", SD_DIST_FORCE_COLLECTIVE="1")
import numpy as np, torch, torch.distributed as tdist
from oracle.ecapa_ref import EcapaRef
from oracle.pipeline_ref import encode_batch_ref
from speech_diarization_amd import audio_io, cluster, diarization_baseline as db, ops, speech_encode, synth
t0 = time.time()
conv = synth.synthetic_conversation(3600.0, 8, seed=0)
wav = {str(tmp_path)!r} + "/meeting.wav"
audio_io.write_wav16(wav, conv.wav, conv.sr)
print(f"meeting generated in {{time.time() - t0:.1f}} s", flush=True)
torch.cuda.set_device(0)
t0 = time.time()
seg1, det1 = db.diarize_audio(wav, 0.35, 0.1, 2, 8, rttm_filepath=wav[:-4] + ".single.rttm", return_details=True)
print(f"world=None: {{det1['embeddings'].shape[0]}} windows, {{len(seg1)}} turns, {{time.time() - t0:.1f}} s", flush=True)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
calls = []
orig = tdist.all_gather_into_tensor
def spy(out, inp, *a, **k):
    calls.append((tuple(inp.shape), inp.is_cuda, out.is_cuda))
    return orig(out, inp, *a, **k)
tdist.all_gather_into_tensor = spy
t0 = time.time()
seg2, det2 = db.diarize_audio(wav, 0.35, 0.1, 2, 8, rttm_filepath=wav[:-4] + ".dist.rttm", return_details=True, world="dist")
print(f"world=dist: {{time.time() - t0:.1f}} s, collective calls {{calls}}", flush=True)
n = det1["embeddings"].shape[0]
assert n > 5000 and calls == [((n, 192), True, True)]                 # ONE all-gather, of device tensors
assert np.array_equal(det1["embeddings"], det2["embeddings"]) and np.array_equal(det1["labels"], det2["labels"])
assert open(wav[:-4] + ".single.rttm").read() == open(wav[:-4] + ".dist.rttm").read() and seg1 == seg2
assert len(set(det1["labels"].tolist())) >= 2
# GPU embeddings vs the CPU oracle on every 16th window (full geometry on the host is ~60 segments/s): same clusters
sub = np.arange(0, n, 16)
y = audio_io.read_audio(wav, sr=16000, mono=True)[0]
rows = db.gather_windows(y, det1["window_starts"][sub], 32000)
sd = synth.make_ecapa_state_dict(speech_encode.SYNTHETIC_SEED)
e_cpu = encode_batch_ref(sd, rows, torch.float32, EcapaRef(sd, torch.float32))
e_gpu = det1["embeddings"][sub]
cd = 1.0 - (e_gpu.astype(np.float64) * e_cpu).sum(1) / (np.linalg.norm(e_gpu, axis=1) * np.linalg.norm(e_cpu, axis=1))
assert cd.max() < 1e-3, cd.max()
labs = []
for e in (e_gpu, e_cpu):
    K = ops.cosine_affinity(torch.from_numpy(cluster.center(e).astype(np.float32)).cuda()).cpu().numpy()
    labs.append(cluster.relabel_by_first_appearance(cluster.spectral(K, 8)))
assert np.array_equal(labs[0], labs[1])
print(f"subsample of {{len(sub)}} windows: max cosine distance GPU vs CPU oracle {{cd.max():.2e}}, identical cluster labels", flush=True)
tdist.barrier(); tdist.destroy_process_group()
print("ok")
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    print(res.stdout[-1500:])
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), (res.stdout[-800:], res.stderr[-2500:])


@pytest.mark.parametrize("split16", [False, True])
def test_config4_affinity_50k_properties(dev, split16):
    """50 000 x 50 000 f32 affinity (10 GB) in one call, exact f32 and the split16x3 kernel of configs[4] (`sd_affinity.hip`: EXACTLY
    symmetric): unit diagonal, symmetry, bounded, planted duplicates / zero rows, and agreement of a row block with the row-block
    entry point."""
    from speech_diarization_amd import ops
    n, d = 50000, 192
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((n, d), generator=g, device=dev)
    x[123] = 0.0
    x[40000] = 3.0 * x[17]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = ops.cosine_affinity(x, split16=split16)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"50k x 50k affinity{' (split16x3)' if split16 else ''}: {dt * 1e3:.1f} ms, {4.0 * n * n / dt / 1e12:.2f} TB/s written, {384.0 * n * n / dt / 1e12:.1f} TFLOP/s")
    diag = torch.diagonal(K)
    keep = torch.ones(n, dtype=torch.bool, device=dev); keep[123] = False
    assert (diag[keep] - 1.0).abs().max() < 1e-5 and diag[123] == 0
    assert bool(torch.all(K[123] == 0)) and bool(torch.all(K[:, 123] == 0))
    assert abs(K[40000, 17].item() - 1.0) < 1e-5
    assert K.abs().max() <= 1.0 + 1e-5
    for lo in (0, 20000, 49000):                           # symmetry, checked block-wise to bound memory
        blk = K[lo:lo + 1000, :]
        assert torch.equal(blk[:, lo:lo + 1000], blk[:, lo:lo + 1000].T)
        assert torch.equal(blk, K[:, lo:lo + 1000].T)          # one accumulator, two stores (sd_affinity.hip, both forms)
    rows = ops.cosine_affinity(x, rows=(31000, 31500), split16=split16)
    assert (rows - K[31000:31500]).abs().max() < 1e-6      # the row-block entry computes every tile on the conv kernels: same products, another summation order
    sub = x[:2000].cpu().numpy()
    from sklearn.metrics.pairwise import cosine_similarity
    assert np.abs(K[:2000, :2000].cpu().numpy() - cosine_similarity(sub)).max() < 2e-6


def test_rccl_all_gather_executes_on_the_device(tmp_path):
    """The build box has one GPU, so RCCL cannot be given two ranks here (it refuses two ranks per device); what CAN run is
    the real thing in a world of one: `init_from_env("nccl")` with `device_id`, then the same `all_gather_into_tensor`
    call the N-GPU job makes, on device tensors, through the sharded product entry (bench.py's step) — in a fresh process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import os, sys
sys.path.insert(0, {root!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", SD_DIST_FORCE_COLLECTIVE="1")
import torch, torch.distributed as tdist
from speech_diarization_amd import dist as sd
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert tdist.get_backend() == "nccl"
x = torch.arange(7 * 192, dtype=torch.float32, device="cuda").view(7, 192)
full = sd.all_gather_embeddings(x, 7)            # the collective runs (forced) and the de-interleave is the identity
torch.cuda.synchronize()
assert full.is_cuda and torch.equal(full, x)
t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
tdist.all_reduce(t, op=tdist.ReduceOp.MAX)      # bench.py's max-over-ranks timing
assert float(t.item()) == 3.5
tdist.barrier(); tdist.destroy_process_group()
print("ok")
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), (res.stdout[-500:], res.stderr[-2000:])
