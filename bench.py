#!/usr/bin/env python3
"""Headline benchmark: segment-embeddings/sec on synthetic 2 s @ 16 kHz segments.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1      (starts the next line as a child process: speech_diarization_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over this rank's batch of synthetic segments
(BASELINE.json configs[1]: 10 000 segments of 32 000 samples, already resident in HBM):
HIP fbank -> HIP ECAPA-TDNN forward (spkrec-ecapa geometry, 20.8 M random-init parameters)
-> [N > 1: RCCL all-gather of the 192-d embeddings] -> cosine affinity of this rank's row
block on-device.  Weak scaling: every rank owns `--segments` segments.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the implicit-GEMM
conv of the wide layers on the f32 matrix cores); `roofline_other_convs` the remaining conv launches;
`roofline_fbank` is the fbank kernel, priced against HBM (its algorithmic bytes).  Kernel
durations are measured live with HIP events on the launch stream (sd_profile_*).
`cpu_baseline` times the CPU oracle (torch f32) on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0          # HBM3E spec
SAMPLES = 32000                # 2 s @ 16 kHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--segments", type=int, default=10000, help="segments per rank per step")
    ap.add_argument("--micro-batch", type=int, default=10000, help="segments per fbank + ECAPA launch (round 3: 10 000 = the whole step in one forward; 5 000 before: -1 %)")
    ap.add_argument("--precision", choices=["f32", "f16", "f32s", "f32ns"], default="f32",
                    help="f32: exact f32 MFMA (configs[1], the headline). f16: f16 operands / f32 accumulate (configs[4]). "
                         "f32s: f32-split16x3 (every frame-level contraction as three f16 MFMA products per value pair, f32-level accuracy). "
                         "f32ns: the wide layers (86 %% of the flops) on exact f32 MFMA, only the narrow convs and the attention logits split")
    ap.add_argument("--no-split-extra", action="store_true", help="skip the extra f32-split16x3 measurement appended to the f32 line")
    ap.add_argument("--no-f16-extra", action="store_true", help="skip the extra f16 measurement appended to the f32 line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for each batch size of the CPU baseline sample")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(state_dict, wav_dev, budget_s, emb_gpu=None, emb_gpu_f16=None):
    """The CPU oracle (torch f32: torch.stft fbank + F.conv1d ECAPA) on the host cores, as SURVEY.md §8(d) /
    BASELINE.md §3 prescribe: batches of 32 (the reference's embed_segments batch, [REF anti_stick_diarize.py:134])
    and of 128 (its reassignment batch, [REF :398]), 20 warm-up segments, >= 200 timed segments per batch size
    (bounded by `budget_s` seconds each).  `value` is the batch-32 rate."""
    from oracle.ecapa_ref import EcapaRef
    from oracle.pipeline_ref import encode_batch_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = EcapaRef(state_dict, torch.float32)
    sample = wav_dev[:256].cpu().numpy()
    encode_batch_ref(state_dict, sample[:20], torch.float32, net)           # 20 warm-up segments (allocator, oneDNN primitives)
    rates = {}
    for batch, n_seg in ((32, 224), (128, 256)):
        done, t0 = 0, time.perf_counter()
        while done < n_seg and (time.perf_counter() - t0) < budget_s:
            encode_batch_ref(state_dict, sample[done % 256: done % 256 + batch], torch.float32, net)
            done += batch
        rates[batch] = (done, time.perf_counter() - t0)
    d32, t32 = rates[32]
    d128, t128 = rates[128]
    # BASELINE.md §3 / SURVEY §8(d): max 1 - cos(e_gpu, e_cpu) over the first 64 bench segments (north_star bar: 1e-3)
    parity = {}
    if emb_gpu is not None:
        e_cpu = encode_batch_ref(state_dict, sample[:64], torch.float32, net).astype(np.float64)
        for key, e in (("max_cosine_distance_vs_gpu", emb_gpu), ("max_cosine_distance_vs_gpu_f16", emb_gpu_f16)):
            if e is not None:
                g = e[:64].double().cpu().numpy()
                parity[key] = float((1.0 - (g * e_cpu).sum(1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(e_cpu, axis=1))).max())
        parity["parity_segments"] = 64
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        model = "unknown"
    return {"value": d32 / t32, "unit": "segments/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{d32} of the same synthetic 2 s segments at batch 32 in {t32:.1f} s ({d32 / t32:.1f}/s) and {d128} at batch 128 in "
                      f"{t128:.1f} s ({d128 / t128:.1f}/s), after 20 warm-up segments; torch-CPU f32 oracle on {model}",
            "batch32": d32 / t32, "batch128": d128 / t128, "cpu_model": model, **parity}


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def load_traffic():
    return load_profile_json("traffic.json")


def counters_match_sources():
    """The PMC counters on the line are file reads (separate rocprofv3 --pmc passes): do they come from the kernels in the tree?  Each file
    records the fingerprint of csrc/ + include/ it was measured on (tools/import_profiles.py); compared with today's sources."""
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("sd_check_profiles", os.path.join(ROOT, "tools", "check_profiles_fresh.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        st = mod.status()
        return {"all": all(v == "ok" for v in st.values()), "files": st}
    except Exception as e:      # the counters are context, never worth failing a bench run for
        return {"all": False, "error": repr(e)}


def reference_batches(state_dict, dev, precision):
    """The hot path the way the reference itself calls it (never `value`): `ecapa_encode_batch(numpy [B, 32000]) -> numpy` at its batches of 32
    [REF anti_stick_diarize.py:134] and 128 [REF anti_stick_diarize.py:398] (pageable host memory in, H2D, fbank + ECAPA, D2H, a synchronisation
    per call), and BASELINE configs[3]'s 16-channel 250 ms hop through `StreamingEmbedder` (hipGraph replay).  Medians after a warm-up."""
    import time
    from speech_diarization_amd import speech_encode, synth
    from speech_diarization_amd.engine import EmbeddingEngine
    from speech_diarization_amd.streaming import StreamingEmbedder
    out = {"precision": precision}
    speech_encode.set_precision(precision)
    for batch, warm, reps in ((32, 80, 80), (128, 30, 30)):
        wavs = synth.synthetic_segments(5, batch, 32000)
        # (host-side generation first: the card's clocks take ~100 calls to settle after an idle stretch, so nothing slow may sit between
        # a warm-up and its measurement)
        many = [synth.synthetic_segments(50 + i, batch, 32000) for i in range(32 if batch == 32 else 12)]
        for _ in range(warm):
            speech_encode.ecapa_encode_batch(wavs)
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            speech_encode.ecapa_encode_batch(wavs)
            lat.append(time.perf_counter() - t0)
        med = float(np.median(lat))
        out[f"numpy_batch{batch}_segments_per_s"] = batch / med
        out[f"numpy_batch{batch}_ms_per_call"] = med * 1e3
        # the same batches as the reference's callers issue them -- a LOOP over independent batches [REF anti_stick_diarize.py:150-171] -- through
        # `ecapa_encode_batches` (what `embed_segments` calls): two batches in flight on two streams, results bitwise those of the calls above
        for _ in range(3):
            got = speech_encode.ecapa_encode_batches(many)
        passes = []
        for _ in range(5):
            t0 = time.perf_counter()
            got = speech_encode.ecapa_encode_batches(many)
            passes.append(time.perf_counter() - t0)
        dt = float(np.median(passes))
        out[f"loop_batch{batch}_two_in_flight_segments_per_s"] = batch * len(many) / dt
        out[f"loop_batch{batch}_two_in_flight_ms_per_batch"] = dt / len(many) * 1e3
        out[f"loop_batch{batch}_bitwise_equal_to_single_calls"] = bool(all(np.array_equal(g, speech_encode.ecapa_encode_batch(w)) for g, w in zip(got[:3], many[:3])))
    eng = EmbeddingEngine(state_dict, dev, max_batch=16, precision=precision)
    st = StreamingEmbedder(eng, channels=16, window_s=2.0, hop_s=0.25, use_graph=True)
    chunk = torch.randn(16, 4000, device=dev) * 0.1
    for _ in range(20):
        st.push(chunk)
    torch.cuda.synchronize()
    lat = []
    for _ in range(100):
        t0 = time.perf_counter()
        st.push(chunk)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    out["stream_16ch_hop_p50_ms"] = float(np.percentile(lat, 50))
    out["stream_16ch_hop_p99_ms"] = float(np.percentile(lat, 99))
    out["note"] = ("the reference's own call sizes, outside the timed region and never `value`: numpy in -> numpy out per call at batch 32 / 128 "
                   "(one synchronous call at a time), the same batches as a loop with two in flight (ecapa_encode_batches, what embed_segments calls), "
                   "and configs[3]'s 16-channel hop (2 s window every 250 ms) as one graph replay")
    return out


def der_vs_cpu(state_dict, dev, precision):
    """The metric's second half (BASELINE.json: "...; DER vs CPU ref"; BASELINE.md §3: label agreement after clustering,
    DER of the GPU RTTM against the CPU RTTM): BASELINE.json configs[0]'s recording (60 s, 2 synthetic speakers, seed 0)
    through the product entry `diarization_baseline.diarize_audio` [REF diarization_baseline.py:236-266] twice - once on
    the HIP path (windows read in place, cosine affinity on the device), once with the CPU oracle encoder injected -
    at the bench's full geometry and weights.  Outside the timed region; the oracle is the checker, not the product."""
    from oracle.ecapa_ref import EcapaRef
    from oracle.pipeline_ref import encode_batch_ref
    from speech_diarization_amd import diarization_baseline as db, rttm, speech_encode, synth
    conv = synth.synthetic_conversation(60.0, 2, seed=0)
    audio = {"waveform": conv.wav, "sample_rate": conv.sr, "uri": "config0"}
    enc = speech_encode.HipEcapaEncoder(state_dict, dev, max_batch=512, precision=precision)
    keep = speech_encode.using_ecapa_encoder
    speech_encode.using_ecapa_encoder = lambda device="cuda": enc
    try:
        t0 = time.perf_counter()
        seg_g, det_g = db.diarize_audio(audio, 0.35, 0.1, 2, 6, return_details=True)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
    finally:
        speech_encode.using_ecapa_encoder = keep
    torch.set_num_threads(usable_cores())
    net = EcapaRef(state_dict, torch.float32)
    cpu = lambda wavs: encode_batch_ref(state_dict, np.asarray(wavs, np.float32), torch.float32, net).astype(np.float32)  # noqa: E731
    t0 = time.perf_counter()
    seg_c, det_c = db.diarize_audio(audio, 0.35, 0.1, 2, 6, encoder=cpu, return_details=True)
    t_cpu = time.perf_counter() - t0
    g, c = det_g["embeddings"].astype(np.float64), det_c["embeddings"].astype(np.float64)
    cosd = 1.0 - (g * c).sum(1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(c, axis=1))
    truth = [(s, e, f"T{k}") for s, e, k in conv.turns]
    return {"der_vs_cpu": float(rttm.der(seg_c, seg_g)), "labels_identical": bool(np.array_equal(det_g["labels"], det_c["labels"])),
            "rttm_turns_identical": seg_g == seg_c, "windows": int(len(det_g["labels"])), "speakers": int(len({k for _, _, k in seg_g})),
            "max_cosine_distance": float(cosd.max()), "der_vs_ground_truth": float(rttm.der(truth, seg_g)),
            "workload": "configs[0]: 60 s 2-speaker synthetic recording (seed 0), diarize_audio, 2 s windows / 0.25 s hop, spectral clustering, "
                        "full ECAPA geometry with the bench's weights; GPU = HIP path at `dtype`, CPU = torch-f32 oracle encoder injected",
            "gpu_wall_s": t_gpu, "cpu_wall_s": t_cpu}


def main():
    args = parse()
    # `python bench.py --gpus N` from a plain shell: the parent (which has not touched the GPU: importing torch does not)
    # starts the N ranks under torch.distributed.run as a child process and relays its output and exit code
    from speech_diarization_amd import launch
    if launch.needs_self_launch(args.gpus):
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    from speech_diarization_amd import _native, ops, synth
    from speech_diarization_amd import dist as sdist
    from speech_diarization_amd.engine import EmbeddingEngine

    # RCCL ("nccl") is the product path.  SD_BENCH_BACKEND=gloo exists only to rehearse the N>1
    # control flow on a one-GPU box (ranks then share the card; RCCL refuses two ranks per device).
    backend = os.environ.get("SD_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"LOCAL_RANK={local_rank} but only {n_dev} GPU(s) visible")
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    rank, local_rank, world = sdist.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    _native.load()
    # "did the collective see N ranks": a sum of ones over the process group, on the device for RCCL
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit(f"all_reduce saw {ranks_seen} ranks, WORLD_SIZE={world}")

    state_dict = synth.make_ecapa_state_dict(1234)
    engine = EmbeddingEngine(state_dict, dev, max_batch=args.micro_batch, precision=args.precision)
    S = args.segments
    # generated on the device by the counter-based generator, seed 0, N(0, 0.1^2) clipped to
    # [-1, 1]; ONE stream of S * world segments, sharded round-robin as SURVEY §8(e) / dist.shard_indices say: rank r owns
    # rows r, r + W, r + 2 W, ...  (so the de-interleaved all-gather is the stream in its original order)
    wav = synth.synthetic_segments_device(0, S, SAMPLES, dev, std=0.1, first_row=rank, row_stride=world)
    n_total = S * world
    lo, hi = sdist.row_block(n_total, rank, world)
    aff = torch.empty((hi - lo, n_total), dtype=torch.float32, device=dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    aff_ms = [0.0]          # average duration of one affinity call (l2norm + tiled product) of the last measure()
    gather = {}             # the exchange step of the last measure(): event-timed like the affinity

    def measure(eng):
        """W warm-up steps, then exactly K timed steps between barriers; max over ranks."""
        aff_events, ag_events = [], []
        last = {}

        def step(timed=False):
            emb = eng.embed(wav)
            # the collective is enqueued behind the forward on torch's current stream (RCCL's own stream waits for it and
            # the current stream waits for RCCL's): an event pair on the current stream brackets pad + all-gather + de-interleave
            if timed:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
            full = sdist.all_gather_embeddings(emb, n_total)
            if timed:
                g1.record()
                ag_events.append((g0, g1))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.cosine_affinity(full, out=aff, rows=(lo, hi))
            if timed:
                e1.record()
                aff_events.append((e0, e1))
            last["full"] = full
            return emb
        for _ in range(args.warmup):
            step()
        fence()
        _native.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            emb = step(True)
        fence()
        dt = time.perf_counter() - t0
        aff_ms[0] = sum(a.elapsed_time(b) for a, b in aff_events) / max(len(aff_events), 1)
        ag_ms = sum(a.elapsed_time(b) for a, b in ag_events) / max(len(ag_events), 1)
        # order of the gathered matrix: row k W + r is rank r's k-th segment, i.e. full[rank::W] are this rank's own embeddings, bitwise
        full = last["full"]
        order_ok = bool(torch.equal(full[rank::world], emb)) and tuple(full.shape) == (n_total, emb.shape[1])
        gather.clear()
        gather.update({"allgather_ms": ag_ms, "order_ok": order_ok, "dt_local": dt})
        if world > 1:
            stats = torch.tensor([dt, ag_ms, 1.0 if order_ok else 0.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            every = [torch.empty_like(stats) for _ in range(world)]
            dist.all_gather(every, stats)
            every = torch.stack(every).cpu()
            gather.update({"ms_per_step_by_rank": [float(v) / args.steps * 1e3 for v in every[:, 0]],
                           "allgather_ms_by_rank": [float(v) for v in every[:, 1]],
                           "order_ok": bool((every[:, 2] == 1.0).all())})
            if not gather["order_ok"]:
                raise SystemExit("all-gather order check failed: full[rank::world] != this rank's embeddings")
            gather["full"] = full
        conv = _native.profile_read(_native.SD_PROF_CONV_GEMM)
        wide = _native.profile_read(_native.SD_PROF_CONV_WIDE)
        fb = _native.profile_read(_native.SD_PROF_FBANK)
        _native.profile_enable(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if not bool(torch.isfinite(emb).all()):
            raise SystemExit("non-finite embeddings")
        return dt, conv, fb, emb, wide

    dt, (conv_ms, conv_n, conv_flops), (fb_ms, fb_n, fb_bytes), emb, (wide_ms, wide_n, wide_flops) = measure(engine)
    affinity_ms = aff_ms[0]
    exchange = dict(gather)         # of the headline measurement (the extra precisions below overwrite `gather`)
    full_main = exchange.pop("full", None)
    if world > 1 and rank == 0:
        # parity of what arrived from the OTHER ranks: rank 0 regenerates the first 16 segments of every other shard from the
        # one stream, embeds them itself and compares with the rows the collective delivered (a different batch around a
        # segment regroups its statistics sums: equal to f32 rounding, < 1e-9 cosine, not bitwise; own rows: bitwise, above)
        k = min(16, S)
        theirs = torch.cat([synth.synthetic_segments_device(0, k, SAMPLES, dev, std=0.1, first_row=r, row_stride=world) for r in range(1, world)])
        mine = engine.embed(theirs).double()
        got = torch.cat([full_main[r::world][:k] for r in range(1, world)]).double()
        cosd = 1.0 - torch.nn.functional.cosine_similarity(mine, got, dim=1)
        exchange["gathered_rows_checked"] = int(got.shape[0])
        exchange["gathered_max_cosine_distance_vs_local_recompute"] = float(cosd.max().item())
        bar = 1e-9 if args.precision in ("f32", "f32s", "f32ns") else 1e-5
        if not float(cosd.max().item()) < bar:
            raise SystemExit(f"gathered embeddings differ from rank 0's recomputation: max cosine distance {float(cosd.max().item()):.3e}")
        del theirs, mine, got
    del full_main
    extra_f16 = None
    emb16 = None
    if args.precision == "f32" and not args.no_f16_extra:
        emb32 = emb.clone()
        del engine
        torch.cuda.empty_cache()
        eng16 = EmbeddingEngine(state_dict, dev, max_batch=args.micro_batch, precision="f16")
        dt16, (n16_ms, n16_n, n16_flops), _, emb16, (c16_ms, c16_n, c16_flops) = measure(eng16)
        cosd = 1.0 - torch.nn.functional.cosine_similarity(emb16.double(), emb32.double(), dim=1)
        c16_tf = c16_flops / (c16_ms * 1e-3) / 1e12 if c16_ms > 0 else 0.0
        extra_f16 = {
            "note": "same step with f16 operands / f32 accumulation on the frame-level convs and f16 activations "
                    "(BASELINE.json configs[4]); operand mantissa = TF32's 10 bits, which the reference enables for its own "
                    "CUDA matmuls/convs [REF diarization_baseline.py:20-21]; NOT the headline value",
            "value": n_total * args.steps / dt16, "unit": "segments/s", "ms_per_step": dt16 / args.steps * 1e3, "dtype": "f16",
            "max_cosine_distance_vs_f32_path": float(cosd.max().item()),
            "roofline": {"kernel": "conv_gemm_f16_t256_kernel", "bound": "mfma", "achieved": c16_tf, "peak": F16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": c16_tf / F16_MFMA_PEAK_TFLOPS, "launches": c16_n,
                         "avg_launch_ms": c16_ms / max(c16_n, 1), "share_of_step_time": c16_ms * 1e-3 / dt16,
                         "traffic": load_profile_json("traffic_f16.json").get("conv_gemm_f16_t256_kernel"),
                         "traffic_source": "file profiles/traffic_f16.json (separate rocprofv3 --pmc passes of `bench.py --precision f16`); NOT measured in this run",
                         "other_conv_kernels": {"kernels": "res2net_chain_f16_kernel, conv_gemm_f16_kernel, skinny/f32 per-segment layers", "launches": n16_n,
                                                "achieved": n16_flops / (n16_ms * 1e-3) / 1e12 if n16_ms > 0 else 0.0, "share_of_step_time": n16_ms * 1e-3 / dt16},
                         "mfma_util_pmc": {k: (v or {}).get("mfma_util") for k, v in load_profile_json("mfma_util_f16.json").items()
                                           if k.startswith("conv_gemm_f16")},
                         "mfma_util_pmc_source": "file profiles/mfma_util_f16.json (separate rocprofv3 --pmc pass of `bench.py --precision f16`); NOT measured in this run"},
        }

    extra_split = None
    if args.precision == "f32" and not args.no_split_extra:
        if extra_f16 is None:
            emb32 = emb.clone()
            del engine
        else:
            del eng16
        torch.cuda.empty_cache()
        engs = EmbeddingEngine(state_dict, dev, max_batch=args.micro_batch, precision="f32s")
        dts, (ns_ms, ns_n, ns_flops), _, embs, (cs_ms, cs_n, cs_flops) = measure(engs)
        cosd = 1.0 - torch.nn.functional.cosine_similarity(embs.double(), emb32.double(), dim=1)
        cs_tf = cs_flops / (cs_ms * 1e-3) / 1e12 if cs_ms > 0 else 0.0
        extra_split = {
            "note": "same step with the frame-level conv layers on the f16 matrix cores at f32-level accuracy: every f32 operand value split hi + lo "
                    "(two f16), three products hi.hi + hi.lo + lo.hi per value pair, f32 accumulation (2^-22 relative per product; exact f32 MFMA: "
                    "2^-24).  Wide layers (stem, tdnn1, tdnn2, MFA: 86 % of the flops): 256x256 LDS-DMA ring kernel on SD_DT_SPLIT16 rows; narrow layers "
                    "(Res2Net convs, attention TDNN): 128x128 kernel that splits the f32 activations while staging them; the attention conv inside the fused "
                    "pooling kernel: a1 split once in LDS.  Activations stay f32 in HBM; SE, statistics, softmax / pooling and the per-segment layers are "
                    "exact f32.  Passes the exact-f32 path's parity "
                    "tests (tests/test_gpu_split16.py).  NOT the headline value",
            "value": n_total * args.steps / dts, "unit": "segments/s", "ms_per_step": dts / args.steps * 1e3, "dtype": "f32-split16x3",
            "max_cosine_distance_vs_f32_path": float(cosd.max().item()),
            "roofline": {"kernel": "conv_gemm_f16_t256_kernel<SPLIT>", "bound": "mfma",
                         "achieved": 3.0 * cs_tf, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 3.0 * cs_tf / F16_MFMA_PEAK_TFLOPS,
                         "f32_equivalent_tflops": cs_tf, "note": "achieved = f16 MFMA flops issued (3 per algorithmic flop) per second; the pack pass "
                         "(split16_pack_kernel, f32 -> hi/lo halves, 8 bytes per value) runs in front of every launch and is not in this figure",
                         "launches": cs_n, "avg_launch_ms": cs_ms / max(cs_n, 1), "share_of_step_time": cs_ms * 1e-3 / dts,
                         "traffic": load_profile_json("traffic_f32s.json").get("conv_gemm_f16_t256_kernel"),
                         "traffic_source": "file profiles/traffic_f32s.json (separate rocprofv3 --pmc passes of `bench.py --precision f32s`); NOT measured in this run",
                         "mfma_util_pmc": (load_profile_json("mfma_util_f32s.json").get("conv_gemm_f16_t256_kernel") or {}).get("mfma_util"),
                         "mfma_util_pmc_source": "file profiles/mfma_util_f32s.json (separate rocprofv3 --pmc pass); NOT measured in this run",
                         "other_conv_kernels": {"kernels": "conv_gemm_split16_n128_kernel (Res2Net convs, attention TDNN; f32-equivalent TFLOP/s), conv_gemm_f32_kernel (affinity)",
                                                "launches": ns_n, "achieved": ns_flops / (ns_ms * 1e-3) / 1e12 if ns_ms > 0 else 0.0,
                                                "share_of_step_time": ns_ms * 1e-3 / dts}},
        }

    extra_ns = None
    if args.precision == "f32" and not args.no_split_extra:
        del engs
        torch.cuda.empty_cache()
        engn = EmbeddingEngine(state_dict, dev, max_batch=args.micro_batch, precision="f32ns")
        dtn, (nn_ms, nn_n, nn_flops), _, embn, (cn_ms, cn_n, cn_flops) = measure(engn)
        cosd = 1.0 - torch.nn.functional.cosine_similarity(embn.double(), emb32.double(), dim=1)
        extra_ns = {
            "note": "the headline step with ONLY the narrow contractions (21 Res2Net convs, attention TDNN, the attention logits inside the fused pooling "
                    "kernel: 14 % of the flops) as split16x3 products; the wide layers (86 % of the flops) stay on the exact-f32 MFMA kernel.  VERDICT r2 item 4's "
                    "alternative (\"or run the chain on split-f16x3 at f32 accuracy\").  NOT the headline value",
            "value": n_total * args.steps / dtn, "unit": "segments/s", "ms_per_step": dtn / args.steps * 1e3, "dtype": "f32 (wide) + f32-split16x3 (narrow)",
            "max_cosine_distance_vs_f32_path": float(cosd.max().item()),
            "wide_kernel": {"kernel": "conv_gemm_f32_t256_kernel", "achieved": cn_flops / (cn_ms * 1e-3) / 1e12 if cn_ms > 0 else 0.0, "unit": "TFLOP/s",
                            "frac": (cn_flops / (cn_ms * 1e-3) / 1e12 if cn_ms > 0 else 0.0) / F32_MFMA_PEAK_TFLOPS, "share_of_step_time": cn_ms * 1e-3 / dtn},
            "narrow_kernels": {"kernel": "conv_gemm_split16_n128_kernel", "launches": nn_n, "f32_equivalent_tflops": nn_flops / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0,
                               "f16_tflops_issued": 3.0 * nn_flops / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0,
                               "frac_of_f16_peak": 3.0 * (nn_flops / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0) / F16_MFMA_PEAK_TFLOPS,
                               "share_of_step_time": nn_ms * 1e-3 / dtn},
        }
        del engn

    if rank == 0:
        traffic = load_traffic()
        value = n_total * args.steps / dt
        # dominant kernel: the 256x256 ring kernel of the wide layers (90 % of the step's flops); the other conv launches
        # (Res2Net convs / chain, stem at f32, attention TDNN, affinity) are reported beside it
        narrow_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        narrow_ms, narrow_n = conv_ms, conv_n
        if wide_n > 0:
            conv_ms, conv_n, conv_flops = wide_ms, wide_n, wide_flops
        conv_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        half = args.precision == "f16"
        mfma_peak = F16_MFMA_PEAK_TFLOPS if half else F32_MFMA_PEAK_TFLOPS
        other_kernel = "conv_gemm_f16_kernel" if half else "conv_gemm_f32_kernel"
        conv_kernel = ("conv_gemm_f16_t256_kernel" if half else "conv_gemm_f32_t256_kernel") if wide_n > 0 else other_kernel   # (launches too small for the wide kernel)
        fb_gbs = fb_bytes / (fb_ms * 1e-3) / 1e9 if fb_ms > 0 else 0.0
        out = {
            "metric": "segment-embeddings/sec (2 s @16 kHz)",
            "value": value,
            "unit": "segments/s",
            "n_gpus": world if backend == "nccl" else min(world, n_dev),    # a gloo rehearsal may put several ranks on one card
            "ranks": world,
            "n_ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "value_per_gpu": value / (world if backend == "nccl" else min(world, n_dev)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else "none (single process)",
            "config": {
                "workload": "configs[1]: 10k synthetic 2 s@16 kHz segments per GPU, HIP fbank + ECAPA-TDNN forward "
                            "(C=1024 spkrec-ecapa geometry, random-init seed 1234), all-gather of 192-d embeddings, "
                            "cosine affinity of the rank's row block on-device",
                "segments_per_gpu": S, "samples_per_segment": SAMPLES, "micro_batch": args.micro_batch,
                "affinity_rows_per_gpu": hi - lo, "affinity_cols": n_total,
                "parallelism": f"segments sharded round-robin (i mod {world}) over {world} rank(s) on {min(world, n_dev)} GPU(s), one all_gather_into_tensor per step",
                "gpus_visible": n_dev,
            },
            "roofline": {
                "kernel": conv_kernel, "bound": "mfma",
                "achieved": conv_tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": conv_tflops / mfma_peak,
                "traffic": traffic.get(conv_kernel),
                "traffic_source": "file profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch); NOT measured in this run",
                "mfma_util_pmc": (load_profile_json("mfma_util.json").get(conv_kernel) or {}).get("mfma_util"),
                "mfma_util_pmc_source": "file profiles/mfma_util.json (separate rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass); NOT measured in this run",
                "launches": conv_n, "avg_launch_ms": conv_ms / max(conv_n, 1),
                "flops_per_launch": conv_flops / max(conv_n, 1),
                "share_of_step_time": conv_ms * 1e-3 / dt,
            },
            "counters_match_sources": counters_match_sources(),
            "roofline_other_convs": {
                "kernel": other_kernel + (" (+ res2net_chain_f16_kernel)" if half else ""), "bound": "mfma", "achieved": narrow_tflops,
                "peak": mfma_peak, "unit": "TFLOP/s", "frac": narrow_tflops / mfma_peak, "launches": narrow_n,
                "avg_launch_ms": narrow_ms / max(narrow_n, 1), "share_of_step_time": narrow_ms * 1e-3 / dt,
                "traffic": traffic.get(other_kernel), "traffic_source": "file profiles/traffic.json; NOT measured in this run",
            },
            "roofline_fbank": {
                "kernel": "fbank_utt16_kernel (ONE launch per batch: waveform -> log-mel incl. the utterance-level top_db floor and mean removal)", "bound": "hbm",
                "achieved": fb_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fb_gbs / HBM_PEAK_GBS,
                "traffic": traffic.get("fbank_utt16_kernel"),
                "traffic_source": "file profiles/traffic.json; NOT measured in this run",
                "launches": fb_n, "avg_launch_ms": fb_ms / max(fb_n, 1),
                "bytes_per_launch": fb_bytes / max(fb_n, 1),
                "share_of_step_time": fb_ms * 1e-3 / dt,
            },
        }
        # cosine affinity of this rank's row block ("report both fractions"): algorithmic bytes
        # 4 rows N (the block written once) + 768 N (the embeddings read once), flops 384 rows N
        a_rows = hi - lo
        a_bytes = 4.0 * a_rows * n_total + 768.0 * n_total
        a_flops = 384.0 * a_rows * n_total
        if affinity_ms > 0:
            out["roofline_affinity"] = {
                "kernel": "l2norm_rows_kernel + affinity_sym_kernel<exact f32> (sd_affinity.hip: upper triangle + mirror) when the block is the full matrix, conv_gemm_f32_kernel for a row block (N > 1 ranks)",
                "rows": a_rows, "cols": n_total, "avg_call_ms": affinity_ms, "share_of_step_time": affinity_ms * 1e-3 * args.steps / dt,
                "bytes": {"bound": "hbm", "achieved": a_bytes / (affinity_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": a_bytes / (affinity_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": a_bytes},
                # the whole-matrix (symmetric) path computes the 128x128 tiles on and above the diagonal only: nt (nt + 1) / 2 of nt^2
                "flops": (lambda nt, full: {
                    "bound": "mfma", "achieved": a_flops * full / (affinity_ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": a_flops * full / (affinity_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                    "computed_flops": a_flops * full, "algorithmic_flops": a_flops,
                    "algorithmic_equivalent_tflops": a_flops / (affinity_ms * 1e-3) / 1e12,
                    "note": "achieved / frac count the flops actually executed (upper-triangle tiles when the block is the whole matrix); "
                            "algorithmic_equivalent_tflops counts the full matrix's 384 N^2"})(
                    (n_total + 127) // 128, ((((n_total + 127) // 128) + 1) / (2.0 * ((n_total + 127) // 128))) if (a_rows == n_total) else 1.0),
                "timing": "torch.cuda.Event pairs on the launch stream around every timed call",
            }
        # the path's ONE exchange step: pad + all_gather_into_tensor of W x ceil(N/W) x 192 f32 + de-interleave
        emb_bytes = int(world * sdist.shard_rows(n_total, world) * emb.shape[1] * 4)
        out["allgather"] = {
            "collective": "all_gather_into_tensor" if world > 1 else "none (world of one: the local embeddings are the gathered matrix)",
            "allgather_ms": exchange["allgather_ms"], "bytes_received_per_rank": emb_bytes if world > 1 else 0,
            "allgather_ms_by_rank": exchange.get("allgather_ms_by_rank", [exchange["allgather_ms"]]),
            "ms_per_step_by_rank": exchange.get("ms_per_step_by_rank", [dt / args.steps * 1e3]),
            "ms_per_step_min": min(exchange.get("ms_per_step_by_rank", [dt / args.steps * 1e3])),
            "ms_per_step_max": max(exchange.get("ms_per_step_by_rank", [dt / args.steps * 1e3])),
            "share_of_step_time": exchange["allgather_ms"] * 1e-3 * args.steps / dt,
            "order_check": "full[rank::world] == this rank's embeddings, bitwise, on every rank" if exchange["order_ok"] else "FAILED",
            "gathered_rows_checked": exchange.get("gathered_rows_checked", 0),
            "gathered_max_cosine_distance_vs_local_recompute": exchange.get("gathered_max_cosine_distance_vs_local_recompute"),
            "timing": "torch.cuda.Event pair on the launch stream around pad + collective + de-interleave of every timed step "
                      "(includes the wait for the slowest rank's forward: a collective cannot finish before its last participant arrives)",
        }
        if extra_f16 is not None:
            out["f16"] = extra_f16
        if extra_split is not None:
            out["f32_split16x3"] = extra_split
        if extra_ns is not None:
            out["f32_narrow_split16x3"] = extra_ns
        if world == 1 and not args.no_cpu_baseline:
            e32 = emb if args.precision == "f32" else None
            out["cpu_baseline"] = cpu_baseline(state_dict, wav, args.cpu_seconds, e32 if (extra_f16 is None and extra_split is None) else emb32,
                                               emb16 if extra_f16 is not None else (emb if args.precision == "f16" else None))
            if extra_split is not None and "max_cosine_distance_vs_gpu" in out["cpu_baseline"]:
                out["cpu_baseline"]["note_split16x3"] = "the f32-split16x3 embeddings sit max_cosine_distance_vs_f32_path from the exact-f32 ones"
            out["reference_batches"] = reference_batches(state_dict, dev, args.precision)
            out["der"] = der_vs_cpu(state_dict, dev, args.precision)
            out["der_vs_cpu"] = out["der"]["der_vs_cpu"]
            out["labels_identical"] = out["der"]["labels_identical"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
