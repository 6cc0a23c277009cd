#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[3] and configs[4] (not the headline bench):

  * streaming: 16 channels x 250 ms hop, 2 s window, hipGraph-captured fbank + ECAPA-TDNN (full
    spkrec-ecapa geometry), per-hop latency p50 / p99 and hops/s, graph replay vs eager launches;
  * 50k x 50k cosine affinity in one call.

    python tools/bench_configs.py [--hops 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def stream_latency(state_dict, dev, precision, use_graph, hops):
    from speech_diarization_amd.engine import EmbeddingEngine
    from speech_diarization_amd.streaming import StreamingEmbedder
    eng = EmbeddingEngine(state_dict, dev, max_batch=16, precision=precision)
    st = StreamingEmbedder(eng, channels=16, window_s=2.0, hop_s=0.25, use_graph=use_graph)
    chunk = (torch.randn(16, 4000, device=dev) * 0.1)
    for _ in range(10):
        st.push(chunk)
    torch.cuda.synchronize()
    lat = []
    for _ in range(hops):
        t0 = time.perf_counter()
        st.push(chunk)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.asarray(lat)
    return {"precision": precision, "graph": use_graph, "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "hops_per_s": float(1e3 / lat.mean()), "realtime_factor_16ch": float(250.0 / np.percentile(lat, 99))}


def host_api_rate(batch, n=32000, reps=100, warm=100, precision="f32"):
    """The reference's own call: numpy in -> numpy out (H2D + fbank + ECAPA + D2H per call), f32 engine.  Every call
    synchronises, so the card idles between calls and drops its clocks: the first ~100 calls after an idle period run
    30-90 % slower than the steady state (measured 5.6 / 4.3 / 3.0 ms per call of 32 segments); `warm` calls come first."""
    from speech_diarization_amd import speech_encode, synth
    speech_encode.set_precision(precision)
    wavs = synth.synthetic_segments(5, batch, n)
    for _ in range(warm):
        speech_encode.ecapa_encode_batch(wavs)
    lat = []
    for _ in range(reps):
        t0 = time.perf_counter()
        speech_encode.ecapa_encode_batch(wavs)
        lat.append(time.perf_counter() - t0)
    lat = np.asarray(lat)
    return {"precision": precision, "batch": batch, "n": n, "segments_per_s": batch / float(np.median(lat)), "ms_per_call_p50": float(np.median(lat)) * 1e3,
            "ms_per_call_p90": float(np.percentile(lat, 90)) * 1e3}


def micro_batch_sweep(state_dict, dev, precision):
    """Resident throughput vs micro-batch ({32, 128, 512, 2048}) on 4096 segments."""
    from speech_diarization_amd.engine import EmbeddingEngine
    wav = (torch.randn(4096, 32000, device=dev) * 0.1).clamp_(-1, 1)
    out = []
    for mb in (32, 128, 512, 2048):
        eng = EmbeddingEngine(state_dict, dev, max_batch=mb, precision=precision)
        eng.embed(wav)
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            eng.embed(wav)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out.append({"micro_batch": mb, "segments_per_s": 4096 / best})
        del eng
        torch.cuda.empty_cache()
    return out


def reassignment_windows(state_dict, dev, precision="f32"):
    """the frame-level reassignment workload of one hour of audio [REF anti_stick_diarize.py:396-430]: 1 s
    windows every 0.1 s = 36 k windows per hour, embedded (a) in place from the ONE resident signal (`embed_windows`:
    230 MB uploaded once) and (b) the round-2 way: a torch advanced-index gather that materialises [36 k, 16 000] f32
    (2.3 GB written and re-read) and `embed`.  Also the H2D volumes of a 1 h meeting at 2 s / 0.25 s windows."""
    from speech_diarization_amd.engine import EmbeddingEngine
    sr, win, step = 16000, 16000, 1600
    y = (torch.randn(3600 * sr, device=dev) * 0.1).clamp_(-1, 1)
    starts = torch.arange(0, y.numel() - win, step, device=dev, dtype=torch.int64)
    eng = EmbeddingEngine(state_dict, dev, max_batch=4096, precision=precision)
    res = {"precision": precision, "windows": int(starts.numel())}
    ar = torch.arange(win, device=dev)
    def in_place():
        return [eng.embed_windows(y, starts[lo:lo + 4096], win) for lo in range(0, starts.numel(), 4096)]
    def gathered():
        return [eng.embed(y[starts[lo:lo + 4096, None] + ar[None, :]]) for lo in range(0, starts.numel(), 4096)]
    for name, fn in (("in_place", in_place), ("gathered", gathered)):
        fn(); torch.cuda.synchronize()
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name + "_windows_per_s"] = starts.numel() / best
        res[name + "_s_per_hour_of_audio"] = best
    n_win_2s = (3600 * sr - 32000) // 4000 + 1
    res["h2d_bytes_1h_meeting"] = {"signal_once": 3600 * sr * 4, "gathered_2s_windows_every_0.25s": n_win_2s * 32000 * 4}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hops", type=int, default=200)
    a = ap.parse_args()
    from speech_diarization_amd import ops, synth
    dev = torch.device("cuda", 0)
    sd = synth.make_ecapa_state_dict(1234)
    out = {"streaming": [stream_latency(sd, dev, p, g, a.hops) for p in ("f32", "f32s", "f16") for g in (True, False)]}
    x = torch.randn(50000, 192, device=dev)
    K = torch.empty(50000, 50000, device=dev)
    for _ in range(3):                                   # (the first calls after other work run 10-50 % slower: clocks)
        ops.cosine_affinity(x, out=K)
    torch.cuda.synchronize()
    t = []
    for _ in range(8):
        t0 = time.perf_counter()
        ops.cosine_affinity(x, out=K)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    dt = min(t)
    out["micro_batch_sweep"] = {p: micro_batch_sweep(sd, dev, p) for p in ("f32", "f32ns", "f32s", "f16")}
    out["reassignment_windows_1h"] = [reassignment_windows(sd, dev, p) for p in ("f32", "f32s", "f16")]
    out["host_api_numpy_in_out"] = [host_api_rate(b, n, precision=p) for p in ("f32", "f32ns", "f32s", "f16") for b, n in ((32, 32000), (128, 32000), (128, 16000))]
    out["affinity_50k"] = {"ms": dt * 1e3, "tflops": 384.0 * 50000 ** 2 / dt / 1e12, "write_tb_s": 4.0 * 50000 ** 2 / dt / 1e12}
    for _ in range(3):
        ops.cosine_affinity(x, out=K, split16=True)
    torch.cuda.synchronize()
    t = []
    for _ in range(8):
        t0 = time.perf_counter()
        ops.cosine_affinity(x, out=K, split16=True)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    dt = min(t)
    out["affinity_50k_split16"] = {"ms": dt * 1e3, "f16_tflops": 3 * 384.0 * 50000 ** 2 / dt / 1e12, "write_tb_s": 4.0 * 50000 ** 2 / dt / 1e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
