"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes exercise the round-robin shard,
the padded all-gather and the de-interleave index math of speech_diarization_amd.dist."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speech_diarization_amd import dist as sdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from speech_diarization_amd import dist as sd
    r, lr, w = sd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    idx = sd.shard_indices(n_total, rank, world)
    # "embedding" of segment i is a row that encodes i, so ordering mistakes are visible
    local = torch.stack([torch.full((192,), float(i)) + torch.arange(192) / 1000.0 for i in idx.tolist()]) if len(idx) else torch.zeros(0, 192)
    full = sd.all_gather_embeddings(local, n_total)
    lo, hi = sd.row_block(n_total, rank, world)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"block_{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 11), (2, 8), (3, 10), (2, 1)])
def test_all_gather_embeddings_restores_segment_order(tmp_path, world, n_total):
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    expect = torch.stack([torch.full((192,), float(i)) + torch.arange(192) / 1000.0 for i in range(n_total)]).numpy()
    covered = []
    for r in range(world):
        full = np.load(tmp_path / f"full_{r}.npy")
        assert full.shape == (n_total, 192)
        assert np.array_equal(full, expect)            # every rank holds all embeddings, original order
        lo, hi = np.load(tmp_path / f"block_{r}.npy")
        covered.extend(range(lo, hi))
    assert covered == list(range(n_total))             # affinity row blocks tile [0, N) exactly once


def test_shard_and_deinterleave_index_math():
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            shards = [sdist.shard_indices(n, r, w) for r in range(w)]
            assert sorted(np.concatenate(shards).tolist()) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            rows = sdist.shard_rows(n, w)
            g = torch.full((w, rows, 2), -1.0)
            for r, s in enumerate(shards):
                g[r, : len(s), 0] = torch.from_numpy(s).float()
            assert sdist.deinterleave(g, n, w)[:, 0].tolist() == [float(i) for i in range(n)]
    assert sdist.world() == (0, 1)
    x = torch.randn(5, 192)
    assert torch.equal(sdist.all_gather_embeddings(x, 5), x)     # single process: identity


# ---- configs[2] through the product entry: diarize_audio(world="dist") under a 2- and 3-process gloo group with an
# injected CPU encoder; the RTTM must be byte-identical to the single-process run [REF diarization_baseline.py:236-266]

def _band_encoder(wavs):
    """Cheap deterministic, row-independent CPU encoder: 192 log band energies of the window's spectrum
    (separates the synthetic voices: their harmonics sit in different bands)."""
    w = np.asarray(wavs, dtype=np.float64)
    spec = np.abs(np.fft.rfft(w * np.hanning(w.shape[1])[None, :], axis=1)) ** 2
    edges = np.linspace(0, 2400 * w.shape[1] // 16000, 193).astype(int)
    bands = np.stack([spec[:, edges[d]:max(edges[d + 1], edges[d] + 1)].mean(axis=1) for d in range(192)], axis=1)
    return np.log(bands + 1e-8).astype(np.float32)


def _diarize_worker(rank, world, port, wav_path, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from speech_diarization_amd import diarization_baseline as db, dist as sd
    sd.init_from_env("gloo")
    calls = []

    def enc(w):
        calls.append(int(w.shape[0]))
        return _band_encoder(w)

    out = os.path.join(out_dir, f"w{world}.rttm")       # same path on every rank: only rank 0 may write it
    segs, det = db.diarize_audio(wav_path, 0.35, 0.1, 2, 6, rttm_filepath=out, encoder=enc, clustering="spectral",
                                 return_details=True, world="dist")
    np.save(os.path.join(out_dir, f"emb_w{world}_r{rank}.npy"), det["embeddings"])
    with open(os.path.join(out_dir, f"segs_w{world}_r{rank}.txt"), "w") as f:
        f.write(repr(segs) + f"\n{sum(calls)}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_diarize_audio_sharded_gives_the_single_process_rttm(tmp_path):
    from speech_diarization_amd import audio_io, diarization_baseline as db, synth
    conv = synth.synthetic_conversation(40.0, n_speakers=2, seed=3)
    wav = tmp_path / "meeting.wav"
    audio_io.write_wav16(wav, conv.wav, conv.sr)
    ref_rttm = tmp_path / "w1.rttm"
    segs1, det1 = db.diarize_audio(wav, 0.35, 0.1, 2, 6, rttm_filepath=ref_rttm, encoder=_band_encoder, clustering="spectral",
                                   return_details=True)
    n = det1["embeddings"].shape[0]
    assert n > 20 and len({k for _, _, k in segs1}) == 2
    for world in (2, 3):
        mp.spawn(_diarize_worker, args=(world, _free_port(), str(wav), str(tmp_path)), nprocs=world, join=True)
        assert (tmp_path / f"w{world}.rttm").read_bytes() == ref_rttm.read_bytes()
        embedded = 0
        for r in range(world):
            assert np.array_equal(np.load(tmp_path / f"emb_w{world}_r{r}.npy"), det1["embeddings"])   # every rank: all rows, window order
            text = (tmp_path / f"segs_w{world}_r{r}.txt").read_text().splitlines()
            assert text[0] == repr(segs1)
            assert int(text[1]) == len(sdist.shard_indices(n, r, world))     # each rank embedded only its shard
            embedded += int(text[1])
        assert embedded == n
