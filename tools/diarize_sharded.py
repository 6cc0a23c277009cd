#!/usr/bin/env python3
"""BASELINE.json configs[2] as a launchable job: one process per GPU, VAD windows sharded round-robin,
ONE RCCL all-gather of the 192-d embeddings, clustering, RTTM from rank 0.

    python tools/diarize_sharded.py meeting.wav --rttm meeting.rttm --gpus 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/diarize_sharded.py meeting.wav --rttm meeting.rttm

The first form starts the second as a child process (the parent never touches the GPU: `launch.self_launch`).

Same entry point and output as the single-process call [REF diarization_baseline.py:236-266]; with
WORLD_SIZE unset it IS the single-process call.  --synthetic N writes an N-second 8-speaker test meeting first.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("audio")
    ap.add_argument("--rttm", default=None)
    ap.add_argument("--backend", default=None, help="nccl (= RCCL, default on a GPU box) or gloo")
    ap.add_argument("--synthetic", type=float, default=0.0, help="generate this many seconds of an 8-speaker meeting at AUDIO (rank 0)")
    ap.add_argument("--min-speakers", type=int, default=2)
    ap.add_argument("--max-speakers", type=int, default=8)
    ap.add_argument("--clustering", default="spectral")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start (one per GPU) when not already under torchrun")
    a = ap.parse_args()
    from speech_diarization_amd import launch
    if launch.needs_self_launch(a.gpus):
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    import torch
    import torch.distributed as tdist
    from speech_diarization_amd import audio_io, diarization_baseline as db, dist, synth
    rank, local_rank, world = dist.init_from_env(a.backend)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if a.synthetic > 0:
        if rank == 0:
            conv = synth.synthetic_conversation(a.synthetic, n_speakers=8, seed=0)
            audio_io.write_wav16(a.audio, conv.wav, conv.sr)
        if world > 1:
            tdist.barrier()
    t0 = time.time()
    segs, det = db.diarize_audio(a.audio, 0.35, 0.1, a.min_speakers, a.max_speakers, rttm_filepath=a.rttm or os.path.splitext(a.audio)[0] + ".rttm",
                                 clustering=a.clustering, return_details=True, world="dist" if world > 1 else None)
    if rank == 0:
        print(f"world={world} backend={tdist.get_backend() if world > 1 else 'none'} windows={det['embeddings'].shape[0]} "
              f"speakers={len({k for _, _, k in segs})} turns={len(segs)} wall={time.time() - t0:.2f}s")
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
