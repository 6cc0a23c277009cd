"""Multi-GPU sharding of the embedding path: one process per GPU, RCCL over xGMI.

Segments are independent (the reference already treats a batch as independent rows,
[REF anti_stick_diarize.py:150-171]), so the path shards with no data-path collective
except ONE exchange: an all-gather of the 192-d embeddings before clustering.

* segment i -> rank i mod W (round-robin balances variable-length VAD segments);
* each rank pads its shard to ceil(N / W) rows, `all_gather_into_tensor` moves
  W x ceil(N/W) x 192 f32 (latency-bound at these sizes: 3.45 MB per rank for a 1 h
  meeting), rows are de-interleaved back to the original order;
* the N x N affinity is never moved over xGMI: every consumer recomputes what it
  needs from the gathered N x 192 matrix (its own row block, or everything on rank 0).

`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on `gloo`
with CPU tensors, which is how the index math is tested without GPUs.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). Returns (rank, local_rank, world)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
    return rank, local_rank, world_size


def shard_indices(n: int, rank: int, world_size: int) -> np.ndarray:
    """Indices of the segments rank `rank` owns: i with i mod W == rank."""
    return np.arange(rank, n, world_size, dtype=np.int64)


def shard_rows(n: int, world_size: int) -> int:
    return (n + world_size - 1) // world_size


def deinterleave(gathered: torch.Tensor, n: int, world_size: int) -> torch.Tensor:
    """gathered [W, rows, D] (rank-major) -> [n, D] in original segment order (row k*W + r <- gathered[r, k])."""
    w, rows, d = gathered.shape
    assert w == world_size
    return gathered.permute(1, 0, 2).reshape(rows * w, d)[:n]


def all_gather_embeddings(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """local: this rank's [len(shard_indices), D] embeddings -> every rank gets the full [n_total, D], original order."""
    rank, w = world()
    if w == 1 and not (dist.is_available() and dist.is_initialized() and os.environ.get("SD_DIST_FORCE_COLLECTIVE") == "1"):
        return local[:n_total]          # (SD_DIST_FORCE_COLLECTIVE=1: run the collective even in a world of one, for tests)
    rows = shard_rows(n_total, w)
    d = local.shape[1]
    send = torch.zeros((rows, d), dtype=local.dtype, device=local.device)
    send[: local.shape[0]] = local
    out = torch.empty((w * rows, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send)
    return deinterleave(out.view(w, rows, d), n_total, w)


def row_block(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [lo, hi) row block of an N-row result owned by `rank` (affinity rows)."""
    per = shard_rows(n, world_size)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)
