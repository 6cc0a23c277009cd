"""One process per GPU, started from a parent that never touches the GPU.

`python bench.py --gpus 8` (or `tools/diarize_sharded.py --gpus 8`) is how a driver starts the
N-GPU job; the ranks themselves must run under `torch.distributed.run` so that RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* are set.  The parent therefore decides from argv and the environment alone,
before any HIP call, and starts torchrun as a CHILD process (never `os.exec*`: replacing a process
image after the GPU was initialised takes the node down on this pool), relays the child's stdout
(rank 0's JSON line) and returns its exit code.

The reference has nothing to mirror here: its entry point hard-codes `device=0`
[REF diarization_baseline.py:245].
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def under_torchrun(env=None) -> bool:
    """True when this process is already one rank of a launched job."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def needs_self_launch(n_gpus: int, env=None) -> bool:
    """A parent asked for N > 1 ranks and is not itself a rank."""
    return n_gpus > 1 and not under_torchrun(env)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun_command(script: str, script_args: list[str], n_ranks: int, port: int, python: str | None = None) -> list[str]:
    """The command line of the child: the contract's own launch form, one rank per GPU of ONE node."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(script_args)


def child_env(env=None) -> dict:
    """Environment of the child: the parent's, with the dmabuf IPC switch RCCL needs on this pool kept / set."""
    out = dict(os.environ if env is None else env)
    out.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        out.pop(k, None)
    return out


def self_launch(script: str, script_args: list[str], n_ranks: int) -> int:
    """Run `script script_args` as n_ranks torchrun ranks in a child process; stdout / stderr pass through. Returns the exit code."""
    cmd = torchrun_command(script, script_args, n_ranks, free_port())
    proc = subprocess.Popen(cmd, env=child_env())
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        try:
            return proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
            return proc.wait()
