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
import signal
import socket
import subprocess
import sys
import threading


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


def _stop(proc: subprocess.Popen, grace_s: float = 30.0) -> int:
    """End the child's whole process group (torchrun + its ranks): SIGTERM, `grace_s` seconds, then SIGKILL."""
    for sig in (signal.SIGTERM, signal.SIGKILL):
        if proc.poll() is not None:
            break
        try:
            os.killpg(proc.pid, sig)            # start_new_session=True: the child's pid is its process-group id
        except ProcessLookupError:
            break
        try:
            proc.wait(timeout=grace_s)
        except subprocess.TimeoutExpired:
            continue
    return proc.wait()


def _run_once(cmd: list[str], env: dict) -> tuple[int, bool]:
    """-> (exit code, the rendezvous port was taken).  SIGTERM / SIGHUP / SIGINT to the parent (a driver's timeout, a
    scheduler, a closed terminal, ^C) end the child's process group before the parent returns 128 + signal: N ranks
    holding N GPUs are never left behind as orphans.  The child is a CHILD (never exec) in a session of its own."""
    proc = subprocess.Popen(cmd, env=env, start_new_session=True, stderr=subprocess.PIPE)
    got = []

    def on_signal(signum, frame):
        got.append(signum)
        raise KeyboardInterrupt

    port_taken = [False]

    def relay():                                   # stderr passes through; the only thing read from it is the bind failure
        for line in iter(proc.stderr.readline, b""):
            if b"EADDRINUSE" in line or b"ddress already in use" in line:
                port_taken[0] = True
            sys.stderr.buffer.write(line)
            sys.stderr.buffer.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    handled = (signal.SIGTERM, signal.SIGHUP, signal.SIGINT)
    main_thread = threading.current_thread() is threading.main_thread()
    old = {s: signal.signal(s, on_signal) for s in handled} if main_thread else {}
    try:
        code = proc.wait()
    except KeyboardInterrupt:
        _stop(proc)
        code = 128 + (got[0] if got else signal.SIGINT)
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    t.join(timeout=5)
    return code, port_taken[0]


def self_launch(script: str, script_args: list[str], n_ranks: int) -> int:
    """Run `script script_args` as n_ranks torchrun ranks in a child process; stdout / stderr pass through. Returns the exit
    code.  `free_port` closes its probe socket before torchrun binds the port, so another job can take it in between: a
    child that dies on the bind is started again on a fresh port (at most three times)."""
    code = 1
    for _ in range(3):
        code, port_taken = _run_once(torchrun_command(script, script_args, n_ranks, free_port()), child_env())
        if code == 0 or not port_taken or code >= 128:
            break
    return code
