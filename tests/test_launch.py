"""`python bench.py --gpus N` / `tools/diarize_sharded.py --gpus N` from a plain shell: the GPU-free parent builds the
torch.distributed.run command of the contract, starts it as a child process, relays rank 0's line and the exit code."""
import json
import os
import subprocess
import sys

from speech_diarization_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "helpers", "launch_probe.py")


def test_command_is_the_contracts_launch_form():
    cmd = launch.torchrun_command("/x/bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29533, python="python3")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29533", "/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "3"]


def test_who_launches():
    assert not launch.needs_self_launch(1, {})
    assert launch.needs_self_launch(8, {})
    assert launch.needs_self_launch(2, {"WORLD_SIZE": "1"})                            # a stray WORLD_SIZE alone is not torchrun
    assert not launch.needs_self_launch(8, {"RANK": "3", "WORLD_SIZE": "8"})           # a rank never launches again
    env = launch.child_env({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "PATH": "/bin"})
    assert "RANK" not in env and "WORLD_SIZE" not in env and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"


def _run(args, **kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=600, env=env, **kw)


def test_parent_starts_two_ranks_and_relays_one_line():
    res = _run([PROBE, "--gpus", "2", "--tag", "a b"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["ranks"] == 2 and out["n_ranks_seen"] == 2 and out["rows"] == 6 and out["order_ok"]
    assert out["argv"] == ["--gpus", "2", "--tag", "a b"]                              # the ranks see the parent's arguments unchanged


def test_parent_relays_a_failing_rank():
    res = _run([PROBE, "--gpus", "2", "--fail"])
    assert res.returncode != 0


def test_sigterm_to_the_parent_ends_every_rank(tmp_path):
    """A driver's timeout or a scheduler signals the PARENT: torchrun and its ranks (which would hold the GPUs) must go with it."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    parent = subprocess.Popen([sys.executable, PROBE, "--gpus", "2", "--hang", str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    pid_files = [tmp_path / "rank0.pid", tmp_path / "rank1.pid"]
    deadline = time.time() + 300
    while not all(f.exists() and f.read_text() for f in pid_files):
        assert time.time() < deadline and parent.poll() is None, "ranks never started"
        time.sleep(0.2)
    pids = [int(f.read_text()) for f in pid_files]
    parent.send_signal(signal.SIGTERM)
    assert parent.wait(timeout=90) == 128 + signal.SIGTERM

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:                                        # a zombie waiting for init to reap it is not a live rank
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False

    deadline = time.time() + 30
    while any(alive(p) for p in pids) and time.time() < deadline:
        time.sleep(0.2)
    assert not any(alive(p) for p in pids)


def test_bench_parent_launches_without_touching_the_gpu():
    """No GPU in this container: the ranks of `bench.py --gpus 2` must each stop at "needs a GPU"; the parent itself never
    asks for one (it would raise before launching) and returns the job's non-zero exit code."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: the ranks would run the whole benchmark")
    res = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--segments", "4", "--steps", "1", "--warmup", "0"])
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs a GPU") >= 2, res.stderr[-3000:]
