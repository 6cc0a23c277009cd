"""Child of tests/test_launch.py: one rank of a self-launched job (gloo, CPU). Rank 0 prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from speech_diarization_amd import dist as sdist, launch  # noqa: E402


def main():
    n = int(sys.argv[sys.argv.index("--gpus") + 1])
    if launch.needs_self_launch(n):
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], n))
    rank, local_rank, world = sdist.init_from_env("gloo")
    ones = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(ones)
    local = torch.full((3, 192), float(rank))
    full = sdist.all_gather_embeddings(local, 3 * world)
    if rank == 0:
        print(json.dumps({"ranks": world, "n_ranks_seen": int(ones.item()), "rows": int(full.shape[0]),
                          "order_ok": bool((full[:, 0] == torch.arange(3 * world) % world).all()), "argv": sys.argv[1:]}), flush=True)
    if "--hang" in sys.argv:                        # a job the parent has to be told to stop: every rank reports its pid, then sleeps
        import time
        d = sys.argv[sys.argv.index("--hang") + 1]
        with open(os.path.join(d, f"rank{rank}.pid"), "w") as f:
            f.write(str(os.getpid()))
        time.sleep(600)
    if "--fail" in sys.argv and rank == world - 1:
        raise SystemExit(7)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
