#!/usr/bin/env python3
"""N x N cosine affinity timings (BASELINE configs[4]): exact f32 (triangle + mirror unless SD_AFFINITY_SYM=0) and split16.

    python tools/time_affinity.py [N ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    from speech_diarization_amd import ops
    dev = torch.device("cuda", 0)
    out = []
    for n in [int(v) for v in sys.argv[1:]] or [8192, 20000, 50000]:
        x = torch.randn(n, 192, device=dev)
        K = torch.empty(n, n, device=dev)
        rec = {"n": n}
        for name, kw in (("f32", {}), ("split16", {"split16": True})):
            for _ in range(2):
                ops.cosine_affinity(x, out=K, **kw)
            torch.cuda.synchronize()
            t = []
            for _ in range(int(os.environ.get('REPS', '5'))):
                t0 = time.perf_counter()
                ops.cosine_affinity(x, out=K, **kw)
                torch.cuda.synchronize()
                t.append(time.perf_counter() - t0)
            dt = min(t)
            rec[name] = {"ms": dt * 1e3, "write_tb_s": 4.0 * n * n / dt / 1e12, "all_ms": [round(v * 1e3, 3) for v in t]}
        out.append(rec)
    print(json.dumps({"sym": os.environ.get("SD_AFFINITY_SYM", "1"), "runs": out}))


if __name__ == "__main__":
    main()
