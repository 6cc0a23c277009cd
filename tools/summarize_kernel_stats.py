#!/usr/bin/env python3
"""rocprofv3 `*_kernel_stats.csv` -> the per-call table kept under profiles/ (kernels per call, average duration, share of a call).

    python tools/summarize_kernel_stats.py <kernel_stats.csv> <calls in the run> ["header line"]
"""
import csv, sys

path, ncalls = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
if len(sys.argv) > 3:
    print(sys.argv[3])
print(f"kernel time per call: {tot / ncalls / 1e6:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t, n = float(r["TotalDurationNs"]), int(r["Calls"])
    print(f"{r['Name'][:80]:80s} {n / ncalls:6.1f}/call  avg {t / n / 1e3:8.1f} us  per call {t / ncalls / 1e3:8.1f} us  {100 * t / tot:5.1f} %")
