#!/usr/bin/env python3
"""(build container) Turn what `tools/refresh_profiles.sh` left under gpurun_out/refresh/ into the judged files under profiles/:
    python tools/import_profiles.py r05
-> profiles/<tag>_{f32,f16,f32s}_kernel_stats.csv, <tag>_pmc_{fetch_size,write_size,mfma_busy}_<p>.csv, <tag>_traffic_<p>.json,
   <tag>_mfma_util_<p>.json and the untagged current copies bench.py reads (traffic.json, traffic_f16.json, traffic_f32s.json,
   mfma_util*.json), each carrying the source fingerprint the GPU box computed from ITS copy of the tree (source_fingerprint.txt)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
tag = sys.argv[1]
fp = open(os.path.join(SRC, "source_fingerprint.txt")).read().strip()
cur = {"f32": "", "f16": "_f16", "f32s": "_f32s"}


def one(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    if not hits:
        raise SystemExit(f"nothing matches {pattern} under {SRC}")
    return hits[-1]


for p in ("f32", "f16", "f32s"):
    if not os.path.isdir(os.path.join(SRC, p)):
        continue
    shutil.copy(one(f"{p}/**/*kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_{p}_kernel_stats.csv"))
    csvs = {}
    for c, short in (("FETCH_SIZE", "fetch_size"), ("WRITE_SIZE", "write_size"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy")):
        csvs[short] = os.path.join(ROOT, "profiles", f"{tag}_pmc_{short}_{p}.csv")
        shutil.copy(one(f"pmc_{c}_{p}/**/*counter_collection.csv"), csvs[short])
    note = f"round {tag[1:]}, bench.py --precision {p}"
    for out in (f"{tag}_traffic_{p}.json", f"traffic{cur[p]}.json"):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "traffic_from_pmc.py"), csvs["fetch_size"], csvs["write_size"],
                               "--out", os.path.join(ROOT, "profiles", out), "--note", note, "--fingerprint", fp], stdout=subprocess.DEVNULL)
    for out in (f"{tag}_mfma_util_{p}.json", f"mfma_util{cur[p]}.json"):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_util_from_pmc.py"), csvs["mfma_busy"],
                               "--out", os.path.join(ROOT, "profiles", out), "--note", note, "--fingerprint", fp], stdout=subprocess.DEVNULL)
    print(p, "imported")
print("source fingerprint", fp[:16])
