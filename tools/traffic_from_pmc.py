#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic.

    python tools/traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        --out profiles/traffic.json

Units and corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(16 B per lane), so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The result
is the average over all launches of a kernel in the profiled run, which is how bench.py's
`roofline.achieved` is averaged.
"""
import argparse
import collections
import csv
import json
import re


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        kn = r["Kernel_Name"]            # Itanium-mangled names (_ZN12_GLOBAL__N_125conv_..._kernelI...) first: the demangled pattern would swallow the prefix
        m = re.search(r"\d([a-z][a-z0-9_]*?_kernel)", kn) if kn.startswith("_Z") else re.search(r"([A-Za-z_0-9]+_kernel(?:<(?:true|false)>)?)", kn)
        if not m:
            continue                        # torch / runtime helper kernels
        name = m.group(1)
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--out", default="profiles/traffic.json")
    ap.add_argument("--note", default="")
    ap.add_argument("--fingerprint", default="", help="source fingerprint of csrc/ + include/ the profiled library was built from (tools/check_profiles_fresh.py)")
    a = ap.parse_args()
    f_tot, f_cnt = per_kernel(a.fetch_csv, "FETCH_SIZE")
    w_tot, w_cnt = per_kernel(a.write_csv, "WRITE_SIZE")
    out = {"_note": "average HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches; " + a.note}
    detail = {}
    for k in sorted(set(f_tot) | set(w_tot)):
        n = max(f_cnt.get(k, 0), w_cnt.get(k, 0), 1)
        fetch = 2.0 * f_tot.get(k, 0.0) * 1024.0 / max(f_cnt.get(k, 0), 1)
        write = w_tot.get(k, 0.0) * 1024.0 / max(w_cnt.get(k, 0), 1)
        out[k] = fetch + write
        detail[k] = {"launches": n, "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
                     "fetch_size_raw_kib_total": f_tot.get(k, 0.0), "write_size_raw_kib_total": w_tot.get(k, 0.0)}
    for k in list(detail):               # also under the bare name when only one instantiation of a kernel ran (bench.py looks kernels up by bare name)
        bare = k.split("<")[0]
        if bare != k and sum(1 for q in detail if q.split("<")[0] == bare) == 1:
            out[bare] = out[k]
    out["_detail"] = detail
    if a.fingerprint:
        out["_source_fingerprint"] = a.fingerprint
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in detail.items():
        print(f"{k:32s} launches={v['launches']:5d} fetch={v['fetch_bytes_per_launch_corrected'] / 1e6:10.2f} MB  write={v['write_bytes_per_launch'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main()
