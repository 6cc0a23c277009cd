#!/usr/bin/env python3
"""rocprofv3-reported matrix-core utilisation per kernel from one --pmc pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv ...
    python tools/mfma_util_from_pmc.py <counter_collection.csv> --out profiles/mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over all
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note), so the kernel's
active cycles are GRBM_GUI_ACTIVE / 8 and the utilisation is
    MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
"""
import argparse
import collections
import csv
import json
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--out", default="profiles/mfma_util.json")
    ap.add_argument("--note", default="")
    ap.add_argument("--fingerprint", default="", help="source fingerprint of csrc/ + include/ the profiled library was built from (tools/check_profiles_fresh.py)")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(a.csv)):
        kn = r["Kernel_Name"]            # Itanium-mangled names first: the demangled pattern would swallow the _ZN12_GLOBAL__N_1.. prefix
        m = re.search(r"\d([a-z][a-z0-9_]*?_kernel)", kn) if kn.startswith("_Z") else re.search(r"([A-Za-z_0-9]+_kernel)\b", kn)
        if m:
            acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {"_note": a.note or "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), summed over all launches of the kernel"}
    for k, v in sorted(acc.items()):
        act = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if act > 0 and busy > 0:
            out[k] = {"mfma_util": busy / (act * 1024.0), "mfma_busy_cycles": busy, "active_cycles": act}
    if a.fingerprint:
        out["_source_fingerprint"] = a.fingerprint
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
