#!/usr/bin/env python3
"""The counters `bench.py` quotes from files (`profiles/traffic*.json`, `profiles/mfma_util*.json`: separate rocprofv3 --pmc passes) must
come from the kernels that are in the tree.  `tools/traffic_from_pmc.py` / `mfma_util_from_pmc.py` record the source fingerprint of
`csrc/` + `include/` (the one `build_native.py` stamps the library with) in each file as `_source_fingerprint`; this script compares it
with today's sources and exits 1 on a mismatch or a missing record.  `tools/refresh_profiles.sh` runs it last; `bench.py` puts the result
on its line as `counters_match_sources`.
    python tools/check_profiles_fresh.py [--stamp]     # --stamp: write today's fingerprint into the files (after a refresh)"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["traffic.json", "traffic_f16.json", "traffic_f32s.json", "mfma_util.json", "mfma_util_f16.json", "mfma_util_f32s.json"]


def fingerprint() -> str:
    spec = importlib.util.spec_from_file_location("sd_build_native", os.path.join(ROOT, "speech-diarization_amd", "build_native.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._fingerprint()


def status() -> dict:
    now = fingerprint()
    out = {}
    for name in FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            out[name] = "missing"
            continue
        rec = json.load(open(path)).get("_source_fingerprint")
        out[name] = "ok" if rec == now else ("no fingerprint recorded" if rec is None else "older than the kernel sources")
    return out


def main() -> int:
    if "--stamp" in sys.argv:
        now = fingerprint()
        for name in FILES:
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                d = json.load(open(path))
                d["_source_fingerprint"] = now
                json.dump(d, open(path, "w"), indent=1)
        print("stamped", now[:16])
        return 0
    st = status()
    for k, v in st.items():
        print(f"{k}: {v}")
    return 0 if all(v == "ok" for v in st.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
