#!/bin/bash
# Round 5, f16 wide kernel (256x256 ring, register epilogue): the hardware dispatch (one workgroup per tile) against the super-tile walk
# (SD_TUNE_T256_LOCKSTEP_TILES = 0: 256 persistent workgroups, per pass 8 row panels x 4 column tiles per XCD):
# time (tools/sweep_f16.py, interleaved processes) and fabric traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), same box.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05f; rm -rf $out; mkdir -p $out
v=speech-diarization_amd/variants/libsd_hip_colmajor.so
shapes="1024,1024 3072,3072"
for round in 1 2; do
  echo "== hardware dispatch"; python tools/sweep_f16.py --B 5000 --rounds 5 --lockstep 1 $shapes
  echo "== super-tile walk (256 persistent workgroups, 8 x 4 tiles per XCD and pass)"; python tools/sweep_f16.py --B 5000 --rounds 5 --lockstep 0 $shapes
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_dispatch -- python3 tools/sweep_f16.py --B 5000 --rounds 2 --lockstep 1 $shapes > $out/pmc_${c}_dispatch.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_lockstep -- python3 tools/sweep_f16.py --B 5000 --rounds 2 --lockstep 0 $shapes > $out/pmc_${c}_lockstep.log 2>&1
  echo "pmc $c done"
done
python3 - <<'PY'
import csv, glob, collections
M = 1005000
alg = {"8": M * 1024 * 2 + 1024 * 1024 * 2, "24": M * 3072 * 2 + 3072 * 3072 * 2}     # activations + weights read once
for tag in ("dispatch", "lockstep"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(f"gpurun_out/r05f/pmc_{c}_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "t256" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    tot[r["Kernel_Name"][:70]][0] += float(r["Counter_Value"]); tot[r["Kernel_Name"][:70]][1] += 1
        for k, (v, n) in sorted(tot.items()):
            print(tag, c, k, f"raw KiB summed over {n} launches (both shapes) {v:.0f}")
PY
