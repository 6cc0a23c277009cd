#!/bin/bash
# Runs ON THE GPU BOX: kernel stats and FETCH_SIZE / WRITE_SIZE passes of the 50 k x 50 k split16x3 affinity (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes).  Output under gpurun_out/aff_prof/.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/aff_prof; rm -rf $out; mkdir -p $out
export REPS=3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/time_affinity.py 50000 > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 tools/time_affinity.py 50000 > $out/pmc_$c.log 2>&1
done
find $out -name "*.csv" | sort
