#!/bin/bash
# Runs ON THE GPU BOX: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA-busy) over the reference's numpy call at one batch size
# (tools/profile_batch32.py), as MI355X_MICROARCH.md prescribes (counters alone, no trace domains).  Output: gpurun_out/pmc_b<B>_<counter>/.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
b=${1:-32}
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_b${b}_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_b${b}_$tag -o p -- python3 tools/profile_batch32.py f32 $b > gpurun_out/pmc_b${b}_$tag.log 2>&1
  echo "pmc $tag done: $(grep segments/s gpurun_out/pmc_b${b}_$tag.log)"
done
find gpurun_out -name "p_counter_collection.csv" | sort
