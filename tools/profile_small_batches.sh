#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the reference's own call, ecapa_encode_batch(numpy [B, 32000]), at B = 16 / 32 / 64 / 128
# (tools/profile_batch32.py: 100 + 200 calls each).  Output: gpurun_out/prof_b<B>/; tools/summarize_kernel_stats.py makes the tables.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prec=${1:-f32}
for b in ${BATCHES:-16 32 64 128}; do
  rm -rf gpurun_out/prof_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$b -o b$b -- python3 tools/profile_batch32.py $prec $b > gpurun_out/prof_b$b.log 2>&1
  grep "segments/s" gpurun_out/prof_b$b.log
done
