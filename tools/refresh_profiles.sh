#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): rocprofv3 kernel stats for the f32 and
# f16 bench steps, then separate PMC passes (one counter group per pass, --kernel-trace/--stats only, as
# MI355X_MICROARCH.md prescribes).  Output under gpurun_out/refresh/; copy what is judged into profiles/.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refresh; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f32 -- python3 bench.py --steps 3 --warmup 1 --no-f16-extra --no-cpu-baseline > $out/f32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f16 -- python3 bench.py --precision f16 --steps 3 --warmup 1 --no-cpu-baseline > $out/f16.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --no-f16-extra --no-cpu-baseline > $out/pmc_$tag.log 2>&1
done

for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_f16 -- python3 bench.py --precision f16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_${c}_f16.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma_f16 -- python3 bench.py --precision f16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_mfma_f16.log 2>&1
find $out -name "*.csv" | sort
