#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): rocprofv3 kernel stats for the f32, f16 and
# f32-split16x3 bench steps, then separate PMC passes (one counter group per pass, --kernel-trace/--stats only, as
# MI355X_MICROARCH.md prescribes).  Output under gpurun_out/refresh/; copy what is judged into profiles/.
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/refresh; mkdir -p $out
common="--no-f16-extra --no-split-extra --no-cpu-baseline"
precisions=${PRECISIONS:-"f32 f16 f32s"}          # PRECISIONS="f32s" bash tools/refresh_profiles.sh: only that engine
rm -rf $out
mkdir -p $out
# the fingerprint of the kernel sources THIS box runs (build_native.py's stamp): tools/import_profiles.py records it in the JSON files, and
# tools/check_profiles_fresh.py / bench.py compare it with the tree, so the bench line can never quote counters of an older kernel unnoticed
python3 - > $out/source_fingerprint.txt <<'PY'
import importlib.util
spec = importlib.util.spec_from_file_location("b", "speech-diarization_amd/build_native.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(m._fingerprint())
PY
for p in $precisions; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$p -- python3 bench.py --precision $p --steps 3 --warmup 1 $common > $out/$p.log 2>&1
  echo "$p stats done"
done
for p in $precisions; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_$p -- python3 bench.py --precision $p --steps 1 --warmup 0 $common > $out/pmc_${tag}_$p.log 2>&1
    echo "pmc $tag $p done"
  done
done
find $out -name "*.csv" | sort
