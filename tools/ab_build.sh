#!/bin/bash
# A/B on the GPU box: build an experiment VARIANT of libsd_hip.so with extra -D flags (the product library is never
# overwritten, so a later test or bench cannot silently run a diagnostic binary), then run a command against it.
# usage: tools/ab_build.sh <name> "<cflags>" <cmd...>
set -e
name="$1"; flags="$2"; shift 2
lib=$(python speech-diarization_amd/build_native.py --variant "$name" "$flags")
echo "=== variant $name [$flags] -> $lib"
SD_EXPERIMENT=1 SD_HIP_LIB="$lib" "$@"
