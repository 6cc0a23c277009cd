#!/bin/bash
# A/B on the GPU box: rebuild libsd_hip.so with extra -D flags, then run a command.  usage: tools/ab_build.sh "<cflags>" <cmd...>
set -e
flags="$1"; shift
SD_EXTRA_CFLAGS="$flags" python speech-diarization_amd/build_native.py --force > /dev/null
echo "=== build [$flags]"
"$@"
