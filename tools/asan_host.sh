#!/bin/bash
# ASan + UBSan pass over the HOST side of the C ABI.  BUILD CONTAINER ONLY: no GPU is needed and sanitizer
# runs are refused on the GPU box.  Builds a variant of libsd_hip.so whose host objects are instrumented (-fsanitize=address,undefined;
# the gfx950 device code is compiled as always: -fno-gpu-sanitize), then runs
#   1. tools/asan_host_driver.c (plain C against include/sd_hip.h: refusal paths, size queries, the schedule's host code), and
#   2. tests/test_abi_and_host.py through ctypes with the sanitizer runtime preloaded.
# Log: profiles/r05_asan_host.log.  Exit code != 0 on any sanitizer report or failed expectation.
set -eo pipefail
cd "$(dirname "$0")/.."
log=profiles/r05_asan_host.log
san="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan"
lib=$(python speech-diarization_amd/build_native.py --variant asan "$san")
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
{
  echo "# $(date -u +%F) ASan + UBSan, host side of libsd_hip.so (variant: $lib; runtime: $rt)"
  echo "## 1. C driver (tools/asan_host_driver.c)"
  /opt/rocm/lib/llvm/bin/clang -std=c11 -Wall -Wextra -Werror -fsanitize=address,undefined -shared-libsan -g \
      -Iinclude tools/asan_host_driver.c -o /tmp/asan_host_driver -L"$(dirname "$lib")" -l:"$(basename "$lib")" -lm \
      -Wl,-rpath,"$(dirname "$lib")" -Wl,-rpath,"$(dirname "$rt")" -Wl,-rpath,/opt/rocm/lib
  /tmp/asan_host_driver
  echo "driver exit code $?"
  echo "## 2. tests/test_abi_and_host.py against the instrumented library (LD_PRELOAD of the runtime)"
  LD_PRELOAD="$rt" SD_EXPERIMENT=1 SD_HIP_LIB="$lib" python -m pytest tests/test_abi_and_host.py -q -p no:cacheprovider 2>&1 | tail -5
} 2>&1 | tee "$log"
if grep -q "ERROR: AddressSanitizer\|runtime error:\|FAIL " "$log"; then echo "sanitizer findings: see $log"; exit 1; fi
echo "clean: $log"
