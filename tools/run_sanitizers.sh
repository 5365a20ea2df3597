#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU side (the oracle and every CPU test that drives it).
# GPU sanitizers are not available on this pool; the HIP kernels are covered by the differential tests instead.
# usage: bash tools/run_sanitizers.sh      (from the repo root; needs gcc's libasan / libubsan)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/oracle" san
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
VUS_ORACLE_LIB="$ROOT/oracle/libvus_oracle_san.so" \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
python -m pytest "$ROOT/tests" -x -q -m "not gpu" -p no:cacheprovider
