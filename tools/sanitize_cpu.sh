#!/bin/bash
# One AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU builds (round-2 VERDICT item 7):
#   * the oracle (oracle/libm17oracle_san.so) under the whole non-GPU test suite,
#   * the C++ operator surface (tests/cxx/mirror_check, incl. the scalar M17Demodulator of detail/scalar_demod.h) under its tests.
# CPU only: the GPU build is never instrumented (gpurun refuses sanitizer runs).  Usage: tools/sanitize_cpu.sh   (from the repo root)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/m17-cxx-demod_amd"
make -s -C "$ROOT/oracle" san
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ -std=c++20 $SAN -ffp-contract=off -I "$PKG/include/m17cxx" "$ROOT/tests/cxx/mirror_check.cpp" -L "$PKG" -lm17hip -L/opt/rocm/lib \
    -Wl,-rpath,"$PKG" -Wl,-rpath,/opt/rocm/lib -o "$ROOT/tests/cxx/mirror_check_san"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export M17_ORACLE_LIB="$ROOT/oracle/libm17oracle_san.so" M17_MIRROR_CHECK="$ROOT/tests/cxx/mirror_check_san"
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" python -m pytest "$ROOT/tests" -x -q -m "not gpu" -p no:cacheprovider "$@"
