#!/bin/bash
# Host-side sanitizer run (CPU box only; never on a GPU box: GPU ASan / xnack+ is refused on this pool).
# Builds libvqhip_asan.so (host code of vq_amd/csrc under ASan + UBSan, device code untouched) and
# libvq_oracle_asan.so with the same clang runtime, then runs the CPU test suite against them.
#   tools/run_asan.sh [pytest args]        default: tests -m "not gpu" -x -q
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CLANG=/opt/rocm/lib/llvm/bin/clang
make -C "$ROOT/vq_amd/csrc" -j"$(nproc)" asan
make -C "$ROOT/oracle" asan
RT="$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)"
export VQHIP_LIB_PATH="$ROOT/vq_amd/libvqhip_asan.so"
export VQ_ORACLE_LIB="$ROOT/oracle/libvq_oracle_asan.so"
# CPython leaks by design and installs its own SIGSEGV handling; OpenMP's thread stacks are fine under ASan
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1:handle_segv=0"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}"
cd "$ROOT"
if [ $# -eq 0 ]; then set -- tests -m "not gpu" -x -q; fi
exec python -m pytest "$@"
