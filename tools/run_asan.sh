#!/bin/bash
# tools/run_asan.sh <command ...>: run a python command over the sanitizer build of the library (make asan): AddressSanitizer + UBSan on the HOST side
# of every translation unit, the gfx950 code objects unchanged.  halt_on_error: the first report ends the process with a non-zero code.
here="$(cd "$(dirname "$0")/.." && pwd)"
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$here/tools/bin/libmi355zk_asan.so" ] || { echo "tools/bin/libmi355zk_asan.so is missing: make asan" >&2; exit 2; }
export LD_PRELOAD="$rt${LD_PRELOAD:+:$LD_PRELOAD}"
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0:exitcode=97${ASAN_OPTIONS:+:$ASAN_OPTIONS}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98${UBSAN_OPTIONS:+:$UBSAN_OPTIONS}"
export MI355ZK_SO="$here/tools/bin/libmi355zk_asan.so"
exec "$@"
