#!/bin/bash
# tools/run_ubsan.sh <command ...>: run a python command over the UBSan build of the library (make ubsan): UndefinedBehaviorSanitizer on the HOST side of
# every translation unit, no interceptors -- the build that can run WITH device work (tools/run_asan.sh cannot: Makefile).  -fno-sanitize-recover: the
# first report aborts the process.
here="$(cd "$(dirname "$0")/.." && pwd)"
[ -f "$here/tools/bin/libmi355zk_ubsan.so" ] || { echo "tools/bin/libmi355zk_ubsan.so is missing: make ubsan" >&2; exit 2; }
rt=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)")
export LD_LIBRARY_PATH="$rt${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98${UBSAN_OPTIONS:+:$UBSAN_OPTIONS}"
export MI355ZK_SO="$here/tools/bin/libmi355zk_ubsan.so"
exec "$@"
