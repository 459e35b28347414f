#!/bin/bash
# tools/run_tsan.sh <command ...>: run a python command over the ThreadSanitizer build of the library (make tsan): the host side of every translation unit
# instrumented, the gfx950 code objects unchanged; works WITH device work (~10 x slower).  Reports go to $TSAN_LOG.<pid> (default gpurun_out/tsanlog.<pid>);
# races inside the uninstrumented HIP runtime / torch are suppressed (tools/tsan.supp).  The process exit code is the command's: read the log.
here="$(cd "$(dirname "$0")/.." && pwd)"
[ -f "$here/tools/bin/libmi355zk_tsan.so" ] || { echo "tools/bin/libmi355zk_tsan.so is missing: make tsan" >&2; exit 2; }
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
mkdir -p "$here/gpurun_out"
# (dlopen through the sanitizer's interceptor resolves $ORIGIN against the sanitizer runtime, not the caller: torch's lazy `libcaffe2_nvrtc.so` is then not found)
tl=$(python3 -c 'import importlib.util as u; print(u.find_spec("torch").submodule_search_locations[0] + "/lib")' 2>/dev/null)
[ -n "$tl" ] && export LD_LIBRARY_PATH="$tl${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
export LD_PRELOAD="$rt${LD_PRELOAD:+:$LD_PRELOAD}"
export TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0:suppressions=$here/tools/tsan.supp:log_path=${TSAN_LOG:-$here/gpurun_out/tsanlog}${TSAN_OPTIONS:+:$TSAN_OPTIONS}"
export MI355ZK_SO="$here/tools/bin/libmi355zk_tsan.so"
exec "$@"
