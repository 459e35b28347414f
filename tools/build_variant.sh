#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" [unit ...]: a second build of the library for a same-box A/B run -- the listed translation
# units (default: msm_g1) recompiled with the extra flags (in parallel), linked with the other objects of build/ into tools/bin/libmi355zk_NAME.so
# (git-ignored, travels with gpurun).  LINKFLAGS: extra flags of the link.  Use: MI355ZK_SO=tools/bin/libmi355zk_NAME.so python tools/...
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2 || true
units=${@:-msm_g1}
mkdir -p build_$name tools/bin
objs=""; pids=""
for o in ntt msm_g1 msm_g2 api host_entry scalar_mul field_ops point_fft point_fft_g2 codec; do
  if [[ " $units " == *" $o "* ]]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result $flags -c phase2-bn254_amd/csrc/$o.hip -o build_$name/$o.o &
    pids="$pids $!"
    objs="$objs build_$name/$o.o"
  else
    objs="$objs build/$o.o"
  fi
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $LINKFLAGS -o tools/bin/libmi355zk_$name.so $objs
echo tools/bin/libmi355zk_$name.so
