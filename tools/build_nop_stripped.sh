#!/bin/bash
# EXPERIMENT (profiles/r05_snop_ab.txt): how much do the `s_nop 0` cost that hipcc's hazard recogniser puts after every empty asm statement of
# ZK_CHAIN_MAD (fieldu.hpp u_mad: the pin that keeps a column sum ONE dependent v_mad_u64_u32 chain)?  Builds a translation unit the way hipcc
# does -- device code to assembly, assembled, linked, bundled, embedded into the host object -- with those s_nop deleted from the assembly in
# between, and links a second library next to the product one:
#     tools/build_nop_stripped.sh ntt msm_g1     ->  tools/bin/libmi355zk_nonop.so   (MI355ZK_SO=... selects it for an A/B)
# An `s_nop 0` is deleted only where it sits between an `;;#ASMEND` and a following instruction with NO asm text in the block (the pins are
# empty); the hazard the recogniser guards against is whatever an asm statement might contain -- here nothing: the instruction stream is
# mad -> mad on one accumulator, which hipcc itself emits without padding wherever it builds such a chain unprompted.
set -e
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
OUT=build/nonop
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result"
OBJS=""
for tu in ntt msm_g1 msm_g2 api field_ops point_fft point_fft_g2 codec; do
  if [[ " $* " == *" $tu "* ]]; then
    hipcc $FLAGS --cuda-device-only -S phase2-bn254_amd/csrc/$tu.hip -o $OUT/$tu.s 2>/dev/null
    python3 - $OUT/$tu.s $OUT/${tu}_stripped.s <<'PY'
import re, sys
src = open(sys.argv[1]).read().splitlines()
out, dropped, i = [], 0, 0
while i < len(src):
    line = src[i]
    if line.strip() == "s_nop 0":
        # look back over blank / comment lines: directly after an EMPTY asm statement?
        j = len(out) - 1
        while j >= 0 and out[j].strip() == "": j -= 1
        if j >= 1 and out[j].strip() == ";;#ASMEND" and out[j - 1].strip() == ";;#ASMSTART":
            dropped += 1
            i += 1
            continue
    out.append(line)
    i += 1
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print(f"{sys.argv[1]}: {dropped} s_nop 0 removed", file=sys.stderr)
PY
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/${tu}_stripped.s -o $OUT/${tu}_dev.o
    $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $OUT/$tu.hsaco $OUT/${tu}_dev.o
    $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$OUT/$tu.hsaco -output=$OUT/$tu.hipfb
    hipcc $FLAGS --cuda-host-only -c phase2-bn254_amd/csrc/$tu.hip -Xclang -fcuda-include-gpubinary -Xclang $OUT/$tu.hipfb -o $OUT/$tu.o
    OBJS="$OBJS $OUT/$tu.o"
  else
    OBJS="$OBJS build/$tu.o"
  fi
done
mkdir -p tools/bin; hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libmi355zk_nonop.so $OBJS
ls -la tools/bin/libmi355zk_nonop.so
