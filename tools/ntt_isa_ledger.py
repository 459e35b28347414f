#!/usr/bin/env python3
"""Static instruction ledger of ntt_pass_wl_kernel<10> (the pass of a 2^20 transform) from the gfx950 ISA hipcc emits:
   python tools/ntt_isa_ledger.py [-DZK_NTT_LDS_PLANES]
Classes: products (v_mad_u64_u32 and the Montgomery glue that only products contain: v_mul_lo_u32, v_lshrrev_b64), limb masks / carries / packing
(v_and, v_lshrrev_b32, v_lshl*, v_alignbit, v_or*), additions and subtractions (v_add*, v_sub*: butterflies, borrow constants AND address arithmetic -- the
ISA does not tell them apart; the planes -> element-major change shows up here), selects / moves, LDS, global memory, scalar / control.
The kernel's stage code runs once per lane (four elements per lane and stage pair), its load and store loops four times: static counts, not per element --
the dynamic total per element is the SQ_INSTS_VALU counter (profiles/*_ntt20_pass_sq_pmc.txt)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "ntt_ledger.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "phase2-bn254_amd", "csrc", "ntt.hip")] + sys.argv[1:], stderr=subprocess.DEVNULL)
body, on = [], False
for line in open(out):
    if re.match(r"^_ZN2zk12_GLOBAL__N_118ntt_pass_wl_kernelILj10E.*:", line): on = True
    if on and ".amdhsa_kernel" in line: break
    if on: body.append(line)
ops = collections.Counter(m.group(1) for l in body for m in [re.match(r"^\s+((?:v|s|ds|global|buffer|scratch)_[a-z0-9_]+)", l)] if m)
def cls(op):
    if op in ("v_mad_u64_u32", "v_mul_lo_u32", "v_lshrrev_b64"): return "products (mad + Montgomery glue)"
    if op.startswith(("v_add", "v_sub", "v_xad", "v_mad_u32", "v_mad_i")): return "add / sub (butterflies, constants, addresses)"
    if op.startswith(("v_and", "v_lshr", "v_lshl", "v_alignbit", "v_or", "v_bfe", "v_ashr", "v_bitop", "v_perm")): return "masks / carries / packing"
    if op.startswith(("v_cndmask", "v_mov", "v_cmp", "v_readfirstlane", "v_accvgpr")): return "selects / moves / compares"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_")): return "global memory"
    if op.startswith("s_nop"): return "s_nop (scalar port)"
    if op.startswith("s_"): return "scalar / control"
    return "other VALU"
led = collections.Counter()
for op, c in ops.items(): led[cls(op)] += c
vgpr = next((l.split(",")[-1].strip() for l in open(out) if "ntt_pass_wl_kernelILj10E" in l and ".num_vgpr" in l), "?")
print(f"# ntt_pass_wl_kernel<10>, gfx950, flags {sys.argv[1:] or '(default: element-major LDS tiles)'}: {vgpr} VGPRs, {sum(ops.values())} instructions (static)")
for k, v in led.most_common(): print(f"{v:7d}  {k}")
print("# by opcode:")
for op, c in ops.most_common(28): print(f"{c:7d}  {op}")
