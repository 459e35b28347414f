#!/usr/bin/env python3
"""Static instruction ledger of one kernel from the gfx950 ISA hipcc emits:
   python tools/isa_ledger.py phase2-bn254_amd/csrc/msm_g1.hip 'msm_accumulate_kernelINS_2FpINS_8FqParams' [hipcc flags]
Classes as in tools/ntt_isa_ledger.py.  Static counts (loops count once): the dynamic totals are the SQ_INSTS_* counters."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
out = os.path.join(tempfile.gettempdir(), "isa_ledger_%s.s" % os.path.basename(src))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, src)] + sys.argv[3:], stderr=subprocess.DEVNULL)
names = sorted({m.group(1) for l in open(out) for m in [re.match(r"^(_Z\w+):", l)] if m and re.search(pat, m.group(1))})
for name in names:
    body, on = [], False
    for line in open(out):
        if line.startswith(name + ":"): on = True
        elif on and (".amdhsa_kernel" in line or re.match(r"^_Z\w+:", line)): break
        if on: body.append(line)
    ops = collections.Counter(m.group(1) for l in body for m in [re.match(r"^\s+((?:v|s|ds|global|buffer|scratch|flat)_[a-z0-9_]+)", l)] if m)
    def cls(op):
        if op in ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32"): return "multiplier (v_mad_u64_u32, v_mul_lo/hi_u32)"
        if op.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64")): return "64-bit shifts / adds"
        if op.startswith(("v_add", "v_sub", "v_xad", "v_mad_u32", "v_mad_i")): return "add / sub"
        if op.startswith(("v_and", "v_lshr", "v_lshl", "v_alignbit", "v_or", "v_bfe", "v_ashr", "v_bitop", "v_perm", "v_xor", "v_not", "v_bfi")): return "masks / shifts / logic"
        if op.startswith(("v_cndmask", "v_mov", "v_cmp", "v_readfirstlane", "v_accvgpr", "v_readlane", "v_writelane")): return "selects / moves / compares"
        if op.startswith("ds_"): return "LDS"
        if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "memory"
        if op.startswith("s_nop"): return "s_nop (scalar port)"
        if op.startswith("s_"): return "scalar / control"
        return "other VALU"
    led = collections.Counter()
    for op, c in ops.items(): led[cls(op)] += c
    res = {k: next((l.split(",")[-1].strip() for l in open(out) if name in l and "." + k in l), "?") for k in ("num_vgpr", "num_agpr", "private_seg_size")}
    print(f"# {name[:110]}: {res['num_vgpr']} VGPRs, {res['num_agpr']} AGPRs, {res['private_seg_size']} B scratch, {sum(ops.values())} instructions (static)")
    for k, v in led.most_common(): print(f"{v:7d}  {k}")
    print("# by opcode:")
    for op, c in ops.most_common(30): print(f"{c:7d}  {op}")
    print()
