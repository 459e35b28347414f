#!/usr/bin/env python3
"""Prover-like exponents at size: a Groth16 witness is full of 0 and 1 (and small values), so window 0 sends a large share of
the points to a handful of buckets.  Times the device-resident G1 multiexp on such a vector next to the uniform one, with the
per-phase kernel times, and checks additivity MSM(s) == MSM(a) + MSM(s - a)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=24); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
k = bench.gen_scalars(n, 5, dev)
bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
uniform = bench.gen_scalars(n, 6, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
kind = torch.randint(0, 10, (n,), device=dev, generator=g)
skew = uniform.clone()
skew[kind < 4] = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=dev)          # 40 % ones
skew[(kind >= 4) & (kind < 7)] = 0                                                   # 30 % zeros
small = torch.randint(0, 256, (n,), device=dev, generator=g, dtype=torch.int64)
m8 = kind == 7
skew[m8] = torch.stack([small, torch.zeros_like(small), torch.zeros_like(small), torch.zeros_like(small)], dim=1)[m8]   # 10 % bytes
out = {"log_n": a.log_n}
for name, sc in (("uniform", uniform), ("prover_like", skew)):
    zk.multiexp(w, (bases, 0), zk.FullDensity(), sc).wait()
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(a.iters): r = zk.multiexp(w, (bases, 0), zk.FullDensity(), sc).wait()
    dt = (time.perf_counter() - t) / a.iters
    L.mi355zk_prof_enable(0)
    kern = {}
    for kn in ("msm_digits", "msm_scatter", "msm_bucket", "msm_sort", "msm_accumulate_heavy", "msm_accumulate", "msm_reduce"):
        ms, cnt = C.c_double(), C.c_long(); L.mi355zk_prof_get(kn.encode(), C.byref(ms), C.byref(cnt))
        kern[kn] = round(ms.value / max(cnt.value, 1), 3)
    out[name] = {"ms": round(dt * 1e3, 3), "Mscalar_mul_per_s": round(n / dt / 1e6, 1), "kernel_ms": kern}
a_ = bench.gen_scalars(n, 8, dev); b_ = skew.clone()
assert L.mi355zk_bn254_fr_sub_assign_dev(C.c_void_p(b_.data_ptr()), C.c_void_p(a_.data_ptr()), n, None) == 0
torch.cuda.synchronize()
import oracle_lib as O
tot = zk.multiexp(w, (bases, 0), zk.FullDensity(), skew).wait()
pa, pb = zk.multiexp(w, (bases, 0), zk.FullDensity(), a_).wait(), zk.multiexp(w, (bases, 0), zk.FullDensity(), b_).wait()
out["additivity_ok"] = bool(np.array_equal(O.G1.to_affine(zk.shard.join_partials(np.stack([pa, pb]))), O.G1.to_affine(tot)))
print(json.dumps(out))
