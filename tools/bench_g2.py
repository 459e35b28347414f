#!/usr/bin/env python3
"""Secondary metric: 2^log_n-point BN254 G2 MSM on one MI355X (bases k_i*G2 generated on the device)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n
scalars = bench.gen_scalars(n, 11, dev); k = bench.gen_scalars(n, 12, dev)
bases = torch.empty((n, 16), dtype=torch.int64, device=dev)
gen = np.ascontiguousarray(inputs.G2_GEN_RAW)
t = time.time()
assert L.mi355zk_bn254_g2_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
torch.cuda.synchronize(); t_gen = time.time() - t
zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
t = time.perf_counter()
for _ in range(a.iters): res = zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
dt = (time.perf_counter() - t) / a.iters
L.mi355zk_prof_enable(0)
kern = {}
for name in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce"):
    ms, cnt = C.c_double(), C.c_long(); L.mi355zk_prof_get(name.encode(), C.byref(ms), C.byref(cnt)); kern[name] = round(ms.value / max(cnt.value, 1), 4)
# closed form check: sum s_i k_i * G2
import bn254_model as M, oracle_lib as O
hs = scalars.cpu().numpy().view(np.uint64); hk = k.cpu().numpy().view(np.uint64)
to_int = lambda x: sum(x[:, i].astype(object) << (64 * i) for i in range(4))
dot = int(sum(s * kk for s, kk in zip(to_int(hs), to_int(hk))) % M.R_ORDER)
want = O.G2.mul(O.G2.from_affine(inputs.G2_GEN_RAW), M.to_limbs(dot))
ok = bool(np.array_equal(O.G2.to_affine(res), O.G2.to_affine(want)))
print(json.dumps({"g2_log_n": a.log_n, "ms": round(dt * 1e3, 3), "Mscalar_mul_per_s": round(n / dt / 1e6, 2), "kernel_ms": kern, "input_gen_s": round(t_gen, 2), "matches_closed_form": ok}))
