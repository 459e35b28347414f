#!/usr/bin/env python3
"""Gate for precomputed window tables with ONE shared bucket set, measured on the existing kernels:
a multiexp over n' = W * n points evaluated for ONE window group out of W (the lowest window: full-width signed digits) is what
a table-mode call over n points would run after its digit kernel -- W * n (digit, table index) pairs into ONE set of 2^(c-1)
buckets, one window's reduction, a one-term join.  Run once per forced window width:
    MI355ZK_MSM_C=c python tools/exp_flat_table.py --log-n L
prints the plain call (the library's own geometry would need MI355ZK_MSM_C unset; here the forced c) and the one-window cell."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n
nw = C.c_int(); c = L.mi355zk_msm_window_bits_groups(n, 1, C.byref(nw)); W = nw.value
nn = W * n
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
k = bench.gen_scalars(nn, 5, dev); s = bench.gen_scalars(nn, 6, dev)
b = torch.empty((nn, 8), dtype=torch.int64, device=dev)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), nn, None) == 0
torch.cuda.synchronize(); del k


def timed(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters * 1e3


def kernels(fn):
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    for _ in range(a.iters): fn()
    L.mi355zk_prof_enable(0)
    out = {}
    for name in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce"):
        ms, cnt = C.c_double(), C.c_long(); L.mi355zk_prof_get(name.encode(), C.byref(ms), C.byref(cnt)); out[name] = round(ms.value / max(cnt.value, 1), 3)
    return out


plain_fn = lambda: zk.multiexp(w, (b[:n], 0), zk.FullDensity(), s[:n]).wait()
plain = timed(plain_fn)
# the W-times longer vector has its own geometry unless c is forced: report what the library used
nw2 = C.c_int(); c2 = L.mi355zk_msm_window_bits_groups(nn, 1, C.byref(nw2))
out = {"log_n": a.log_n, "c": c, "windows": W, "plain_ms": round(plain, 3), "plain_kernel_ms": kernels(plain_fn)}
if nw2.value == W:
    cell_fn = lambda: zk.multiexp(w, (b, 0), zk.FullDensity(), s, window_group=(W, 0)).wait()
    cell = timed(cell_fn)
    out.update({"flat_pairs": nn, "flat_one_window_ms": round(cell, 3), "flat_kernel_ms": kernels(cell_fn)})
else:
    out["note"] = f"geometry differs for the long vector (c={c2}, W={nw2.value}): force MI355ZK_MSM_C"
print(json.dumps(out), flush=True)
