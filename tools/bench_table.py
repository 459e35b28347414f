#!/usr/bin/env python3
"""Table mode (mi355zk_bn254_g{1,2}_msm_table_dev) against the plain device-resident multiexp on one MI355X: a 2^log_n-point vector,
uniform exponents, everything resident.  MI355ZK_MSM_TABLE_C=c forces the table's window width (the sweep behind table_window_bits)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench, oracle_lib as O

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--group", type=int, default=1)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n; grp = a.group
G = O.G1 if grp == 1 else O.G2
gen = np.ascontiguousarray(inputs.G1_GEN_RAW if grp == 1 else inputs.G2_GEN_RAW)
k = bench.gen_scalars(n, 5, dev); s = bench.gen_scalars(n, 6, dev)
b = torch.empty((n, 8 * grp), dtype=torch.int64, device=dev)
fn = L.mi355zk_bn254_g1_batch_mul_dev if grp == 1 else L.mi355zk_bn254_g2_batch_mul_dev
assert fn(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
torch.cuda.synchronize(); del k
t0 = time.perf_counter(); t = zk.MsmTable(b); torch.cuda.synchronize(); build_ms = (time.perf_counter() - t0) * 1e3


def timed(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters * 1e3, r


def kernels(fn):
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    for _ in range(a.iters): fn()
    L.mi355zk_prof_enable(0)
    out = {}
    for name in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce"):
        ms, cnt = C.c_double(), C.c_long(); L.mi355zk_prof_get(name.encode(), C.byref(ms), C.byref(cnt)); out[name] = round(ms.value / max(cnt.value, 1), 3)
    return out


plain_fn = lambda: zk.multiexp(w, (b, 0), zk.FullDensity(), s).wait()
table_fn = lambda: zk.multiexp(w, (t, 0), zk.FullDensity(), s).wait()
pm, pr = timed(plain_fn); tm, tr = timed(table_fn)
print(json.dumps({"group": grp, "log_n": a.log_n, "plain_ms": round(pm, 3), "table_ms": round(tm, 3), "speedup": round(pm / tm, 3),
                  "table_window_bits": t.window_bits, "table_windows": t.n_windows, "table_MB": round(t.table.numel() * 8 / 2**20, 1),
                  "table_build_ms": round(build_ms, 1), "plain_kernel_ms": kernels(plain_fn), "table_kernel_ms": kernels(table_fn),
                  "same_point": bool(np.array_equal(G.to_affine(pr), G.to_affine(tr)))}), flush=True)
