#!/usr/bin/env python3
"""What one rank of an N-GPU run does, timed alone on one MI355X: the cell (point range x window group) shard.plan(N) gives
rank 0 of a 2^log_n-point G1 MSM, next to a plain 1/N point range -- the strong-scaling ceiling of bench.py --gpus N before
any communication (the exchange is one all-gather of 96-byte partials)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=26); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
k = bench.gen_scalars(n, 5, dev); s = bench.gen_scalars(n, 6, dev)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
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


full = timed(lambda: zk.multiexp(w, (b, 0), zk.FullDensity(), s).wait())
out = {"log_n": a.log_n, "one_gpu_ms": round(full, 2)}
for world in (2, 4, 8):
    pg, wg = zk.shard.plan(world); nl = n // pg
    # every window group is timed; the slowest one is the rank the others wait for (the groups differ in their digit chains and in
    # the top window's short digits)
    cells = {g: timed(lambda g=g: zk.multiexp(w, (b[:nl], 0), zk.FullDensity(), s[:nl], window_group=(wg, g)).wait()) for g in range(wg)}
    slow = max(cells, key=cells.get)
    cell_fn = lambda: zk.multiexp(w, (b[:nl], 0), zk.FullDensity(), s[:nl], window_group=(wg, slow)).wait()
    cell = cells[slow]
    pts = timed(lambda: zk.multiexp(w, (b[:n // world], 0), zk.FullDensity(), s[:n // world]).wait())
    nw = C.c_int(); c = L.mi355zk_msm_window_bits_groups(nl, wg, C.byref(nw))
    out[f"n{world}"] = {"plan": f"{pg} point range(s) x {wg} window group(s)", "windows": nw.value, "field_bits": c, "cell_ms": round(cell, 2), "cell_ms_by_group": [round(cells[g], 2) for g in range(wg)],
                        "speedup": round(full / cell, 2), "efficiency": round(full / cell / world, 3), "cell_kernel_ms": kernels(cell_fn),
                        "point_ranges_only_ms": round(pts, 2), "point_ranges_only_efficiency": round(full / pts / world, 3)}
print(json.dumps(out))
