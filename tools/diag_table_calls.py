#!/usr/bin/env python3
"""Per-call wall times of plain and table-mode multiexps (2^20 points), first call included: where a leg's cold start ends and whether
single calls stall (BENCH_r04 / r05: the G1 table leg after ONE warm-up call, a 38-64 ms call inside the G2 table leg)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << 20
out = {}
big = torch.empty(int(os.environ.get("DIAG_PREALLOC_GB", "8")) << 27, dtype=torch.int64, device=dev); del big   # the allocator state bench.py leaves: GiBs cached
for g, limbs, gen in ((1, 8, inputs.G1_GEN_RAW), (2, 16, inputs.G2_GEN_RAW)):
    k = bench.gen_scalars(n, 21 + g, dev); sc = bench.gen_scalars(n, 11 + g, dev)
    b = torch.empty((n, limbs), dtype=torch.int64, device=dev)
    mul = L.mi355zk_bn254_g1_batch_mul_dev if g == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert mul(C.c_void_p(b.data_ptr()), np.ascontiguousarray(gen).ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    def run(src, count):
        ts = []
        for _ in range(count):
            t = time.perf_counter(); zk.multiexp(w, (src, 0), zk.FullDensity(), sc).wait(); ts.append(round((time.perf_counter() - t) * 1e3, 3))
        return ts
    out[f"g{g}_plain"] = run(b, 40)
    t = time.perf_counter(); tb = zk.MsmTable(b); torch.cuda.synchronize(); out[f"g{g}_table_build_ms"] = round((time.perf_counter() - t) * 1e3, 2)
    out[f"g{g}_table"] = run(tb, 60)
    del tb
print(json.dumps(out))
