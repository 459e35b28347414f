#!/usr/bin/env python3
"""The streamed host-buffer G1 multiexp at 2^26 (pinned bases on the device, page-locked exponents crossing PCIe inside the call -- SURVEY 8d's
own definition of the metric) under different chunk schedules: MI355ZK_HOST_CHUNK_FIRST (log2 of the first chunk) x MI355ZK_HOST_CHUNK_GROWTH
(percent), and equal chunks (MI355ZK_HOST_CHUNK_TEST).  Per configuration: wall time per call and the per-group kernel time the library's
HIP-event hooks saw (sum over the chunks of a call)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
log_n = int(os.environ.get("EXP_LOG_N", "26"))
n = 1 << log_n
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
sh = min(n, 1 << 22)
for s in range(n // sh):
    k = bench.gen_scalars(sh, 50 + s, dev)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b[s * sh:(s + 1) * sh].data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), sh, None) == 0
sc = torch.cat([bench.gen_scalars(sh, 90 + s, dev) for s in range(n // sh)])
res = zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
t = time.perf_counter()
for _ in range(3): zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
resident_ms = (time.perf_counter() - t) / 3 * 1e3
hb = b.cpu().numpy().view(np.uint64)
hs_t = torch.empty(sc.shape, dtype=torch.int64, pin_memory=True); hs_t.copy_(sc); torch.cuda.synchronize()
hs = hs_t.numpy().view(np.uint64)
del b, sc
zk.pin_bases(hb)
zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
import oracle_lib as O
want = O.G1.to_affine(res)
out = {"log_n": log_n, "resident_ms": round(resident_ms, 2), "configs": []}
configs = [a.split(",") for a in sys.argv[1:]] or [["", "", ""]]
names = ("msm_digits", "msm_sort", "msm_part_scan", "msm_scatter", "msm_bucket", "msm_accumulate_heavy", "msm_accumulate", "msm_reduce")
for first, grow, test in configs:
    for kname, v in (("MI355ZK_HOST_CHUNK_FIRST", first), ("MI355ZK_HOST_CHUNK_GROWTH", grow), ("MI355ZK_HOST_CHUNK_TEST", test)):
        if v: os.environ[kname] = v
        else: os.environ.pop(kname, None)
    got = zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
    ok = bool(np.array_equal(O.G1.to_affine(got), want))
    t = time.perf_counter()
    for _ in range(4): zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
    ms = (time.perf_counter() - t) / 4 * 1e3
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    for _ in range(2): zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
    L.mi355zk_prof_enable(0)
    kern = {}
    for name in names:
        tot, cnt = C.c_double(), C.c_long()
        L.mi355zk_prof_get(name.encode(), C.byref(tot), C.byref(cnt))
        kern[name] = [round(tot.value / 2, 3), cnt.value // 2]   # ms per call, launches per call
    out["configs"].append({"first_log": first, "growth_pct": grow, "equal_chunks_of": test, "ms_per_call": round(ms, 2), "same_point": ok, "kernel_ms_per_call": kern})
    print(json.dumps(out["configs"][-1]), file=sys.stderr, flush=True)
print(json.dumps(out))
