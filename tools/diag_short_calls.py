#!/usr/bin/env python3
"""Where a SHORT multiexp spends its wall time: per-call wall (no events), the kernel groups' HIP-event times of the same call shape, and the host join
(MI355ZK_TRACE_MSM prints it per call).  G1 / G2 at 2^12, 2^14, 2^16."""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if "--child" in sys.argv:
    import numpy as np, torch
    import phase2_bn254_amd as zk, inputs, bench
    L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
    g, ln = int(sys.argv[2]), int(sys.argv[3])
    n = 1 << ln
    k = bench.gen_scalars(n, 21 + g, dev); sc = bench.gen_scalars(n, 11 + g, dev)
    b = torch.empty((n, 8 * g), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if g == 1 else inputs.G2_GEN_RAW)
    mul = L.mi355zk_bn254_g1_batch_mul_dev if g == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert mul(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    for _ in range(60): zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
    t = time.perf_counter()
    for _ in range(200): zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
    wall = (time.perf_counter() - t) / 200
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    for _ in range(50): zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
    L.mi355zk_prof_enable(0)
    kern = bench._prof(L, ("msm_digits", "msm_sort", "msm_accumulate_heavy", "msm_accumulate", "msm_reduce"))
    print(json.dumps({"group": g, "log_n": ln, "wall_ms": round(wall * 1e3, 4), "kernel_ms": {a: round(v, 4) for a, v in kern.items() if v is not None},
                      "kernel_sum_ms": round(sum(v for v in kern.values() if v is not None), 4)}))
    sys.exit(0)
for g in (1, 2):
    for ln in (12, 14, 16):
        env = dict(os.environ, MI355ZK_TRACE_MSM="1")
        r = subprocess.run([sys.executable, __file__, "--child", str(g), str(ln)], capture_output=True, text=True, env=env)
        joins = [float(l.split(":")[-1].split("us")[0]) for l in r.stderr.splitlines() if "host join" in l]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
        if joins: d["host_join_us_median"] = sorted(joins)[len(joins) // 2]
        print(json.dumps(d))
