#!/usr/bin/env python3
"""2^log_n Fr transforms in batches (mi355zk_bn254_fr_domain_op_batch_dev, one launch per pass over all the arrays) against one at a time:
ms per transform.   python tools/bench_ntt_batch.py [--log-n 20] [--batches 1 2 3 4 8]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs
ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 3, 4, 8]); ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
w = zk.Worker(0); n = 1 << a.log_n
host = inputs.random_fr_mont(n, seed=5)
res = {"log_n": a.log_n, "unit": "ms per transform"}
for k in a.batches:
    doms = [zk.EvaluationDomain(torch.from_numpy(host.view(np.int64)).cuda().clone(), a.log_n) for _ in range(k)]
    for op in ("fft", "ifft", "coset_fft", "icoset_fft"):
        fn = getattr(zk.EvaluationDomain, op + "_many")
        t_warm = time.perf_counter() + 0.06
        while time.perf_counter() < t_warm: fn(w, doms)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.iters): fn(w, doms)
        torch.cuda.synchronize()
        res.setdefault(op, {})[f"batch_{k}"] = round((time.perf_counter() - t) / a.iters / k * 1e3, 4)
print(json.dumps(res))
