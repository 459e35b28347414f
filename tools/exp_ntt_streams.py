#!/usr/bin/env python3
"""Experiment: K independent 2^log_n transforms queued on ONE stream against the same K on K streams (their workgroups then interleave on the CUs:
do the load / compute / store phases of different transforms overlap?).   python tools/exp_ntt_streams.py [--log-n 20] [--k 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs
ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--k", type=int, default=3); ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
w = zk.Worker(0); n = 1 << a.log_n
host = inputs.random_fr_mont(n, seed=5)
doms = [zk.EvaluationDomain(torch.from_numpy(host.view(np.int64)).cuda().clone(), a.log_n) for _ in range(a.k)]
streams = [torch.cuda.Stream() for _ in range(a.k)]
res = {"log_n": a.log_n, "k": a.k}
for op in ("fft", "coset_fft"):
    def one_stream():
        for d in doms: getattr(d, op)(w)
    def k_streams():
        for d, s in zip(doms, streams):
            with torch.cuda.stream(s): getattr(d, op)(w)
    for name, fn in (("one_stream", one_stream), ("k_streams", k_streams)):
        t_warm = time.perf_counter() + 0.06
        while time.perf_counter() < t_warm: fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.iters): fn()
        torch.cuda.synchronize()
        res[f"{op}_{name}_ms_per_transform"] = round((time.perf_counter() - t) / a.iters / a.k * 1e3, 4)
print(json.dumps(res))
