#!/usr/bin/env python3
"""Experiment: how fast does a small-footprint memory-bound stream (torch tensor copies: 256-lane workgroups, few VGPRs, no LDS) run BESIDE the bucket accumulation of a
2^26 multiexp, and what does it cost the multiexp -- with the shipped accumulation (four waves per SIMD: 500 of a SIMD's 512 VGPRs) and with the three-wave variant
(MI355ZK_SO=tools/bin/libmi355zk_w3.so: 128 VGPRs per SIMD left free).  The question behind it: can the partition of the next window group hide under the accumulation."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << 26
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
sh = 1 << 22
for s in range(n // sh):
    k = bench.gen_scalars(sh, 50 + s, dev)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b[s * sh:(s + 1) * sh].data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), sh, None) == 0
sc = torch.cat([bench.gen_scalars(sh, 90 + s, dev) for s in range(n // sh)])
src = torch.empty(1 << 29, dtype=torch.int64, device=dev); dst = torch.empty_like(src)   # 4 GiB each
torch.cuda.synchronize()
def msm_loop(reps, out):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
        t = time.perf_counter()
        for _ in range(reps): zk.multiexp(w, (b, 0), zk.FullDensity(), sc).wait()
        out["msm_ms"] = (time.perf_counter() - t) / reps * 1e3
stop = False
def copy_loop(out):
    st = torch.cuda.Stream()
    cnt = 0
    with torch.cuda.stream(st):
        t = time.perf_counter()
        while not stop:
            dst.copy_(src); st.synchronize(); cnt += 1
        out["copy_GBs"] = cnt * 2 * src.numel() * 8 / (time.perf_counter() - t) / 1e9
        out["copies"] = cnt
res = {}
o = {}; msm_loop(5, o); res["msm_alone_ms"] = round(o["msm_ms"], 2)
o = {}; stop = False
th = threading.Thread(target=copy_loop, args=(o,)); th.start(); time.sleep(1.0); stop = True; th.join(); res["copy_alone_GBs"] = round(o["copy_GBs"])
o1, o2 = {}, {}; stop = False
th = threading.Thread(target=copy_loop, args=(o2,)); th.start()
msm_loop(8, o1); stop = True; th.join()
res["msm_beside_copy_ms"] = round(o1["msm_ms"], 2); res["copy_beside_msm_GBs"] = round(o2["copy_GBs"]); res["copies"] = o2["copies"]
res["lib"] = os.environ.get("MI355ZK_SO", "default")
print(json.dumps(res))
