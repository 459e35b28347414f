#!/usr/bin/env python3
"""Four 2^20-point G1 multiexps on device-resident inputs, nothing else: run under `rocprofv3 --kernel-trace` and read the launch
order / durations of one call with `tools/rocpd_summary.py <db> --timeline 27` (profiles/*_msm20_timeline.txt)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes as C
import phase2_bn254_amd as zk, inputs, bench
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("TRACE_LOG_N", "20"))
k = bench.gen_scalars(n, 5, dev); s = bench.gen_scalars(n, 6, dev)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
for _ in range(4): zk.multiexp(w, (b, 0), zk.FullDensity(), s).wait()
