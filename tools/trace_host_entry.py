#!/usr/bin/env python3
"""Three streamed host-buffer G1 multiexps at 2^TRACE_LOG_N (default 26) over a pinned base vector, nothing else: run under
`rocprofv3 --kernel-trace` and read the launches of the last call with `tools/rocpd_summary.py <db> --timeline N`
(what every chunk of the streamed call costs: profiles/r04_host_entry_timeline.txt)."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes as C
import phase2_bn254_amd as zk, inputs, bench
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("TRACE_LOG_N", "26"))
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
sh = min(n, 1 << 22)
for s in range(n // sh):
    k = bench.gen_scalars(sh, 50 + s, dev)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b[s * sh:(s + 1) * sh].data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), sh, None) == 0
sc = torch.cat([bench.gen_scalars(sh, 90 + s, dev) for s in range(n // sh)])
hb = b.cpu().numpy().view(np.uint64)
hs_t = torch.empty(sc.shape, dtype=torch.int64, pin_memory=True); hs_t.copy_(sc); torch.cuda.synchronize()
hs = hs_t.numpy().view(np.uint64)
del b, sc
zk.pin_bases(hb)
for _ in range(3): zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
