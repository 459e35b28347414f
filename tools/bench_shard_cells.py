import sys, os, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device('cuda', 0)
log_n = 26; n = 1 << log_n
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
k = bench.gen_scalars(n, 5, dev); s = bench.gen_scalars(n, 6, dev)
b = torch.empty((n, 8), dtype=torch.int64, device=dev)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
torch.cuda.synchronize(); del k
def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
full = t(lambda: zk.multiexp(w, (b, 0), zk.FullDensity(), s).wait())
print("1 GPU full 2^26: %.2f ms" % full)
for world in (2, 4, 8):
    pg, wg = zk.shard.plan(world)
    nl = n // pg
    cell = t(lambda: zk.multiexp(w, (b[:nl], 0), zk.FullDensity(), s[:nl], window_group=(wg, 0)).wait())
    pts = t(lambda: zk.multiexp(w, (b[:n // world], 0), zk.FullDensity(), s[:n // world]).wait())
    nw = C.c_int(); c = L.mi355zk_msm_window_bits_groups(nl, wg, C.byref(nw))
    print("world %d: plan %dx%d (field %d bits, W=%d): cell %.2f ms -> speedup %.2f (%.0f%%);  point-range only: %.2f ms -> %.2f (%.0f%%)" % (world, pg, wg, c, nw.value, cell, full / cell, 100 * full / cell / world, pts, full / pts, 100 * full / pts / world))
for world in (4, 8):
    pg, wg = zk.shard.plan(world); nl = n // pg
    L.mi355zk_prof_reset(); L.mi355zk_prof_enable(1)
    for _ in range(3): zk.multiexp(w, (b[:nl], 0), zk.FullDensity(), s[:nl], window_group=(wg, 0)).wait()
    L.mi355zk_prof_enable(0)
    kern = {}
    for name in ("msm_digits", "msm_sort", "msm_accumulate_heavy", "msm_accumulate", "msm_reduce"):
        ms, cnt = C.c_double(), C.c_long(); L.mi355zk_prof_get(name.encode(), C.byref(ms), C.byref(cnt)); kern[name] = round(ms.value / max(cnt.value, 1), 3)
    print("world", world, "cell kernels:", kern, "sum", round(sum(kern.values()), 2))
