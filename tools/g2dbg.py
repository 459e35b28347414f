import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, ctypes as C
import phase2_bn254_amd as zk, inputs, oracle_lib as O
w = zk.Worker(0)
L = zk.lib.load()
L.mi355zk_prof_enable(1)
for n in (1, 33):
    bases = inputs.bases_progression_cpu(2, n, seed=n); scalars = inputs.random_scalars(n, seed=11*n+1)
    t=time.time(); got = zk.multiexp(w, (bases,0), zk.FullDensity(), scalars).wait(); dt=time.time()-t
    rc, want = O.G2.multiexp(bases, scalars)
    print("n",n,"time",dt,"ok",np.array_equal(O.G2.to_affine(got), O.G2.to_affine(want)), flush=True)
    for k in (b"msm_digits", b"msm_sort", b"msm_accumulate", b"msm_reduce"):
        ms=C.c_double(); cnt=C.c_long(); L.mi355zk_prof_get(k, C.byref(ms), C.byref(cnt)); print("  ",k,ms.value,cnt.value, flush=True)
