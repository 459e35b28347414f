#!/usr/bin/env python3
"""Montgomery-product rate of the library on the device (+ parity of a*b^iters against the oracle)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime load order)
import bn254_model as M, oracle_lib as O
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "phase2-bn254_amd", "libmi355zk.so")
L = C.CDLL(so)
f = L.mi355zk_ubench_fp_mul
f.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
a = np.array(M.to_limbs(0x1234567890abcdef1234567890abcdef1234567890abcdef % M.Q), dtype=np.uint64)
b = np.array(M.to_limbs(M.Q - 0xfedcba9876543210), dtype=np.uint64)
for which, name in ((0, "Fq"), (1, "Fr")):
    for blocks, iters in ((256 * 4, 2000), (256 * 8, 2000), (256 * 16, 2000)):
        out = np.zeros(16, np.uint64); ms = C.c_float()
        rc = f(which, blocks, iters, a.ctypes.data, b.ctypes.data, out.ctypes.data, C.byref(ms)); assert rc == 0
        want = a.copy()
        if blocks == 1024:
            for _ in range(iters): want = O.fe_mul(which, want, b)
            ok = bool(np.array_equal(out[:4], want))
        else:
            ok = None
        muls = blocks * 256 * 4 * iters
        print(f"{os.path.basename(so)} {name} blocks={blocks} ({blocks // 256 * 4 / 4:.0f} waves/SIMD): {ms.value:8.3f} ms  {muls / ms.value / 1e6:8.2f} G mul/s  parity={ok}", flush=True)
