"""The GLV split the G1 per-point scalar multiplications use (phase2-bn254_amd/csrc/glv.hpp), on the HOST through the library's
self-test hook, against big integers: k1 + k2 * lambda == k (mod r), |k1|, |k2| < 2^128, and the endomorphism constants
themselves: phi(x, y) = (beta x, y) equals lambda * (x, y) on the generator (big-int group law of tests/bn254_model.py)."""
import random

import numpy as np

import bn254_model as M

LAMBDA = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
BETA = 0x59E26BCEA0D48BACD4F263F1ACDB5C4F5763473177FFFFFE


def test_endomorphism_constants():
    assert pow(BETA, 3, M.Q) == 1 and BETA != 1
    assert (LAMBDA * LAMBDA + LAMBDA + 1) % M.R_ORDER == 0
    g = (1, 2)
    assert M.ec_mul(M.FQ_OPS, g, LAMBDA) == (BETA * g[0] % M.Q, g[1])


def test_glv_split_identity_and_size():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    rnd = random.Random(5)
    R = M.R_ORDER
    ks = [0, 1, 2, R - 1, R - 2, LAMBDA, LAMBDA + 1, R - LAMBDA, (1 << 128) - 1, 1 << 128, (1 << 253), (R - 1) // 2] + [rnd.randrange(R) for _ in range(3000)]
    for k in ks:
        sc = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        out = np.zeros(12, np.uint32)
        assert lib.mi355zk_selftest_glv_split(sc.ctypes.data, out.ctypes.data) == 0
        k1 = sum(int(out[i]) << (32 * i) for i in range(5))
        k2 = sum(int(out[5 + i]) << (32 * i) for i in range(5))
        assert k1 < (1 << 128) and k2 < (1 << 128), hex(k)
        if out[10]:
            k1 = -k1
        if out[11]:
            k2 = -k2
        assert (k1 + k2 * LAMBDA - k) % R == 0, hex(k)
