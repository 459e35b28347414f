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


MU = 0x6F4D8248EEB859FBF83E9682E87CFD46


def test_g2_split_and_psi():
    """G2: k = k1 + k2 mu with both halves non-negative and < 2^128, and psi(P) -- through the table-entry path of the kernels --
    equals mu * P (oracle group law) on random points of the order-r subgroup."""
    import inputs
    import oracle_lib as O
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    assert MU == M.Q % M.R_ORDER
    rnd = random.Random(6)
    R = M.R_ORDER
    for k in [0, 1, MU - 1, MU, MU + 1, R - 1, (1 << 253), MU * MU % R] + [rnd.randrange(R) for _ in range(3000)]:
        sc = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        out = np.zeros(10, np.uint32)
        assert lib.mi355zk_selftest_glv2_split(sc.ctypes.data, out.ctypes.data) == 0
        k1 = sum(int(out[i]) << (32 * i) for i in range(5))
        k2 = sum(int(out[5 + i]) << (32 * i) for i in range(5))
        assert k1 < (1 << 128) and k2 < (1 << 128) and k1 + k2 * MU == k, hex(k)
    pts = inputs.bases_cpu(2, 6, seed=41)
    mu_limbs = np.array(M.to_limbs(MU), dtype=np.uint64)
    for p in pts:
        out = np.zeros(24, np.uint64)
        assert lib.mi355zk_selftest_g2_psi(np.ascontiguousarray(p).ctypes.data, out.ctypes.data) == 0
        want = O.G2.to_affine(O.G2.mul(O.G2.from_affine(p), mu_limbs))
        assert np.array_equal(O.G2.to_affine(out), want)


def test_width5_non_adjacent_form():
    """glv_wnaf5 (the digit string of the one-scalar-for-all-points batch_exp kernel): m = sum d_j 2^j, every non-zero digit odd and at most 15
    in magnitude, at least four zeros after each, the returned top index right; magnitudes up to 2^159 (the split's halves are < 2^128)."""
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    rnd = random.Random(55)
    cases = [0, 1, 2, 15, 16, 17, 31, 32, (1 << 128) - 1, 1 << 127, (1 << 159) - 1, 0xAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA, 0x5555555555555555]
    cases += [rnd.getrandbits(rnd.choice([8, 64, 127, 128, 140, 159])) for _ in range(300)]
    for m in cases:
        limbs = np.array([(m >> (32 * i)) & 0xFFFFFFFF for i in range(5)], dtype=np.uint32)
        dig = np.zeros(164, dtype=np.int8)
        top = lib.mi355zk_selftest_glv_wnaf5(limbs.ctypes.data, dig.ctypes.data)
        assert sum(int(d) << j for j, d in enumerate(dig)) == m
        nz = [j for j, d in enumerate(dig) if d]
        assert top == (nz[-1] if nz else -1)
        assert all(int(dig[j]) % 2 != 0 and abs(int(dig[j])) <= 15 for j in nz)
        assert all(b - a >= 5 for a, b in zip(nz, nz[1:]))
