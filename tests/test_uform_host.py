"""The kernels' U-form arithmetic (29-bit lazy limbs: csrc/fieldu.hpp, curveu.hpp) compiled for the HOST
and checked against Python big ints / the oracle, including the stated limb and value bounds."""
import ctypes as C
import random

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

MASK = (1 << 29) - 1
rnd = random.Random(29)
MODS = [(0, M.Q), (1, M.R_ORDER)]


def limbs29(v, n=9):
    return [(v >> (29 * i)) & MASK for i in range(n - 1)] + [v >> (29 * (n - 1))]


def val(l):
    return sum(int(x) << (29 * i) for i, x in enumerate(l))


def redundant(v, extra_bits):
    """a non-normalised limb vector with the same value: limbs up to 2^(29+extra_bits)"""
    l = limbs29(v)
    for i in range(8):
        if l[i + 1] > 0:
            take = rnd.randrange(0, min(l[i + 1], (1 << extra_bits) - 1) + 1)
            l[i + 1] -= take
            l[i] += take << 29
    assert val(l) == v
    return l


@pytest.fixture(scope="module")
def lib():
    import phase2_bn254_amd as zk

    return zk.lib.load()


def _u32(l):
    return np.array(l, dtype=np.uint32)


@pytest.mark.parametrize("which,p", MODS)
def test_u_mul(lib, which, p):
    rinv = pow(1 << 261, -1, p)
    for trial in range(400):
        # values up to 10p, limbs up to 2^30 (the loosest operands the kernels feed u_mul)
        va, vb = rnd.randrange(10 * p), rnd.randrange(10 * p)
        a = _u32(redundant(va, 1) if trial % 2 else limbs29(va))
        b = _u32(redundant(vb, 1) if trial % 3 == 0 else limbs29(vb))
        out = np.zeros(9, np.uint32)
        assert lib.mi355zk_selftest_u_mul(which, a.ctypes.data, b.ctypes.data, out.ctypes.data) == 0
        r = val(out)
        assert r % p == va * vb * rinv % p
        assert r < va * vb // (1 << 261) + p + 1 and r < 2 * p   # the bound the callers rely on
        assert all(int(x) <= MASK for x in out[:8])                # N-form
    # extreme limbs: one operand N-form with all limbs 2^29-1 (value ~2^261 is out of range for the value bound but not for the accumulator)
    a = _u32([MASK] * 8 + [(10 * p) >> 232])
    b = _u32([(1 << 30) - 1] * 8 + [0])
    out = np.zeros(9, np.uint32)
    lib.mi355zk_selftest_u_mul(which, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert val(out) % p == val(a) * val(b) * rinv % p


@pytest.mark.parametrize("which,p", MODS)
def test_u_mul_by_a_constant_with_its_quotient(lib, which, p):
    """u_mul_shoup (the NTT's product by a table twiddle): the plain product a * w reduced with the precomputed quotient floor(w 2^261 / p);
    operands as loose as the stage code feeds it (values up to 160p, limbs up to 2^31), the result below 2p in N-form."""
    edge_w = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 253, (1 << 29) - 1]
    for trial in range(400):
        vw = edge_w[trial] if trial < len(edge_w) else rnd.randrange(p)
        bound = (160 * p, 30 * p, 2 * p, p)[trial % 4]
        va = rnd.randrange(bound) if trial % 7 else bound - 1
        a = _u32(redundant(va, 2) if trial % 2 else limbs29(va))
        w = np.array([(vw >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        out, wq = np.zeros(9, np.uint32), np.zeros(9, np.uint32)
        assert lib.mi355zk_selftest_u_mul_shoup(which, a.ctypes.data, w.ctypes.data, out.ctypes.data, wq.ctypes.data) == 0
        assert val(wq) == (vw << 261) // p and all(int(x) <= MASK for x in wq)
        r = val(out)
        assert r % p == va * vw % p
        assert r < 2 * p and all(int(x) <= MASK for x in out[:8])
        q = (va * vw - r) // p
        assert va * vw // p - 2 <= q <= va * vw // p
    # the widest limbs the columns admit: every limb of a at 2^31 - 1
    a = _u32([(1 << 31) - 1] * 8 + [(100 * p) >> 232])
    vw = p - 1
    w = np.array([(vw >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
    out, wq = np.zeros(9, np.uint32), np.zeros(9, np.uint32)
    lib.mi355zk_selftest_u_mul_shoup(which, a.ctypes.data, w.ctypes.data, out.ctypes.data, wq.ctypes.data)
    assert val(out) % p == val(a) * vw % p and val(out) < 2 * p


@pytest.mark.parametrize("which,p", MODS)
@pytest.mark.parametrize("k,s", [(1, 1), (2, 1), (4, 1), (4, 2), (4, 3), (8, 1)])
def test_u_sub(lib, which, p, k, s):
    for _ in range(200):
        va = rnd.randrange(2 * p)
        vb = rnd.randrange(k * p + 1)
        a = _u32(limbs29(va))
        bl = limbs29(vb)
        if s > 1:  # b with limbs up to s * 2^29
            bl = redundant(vb, 1) if s == 2 else [x for x in redundant(vb, 1)]
        b = _u32(bl)
        assert all(x < s * (1 << 29) for x in bl[:8])
        out = np.zeros(9, np.uint32)
        assert lib.mi355zk_selftest_u_sub(which, k, s, a.ctypes.data, b.ctypes.data, out.ctypes.data) == 0
        assert val(out) == va + k * p - vb
        assert all(int(x) <= MASK for x in out[:8])
    # boundary: b == k*p exactly and a == 0
    a, b, out = _u32([0] * 9), _u32(limbs29(k * p)), np.zeros(9, np.uint32)
    lib.mi355zk_selftest_u_sub(which, k, s, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert val(out) == 0


@pytest.mark.parametrize("which,p", MODS)
def test_u_pack_roundtrip_and_canonical_reduce(lib, which, p):
    for v in [0, 1, p - 1, rnd.randrange(p), rnd.randrange(p)]:
        a = np.array(M.to_limbs(v), dtype=np.uint64)
        u = np.zeros(9, np.uint32)
        lib.mi355zk_selftest_u_pack(which, a.ctypes.data, u.ctypes.data, None, None)
        assert val(u) == v
    for v in [0, 1, p - 1, p, p + 1, 2 * p - 1, rnd.randrange(2 * p)]:
        u = _u32(limbs29(v))
        out = np.zeros(4, np.uint64)
        lib.mi355zk_selftest_u_pack(which, None, None, u.ctypes.data, out.ctypes.data)
        assert M.from_limbs(out) == v % p


@pytest.mark.parametrize("which,p", MODS)
def test_u_reduce_below_32p_without_a_product(lib, which, p):
    """u_to_std_lt32p: the NTT's closing reduction (values < 30p after ten butterfly stages) by a quotient estimate from the top
    limb, a limbwise v - q p and two conditional subtractions; every multiple of p up to 32p and its neighbours, and random values."""
    cases = [0, 1, 32 * p - 1]
    for k in range(1, 32):
        cases += [k * p - 1, k * p, k * p + 1]
    cases += [rnd.randrange(32 * p) for _ in range(3000)]
    for v in cases:
        u = _u32(limbs29(v))
        out = np.zeros(4, np.uint64)
        assert lib.mi355zk_selftest_u_reduce32(which, u.ctypes.data, out.ctypes.data) == 0
        assert M.from_limbs(out) == v % p, v // p


def _xyzz_to_affine(x):
    X, Y, ZZ, ZZZ = (M.from_mont(M.from_limbs(x[4 * i:4 * i + 4]), M.Q) for i in range(4))
    if ZZ == 0:
        return None
    return (X * pow(ZZ, -1, M.Q) % M.Q, Y * pow(ZZ, -1 if False else -1, M.Q) * 0 + Y * pow(ZZZ, -1, M.Q) % M.Q)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_g1_bucket_accumulation_on_host(lib, mode):
    """sum of signed points, incl. the same point twice in a row (doubling branch), P then -P (infinity) and
    restarting from infinity: saturated XYZZ (mode 0), U-form XYZZ (mode 1), the U-form accumulator CARRIED through an
    R-domain record after every third point (mode 2: xyzzu_to_r / xyzzu_from_r, the chunks of a streamed multiexp) and the
    PAIR-per-bucket addition replayed as its two lanes (mode 3: the rounds of pair_add_mixed) against the big-int model."""
    n = 40
    raw = inputs.bases_cpu(1, n, seed=123)
    pts = [M.g1_affine_from_raw(r) for r in raw]
    seq = [(0, 0), (0, 0), (1, 0), (1, 1), (2, 1), (2, 0), (3, 0)]           # dbl; add; cancel ...
    seq += [(i, rnd.randrange(2)) for i in range(4, n)]
    seq += [(5, 0), (5, 0), (5, 1), (5, 1)]
    for prefix in (1, 2, 4, 6, 7, len(seq)):
        sub = seq[:prefix]
        arr = np.ascontiguousarray(np.stack([raw[i] for i, _ in sub]))
        neg = np.array([s for _, s in sub], dtype=np.uint8)
        out = np.zeros(16, np.uint64)
        assert lib.mi355zk_selftest_g1_accumulate(mode, arr.ctypes.data, neg.ctypes.data, len(sub), out.ctypes.data) == 0
        want = None
        for i, s in sub:
            want = M.ec_add(M.FQ_OPS, want, M.ec_neg(M.FQ_OPS, pts[i]) if s else pts[i])
        assert _xyzz_to_affine(out) == want, (mode, prefix)
        # coordinates are canonical (< q) in the memory format
        assert all(M.from_limbs(out[4 * i:4 * i + 4]) < M.Q for i in range(4))


def test_u_accumulation_long_chain_keeps_invariants(lib):
    """2000 random signed additions through the U-form accumulator == the saturated accumulator == model."""
    n = 2000
    raw = inputs.bases_progression_cpu(1, n, seed=321)
    neg = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    outs = []
    for mode in (0, 1, 2, 3):
        out = np.zeros(16, np.uint64)
        assert lib.mi355zk_selftest_g1_accumulate(mode, raw.ctypes.data, neg.ctypes.data, n, out.ctypes.data) == 0
        outs.append(_xyzz_to_affine(out))
    assert outs[0] == outs[1] == outs[2] == outs[3] and outs[0] is not None
    acc = O.G1.from_affine(np.zeros(8, np.uint64))
    for i in range(n):
        pt = raw[i].copy()
        if neg[i]:
            y = M.from_limbs(pt[4:])
            pt[4:] = M.to_limbs((M.Q - y) % M.Q)
        acc = O.G1.add_mixed(acc, pt)
    assert M.g1_jac_from_raw(acc) == outs[1]


@pytest.mark.parametrize("mode", [0, 1])
def test_g1_record_sums_in_the_r_domain_on_host(lib, mode):
    """The records of the G1 bucket reduction (curveu.hpp: 2^261-domain XYZZ): buckets accumulated by the mixed addition, turned
    into records and summed by the FULL U-form addition -- kept in registers (mode 0) or stored and re-loaded after every
    addition (mode 1) -- against the big-int model.  Covers: ordinary sums; empty buckets (infinity operands on either side);
    the same point in two buckets, once with ZZ = 1 and once behind a cancelled detour so that ZZ != 1 (doubling branch); a
    bucket and its negative (infinity); restarting from infinity; and a long chain (invariants)."""
    n = 48
    raw = inputs.bases_cpu(1, n, seed=777)
    pts = [M.g1_affine_from_raw(r) for r in raw]
    cases = [
        [(0, 0, 0), (1, 0, 1), (2, 1, 1), (3, 0, 2)],                                  # P0 + (P1 - P2) + P3
        [(0, 0, 1), (1, 0, 3)],                                                        # empty buckets 0 and 2
        [(0, 0, 0), (0, 0, 1)],                                                        # P0 + P0, both ZZ = 1
        [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1)],                                  # (P0 + P1 - P1) + P0: doubling with ZZ != 1
        [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1)],                                  # (P0 + P1) + (-P0 - P1) = infinity
        [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1), (2, 0, 2), (3, 1, 3)],            # ... and on from infinity
        [(i, rnd.randrange(2), rnd.randrange(12)) for i in range(n)],                  # 12 buckets of ~4
        [(i % n, rnd.randrange(2), j // 3) for j, i in enumerate(range(600))],         # 200 records in one chain
    ]
    for sub in cases:
        n_groups = max(g for _, _, g in sub) + 1
        arr = np.ascontiguousarray(np.stack([raw[i] for i, _, _ in sub]))
        neg = np.array([s for _, s, _ in sub], dtype=np.uint8)
        grp = np.array([g for _, _, g in sub], dtype=np.uint32)
        out = np.zeros(16, np.uint64)
        assert lib.mi355zk_selftest_g1_record_sum(mode, arr.ctypes.data, neg.ctypes.data, grp.ctypes.data, len(sub), n_groups, out.ctypes.data) == 0
        want = None
        for g in range(n_groups):                       # group order, as the library sums them (the group law is associative anyway)
            b = None
            for i, s, gg in sub:
                if gg == g:
                    b = M.ec_add(M.FQ_OPS, b, M.ec_neg(M.FQ_OPS, pts[i]) if s else pts[i])
            want = M.ec_add(M.FQ_OPS, want, b)
        assert _xyzz_to_affine(out) == want, (mode, sub[:6])
        assert all(M.from_limbs(out[4 * i:4 * i + 4]) < M.Q for i in range(4))


def _xyzz2_to_affine(x):
    c = [M.from_mont(M.from_limbs(x[4 * i:4 * i + 4]), M.Q) for i in range(8)]
    X, Y, ZZ, ZZZ = (c[0], c[1]), (c[2], c[3]), (c[4], c[5]), (c[6], c[7])
    if ZZ == (0, 0):
        return None
    return (M.f2_mul(X, M.f2_inv(ZZ)), M.f2_mul(Y, M.f2_inv(ZZZ)))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_g2_bucket_accumulation_on_host(lib, mode):
    """(mode 3: the pair-per-bucket addition over Fq2 replayed as its two lanes, the doubling through xyzzu2_double_affine)"""
    n = 24
    raw = inputs.bases_cpu(2, n, seed=223)
    pts = [M.g2_affine_from_raw(r) for r in raw]
    seq = [(0, 0), (0, 0), (1, 0), (1, 1), (2, 1), (2, 0), (3, 0)]
    seq += [(i, rnd.randrange(2)) for i in range(4, n)]
    seq += [(5, 1), (5, 1), (5, 0), (5, 0)]
    for prefix in (1, 2, 4, 6, 7, len(seq)):
        sub = seq[:prefix]
        arr = np.ascontiguousarray(np.stack([raw[i] for i, _ in sub]))
        neg = np.array([s for _, s in sub], dtype=np.uint8)
        out = np.zeros(32, np.uint64)
        assert lib.mi355zk_selftest_g2_accumulate(mode, arr.ctypes.data, neg.ctypes.data, len(sub), out.ctypes.data) == 0
        want = None
        for i, s in sub:
            want = M.ec_add(M.FQ2_OPS, want, M.ec_neg(M.FQ2_OPS, pts[i]) if s else pts[i])
        assert _xyzz2_to_affine(out) == want, (mode, prefix)
        assert all(M.from_limbs(out[4 * i:4 * i + 4]) < M.Q for i in range(8))


@pytest.mark.parametrize("mode", [0, 1])
def test_g2_record_sums_in_the_r_domain_on_host(lib, mode):
    """G2 twin of test_g1_record_sums_in_the_r_domain_on_host (the full U-form addition over Fq2 and its rare branches)."""
    n = 24
    raw = inputs.bases_cpu(2, n, seed=778)
    pts = [M.g2_affine_from_raw(r) for r in raw]
    cases = [
        [(0, 0, 0), (1, 0, 1), (2, 1, 1), (3, 0, 2)],
        [(0, 0, 1), (1, 0, 3)],
        [(0, 0, 0), (0, 0, 1)],
        [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1)],
        [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1)],
        [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1), (2, 0, 2), (3, 1, 3)],
        [(i, rnd.randrange(2), rnd.randrange(8)) for i in range(n)],
        [(i % n, rnd.randrange(2), j // 3) for j, i in enumerate(range(240))],
    ]
    for sub in cases:
        n_groups = max(g for _, _, g in sub) + 1
        arr = np.ascontiguousarray(np.stack([raw[i] for i, _, _ in sub]))
        neg = np.array([s for _, s, _ in sub], dtype=np.uint8)
        grp = np.array([g for _, _, g in sub], dtype=np.uint32)
        out = np.zeros(32, np.uint64)
        assert lib.mi355zk_selftest_g2_record_sum(mode, arr.ctypes.data, neg.ctypes.data, grp.ctypes.data, len(sub), n_groups, out.ctypes.data) == 0
        want = None
        for g in range(n_groups):
            b = None
            for i, s, gg in sub:
                if gg == g:
                    b = M.ec_add(M.FQ2_OPS, b, M.ec_neg(M.FQ2_OPS, pts[i]) if s else pts[i])
            want = M.ec_add(M.FQ2_OPS, want, b)
        assert _xyzz2_to_affine(out) == want, (mode, sub[:6])
        assert all(M.from_limbs(out[4 * i:4 * i + 4]) < M.Q for i in range(8))


def test_g2_u_accumulation_long_chain(lib):
    n = 600
    raw = inputs.bases_progression_cpu(2, n, seed=421)
    neg = np.array([rnd.randrange(2) for _ in range(n)], dtype=np.uint8)
    outs = []
    for mode in (0, 1, 2, 3):
        out = np.zeros(32, np.uint64)
        assert lib.mi355zk_selftest_g2_accumulate(mode, raw.ctypes.data, neg.ctypes.data, n, out.ctypes.data) == 0
        outs.append(_xyzz2_to_affine(out))
    assert outs[0] == outs[1] == outs[2] == outs[3] and outs[0] is not None


def test_g2_scalar_mul_on_u_form_jacobian_host(lib):
    """The U-form Fq2 Jacobian arithmetic (jacu2_double, jacu2_add_tab, table entries) run on the HOST through the same
    windowed program as batch_exp_win_u2_kernel, against the oracle's mul_assign + into_affine: random scalars, 0, 1, 2, r - 1,
    digits that hit every table entry and both signs."""
    pts = inputs.bases_progression_cpu(2, 6, seed=77)
    ks = [0, 1, 2, 8, 9, 0x8888, 0xFFFFFFFF, M.R_ORDER - 1, M.R_ORDER - 2] + [rnd.randrange(M.R_ORDER) for _ in range(12)]
    for idx, k in enumerate(ks):
        p = pts[idx % len(pts)]
        kl = np.array(M.to_limbs(k), dtype=np.uint64)
        out = np.zeros(24, np.uint64)
        assert lib.mi355zk_selftest_g2_scalar_mul_u(p.ctypes.data, kl.ctypes.data, out.ctypes.data) == 0
        want = O.G2.to_affine(O.G2.mul(O.G2.from_affine(p), kl))
        assert np.array_equal(O.G2.to_affine(out), want), hex(k)
