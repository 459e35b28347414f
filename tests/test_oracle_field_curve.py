"""Oracle field + group law vs the independent Python big-int model, and the group-law identities the
reference's own tests assert (pairing/src/bn256/ec.rs:1571-1720, pairing/src/tests/curve.rs)."""
import random

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

rnd = random.Random(0x5DBE6259)  # first seed word of the reference's curve tests


@pytest.mark.parametrize("which,p", [(O.FQ, M.Q), (O.FR, M.R_ORDER)])
def test_field_ops(which, p):
    rinv = pow(M.MONT_R, -1, p)
    for _ in range(300):
        a, b = rnd.randrange(p), rnd.randrange(p)
        assert M.from_limbs(O.fe_mul(which, M.to_limbs(a), M.to_limbs(b))) == a * b * rinv % p
        assert M.from_limbs(O.fe_add(which, M.to_limbs(a), M.to_limbs(b))) == (a + b) % p
        assert M.from_limbs(O.fe_sub(which, M.to_limbs(a), M.to_limbs(b))) == (a - b) % p
    for a in (1, 2, p - 1, rnd.randrange(1, p)):
        got = O.fe_inv(which, M.to_limbs(M.to_mont(a, p)))
        assert M.from_mont(M.from_limbs(got), p) == pow(a, -1, p)
    assert O.fe_inv(which, [0, 0, 0, 0]) is None
    # edge operands
    for a, b in ((0, 0), (p - 1, p - 1), (p - 1, 1), (0, p - 1)):
        assert M.from_limbs(O.fe_mul(which, M.to_limbs(a), M.to_limbs(b))) == a * b * rinv % p
        assert M.from_limbs(O.fe_add(which, M.to_limbs(a), M.to_limbs(b))) == (a + b) % p
        assert M.from_limbs(O.fe_sub(which, M.to_limbs(a), M.to_limbs(b))) == (a - b) % p


def _f2(a):
    return M.to_limbs(M.to_mont(a[0], M.Q)) + M.to_limbs(M.to_mont(a[1], M.Q))


def _unf2(l):
    return (M.from_mont(M.from_limbs(l[:4]), M.Q), M.from_mont(M.from_limbs(l[4:]), M.Q))


def test_fq2_ops():
    for _ in range(100):
        a = (rnd.randrange(M.Q), rnd.randrange(M.Q))
        b = (rnd.randrange(M.Q), rnd.randrange(M.Q))
        assert _unf2(O.fq2_mul(_f2(a), _f2(b))) == M.f2_mul(a, b)
        assert _unf2(O.fq2_sqr(_f2(a))) == M.f2_mul(a, a)
        assert _unf2(O.fq2_inv(_f2(a))) == M.f2_inv(a)


GROUPS = [(O.G1, M.FQ_OPS, M.G1_GEN, inputs.G1_GEN_RAW, M.g1_jac_from_raw, M.g1_affine_from_raw, M.g1_affine_to_raw),
          (O.G2, M.FQ2_OPS, M.G2_GEN, inputs.G2_GEN_RAW, M.g2_jac_from_raw, M.g2_affine_from_raw, M.g2_affine_to_raw)]


@pytest.mark.parametrize("G,F,gen,gen_raw,jac_from,aff_from,aff_to", GROUPS)
def test_scalar_mul_matches_model(G, F, gen, gen_raw, jac_from, aff_from, aff_to):
    for k in (1, 2, 3, rnd.randrange(M.R_ORDER), M.R_ORDER - 1):
        pj = G.mul(G.from_affine(gen_raw), M.to_limbs(k))
        assert jac_from(pj) == M.ec_mul(F, gen, k)
        assert aff_from(G.to_affine(pj)) == M.ec_mul(F, gen, k)
    # r * G = infinity (ec.rs:1696-1720)
    assert jac_from(G.mul(G.from_affine(gen_raw), M.to_limbs(M.R_ORDER))) is None
    assert not G.to_affine(G.mul(G.from_affine(gen_raw), M.to_limbs(M.R_ORDER))).any()


@pytest.mark.parametrize("G,F,gen,gen_raw,jac_from,aff_from,aff_to", GROUPS)
def test_addition_doubling_mixed_agree(G, F, gen, gen_raw, jac_from, aff_from, aff_to):
    """a + a == 2a == a +mixed a (ec.rs:1571-1659); a + (-a) == 0 incl. mixed (ec.rs:1661-1694)."""
    for _ in range(5):
        k1, k2 = rnd.randrange(1, M.R_ORDER), rnd.randrange(1, M.R_ORDER)
        a = G.mul(G.from_affine(gen_raw), M.to_limbs(k1))
        b = G.mul(G.from_affine(gen_raw), M.to_limbs(k2))
        pa, pb = M.ec_mul(F, gen, k1), M.ec_mul(F, gen, k2)
        assert jac_from(G.add(a, b)) == M.ec_add(F, pa, pb)
        assert jac_from(G.add(a, a)) == M.ec_add(F, pa, pa) == jac_from(G.double(a))
        a_aff = G.to_affine(a)
        assert jac_from(G.add_mixed(a, a_aff)) == M.ec_add(F, pa, pa)
        assert jac_from(G.add_mixed(b, a_aff)) == M.ec_add(F, pa, pb)
        neg_a = np.array(aff_to(M.ec_neg(F, pa)), dtype=np.uint64)
        assert jac_from(G.add_mixed(a, neg_a)) is None
        assert jac_from(G.add(a, G.from_affine(neg_a))) is None
        zero = G.from_affine(np.zeros(G.aff, np.uint64))
        assert jac_from(zero) is None
        assert jac_from(G.add(zero, a)) == pa and jac_from(G.add(a, zero)) == pa
        assert jac_from(G.add_mixed(zero, a_aff)) == pa
        assert G.eq(G.add(a, b), G.add(b, a)) and not G.eq(a, b)


@pytest.mark.parametrize("G,F,gen,gen_raw,jac_from,aff_from,aff_to", GROUPS)
def test_batch_normalization(G, F, gen, gen_raw, jac_from, aff_from, aff_to):
    """batch_normalization == per-point into_affine (pairing/src/tests/curve.rs:347-384), with
    infinity and already-normalised elements mixed in."""
    pts = []
    for i in range(12):
        if i % 5 == 0:
            pts.append(G.from_affine(np.zeros(G.aff, np.uint64)))
        elif i % 5 == 1:
            pts.append(G.from_affine(gen_raw))
        else:
            pts.append(G.mul(G.from_affine(gen_raw), M.to_limbs(rnd.randrange(1, M.R_ORDER))))
    v = G.batch_normalization(np.concatenate(pts)).reshape(len(pts), -1)
    for before, after in zip(pts, v):
        assert jac_from(before) == jac_from(after)
        if jac_from(before) is not None:
            assert np.array_equal(after[: G.aff], G.to_affine(before))
