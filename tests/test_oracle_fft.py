"""Oracle FFT (restatement of bellman/src/domain.rs) vs golden vectors (definition DFT) and the
identities the reference's own tests assert (domain.rs:380-496)."""
import numpy as np
import pytest

import bn254_model as M
import golden_util as GU
import inputs
import oracle_lib as O


def test_golden_vectors():
    n = 0
    for c in GU.ntt_cases():
        got = O.fr_serial_fft(c["input"], c["log_n"], c["omega"]).reshape(-1, 4)
        assert np.array_equal(got, c["fft"]), c["log_n"]
        for op in ("fft", "ifft", "coset_fft", "icoset_fft"):
            assert np.array_equal(O.fr_domain_op(c["input"], c["log_n"], op).reshape(-1, 4), c[op]), (c["log_n"], op)
        n += 1
    assert n >= 7


@pytest.mark.parametrize("log_n", [1, 4, 9, 12])
def test_fft_consistency(log_n):
    """domain.rs:427-463: ifft(fft(a)) == a and icoset_fft(coset_fft(a)) == a."""
    a = inputs.random_fr_mont(1 << log_n, seed=log_n)
    assert np.array_equal(O.fr_domain_op(O.fr_domain_op(a, log_n, "fft"), log_n, "ifft").reshape(-1, 4), a)
    assert np.array_equal(O.fr_domain_op(O.fr_domain_op(a, log_n, "coset_fft"), log_n, "icoset_fft").reshape(-1, 4), a)


@pytest.mark.parametrize("log_n,log_cpus", [(3, 1), (6, 2), (9, 2), (10, 3)])
def test_parallel_fft_consistency(log_n, log_cpus):
    """domain.rs:465-496: parallel_fft == serial_fft."""
    a = inputs.random_fr_mont(1 << log_n, seed=40 + log_n)
    omega = O.fr_domain(log_n)[0]
    assert np.array_equal(O.fr_parallel_fft(a, log_n, omega, log_cpus), O.fr_serial_fft(a, log_n, omega))


def test_polynomial_arith():
    """domain.rs:380-425: naive polynomial product == ifft(fft(a) .* fft(b)) (sizes up to 2^5 coefficients)."""
    import random
    rnd = random.Random(3)
    for la in (1, 3, 7):
        for lb in (1, 5, 8):
            a = [rnd.randrange(M.R_ORDER) for _ in range(la)]
            b = [rnd.randrange(M.R_ORDER) for _ in range(lb)]
            naive = [0] * (la + lb)
            for i, x in enumerate(a):
                for j, y in enumerate(b):
                    naive[i + j] = (naive[i + j] + x * y) % M.R_ORDER
            log_n = max(1, (la + lb - 1).bit_length())
            n = 1 << log_n
            mont = lambda v: np.array([M.to_limbs(M.to_mont(t, M.R_ORDER)) for t in v + [0] * (n - len(v))], dtype=np.uint64)  # noqa: E731
            fa, fb = O.fr_domain_op(mont(a), log_n, "fft"), O.fr_domain_op(mont(b), log_n, "fft")
            prod = O.fe_mul_many(O.FR, fa, fb)
            back = O.fr_domain_op(prod, log_n, "ifft").reshape(-1, 4)
            got = [M.from_mont(M.from_limbs(r), M.R_ORDER) for r in back]
            assert got[: la + lb] == naive and not any(got[la + lb:])


@pytest.mark.parametrize("group", [1, 2])
def test_point_fft_oracle_against_model(group):
    """EvaluationDomain<Point<G>> (group.rs:22-51): the oracle's point FFT is the scalar FFT 'in the exponent'.
    For v_i = a_i * G:  fft(v)_k = (sum_i a_i w^(ik)) * G, checked with the independent model's DFT; and
    ifft(fft(v)) == v; parallel_fft shape == serial (domain.rs:465-496)."""
    import random
    rnd = random.Random(11 + group)
    G = O.G1 if group == 1 else O.G2
    F, gen = (M.FQ_OPS, M.G1_GEN) if group == 1 else (M.FQ2_OPS, M.G2_GEN)
    gen_raw = inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW
    to_raw = M.g1_affine_to_raw if group == 1 else M.g2_affine_to_raw
    log_n = 3
    n = 1 << log_n
    a = [rnd.randrange(M.R_ORDER) for _ in range(n)]
    a[2] = 0
    pts = G.mul_many_affine(gen_raw, np.array([M.to_limbs(v) for v in a], dtype=np.uint64))
    got = O.point_domain_op(group, pts, log_n, "fft")
    want = M.dft(a, M.domain_omega(log_n))
    for k in range(n):
        assert list(got[k]) == to_raw(M.ec_mul(F, gen, want[k])), k
    back = O.point_domain_op(group, got, log_n, "ifft")
    assert np.array_equal(back, pts)
    assert np.array_equal(O.point_domain_op(group, pts, log_n, "fft", log_cpus=1), got)
