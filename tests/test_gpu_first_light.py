"""First-light GPU parity: NTT and MSM through the C ABI vs the CPU oracle (bit exact)."""
import numpy as np
import pytest

import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 16])
@pytest.mark.parametrize("op", ["fft", "ifft", "coset_fft", "icoset_fft"])
def test_domain_ops_match_oracle(zk, worker, log_n, op):
    a = inputs.random_fr_mont(1 << log_n, seed=100 + log_n)
    want = O.fr_domain_op(a, log_n, op).reshape(-1, 4)
    dom = zk.EvaluationDomain.from_coeffs(a)
    getattr(dom, op)(worker)
    assert np.array_equal(dom.into_coeffs(), want)


@pytest.mark.parametrize("n", [1, 2, 31, 32, 100, 1000, 5000])
def test_g1_multiexp_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(1, n, seed=n)
    scalars = inputs.random_scalars(n, seed=7 * n + 1)
    rc, want = O.G1.multiexp(bases, scalars)
    assert rc == 0
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))


@pytest.mark.parametrize("n", [1, 33, 500])
def test_g2_multiexp_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(2, n, seed=n)
    scalars = inputs.random_scalars(n, seed=11 * n + 1)
    rc, want = O.G2.multiexp(bases, scalars)
    assert rc == 0
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert np.array_equal(O.G2.to_affine(got), O.G2.to_affine(want))
