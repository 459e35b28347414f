"""FFT over curve points (EvaluationDomain<Point<G1>>, the Lagrange-basis conversion of
prepare_phase2).  Bit exact against the oracle's point FFT + batch_normalization, plus the reference's identity
ifft(fft(v)) == v, and the defining property on the tau-power structure prepare_phase2 feeds it."""
import ctypes as C

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(zk, pts, log_n, inverse, group=1):
    import torch

    d = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()
    fn = zk.lib.load().mi355zk_bn254_g1_point_fft_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_point_fft_dev
    assert fn(C.c_void_p(d.data_ptr()), log_n, inverse, None) == 0
    return d.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8])
@pytest.mark.parametrize("op", ["fft", "ifft"])
def test_point_fft_matches_oracle(zk, worker, log_n, op):
    n = 1 << log_n
    pts = inputs.bases_progression_cpu(1, n, seed=40 + log_n)
    if n >= 4:
        pts[3] = 0  # an infinity coefficient
    want = O.point_domain_op(1, pts, log_n, op)
    got = _run(zk, pts, log_n, 1 if op == "ifft" else 0)
    assert np.array_equal(got, want)


def test_point_fft_roundtrip_and_lagrange_property(zk, worker):
    """ifft(fft(v)) == v (domain.rs:427-463 on points); and for v_i = tau^i * G (the tau-table of prepare_phase2) the
    ifft gives L_j(tau) * G with L_j the Lagrange polynomials of the domain: sum_j L_j(tau) = 1, so the outputs sum to G."""
    log_n = 10
    n = 1 << log_n
    tau = 0x1234567890ABCDEF1234567 % M.R_ORDER
    ks = np.array([M.to_limbs(pow(tau, i, M.R_ORDER)) for i in range(n)], dtype=np.uint64)
    pts = O.G1.mul_many_affine(inputs.G1_GEN_RAW, ks)
    lag = _run(zk, pts, log_n, 1)
    acc = O.G1.from_affine(np.zeros(8, np.uint64))
    for p in lag:
        acc = O.G1.add_mixed(acc, p)
    assert np.array_equal(O.G1.to_affine(acc), inputs.G1_GEN_RAW)
    # spot-check against the closed form L_j(tau) = (tau^n - 1) * w^j / (n * (tau - w^j))
    w = M.domain_omega(log_n)
    for j in (0, 1, 77, n - 1):
        wj = pow(w, j, M.R_ORDER)
        lj = (pow(tau, n, M.R_ORDER) - 1) * wj % M.R_ORDER * pow(n * (tau - wj) % M.R_ORDER, -1, M.R_ORDER) % M.R_ORDER
        want = O.G1.to_affine(O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(lj)))
        assert np.array_equal(lag[j], want), j
    back = _run(zk, lag, log_n, 0)
    assert np.array_equal(back, pts)


@pytest.mark.parametrize("log_n", [0, 1, 3, 6])
@pytest.mark.parametrize("op", ["fft", "ifft"])
@pytest.mark.parametrize("trusted", [0, 2])
def test_g2_point_fft_matches_oracle(zk, worker, log_n, op, trusted):
    """The G2 leg of prepare_phase2 (coeffs_g2): bit exact against the oracle's Point<G2> FFT + batch_normalization -- through the plain
    windows (default) and, these points being in the subgroup, under MI355ZK_G2_TRUSTED_SUBGROUP (mode bit 1: the psi-split twiddles)."""
    n = 1 << log_n
    pts = inputs.bases_progression_cpu(2, n, seed=60 + log_n)
    if n >= 4:
        pts[2] = 0  # an infinity coefficient
    want = O.point_domain_op(2, pts, log_n, op)
    got = _run(zk, pts, log_n, (1 if op == "ifft" else 0) | trusted, group=2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("trusted", [0, 2])
def test_g2_point_fft_roundtrip(zk, worker, trusted):
    log_n = 9
    pts = inputs.bases_progression_cpu(2, 1 << log_n, seed=71)
    lag = _run(zk, pts, log_n, 1 | trusted, group=2)
    assert not np.array_equal(lag, pts)
    assert np.array_equal(_run(zk, lag, log_n, 0 | trusted, group=2), pts)


@pytest.mark.parametrize("group,trusted", [(1, 0), (2, 0), (2, 2)])
def test_point_ifft_matches_oracle_at_2e12(zk, worker, group, trusted):
    """prepare_phase2's Lagrange conversion at 2^12 points (the REQUIRED_POWER of BASELINE config 1), G1 and G2, every record
    against the oracle's Point<G> FFT + batch_normalization; an infinity coefficient included."""
    log_n = 12
    pts = inputs.bases_progression_cpu(group, 1 << log_n, seed=90 + group)
    pts[1234] = 0
    want = O.point_domain_op(group, pts, log_n, "ifft")
    got = _run(zk, pts, log_n, 1 | trusted, group=group)
    assert np.array_equal(got, want)
