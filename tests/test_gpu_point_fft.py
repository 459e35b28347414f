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


def _scalars_to_dev(vals):
    """list of Python ints (< 2^256) -> (n, 4) int64 device tensor of canonical limbs"""
    import torch

    buf = b"".join(v.to_bytes(32, "little") for v in vals)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.int64).reshape(-1, 4).copy()).cuda()


@pytest.mark.parametrize("group,log_n,trusted", [(1, 16, 0), (1, 20, 0), (2, 16, 0), (2, 16, 2), (2, 18, 0)])
def test_point_ifft_at_size_on_a_tau_table(zk, worker, group, log_n, trusted):
    """prepare_phase2's Lagrange conversion (powersoftau/src/bin/prepare_phase2.rs:68-105; bellman/src/group.rs:22-51 under
    domain.rs:154-173) at the sizes the reference runs it: G1 at 2^16 and 2^20, G2 at 2^16 and 2^18 (plain windows, and the psi split under the
    caller's promise).  Input: the tau-table v_i = tau^i G; the ifft is then out_j = L_j(tau) G, L_j the Lagrange polynomials of the domain.
      (1) >= 256 indices against the closed form L_j(tau) = (tau^n - 1) w^j / (n (tau - w^j)) through the oracle's mul + into_affine;
      (2) sum_j out_j == G (the L_j sum to one) -- through the library's dense_multiexp with unit exponents;
      (3) EVERY output record in one linear form, independent of the forward point FFT:  sum_j w^(jk) out_j == tau^k G  (row k of the DFT
          that inverts the ifft), k = 1 and a large odd k, again by dense_multiexp (parity-tested on its own);
      (4) fft(ifft(v)) == v, record for record (domain.rs:427-463)."""
    import torch

    G = O.G1 if group == 1 else O.G2
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    L = zk.lib.load()
    r = M.R_ORDER
    n = 1 << log_n
    tau = 0x2B3C4D5E6F708192A3B4C5D6E7F8091A2B3C4D5E6F708192A3B4C5D6E7F809 % r
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * tau % r
    d_k = _scalars_to_dev(pw)
    pts = torch.empty((n, 8 * group), dtype=torch.int64, device="cuda")
    mul = L.mi355zk_bn254_g1_batch_mul_dev if group == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    fft = L.mi355zk_bn254_g1_point_fft_dev if group == 1 else L.mi355zk_bn254_g2_point_fft_dev
    assert mul(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(d_k.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(pts[1].cpu().numpy().view(np.uint64), G.to_affine(G.mul(G.from_affine(gen), np.array(M.to_limbs(tau), dtype=np.uint64))))
    lag = pts.clone()
    assert fft(C.c_void_p(lag.data_ptr()), log_n, 1 | trusted, None) == 0
    # (1) the closed form on a sample
    w = M.domain_omega(log_n)
    rng = np.random.default_rng(9100 + log_n + group)
    idx = np.unique(np.concatenate([[0, 1, 2, n // 2 - 1, n // 2, n - 2, n - 1], rng.integers(0, n, size=270)]))
    assert len(idx) >= 256
    h = lag[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    tn1 = (pow(tau, n, r) - 1) % r
    g_jac = G.from_affine(gen)
    for i, j in enumerate(idx):
        wj = pow(w, int(j), r)
        lj = tn1 * wj % r * pow(n * (tau - wj) % r, -1, r) % r
        assert np.array_equal(h[i], G.to_affine(G.mul(g_jac, np.array(M.to_limbs(lj), dtype=np.uint64)))), int(j)
    # (2) the outputs sum to G
    ones = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ones[:, 0] = 1
    assert np.array_equal(G.to_affine(zk.ceremony.dense_multiexp(lag, ones)), gen)
    # (3) rows of the inverting DFT over every output record
    for k in (1, (n // 2 + 12345) | 1):
        wk = pow(w, k, r)
        col = [1] * n
        for j in range(1, n):
            col[j] = col[j - 1] * wk % r
        got = zk.ceremony.dense_multiexp(lag, _scalars_to_dev(col))
        assert np.array_equal(G.to_affine(got), pts[k].cpu().numpy().view(np.uint64)), k
    # (4) the forward transform brings the table back
    back = lag.clone()
    assert fft(C.c_void_p(back.data_ptr()), log_n, 0 | trusted, None) == 0
    assert torch.equal(back, pts)
