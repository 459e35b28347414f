"""G2 records outside the order-r subgroup (and records on no curve at all) through every scalar-multiplication entry point.

The reference's wNAF `mul` is exact for every point of the twist and its bn256 decoders only check the curve equation
(pairing/src/bn256/ec.rs:1136-1344), so the G2 kernels run PLAIN windows by default and return the oracle's bytes for such records.
The psi split (k = k1 + k2 mu with psi(P) for mu P, glv.hpp -- true in the subgroup only) is what the caller opts into with
MI355ZK_G2_TRUSTED_SUBGROUP; the check psi(P) == mu P (mu P by a plain double-and-add: mi355zk_bn254_g2_subgroup_check_dev, on the host
mi355zk_selftest_g2_in_subgroup) is how a caller with untrusted G2 data earns the right to give that promise."""
import ctypes as C

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O


def _twist_point(seed: int):
    """an on-curve point of the twist that is (with overwhelming probability) NOT in the order-r subgroup: x = (seed, 1), y = a
    square root of x^3 + b' if there is one (oracle Fq2::sqrt); r * P != infinity confirms it."""
    b = O.g2_coeff_b()
    to_mont = lambda v: np.array(M.to_limbs(M.to_mont(v, M.Q)), dtype=np.uint64)  # noqa: E731
    for c0 in range(seed, seed + 200):
        x = np.concatenate([to_mont(c0), to_mont(1)])
        rhs = O.fq2_mul(O.fq2_sqr(x), x)
        rhs = np.concatenate([O.fe_add(0, rhs[:4], b[:4]), O.fe_add(0, rhs[4:], b[4:])])
        y = O.fq2_sqrt(rhs)
        if y is None:
            continue
        p = np.concatenate([x, y])
        rc, _, back = O.decode_points(2, O.encode_points(2, p.reshape(1, 16), False), compressed=False, checked=True)
        assert rc == 0 and np.array_equal(back[0], p)            # on the curve (the reference's own check accepts it)
        if O.G2.to_affine(O.G2.mul(O.G2.from_affine(p), np.array(M.to_limbs(M.R_ORDER), dtype=np.uint64))).any():
            return p
    raise AssertionError("no twist point found")


@pytest.fixture(scope="module")
def lib():
    import phase2_bn254_amd as zk

    return zk.lib.load()


def test_subgroup_points_pass_and_a_cofactor_point_fails_on_host(lib):
    pts = inputs.bases_progression_cpu(2, 6, seed=4711)
    for p in pts:
        assert lib.mi355zk_selftest_g2_in_subgroup(np.ascontiguousarray(p).ctypes.data_as(C.c_void_p)) == 1
    assert lib.mi355zk_selftest_g2_in_subgroup(np.ascontiguousarray(inputs.G2_GEN_RAW).ctypes.data_as(C.c_void_p)) == 1
    assert lib.mi355zk_selftest_g2_in_subgroup(np.zeros(16, np.uint64).ctypes.data_as(C.c_void_p)) == 1      # the identity
    for seed in (3, 1000):
        bad = _twist_point(seed)
        assert lib.mi355zk_selftest_g2_in_subgroup(bad.ctypes.data_as(C.c_void_p)) == 0
        # a subgroup point plus a cofactor point: on the twist, not in the subgroup
        mixed = O.G2.to_affine(O.G2.add_mixed(O.G2.from_affine(pts[0]), bad))
        assert lib.mi355zk_selftest_g2_in_subgroup(mixed.ctypes.data_as(C.c_void_p)) == 0


def test_the_membership_test_is_sound_for_bn254():
    """scalar_mul.hip g2_in_subgroup: P in G2 <=> f(psi) P == 0 with f = (x + 1) + x X + x X^2 - 2 x X^3.  psi satisfies chi = X^2 - t X + q on all of
    E'(Fq2), so f(psi) P == 0 implies Res(f, chi) P == 0; the order of P then divides gcd(Res, #E'(Fq2)) -- which is r: a point that passes
    is in the subgroup, and a member passes because r | Res means f(q) == 0 mod r for psi's eigenvalue q.  The same for rounds 3-4's X - 6 x^2."""
    import math

    x = 0x44E992B44A6909F1
    q, r, t = 36 * x**4 + 36 * x**3 + 24 * x**2 + 6 * x + 1, 36 * x**4 + 36 * x**3 + 18 * x**2 + 6 * x + 1, 6 * x**2 + 1
    assert q == M.Q and r == M.R_ORDER
    order = r * (2 * q - r)                                       # #E'(Fq2) of the sextic twist that carries G2

    def resultant_with_chi(f):                                    # f: integer coefficients, low degree first; Res(f, chi) by reducing mod chi
        # work in Z[X] / (chi): X^2 = t X - q.  Res(chi, f) = prod over the two roots of chi of f(root) = norm of f(X) in Z[X]/(chi) (chi monic)
        a, b = 0, 0                                               # f(X) = a + b X mod chi
        for c in reversed(f):
            a, b = c - q * b, a + t * b                           # (a + b X) X + c
        # norm of a + b X over the roots s1, s2 (s1 + s2 = t, s1 s2 = q): (a + b s1)(a + b s2) = a^2 + a b t + b^2 q
        return a * a + a * b * t + b * b * q

    for f in ([x + 1, x, x, -2 * x], [-(t - 1), 1]):
        res = resultant_with_chi(f)
        assert res % r == 0 and math.gcd(res, 2 * q - r) == 1 and math.gcd(res, order) == r
        assert sum(c * pow(q, k, r) for k, c in enumerate(f)) % r == 0   # members pass: psi acts on G2 as q


def _limbs(v):
    return np.array(M.to_limbs(v % (1 << 256)), dtype=np.uint64)


def _g2_vector_with_cofactor_points(n, seed):
    """subgroup points with on-twist, out-of-subgroup records among them: two cofactor points, a subgroup point plus a cofactor point, and
    an infinity record"""
    pts = inputs.bases_progression_cpu(2, n, seed=seed)
    bad = [_twist_point(77), _twist_point(1000)]
    pts[1] = bad[0]
    pts[n // 2] = bad[1]
    pts[n - 2] = O.G2.to_affine(O.G2.add_mixed(O.G2.from_affine(pts[0]), bad[0]))
    pts[3] = 0
    return pts


@pytest.mark.gpu
def test_subgroup_check_on_device(zk, worker):
    import torch

    n = 3000
    pts = inputs.bases_progression_cpu(2, n, seed=4712)
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    assert zk.ceremony.g2_subgroup_check(d) == -1
    bad = _twist_point(77)
    pts[2000] = bad
    pts[2777] = bad
    pts[5] = 0                                                    # infinity is a member
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    assert zk.ceremony.g2_subgroup_check(d) == 2000


@pytest.mark.gpu
@pytest.mark.parametrize("same_scalar", [False, True])
def test_g2_batch_exp_is_the_reference_mul_for_every_point_of_the_twist(zk, worker, same_scalar):
    """the reference's batch_exp is wNAF (pairing/src/wnaf.rs:4-71 under batched_accumulator.rs:1147-1158 and
    phase2/src/parameters.rs:436-464) -- the group law, exact for ANY point of the twist -- and its decoders test the curve equation only
    (ec.rs:133-150).  The default G2 batch_exp (plain windows, no psi split) returns the oracle's k * P for on-twist records with a
    cofactor component, per-point and shared scalar, device and host-buffer entry points; scalars 0 / 1 / r - 1 / r - 2 / mu included
    ((r - 1) P != -P for such a point).  The promise flag is what trades this away: on the SAME vector it returns the oracle's point for the
    subgroup records and a different one for the others."""
    import torch

    n = 48
    pts = _g2_vector_with_cofactor_points(n, seed=4720)
    R = M.R_ORDER
    ks = inputs.random_scalars(n, seed=4721)
    special = [0, 1, R - 1, R - 2, M.Q % R, (1 << 128) + 5]
    for i, v in enumerate(special):
        ks[i] = _limbs(v)
    ks[n // 2] = _limbs(R - 1)                                    # the second cofactor point times r - 1
    if same_scalar:
        ks = np.tile(ks[n - 1], (n, 1))
    want = np.stack([O.G2.to_affine(O.G2.mul(O.G2.from_affine(pts[i]), ks[i])) for i in range(n)])
    d_p, d_k = torch.from_numpy(pts.view(np.int64)).cuda(), torch.from_numpy((ks[:1] if same_scalar else ks).view(np.int64)).cuda()
    got = zk.ceremony.batch_exp(d_p, d_k, same_scalar=same_scalar).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    got_h = zk.ceremony.batch_exp_host(pts, ks[:1] if same_scalar else ks, same_scalar=same_scalar)
    assert np.array_equal(got_h, want)
    promised = zk.ceremony.batch_exp(d_p, d_k, same_scalar=same_scalar, trusted_subgroup=True).cpu().numpy().view(np.uint64)
    outside = {1, n // 2, n - 2}
    for i in range(n):
        if i not in outside:
            assert np.array_equal(promised[i], want[i]), i
    if same_scalar:
        assert any(not np.array_equal(promised[i], want[i]) for i in outside)   # the broken promise shows: that is why it is not the default
    # r * P is NOT infinity for a cofactor point, and the default path says so (the split path would reduce r to zero)
    rP = zk.ceremony.batch_exp(d_p[1:2].contiguous(), torch.from_numpy(_limbs(R).reshape(1, 4).view(np.int64)).cuda(), same_scalar=True).cpu().numpy().view(np.uint64)
    assert rP.any() and np.array_equal(rP[0], O.G2.to_affine(O.G2.mul(O.G2.from_affine(pts[1]), _limbs(R))))


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["fft", "ifft"])
def test_g2_point_fft_is_exact_for_points_outside_the_subgroup(zk, worker, op):
    """EvaluationDomain<Point<G2>>::{fft, ifft} (bellman/src/group.rs:38-51: `group_mul_assign` is `mul_assign`, the group law) on a vector
    holding on-twist, out-of-subgroup records: the oracle's bytes."""
    import torch

    log_n = 5
    pts = _g2_vector_with_cofactor_points(1 << log_n, seed=4730)
    want = O.point_domain_op(2, pts, log_n, op)
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    got = (zk.ceremony.point_ifft(d) if op == "ifft" else zk.ceremony.point_fft(d)).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [9, 40])
def test_g2_sparse_matvec_is_exact_for_points_outside_the_subgroup(zk, worker, rows):
    """The QAP sums of MPCParameters::new (phase2/src/parameters.rs:281-294: `coeffs_g2[lag].mul(coeff)` added up) over bases with a cofactor
    component; coefficients 0 / 1 / r - 1 (a full multiplication here: (r - 1) P != -P) / general.  Device and host-buffer forms."""
    import torch

    n = 24
    pts = _g2_vector_with_cofactor_points(n, seed=4740)
    # rows = 9: fewer than 2 terms per base -- every term through the plain windows; rows = 40: the bases are reused (nnz >= 2 n_bases), so the
    # membership test runs over the bases and the terms split by it (members: psi split and the r - 1 shortcut; the others: plain windows)
    rng = np.random.default_rng(4741 + rows)
    lens = rng.integers(0, 6, rows)
    lens[2] = 0
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, n, nnz).astype(np.int32)
    col[:4] = [1, n // 2, n - 2, 3]                               # the records outside the subgroup (and the infinity) are used
    assert rows == 9 or nnz >= 2 * n
    coeff = inputs.random_scalars(nnz, seed=4742)
    for t, v in enumerate([M.R_ORDER - 1, 1, M.R_ORDER - 1, 7, 0]):
        coeff[t] = _limbs(v)
    if nnz > 8:                                                   # r - 1 and 1 on a MEMBER base too (the shortcut -P / P where it is allowed)
        col[5], col[6] = 0, 4
        coeff[5], coeff[6] = _limbs(M.R_ORDER - 1), _limbs(1)
    want = np.zeros((rows, 16), np.uint64)
    for r in range(rows):
        acc = O.G2.from_affine(np.zeros(16, np.uint64))
        for t in range(row_ptr[r], row_ptr[r + 1]):
            acc = O.G2.add(acc, O.G2.mul(O.G2.from_affine(pts[col[t]]), coeff[t]))
        want[r] = O.G2.to_affine(acc)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else a.dtype)).cuda()  # noqa: E731
    got = zk.ceremony.eval_qap(dev(pts), dev(row_ptr), dev(col), dev(coeff)).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    got_h = zk.ceremony.eval_qap_host(pts, row_ptr.astype(np.uint32), col.astype(np.uint32), coeff)
    assert np.array_equal(got_h, want)


@pytest.mark.gpu
@pytest.mark.parametrize("group", [1, 2])
def test_batch_exp_of_records_that_are_on_no_curve(zk, worker, group):
    """`compute_constrained` reads its challenge with CheckForCorrectness::No (powersoftau/src/bin/compute_constrained.rs:16): a coordinate
    pair that satisfies no curve equation is admitted and multiplied by wNAF -- the chord-and-tangent law of y^2 = x^3 + (y0^2 - x0^3), which
    no formula names.  batch_exp returns the oracle's mul_assign for such records too (G1: the on-curve test in front of the phi split sends
    them through the plain windows; G2: the plain windows are the default), including a record with y == 0 (a point of order two there:
    2 P is infinity, ec.rs:301-358 gives Z3 = 0) and one with x == 0; honest records in the same call are unaffected."""
    import torch

    G = O.G1 if group == 1 else O.G2
    w = 8 * group
    n = 40
    pts = inputs.bases_progression_cpu(group, n, seed=4750 + group)
    rng = np.random.default_rng(4751)
    mont = lambda v: np.array(M.to_limbs(M.to_mont(v, M.Q)), dtype=np.uint64)  # noqa: E731
    def rec(*coords):
        return np.concatenate([mont(c) for c in coords])
    off = {}
    for i in (2, 9, 17, 30):
        off[i] = rec(*[int(rng.integers(1, 1 << 62)) for _ in range(2 * group)])      # random small coordinates: on no curve
    off[5] = rec(12345, 0) if group == 1 else rec(12345, 678, 0, 0)                   # y == 0
    off[6] = rec(0, 777) if group == 1 else rec(0, 0, 777, 3)                         # x == 0
    for i, p in off.items():
        pts[i] = p
    ks = inputs.random_scalars(n, seed=4752)
    ks[5] = _limbs(7)                                             # odd multiple of the order-two record: the record itself
    ks[2] = _limbs(M.R_ORDER - 1)
    for same in (False, True):
        kk = np.tile(ks[9], (n, 1)) if same else ks
        want = np.stack([G.to_affine(G.mul(G.from_affine(pts[i]), kk[i])) for i in range(n)])
        d_p, d_k = torch.from_numpy(pts.view(np.int64)).cuda(), torch.from_numpy((kk[:1] if same else kk).view(np.int64)).cuda()
        got = zk.ceremony.batch_exp(d_p, d_k, same_scalar=same).cpu().numpy().view(np.uint64)
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (group, same, i, i in off)
    even = zk.ceremony.batch_exp(torch.from_numpy(pts[5:6].view(np.int64)).cuda(), torch.from_numpy(_limbs(6).reshape(1, 4).view(np.int64)).cuda()).cpu().numpy()
    assert not even.any()                                         # 6 * (x, 0) = infinity


@pytest.mark.gpu
def test_g2_batch_mul_of_a_base_outside_the_subgroup(zk, worker):
    """mi355zk_bn254_g2_batch_mul_dev (one base by value, many scalars): the base's membership is decided on the host and only a member
    takes the split kernel."""
    import torch

    bad = _twist_point(77)
    ks = inputs.random_scalars(33, seed=4760)
    ks[0] = _limbs(M.R_ORDER - 1)
    out = torch.empty((33, 16), dtype=torch.int64, device="cuda")
    fn = zk.lib.load().mi355zk_bn254_g2_batch_mul_dev
    for base in (bad, inputs.G2_GEN_RAW):
        b = np.ascontiguousarray(base)
        assert fn(C.c_void_p(out.data_ptr()), b.ctypes.data_as(C.c_void_p), C.c_void_p(torch.from_numpy(ks.view(np.int64)).cuda().data_ptr()), 33, None) == 0
        got = out.cpu().numpy().view(np.uint64)
        for i in range(33):
            assert np.array_equal(got[i], O.G2.to_affine(O.G2.mul(O.G2.from_affine(b), ks[i]))), i


@pytest.mark.gpu
def test_table_mode_is_exact_for_a_g2_point_outside_the_subgroup(zk, worker):
    """Round 4: the window table is built by PLAIN doublings, so table mode agrees with the plain bucket call and with the
    reference's multiexp (exact for every point of the twist) even when a base lies outside the order-r subgroup -- rounds 2-3 built the
    table through the psi split and returned a different point for such a vector."""
    import torch

    n = 700
    pts = inputs.bases_progression_cpu(2, n, seed=4713)
    bad = _twist_point(77)
    pts[123] = bad
    pts[600] = bad
    sc = inputs.random_scalars(n, seed=4714)
    rc, want = O.G2.multiexp(pts, sc, threads=4)
    assert rc == 0
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    plain = zk.multiexp(worker, (d_pts, 0), zk.FullDensity(), d_sc).wait()
    table = zk.MsmTable(d_pts)
    tab = zk.multiexp(worker, (table, 0), zk.FullDensity(), d_sc).wait()
    assert np.array_equal(O.G2.to_affine(plain), O.G2.to_affine(want))
    assert np.array_equal(O.G2.to_affine(tab), O.G2.to_affine(want))
    # the table's records of the cofactor point are its doublings (oracle: plain double-and-add)
    recs = table.table.cpu().numpy().view(np.uint64).reshape(table.n_windows, n, 16)
    c, W = table.window_bits, table.n_windows
    rest = 254 - (c - 1)
    widths = [rest // (W - 1) + (1 if w < rest % (W - 1) else 0) for w in range(W - 1)]
    shift = 0
    for w in range(min(W, 4)):
        k = np.array(M.to_limbs(1 << shift), dtype=np.uint64)
        assert np.array_equal(recs[w, 123], O.G2.to_affine(O.G2.mul(O.G2.from_affine(bad), k))), w
        if w < W - 1:
            shift += widths[w]


@pytest.mark.gpu
@pytest.mark.parametrize("same_scalar", [False, True])
def test_plain_and_split_kernels_agree_on_subgroup_points_at_size(zk, worker, same_scalar):
    """2^15 subgroup points (k_i * G2): the default (plain windows) and the promised (psi split) kernels return the same records, and both the
    closed form (s_i k_i mod r) * G2 -- checked on every record through batch_mul of the generator, and on a sample against the oracle's mul."""
    import torch

    n = 1 << 15
    import bench

    dev = torch.device("cuda", 0)
    k = bench.gen_scalars(n, 4801, dev)
    s = bench.gen_scalars(1 if same_scalar else n, 4802, dev)
    L = zk.lib.load()
    gen = np.ascontiguousarray(inputs.G2_GEN_RAW)
    pts = torch.empty((n, 16), dtype=torch.int64, device=dev)
    assert L.mi355zk_bn254_g2_batch_mul_dev(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    plain = zk.ceremony.batch_exp(pts, s, same_scalar=same_scalar)
    split = zk.ceremony.batch_exp(pts, s, same_scalar=same_scalar, trusted_subgroup=True)
    assert torch.equal(plain, split)
    to_int = lambda a: sum(a[:, i].astype(object) << (64 * i) for i in range(4))  # noqa: E731
    hk, hs = to_int(k.cpu().numpy().view(np.uint64)), to_int(s.cpu().numpy().view(np.uint64))
    prod = np.array([M.to_limbs(int(kk) * int(hs[0 if same_scalar else i]) % M.R_ORDER) for i, kk in enumerate(hk)], dtype=np.uint64)
    want = torch.empty_like(pts)
    d_prod = torch.from_numpy(prod.view(np.int64)).to(dev)
    assert L.mi355zk_bn254_g2_batch_mul_dev(C.c_void_p(want.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(d_prod.data_ptr()), n, None) == 0
    assert torch.equal(plain, want)
    got = plain.cpu().numpy().view(np.uint64)
    for i in (0, 1, n // 3, n - 1):
        assert np.array_equal(got[i], O.G2.to_affine(O.G2.mul(O.G2.from_affine(inputs.G2_GEN_RAW), prod[i]))), i
