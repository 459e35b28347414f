"""G2 subgroup membership (mi355zk_bn254_g2_subgroup_check_dev, run on the host through mi355zk_selftest_g2_in_subgroup).

The G2 scalar-multiplication kernels split k = k1 + k2 mu and use psi(P) for mu P (glv.hpp), which holds in the order-r subgroup
only; the reference's wNAF `mul` is exact for every point of the twist and its bn256 decoders only check the curve equation
(pairing/src/bn256/ec.rs:1136-1344).  The check psi(P) == mu P (mu P by a plain double-and-add) is what a caller with untrusted
G2 data uses to establish the precondition stated in include/mi355zk.h."""
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


@pytest.mark.gpu
def test_subgroup_check_on_device_and_why_it_matters(zk, worker):
    """The device check finds the lowest offending record; and the precondition is real: for the cofactor point the split-based
    batch_exp does NOT return the reference's k * P (oracle mul_assign), for subgroup points it does."""
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
    k = np.array([M.to_limbs(0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % M.R_ORDER)], dtype=np.uint64)
    got = zk.ceremony.batch_exp(d[1999:2001].contiguous(), torch.from_numpy(k.view(np.int64)).cuda(), same_scalar=True).cpu().numpy().view(np.uint64)
    want = np.stack([O.G2.to_affine(O.G2.mul(O.G2.from_affine(pts[i]), k[0])) for i in (1999, 2000)])
    assert np.array_equal(got[0], want[0])                        # in the subgroup: the reference's answer
    assert not np.array_equal(got[1], want[1])                    # outside: psi(P) != mu P, the stated precondition


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
