"""The single-process multi-GPU mode of the C ABI (mi355zk_init with n_devices > 1, include/mi355zk.h): a host-buffer multiexp is cut
into one point range per device, every cell runs the single-GPU pipeline from its own host thread, and the Jacobian partials are
joined on the host.  The consumer it exists for is ONE Rust process (phase2/src/bin/prove.rs -> bellman/src/groth16/prover.rs:250-298
-> multiexp.rs:330-355).  On a one-GPU box the device set repeats id 0 ({0,0}, {0,0,0,0}, {0} x 8): logical devices sharing a GPU,
the same trick the gloo tests use for ranks.  Everything is compared with the CPU oracle: affine result, SynthesisError and index."""
import ctypes as C

import numpy as np
import pytest

import golden_util as GU
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def multi(zk, worker, monkeypatch):
    """factory: a Worker over `k` logical devices (all physical device 0), cutting calls from 64 exponents on"""
    monkeypatch.setenv("MI355ZK_MULTI_MIN_LOG", "6")

    def make(k):
        w = zk.Worker(devices=[0] * k)
        assert zk.lib.load().mi355zk_device_count() == k
        return w

    yield make
    zk.unpin_bases(None)
    zk.Worker(0)  # back to one device for the tests that follow
    assert zk.lib.load().mi355zk_device_count() == 1


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("k", [2, 4, 8])
def test_cells_over_repeated_device_ids_match_the_oracle(zk, multi, group, k):
    G = O.G1 if group == 1 else O.G2
    n, off = 3000, 5
    rng = np.random.default_rng(4100 + k + group)
    bits = rng.random(n) < 0.6
    sel = np.nonzero(bits)[0]
    bases = inputs.bases_progression_cpu(group, len(sel) + off, seed=4101 + group)
    scalars = inputs.random_scalars(n, seed=4102 + k)
    scalars[::17] = 0
    scalars[5::23] = np.array([1, 0, 0, 0], dtype=np.uint64)
    dens = GU.density_words(bits)
    w = multi(k)
    # density map + source offset: every cell starts at the prefix popcount of its first exponent
    rc, want = G.multiexp(bases, scalars, density=dens, density_bits=n, base_offset=off, threads=4)
    assert rc == 0
    dm = zk.DensityTracker.from_bools(bits)
    got = zk.multiexp(w, (bases, off), dm, scalars).wait()                      # unpinned: every cell uploads its slice of the bases
    assert np.array_equal(G.to_affine(got), G.to_affine(want))
    zk.pin_bases(bases)
    for _ in range(2):                                                          # pinned: kept on the device(s) after the first call
        assert np.array_equal(G.to_affine(zk.multiexp(w, (bases, off), dm, scalars).wait()), G.to_affine(want))
    zk.unpin_bases(bases)
    # FullDensity over a prefix
    m = len(sel)
    rc, want = G.multiexp(bases, scalars[:m], threads=4)
    assert rc == 0
    assert np.array_equal(G.to_affine(zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars[:m]).wait()), G.to_affine(want))


@pytest.mark.parametrize("k", [2, 8])
def test_errors_carry_the_global_exponent_index(zk, multi, k):
    """UnexpectedIdentity in a later cell, two identities (the lower exponent wins whatever cell finishes first), Eof planned for
    the whole call, identity before Eof, a non-canonical exponent: code and index of the single-device call (and of the oracle)."""
    n, off = 2500, 3
    rng = np.random.default_rng(4200 + k)
    bits = rng.random(n) < 0.7
    sel = np.nonzero(bits)[0]
    bases = inputs.bases_progression_cpu(1, len(sel) + off, seed=4201)
    scalars = inputs.random_scalars(n, seed=4202)
    dens = GU.density_words(bits)
    dm = zk.DensityTracker.from_bools(bits)
    w = multi(k)
    bad = bases.copy()
    r_hi, r_lo = len(sel) * 7 // 8, len(sel) * 5 // 8
    bad[off + r_hi] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bad, off), dm, scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == int(sel[r_hi])
    bad[off + r_lo] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bad, off), dm, scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == int(sel[r_lo])
    assert O.G1.multiexp(bad, scalars, density=dens, density_bits=n, base_offset=off)[0] == 1
    # a zero exponent skips its identity base without looking at it (multiexp.rs:95-96)
    sc = scalars.copy()
    sc[sel[r_lo]] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bad, off), dm, sc).wait()
    assert e.value.index == int(sel[r_hi])
    # FullDensity with more exponents than bases: Eof at the first exponent without a base
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bases, off), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == len(sel)
    # ... and an identity among the exponents before it wins
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bad, off), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == r_lo
    # a non-canonical exponent in the last cell
    sc = scalars.copy()
    t = int(sel[-3])   # (a selected one: an exponent the density map skips is not looked at)
    sc[t, 3] |= np.uint64(1 << 62)
    with pytest.raises(ValueError):
        zk.multiexp(w, (bases, off), dm, sc).wait()
    assert zk.lib.load().mi355zk_last_error_index() == t


@pytest.mark.parametrize("plan", ["2x2", "1x4", "2x4"])
def test_window_group_cells(zk, multi, monkeypatch, plan):
    """MI355ZK_MULTI_PLAN: point ranges x groups of scalar windows (shard.py's cells) inside one process."""
    monkeypatch.setenv("MI355ZK_MULTI_PLAN", plan)
    n = 5000
    bases = inputs.bases_progression_cpu(1, n, seed=4301)
    scalars = inputs.random_scalars(n, seed=4302)
    rc, want = O.G1.multiexp(bases, scalars, threads=8)
    assert rc == 0
    w = multi(8)
    assert np.array_equal(O.G1.to_affine(zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()), O.G1.to_affine(want))
    bad = bases.copy()
    bad[4000] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(w, (bad, 0), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == 4000


def test_short_calls_take_the_devices_in_turn_and_concurrent_callers_work(zk, worker, monkeypatch):
    """Below the cutting threshold a call runs whole on the next device of the set; eight host threads at once (the prover's eight
    multiexps, prover.rs:250-298), some cut into cells and some not."""
    import threading

    monkeypatch.setenv("MI355ZK_MULTI_MIN_LOG", "11")
    w = zk.Worker(devices=[0, 0, 0, 0])
    try:
        cases = []
        for t in range(8):
            n = 700 if t % 2 else 4096
            bases = inputs.bases_progression_cpu(1, n, seed=4400 + t)
            scalars = inputs.random_scalars(n, seed=4410 + t)
            rc, want = O.G1.multiexp(bases, scalars, threads=4)
            assert rc == 0
            cases.append((bases, scalars, O.G1.to_affine(want)))
        got = [None] * 8

        def run(t):
            got[t] = O.G1.to_affine(zk.multiexp(w, (cases[t][0], 0), zk.FullDensity(), cases[t][1]).wait())

        th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for t in range(8):
            assert np.array_equal(got[t], cases[t][2]), t
    finally:
        zk.Worker(0)


def test_init_rejects_devices_that_do_not_exist(zk, worker):
    lib = zk.lib.load()
    ids = (C.c_int * 2)(0, 1 << 20)
    assert lib.mi355zk_init(ids, 2) == zk.lib.ERR_BAD_ARGS
    assert lib.mi355zk_init(None, 2) == zk.lib.ERR_BAD_ARGS
    assert lib.mi355zk_device_count() == 1   # a rejected set changes nothing


def test_eight_cells_at_2e22_same_point_as_one_device(zk, worker, monkeypatch):
    """At a size where the cells are real work (2^19 exponents each, streamed upload, pinned bases): the same affine point as the
    device-resident single-GPU call, with and without a density map."""
    import torch

    import bench

    log_n = 22
    n = 1 << log_n
    dev = torch.device("cuda", 0)
    k = bench.gen_scalars(n, 4501, dev)
    d_bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert zk.lib.load().mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(d_bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    d_scalars = bench.gen_scalars(n, 4502, dev)
    want = O.G1.to_affine(zk.multiexp(worker, (d_bases, 0), zk.FullDensity(), d_scalars).wait())
    rng = np.random.default_rng(4503)
    bits = rng.random(n + 1000) < 0.9
    m = int(np.searchsorted(np.cumsum(bits), n))  # exponents that consume (almost) all bases
    bits = bits[:m]
    dm = zk.DensityTracker.from_bools(bits)
    d_sc2 = bench.gen_scalars(m, 4504, dev)
    want_d = O.G1.to_affine(zk.multiexp(worker, (d_bases, 0), dm, d_sc2).wait())
    h_bases = d_bases.cpu().numpy().view(np.uint64)
    h_scalars = d_scalars.cpu().numpy().view(np.uint64)
    h_sc2 = d_sc2.cpu().numpy().view(np.uint64)
    del d_bases, d_scalars, d_sc2, k
    w = zk.Worker(devices=[0] * 8)
    try:
        zk.pin_bases(h_bases)
        for _ in range(2):
            assert np.array_equal(O.G1.to_affine(zk.multiexp(w, (h_bases, 0), zk.FullDensity(), h_scalars).wait()), want)
        assert np.array_equal(O.G1.to_affine(zk.multiexp(w, (h_bases, 0), dm, h_sc2).wait()), want_d)
    finally:
        zk.unpin_bases(None)
        zk.Worker(0)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("k", [1, 3, 8])
def test_batch_exp_on_host_buffers_over_the_device_set(zk, worker, group, k):
    """mi355zk_bn254_g{1,2}_batch_exp: phase2 `contribute` / powersoftau `batch_exp` for a single-process caller -- the points shard by
    contiguous range over the device set, no exchange.  Per-point scalars and one shared scalar, an infinity record among the points,
    against the oracle's mul_assign + into_affine; identical for 1, 3 and 8 logical devices."""
    G = O.G1 if group == 1 else O.G2
    n = 4200 if group == 1 else 3100          # (>= 1024 points per range, or the call is not cut)
    bases = inputs.bases_progression_cpu(group, n, seed=4600 + group)
    bases[17] = 0
    sc = inputs.random_scalars(n, seed=4601)
    sc[5] = 0
    sc[6] = np.array([1, 0, 0, 0], dtype=np.uint64)
    w = zk.Worker(devices=[0] * k) if k > 1 else zk.Worker(0)
    try:
        got = zk.ceremony.batch_exp_host(bases, sc)
        one = zk.ceremony.batch_exp_host(bases, sc[3:4], same_scalar=True)
    finally:
        zk.Worker(0)
    idx = list(range(0, 40)) + list(range(n // 3 - 5, n // 3 + 5)) + list(range(n - 20, n))   # around every range boundary for k = 3
    for i in idx:
        assert np.array_equal(got[i], G.to_affine(G.mul(G.from_affine(bases[i]), sc[i]))), i
        assert np.array_equal(one[i], G.to_affine(G.mul(G.from_affine(bases[i]), sc[3]))), i
    # the whole vector against the device-resident entry point
    import torch

    d = zk.ceremony.batch_exp(torch.from_numpy(bases.view(np.int64)).cuda(), torch.from_numpy(sc.view(np.int64)).cuda())
    assert np.array_equal(d.cpu().numpy().view(np.uint64), got)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("k", [1, 3])
def test_merge_pairs_on_host_buffers_over_the_device_set(zk, worker, group, k, monkeypatch):
    """mi355zk_bn254_g{1,2}_merge_pairs / _dense_multiexp on host buffers (powersoftau's verification multiexps for a single-process caller):
    the vectors are cut into pieces (MI355ZK_DENSE_PIECE_TEST: 700 points instead of 2^22), dealt to two host threads per device, and the
    partials are added on the host.  power_pairs shape (v2 = v1 shifted by one record, overlapping views), an infinity entry, a zero
    scalar -- against the oracle."""
    monkeypatch.setenv("MI355ZK_DENSE_PIECE_TEST", "700")
    G = O.G1 if group == 1 else O.G2
    n = 5000 if group == 1 else 1500
    v = inputs.bases_progression_cpu(group, n + 1, seed=4700 + group)
    v[1234] = 0
    rho = inputs.random_scalars(n, seed=4701)
    rho[5] = 0

    def ref(bases):
        sc = rho.copy()
        sc[~bases.any(axis=1)] = 0
        rc, out = G.multiexp(bases, sc, threads=4)
        assert rc == 0
        return G.to_affine(out)

    zk.Worker(devices=[0] * k) if k > 1 else zk.Worker(0)
    try:
        s, sx = zk.ceremony.merge_pairs_host(v[:n], v[1:], rho)
        one = zk.ceremony.dense_multiexp_host(v[:n], rho)
        empty = zk.ceremony.dense_multiexp_host(v[:0], rho[:0])
    finally:
        zk.Worker(0)
    assert np.array_equal(G.to_affine(s), ref(v[:n])) and np.array_equal(G.to_affine(sx), ref(v[1:n + 1]))
    assert np.array_equal(G.to_affine(one), ref(v[:n]))
    assert not empty[8 * group:].any()   # Z == 0


def test_merge_pairs_host_at_2e23_matches_the_device_resident_call(zk, worker):
    """Two real pieces of 2^22 points through two host threads: the same affine sums as ONE device-resident merge_pairs call."""
    import torch

    import bench

    n = (1 << 23) - 1
    dev = torch.device("cuda", 0)
    k = bench.gen_scalars(n + 1, 4801, dev)
    d_v = torch.empty((n + 1, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert zk.lib.load().mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(d_v.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n + 1, None) == 0
    d_rho = bench.gen_scalars(n, 4802, dev)
    want_s, want_sx = zk.ceremony.power_pairs(d_v, d_rho)
    h_v = d_v.cpu().numpy().view(np.uint64)
    h_rho = d_rho.cpu().numpy().view(np.uint64)
    s, sx = zk.ceremony.merge_pairs_host(h_v[:n], h_v[1:], h_rho)
    assert np.array_equal(O.G1.to_affine(s), O.G1.to_affine(want_s)) and np.array_equal(O.G1.to_affine(sx), O.G1.to_affine(want_sx))


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("k", [1, 3])
def test_qap_evaluation_on_host_buffers_over_the_device_set(zk, worker, group, k):
    """mi355zk_bn254_g{1,2}_sparse_matvec: the per-variable sums of MPCParameters::new for a single-process caller -- every device of the set
    takes a contiguous range of rows.  600 rows of 0 .. 6 terms (and one long row), +-1 / zero coefficients, an infinity base: the same
    records as the device-resident call, sampled rows against the oracle; a bad column index is bad arguments from whichever range holds it."""
    import torch

    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    nb = 96
    bases = inputs.bases_progression_cpu(group, nb, seed=4900 + group)
    bases[7] = 0
    rng = np.random.default_rng(4901)
    n_rows = 600
    row_len = rng.integers(0, 7, size=n_rows)
    row_len[311] = 200
    row_ptr = np.concatenate([[0], np.cumsum(row_len)]).astype(np.uint32)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, nb, size=nnz).astype(np.uint32)
    coeff = inputs.random_scalars(nnz, seed=4902)
    small = rng.integers(0, 4, size=nnz)
    coeff[small == 0] = np.array([1, 0, 0, 0], dtype=np.uint64)
    coeff[small == 1] = np.array(M.to_limbs(M.R_ORDER - 1), dtype=np.uint64)
    coeff[11] = 0
    zk.Worker(devices=[0] * k) if k > 1 else zk.Worker(0)
    try:
        got = zk.ceremony.eval_qap_host(bases, row_ptr, col, coeff)
        bad_col = col.copy()
        bad_col[nnz - 3] = nb
        with pytest.raises(ValueError):
            zk.ceremony.eval_qap_host(bases, row_ptr, bad_col, coeff)
    finally:
        zk.Worker(0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else np.int32)).cuda()  # noqa: E731
    want = zk.ceremony.eval_qap(d(bases), d(row_ptr), d(col), d(coeff)).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    for r in list(range(0, 12)) + [199, 200, 201, 311, 399, 400, 401, n_rows - 1]:
        acc = G.from_affine(np.zeros(G.aff, np.uint64))
        for t in range(row_ptr[r], row_ptr[r + 1]):
            acc = G.add(acc, G.mul(G.from_affine(bases[col[t]]), coeff[t]))
        assert np.array_equal(got[r], G.to_affine(acc)), r


def test_each_device_keeps_only_the_slice_its_cell_consumes(zk, multi):
    """"the tau-table slice stays resident on its GPU".  A pinned vector that multi-GPU calls run over is cached as one slice per
    (logical) device -- n / N records each, together the vector once -- not as a whole copy per device (round 4: N x the vector); with a
    density map the slices are the prefix-popcount ranges.  A short call (below the cutting threshold) runs whole on ONE device and the
    next short calls over the same vector go to the device that holds it."""
    lib = zk.lib.load()
    n = 4096
    bases = inputs.bases_progression_cpu(1, n, seed=4601)
    scalars = inputs.random_scalars(n, seed=4602)
    rc, want = O.G1.multiexp(bases, scalars, threads=4)
    assert rc == 0
    d, t = C.c_size_t(0), C.c_size_t(0)
    for k in (2, 4, 8):
        w = multi(k)
        zk.pin_bases(bases)
        for _ in range(2):
            assert np.array_equal(O.G1.to_affine(zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()), O.G1.to_affine(want))
        assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 1
        assert d.value == n * 64, (k, d.value)                               # the vector ONCE over all devices, in k slices
        zk.unpin_bases(bases)
        assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 0
    # density map: the cells' slices follow the prefix popcounts and add up to the consumed bases
    rng = np.random.default_rng(4603)
    bits = rng.random(n) < 0.5
    m = int(bits.sum())
    dm = zk.DensityTracker.from_bools(bits)
    rc, want = O.G1.multiexp(bases[:m], scalars, density=GU.density_words(bits), density_bits=n, threads=4)
    assert rc == 0
    w = multi(4)
    b = np.ascontiguousarray(bases[:m])
    zk.pin_bases(b)
    assert np.array_equal(O.G1.to_affine(zk.multiexp(w, (b, 0), dm, scalars).wait()), O.G1.to_affine(want))
    assert lib.mi355zk_bases_cache_info(b.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 1 and d.value == m * 64
    zk.unpin_bases(None)


def test_invalidate_reaches_every_slice_of_an_implicitly_cached_vector():
    """under MI355ZK_BASES_CACHE_IMPLICIT=1 (every vector treated as pinned) a multi-device call caches one SLICE per device of a vector
    that is in no pin list; mi355zk_bases_cache_invalidate(vector) must reach all of them -- round 5 left the slices on devices 1 .. N-1 cached, to be
    caught by the sampled fingerprint only.  The implicit mode is read once per process, so this runs in a child: cache the slices over three logical
    devices, see them all under the vector's name, rewrite the vector in place, invalidate, see none, and get the new vector's result."""
    import os
    import subprocess
    import sys

    code = r'''
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
os.environ["MI355ZK_BASES_CACHE_IMPLICIT"] = "1"; os.environ["MI355ZK_MULTI_MIN_LOG"] = "6"
import inputs, oracle_lib as O
import phase2_bn254_amd as zk
lib = zk.lib.load()
w = zk.Worker(devices=[0, 0, 0])
n = 3000
bases = inputs.bases_progression_cpu(1, n, seed=4701)
scalars = inputs.random_scalars(n, seed=4702)
d, t = C.c_size_t(0), C.c_size_t(0)
rc, want = O.G1.multiexp(bases, scalars, threads=4)
got = zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))
assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 1 and d.value == n * 64, d.value   # all three slices answer to the vector
bases[:] = inputs.bases_progression_cpu(1, n, seed=4703)      # the caller rewrites its vector in place ...
lib.mi355zk_bases_cache_invalidate(bases.ctypes.data_as(C.c_void_p))   # ... and says so
assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 0, d.value
rc, want = O.G1.multiexp(bases, scalars, threads=4)
got = zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))
print("implicit-slices-ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "implicit-slices-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
