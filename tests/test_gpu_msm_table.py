"""GPU parity tests for TABLE MODE (mi355zk_bn254_g{1,2}_msm_table_build_dev / _msm_table_dev, include/mi355zk.h): the multiexp of
bellman/src/multiexp.rs:330-355 over a precomputed window table of the base vector -- same result (bit-exact after affine
normalisation) and the same Source / density errors (source.rs:44-118) as the plain call and the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import golden_util as GU
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _table(zk, bases):
    return zk.MsmTable(_dev(bases))


@pytest.mark.parametrize("n", [1, 2, 31, 33, 1000, 5000, 1 << 14])
def test_g1_table_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(1, n, seed=n + 3)
    scalars = inputs.random_scalars(n, seed=7 * n + 2)
    rc, want = O.G1.multiexp(bases, scalars, threads=8)
    assert rc == 0
    got = zk.multiexp(worker, (_table(zk, bases), 0), zk.FullDensity(), _dev(scalars)).wait()
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))


@pytest.mark.parametrize("n", [1, 33, 500, 4096, 1 << 14])
def test_g2_table_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(2, n, seed=n + 5)
    scalars = inputs.random_scalars(n, seed=11 * n + 2)
    rc, want = O.G2.multiexp(bases, scalars, threads=8)
    assert rc == 0
    got = zk.multiexp(worker, (_table(zk, bases), 0), zk.FullDensity(), _dev(scalars)).wait()
    assert np.array_equal(O.G2.to_affine(got), O.G2.to_affine(want))


@pytest.mark.parametrize("group", [1, 2])
def test_table_records_are_the_shifted_bases(zk, worker, group):
    """table[w * n + i] = 2^shift_w * bases[i]: every window of a small table against the oracle's scalar multiplication; the
    identity stays the identity."""
    G = O.G1 if group == 1 else O.G2
    n = 7
    bases = inputs.bases_progression_cpu(group, n, seed=91)
    bases[4] = 0
    t = _table(zk, bases)
    tab = t.table.cpu().numpy().view(np.uint64).reshape(t.n_windows, n, -1)
    assert np.array_equal(tab[0], bases)
    # the layout make_geom gives c-bit windows: 254 - (c - 1) bits spread over the W - 1 lower windows
    c, W = t.window_bits, t.n_windows
    rest = 254 - (c - 1)
    base, rem = divmod(rest, W - 1)
    shift = 0
    for w in range(1, W):
        shift += base + (1 if w - 1 < rem else 0)
        k = np.array([(1 << shift >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
        for i in (0, 3, 4, 6):
            want = G.to_affine(G.mul(G.from_affine(bases[i]), k)) if bases[i].any() else np.zeros_like(bases[i])
            assert np.array_equal(tab[w, i], want), (w, i)


@pytest.mark.parametrize("group", [1, 2])
def test_table_density_offset_and_trivial_exponents(zk, worker, group):
    G = O.G1 if group == 1 else O.G2
    n = 3000 if group == 1 else 700
    rng = np.random.default_rng(15)
    bits = rng.integers(0, 2, size=n).astype(bool)
    used = int(bits.sum())
    bases = inputs.bases_progression_cpu(group, used + 5, seed=33)
    scalars = inputs.random_scalars(n, seed=34)
    scalars[::17] = 0
    scalars[5::23] = np.array([1, 0, 0, 0], dtype=np.uint64)
    rc, want = G.multiexp(bases, scalars, density=GU.density_words(bits), density_bits=n, base_offset=5)
    assert rc == 0
    got = zk.multiexp(worker, (_table(zk, bases), 5), zk.DensityTracker.from_bools(bits), _dev(scalars)).wait()
    assert np.array_equal(G.to_affine(got), G.to_affine(want))


def test_table_errors_name_the_lowest_exponent(zk, worker):
    """The accumulation meets an identity by its TABLE index, which orders by window first; the error must name the lowest
    EXPONENT (source.rs:50-52).  Exponent 5 has digits in a high window only, exponent 9 in window 0: 5 is reported."""
    n = 40
    bases = inputs.bases_progression_cpu(1, n - 2, seed=5)
    scalars = inputs.random_scalars(n, seed=6)
    scalars[5] = np.array([0, 0, 0, 1 << 8], dtype=np.uint64)   # 2^200
    scalars[9] = np.array([1, 0, 0, 0], dtype=np.uint64)
    bad = bases.copy()
    bad[5] = 0
    bad[9] = 0
    t = _table(zk, bad)
    with pytest.raises(zk.SynthesisError) as e:   # identity at 5 beats identity at 9 and Eof at n - 2
        zk.multiexp(worker, (t, 0), zk.FullDensity(), _dev(scalars)).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == 5
    sc = scalars.copy()
    sc[5] = 0   # zero exponent: the identity base is skipped without being looked at (multiexp.rs:95-96)
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (t, 0), zk.FullDensity(), _dev(sc)).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == 9
    sc[9] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (t, 0), zk.FullDensity(), _dev(sc)).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == n - 2
    rc, _ = O.G1.multiexp(bad, sc)
    assert rc == 2


@pytest.mark.parametrize("group", [1, 2])
def test_table_identity_error_index_under_a_density_map(zk, worker, group):
    G = O.G1 if group == 1 else O.G2
    n = 200
    rng = np.random.default_rng(78)
    bits = rng.random(n) < 0.5
    sel = np.nonzero(bits)[0]
    bases = inputs.bases_progression_cpu(group, len(sel) + 3, seed=43)
    scalars = inputs.random_scalars(n, seed=44)
    dm = zk.DensityTracker.from_bools(bits)
    bad = bases.copy()
    bad[3 + 70] = 0
    bad[3 + 41] = 0   # owned by exponent sel[41]: reported
    bad[3 + 10] = 0   # owned by exponent sel[10], whose scalar is zero: skipped
    bad[1] = 0        # below the source offset: never read
    sc = scalars.copy()
    sc[sel[10]] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (_table(zk, bad), 3), dm, _dev(sc)).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == int(sel[41])
    rc, want = G.multiexp(bases, sc, density=GU.density_words(bits), density_bits=n, base_offset=3)
    got = zk.multiexp(worker, (_table(zk, bases), 3), dm, _dev(sc)).wait()
    assert rc == 0 and np.array_equal(G.to_affine(got), G.to_affine(want))


def test_table_montgomery_scalars_and_noncanonical_exponent(zk, worker):
    import torch

    n = 2000
    bases = inputs.bases_progression_cpu(1, n, seed=61)
    mont = inputs.random_fr_mont(n, seed=62)
    canon = O.fr_into_repr(mont) if hasattr(O, "fr_into_repr") else None
    t = _table(zk, bases)
    got = zk.multiexp(worker, (t, 0), zk.FullDensity(), _dev(mont), scalars_montgomery=True).wait()
    plain = zk.multiexp(worker, (_dev(bases), 0), zk.FullDensity(), _dev(mont), scalars_montgomery=True).wait()
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(plain))
    if canon is not None:
        rc, want = O.G1.multiexp(bases, canon, threads=8)
        assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))
    bad = inputs.random_scalars(n, seed=63)
    bad[77, 3] |= np.uint64(1 << 62)   # >= 2^254: not a canonical FrRepr
    fut = zk.multiexp(worker, (t, 0), zk.FullDensity(), _dev(bad))
    with pytest.raises(ValueError):
        fut.wait()


@pytest.mark.parametrize("group,log_n", [(1, 16), (1, 20), (2, 16), (2, 18)])
def test_table_equals_plain_at_size_with_prover_like_exponents(zk, worker, group, log_n):
    """At the sizes table mode is for, uniform and witness-shaped exponents (40 % ones, 30 % zeros, 10 % bytes: the heavy-bucket
    path in ONE shared bucket set): the same affine point as the plain device-resident call, and as the host-buffer call on a
    2^12 prefix checked against the oracle."""
    import torch
    import bench

    dev = torch.device("cuda", 0)
    n = 1 << log_n
    lib = zk.lib.load()
    k = bench.gen_scalars(n, 71, dev)
    b = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    fn = lib.mi355zk_bn254_g1_batch_mul_dev if group == 1 else lib.mi355zk_bn254_g2_batch_mul_dev
    assert fn(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    G = O.G1 if group == 1 else O.G2
    t = zk.MsmTable(b)
    s = bench.gen_scalars(n, 72, dev)
    g = torch.Generator(device=dev); g.manual_seed(73)
    kind = torch.randint(0, 10, (n,), device=dev, generator=g)
    w = s.clone()
    w[kind < 3] = 0
    w[(kind >= 3) & (kind < 7)] = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=dev)
    byt = (kind == 7)
    w[byt, 1:] = 0
    w[byt, 0] = w[byt, 0] & 0xFF
    for sc in (s, w):
        got = zk.multiexp(worker, (t, 0), zk.FullDensity(), sc).wait()
        want = zk.multiexp(worker, (b, 0), zk.FullDensity(), sc).wait()
        assert np.array_equal(G.to_affine(got), G.to_affine(want))
    m = 1 << 12
    hb = b[:m].cpu().numpy().view(np.uint64)
    hs = w[:m].cpu().numpy().view(np.uint64)
    rc, want = G.multiexp(hb, hs, threads=8)
    got = zk.multiexp(worker, (t, 0), zk.FullDensity(), w[:m].contiguous()).wait()   # fewer exponents than bases: a prefix
    assert rc == 0 and np.array_equal(G.to_affine(got), G.to_affine(want))


@pytest.mark.parametrize("group", [1, 2])
def test_host_entry_runs_pinned_vectors_in_table_mode(zk, worker, group):
    """The host-buffer entry point (what a bellman shim calls with `Arc<Vec<G>>` + `Vec<FrRepr>`): a vector pinned WITH TABLES
    (mi355zk_bases_cache_pin_tables) is uploaded by the first call, gets its window table on the second and is evaluated in table
    mode from then on -- same point as the oracle every time, with a density map and a source offset, and the Source errors
    keep their exponent index; unpinning frees both copies."""
    G = O.G1 if group == 1 else O.G2
    lib = zk.lib.load()
    n = 5000 if group == 1 else 1500
    rng = np.random.default_rng(3100 + group)
    bits = rng.random(n) < 0.6
    sel = np.nonzero(bits)[0]
    bases = np.ascontiguousarray(inputs.bases_progression_cpu(group, len(sel) + 4, seed=3101))
    scalars = inputs.random_scalars(n, seed=3102)
    scalars[::13] = 0
    dm = zk.DensityTracker.from_bools(bits)
    rc, want = G.multiexp(bases, scalars, density=GU.density_words(bits), density_bits=n, base_offset=4)
    assert rc == 0
    want = G.to_affine(want)

    def info():
        d, t = C.c_size_t(), C.c_size_t()
        found = lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t))
        return found, d.value, t.value

    zk.pin_bases(bases, tables=True)
    try:
        assert np.array_equal(G.to_affine(zk.multiexp(worker, (bases, 4), dm, scalars).wait()), want)   # uploads
        assert info()[0] == 1 and info()[2] == 0
        assert np.array_equal(G.to_affine(zk.multiexp(worker, (bases, 4), dm, scalars).wait()), want)   # builds the table, table mode
        found, d_bytes, t_bytes = info()
        c, w = C.c_uint32(), C.c_uint32()
        assert lib.mi355zk_msm_table_geometry(len(bases), group, C.byref(c), C.byref(w)) == 0
        assert found == 1 and t_bytes == w.value * d_bytes
        assert np.array_equal(G.to_affine(zk.multiexp(worker, (bases, 4), dm, scalars).wait()), want)   # table mode again
        # FullDensity over a prefix of the vector, no offset
        rc2, want2 = G.multiexp(bases, scalars[: len(bases)])
        assert rc2 == 0
        assert np.array_equal(G.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars[: len(bases)]).wait()), G.to_affine(want2))
        # more exponents than bases: Eof at the first exponent without a base, after the identity check (none here)
        with pytest.raises(zk.SynthesisError) as e:
            zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
        assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == len(bases)
    finally:
        zk.unpin_bases(bases)
    assert info()[0] == 0


def _to_int(a):
    return [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in a]


@pytest.mark.parametrize("group,log_n", [(1, 23), (2, 20)])
def test_table_mode_closed_form_and_additivity_at_size(zk, worker, group, log_n):
    """Size-independent properties at the sizes table mode is for (the c = 22 layout of 2^23 G1 points; 2^20 G2 points): bases
    k_i * G, so the multiexp has the closed form (sum s_i k_i) * G; additivity over point ranges through the source offset (the
    second range reads the SAME table at an offset); a density map selecting half of the exponents against the plain call."""
    import torch

    import bench
    import bn254_model as M

    dev = torch.device("cuda", 0)
    n = 1 << log_n
    G = O.G1 if group == 1 else O.G2
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    lib = zk.lib.load()
    s = bench.gen_scalars(n, 4101, dev)
    k = bench.gen_scalars(n, 4102, dev)
    b = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
    fn = lib.mi355zk_bn254_g1_batch_mul_dev if group == 1 else lib.mi355zk_bn254_g2_batch_mul_dev
    assert fn(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    t = zk.MsmTable(b)
    total = zk.multiexp(worker, (t, 0), zk.FullDensity(), s).wait()
    hs, hk = s.cpu().numpy().view(np.uint64), k.cpu().numpy().view(np.uint64)
    dot = sum(x * y for x, y in zip(_to_int(hs), _to_int(hk))) % M.R_ORDER
    want = G.mul(G.from_affine(gen), M.to_limbs(dot))
    assert np.array_equal(G.to_affine(total), G.to_affine(want))
    h = n // 2 + 4321
    lo = zk.multiexp(worker, (t, 0), zk.FullDensity(), s[:h].contiguous()).wait()
    hi = zk.multiexp(worker, (t, h), zk.FullDensity(), s[h:].contiguous()).wait()
    assert np.array_equal(G.to_affine(zk.shard.join_partials(np.stack([lo, hi]))), G.to_affine(total))
    rng = np.random.default_rng(4103)
    m = 1 << 18
    bits = rng.random(m) < 0.5
    dm = zk.DensityTracker.from_bools(bits)
    got = zk.multiexp(worker, (t, 7), dm, s[:m].contiguous()).wait()
    ref = zk.multiexp(worker, (b, 7), dm, s[:m].contiguous()).wait()
    assert np.array_equal(G.to_affine(got), G.to_affine(ref))


def test_a_handful_of_exponents_over_a_long_pinned_vector_stays_plain(zk, worker):
    """The prover's input multiexps: a few exponents over the 2^k-point a / b queries of its Parameters.  A table's window width comes
    from the VECTOR's length, so table mode would zero and reduce 2^19 buckets for a handful of points: host-buffer calls with fewer
    than n_bases / 8 exponents stay on the plain path (which picks its window from n) even when the vector is pinned with tables --
    and build no table on their account.  Same result either way."""
    n_bases = 1 << 14
    bases = inputs.bases_progression_cpu(1, n_bases, seed=8801)
    lib = zk.lib.load()
    zk.pin_bases(bases, tables=True)
    try:
        few = inputs.random_scalars(40, seed=8802)
        rc, want = O.G1.multiexp(bases, few, threads=2)
        assert rc == 0
        for _ in range(3):
            assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), few).wait()), O.G1.to_affine(want))
        dev_b, tab_b = C.c_size_t(), C.c_size_t()
        assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(dev_b), C.byref(tab_b)) == 1
        assert dev_b.value == n_bases * 64 and tab_b.value == 0
        many = inputs.random_scalars(n_bases, seed=8803)
        rc, want = O.G1.multiexp(bases, many, threads=8)
        assert rc == 0
        for _ in range(2):
            assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), many).wait()), O.G1.to_affine(want))
        assert lib.mi355zk_bases_cache_info(bases.ctypes.data_as(C.c_void_p), C.byref(dev_b), C.byref(tab_b)) == 1
        assert tab_b.value > dev_b.value    # the full-length call built and used the table
    finally:
        zk.unpin_bases(bases)
