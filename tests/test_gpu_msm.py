"""GPU parity tests for the multiexp path, all through the C ABI (libmi355zk.so):
bit-exact (after affine normalisation, which is how the reference defines equality of projective
points: ec.rs:45-85, 596-629) against the committed golden vectors and the CPU oracle, the Source /
density error contract, and size-independent properties at BASELINE.json's full sizes."""
import ctypes as C

import numpy as np
import pytest

import golden_util as GU
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _density(zk, c):
    return zk.DensityTracker.from_bools(c["density"]) if c["density"] is not None else zk.FullDensity()


@pytest.mark.parametrize("group", [1, 2])
def test_golden_vectors(zk, worker, group):
    G = O.G1 if group == 1 else O.G2
    for c in GU.msm_cases(group):
        fut = zk.multiexp(worker, (c["bases"], c["base_offset"]), _density(zk, c), c["scalars"])
        if c["rc"] == 0:
            assert np.array_equal(G.to_affine(fut.wait()), c["expected"]), c["name"]
        else:
            with pytest.raises(zk.SynthesisError) as e:
                fut.wait()
            want = zk.SynthesisError.UNEXPECTED_IDENTITY if c["rc"] == 1 else zk.SynthesisError.IO_UNEXPECTED_EOF
            assert e.value.kind == want, c["name"]


@pytest.mark.parametrize("n", [1, 2, 31, 32, 100, 1000, 5000, 1 << 14])
def test_g1_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(1, n, seed=n)
    scalars = inputs.random_scalars(n, seed=7 * n + 1)
    rc, want = O.G1.multiexp(bases, scalars, threads=8)
    assert rc == 0
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))


@pytest.mark.parametrize("n", [1, 33, 500, 4096, 1 << 16])
def test_g2_matches_oracle(zk, worker, n):
    bases = inputs.bases_progression_cpu(2, n, seed=n)
    scalars = inputs.random_scalars(n, seed=11 * n + 1)
    rc, want = O.G2.multiexp(bases, scalars, threads=8)
    assert rc == 0
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert np.array_equal(O.G2.to_affine(got), O.G2.to_affine(want))


def test_density_half_and_offset_matches_oracle(zk, worker):
    """BASELINE config 2's DensityTracker variant (~50 % density) with a non-zero source offset."""
    n = 3000
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, size=n).astype(bool)
    used = int(bits.sum())
    bases = inputs.bases_progression_cpu(1, used + 5, seed=31)
    scalars = inputs.random_scalars(n, seed=32)
    scalars[::17] = 0
    scalars[5::23] = np.array([1, 0, 0, 0], dtype=np.uint64)
    rc, want = O.G1.multiexp(bases, scalars, density=GU.density_words(bits), density_bits=n, base_offset=5)
    assert rc == 0
    got = zk.multiexp(worker, (bases, 5), zk.DensityTracker.from_bools(bits), scalars).wait()
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))


def test_error_index_and_order(zk, worker):
    bases = inputs.bases_cpu(1, 6, seed=5)
    scalars = inputs.random_scalars(8, seed=6)
    bases[2] = 0
    with pytest.raises(zk.SynthesisError) as e:  # identity at 2 beats Eof at 6
        zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == 2
    scalars[2] = 0  # zero exponent: the identity base is skipped without being looked at (multiexp.rs:95-96)
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == 6
    rc, _ = O.G1.multiexp(bases, scalars)
    assert rc == 2


@pytest.mark.parametrize("group", [1, 2])
def test_identity_error_index_under_a_density_map(zk, worker, group):
    """The kernels find an identity base by its BASE index; the error must name the EXPONENT that owns it: the
    (base - offset)-th selected one (source.rs:101-118).  Two identities: the lower exponent wins; an identity whose
    exponent is zero or not selected is never looked at."""
    G = O.G1 if group == 1 else O.G2
    n = 200
    rng = np.random.default_rng(77)
    bits = rng.random(n) < 0.5
    sel = np.nonzero(bits)[0]
    bases = inputs.bases_progression_cpu(group, len(sel) + 3, seed=41)
    scalars = inputs.random_scalars(n, seed=42)
    dm = zk.DensityTracker.from_bools(bits)
    got = zk.multiexp(worker, (bases, 3), dm, scalars).wait()
    rc, want = G.multiexp(bases, scalars, density=GU.density_words(bits), density_bits=n, base_offset=3)
    assert rc == 0 and np.array_equal(G.to_affine(got), G.to_affine(want))
    bad = bases.copy()
    bad[3 + 70] = 0   # owned by exponent sel[70]
    bad[3 + 41] = 0   # owned by exponent sel[41]: reported
    bad[3 + 10] = 0   # owned by exponent sel[10], whose scalar is zero: skipped (multiexp.rs:95-96)
    bad[1] = 0        # below the source offset: never read
    sc = scalars.copy()
    sc[sel[10]] = 0
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (bad, 3), dm, sc).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == int(sel[41])
    rc_o, _ = G.multiexp(bad, sc, density=GU.density_words(bits), density_bits=n, base_offset=3)
    assert rc_o == 1


def test_non_canonical_exponent_is_bad_arguments(zk, worker):
    """An exponent >= 2^254 is not a FrRepr the reference could hand over (into_repr() < r); its top digit would overflow the
    bucket field.  The library reports bad arguments and the exponent's index instead of corrupting another window."""
    n = 3000
    bases = inputs.bases_progression_cpu(1, n, seed=61)
    scalars = inputs.random_scalars(n, seed=62)
    scalars[1234, 3] |= np.uint64(1 << 62)
    with pytest.raises(ValueError):
        zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert zk.lib.load().mi355zk_last_error_index() == 1234


def test_empty_input_is_identity(zk, worker):
    """No exponent evaluated: the reference's Projective::zero() = (0, 1, 0) in Montgomery form (ec.rs:229-235), from the host-buffer
    entry, the device-resident entry and -- Eof at index 0 -- a call whose bases ran out before the first exponent."""
    import torch

    one = np.array(O.G1.from_affine(inputs.G1_GEN_RAW)[8:12], dtype=np.uint64)   # Z of an affine point lifted to Jacobian = Fq::one()
    zero = np.concatenate([np.zeros(4, np.uint64), one, np.zeros(4, np.uint64)])
    got = zk.multiexp(worker, (np.zeros((0, 8), np.uint64), 0), zk.FullDensity(), np.zeros((0, 4), np.uint64)).wait()
    assert np.array_equal(got, zero)
    d = torch.zeros((4, 8), dtype=torch.int64, device="cuda")
    got = zk.multiexp(worker, (d, 0), zk.FullDensity(), torch.zeros((0, 4), dtype=torch.int64, device="cuda")).wait()
    assert np.array_equal(got, zero)
    lib = zk.lib.load()
    out = np.full(12, 7, dtype=np.uint64)
    sc = inputs.random_scalars(5, seed=3)
    bases = inputs.bases_cpu(1, 2, seed=4)
    rc = lib.mi355zk_bn254_g1_msm(bases.ctypes.data_as(C.c_void_p), 2, 2, sc.ctypes.data_as(C.c_void_p), 5, None, 0, out.ctypes.data_as(C.c_void_p))
    assert rc == zk.lib.ERR_UNEXPECTED_EOF and lib.mi355zk_last_error_index() == 0 and np.array_equal(out, zero)


def test_skewed_scalars_one_heavy_bucket(zk, worker):
    """All scalars equal: every point of a window lands in ONE bucket (worst-case load imbalance)."""
    n = 20000
    bases = inputs.bases_progression_cpu(1, n, seed=41)
    scalars = np.tile(inputs.random_scalars(1, seed=42), (n, 1))
    rc, want = O.G1.multiexp(bases, scalars, threads=8)
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))


def test_deterministic_run_twice(zk, worker):
    """The group element is reproducible; its Jacobian representative need not be (the order inside a bucket comes from LDS
    atomics in the partition kernels, and include/mi355zk.h allows any representative: projective equality is by value,
    ec.rs:45-85)."""
    n = 50000
    bases = inputs.bases_progression_cpu(1, n, seed=51)
    scalars = inputs.random_scalars(n, seed=52)
    a = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    b = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    assert np.array_equal(O.G1.to_affine(a), O.G1.to_affine(b))
    assert O.G1.eq(a, b)


def _dev_inputs(zk, log_n, seed, group=1):
    """2^log_n bases k_i*G and scalars generated on the device (as bench.py does)."""
    import torch

    import bench

    n = 1 << log_n
    dev = torch.device("cuda", 0)
    scalars = bench.gen_scalars(n, seed, dev)
    k = bench.gen_scalars(n, seed + 1, dev)
    bases = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    fn = zk.lib.load().mi355zk_bn254_g1_batch_mul_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_batch_mul_dev
    rc = fn(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n,
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return bases, scalars, k


def test_batch_mul_matches_oracle(zk, worker):
    bases, _, k = _dev_inputs(zk, 9, seed=61)
    want = O.G1.mul_many_affine(inputs.G1_GEN_RAW, k.cpu().numpy().view(np.uint64))
    assert np.array_equal(bases.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n", [64, 4096, 1 << 16])
def test_repeated_bases_collide_in_the_bucket_reduction(zk, worker, group, n):
    """One point, and its negative, repeated n times under scalars that are small multiples of a few window-sized steps: the
    buckets hold k*P for small k, so bucket sums, running sums and tree partial sums keep meeting EQUAL and OPPOSITE operands --
    the doubling and infinity branches of the full addition the bucket reduction runs on R-domain records (curveu.hpp
    xyzzr_add), and of the mixed addition in the accumulation (same point twice in a row in one bucket).  The sum has the closed
    form (sum of +/- k_i mod r) * P."""
    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    rng = np.random.default_rng(n + group)
    gen = inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW
    p_aff = G.mul_many_affine(gen, inputs.random_scalars(1, seed=99))[0]
    neg = p_aff.copy()
    w = p_aff.size // 2
    for c in range(w // 4):                                           # -P: negate every Fq component of y
        y = M.from_limbs(neg[w + 4 * c:w + 4 * c + 4])
        neg[w + 4 * c:w + 4 * c + 4] = M.to_limbs((M.Q - y) % M.Q)
    signs = rng.integers(0, 2, size=n)
    bases = np.ascontiguousarray(np.stack([neg if sg else p_aff for sg in signs]))
    for variant in range(3):
        if variant == 0:      # tiny scalars: everything lands in the first few buckets of window 0
            ks = [int(v) for v in rng.integers(1, 4, size=n)]
        elif variant == 1:    # a handful of values spread over every window
            vals = [1, 2, (1 << 13) + 1, (1 << 64) + (1 << 21), M.R_ORDER - 1, M.R_ORDER - 2, (M.R_ORDER - 1) // 2]
            ks = [vals[int(v)] for v in rng.integers(0, len(vals), size=n)]
        else:                 # pairs k, then the same k on the opposite point: total cancellation bucket by bucket
            half = [int(v) for v in rng.integers(1, 1 << 20, size=n // 2)]
            ks = half + half
            bases = np.ascontiguousarray(np.stack([p_aff] * (n // 2) + [neg] * (n // 2)))
            signs = np.array([0] * (n // 2) + [1] * (n // 2))
        scalars = np.array([M.to_limbs(k) for k in ks], dtype=np.uint64)
        total = sum((-k if sg else k) for k, sg in zip(ks, signs)) % M.R_ORDER
        got = G.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait())
        want = G.to_affine(G.mul(G.from_affine(p_aff), np.array(M.to_limbs(total), dtype=np.uint64)))
        assert np.array_equal(got, want), (group, n, variant)


@pytest.mark.parametrize("log_n", [20, 22])
def test_full_size_device_resident_properties(zk, worker, log_n):
    """BASELINE config 2 (2^20) and beyond, inputs resident in HBM.  Size-independent checks:
      - bases are k_i*G, so MSM(s, k*G) == (sum s_i*k_i mod r) * G        (closed form, via the oracle's mul)
      - additivity over point ranges: MSM(all) == MSM(first half) + MSM(second half)
      - the 2^14 prefix equals the CPU oracle's multiexp bit for bit."""
    import bn254_model as M

    bases, scalars, k = _dev_inputs(zk, log_n, seed=71 + log_n)
    n = 1 << log_n
    total = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    hs = scalars.cpu().numpy().view(np.uint64)
    hk = k.cpu().numpy().view(np.uint64)
    to_int = lambda a: sum(a[:, i].astype(object) << (64 * i) for i in range(4))  # noqa: E731
    dot = int(sum(s * kk for s, kk in zip(to_int(hs), to_int(hk))) % M.R_ORDER)
    want = O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(dot))
    assert np.array_equal(O.G1.to_affine(total), O.G1.to_affine(want))
    h = n // 2
    a = zk.multiexp(worker, (bases[:h], 0), zk.FullDensity(), scalars[:h]).wait()
    b = zk.multiexp(worker, (bases, h), zk.FullDensity(), scalars[h:]).wait()  # second half through the source offset
    assert np.array_equal(O.G1.to_affine(zk.shard.join_partials(np.stack([a, b]))), O.G1.to_affine(total))
    m = 1 << 14
    rc, ref = O.G1.multiexp(bases[:m].cpu().numpy().view(np.uint64), hs[:m], threads=8)
    got = zk.multiexp(worker, (bases[:m], 0), zk.FullDensity(), scalars[:m]).wait()
    assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(ref))


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("same_scalar", [0, 1])
@pytest.mark.parametrize("trusted", [0, 2])
def test_batch_exp_matches_oracle(zk, worker, group, same_scalar, trusted):
    """out[i] = k[i] * P[i] (powersoftau batch_exp) / k * P[i] (phase2 contribute), affine out.
    Bit exact against the oracle's mul_assign + into_affine, incl. scalar 0 / 1 / r-1 and an infinity base.  trusted = 2
    (MI355ZK_G2_TRUSTED_SUBGROUP): the psi-split G2 kernel (these bases are in the subgroup); accepted and ignored on G1."""
    import torch

    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    n = 300 if group == 1 else 64
    bases = inputs.bases_progression_cpu(group, n, seed=90 + group)
    bases[7] = 0  # infinity base -> infinity out
    ks = inputs.random_scalars(n, seed=91)
    ks[0] = 0
    ks[1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    ks[2] = np.array(M.to_limbs(M.R_ORDER - 1), dtype=np.uint64)
    if same_scalar:
        ks = np.tile(ks[5], (n, 1))
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_o = torch.empty_like(d_b)
    fn = zk.lib.load().mi355zk_bn254_g1_batch_exp_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_batch_exp_dev
    assert fn(C.c_void_p(d_o.data_ptr()), C.c_void_p(d_b.data_ptr()), C.c_void_p(d_k.data_ptr()), n, same_scalar | trusted, None) == 0
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint64)
    for i in range(n):
        want = G.to_affine(G.mul(G.from_affine(bases[i]), ks[i]))
        assert np.array_equal(got[i], want), i
    assert fn(C.c_void_p(d_o.data_ptr()), C.c_void_p(d_b.data_ptr()), C.c_void_p(d_k.data_ptr()), n, same_scalar | 4, None) == 3   # an unknown mode bit


@pytest.mark.parametrize("group,trusted", [(1, 0), (2, 0), (2, 2)])
def test_batch_exp_scalars_around_the_endomorphism_eigenvalues(zk, worker, group, trusted):
    """The per-point scalar multiplications split the scalar by the curve's endomorphism (glv.hpp: k = k1 + k2 lambda on G1,
    k = k1 + k2 mu on G2).  Scalars on and around the eigenvalue and its multiples, around 2^128, and with one half of the split
    equal to zero, per point and as the one shared scalar, against the oracle's plain double-and-add."""
    import torch

    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    R = M.R_ORDER
    ev = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD if group == 1 else M.Q % R
    vals = [ev, ev + 1, ev - 1, R - ev, 2 * ev % R, 3 * ev % R, (ev * ev) % R, (ev * ev + 1) % R, (1 << 128) - 1, 1 << 128, (1 << 128) + 1,
            (1 << 127), (ev << 64) % R, 7, 8, 9, 15, 16, 17, R - 2, (R - 1) // 2, (R + 1) // 2, 0x8888888888888888888888888888888888888888 % R,
            0x0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F % R,
            R, R + 1, (1 << 256) - 1, (1 << 255) + 12345]                  # not canonical: still k * P (plain windows take all 256 bits; the split is exact for any 256-bit k)
    n = len(vals)
    bases = inputs.bases_progression_cpu(group, n, seed=190 + group)
    ks = np.array([M.to_limbs(v) for v in vals], dtype=np.uint64)
    fn = zk.lib.load().mi355zk_bn254_g1_batch_exp_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_batch_exp_dev
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    d_o = torch.empty_like(d_b)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    assert fn(C.c_void_p(d_o.data_ptr()), C.c_void_p(d_b.data_ptr()), C.c_void_p(d_k.data_ptr()), n, trusted, None) == 0
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert np.array_equal(got[i], G.to_affine(G.mul(G.from_affine(bases[i]), np.array(M.to_limbs(vals[i] % R), dtype=np.uint64)))), hex(vals[i])
    # the same values as the ONE scalar of a phase2-style call (G1: the sliding-window kernel -- every value, and 0 / 1 / 2 / 3 / r - 1)
    for v in (vals + [0, 1, 2, 3, R - 1, 31, 32, 33, (1 << 127) - 1] if group == 1 else vals[:8]):
        one = torch.from_numpy(np.array([M.to_limbs(v)], dtype=np.uint64).view(np.int64)).cuda()
        assert fn(C.c_void_p(d_o.data_ptr()), C.c_void_p(d_b.data_ptr()), C.c_void_p(one.data_ptr()), n, 1 | trusted, None) == 0
        torch.cuda.synchronize()
        got = d_o.cpu().numpy().view(np.uint64)
        k = np.array(M.to_limbs(v % R), dtype=np.uint64)
        for i in range(0, n, 5):
            assert np.array_equal(got[i], G.to_affine(G.mul(G.from_affine(bases[i]), k))), hex(v)


def test_concurrent_calls_from_several_host_threads(zk, worker):
    """The prover queues 8 multiexps before the first wait() (prover.rs:250-298): the entry points must be
    re-entrant.  4 host threads run G1 / G2 multiexps and domain ops at the same time; every result is checked."""
    import threading

    jobs = []
    for t in range(4):
        n = 700 + 100 * t
        g = 1 + (t & 1)
        bases = inputs.bases_progression_cpu(g, n, seed=500 + t)
        scalars = inputs.random_scalars(n, seed=600 + t)
        G = O.G1 if g == 1 else O.G2
        rc, want = G.multiexp(bases, scalars, threads=2)
        a = inputs.random_fr_mont(1 << (10 + t), seed=700 + t)
        jobs.append((G, bases, scalars, G.to_affine(want), a, O.fr_domain_op(a, 10 + t, "icoset_fft").reshape(-1, 4)))
    errors = []

    def run(job):
        G, bases, scalars, want, a, want_fft = job
        try:
            for _ in range(3):
                got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
                assert np.array_equal(G.to_affine(got), want)
                dom = zk.EvaluationDomain.from_coeffs(a)
                dom.icoset_fft(worker)
                assert np.array_equal(dom.into_coeffs(), want_fft)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_prover_like_scalars_many_zeros_and_ones(zk, worker):
    """Groth16 witnesses are full of 0 and 1: every 1 lands in bucket 1 of window 0 (one huge bucket).  The
    segment-parallel heavy-bucket path must give the oracle's answer (and not serialise on one lane)."""
    import time

    n = 1 << 17
    bases = inputs.bases_progression_cpu(1, n, seed=801)
    scalars = inputs.random_scalars(n, seed=802)
    rng = np.random.default_rng(803)
    kind = rng.integers(0, 10, size=n)
    scalars[kind < 4] = np.array([1, 0, 0, 0], dtype=np.uint64)
    scalars[(kind >= 4) & (kind < 7)] = 0
    scalars[kind == 7] = np.array([2, 0, 0, 0], dtype=np.uint64)
    rc, want = O.G1.multiexp(bases, scalars, threads=8)
    assert rc == 0
    zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    t = time.perf_counter()
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    dt = time.perf_counter() - t
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))
    assert dt < 0.5, f"skewed multiexp took {dt * 1e3:.1f} ms (host-buffer entry point incl. H2D)"


@pytest.mark.parametrize("group", [1, 2])
def test_merge_pairs_and_dense_multiexp(zk, worker, group):
    """powersoftau / phase2 verification.  power_pairs(v) = merge_pairs(v[0..n-1], v[1..]) with one
    random scalar vector; infinity entries add nothing (dense_multiexp has no identity check).  Both sums must equal the
    oracle's per-vector multiexp, bit exact after normalisation."""
    import torch

    G = O.G1 if group == 1 else O.G2
    n = 3000 if group == 1 else 400
    v = inputs.bases_progression_cpu(group, n + 1, seed=950 + group)
    v[17] = 0  # an infinity entry: allowed here
    rho = inputs.random_scalars(n, seed=951)
    rho[5] = 0
    d_v = torch.from_numpy(v.view(np.int64)).cuda()
    d_rho = torch.from_numpy(rho.view(np.int64)).cuda()
    L = zk.lib.load()
    s, sx = np.zeros(12 * group, np.uint64), np.zeros(12 * group, np.uint64)
    fn = L.mi355zk_bn254_g1_merge_pairs_dev if group == 1 else L.mi355zk_bn254_g2_merge_pairs_dev
    rec = 64 * group
    assert fn(C.c_void_p(d_v.data_ptr()), C.c_void_p(d_v.data_ptr() + rec), C.c_void_p(d_rho.data_ptr()), n, None,
              s.ctypes.data_as(C.c_void_p), sx.ctypes.data_as(C.c_void_p)) == 0
    # oracle: zero scalar where the base is infinity (dense semantics), then the bellman multiexp restatement
    def ref(bases):
        sc = rho.copy()
        sc[~bases.any(axis=1)] = 0
        rc, out = G.multiexp(bases, sc, threads=4)
        assert rc == 0
        return G.to_affine(out)
    assert np.array_equal(G.to_affine(s), ref(v[:n]))
    assert np.array_equal(G.to_affine(sx), ref(v[1:n + 1]))
    one = np.zeros(12 * group, np.uint64)
    fn1 = L.mi355zk_bn254_g1_dense_multiexp_dev if group == 1 else L.mi355zk_bn254_g2_dense_multiexp_dev
    assert fn1(C.c_void_p(d_v.data_ptr()), C.c_void_p(d_rho.data_ptr()), n, None, one.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(G.to_affine(one), ref(v[:n]))


@pytest.mark.parametrize("group,trusted", [(1, 0), (2, 0), (2, 2)])
def test_sparse_matvec_qap_evaluation(zk, worker, group, trusted):
    """out[v] = sum over the terms of variable v of coeff * Lagrange point (parameters.rs:281-294),
    then batch_normalization.  CSR rows of very different lengths (empty rows, one long row such as the constant-one
    variable, +-1 and zero coefficients, an infinity base), bit exact against the oracle's mul_assign / add_assign."""
    import torch

    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    nb = 64 if group == 1 else 24
    bases = inputs.bases_progression_cpu(group, nb, seed=1300 + group)
    bases[5] = 0
    rng = np.random.default_rng(1301)
    row_len = [0, 1, 3, 0, (900 if group == 1 else 150), 2, 7, 1]
    row_ptr = np.concatenate([[0], np.cumsum(row_len)]).astype(np.uint32)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, nb, size=nnz).astype(np.uint32)
    coeff = inputs.random_scalars(nnz, seed=1302)
    small = rng.integers(0, 4, size=nnz)
    coeff[small == 0] = np.array([1, 0, 0, 0], dtype=np.uint64)
    coeff[small == 1] = np.array(M.to_limbs(M.R_ORDER - 1), dtype=np.uint64)
    coeff[3] = 0
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else np.int32)).cuda()  # noqa: E731
    d_bases, d_rp, d_col, d_cf = d(bases), d(row_ptr), d(col), d(coeff)
    d_out = torch.zeros((len(row_len), 8 * group), dtype=torch.int64, device="cuda")
    fn = zk.lib.load().mi355zk_bn254_g1_sparse_matvec_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_sparse_matvec_dev
    assert fn(C.c_void_p(d_out.data_ptr()), C.c_void_p(d_bases.data_ptr()), nb, C.c_void_p(d_rp.data_ptr()), C.c_void_p(d_col.data_ptr()),
              C.c_void_p(d_cf.data_ptr()), len(row_len), nnz, None, trusted) == 0
    # the index arrays are validated: one column beyond the bases, or a non-monotone row_ptr, is "bad arguments" (3), not a wild gather
    assert fn(C.c_void_p(d_out.data_ptr()), C.c_void_p(d_bases.data_ptr()), int(col.max()), C.c_void_p(d_rp.data_ptr()), C.c_void_p(d_col.data_ptr()),
              C.c_void_p(d_cf.data_ptr()), len(row_len), nnz, None, trusted) == 3
    bad_rp = row_ptr.copy()
    bad_rp[2], bad_rp[3] = bad_rp[3], bad_rp[2] - 1
    assert fn(C.c_void_p(d_out.data_ptr()), C.c_void_p(d_bases.data_ptr()), nb, C.c_void_p(d(bad_rp).data_ptr()), C.c_void_p(d_col.data_ptr()),
              C.c_void_p(d_cf.data_ptr()), len(row_len), nnz, None, trusted) == 3
    got = d_out.cpu().numpy().view(np.uint64)
    for r in range(len(row_len)):
        acc = G.from_affine(np.zeros(G.aff, np.uint64))
        for t in range(row_ptr[r], row_ptr[r + 1]):
            acc = G.add(acc, G.mul(G.from_affine(bases[col[t]]), coeff[t]))
        assert np.array_equal(got[r], G.to_affine(acc)), r


def test_baseline_size_2e26_closed_form_and_linearity(zk, worker):
    """BASELINE.json's headline size: 2^26 points resident in HBM (4 GiB of bases + 2 GiB of scalars).
    Size-independent checks at the FULL size:
      - closed form: the bases are k_i*G, so MSM == (sum_i s_i*k_i mod r) * G.  The 2^26-term dot product is taken
        on the device as a checksum of checksums: t_i = s_i*k_i/R (the pointwise Montgomery product on the canonical
        limbs), and sum_i t_i is output 0 of a 2^26-point fft of t (X[0] = sum a_i) -- two kernels that are
        parity-tested on their own;
      - linearity in the exponents: MSM(s) == MSM(a) + MSM(s - a);
      - additivity over point ranges through the Source offset: MSM(all) == MSM(first half) + MSM(second half)."""
    import torch

    import bench
    import bn254_model as M

    log_n = 26
    n = 1 << log_n
    L = zk.lib.load()
    dev = torch.device("cuda", 0)
    shard = 1 << 20
    scalars = torch.empty((n, 4), dtype=torch.int64, device=dev)
    k = torch.empty((n, 4), dtype=torch.int64, device=dev)
    bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    for s in range(n // shard):
        scalars[s * shard:(s + 1) * shard] = bench.gen_scalars(shard, 7_000_003 * s + 5, dev)
        k[s * shard:(s + 1) * shard] = bench.gen_scalars(shard, 9_000_011 * s + 3, dev)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    total = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    # closed form
    t = scalars.clone()
    assert L.mi355zk_bn254_fr_mul_assign_dev(C.c_void_p(t.data_ptr()), C.c_void_p(k.data_ptr()), n, None) == 0
    assert L.mi355zk_bn254_fr_domain_op_dev(C.c_void_p(t.data_ptr()), log_n, 0, None) == 0
    torch.cuda.synchronize()
    x0 = M.from_limbs([int(v) for v in t[0].cpu().numpy().view(np.uint64)])
    del t, k
    dot = x0 * M.MONT_R % M.R_ORDER
    want = O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(dot))
    assert np.array_equal(O.G1.to_affine(total), O.G1.to_affine(want))
    # linearity
    a = torch.empty_like(scalars)
    for s in range(n // shard):
        a[s * shard:(s + 1) * shard] = bench.gen_scalars(shard, 11_000_027 * s + 1, dev)
    b = scalars.clone()
    assert L.mi355zk_bn254_fr_sub_assign_dev(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    pa = zk.multiexp(worker, (bases, 0), zk.FullDensity(), a).wait()
    pb = zk.multiexp(worker, (bases, 0), zk.FullDensity(), b).wait()
    del a, b
    assert np.array_equal(O.G1.to_affine(zk.shard.join_partials(np.stack([pa, pb]))), O.G1.to_affine(total))
    # additivity over point ranges
    h = n // 2
    lo = zk.multiexp(worker, (bases[:h], 0), zk.FullDensity(), scalars[:h]).wait()
    hi = zk.multiexp(worker, (bases, h), zk.FullDensity(), scalars[h:]).wait()
    assert np.array_equal(O.G1.to_affine(zk.shard.join_partials(np.stack([lo, hi]))), O.G1.to_affine(total))


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("log_n", [6, 12, 20])
def test_window_group_partials_add_up(zk, worker, group, log_n):
    """Multi-GPU sharding by scalar windows (shard.plan): the partials of all (point range, window group) cells add up to the
    multiexp, for every split a power-of-two world produces, with a density map as well."""
    n = 1 << log_n
    G = O.G1 if group == 1 else O.G2
    if log_n <= 12:
        import torch

        bases_h = inputs.bases_progression_cpu(group, n + 3, seed=1500 + log_n)
        sc_h = inputs.random_scalars(n, seed=1501)
        sc_h[::7] = 0
        bases = torch.from_numpy(bases_h.view(np.int64)).cuda()
        scalars = torch.from_numpy(sc_h.view(np.int64)).cuda()
        rc, want = G.multiexp(bases_h, sc_h, base_offset=3)
        assert rc == 0
        off = 3
    else:
        # the geometry N = 2 / 4 / 8 ranks really run (window count, bucket count, partition shape of a 2^20-point call)
        bases, scalars, _ = _dev_inputs(zk, log_n, seed=1510, group=group)
        want = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
        off = 0
    for world in (2, 4, 8):
        pg, wg = zk.shard.plan(world)
        parts = []
        for rank in range(world):
            _, p, _, w = zk.shard.rank_groups(world, rank)
            lo, hi = zk.shard.shard_range(n, pg, p)
            part = zk.multiexp(worker, (bases, off + lo), zk.FullDensity(), scalars[lo:hi], window_group=(wg, w)).wait()
            parts.append(part)
        got = zk.shard.join_partials(np.stack(parts))
        assert np.array_equal(G.to_affine(got), G.to_affine(want)), (world, pg, wg)
    if log_n == 12 and group == 1:  # density map + window groups
        rng = np.random.default_rng(3)
        bits = rng.random(n) < 0.5
        rc, want_d = G.multiexp(bases_h, sc_h, density=GU.density_words(bits), density_bits=n, base_offset=3)
        dm = zk.DensityTracker.from_bools(bits)
        parts = [zk.multiexp(worker, (bases, 3), dm, scalars, window_group=(4, w)).wait() for w in range(4)]
        assert rc == 0 and np.array_equal(G.to_affine(zk.shard.join_partials(np.stack(parts))), G.to_affine(want_d))


def _to_int(a):
    return sum(a[:, i].astype(object) << (64 * i) for i in range(4))


def test_config2_density_tracker_at_2e20(zk, worker):
    """BASELINE config 2's DensityTracker variant at its stated size: 2^20 exponents, ~50 % density, a non-zero source
    offset, compacted bases k_j*G resident in HBM, a prover-like sprinkle of 0 / 1 scalars.
      - closed form at the full size: sum over the selected i of s_i * k_rank(i) (source.rs:101-118 defines rank)
      - the 2^14-exponent prefix (same map, same bases) against the CPU oracle, bit exact
      - additivity: MSM(s) == MSM(a) + MSM(s - a) under the same map."""
    import torch

    import bench
    import bn254_model as M

    log_n, off = 20, 7
    n = 1 << log_n
    rng = np.random.default_rng(2101)
    bits = rng.random(n) < 0.5
    used = int(bits.sum())
    dev = torch.device("cuda", 0)
    k = bench.gen_scalars(used + off, 2102, dev)
    bases = torch.empty((used + off, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert zk.lib.load().mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()),
                                                       used + off, None) == 0
    hs = inputs.random_scalars(n, seed=2103)
    kind = rng.integers(0, 10, size=n)
    hs[kind == 0] = 0
    hs[kind == 1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    scalars = torch.from_numpy(hs.view(np.int64)).to(dev)
    dm = zk.DensityTracker.from_bools(bits)
    total = zk.multiexp(worker, (bases, off), dm, scalars).wait()
    hk = _to_int(k.cpu().numpy().view(np.uint64))[off:]
    sel = _to_int(hs)[bits]
    dot = int(sum(int(a) * int(b) for a, b in zip(sel, hk)) % M.R_ORDER)
    want = O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(dot))
    assert np.array_equal(O.G1.to_affine(total), O.G1.to_affine(want))
    m = 1 << 14
    hb = bases.cpu().numpy().view(np.uint64)
    rc, ref = O.G1.multiexp(hb, hs[:m], density=GU.density_words(bits[:m]), density_bits=m, base_offset=off, threads=8)
    got = zk.multiexp(worker, (bases, off), zk.DensityTracker.from_bools(bits[:m]), scalars[:m]).wait()
    assert rc == 0 and np.array_equal(O.G1.to_affine(got), O.G1.to_affine(ref))
    a = bench.gen_scalars(n, 2104, dev)
    b = scalars.clone()
    assert zk.lib.load().mi355zk_bn254_fr_sub_assign_dev(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    pa = zk.multiexp(worker, (bases, off), dm, a).wait()
    pb = zk.multiexp(worker, (bases, off), dm, b).wait()
    assert np.array_equal(O.G1.to_affine(zk.shard.join_partials(np.stack([pa, pb]))), O.G1.to_affine(total))
    # one base too few: UnexpectedEof at the exponent that owns the missing base (source.rs:46-48)
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (bases[:used + off - 1], off), dm, scalars).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == int(np.nonzero(bits)[0][-1])


def test_g2_msm_at_2e20_closed_form(zk, worker):
    """2^20-point G2 multiexp (prover.rs:297-298's B_G2 at BASELINE config 5's size), bases k_i*G2 resident in HBM:
    closed form (sum s_i k_i) * G2, additivity over point ranges through the source offset, and the 2^12 prefix against the oracle."""
    import torch

    import bench
    import bn254_model as M

    log_n = 20
    n = 1 << log_n
    dev = torch.device("cuda", 0)
    scalars = bench.gen_scalars(n, 2201, dev)
    k = bench.gen_scalars(n, 2202, dev)
    bases = torch.empty((n, 16), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G2_GEN_RAW)
    assert zk.lib.load().mi355zk_bn254_g2_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    total = zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    hs, hk = scalars.cpu().numpy().view(np.uint64), k.cpu().numpy().view(np.uint64)
    dot = int(sum(int(a) * int(b) for a, b in zip(_to_int(hs), _to_int(hk))) % M.R_ORDER)
    want = O.G2.mul(O.G2.from_affine(inputs.G2_GEN_RAW), M.to_limbs(dot))
    assert np.array_equal(O.G2.to_affine(total), O.G2.to_affine(want))
    h = n // 2 + 12345
    lo = zk.multiexp(worker, (bases[:h], 0), zk.FullDensity(), scalars[:h]).wait()
    hi = zk.multiexp(worker, (bases, h), zk.FullDensity(), scalars[h:]).wait()
    assert np.array_equal(O.G2.to_affine(zk.shard.join_partials(np.stack([lo, hi]))), O.G2.to_affine(total))
    m = 1 << 12
    rc, ref = O.G2.multiexp(bases[:m].cpu().numpy().view(np.uint64), hs[:m], threads=8)
    got = zk.multiexp(worker, (bases[:m], 0), zk.FullDensity(), scalars[:m]).wait()
    assert rc == 0 and np.array_equal(O.G2.to_affine(got), O.G2.to_affine(ref))


@pytest.mark.parametrize("group", [1, 2])
def test_montgomery_scalars_fused_into_repr(zk, worker, group):
    """the prover holds Montgomery-form Fr (`Vec<Scalar<E>>`) and converts with into_repr() before every
    multiexp (prover.rs:89-129).  MI355ZK_MSM_SCALARS_MONTGOMERY fuses that pass into the digit extraction: same result as
    the canonical call and as the oracle; mi355zk_bn254_fr_into_repr_dev is the standalone pass.  Includes 0, 1, r - 1 and a
    density map."""
    import torch

    import bn254_model as M

    G = O.G1 if group == 1 else O.G2
    n = 5000 if group == 1 else 700
    bases = inputs.bases_progression_cpu(group, n, seed=2300 + group)
    canon = inputs.random_scalars(n, seed=2301)
    canon[0] = 0
    canon[1] = np.array([1, 0, 0, 0], dtype=np.uint64)
    canon[2] = np.array(M.to_limbs(M.R_ORDER - 1), dtype=np.uint64)
    r2 = np.array(M.to_limbs(pow(2, 512, M.R_ORDER)), dtype=np.uint64)
    mont = O.fe_mul_many(O.FR, canon, np.tile(r2, (n, 1))).reshape(n, 4)       # from_repr: c * R mod r
    d_bases = torch.from_numpy(bases.view(np.int64)).cuda()
    d_mont = torch.from_numpy(mont.view(np.int64)).cuda()
    rng = np.random.default_rng(2302)
    bits = rng.random(n) < 0.6
    for dm, dens in ((zk.FullDensity(), None), (zk.DensityTracker.from_bools(bits), bits)):
        rc, want = G.multiexp(bases, canon, density=None if dens is None else GU.density_words(dens), density_bits=None if dens is None else n,
                              threads=8)
        assert rc == 0
        got = zk.multiexp(worker, (d_bases, 0), dm, d_mont, scalars_montgomery=True).wait()
        assert np.array_equal(G.to_affine(got), G.to_affine(want))
    out = torch.empty_like(d_mont)
    assert zk.lib.load().mi355zk_bn254_fr_into_repr_dev(C.c_void_p(out.data_ptr()), C.c_void_p(d_mont.data_ptr()), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), canon)


def test_host_entry_streamed_upload_and_bases_cache(zk, worker):
    """The host-buffer entry point (mi355zk_bn254_g1_msm, what a bellman shim calls with `Arc<Vec<G1Affine>>` + `Vec<FrRepr>`):
    2^23 exponents are cut into chunks whose upload overlaps the previous chunk's kernels, and the base vector stays cached on
    the device once the caller has PINNED it (mi355zk_bases_cache_pin; unpinned vectors are uploaded on every call, so a record
    rewritten in place is seen).  Same group element as the device-resident call -- with a density map and
    a source offset, on the first (uploading) and the second (cached) call; a changed CRS at the same address is noticed; the
    Source errors keep their global exponent index across chunks."""
    import torch

    import bench

    log_n, off = 23, 3
    n = 1 << log_n
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2401)
    bits = rng.random(n) < 0.75
    used = int(bits.sum())
    k = bench.gen_scalars(used + off, 2402, dev)
    d_bases = torch.empty((used + off, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert zk.lib.load().mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(d_bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()),
                                                       used + off, None) == 0
    d_scalars = bench.gen_scalars(n, 2403, dev)
    dm = zk.DensityTracker.from_bools(bits)
    want = O.G1.to_affine(zk.multiexp(worker, (d_bases, off), dm, d_scalars).wait())
    h_bases = d_bases.cpu().numpy().view(np.uint64)
    h_scalars = d_scalars.cpu().numpy().view(np.uint64)
    # NOT pinned: the plain host-buffer entry uploads its bases on every call, so a record rewritten in place between two calls
    # -- one the cache's fingerprint does not sample -- is seen (the reference reads the vector it is given)
    plain = zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()
    assert np.array_equal(O.G1.to_affine(plain), want)
    mid = off + 12345
    saved = h_bases[mid].copy()
    h_bases[mid] = h_bases[mid + 1]
    d_bases[mid] = d_bases[mid + 1]
    want_mut = O.G1.to_affine(zk.multiexp(worker, (d_bases, off), dm, d_scalars).wait())
    assert not np.array_equal(want_mut, want)
    assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()), want_mut)
    h_bases[mid] = saved
    d_bases[mid] = torch.from_numpy(saved.view(np.int64)).to(dev)
    # pinned (the shim's promise that the Arc<Vec<G>> is immutable): the second call is served from the device copy
    zk.pin_bases(h_bases)
    first = zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()       # uploads bases + scalars, chunk by chunk
    second = zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()      # bases served from the cache
    assert np.array_equal(O.G1.to_affine(first), want) and np.array_equal(O.G1.to_affine(second), want)
    # FullDensity over a prefix of the same (cached) vector
    want_fd = O.G1.to_affine(zk.multiexp(worker, (d_bases, 0), zk.FullDensity(), d_scalars[:used]).wait())
    assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (h_bases, 0), zk.FullDensity(), h_scalars[:used]).wait()), want_fd)
    # a different CRS at the same address without an invalidate (a broken promise): the fingerprint of the sampled records is
    # the second line of defence -- record 0 is always sampled
    h_bases[0] = h_bases[1]
    d_bases[0] = d_bases[1]
    want2 = O.G1.to_affine(zk.multiexp(worker, (d_bases, off), dm, d_scalars).wait())
    assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()), want2)
    # an identity base owned by an exponent of the SECOND chunk: the error carries the global exponent index
    sel = np.nonzero(bits)[0]
    target = int(sel[len(sel) * 3 // 4])
    h_bases[off + len(sel) * 3 // 4] = 0
    zk.unpin_bases(h_bases)   # a record rewritten in place: the promise ends first (the vector is uploaded again from here on)
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (h_bases, off), dm, h_scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == target
    # one base short: Eof at the last selected exponent (in the last chunk)
    h_bases[off + len(sel) * 3 // 4] = h_bases[1]
    zk.lib.load().mi355zk_bases_cache_invalidate(None)
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (h_bases[:used + off - 1].copy(), off), dm, h_scalars).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == int(sel[-1])


@pytest.mark.parametrize("group", [1, 2])
def test_streamed_chunks_against_the_oracle(zk, worker, group, monkeypatch):
    """The chunked host-buffer call (one bucket array carried across the chunks, msm_accumulate_kernel<.., CARRY>) against the CPU
    ORACLE, at a size the oracle finishes in seconds: MI355ZK_HOST_CHUNK_TEST cuts a 3000-exponent call into chunks of 512.
    Density map with a source offset, exponents 0 and 1, a base repeated across chunk boundaries (the carried bucket meets its own
    point again: the doubling branch), P and -P in different chunks (a carried bucket that becomes infinity); pinned and unpinned
    vectors; UnexpectedIdentity in a later chunk and Eof keep the unchunked call's exponent index."""
    import bn254_model as M

    monkeypatch.setenv("MI355ZK_HOST_CHUNK_TEST", "512")
    G = O.G1 if group == 1 else O.G2
    n, off = 3000, 4
    rng = np.random.default_rng(2700 + group)
    bits = rng.random(n) < 0.7
    sel = np.nonzero(bits)[0]
    used = len(sel)
    bases = inputs.bases_progression_cpu(group, used + off, seed=2701 + group)
    scalars = inputs.random_scalars(n, seed=2702 + group)
    scalars[::19] = 0
    scalars[3::29] = np.array([1, 0, 0, 0], dtype=np.uint64)
    # the same base under the same exponent in chunks 0, 2 and 4 (same bucket in every window: P + P across a carry), and its
    # negative under that exponent in chunk 5 ... twice, so that one pair cancels a carried bucket to infinity
    r0, r2, r4, r5a, r5b = (int(np.searchsorted(sel, x)) for x in (40, 1100, 2100, 2600, 2700))
    for r in (r2, r4):
        bases[off + r] = bases[off + r0]
        scalars[sel[r]] = scalars[sel[r0]]
    neg = bases[off + r0].copy()
    half = 4 * group
    y = [M.from_limbs(neg[half + 4 * k: half + 4 * k + 4]) for k in range(group)]
    for k in range(group):
        neg[half + 4 * k: half + 4 * k + 4] = M.to_limbs((M.Q - y[k]) % M.Q)
    for r in (r5a, r5b):
        bases[off + r] = neg
        scalars[sel[r]] = scalars[sel[r0]]
    dens = GU.density_words(bits)
    rc, want = G.multiexp(bases, scalars, density=dens, density_bits=n, base_offset=off, threads=4)
    assert rc == 0
    dm = zk.DensityTracker.from_bools(bits)
    got = zk.multiexp(worker, (bases, off), dm, scalars).wait()                       # unpinned: the bases travel chunk by chunk
    assert np.array_equal(G.to_affine(got), G.to_affine(want))
    zk.pin_bases(bases)
    for _ in range(2):                                                                 # pinned: first call fills the cache, second uses it
        assert np.array_equal(G.to_affine(zk.multiexp(worker, (bases, off), dm, scalars).wait()), G.to_affine(want))
    zk.unpin_bases(bases)
    # FullDensity, more exponents than bases: Eof at the unchunked index, result of the prefix still correct up to there
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (bases, off), zk.FullDensity(), scalars).wait()
    assert e.value.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.value.index == used
    # an identity base owned by an exponent of chunk 3
    bad = bases.copy()
    target = int(sel[np.searchsorted(sel, 1700)])
    bad[off + int(np.searchsorted(sel, 1700))] = 0
    assert scalars[target].any()
    rc, _ = G.multiexp(bad, scalars, density=dens, density_bits=n, base_offset=off)
    assert rc == 1
    with pytest.raises(zk.SynthesisError) as e:
        zk.multiexp(worker, (bad, off), dm, scalars).wait()
    assert e.value.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.value.index == target


def test_host_entry_streamed_g2_and_heavy_buckets_in_later_chunks(zk, worker):
    """The streamed host-buffer call carries ONE bucket array across its chunks (msm_accumulate_kernel<.., CARRY>,
    msm_heavy_combine_kernel with carry): G2 at 2^23 points (four chunks), and G1 with prover-like exponents whose byte-valued
    and one-valued exponents sit in the SECOND half of the vector -- heavy buckets that only appear in the later chunks, continued
    through the segment-parallel path.  Same group element as the device-resident single-pass call."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    # ---- G2
    log_n = 23
    bases, scalars, _ = _dev_inputs(zk, log_n, seed=2601, group=2)
    want = O.G2.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait())
    hb, hs = bases.cpu().numpy().view(np.uint64), scalars.cpu().numpy().view(np.uint64)
    zk.pin_bases(hb)
    assert np.array_equal(O.G2.to_affine(zk.multiexp(worker, (hb, 0), zk.FullDensity(), hs).wait()), want)    # uploads the bases chunk by chunk
    assert np.array_equal(O.G2.to_affine(zk.multiexp(worker, (hb, 0), zk.FullDensity(), hs).wait()), want)    # bases cached: growing chunks
    zk.unpin_bases(hb)
    del bases, scalars, hb, hs
    # ---- G1, skew in the later chunks
    n = 1 << 23
    bases, scalars, _ = _dev_inputs(zk, 23, seed=2611)
    sc = scalars.cpu().numpy().view(np.uint64).copy()
    rng = np.random.default_rng(2612)
    half = n // 2
    kind = rng.random(half)
    tail = sc[half:]
    tail[kind < 0.4] = np.array([1, 0, 0, 0], dtype=np.uint64)
    tail[(kind >= 0.4) & (kind < 0.6)] = 0
    small = (kind >= 0.6) & (kind < 0.8)
    tail[small] = 0
    tail[small, 0] = rng.integers(2, 256, size=int(small.sum()), dtype=np.uint64)
    d_sc = torch.from_numpy(sc.view(np.int64)).to(dev)
    want = O.G1.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), d_sc).wait())
    hb = bases.cpu().numpy().view(np.uint64)
    zk.pin_bases(hb)
    zk.multiexp(worker, (hb[:4096], 0), zk.FullDensity(), sc[:4096]).wait()
    for _ in range(2):
        assert np.array_equal(O.G1.to_affine(zk.multiexp(worker, (hb, 0), zk.FullDensity(), sc).wait()), want)
    zk.unpin_bases(None)


def test_prover_like_exponents_at_2e22_take_the_big_bin_path(zk, worker):
    """A Groth16 witness at size: 40 % ones, 30 % zeros, 10 % bytes, the rest uniform.  Window 0 then sends ~half of all points
    into ONE coarse bin of the partition (far more than a workgroup holds in registers: the segmented msm_bigbin_* kernels) and
    millions into single buckets (the segment-parallel heavy-bucket accumulation).  Closed form (sum s_i k_i) G, and the skewed
    vector must not be slower than the uniform one (it has fewer non-zero digits)."""
    import time

    import torch

    import bench
    import bn254_model as M

    bases, uniform, k = _dev_inputs(zk, 22, seed=2501)
    n = 1 << 22
    dev = bases.device
    g = torch.Generator(device=dev)
    g.manual_seed(2502)
    kind = torch.randint(0, 10, (n,), device=dev, generator=g)
    skew = uniform.clone()
    skew[kind < 4] = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=dev)
    skew[(kind >= 4) & (kind < 7)] = 0
    small = torch.randint(0, 256, (n,), device=dev, generator=g, dtype=torch.int64)
    m8 = kind == 7
    skew[m8] = torch.stack([small, torch.zeros_like(small), torch.zeros_like(small), torch.zeros_like(small)], dim=1)[m8]
    got = zk.multiexp(worker, (bases, 0), zk.FullDensity(), skew).wait()
    hs, hk = skew.cpu().numpy().view(np.uint64), k.cpu().numpy().view(np.uint64)
    dot = int(sum(int(a) * int(b) for a, b in zip(_to_int(hs), _to_int(hk))) % M.R_ORDER)
    want = O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(dot))
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want))

    def timed(sc):
        zk.multiexp(worker, (bases, 0), zk.FullDensity(), sc).wait()
        t = time.perf_counter()
        for _ in range(3):
            zk.multiexp(worker, (bases, 0), zk.FullDensity(), sc).wait()
        return (time.perf_counter() - t) / 3

    t_uniform, t_skew = timed(uniform), timed(skew)
    assert t_skew < 1.2 * t_uniform, f"prover-like exponents: {t_skew * 1e3:.2f} ms against {t_uniform * 1e3:.2f} ms uniform"


@pytest.mark.parametrize("group", [1, 2])
def test_pair_and_lane_kernels_agree(group):
    """The accumulation, both kernels at every size: MI355ZK_G{1,2}_PAIR=1 (a pair of lanes per bucket -- the even lane keeps (X, ZZ),
    the odd lane (Y, ZZZ): msm_accumulate_pair_kernel / _g1_kernel) and =0 (one lane per bucket), each in its own process (the switch is
    read once).  Every case of tests/pair_mode_check.py is held against the CPU oracle inside the child where the oracle is quick
    (n <= 4096, equal points colliding in a bucket, an identity base's error); the 2^17 / 2^19-point results and the streamed
    (carried-bucket) call must come out byte-identical from both kernels."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, **{"MI355ZK_G%d_PAIR" % group: mode})
        p = subprocess.run([sys.executable, os.path.join(here, "pair_mode_check.py"), str(group)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (mode, p.stdout[-2000:], p.stderr[-4000:])
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["0"].keys() == res["1"].keys()
    for k in res["0"]:
        assert res["0"][k] == res["1"][k], k
    assert "123" in res["0"]["identity"]


_HEAVY_PAST_REACH = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench, bn254_model as M, oracle_lib as O
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << 20
g = np.zeros(5 + 128, np.uint32)
assert L.mi355zk_selftest_msm_digits(n, 1, None, 0, 0, 0, None, g.ctypes.data) == 0
c, W, nb, rmul = (int(x) for x in g[:4])
assert c == 14 and rmul == 1 and W * nb <= (1 << 18), (c, W, nb, rmul)      # the split launch is on: <= 2^18 buckets over all windows
width = [int(x) for x in g[5:5 + W]]; shift = [int(x) for x in g[5 + W:5 + 2 * W]]
rng = np.random.default_rng(9101)
HOT = 3686                                   # 45 % of a window's 8192 buckets; 19 x 3686 = 70 034 hot buckets of 2^20 / 3686 = 284 entries each
limbs = np.zeros((n, 4), np.uint64)
for wd, sh in zip(width, shift):
    hot = rng.choice(np.arange(1, (1 << (wd - 1)) - 1, dtype=np.uint64), size=min(HOT, (1 << (wd - 1)) - 2), replace=False)   # positive digits below the sign boundary: no carries
    d = hot[rng.permutation(n) % len(hot)]     # every hot digit exactly n / 3686 = 284 or 285 times: all of them over the threshold
    li, off = sh // 64, sh % 64
    limbs[:, li] |= d << np.uint64(off)
    if off + wd > 64: limbs[:, li + 1] |= d >> np.uint64(64 - off)
scalars = torch.from_numpy(limbs.view(np.int64)).to(dev)
k = bench.gen_scalars(n, 9102, dev)
bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
torch.cuda.synchronize()
got = zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
os.environ["MI355ZK_MSM_SPLIT"] = "0"        # (read per call) the lane-per-bucket launch alone
plain = zk.multiexp(w, (bases, 0), zk.FullDensity(), scalars).wait()
to_int = lambda a: sum(a[:, i].astype(object) << (64 * i) for i in range(4))
dot = int(sum(s * kk for s, kk in zip(to_int(limbs), to_int(k.cpu().numpy().view(np.uint64)))) % M.R_ORDER)
want = O.G1.to_affine(O.G1.mul(O.G1.from_affine(inputs.G1_GEN_RAW), M.to_limbs(dot)))
print(json.dumps({"split": bool(np.array_equal(O.G1.to_affine(got), want)), "plain": bool(np.array_equal(O.G1.to_affine(plain), want))}))
'''


def test_more_over_long_buckets_than_the_heavy_path_reaches(zk, worker):
    """with the quad-per-bucket launch active the lane-per-bucket launch skips EVERY bucket of order[0 .. max(split_hb, hb)) that is
    longer than split_t; the segment-parallel path takes the over-long ones of order[0 .. hb) only, hb <= 65 536 -- an over-long bucket past
    that reach was accumulated by nobody (round 4's library returns a wrong point for this input).  2^20 exponents at c = 14 (19 windows x 8192
    buckets <= 2^18: the split launch is on) whose digits take 3686 values per window: 70 034 buckets of 284 entries against a heavy threshold of
    274.  Closed form (bases k_i * G), with and without the split launch; a process of its own because MI355ZK_MSM_C is read once."""
    import json
    import os
    import subprocess
    import sys

    env = dict(os.environ, MI355ZK_MSM_C="14")
    env.pop("MI355ZK_MSM_SPLIT", None)
    r = subprocess.run([sys.executable, "-c", _HEAVY_PAST_REACH], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and line, r.stderr[-800:]
    assert json.loads(line[-1]) == {"split": True, "plain": True}
