"""GPU parity tests for the Fr NTT path through the C ABI: byte-exact on all 2^log_n x 32 bytes against
the golden vectors (definition DFT) and the oracle's serial_fft, for every pass structure the kernel
uses (1, 2 and 3 passes; uneven splits), plus the reference tests' identities at full size."""
import numpy as np
import pytest

import golden_util as GU
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu
OPS = ["fft", "ifft", "coset_fft", "icoset_fft"]


def test_golden_vectors(zk, worker):
    for c in GU.ntt_cases():
        a = c["input"].copy()
        zk.bellman.best_fft(a, worker, c["omega"], c["log_n"])  # domain.rs:263 with an explicit omega
        assert np.array_equal(a, c["fft"]), c["log_n"]
        for op in OPS:
            dom = zk.EvaluationDomain.from_coeffs(c["input"])
            getattr(dom, op)(worker)
            assert np.array_equal(dom.into_coeffs(), c[op]), (c["log_n"], op)


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 15, 16, 19, 20])
@pytest.mark.parametrize("op", OPS)
def test_domain_ops_match_oracle(zk, worker, log_n, op):
    a = inputs.random_fr_mont(1 << log_n, seed=100 + log_n)
    want = O.fr_domain_op(a, log_n, op).reshape(-1, 4)
    dom = zk.EvaluationDomain.from_coeffs(a)
    getattr(dom, op)(worker)
    assert np.array_equal(dom.into_coeffs(), want)


@pytest.mark.parametrize("log_n", [21, 22])
def test_long_row_transforms_match_oracle(zk, worker, log_n):
    """2^21 and 2^22 run as two passes of 2048-point rows (11 + 10, 11 + 11: the radix-4 kernel with a lone first stage and the
    stage-2 product skip given up for the value bound); the three-pass plans are covered at 2^24 and 2^26 below."""
    a = inputs.random_fr_mont(1 << log_n, seed=200 + log_n)
    want = O.fr_domain_op(a, log_n, "fft", log_cpus=3).reshape(-1, 4)  # parallel_fft shape == serial_fft (domain.rs:465-496)
    dom = zk.EvaluationDomain.from_coeffs(a)
    dom.fft(worker)
    assert np.array_equal(dom.into_coeffs(), want)


def test_ragged_length_is_zero_padded(zk, worker):
    """from_coeffs pads to the next power of two with zeros (domain.rs:89)."""
    a = inputs.random_fr_mont(1000, seed=7)
    dom = zk.EvaluationDomain.from_coeffs(a)
    assert dom.exp == 10
    dom.coset_fft(worker)
    padded = np.concatenate([a, np.zeros((24, 4), np.uint64)])
    assert np.array_equal(dom.into_coeffs(), O.fr_domain_op(padded, 10, "coset_fft").reshape(-1, 4))


@pytest.mark.parametrize("log_n", [20, 24])
def test_device_resident_roundtrips(zk, worker, log_n):
    """domain.rs:427-463 at BASELINE config 3's size and beyond, data resident in HBM:
    ifft(fft(a)) == a, icoset_fft(coset_fft(a)) == a, and linearity fft(a + b) == fft(a) + fft(b)
    checked through the oracle's field add on a sample."""
    import torch

    n = 1 << log_n
    host = inputs.random_fr_mont(n, seed=300 + log_n)
    d = torch.from_numpy(host.view(np.int64)).cuda()
    dom = zk.EvaluationDomain(d.clone(), log_n)
    dom.fft(worker)
    f_a = dom.coeffs.clone()
    dom.ifft(worker)
    assert torch.equal(dom.coeffs, d)
    dom.coset_fft(worker)
    dom.icoset_fft(worker)
    assert torch.equal(dom.coeffs, d)
    # spot-check fft output against the definition on a few output indices: X[k] = sum_i a[i] w^(ik)
    if log_n == 20:
        want = O.fr_domain_op(host, log_n, "fft").reshape(-1, 4)
        assert np.array_equal(f_a.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("log_n", [23, 26])
def test_large_transforms_roundtrip_and_halve(zk, worker, log_n):
    """Sizes the oracle cannot reach in seconds (three passes with 9-bit rows at 2^26: the radix-4 kernel with a lone first stage),
    by size-independent properties: ifft(fft(a)) == a, icoset_fft(coset_fft(a)) == a, and the decimation identity that ties a
    transform to the one of half its size -- the even outputs of fft_n(a) are fft_{n/2}(a[:n/2] + a[n/2:]) (omega_{n/2} =
    omega_n^2 in bellman's domains, domain.rs:60-75) -- so that, level by level, the large sizes hang on the oracle-checked ones."""
    import ctypes as C

    import torch

    import bench

    L = zk.lib.load()
    n = 1 << log_n
    d = bench.gen_scalars(n, 4200 + log_n, torch.device("cuda", 0))      # canonical values < r are valid Montgomery elements
    dom = zk.EvaluationDomain(d.clone(), log_n)
    dom.fft(worker)
    f = dom.coeffs.clone()
    dom.ifft(worker)
    assert torch.equal(dom.coeffs, d)
    dom.coset_fft(worker)
    dom.icoset_fft(worker)
    assert torch.equal(dom.coeffs, d)
    del dom
    lo, hi = d[:n // 2].clone(), d[n // 2:].clone()
    neg = torch.zeros_like(hi)
    assert L.mi355zk_bn254_fr_sub_assign_dev(C.c_void_p(neg.data_ptr()), C.c_void_p(hi.data_ptr()), n // 2, None) == 0      # -hi
    assert L.mi355zk_bn254_fr_sub_assign_dev(C.c_void_p(lo.data_ptr()), C.c_void_p(neg.data_ptr()), n // 2, None) == 0      # lo + hi
    half = zk.EvaluationDomain(lo, log_n - 1)
    half.fft(worker)
    assert torch.equal(half.coeffs, f[0::2].contiguous())


@pytest.mark.parametrize("log_n,op", [(21, "ifft"), (21, "coset_fft"), (21, "icoset_fft"), (22, "coset_fft"), (22, "icoset_fft"), (23, "ifft"),
                                      (23, "coset_fft"), (23, "icoset_fft"), (24, "icoset_fft"), (25, "coset_fft"), (25, "icoset_fft"), (25, "ifft")])
def test_scaled_transforms_without_a_full_table_match_oracle(zk, worker, log_n, op):
    """(round 5) From 2^21 on there is no full inter-pass table; the scale factors of ifft / coset_fft / icoset_fft are folded into small
    tables: the first pass's butterfly twiddles carry the row part of g^i (odd row lengths at 2^21 / 2^25: the lone stage 0 has a twiddle then)
    and one product the column part; the last pass multiplies by (ginv^stride)^k and a per-row constant; minv alone rides on the twiddle
    of the pass before the last.  Whole arrays against the oracle."""
    a = inputs.random_fr_mont(1 << log_n, seed=500 + log_n)
    want = O.fr_domain_op(a, log_n, op, log_cpus=3).reshape(-1, 4)
    dom = zk.EvaluationDomain.from_coeffs(a)
    getattr(dom, op)(worker)
    assert np.array_equal(dom.into_coeffs(), want)


@pytest.mark.parametrize("op", ["ifft", "coset_fft"])
def test_2e24_three_pass_uneven_split_matches_oracle(zk, worker, op):
    """2^24 elements (512 MiB): three passes with an uneven digit split, the fused scalings on the first / last pass; all
    2^24 x 32 bytes against the oracle (its parallel_fft shape, asserted equal to serial_fft by domain.rs:465-496)."""
    log_n = 24
    a = inputs.random_fr_mont(1 << log_n, seed=400)
    want = O.fr_domain_op(a, log_n, op, log_cpus=3).reshape(-1, 4)
    dom = zk.EvaluationDomain.from_coeffs(a)
    getattr(dom, op)(worker)
    assert np.array_equal(dom.into_coeffs(), want)


@pytest.mark.parametrize("log_n,batch", [(0, 2), (3, 3), (10, 3), (12, 9), (16, 3), (20, 3), (20, 2), (21, 2)])
def test_batched_domain_ops_match_oracle(zk, worker, log_n, batch):
    """(round 5) mi355zk_bn254_fr_domain_op_batch_dev: the same operation on `batch` arrays, one launch per pass over all of them (prover.rs:217-241
    transforms a, b and c one after the other).  Every array against the oracle, for the four operations; nine arrays cross the eight-per-launch chunk."""
    import torch

    for op in OPS:
        hosts = [inputs.random_fr_mont(1 << log_n, seed=900 + 10 * log_n + t) for t in range(batch)]
        doms = [zk.EvaluationDomain(torch.from_numpy(h.view(np.int64)).cuda(), log_n) for h in hosts]
        getattr(zk.EvaluationDomain, op + "_many")(worker, doms)
        for h, d in zip(hosts, doms):
            want = O.fr_domain_op(h, log_n, op, log_cpus=3 if log_n >= 20 else 31).reshape(-1, 4)
            assert np.array_equal(d.coeffs.cpu().numpy().view(np.uint64), want), (op, log_n)


def test_batched_domain_op_rejects_bad_batches(zk, worker):
    import ctypes as C

    import torch

    L = zk.lib.load()
    a = torch.zeros((16, 4), dtype=torch.int64, device="cuda")
    two = (C.c_void_p * 2)(a.data_ptr(), a.data_ptr())
    assert L.mi355zk_bn254_fr_domain_op_batch_dev(two, 2, 4, zk.lib.OP_FFT, None) == 3        # one array twice
    nul = (C.c_void_p * 2)(a.data_ptr(), None)
    assert L.mi355zk_bn254_fr_domain_op_batch_dev(nul, 2, 4, zk.lib.OP_FFT, None) == 3
    assert L.mi355zk_bn254_fr_domain_op_batch_dev(two, 0, 4, zk.lib.OP_FFT, None) == 3
    one = (C.c_void_p * 1)(a.data_ptr())
    assert L.mi355zk_bn254_fr_domain_op_batch_dev(one, 1, 4, 99, None) == 3


def _mont(v):
    import bn254_model as M

    return np.array(M.to_limbs(v * (1 << 256) % M.R_ORDER), dtype=np.uint64)


def _powers(g, n):
    """[g^0 .. g^(n-1)] as Montgomery rows, by doubling through the oracle's field product"""
    import bn254_model as M

    p = _mont(1).reshape(1, 4)
    m = 1
    while m < n:
        step = np.tile(_mont(pow(g, m, M.R_ORDER)), (m, 1))
        p = np.concatenate([p, O.fe_mul_many(O.FR, p, step).reshape(-1, 4)])
        m *= 2
    return p[:n]


@pytest.mark.parametrize("log_n", [3, 10, 12, 16, 20, 21])
def test_scaled_transform_with_arbitrary_factors_matches_oracle(zk, worker, log_n):
    """(round 5) mi355zk_bn254_fr_ntt_scaled_dev: a[i] *= g^i, X = NTT_omega(a), X[k] *= c * h^k for ANY g, c, h (distribute_powers takes any g:
    domain.rs:176-189) against the oracle's serial_fft with the scalings done by its field product.  Seven factor sets on ONE root: the
    per-root store of folded tables holds four, so the later ones evict the earlier ones, and the first set is run again at the end."""
    import ctypes as C

    import bn254_model as M
    import torch

    L = zk.lib.load()
    n = 1 << log_n
    r = M.R_ORDER
    omega = pow(M.FR_ROOT_OF_UNITY, 1 << (M.FR_S - log_n), r)
    a = inputs.random_fr_mont(n, seed=1300 + log_n)
    rng = np.random.default_rng(1400 + log_n)
    rnd = lambda: int.from_bytes(rng.bytes(32), "little") % (r - 2) + 2
    sets = [(rnd(), None, None), (None, rnd(), None), (None, rnd(), rnd()), (rnd(), rnd(), rnd()), (None, None, rnd()), (rnd(), rnd(), None), (r - 1, 1, r - 1)]
    if log_n >= 20:
        sets = sets[:5] if log_n == 20 else sets[2:4]
    sets = sets + [sets[0]]
    for g, c, h in sets:
        x = a
        if g is not None:
            x = O.fe_mul_many(O.FR, x, _powers(g, n)).reshape(-1, 4)
        x = O.fr_serial_fft(x, log_n, _mont(omega)).reshape(-1, 4) if log_n < 20 else O.fr_parallel_fft(x, log_n, _mont(omega), 3).reshape(-1, 4)
        if h is not None:
            x = O.fe_mul_many(O.FR, x, _powers(h, n)).reshape(-1, 4)
        if c is not None:
            x = O.fe_mul_many(O.FR, x, np.tile(_mont(c), (n, 1))).reshape(-1, 4)
        d = torch.from_numpy(a.view(np.int64)).cuda()
        keep = [_mont(v) if v is not None else None for v in (omega, g, c, h)]
        ptr = lambda k: None if k is None else k.ctypes.data_as(C.c_void_p)
        rc = L.mi355zk_bn254_fr_ntt_scaled_dev(C.c_void_p(d.data_ptr()), log_n, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), None)
        assert rc == 0
        assert np.array_equal(d.cpu().numpy().view(np.uint64), x), (log_n, g is not None, c is not None, h is not None)


def test_divide_by_z_on_coset_and_z(zk, worker):
    """domain.rs:207-234: z(tau) = tau^m - 1; divide_by_z_on_coset multiplies every coefficient by z(g)^-1, g = 7."""
    import torch

    import bn254_model as M

    log_n = 12
    a = inputs.random_fr_mont(1 << log_n, seed=500)
    dom = zk.EvaluationDomain(torch.from_numpy(a.view(np.int64)).cuda(), log_n)
    r = M.R_ORDER
    tau = 0x123456789ABCDEF0123 % r
    mont = lambda v: np.array(M.to_limbs(v * M.MONT_R % r), dtype=np.uint64)  # noqa: E731
    assert np.array_equal(dom.z(mont(tau)), mont((pow(tau, 1 << log_n, r) - 1) % r))
    dom.divide_by_z_on_coset(worker)
    zinv = mont(pow((pow(7, 1 << log_n, r) - 1) % r, -1, r))
    want = O.fe_mul_many(O.FR, a, np.tile(zinv, (1 << log_n, 1))).reshape(-1, 4)
    assert np.array_equal(dom.coeffs.cpu().numpy().view(np.uint64), want)


def test_table_cache_drop_between_the_lookups_of_one_transform(zk, worker):
    """The (device, size, root) table cache holds 64 entries per device and drops them all when a transform needs room.  A coset
    transform on a fresh size looks up two new tables (omega and g): the drop must happen BEFORE the first lookup, never between the two
    (the tables the first one returned would be freed under the launch -- found by tools/fuzz_ntt.py once a process had used more than
    64 (size, root) pairs).  Two sweeps of coset_fft / icoset_fft over 2^1 .. 2^17 (two new entries per call, 68 per sweep) with a
    one-entry call between them, so that the limit is met at both parities of the count whatever the earlier tests left in the cache;
    every transform is checked against the oracle."""
    def run(log_n, op, seed):
        a = inputs.random_fr_mont(1 << log_n, seed=seed)
        dom = zk.EvaluationDomain.from_coeffs(a)
        getattr(dom, op)(worker)
        assert np.array_equal(dom.into_coeffs(), O.fr_domain_op(a, log_n, op).reshape(-1, 4)), (log_n, op)

    for sweep in range(2):
        for log_n in range(1, 18):
            run(log_n, "coset_fft", 7000 + 100 * sweep + log_n)
            run(log_n, "icoset_fft", 7050 + 100 * sweep + log_n)
        run(18, "fft", 7300 + sweep)         # one entry: the next sweep meets the limit at the other parity
    run(18, "coset_fft", 7400)
    run(19, "icoset_fft", 7401)


def test_scratch_buffers_are_bounded_over_many_streams(zk, worker):
    """The inter-pass scratch of a transform (and the Z / table scratch of batch_exp) is kept per (device, stream); a caller that
    makes a stream per call must not pin a buffer per stream for ever: past 16 streams the buffers are dropped (device idle) and made
    again.  40 streams, a two-pass transform and a batch_exp on each, results against the oracle."""
    import torch

    import bn254_model as M

    log_n = 12
    a = inputs.random_fr_mont(1 << log_n, seed=7500)
    want = O.fr_domain_op(a, log_n, "coset_fft").reshape(-1, 4)
    bases = inputs.bases_progression_cpu(1, 64, seed=7501)
    k = np.array([M.to_limbs(0x1234567890ABCDEF1234567890ABCDEF)], dtype=np.uint64)
    want_exp = np.stack([O.G1.to_affine(O.G1.mul(O.G1.from_affine(bases[i]), k[0])) for i in (0, 63)])
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    d_k = torch.from_numpy(k.view(np.int64)).cuda()
    streams = [torch.cuda.Stream() for _ in range(40)]
    for s in streams:
        with torch.cuda.stream(s):
            dom = zk.EvaluationDomain.from_coeffs(a)
            dom.coset_fft(worker)
            got = dom.into_coeffs()
            out = zk.ceremony.batch_exp(d_b, d_k, same_scalar=True)
            s.synchronize()
        assert np.array_equal(got, want)
        assert np.array_equal(out.cpu().numpy().view(np.uint64)[[0, 63]], want_exp)
