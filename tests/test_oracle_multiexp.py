"""Oracle multiexp (restatement of bellman/src/multiexp.rs) vs the golden vectors of the independent
Python model, vs the reference tests' own check (naive sum == multiexp, multiexp.rs:479-518), and the
Source / density error contract (source.rs:36-118)."""
import numpy as np
import pytest

import golden_util as GU
import inputs
import oracle_lib as O


@pytest.mark.parametrize("group", [1, 2])
def test_golden_vectors(group):
    G = O.G1 if group == 1 else O.G2
    n_cases = 0
    for c in GU.msm_cases(group):
        dens = GU.density_words(c["density"]) if c["density"] is not None else None
        dbits = len(c["density"]) if c["density"] is not None else None
        for threads in (1, 4):
            rc, out = G.multiexp(c["bases"], c["scalars"], density=dens, density_bits=dbits, base_offset=c["base_offset"],
                                 threads=threads, n_bases=c["bases"].shape[0])
            assert rc == c["rc"], c["name"]
            if rc == 0:
                assert np.array_equal(G.to_affine(out), c["expected"]), c["name"]
        n_cases += 1
    assert n_cases >= 17


@pytest.mark.parametrize("group,n", [(1, 300), (1, 2000), (2, 200)])
def test_naive_equals_multiexp(group, n):
    """multiexp.rs:479-518 `test_with_bls12` restated on BN254: naive sum(base * exp) == multiexp."""
    G = O.G1 if group == 1 else O.G2
    bases = inputs.bases_progression_cpu(group, n, seed=77 + n)
    scalars = inputs.random_scalars(n, seed=78 + n)
    rc, fast = G.multiexp(bases, scalars, threads=4)
    assert rc == 0
    assert G.eq(fast, G.naive_multiexp(bases, scalars))


def test_window_choice_matches_reference_rule():
    # multiexp.rs:341-345: c = 3 if n < 32 else ceil(ln n)
    assert O.multiexp_window_bits(1) == 3 and O.multiexp_window_bits(31) == 3
    assert O.multiexp_window_bits(32) == 4
    assert O.multiexp_window_bits(1 << 16) == 12 and O.multiexp_window_bits(1 << 20) == 14 and O.multiexp_window_bits(1 << 26) == 19


def test_error_order_lowest_index_wins():
    """identity at index 2 and bases exhausted at index 4: the error at the lower index is reported."""
    bases = inputs.bases_cpu(1, 4, seed=5)
    bases[2] = 0
    scalars = inputs.random_scalars(6, seed=6)
    rc, _ = O.G1.multiexp(bases, scalars)
    assert rc == 1
    bases = inputs.bases_cpu(1, 4, seed=5)
    bases[3] = 0
    scalars[3] = 0  # zero scalar: identity base not looked at (multiexp.rs:95-96) -> Eof at index 4
    rc, _ = O.G1.multiexp(bases, scalars)
    assert rc == 2


def test_density_shorter_than_exponents_stops_at_zip():
    """multiexp_inner iterates exponents.zip(density) (multiexp.rs:92): extra exponents are ignored."""
    bases = inputs.bases_cpu(1, 3, seed=9)
    scalars = inputs.random_scalars(8, seed=10)
    dens = GU.density_words([1, 0, 1, 1])
    rc, a = O.G1.multiexp(bases, scalars, density=dens, density_bits=4)
    rc2, b = O.G1.multiexp(bases, scalars[:4], density=dens, density_bits=4)
    assert rc == 0 and rc2 == 0 and O.G1.eq(a, b)


@pytest.mark.parametrize("group,n,cpus", [(1, 5, 1), (1, 31, 3), (1, 700, 1), (1, 700, 5), (1, 2500, 16), (2, 300, 4)])
def test_dense_multiexp_like_powersoftau(group, n, cpus):
    """powersoftau::utils::dense_multiexp (powersoftau/src/utils.rs:189-292) restated: all `cpus` threads on one region at a time
    over chunks of n / cpus + 1 bases.  Same group element as bellman's multiexp over the same pairs (the reference's own test
    of the two shapes is `dense == sparse`, multiexp.rs:517,589) and as the naive sum; exponents 0 and 1 take their shortcuts; a
    base at infinity adds nothing (no Source, no UnexpectedIdentity)."""
    G = O.G1 if group == 1 else O.G2
    bases = inputs.bases_progression_cpu(group, n, seed=177 + n)
    scalars = inputs.random_scalars(n, seed=178 + n)
    scalars[0] = 0
    scalars[1] = [1, 0, 0, 0]
    if n > 8:
        scalars[7] = [1, 0, 0, 0]
    dense = G.dense_multiexp(bases, scalars, cpus=cpus)
    rc, sparse = G.multiexp(bases, scalars, threads=2)
    assert rc == 0 and G.eq(dense, sparse) and G.eq(dense, G.naive_multiexp(bases, scalars))
    if n > 8:
        holed = bases.copy()
        holed[3] = 0                      # infinity: skipped silently
        zeroed = scalars.copy()
        zeroed[3] = 0
        assert G.eq(G.dense_multiexp(holed, scalars, cpus=cpus), G.dense_multiexp(bases, zeroed, cpus=cpus))
