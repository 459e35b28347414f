"""Host-side mirror of the reference interface (phase2-bn254_amd/bellman.py, shard.py): everything that
does not need a GPU."""
import numpy as np
import pytest

import golden_util as GU


def test_density_tracker_matches_reference_semantics():
    import phase2_bn254_amd as zk

    d = zk.DensityTracker()  # source.rs:120-140
    for _ in range(70):
        d.add_element()
    for i in (0, 3, 3, 31, 32, 69):
        d.inc(i)
    assert d.get_total_density() == 5 and d.get_query_size() == 70
    words, bits = d.words()
    assert bits == 70 and np.array_equal(words, GU.density_words([i in (0, 3, 31, 32, 69) for i in range(70)]))
    assert zk.FullDensity().get_query_size() is None and zk.FullDensity().words() == (None, 0)


def test_evaluation_domain_from_coeffs():
    import phase2_bn254_amd as zk

    for n, exp in ((0, 0), (1, 0), (2, 1), (3, 2), (5, 3), (1024, 10), (1025, 11)):
        dom = zk.EvaluationDomain.from_coeffs(np.ones((n, 4), dtype=np.uint64))  # domain.rs:52-99
        assert dom.exp == exp and dom.coeffs.shape == (1 << exp, 4)
        assert (dom.coeffs[:n] == 1).all() and not dom.coeffs[n:].any()  # padded with group_zero (:89)

    class Huge:  # a coefficient vector longer than 2^28 - 1 without allocating it
        shape = (1 << 28, 4)

    with pytest.raises(zk.SynthesisError) as e:
        zk.EvaluationDomain.from_coeffs(Huge())
    assert e.value.kind == zk.SynthesisError.POLYNOMIAL_DEGREE_TOO_LARGE  # domain.rs:66-68


def test_multiexp_asserts_query_size():
    import phase2_bn254_amd as zk

    d = zk.DensityTracker.from_bools([1, 0, 1])
    with pytest.raises(AssertionError):  # multiexp.rs:347-352
        zk.multiexp(None, (np.zeros((2, 8), np.uint64), 0), d, np.zeros((4, 4), np.uint64))


def test_shard_ranges_cover_and_density_offsets():
    import phase2_bn254_amd as zk

    for n, world in ((100, 8), (1 << 20, 8), (7, 2), (5, 8)):
        rs = [zk.shard.shard_range(n, world, r) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    bits = [1, 0, 1, 1, 0, 0, 1, 1]
    assert zk.shard.density_base_offsets(bits, 2) == [0, 3] and zk.shard.density_base_offsets(bits, 4) == [0, 1, 3, 3]


def test_shard_plan_covers_every_rank_once():
    import phase2_bn254_amd as zk

    assert [zk.shard.plan(w) for w in (1, 2, 4, 8, 16, 3, 6)] == [(1, 1), (1, 2), (1, 4), (2, 4), (4, 4), (3, 1), (6, 1)]
    for world in (1, 2, 4, 8, 16, 5):
        cells = set()
        for rank in range(world):
            pg, p, wg, w = zk.shard.rank_groups(world, rank)
            assert pg * wg == world and 0 <= p < pg and 0 <= w < wg
            cells.add((p, w))
        assert len(cells) == world


def test_parameter_file_reader_fails_on_truncation_like_read_exact():
    """`_Reader.u32` / `.raw` (ceremony.py): the reference's read_u32 / read_exact return UnexpectedEof on a truncated file
    (bellman/src/groth16/mod.rs:296-383, phase2/src/parameters.rs:683-706); slicing a tensor past its end silently returns fewer
    bytes and int.from_bytes(b"") is 0, so every read checks the remaining length first."""
    import torch

    import phase2_bn254_amd as zk

    data = torch.arange(10, dtype=torch.uint8)
    rd = zk.ceremony._Reader(data)
    assert rd.u32() == 0x00010203 and rd.off == 4
    assert bytes(rd.raw(4).numpy()) == bytes([4, 5, 6, 7])
    with pytest.raises(ValueError, match="too short"):
        rd.u32()          # two bytes left
    with pytest.raises(ValueError, match="too short"):
        rd.raw(3)
    assert bytes(rd.raw(2).numpy()) == bytes([8, 9])
    with pytest.raises(ValueError, match="too short"):
        rd.raw(1)
    with pytest.raises(ValueError, match="too short"):
        zk.ceremony._Reader(torch.zeros(0, dtype=torch.uint8)).u32()
    with pytest.raises(ValueError, match="too short"):
        zk.ceremony._be32(data, 8)
