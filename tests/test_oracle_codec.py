"""The oracle's restatement of the reference's point encodings (oracle/codec.h; pairing/src/bn256/ec.rs:763-946,
1136-1344) pinned against an independent Python big-integer model of the wire format, the curve equations and the
algebraic definition of the square roots, plus the literals the reference holds (B_COEFF_FQ2 fq.rs:18-31, the
NEGATIVE_ONE quirk fq.rs:434-439).  CPU only."""
import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

Q = M.Q


def _be(x: int) -> bytes:
    return int(x).to_bytes(32, "big")


def _model_encode(group, p, compressed):
    """EncodedPoint::from_affine on integer coordinates."""
    size = O.ENC_SIZE[(group, compressed)]
    if p is None:
        return bytes([0x40]) + bytes(size - 1)
    if group == 1:
        x, y = p
        out = bytearray(_be(x) + (b"" if compressed else _be(y)))
        larger = y > (Q - y) % Q
    else:
        (x0, x1), (y0, y1) = p
        out = bytearray(_be(x1) + _be(x0) + (b"" if compressed else _be(y1) + _be(y0)))
        ny = ((Q - y0) % Q, (Q - y1) % Q)
        larger = (y1, y0) > (ny[1], ny[0])  # Fq2 order: c1 first (fq2.rs:20-31)
    if compressed and larger:
        out[0] |= 0x80
    return bytes(out)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("compressed", [False, True])
def test_encode_matches_wire_format_model_and_roundtrips(group, compressed):
    n = 24
    pts = inputs.bases_cpu(group, n, seed=500 + group)
    pts[5] = 0  # infinity
    from_raw = M.g1_affine_from_raw if group == 1 else M.g2_affine_from_raw
    enc = O.encode_points(group, pts, compressed)
    for i in range(n):
        assert bytes(enc[i]) == _model_encode(group, from_raw(pts[i]), compressed), i
    rc, idx, dec = O.decode_points(group, enc, compressed, checked=True)
    assert rc == 0 and idx == -1
    assert np.array_equal(dec, pts)
    # both roots of a compressed record decode to P and -P
    if compressed:
        flipped = enc.copy()
        flipped[:, 0] ^= 0x80
        flipped[5, 0] = 0x40
        rc, _, dec2 = O.decode_points(group, flipped, True)
        assert rc == 0
        G = O.G1 if group == 1 else O.G2
        for i in (0, 1, 7):
            s = G.add_mixed(G.from_affine(pts[i]), dec2[i])
            assert np.array_equal(G.to_affine(s), np.zeros(G.aff, np.uint64)), i  # P + (-P) = infinity


def test_generators_encode_to_the_known_coordinates():
    enc = O.encode_points(1, inputs.G1_GEN_RAW, False)[0]
    assert bytes(enc) == _be(1) + _be(2)  # G1 generator (1, 2): ec.rs G1 generator literals fq.rs:39-50
    enc2 = O.encode_points(2, inputs.G2_GEN_RAW, False)[0]
    (x0, x1), (y0, y1) = M.G2_GEN
    assert bytes(enc2) == _be(x1) + _be(x0) + _be(y1) + _be(y0)


def test_fq_sqrt_definition():
    rng = np.random.default_rng(7)
    for _ in range(20):
        a = int.from_bytes(rng.bytes(40), "big") % Q
        sq = a * a % Q
        r = O.fq_sqrt(np.array(M.to_limbs(M.to_mont(sq, Q)), np.uint64))
        assert r is not None and M.from_mont(M.from_limbs(r), Q) in (a, (Q - a) % Q)
    # Euler: a non-residue has no root (ff_derive's a0 == -1 test)
    nonres = next(v for v in range(2, 50) if pow(v, (Q - 1) // 2, Q) == Q - 1)
    assert O.fq_sqrt(np.array(M.to_limbs(M.to_mont(nonres, Q)), np.uint64)) is None
    assert np.array_equal(O.fq_sqrt(np.zeros(4, np.uint64)), np.zeros(4, np.uint64))


def test_fq2_sqrt_and_the_negative_one_quirk():
    rng = np.random.default_rng(8)
    for _ in range(10):
        a = (int.from_bytes(rng.bytes(40), "big") % Q, int.from_bytes(rng.bytes(40), "big") % Q)
        sq = M.f2_mul(a, a)
        raw = np.array(M.to_limbs(M.to_mont(sq[0], Q)) + M.to_limbs(M.to_mont(sq[1], Q)), np.uint64)
        r = O.fq2_sqrt(raw)
        got = (M.from_mont(M.from_limbs(r[:4]), Q), M.from_mont(M.from_limbs(r[4:]), Q))
        assert M.f2_mul(got, got) == sq
    # a non-residue of Fq2: norm is a non-residue of Fq.  The reference's NEGATIVE_ONE is -(2^256 mod r) mod r
    # (fq.rs:434-439) instead of -(2^256 mod q): its Fq2::sqrt answers Some(garbage) here -- reproduced.
    assert (-(2**256 % M.R_ORDER)) % M.R_ORDER == 0x2259D6B14729C0FA51E1A2470908122EF13771B2DA58A367974BC177A0000006
    nr = next((c0, 1) for c0 in range(1, 60) if pow((c0 * c0 + 1) % Q, (Q - 1) // 2, Q) == Q - 1)
    raw = np.array(M.to_limbs(M.to_mont(nr[0], Q)) + M.to_limbs(M.to_mont(nr[1], Q)), np.uint64)
    r = O.fq2_sqrt(raw)
    assert r is not None
    got = (M.from_mont(M.from_limbs(r[:4]), Q), M.from_mont(M.from_limbs(r[4:]), Q))
    assert M.f2_mul(got, got) != nr


def test_g2_coeff_b_is_the_reference_literal():
    lit_c0 = [0x3BF938E377B802A8, 0x020B1B273633535D, 0x26B7EDF049755260, 0x2514C6324384A86D]  # fq.rs:18-31
    lit_c1 = [0x38E7ECCCD1DCFF67, 0x65F0B37D93CE0D3E, 0xD749D0DD22AC00AA, 0x0141B9CE4A688D4D]
    assert [int(v) for v in O.g2_coeff_b()] == lit_c0 + lit_c1


@pytest.mark.parametrize("group", [1, 2])
def test_decode_errors_follow_the_reference(group):
    csz, usz = O.ENC_SIZE[(group, True)], O.ENC_SIZE[(group, False)]
    good = O.encode_points(group, inputs.bases_cpu(group, 3, seed=520 + group), False)
    # infinity flag with stray bits -> UnexpectedInformation (8)
    bad = good.copy(); bad[1] = 0; bad[1, 0] = 0x40; bad[1, usz - 1] = 1
    assert O.decode_points(group, bad, False)[:2] == (8, 1)
    # "greatest" bit on an uncompressed record: G1 UnexpectedInformation (ec.rs:797-801), G2 UnexpectedCompressionMode (ec.rs:1158-1161)
    bad = good.copy(); bad[2, 0] |= 0x80
    assert O.decode_points(group, bad, False)[:2] == (8 if group == 1 else 7, 2)
    # coordinate >= q -> CoordinateDecodingError (6); q itself with the flag bits masked is 0x30644e72... < 2^254
    bad = good.copy(); bad[0, :32] = np.frombuffer(_be(Q), np.uint8)
    assert O.decode_points(group, bad, False)[:2] == (6, 0)
    # off-curve y: NotOnCurve (4) only when checked
    bad = good.copy(); bad[1, usz - 1] ^= 1
    assert O.decode_points(group, bad, False, checked=True)[:2] == (4, 1)
    rc, idx, pts = O.decode_points(group, bad, False, checked=False)
    assert rc == 0 and pts[1].any()
    # first failure wins
    bad[0, 0] |= 0x80
    assert O.decode_points(group, bad, False)[1] == 0
    if group == 1:
        # compressed x with no point on the curve: x^3 + 3 a non-residue
        x = next(v for v in range(1, 100) if pow((v**3 + 3) % Q, (Q - 1) // 2, Q) == Q - 1)
        rec = np.frombuffer(_be(x), np.uint8).reshape(1, csz)
        assert O.decode_points(1, rec, True)[:2] == (4, 0)
    inf = np.zeros((1, csz), np.uint8); inf[0, 0] = 0x40
    rc, _, p = O.decode_points(group, inf, True)
    assert rc == 0 and not p.any()
