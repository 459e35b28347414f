"""SURVEY 8(f) row 4, point codecs: the HIP encode / decode kernels against the oracle's restatement of
pairing/src/bn256/ec.rs:763-946, 1136-1344, byte for byte, through the C ABI -- valid points, infinity, both roots,
every GroupDecodingError the reference raises (first failing record wins), and the Fq2::sqrt quirk."""
import ctypes as C

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _encode(zk, group, pts, compressed):
    import torch

    d_in = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()
    n = pts.shape[0]
    d_out = torch.zeros((n, O.ENC_SIZE[(group, compressed)]), dtype=torch.uint8, device="cuda")
    fn = zk.lib.load().mi355zk_bn254_g1_encode_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_encode_dev
    assert fn(C.c_void_p(d_out.data_ptr()), C.c_void_p(d_in.data_ptr()), n, 1 if compressed else 0, None) == 0
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def _decode(zk, group, data, compressed, checked=True):
    import torch

    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.shape[0]
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.full((n, 8 * group), -1, dtype=torch.int64, device="cuda")
    err = C.c_longlong(-7)
    fn = zk.lib.load().mi355zk_bn254_g1_decode_dev if group == 1 else zk.lib.load().mi355zk_bn254_g2_decode_dev
    rc = fn(C.c_void_p(d_out.data_ptr()), C.c_void_p(d_in.data_ptr()), n, 1 if compressed else 0, 1 if checked else 0, None, C.byref(err))
    return rc, err.value, d_out.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("compressed", [False, True])
def test_encode_decode_match_oracle(zk, worker, group, compressed):
    n = 700 if group == 1 else 150
    pts = inputs.bases_progression_cpu(group, n, seed=600 + group)
    pts[3] = 0
    pts[n - 1] = 0
    want = O.encode_points(group, pts, compressed)
    got = _encode(zk, group, pts, compressed)
    assert np.array_equal(got, want)
    rc, idx, dec = _decode(zk, group, want, compressed)
    assert (rc, idx) == (0, -1)
    assert np.array_equal(dec, pts)
    if compressed:  # the other root
        flipped = want.copy()
        flipped[:, 0] ^= 0x80
        flipped[3, 0] = flipped[n - 1, 0] = 0x40
        rc_o, _, dec_o = O.decode_points(group, flipped, True)
        rc, idx, dec = _decode(zk, group, flipped, True)
        assert rc == rc_o == 0 and np.array_equal(dec, dec_o)


@pytest.mark.parametrize("group", [1, 2])
def test_decode_errors_match_oracle(zk, worker, group):
    usz, csz = O.ENC_SIZE[(group, False)], O.ENC_SIZE[(group, True)]
    good = O.encode_points(group, inputs.bases_progression_cpu(group, 40, seed=620 + group), False)
    q_be = np.frombuffer(int(M.Q).to_bytes(32, "big"), np.uint8)
    cases = []
    b = good.copy(); b[17] = 0; b[17, 0] = 0x40; b[17, usz - 1] = 1; cases.append((b, False, True))      # UnexpectedInformation
    b = good.copy(); b[9, 0] |= 0x80; cases.append((b, False, True))                                       # 8 (G1) / 7 (G2)
    b = good.copy(); b[30, :32] = q_be; cases.append((b, False, True))                                     # CoordinateDecodingError (x / x.c1)
    b = good.copy(); b[30, usz - 32:] = q_be; cases.append((b, False, True))                               # CoordinateDecodingError (y / y.c0)
    b = good.copy(); b[5, usz - 1] ^= 1; cases.append((b, False, True)); cases.append((b, False, False))   # NotOnCurve iff checked
    b = good.copy(); b[5, usz - 1] ^= 1; b[2, 0] |= 0x80; b[33, :32] = q_be; cases.append((b, False, True))  # first failure wins
    for data, compressed, checked in cases:
        rc_o, idx_o, dec_o = O.decode_points(group, data, compressed, checked)
        rc, idx, dec = _decode(zk, group, data, compressed, checked)
        assert (rc, idx) == (rc_o, idx_o)
        if rc == 0:
            assert np.array_equal(dec, dec_o)
        else:  # records before the first failure are decoded; the failing record is infinity
            assert np.array_equal(dec[:idx], dec_o[:idx]) and not dec[idx].any()
    # compressed: an x with no point (G1: NotOnCurve; G2: the reference's sqrt quirk returns an off-curve "point")
    comp = O.encode_points(group, inputs.bases_progression_cpu(group, 8, seed=640), True)
    if group == 1:
        x = next(v for v in range(1, 100) if pow((v**3 + 3) % M.Q, (M.Q - 1) // 2, M.Q) == M.Q - 1)
        comp[6] = np.frombuffer(int(x).to_bytes(32, "big"), np.uint8)
    else:
        comp[6] = 0
        comp[6, csz - 1] = 5  # x = 5: x^3 + b' is a non-residue or not -- the oracle decides, the kernel must agree
        comp[2] = 0
        comp[2, csz - 1] = 1
        comp[4] = 0
        comp[4, 31] = 1       # x = u
    rc_o, idx_o, dec_o = O.decode_points(group, comp, True)
    rc, idx, dec = _decode(zk, group, comp, True)
    assert (rc, idx) == (rc_o, idx_o)
    if group == 1:
        assert (rc, idx) == (4, 6)
    else:
        assert rc == 0 and np.array_equal(dec, dec_o)


def test_decode_is_the_inverse_of_encode_at_size(zk, worker):
    """2^16 G1 points produced on the device: decode(encode(P)) == P in both modes (size-independent round trip)."""
    import torch

    n = 1 << 16
    L = zk.lib.load()
    k = torch.from_numpy(inputs.random_scalars(n, seed=660).view(np.int64)).cuda()
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    for compressed in (0, 1):
        enc = torch.zeros((n, 32 if compressed else 64), dtype=torch.uint8, device="cuda")
        back = torch.zeros_like(pts)
        assert L.mi355zk_bn254_g1_encode_dev(C.c_void_p(enc.data_ptr()), C.c_void_p(pts.data_ptr()), n, compressed, None) == 0
        err = C.c_longlong(0)
        assert L.mi355zk_bn254_g1_decode_dev(C.c_void_p(back.data_ptr()), C.c_void_p(enc.data_ptr()), n, compressed, 1, None, C.byref(err)) == 0
        assert err.value == -1 and torch.equal(back, pts)
