"""Host-side mirror of the ceremony-side callers next to the hot path; thin wrappers over the
C ABI for device-resident data (torch CUDA tensors of dtype int64/uint8), same names and argument meaning as the
reference functions they replace:

  batch_exp(bases, exps)                   powersoftau/src/batched_accumulator.rs:1130-1181 (exp_i * coeff folded by the caller)
  batch_exp(bases, coeff, same_scalar)     phase2/src/parameters.rs:423-470 (every point by delta^-1)
  dense_multiexp(bases, exponents)         powersoftau/src/utils.rs:189-292
  merge_pairs(v1, v2, rho)                 powersoftau/src/utils.rs:112-131, phase2/src/utils.rs:59-105 (rho drawn by the caller)
  power_pairs(v, rho)                      powersoftau/src/utils.rs:133-135
  eval_qap(bases, row_ptr, col, coeff)     the per-variable sums of MPCParameters::new, phase2/src/parameters.rs:225-294
  point_fft / point_ifft(points)           EvaluationDomain<Point<G>>::{fft, ifft}, powersoftau/src/bin/prepare_phase2.rs:68-131
  decode_points / encode_points            EncodedPoint::{into_affine[_unchecked], from_affine}, pairing/src/bn256/ec.rs:763-946, 1136-1344

Flows assembled from them:  read_accumulator / write_accumulator, read_phase1radix2m / write_phase1radix2m (file containers),
prepare_phase2 (prepare_phase2.rs:60-160), contribute_accumulator (compute_constrained), eval_qap_polynomials (MPCParameters::new eval).

Points are raw affine records (n x 8 int64 for G1, n x 16 for G2; all-zero = infinity), scalars canonical FrRepr
(n x 4 int64).  The group (1 or 2) is taken from the record width.  Work is issued on torch's current stream.

trusted_subgroup (batch_exp, eval_qap, point_fft / point_ifft and the flows over them): the caller's PROMISE that every G2 record lies
in the order-r subgroup (MI355ZK_G2_TRUSTED_SUBGROUP; honest ceremony data does, g2_subgroup_check establishes it) -- the G2 kernels then
split their scalars over the twist's endomorphism and run ~1.3x faster.  Default False: plain windows, the reference's wNAF result
(pairing/src/wnaf.rs:4-71) for EVERY record its decoders admit.  Ignored for G1.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .bellman import DeviceError, _stream_ptr


class GroupDecodingError(Exception):
    """pairing/src/lib.rs GroupDecodingError, as raised by EncodedPoint::into_affine."""

    KINDS = {4: "NotOnCurve", 5: "NotInSubgroup", 6: "CoordinateDecodingError", 7: "UnexpectedCompressionMode", 8: "UnexpectedInformation"}

    def __init__(self, code: int, index: int):
        super().__init__(f"{self.KINDS.get(code, code)} at point {index}")
        self.kind = self.KINDS.get(code, str(code))
        self.index = index


def _p(t):
    return C.c_void_p(t.data_ptr())


def _group(points) -> int:
    w = points.shape[-1]
    if w not in (8, 16):
        raise ValueError("points must be (n, 8) G1 or (n, 16) G2 raw affine records")
    return w // 8


def _fn(name: str, group: int):
    return getattr(_lib.load(), f"mi355zk_bn254_g{group}_{name}")


def _check(rc: int, what: str):
    if rc != 0:
        raise DeviceError(f"{what} failed (rc={rc})")


def _mode(first: bool, trusted_subgroup: bool) -> int:
    return (1 if first else 0) | (_lib.G2_TRUSTED_SUBGROUP if trusted_subgroup else 0)


def batch_exp(bases, exps, same_scalar: bool = False, trusted_subgroup: bool = False):
    """out[i] = exps[i] * bases[i]  (same_scalar: exps is one scalar applied to every point); affine, normalised."""
    import torch

    g = _group(bases)
    out = torch.empty_like(bases)
    _check(_fn("batch_exp_dev", g)(_p(out), _p(bases), _p(exps), bases.shape[0], _mode(same_scalar, trusted_subgroup), _stream_ptr()), "batch_exp")
    return out


def batch_exp_host(bases: np.ndarray, exps: np.ndarray, same_scalar: bool = False, trusted_subgroup: bool = False) -> np.ndarray:
    """batch_exp on HOST arrays ((n, 8) / (n, 16) u64 records, (n, 4) or (1, 4) u64 scalars): mi355zk_bn254_g{1,2}_batch_exp, which spreads
    the points over the device set of the last Worker (contiguous point ranges, no exchange) -- phase2 `contribute` in ONE process on N GPUs."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    exps = np.ascontiguousarray(exps, dtype=np.uint64)
    g = {8: 1, 16: 2}[bases.shape[1]]
    out = np.empty_like(bases)
    fn = _lib.load().mi355zk_bn254_g1_batch_exp if g == 1 else _lib.load().mi355zk_bn254_g2_batch_exp
    _check(fn(out.ctypes.data_as(C.c_void_p), bases.ctypes.data_as(C.c_void_p), exps.ctypes.data_as(C.c_void_p), bases.shape[0], _mode(same_scalar, trusted_subgroup)),
           "batch_exp (host buffers)")
    return out


def _to_host_point(g: int):
    return np.zeros(12 * g, dtype=np.uint64)


def dense_multiexp(bases, exponents) -> np.ndarray:
    """sum_i exponents[i] * bases[i] with powersoftau's contract (infinity bases add nothing); Jacobian limbs on the host."""
    g = _group(bases)
    out = _to_host_point(g)
    _check(_fn("dense_multiexp_dev", g)(_p(bases), _p(exponents), bases.shape[0], _stream_ptr(), out.ctypes.data_as(C.c_void_p)), "dense_multiexp")
    return out


def merge_pairs(v1, v2, rho):
    """(sum rho_i * v1[i], sum rho_i * v2[i]) from one digit extraction and one sort; Jacobian limbs on the host."""
    g = _group(v1)
    s, sx = _to_host_point(g), _to_host_point(g)
    _check(_fn("merge_pairs_dev", g)(_p(v1), _p(v2), _p(rho), v1.shape[0], _stream_ptr(), s.ctypes.data_as(C.c_void_p),
                                     sx.ctypes.data_as(C.c_void_p)), "merge_pairs")
    return s, sx


def merge_pairs_host(v1: np.ndarray, v2: np.ndarray, rho: np.ndarray):
    """merge_pairs on HOST arrays: mi355zk_bn254_g{1,2}_merge_pairs -- pieces of 2^22 points over the device set of the last Worker, partials
    added on the host (powersoftau's verify_transform in ONE process on N GPUs).  v1 / v2 may be overlapping views (power_pairs)."""
    g = {8: 1, 16: 2}[v1.shape[1]]
    assert v1.flags["C_CONTIGUOUS"] and v2.flags["C_CONTIGUOUS"] and v1.dtype == np.uint64 and v2.dtype == np.uint64 and v1.shape == v2.shape
    rho = np.ascontiguousarray(rho, dtype=np.uint64)
    s, sx = _to_host_point(g), _to_host_point(g)
    fn = _lib.load().mi355zk_bn254_g1_merge_pairs if g == 1 else _lib.load().mi355zk_bn254_g2_merge_pairs
    _check(fn(v1.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p), rho.ctypes.data_as(C.c_void_p), v1.shape[0], s.ctypes.data_as(C.c_void_p),
              sx.ctypes.data_as(C.c_void_p)), "merge_pairs (host buffers)")
    return s, sx


def dense_multiexp_host(bases: np.ndarray, exponents: np.ndarray) -> np.ndarray:
    """dense_multiexp on HOST arrays (mi355zk_bn254_g{1,2}_dense_multiexp), over the device set like merge_pairs_host."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    exponents = np.ascontiguousarray(exponents, dtype=np.uint64)
    g = {8: 1, 16: 2}[bases.shape[1]]
    out = _to_host_point(g)
    fn = _lib.load().mi355zk_bn254_g1_dense_multiexp if g == 1 else _lib.load().mi355zk_bn254_g2_dense_multiexp
    _check(fn(bases.ctypes.data_as(C.c_void_p), exponents.ctypes.data_as(C.c_void_p), bases.shape[0], out.ctypes.data_as(C.c_void_p)), "dense_multiexp (host buffers)")
    return out


def power_pairs(v, rho):
    """merge_pairs(v[:-1], v[1:]) (utils.rs:133-135); rho has len(v) - 1 scalars."""
    return merge_pairs(v[:-1], v[1:], rho)


def eval_qap(bases, row_ptr, col, coeff, trusted_subgroup: bool = False):
    """out[v] = sum over the terms t of variable v of coeff[t] * bases[col[t]] (CSR: row_ptr int32 (n_rows + 1), col int32), affine."""
    import torch

    g = _group(bases)
    n_rows = row_ptr.shape[0] - 1
    for name, t in (("row_ptr", row_ptr), ("col", col)):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous int32 tensor")   # (an int64 tensor would be read as pairs of int32)
    if not (bases.is_contiguous() and coeff.is_contiguous()) or coeff.shape[0] != col.shape[0]:
        raise ValueError("bases / coeff must be contiguous, one coefficient per term")
    out = torch.zeros((n_rows, 8 * g), dtype=bases.dtype, device=bases.device)
    rc = _fn("sparse_matvec_dev", g)(_p(out), _p(bases), bases.shape[0], _p(row_ptr), _p(col), _p(coeff), n_rows, col.shape[0], _stream_ptr(),
                                     _mode(False, trusted_subgroup))
    if rc == _lib.ERR_BAD_ARGS:
        raise ValueError("eval_qap: a column index is out of range or row_ptr is not a CSR offset array")
    _check(rc, "eval_qap")
    return out


def eval_qap_host(bases: np.ndarray, row_ptr: np.ndarray, col: np.ndarray, coeff: np.ndarray, trusted_subgroup: bool = False) -> np.ndarray:
    """eval_qap on HOST arrays (bases (n, 8|16) u64, row_ptr / col uint32, coeff (nnz, 4) u64): mi355zk_bn254_g{1,2}_sparse_matvec, which gives
    every device of the last Worker's set a contiguous range of rows (MPCParameters::new in ONE process on N GPUs)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    coeff = np.ascontiguousarray(coeff, dtype=np.uint64)
    g = {8: 1, 16: 2}[bases.shape[1]]
    n_rows = row_ptr.shape[0] - 1
    out = np.zeros((n_rows, 8 * g), dtype=np.uint64)
    fn = _lib.load().mi355zk_bn254_g1_sparse_matvec if g == 1 else _lib.load().mi355zk_bn254_g2_sparse_matvec
    rc = fn(out.ctypes.data_as(C.c_void_p), bases.ctypes.data_as(C.c_void_p), bases.shape[0], row_ptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
            coeff.ctypes.data_as(C.c_void_p), n_rows, col.shape[0], _mode(False, trusted_subgroup))
    if rc == _lib.ERR_BAD_ARGS:
        raise ValueError("eval_qap: a column index is out of range or row_ptr is not a CSR offset array")
    _check(rc, "eval_qap (host buffers)")
    return out


def _point_fft(points, inverse: int, trusted_subgroup: bool = False):
    g = _group(points)
    n = points.shape[0]
    if n == 0 or n & (n - 1):
        raise ValueError("the number of points must be a power of two")
    _check(_fn("point_fft_dev", g)(_p(points), n.bit_length() - 1, _mode(bool(inverse), trusted_subgroup), _stream_ptr()), "point fft")
    return points


def point_fft(points, trusted_subgroup: bool = False):
    """in place; returns its argument"""
    return _point_fft(points, 0, trusted_subgroup)


def point_ifft(points, trusted_subgroup: bool = False):
    """in place, including the 1/m scaling and the normalisation to affine (prepare_phase2.rs:102-131)"""
    return _point_fft(points, 1, trusted_subgroup)


_ENC_SIZE = {(1, False): 64, (1, True): 32, (2, False): 128, (2, True): 64}


def g2_subgroup_check(points) -> int:
    """mi355zk_bn254_g2_subgroup_check_dev: -1 if every (n, 16) G2 record lies in the order-r subgroup, else the lowest index that
    does not.  Not a reference function (pairing_ce's bn256 decoders check the curve equation only, ec.rs:1136-1344): it establishes
    the precondition of the G2 scalar-multiplication kernels (their psi split is exact in the subgroup only) for untrusted data."""
    import ctypes as C

    assert _group(points) == 2
    bad = C.c_longlong(-1)
    _check(_lib.load().mi355zk_bn254_g2_subgroup_check_dev(_p(points.contiguous()), points.shape[0], _stream_ptr(), C.byref(bad)), "g2_subgroup_check")
    return int(bad.value)


def encode_points(points, compressed: bool):
    import torch

    g = _group(points)
    out = torch.zeros((points.shape[0], _ENC_SIZE[(g, bool(compressed))]), dtype=torch.uint8, device=points.device)
    _check(_fn("encode_dev", g)(_p(out), _p(points), points.shape[0], 1 if compressed else 0, _stream_ptr()), "encode_points")
    return out


def decode_points(data, group: int, compressed: bool, checked: bool = True):
    """(n, size) uint8 wire records -> (n, 8 * group) raw affine records; raises GroupDecodingError for the first bad record."""
    import torch

    size = _ENC_SIZE[(group, bool(compressed))]
    if data.shape[-1] != size:
        raise ValueError(f"expected records of {size} bytes")
    out = torch.empty((data.shape[0], 8 * group), dtype=torch.int64, device=data.device)
    idx = C.c_longlong(-1)
    rc = _fn("decode_dev", group)(_p(out), _p(data), data.shape[0], 1 if compressed else 0, 1 if checked else 0, _stream_ptr(), C.byref(idx))
    if rc in GroupDecodingError.KINDS:
        raise GroupDecodingError(rc, idx.value)
    _check(rc, "decode_points")
    return out


# ---------------------------------------------------------------------------------------------------------------
# File containers around the codecs: byte layouts only -- the payload goes through the codec kernels.
#   accumulator / challenge / response body   powersoftau/src/batched_accumulator.rs:88-170 (calculate_mmap_position),
#                                             powersoftau/src/parameters.rs:74-105: 64-byte hash, then TauG1[2^(p+1) - 1],
#                                             TauG2[2^p], AlphaG1[2^p], BetaG1[2^p], BetaG2[1]; every element compressed or not
#   phase1radix2m{p}                          written by powersoftau/src/bin/prepare_phase2.rs:160-240, read by
#                                             phase2/src/parameters.rs:147-217: alpha_g1, beta_g1, beta_g2, coeffs_g1[m],
#                                             coeffs_g2[m], alpha_coeffs_g1[m], beta_coeffs_g1[m], h[m - 1], all uncompressed
class DeserializationError(Exception):
    """powersoftau/src/utils.rs DeserializationError::PointAtInfinity / parameters.rs:164 "point at infinity"."""


HASH_SIZE = 64  # parameters.rs:74

_ACC_FIELDS = (("tau_g1", 1, lambda p: (2 << p) - 1), ("tau_g2", 2, lambda p: 1 << p), ("alpha_g1", 1, lambda p: 1 << p),
               ("beta_g1", 1, lambda p: 1 << p), ("beta_g2", 2, lambda p: 1))


def accumulator_layout(power: int, compressed: bool):
    """[(name, group, count, byte offset)], total bytes (without a response's trailing public key)."""
    off, out = HASH_SIZE, []
    for name, g, cnt in _ACC_FIELDS:
        out.append((name, g, cnt(power), off))
        off += cnt(power) * _ENC_SIZE[(g, bool(compressed))]
    return out, off


def _no_infinity(points, what: str):
    if bool((points == 0).all(dim=1).any().item()):
        raise DeserializationError(f"PointAtInfinity in {what}")
    return points


def read_accumulator(data, power: int, compressed: bool, checked: bool = True):
    """data: 1-D uint8 device tensor holding the file.  Returns {"hash": 64 bytes, "tau_g1": (2^(p+1)-1, 8), ...} of raw affine
    device records; GroupDecodingError for an undecodable element, DeserializationError for the point at infinity
    (batched_accumulator.rs read_points_chunk)."""
    layout, total = accumulator_layout(power, compressed)
    if data.numel() < total:
        raise ValueError(f"accumulator of power {power} needs {total} bytes")
    out = {"hash": data[:HASH_SIZE].clone()}
    for name, g, cnt, off in layout:
        sz = _ENC_SIZE[(g, bool(compressed))]
        out[name] = _no_infinity(decode_points(data[off:off + cnt * sz].view(cnt, sz), g, compressed, checked), name)
    return out


def write_accumulator(acc, compressed: bool):
    """The inverse of read_accumulator: one uint8 device tensor (hash, then the five vectors)."""
    import torch

    power = acc["tau_g2"].shape[0].bit_length() - 1
    layout, total = accumulator_layout(power, compressed)
    data = torch.zeros(total, dtype=torch.uint8, device=acc["tau_g1"].device)
    data[:HASH_SIZE] = acc["hash"]
    for name, g, cnt, off in layout:
        if acc[name].shape[0] != cnt:
            raise ValueError(f"{name}: {acc[name].shape[0]} elements, layout wants {cnt}")
        enc = encode_points(acc[name], compressed)
        data[off:off + enc.numel()] = enc.reshape(-1)
    return data


_RADIX_FIELDS = (("alpha_g1", 1, lambda m: 1), ("beta_g1", 1, lambda m: 1), ("beta_g2", 2, lambda m: 1), ("coeffs_g1", 1, lambda m: m),
                 ("coeffs_g2", 2, lambda m: m), ("alpha_coeffs_g1", 1, lambda m: m), ("beta_coeffs_g1", 1, lambda m: m),
                 ("h", 1, lambda m: m - 1))


def read_phase1radix2m(data, m: int):
    """data: 1-D uint8 device tensor of a phase1radix2m{log2 m} file -> dict of raw affine device records (parameters.rs:147-217:
    into_affine_unchecked + the point-at-infinity test)."""
    off, out = 0, {}
    for name, g, cnt in _RADIX_FIELDS:
        n, sz = cnt(m), _ENC_SIZE[(g, False)]
        if data.numel() < off + n * sz:
            raise ValueError("phase1radix2m file too short")
        out[name] = _no_infinity(decode_points(data[off:off + n * sz].view(n, sz), g, False, checked=False), name)
        off += n * sz
    return out


def write_phase1radix2m(params):
    import torch

    return torch.cat([encode_points(params[name], False).reshape(-1) for name, _, _ in _RADIX_FIELDS])


def prepare_phase2(acc, m: int, trusted_subgroup: bool = False):
    """The device work of powersoftau/src/bin/prepare_phase2.rs:60-160 for one degree m = 2^k: Lagrange-basis conversion of the
    tau powers (four point iffts) and the H bases  h[i] = tau_g1[i + m] - tau_g1[i]  (:137-147), all normalised to affine; returns
    the dict write_phase1radix2m serialises.  `acc` is what read_accumulator returns."""
    import torch

    if m < 1 or m & (m - 1) or 2 * m - 1 > acc["tau_g1"].shape[0]:
        raise ValueError("m must be a power of two within the accumulator's powers")
    dev = acc["tau_g1"].device
    out = {"alpha_g1": acc["alpha_g1"][:1].clone(), "beta_g1": acc["beta_g1"][:1].clone(), "beta_g2": acc["beta_g2"][:1].clone()}
    for name, src in (("coeffs_g1", "tau_g1"), ("coeffs_g2", "tau_g2"), ("alpha_coeffs_g1", "alpha_g1"), ("beta_coeffs_g1", "beta_g1")):
        out[name] = point_ifft(acc[src][:m].clone(), trusted_subgroup)
    # h[i] = 1 * tau_g1[i + m] + (r - 1) * tau_g1[i]: a two-term row of the sparse matrix x point vector product
    n_h = m - 1
    r_minus_1 = [0x43E1F593F0000000, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029]
    one = [1, 0, 0, 0]
    to_i64 = lambda limbs: [v - (1 << 64) if v >= (1 << 63) else v for v in limbs]  # noqa: E731
    coeff = torch.tensor([to_i64(one), to_i64(r_minus_1)], dtype=torch.int64, device=dev).repeat(max(n_h, 1), 1)[:2 * n_h]
    idx = torch.arange(n_h, device=dev, dtype=torch.int32)
    col = torch.stack([idx + m, idx], dim=1).reshape(-1).contiguous()
    row_ptr = (2 * torch.arange(n_h + 1, device=dev, dtype=torch.int32)).contiguous()
    out["h"] = eval_qap(acc["tau_g1"][:2 * m - 1].contiguous(), row_ptr, col, coeff.contiguous()) if n_h else acc["tau_g1"][:0].clone()
    return out


_R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # fr.rs:4

# G1Affine::one() / G2Affine::one() as raw affine records (Montgomery limbs): the reference's literals
# pairing/src/bn256/fq.rs:39-50 (G1_GENERATOR_X, _Y = (1, 2)) and :60-83 (G2_GENERATOR_X_C0, X_C1, Y_C0, Y_C1)
G1_ONE_RAW = np.array([0xD35D438DC58F0D9D, 0x0A78EB28F5C70B3D, 0x666EA36F7879462C, 0x0E0A77C19A07DF2F,
                       0xA6BA871B8B1E1B3A, 0x14F1D651EB8E167B, 0xCCDD46DEF0F28C58, 0x1C14EF83340FBE5E], dtype=np.uint64)
G2_ONE_RAW = np.array([0x8E83B5D102BC2026, 0xDCEB1935497B0172, 0xFBB8264797811ADF, 0x19573841AF96503B,
                       0xAFB4737DA84C6140, 0x6043DD5A5802D8C4, 0x09E950FC52A02F86, 0x14FEF0833AEA7B6B,
                       0x619DFA9D886BE9F6, 0xFE7FD297F59E9B78, 0xFF9E1A62231B7DFE, 0x28FD7EEBAE9E4206,
                       0x64095B56C71856EE, 0xDC57F922327D3CBB, 0x55F935BE33351076, 0x0DA4A0E693FD6482], dtype=np.uint64)


def blank_hash() -> bytes:
    """powersoftau/src/utils.rs:138-140: BLAKE2b-512 of the empty string, the `hash` field of a fresh challenge."""
    import hashlib

    return hashlib.blake2b(b"", digest_size=64).digest()


def calculate_hash(data) -> bytes:
    """powersoftau/src/utils.rs:20-27: BLAKE2b-512 over the whole file (`data`: bytes, numpy uint8 or a uint8 tensor).
    Hashing is host work in the reference too (a sequential compression function; no device counterpart)."""
    import hashlib

    if hasattr(data, "cpu"):
        data = data.cpu().numpy()
    return hashlib.blake2b(np.ascontiguousarray(data, dtype=np.uint8).tobytes(), digest_size=64).digest()


def new_accumulator(power: int, device):
    """BatchedAccumulator::generate_initial (batched_accumulator.rs:1295-1347), the body of `new_constrained`: every element
    of every vector is the group generator, the hash field is blank_hash().  write_accumulator(acc, compressed=False) of the
    result is the challenge file of new_constrained.rs:42-77 (COMPRESS_NEW_CHALLENGE = No: `accumulator_size` bytes)."""
    import torch

    g1 = torch.from_numpy(G1_ONE_RAW.view(np.int64)).to(device)
    g2 = torch.from_numpy(G2_ONE_RAW.view(np.int64)).to(device)
    n = 1 << power
    return {"hash": torch.frombuffer(bytearray(blank_hash()), dtype=torch.uint8).to(device),
            "tau_g1": g1.repeat(2 * n - 1, 1), "tau_g2": g2.repeat(n, 1), "alpha_g1": g1.repeat(n, 1), "beta_g1": g1.repeat(n, 1),
            "beta_g2": g2.repeat(1, 1)}


def _limbs_i64(x: int):
    out = []
    for i in range(4):
        v = (x >> (64 * i)) & ((1 << 64) - 1)
        out.append(v - (1 << 64) if v >= (1 << 63) else v)
    return out


def scalar_powers(base: int, n: int, device, coeff: int = 1):
    """(n, 4) canonical FrRepr rows  coeff * base^i mod r,  i < n, built on the device by repeated doubling with the pointwise
    Montgomery product (what batched_accumulator.rs:1230-1260 computes per chunk with `pow` and a running product)."""
    import torch

    L = _lib.load()
    mont = lambda v: (v % _R_ORDER) * (1 << 256) % _R_ORDER  # noqa: E731
    pw = torch.empty((n, 4), dtype=torch.int64, device=device)
    pw[0] = torch.tensor(_limbs_i64(mont(coeff)), dtype=torch.int64, device=device)
    have = 1
    while have < n:
        take = min(have, n - have)
        step = torch.tensor(_limbs_i64(mont(pow(base, have, _R_ORDER))), dtype=torch.int64, device=device).repeat(take, 1)
        blk = pw[:take].clone()
        _check(L.mi355zk_bn254_fr_mul_assign_dev(_p(blk), _p(step), take, _stream_ptr()), "fr_mul_assign")
        pw[have:have + take] = blk
        have += take
    one = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=device).repeat(n, 1)
    _check(L.mi355zk_bn254_fr_mul_assign_dev(_p(pw), _p(one), n, _stream_ptr()), "fr_mul_assign")  # x R * 1 / R = x: canonical
    return pw


def contribute_accumulator(acc, tau: int, alpha: int, beta: int, trusted_subgroup: bool = False):
    """The device work of powersoftau `compute_constrained` (batched_accumulator.rs:1119-1292: every tau power by tau^i, the alpha /
    beta vectors by alpha tau^i / beta tau^i, beta_g2 by beta), `batch_exp` + normalisation per vector.  Returns a new dict."""
    dev = acc["tau_g1"].device
    n1, n = acc["tau_g1"].shape[0], acc["tau_g2"].shape[0]
    tp = scalar_powers(tau, n1, dev)
    out = {"hash": acc["hash"].clone()}
    out["tau_g1"] = batch_exp(acc["tau_g1"], tp)
    out["tau_g2"] = batch_exp(acc["tau_g2"], tp[:n].contiguous(), trusted_subgroup=trusted_subgroup)
    out["alpha_g1"] = batch_exp(acc["alpha_g1"], scalar_powers(tau, n, dev, coeff=alpha))
    out["beta_g1"] = batch_exp(acc["beta_g1"], scalar_powers(tau, n, dev, coeff=beta))
    out["beta_g2"] = batch_exp(acc["beta_g2"], scalar_powers(tau, 1, dev, coeff=beta), trusted_subgroup=trusted_subgroup)
    return out


def eval_qap_polynomials(radix, at, bt, ct, trusted_subgroup: bool = False):
    """The `eval` of MPCParameters::new (phase2/src/parameters.rs:225-300) as four sparse products over the Lagrange bases of a
    phase1radix2m file (`radix`: what read_phase1radix2m returns).  at / bt / ct: CSR triples (row_ptr int32 (n_vars + 1),
    col int32 (nnz) = Lagrange index, coeff (nnz, 4) canonical) -- the QAP polynomials of keypair_assembly.rs:15-25, one row
    per variable.  Returns affine (a_g1, b_g1, b_g2, ext) with
        a_g1[v] = sum_at c L_g1[lag],  b_g1[v] = sum_bt c L_g1[lag],  b_g2[v] = sum_bt c L_g2[lag],
        ext[v]  = sum_at c beta_L_g1[lag] + sum_bt c alpha_L_g1[lag] + sum_ct c L_g1[lag]          (ic / l before the delta split)."""
    import torch

    a_g1 = eval_qap(radix["coeffs_g1"], *at)
    b_g1 = eval_qap(radix["coeffs_g1"], *bt)
    b_g2 = eval_qap(radix["coeffs_g2"], *bt, trusted_subgroup=trusted_subgroup)
    # ext: one product over the concatenated bases [beta_L | alpha_L | L] and the row-wise concatenated term lists
    m = radix["coeffs_g1"].shape[0]
    bases = torch.cat([radix["beta_coeffs_g1"], radix["alpha_coeffs_g1"], radix["coeffs_g1"]])
    n_vars = at[0].shape[0] - 1
    lens = [(t[0][1:] - t[0][:-1]).to(torch.int64) for t in (at, bt, ct)]
    total = lens[0] + lens[1] + lens[2]
    row_ptr = torch.zeros(n_vars + 1, dtype=torch.int64, device=bases.device)
    row_ptr[1:] = torch.cumsum(total, 0)
    nnz = int(row_ptr[-1].item())
    col = torch.empty(nnz, dtype=torch.int32, device=bases.device)
    coeff = torch.empty((nnz, 4), dtype=torch.int64, device=bases.device)
    start = row_ptr[:-1].clone()
    for k, (t, ln) in enumerate(zip((at, bt, ct), lens)):
        # destination of term j of row v: start[v] + (j - t.row_ptr[v])
        rows = torch.repeat_interleave(torch.arange(n_vars, device=bases.device), ln)
        within = torch.arange(t[1].shape[0], device=bases.device) - t[0][:-1].to(torch.int64)[rows]
        dst = start[rows] + within
        col[dst] = t[1] + k * m
        coeff[dst] = t[2]
        start = start + ln
    ext = eval_qap(bases, row_ptr.to(torch.int32), col, coeff)
    return a_g1, b_g1, b_g2, ext


# ---------------------------------------------------------------------------------------------------------------
# Groth16 parameter files: bellman `Parameters::{write, read}` (bellman/src/groth16/mod.rs:104-198, 252-383) and
# phase2 `MPCParameters::{write, read}` (phase2/src/parameters.rs:661-706; PublicKey phase2/src/keypair.rs:49-120).  Layout, every
# point UNCOMPRESSED:
#   vk: alpha_g1, beta_g1 (G1), beta_g2, gamma_g2 (G2), delta_g1 (G1), delta_g2 (G2), u32 BE |ic|, ic[] (G1)
#   then five vectors, each u32 BE length + points: h (G1), l (G1), a (G1), b_g1 (G1), b_g2 (G2)
#   MPC files append cs_hash[64], u32 BE number of contributions, and per contribution delta_after, s, s_delta (G1), r_delta (G2),
#   transcript[64].
# Offsets and the length words are host work; every point goes through the codec kernels.
def _be32(data, off: int) -> int:
    if off < 0 or off + 4 > data.numel():
        raise ValueError("parameter file too short in a length word")      # read_u32: io::ErrorKind::UnexpectedEof
    return int.from_bytes(bytes(data[off:off + 4].cpu().numpy()), "big")


class _Reader:
    """Cursor over a parameter file.  Every read checks the remaining length first: slicing a tensor past its end returns
    FEWER bytes without complaint (and int.from_bytes(b"") is 0), whereas the reference's read_u32 / read_exact fail with
    UnexpectedEof on a truncated file (groth16/mod.rs:296-383, phase2/src/parameters.rs:683-706)."""

    def __init__(self, data):
        self.data, self.off = data, 0

    def _need(self, n: int, what: str):
        if n < 0 or self.off + n > self.data.numel():
            raise ValueError(f"parameter file too short in {what}")        # io::ErrorKind::UnexpectedEof

    def points(self, n: int, group: int, checked: bool, no_infinity: bool, what: str):
        sz = _ENC_SIZE[(group, False)]
        self._need(n * sz, what)
        pts = decode_points(self.data[self.off:self.off + n * sz].view(n, sz), group, False, checked)
        self.off += n * sz
        return _no_infinity(pts, what) if no_infinity else pts

    def u32(self, what: str = "a length word") -> int:
        self._need(4, what)
        v = _be32(self.data, self.off)
        self.off += 4
        return v

    def raw(self, n: int, what: str = "a hash"):
        self._need(n, what)
        out = self.data[self.off:self.off + n].clone()
        self.off += n
        return out


def read_parameters(data, disallow_points_at_infinity: bool = True, checked: bool = True, _reader=None):
    """Parameters::read (groth16/mod.rs:296-383).  data: 1-D uint8 device tensor.  Returns {"vk": {...device records...,
    "ic": (n, 8)}, "h", "l", "a", "b_g1": (n, 8), "b_g2": (n, 16)} of raw affine device records.  The verifying key is always
    read checked and its ic rejects the point at infinity (VerifyingKey::read, :143-198); the five vectors follow the two flags."""
    rd = _reader or _Reader(data)
    vk = {}
    for name, g in (("alpha_g1", 1), ("beta_g1", 1), ("beta_g2", 2), ("gamma_g2", 2), ("delta_g1", 1), ("delta_g2", 2)):
        vk[name] = rd.points(1, g, True, False, name)
    vk["ic"] = rd.points(rd.u32("the length of ic"), 1, True, True, "ic")
    out = {"vk": vk}
    for name, g in (("h", 1), ("l", 1), ("a", 1), ("b_g1", 1), ("b_g2", 2)):
        out[name] = rd.points(rd.u32(f"the length of {name}"), g, checked, disallow_points_at_infinity, name)
    return out


def write_parameters(params):
    """Parameters::write (groth16/mod.rs:253-294): one uint8 device tensor."""
    import torch

    dev = params["h"].device
    be = lambda n: torch.frombuffer(bytearray(int(n).to_bytes(4, "big")), dtype=torch.uint8).to(dev)  # noqa: E731
    enc = lambda pts: encode_points(pts.contiguous(), False).reshape(-1)  # noqa: E731
    vk = params["vk"]
    parts = [enc(vk[k]) for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2")]
    parts += [be(vk["ic"].shape[0]), enc(vk["ic"])]
    for name in ("h", "l", "a", "b_g1", "b_g2"):
        parts += [be(params[name].shape[0]), enc(params[name])]
    return torch.cat(parts)


def read_mpc_parameters(data, disallow_points_at_infinity: bool = True, checked: bool = True):
    """MPCParameters::read (phase2/src/parameters.rs:683-706): the Groth16 parameters, cs_hash and the contributions' public keys
    (every public-key point checked and never the point at infinity, keypair.rs:64-120)."""
    rd = _Reader(data)
    out = {"params": read_parameters(data, disallow_points_at_infinity, checked, _reader=rd)}
    out["cs_hash"] = rd.raw(64, "cs_hash")
    out["contributions"] = []
    for _ in range(rd.u32("the number of contributions")):
        pk = {"delta_after": rd.points(1, 1, True, True, "delta_after"), "s": rd.points(1, 1, True, True, "s"),
              "s_delta": rd.points(1, 1, True, True, "s_delta"), "r_delta": rd.points(1, 2, True, True, "r_delta")}
        pk["transcript"] = rd.raw(64, "a transcript")
        out["contributions"].append(pk)
    return out


def write_mpc_parameters(mpc):
    """MPCParameters::write (phase2/src/parameters.rs:663-678)."""
    import torch

    dev = mpc["cs_hash"].device
    enc = lambda pts: encode_points(pts.contiguous(), False).reshape(-1)  # noqa: E731
    parts = [write_parameters(mpc["params"]), mpc["cs_hash"],
             torch.frombuffer(bytearray(len(mpc["contributions"]).to_bytes(4, "big")), dtype=torch.uint8).to(dev)]
    for pk in mpc["contributions"]:
        parts += [enc(pk["delta_after"]), enc(pk["s"]), enc(pk["s_delta"]), enc(pk["r_delta"]), pk["transcript"]]
    return torch.cat(parts)


def contribute_parameters(params, delta: int):
    """The device work of MPCParameters::contribute (phase2/src/parameters.rs:414-522) for a given private delta: every point of
    l and h times delta^-1 (`batch_exp`, affine out), delta_g1 and delta_g2 times delta.  Returns new parameters; the key pair,
    its transcript hash and the public key record (keypair.rs: hash-to-G2, BLAKE2b) stay with the caller."""
    dev = params["h"].device
    import torch

    to_dev = lambda v: torch.tensor([_limbs_i64(v % _R_ORDER)], dtype=torch.int64, device=dev)  # noqa: E731
    d_inv, d = to_dev(pow(delta, -1, _R_ORDER)), to_dev(delta)
    out = dict(params)
    out["l"] = batch_exp(params["l"], d_inv, same_scalar=True)
    out["h"] = batch_exp(params["h"], d_inv, same_scalar=True)
    vk = dict(params["vk"])
    vk["delta_g1"] = batch_exp(params["vk"]["delta_g1"], d, same_scalar=True)
    vk["delta_g2"] = batch_exp(params["vk"]["delta_g2"], d, same_scalar=True)   # (one point: the plain windows)
    out["vk"] = vk
    return out
