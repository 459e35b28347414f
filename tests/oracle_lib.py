"""ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All arrays are numpy uint64 (Montgomery limbs, little endian), shapes documented per function.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(_ROOT, "oracle", f) for f in ("bn254_oracle.c", "field.h", "tmpl_curve.h", "tmpl_multiexp.h", "tmpl_fft.h", "tmpl_fft_undef.h", "codec.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        for name in ("oracle_g1_multiexp", "oracle_g2_multiexp"):
            f = getattr(_lib, name)
            f.restype = C.c_int
            f.argtypes = [_u64p, C.c_size_t, C.c_size_t, _u64p, C.c_size_t, _u32p, C.c_size_t, C.c_int, _u64p]
        _lib.oracle_dummy_multiexp.restype = C.c_int
        _lib.oracle_dummy_multiexp.argtypes = [_u32p, C.c_size_t, C.c_size_t, _u64p, C.c_size_t, _u32p, C.c_size_t, _u32p]
        _lib.oracle_multiexp_window_bits.restype = C.c_uint32
        _lib.oracle_multiexp_window_bits.argtypes = [C.c_size_t]
        _lib.oracle_dummy_domain_omega.restype = C.c_uint32
    return _lib


def _p64(a):
    return a.ctypes.data_as(_u64p) if a is not None else None


def _p32(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


def _arr(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if shape is not None:
        a = a.reshape(shape)
    return a


FQ, FR = 0, 1


def fe_mul(which, a, b):
    a, b = _arr(a), _arr(b)
    r = np.zeros(4, np.uint64)
    lib().oracle_fe_mul(which, _p64(r), _p64(a), _p64(b))
    return r


def fe_mul_many(which, a, b):
    a, b = _arr(a), _arr(b)
    r = np.zeros_like(a)
    lib().oracle_fe_mul_many(which, _p64(r), _p64(a), _p64(b), C.c_size_t(a.size // 4))
    return r


def fe_add(which, a, b):
    a, b = _arr(a), _arr(b)
    r = np.zeros(4, np.uint64)
    lib().oracle_fe_add(which, _p64(r), _p64(a), _p64(b))
    return r


def fe_sub(which, a, b):
    a, b = _arr(a), _arr(b)
    r = np.zeros(4, np.uint64)
    lib().oracle_fe_sub(which, _p64(r), _p64(a), _p64(b))
    return r


def fe_inv(which, a):
    a = _arr(a)
    r = np.zeros(4, np.uint64)
    ok = lib().oracle_fe_inv(which, _p64(r), _p64(a))
    return r if ok else None


def fe_from_canonical(which, a):
    a = _arr(a)
    r = np.zeros(4, np.uint64)
    lib().oracle_fe_from_canonical(which, _p64(r), _p64(a))
    return r


def fe_to_canonical(which, a):
    a = _arr(a)
    r = np.zeros(4, np.uint64)
    lib().oracle_fe_to_canonical(which, _p64(r), _p64(a))
    return r


def fq2_mul(a, b):
    a, b = _arr(a), _arr(b)
    r = np.zeros(8, np.uint64)
    lib().oracle_fq2_mul(_p64(r), _p64(a), _p64(b))
    return r


def fq2_sqr(a):
    a = _arr(a)
    r = np.zeros(8, np.uint64)
    lib().oracle_fq2_sqr(_p64(r), _p64(a))
    return r


def fq2_inv(a):
    a = _arr(a)
    r = np.zeros(8, np.uint64)
    ok = lib().oracle_fq2_inv(_p64(r), _p64(a))
    return r if ok else None


def fr_root_of_unity():
    r = np.zeros(4, np.uint64)
    lib().oracle_fr_root_of_unity(_p64(r))
    return r


class Group:
    """g = 1 or 2.  Jacobian points: 12*g... (12 u64 for G1, 24 for G2); affine raw: 8 / 16 u64."""

    def __init__(self, g: int):
        self.g = g
        self.aff = 8 * g
        self.jac = 12 * g
        self.pre = f"oracle_g{g}_"

    def _f(self, name):
        return getattr(lib(), self.pre + name)

    def double(self, p):
        p = _arr(p).copy()
        self._f("double")(_p64(p))
        return p

    def add(self, p, o):
        p, o = _arr(p).copy(), _arr(o)
        self._f("add")(_p64(p), _p64(o))
        return p

    def add_mixed(self, p, o):
        p, o = _arr(p).copy(), _arr(o)
        self._f("add_mixed")(_p64(p), _p64(o))
        return p

    def mul(self, p, k):
        p, k = _arr(p).copy(), _arr(k)
        self._f("mul")(_p64(p), _p64(k))
        return p

    def to_affine(self, p):
        p = _arr(p)
        r = np.zeros(self.aff, np.uint64)
        self._f("to_affine")(_p64(r), _p64(p))
        return r

    def from_affine(self, a):
        a = _arr(a)
        r = np.zeros(self.jac, np.uint64)
        self._f("from_affine")(_p64(r), _p64(a))
        return r

    def eq(self, a, b):
        a, b = _arr(a), _arr(b)
        return bool(self._f("eq")(_p64(a), _p64(b)))

    def batch_normalization(self, v):
        v = _arr(v).copy()
        self._f("batch_normalization")(_p64(v), C.c_size_t(v.size // self.jac))
        return v

    def mul_many_affine(self, base_affine, ks):
        base_affine, ks = _arr(base_affine), _arr(ks)
        n = ks.size // 4
        out = np.zeros((n, self.aff), np.uint64)
        self._f("mul_many_affine")(_p64(out), _p64(base_affine), _p64(ks), C.c_size_t(n))
        return out

    def arith_progression_affine(self, start_affine, step_affine, n):
        start_affine, step_affine = _arr(start_affine), _arr(step_affine)
        out = np.zeros((n, self.aff), np.uint64)
        self._f("arith_progression_affine")(_p64(out), _p64(start_affine), _p64(step_affine), C.c_size_t(n))
        return out

    def multiexp(self, bases, scalars, density=None, density_bits=None, base_offset=0, threads=1, n_bases=None):
        """returns (rc, out_xyz).  bases (n_bases, 8g) u64; scalars (n, 4) canonical FrRepr u64;
        density: numpy uint32 words (bit i = word i/32, bit i%32) or None for FullDensity."""
        bases, scalars = _arr(bases), _arr(scalars)
        nb = bases.size // self.aff if n_bases is None else n_bases
        ns = scalars.size // 4
        if density is not None:
            density = np.ascontiguousarray(density, dtype=np.uint32)
            if density_bits is None:
                density_bits = ns
        out = np.zeros(self.jac, np.uint64)
        rc = self._f("multiexp")(_p64(bases), C.c_size_t(nb), C.c_size_t(base_offset), _p64(scalars), C.c_size_t(ns),
                                 _p32(density), C.c_size_t(density_bits or 0), C.c_int(threads), _p64(out))
        return rc, out

    def dense_multiexp(self, bases, scalars, cpus=1):
        """powersoftau::utils::dense_multiexp (powersoftau/src/utils.rs:189-292): one base per exponent, infinity bases add
        nothing; `cpus` = num_cpus::get() (threads per region)."""
        bases, scalars = _arr(bases), _arr(scalars)
        n = scalars.size // 4
        assert bases.size // self.aff == n
        out = np.zeros(self.jac, np.uint64)
        self._f("dense_multiexp")(_p64(bases), _p64(scalars), C.c_size_t(n), C.c_int(cpus), _p64(out))
        return out

    def naive_multiexp(self, bases, scalars):
        bases, scalars = _arr(bases), _arr(scalars)
        out = np.zeros(self.jac, np.uint64)
        self._f("naive_multiexp")(_p64(bases), _p64(scalars), C.c_size_t(scalars.size // 4), _p64(out))
        return out


G1 = Group(1)
G2 = Group(2)


def point_domain_op(group: int, affine_pts, log_n: int, op: str, log_cpus: int = 31):
    """EvaluationDomain<Point<G>>::{fft, ifft} then batch_normalization (prepare_phase2.rs:68-131):
    (2^log_n, 8g) affine raw records in -> affine raw records out."""
    G = G1 if group == 1 else G2
    affine_pts = _arr(affine_pts).reshape(-1, G.aff)
    jac = np.ascontiguousarray(np.stack([G.from_affine(p) for p in affine_pts]))
    fn = lib().oracle_g1_point_domain_op if group == 1 else lib().oracle_g2_point_domain_op
    rc = fn(_p64(jac), C.c_uint32(log_n), C.c_int(OPS[op]), C.c_uint32(log_cpus))
    if rc:
        raise ValueError(f"oracle point_domain_op rc={rc}")
    return np.stack([G.to_affine(p) for p in jac.reshape(-1, G.jac)])


def multiexp_set_window_bits(c: int):
    """BENCH ONLY: force the window width of the BN254 multiexps (0: the reference's rule) -- the 2^26 headline's c = 19 on a 2^22 sample"""
    lib().oracle_multiexp_set_window_bits(C.c_uint32(c))


def multiexp_window_bits(n):
    return int(lib().oracle_multiexp_window_bits(C.c_size_t(n)))


def fr_serial_fft(a, log_n, omega):
    a, omega = _arr(a).copy(), _arr(omega)
    lib().oracle_fr_serial_fft(_p64(a), C.c_uint32(log_n), _p64(omega))
    return a


def fr_parallel_fft(a, log_n, omega, log_cpus):
    a, omega = _arr(a).copy(), _arr(omega)
    lib().oracle_fr_parallel_fft(_p64(a), C.c_uint32(log_n), _p64(omega), C.c_uint32(log_cpus))
    return a


def fr_domain(log_n):
    """(omega, omegainv, geninv, minv) Montgomery limbs, or None if log_n > S (PolynomialDegreeTooLarge)."""
    outs = [np.zeros(4, np.uint64) for _ in range(4)]
    rc = lib().oracle_fr_domain(C.c_uint32(log_n), *[_p64(o) for o in outs])
    return None if rc else tuple(outs)


OPS = {"fft": 0, "ifft": 1, "coset_fft": 2, "icoset_fft": 3}


def fr_domain_op(a, log_n, op, log_cpus=31):
    """In-place semantic of EvaluationDomain::{fft,ifft,coset_fft,icoset_fft}; returns the new array.
    log_cpus >= log_n selects serial_fft (the normative definition)."""
    a = _arr(a).copy()
    rc = lib().oracle_fr_domain_op(_p64(a), C.c_uint32(log_n), C.c_int(OPS[op]), C.c_uint32(log_cpus))
    if rc:
        raise ValueError(f"oracle_fr_domain_op rc={rc}")
    return a


def dummy_multiexp(bases, scalars, density=None, density_bits=None, base_offset=0):
    bases = np.ascontiguousarray(bases, dtype=np.uint32)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    if density is not None:
        density = np.ascontiguousarray(density, dtype=np.uint32)
        if density_bits is None:
            density_bits = scalars.size
    out = np.zeros(1, np.uint32)
    rc = lib().oracle_dummy_multiexp(_p32(bases), C.c_size_t(bases.size), C.c_size_t(base_offset), _p64(scalars), C.c_size_t(scalars.size),
                                     _p32(density), C.c_size_t(density_bits or 0), _p32(out))
    return rc, int(out[0])


def dummy_domain_op(a, log_n, op, log_cpus=31):
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    rc = lib().oracle_dummy_domain_op(_p32(a), C.c_uint32(log_n), C.c_int(OPS[op]), C.c_uint32(log_cpus))
    if rc:
        raise ValueError(f"oracle_dummy_domain_op rc={rc}")
    return a


def dummy_domain_omega(log_n):
    return int(lib().oracle_dummy_domain_omega(C.c_uint32(log_n)))


# ---- point codecs (oracle/codec.h)
ENC_SIZE = {(1, False): 64, (1, True): 32, (2, False): 128, (2, True): 64}


def encode_points(group: int, affine, compressed: bool) -> np.ndarray:
    """(n, 8g) raw affine records -> (n, size) uint8 wire records (EncodedPoint::from_affine)."""
    G = G1 if group == 1 else G2
    affine = _arr(affine).reshape(-1, G.aff)
    out = np.zeros((affine.shape[0], ENC_SIZE[(group, bool(compressed))]), np.uint8)
    fn = lib().oracle_g1_encode if group == 1 else lib().oracle_g2_encode
    fn(out.ctypes.data_as(C.c_void_p), _p64(affine), C.c_size_t(affine.shape[0]), C.c_int(1 if compressed else 0))
    return out


def decode_points(group: int, data, compressed: bool, checked: bool = True):
    """wire records -> (rc, err_index, (n, 8g) raw affine records); rc = 0 or the GroupDecodingError code of the first failure."""
    G = G1 if group == 1 else G2
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, ENC_SIZE[(group, bool(compressed))])
    out = np.zeros((data.shape[0], G.aff), np.uint64)
    err = C.c_longlong(-1)
    fn = lib().oracle_g1_decode if group == 1 else lib().oracle_g2_decode
    fn.restype = C.c_int
    rc = fn(_p64(out), data.ctypes.data_as(C.c_void_p), C.c_size_t(data.shape[0]), C.c_int(1 if compressed else 0), C.c_int(1 if checked else 0),
            C.byref(err))
    return rc, err.value, out


def fq_sqrt(a):
    a = _arr(a)
    r = np.zeros(4, np.uint64)
    return r if lib().oracle_fq_sqrt(_p64(r), _p64(a)) else None


def fq2_sqrt(a):
    a = _arr(a)
    r = np.zeros(8, np.uint64)
    return r if lib().oracle_fq2_sqrt(_p64(r), _p64(a)) else None


def g2_coeff_b():
    r = np.zeros(8, np.uint64)
    lib().oracle_g2_coeff_b(_p64(r))
    return r
