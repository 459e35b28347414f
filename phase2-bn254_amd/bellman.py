"""Host-side mirror of the reference's interface for the accelerated path (thin; all compute is in
libmi355zk.so through the C ABI).  The C++ twin of this file is host/bellman.hpp.

Reference interface mirrored (same names, argument meaning, error behaviour):
  bellman/src/multiexp.rs:330-340   multiexp(pool, bases, density_map, exponents) -> Future<Projective>
  bellman/src/source.rs:36-140      (Arc<Vec<G>>, usize) source builder, FullDensity, DensityTracker
  bellman/src/domain.rs:30-203      EvaluationDomain::{from_coeffs, fft, ifft, coset_fft, icoset_fft, ...}
  bellman/src/multicore.rs:17-72    Worker
  bellman/src/cs.rs:156-173         SynthesisError variants raised on this path

Data may live on the host (numpy uint64 arrays -> host-buffer entry points) or already in HBM
(torch CUDA tensors of dtype int64/uint64 -> `_dev` entry points on torch's current stream).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

FR_S = 28  # Fr::S, pairing/src/bn256/fr.rs:31-34


class SynthesisError(Exception):
    """bellman/src/cs.rs:156-173 (the variants this path can raise)."""

    UNEXPECTED_IDENTITY = "UnexpectedIdentity"          # source.rs:50-52
    IO_UNEXPECTED_EOF = "IoError(UnexpectedEof)"        # source.rs:46-48,62-64
    POLYNOMIAL_DEGREE_TOO_LARGE = "PolynomialDegreeTooLarge"  # domain.rs:66-79
    UNCONSTRAINED_VARIABLE = "UnconstrainedVariable"    # phase2/src/parameters.rs:340-346 (MPCParameters::new)

    def __init__(self, kind: str, index: int = -1):
        super().__init__(kind if index < 0 else f"{kind} at exponent {index}")
        self.kind = kind
        self.index = index


class DeviceError(RuntimeError):
    """HIP failure inside libmi355zk (no reference counterpart; there is no CPU fallback)."""


class Worker:
    """bellman/src/multicore.rs:17-35.  The reference's Worker is a CPU thread pool; here it names the GPU(s) the calling process
    drives.  Worker(3): one GPU (one process per GPU, shard.py).  Worker(devices=[0, 1, .., 7]): the single-process multi-GPU mode
    of mi355zk_init -- host-buffer multiexps of >= 2^20 exponents are cut into one point range per device and joined on the host,
    shorter ones go to the devices in turn (include/mi355zk.h).  The last Worker constructed defines the library's device set."""

    def __init__(self, device: int | None = None, devices=None):
        if devices is not None:
            devices = [int(d) for d in devices]
            assert devices and device is None
            ids = (C.c_int * len(devices))(*devices)
            self.device, self.devices, n = devices[0], devices, len(devices)
        else:
            ids = (C.c_int * 1)(device) if device is not None else None
            self.device, self.devices, n = device, [device] if device is not None else [], 1 if device is not None else 0
        rc = _lib.load().mi355zk_init(ids, n)
        if rc != 0:
            raise DeviceError(f"mi355zk_init failed (rc={rc})")

    def log_num_cpus(self) -> int:  # multicore.rs:37-39; serial/parallel FFT split is moot on the GPU
        return 0


class FullDensity:
    """source.rs:80-99"""

    def get_query_size(self):
        return None

    def words(self):
        return None, 0


class DensityTracker:
    """source.rs:101-140 (BitVec-backed in the reference; here a numpy uint32 word array with the
    ABI's bit order: bit i = word i/32, bit i%32)."""

    def __init__(self):
        self._bits = np.zeros(64, dtype=np.uint8)   # one byte per element, grown geometrically (add_element is called per variable)
        self._n = 0
        self.total_density = 0
        self._packed = None                         # words() of the current contents (a proof asks for the same map 2-3 times)

    def add_element(self):
        if self._n == self._bits.size:
            self._bits = np.concatenate([self._bits, np.zeros(self._bits.size, dtype=np.uint8)])
        self._bits[self._n] = 0
        self._n += 1
        self._packed = None

    def inc(self, idx: int):
        if idx >= self._n:
            raise IndexError(idx)
        if not self._bits[idx]:
            self._bits[idx] = 1
            self.total_density += 1
            self._packed = None

    def get_total_density(self) -> int:
        return self.total_density

    def get_query_size(self):
        return self._n

    @classmethod
    def from_bools(cls, bools) -> "DensityTracker":
        d = cls()
        d._bits = np.ascontiguousarray(np.asarray(bools, dtype=bool), dtype=np.uint8).reshape(-1)
        d._n = d._bits.size
        if d._n == 0:
            d._bits = np.zeros(64, dtype=np.uint8)
        d.total_density = int(d._bits[:d._n].sum())
        return d

    def words(self):
        if self._packed is None:
            n = self._n
            w = np.zeros((n + 31) // 32 or 1, dtype=np.uint32)
            if n:
                pad = np.zeros(w.size * 32, dtype=np.uint8)
                pad[:n] = self._bits[:n]
                w[:] = np.packbits(pad.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
            self._packed = (w, n)
        return self._packed


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Ready:
    """A completed future (futures 0.1 `future::result`): multiexp() is synchronous here, which is
    what singlecore::Worker::compute already is in the reference (singlecore.rs:33-47)."""

    def __init__(self, value=None, error: Exception | None = None):
        self._value, self._error = value, error

    def wait(self):
        if self._error is not None:
            raise self._error
        return self._value


def pin_bases(arr, tables: bool = False) -> None:
    """mi355zk_bases_cache_pin: declare the HOST base vector `arr` ((n, 8) / (n, 16) u64, C-contiguous) immutable until
    unpin_bases(arr) -- the `Arc<Vec<G>>` of a Parameters object (groth16/mod.rs:216-238).  Host-buffer multiexps over it then
    keep their uploaded copy on the device.  Without the promise every call uploads its bases again."""
    assert not _is_torch(arr) and arr.flags["C_CONTIGUOUS"] and arr.dtype == np.uint64
    fn = _lib.load().mi355zk_bases_cache_pin_tables if tables else _lib.load().mi355zk_bases_cache_pin  # tables: + the window table (table mode)
    rc = fn(arr.ctypes.data_as(C.c_void_p), arr.shape[0], {8: 1, 16: 2}[arr.shape[1]])
    if rc != 0:
        raise ValueError("mi355zk_bases_cache_pin: bad arguments")


def unpin_bases(arr=None) -> None:
    """mi355zk_bases_cache_invalidate: the promise ends (before rewriting or freeing the vector); None: every vector."""
    _lib.load().mi355zk_bases_cache_invalidate(arr.ctypes.data_as(C.c_void_p) if arr is not None else None)


class MsmTable:
    """The window table of a device-resident base vector (mi355zk_bn254_g{1,2}_msm_table_build_dev): pass `(table, offset)` to
    multiexp() where `(bases, offset)` would go -- same result and errors, evaluated in table mode (one bucket set for all windows).
    For vectors that stay put between calls: the `Arc<Vec<G>>` of groth16::Parameters (groth16/mod.rs:216-238)."""

    def __init__(self, bases):
        import torch

        assert _is_torch(bases) and bases.is_cuda and bases.is_contiguous() and bases.shape[1] in (8, 16)
        lib = _lib.load()
        self.n_bases, self.limbs = int(bases.shape[0]), int(bases.shape[1])
        self.group = {8: 1, 16: 2}[self.limbs]
        c, w = C.c_uint32(), C.c_uint32()
        if lib.mi355zk_msm_table_geometry(self.n_bases, self.group, C.byref(c), C.byref(w)) != 0:
            raise ValueError("mi355zk_msm_table_geometry: bad arguments")
        self.window_bits, self.n_windows = int(c.value), int(w.value)
        self.table = torch.empty((self.n_windows * self.n_bases, self.limbs), dtype=torch.int64, device=bases.device)
        fn = lib.mi355zk_bn254_g1_msm_table_build_dev if self.group == 1 else lib.mi355zk_bn254_g2_msm_table_build_dev
        with torch.cuda.device(bases.device):
            rc = fn(C.c_void_p(bases.data_ptr()), self.n_bases, C.c_void_p(self.table.data_ptr()), self.table.numel() * 8, _stream_ptr())
        if rc != 0:
            raise DeviceError(f"mi355zk msm_table_build rc={rc}")


def multiexp(pool: Worker, bases, density_map, exponents, window_group=None, scalars_montgomery: bool = False) -> _Ready:
    """bellman/src/multiexp.rs:330.  (window_group = (groups, index), device-resident data only: the partial sum over one
    of `groups` equal groups of scalar windows -- multi-GPU sharding by windows, shard.py; None = the whole multiexp.
    scalars_montgomery, device-resident data only: `exponents` are Montgomery-form Fr elements, i.e. the prover's vectors BEFORE
    scalars_into_representations / field_elements_into_representations (prover.rs:89-129); the conversion is fused into the call.)  `bases` = (array, offset) like `(Arc<Vec<G>>, usize)`:
    array of shape (n_bases, 8) u64 for G1Affine raw records or (n_bases, 16) for G2Affine;
    `exponents` = (n, 4) u64 canonical FrRepr; `density_map` = FullDensity() or a DensityTracker.
    Returns a ready future whose wait() yields the Jacobian X||Y||Z limbs (12 / 24 u64)."""
    arr, offset = bases
    qs = density_map.get_query_size()
    n_exp = int(exponents.shape[0])
    if qs is not None:
        assert qs == n_exp  # multiexp.rs:347-352
    words, dbits = density_map.words()
    lib = _lib.load()
    if isinstance(arr, MsmTable):
        import torch

        if window_group is not None and tuple(window_group) != (1, 0):
            raise ValueError("table mode evaluates all windows in one bucket set: no window groups")
        assert exponents.is_cuda and exponents.is_contiguous()
        out = np.zeros(12 * arr.group, dtype=np.uint64)
        fn = lib.mi355zk_bn254_g1_msm_table_dev if arr.group == 1 else lib.mi355zk_bn254_g2_msm_table_dev
        with torch.cuda.device(arr.table.device):
            rc = fn(C.c_void_p(arr.table.data_ptr()), arr.n_bases, offset, C.c_void_p(exponents.data_ptr()), n_exp,
                    words.ctypes.data_as(C.c_void_p) if words is not None else None, dbits,
                    _lib.MSM_SCALARS_MONTGOMERY if scalars_montgomery else 0, _stream_ptr(), out.ctypes.data_as(C.c_void_p))
    elif _is_torch(arr):
        import torch

        assert arr.is_cuda and exponents.is_cuda and arr.is_contiguous() and exponents.is_contiguous()
        limbs = arr.shape[1]
        group = {8: 1, 16: 2}[limbs]
        out = np.zeros(12 * group, dtype=np.uint64)
        wg, wi = window_group if window_group is not None else (1, 0)
        fn = lib.mi355zk_bn254_g1_msm_ex_dev if group == 1 else lib.mi355zk_bn254_g2_msm_ex_dev
        with torch.cuda.device(arr.device):
            rc = fn(C.c_void_p(arr.data_ptr()), arr.shape[0], offset, C.c_void_p(exponents.data_ptr()), n_exp,
                    words.ctypes.data_as(C.c_void_p) if words is not None else None, dbits,
                    _lib.MSM_SCALARS_MONTGOMERY if scalars_montgomery else 0, int(wg), int(wi), _stream_ptr(), out.ctypes.data_as(C.c_void_p))
    else:
        if (window_group is not None and tuple(window_group) != (1, 0)) or scalars_montgomery:
            raise ValueError("window groups / Montgomery scalars need device-resident inputs")
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        exponents = np.ascontiguousarray(exponents, dtype=np.uint64)
        limbs = arr.shape[1] if arr.ndim == 2 else 8
        group = {8: 1, 16: 2}[limbs]
        out = np.zeros(12 * group, dtype=np.uint64)
        fn = lib.mi355zk_bn254_g1_msm if group == 1 else lib.mi355zk_bn254_g2_msm
        rc = fn(arr.ctypes.data_as(C.c_void_p), arr.shape[0], offset, exponents.ctypes.data_as(C.c_void_p), n_exp,
                words.ctypes.data_as(C.c_void_p) if words is not None else None, dbits, out.ctypes.data_as(C.c_void_p))
    if rc == _lib.OK:
        return _Ready(out)
    idx = int(lib.mi355zk_last_error_index())
    if rc == _lib.ERR_UNEXPECTED_IDENTITY:
        return _Ready(error=SynthesisError(SynthesisError.UNEXPECTED_IDENTITY, idx))
    if rc == _lib.ERR_UNEXPECTED_EOF:
        return _Ready(error=SynthesisError(SynthesisError.IO_UNEXPECTED_EOF, idx))
    if rc == _lib.ERR_BAD_ARGS:
        return _Ready(error=ValueError("mi355zk: bad arguments"))
    return _Ready(error=DeviceError(f"mi355zk device failure rc={rc}"))


class EvaluationDomain:
    """bellman/src/domain.rs:30-203 for G = Scalar<Bn256>: coefficients are (m, 4) u64 Montgomery Fr
    limbs, either a numpy array or a torch CUDA tensor (int64 view)."""

    def __init__(self, coeffs, exp: int):
        self.coeffs = coeffs
        self.exp = exp

    @classmethod
    def from_coeffs(cls, coeffs) -> "EvaluationDomain":
        n = int(coeffs.shape[0])
        if n > (1 << FR_S) - 1:  # domain.rs:66-68
            raise SynthesisError(SynthesisError.POLYNOMIAL_DEGREE_TOO_LARGE)
        m, exp = 1, 0
        while m < n:  # domain.rs:70-79
            m *= 2
            exp += 1
            if exp > FR_S:
                raise SynthesisError(SynthesisError.POLYNOMIAL_DEGREE_TOO_LARGE)
        if m != n:  # coeffs.resize(m, G::group_zero()), domain.rs:89
            if _is_torch(coeffs):
                import torch

                pad = torch.zeros((m - n, 4), dtype=coeffs.dtype, device=coeffs.device)
                coeffs = torch.cat([coeffs, pad]).contiguous()
            else:
                coeffs = np.concatenate([np.asarray(coeffs, dtype=np.uint64), np.zeros((m - n, 4), dtype=np.uint64)])
        elif not _is_torch(coeffs):
            coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).copy()
        else:
            # `from_coeffs` takes the Vec by value (domain.rs:52): the domain owns its coefficients and the transforms work in
            # place, so the caller's tensor is copied -- a ProvingAssignment can then be proved twice
            coeffs = coeffs.contiguous().clone()
        return cls(coeffs, exp)

    def as_ref(self):
        return self.coeffs

    def into_coeffs(self):
        return self.coeffs

    def _op(self, op: int):
        lib = _lib.load()
        if _is_torch(self.coeffs):
            import torch

            assert self.coeffs.is_cuda and self.coeffs.is_contiguous()
            with torch.cuda.device(self.coeffs.device):
                rc = lib.mi355zk_bn254_fr_domain_op_dev(C.c_void_p(self.coeffs.data_ptr()), self.exp, op, _stream_ptr())
        else:
            rc = lib.mi355zk_bn254_fr_domain_op(self.coeffs.ctypes.data_as(C.c_void_p), self.exp, op)
        if rc != 0:
            raise DeviceError(f"mi355zk NTT failed rc={rc}")

    @staticmethod
    def _op_many(domains, op: int):
        """The same operation on several device-resident domains of one size (prover.rs:217-241: a, b, c): one launch per pass over all of
        them (mi355zk_bn254_fr_domain_op_batch_dev) -- the results are those of the separate calls, byte for byte."""
        domains = list(domains)
        if not domains:
            return
        d0 = domains[0]
        if len(domains) == 1 or not all(_is_torch(d.coeffs) and d.coeffs.is_cuda and d.exp == d0.exp and d.coeffs.device == d0.coeffs.device for d in domains):
            for d in domains:
                d._op(op)
            return
        import torch

        ptrs = (C.c_void_p * len(domains))(*[d.coeffs.data_ptr() for d in domains])
        assert all(d.coeffs.is_contiguous() for d in domains)
        with torch.cuda.device(d0.coeffs.device):
            rc = _lib.load().mi355zk_bn254_fr_domain_op_batch_dev(ptrs, len(domains), d0.exp, op, _stream_ptr())
        if rc != 0:
            raise DeviceError(f"mi355zk batched NTT failed rc={rc}")

    @staticmethod
    def fft_many(worker: Worker, domains):
        EvaluationDomain._op_many(domains, _lib.OP_FFT)

    @staticmethod
    def ifft_many(worker: Worker, domains):
        EvaluationDomain._op_many(domains, _lib.OP_IFFT)

    @staticmethod
    def coset_fft_many(worker: Worker, domains):
        EvaluationDomain._op_many(domains, _lib.OP_COSET_FFT)

    @staticmethod
    def icoset_fft_many(worker: Worker, domains):
        EvaluationDomain._op_many(domains, _lib.OP_ICOSET_FFT)

    def fft(self, worker: Worker):  # domain.rs:154
        self._op(_lib.OP_FFT)

    def ifft(self, worker: Worker):  # domain.rs:159
        self._op(_lib.OP_IFFT)

    def coset_fft(self, worker: Worker):  # domain.rs:191
        self._op(_lib.OP_COSET_FFT)

    def icoset_fft(self, worker: Worker):  # domain.rs:197
        self._op(_lib.OP_ICOSET_FFT)

    # ---- the elementwise steps of the prover's H pipeline (prover.rs:217-241), device-resident coefficients only
    def _dev_coeffs(self):
        if not _is_torch(self.coeffs):
            raise ValueError("device-resident coefficients (a torch CUDA tensor) required")
        return C.c_void_p(self.coeffs.data_ptr())

    def _on_device(self):
        """context manager: the coefficients' GPU is current (kernels and `_stream_ptr()` then refer to ITS stream, not to the
        stream of whatever device the caller last selected)"""
        import torch

        self._dev_coeffs()
        return torch.cuda.device(self.coeffs.device)

    def z(self, tau):
        """domain.rs:207-212: tau^m - 1 (tau, result: 4 u64 Montgomery limbs)."""
        tau = np.ascontiguousarray(tau, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().mi355zk_bn254_fr_domain_z(self.exp, tau.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise DeviceError(f"mi355zk domain_z failed rc={rc}")
        return out

    def divide_by_z_on_coset(self, worker: Worker):  # domain.rs:217-234
        with self._on_device():
            rc = _lib.load().mi355zk_bn254_fr_divide_by_z_on_coset_dev(self._dev_coeffs(), self.exp, _stream_ptr())
        if rc != 0:
            raise DeviceError(f"mi355zk divide_by_z_on_coset failed rc={rc}")

    def mul_assign(self, worker: Worker, other: "EvaluationDomain"):  # domain.rs:236-249
        assert self.coeffs.shape[0] == other.coeffs.shape[0]
        assert other.coeffs.device == self.coeffs.device
        with self._on_device():
            rc = _lib.load().mi355zk_bn254_fr_mul_assign_dev(self._dev_coeffs(), other._dev_coeffs(), self.coeffs.shape[0], _stream_ptr())
        if rc != 0:
            raise DeviceError(f"mi355zk mul_assign failed rc={rc}")

    def sub_assign(self, worker: Worker, other: "EvaluationDomain"):  # domain.rs:251-260
        assert self.coeffs.shape[0] == other.coeffs.shape[0]
        assert other.coeffs.device == self.coeffs.device
        with self._on_device():
            rc = _lib.load().mi355zk_bn254_fr_sub_assign_dev(self._dev_coeffs(), other._dev_coeffs(), self.coeffs.shape[0], _stream_ptr())
        if rc != 0:
            raise DeviceError(f"mi355zk sub_assign failed rc={rc}")


def best_fft(a, worker: Worker, omega, log_n: int):
    """bellman/src/domain.rs:263: in-place transform of `a` (numpy (2^log_n, 4) u64) with root `omega`."""
    rc = _lib.load().mi355zk_bn254_fr_ntt(a.ctypes.data_as(C.c_void_p), log_n,
                                          np.ascontiguousarray(omega, dtype=np.uint64).ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise DeviceError(f"mi355zk NTT failed rc={rc}")
