"""The caller of the hot path, restated over the device library: `create_proof` of bellman's Groth16 prover
(bellman/src/groth16/prover.rs:202-343) -- the one place where the Fr NTTs, the elementwise domain steps, the scalar
conversions and the eight multiexps (G1 and G2, FullDensity and DensityTracker sources) of this backend meet.

Mirrored interfaces (same names, argument meaning, order of operations):
  groth16/prover.rs:202-343   ProvingAssignment::create_proof(params, r, s)
  groth16/mod.rs:429-483      ParameterSource for &Parameters: get_vk / get_h / get_l / get_a / get_b_g1 / get_b_g2
  groth16/prover.rs:89-129    scalars_into_representations / field_elements_into_representations: FUSED into the multiexps
                              (MI355ZK_MSM_SCALARS_MONTGOMERY), the vectors never leave HBM and are never rewritten

Data: everything large is device resident (torch CUDA int64 tensors holding u64 limbs): the evaluation vectors a, b, c and the
assignments are MONTGOMERY Fr (the prover's `Vec<Scalar<E>>` / `Vec<E::Fr>`), the parameter vectors raw affine records.  The
H polynomial goes ifft -> coset_fft -> mul / sub / divide_by_z_on_coset -> icoset_fft -> multiexp without a host round trip.
What stays on the host is what the reference does once per proof on single points (vk.delta_g1.mul(r) ...): here through the
library's host-side add / mul / to_affine helpers.
"""
from __future__ import annotations

import ctypes as C
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import ceremony
from . import lib as _lib
from .bellman import DensityTracker, EvaluationDomain, FullDensity, SynthesisError, Worker, multiexp

_R_ORDER = ceremony._R_ORDER


class Parameters:
    """groth16::Parameters (groth16/mod.rs:216-238) as a ParameterSource (:429-483).  vk: dict of raw affine host records
    (alpha_g1, beta_g1, delta_g1: 8 u64; beta_g2, delta_g2: 16 u64); h, l, a, b_g1: (n, 8) device tensors; b_g2: (n, 16)."""

    def __init__(self, vk, h, l, a, b_g1, b_g2):  # noqa: E741
        self.vk, self.h, self.l, self.a, self.b_g1, self.b_g2 = vk, h, l, a, b_g1, b_g2

    def with_tables(self, g1_min: int = 1 << 19, g2_min: int = 1 << 16) -> "Parameters":
        """The same parameters with WINDOW TABLES (bellman.MsmTable: table mode, include/mi355zk.h) in place of the vectors long
        enough to gain from them -- the vectors of a circuit's Parameters are the same for every proof, which is what a table needs.
        Costs n_windows (13 - 15) times the vector's memory; the proofs are byte-identical."""
        from .bellman import MsmTable

        def tab(v, lo):
            return MsmTable(v) if not isinstance(v, MsmTable) and int(v.shape[0]) >= lo else v

        return Parameters(self.vk, tab(self.h, g1_min), tab(self.l, g1_min), tab(self.a, g1_min), tab(self.b_g1, g1_min), tab(self.b_g2, g2_min))

    def get_vk(self, _num_inputs):
        return self.vk

    def get_h(self, _n):
        return (self.h, 0)

    def get_l(self, _n):
        return (self.l, 0)

    def get_a(self, num_inputs, _num_aux):
        return (self.a, 0), (self.a, num_inputs)

    def get_b_g1(self, num_inputs, _num_aux):
        return (self.b_g1, 0), (self.b_g1, num_inputs)

    def get_b_g2(self, num_inputs, _num_aux):
        return (self.b_g2, 0), (self.b_g2, num_inputs)


class ProvingAssignment:
    """groth16/prover.rs:131-151: the densities and the evaluation / assignment vectors synthesis leaves behind."""

    def __init__(self, a, b, c, input_assignment, aux_assignment, a_aux_density: DensityTracker, b_input_density: DensityTracker,
                 b_aux_density: DensityTracker):
        self.a, self.b, self.c = a, b, c
        self.input_assignment, self.aux_assignment = input_assignment, aux_assignment
        self.a_aux_density, self.b_input_density, self.b_aux_density = a_aux_density, b_input_density, b_aux_density


# ---- single-point group operations of the proof assembly (prover.rs:300-333)
def _limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def _is_zero(aff: np.ndarray) -> bool:
    return not np.asarray(aff).any()


def _from_affine(aff: np.ndarray) -> np.ndarray:
    """raw affine record -> Jacobian X || Y || Z with Z = one (the all-zero record -> Z = 0)"""
    aff = np.ascontiguousarray(aff, dtype=np.uint64)
    g = aff.size // 8
    jac = np.zeros(12 * g, dtype=np.uint64)
    if not _is_zero(aff):
        jac[:8 * g] = aff
        jac[8 * g:8 * g + 4] = ceremony.G1_ONE_RAW[:4]  # Fq one in Montgomery form = G1_GENERATOR_X (fq.rs:39-44); Fq2 one = (one, 0)
    return jac


def _to_affine(jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    g = jac.size // 12
    out = np.zeros(8 * g, dtype=np.uint64)
    fn = _lib.load().mi355zk_bn254_g1_to_affine if g == 1 else _lib.load().mi355zk_bn254_g2_to_affine
    assert fn(out.ctypes.data_as(C.c_void_p), jac.ctypes.data_as(C.c_void_p)) == 0
    return out


def _add(acc: np.ndarray, other: np.ndarray) -> np.ndarray:
    acc = np.ascontiguousarray(acc, dtype=np.uint64).copy()
    other = np.ascontiguousarray(other, dtype=np.uint64)
    fn = _lib.load().mi355zk_bn254_g1_add if acc.size == 12 else _lib.load().mi355zk_bn254_g2_add
    assert fn(acc.ctypes.data_as(C.c_void_p), other.ctypes.data_as(C.c_void_p)) == 0
    return acc


def _mul(point, k: int, device=None) -> np.ndarray:
    """k * point for one point (affine record or Jacobian), Jacobian out: CurveAffine::mul / CurveProjective::mul_assign, done
    where the reference does it -- on the host, once per proof (the library's mi355zk_bn254_g{1,2}_mul)"""
    point = np.ascontiguousarray(point, dtype=np.uint64)
    jac = (_from_affine(point) if point.size in (8, 16) else point).copy()
    fn = _lib.load().mi355zk_bn254_g1_mul if jac.size == 12 else _lib.load().mi355zk_bn254_g2_mul
    kk = _limbs(k % _R_ORDER)
    assert fn(jac.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p)) == 0
    return jac


# The eight multiexps of a proof are queued from eight host threads (the reference queues them on its CpuPool before the first
# wait(), prover.rs:250-298).  The threads and their streams outlive the proof: starting threads and creating streams costs as much
# as a small multiexp.
_executor = None
_executor_lock = threading.Lock()
_tls = threading.local()


def _pool8() -> ThreadPoolExecutor:
    global _executor
    with _executor_lock:
        if _executor is None:
            _executor = ThreadPoolExecutor(max_workers=8, thread_name_prefix="mi355zk-multiexp")
    return _executor


def _thread_stream(device):
    import torch

    streams = getattr(_tls, "streams", None)
    if streams is None:
        streams = _tls.streams = {}
    key = (device.type, device.index)
    if key not in streams:
        streams[key] = torch.cuda.Stream(device=device)
    return streams[key]


def create_proof(pool: Worker, params: Parameters, prover: ProvingAssignment, r: int, s: int, concurrent: bool = True):
    """prover.rs:202-343.  r, s: canonical integers mod the group order.  Returns the proof (a, b, c) as raw affine records
    (8, 16, 8 u64).  concurrent: the eight multiexps are submitted from eight host threads before the first wait(), which is
    how the reference queues them on its CpuPool (prover.rs:250-298); the library serialises them on the device."""
    import torch

    vk = params.get_vk(prover.input_assignment.shape[0])
    device = prover.a.device
    ex = _pool8() if concurrent else None

    def submit(bases, density, exponents):
        if ex is None:
            with torch.cuda.device(device):
                fut = multiexp(pool, bases, density, exponents, scalars_montgomery=True)
            return fut.wait

        def call():
            # its own stream: small multiexps (the library leases them a workspace of their own) overlap with the long ones instead of
            # queueing behind them on one stream; inputs are complete (the caller synchronised), results come back on the host
            with torch.cuda.device(device), torch.cuda.stream(_thread_stream(device)):
                return multiexp(pool, bases, density, exponents, scalars_montgomery=True)

        f = ex.submit(call)
        return lambda: f.result().wait()

    # ---- h (prover.rs:217-247)
    a = EvaluationDomain.from_coeffs(prover.a)
    b = EvaluationDomain.from_coeffs(prover.b)
    c = EvaluationDomain.from_coeffs(prover.c)
    # (prover.rs:220-231 transforms a, b and c one after the other; the three are independent, and one launch per pass over all three lets
    # one transform's loads and stores run under another's butterflies -- same bytes)
    EvaluationDomain.ifft_many(pool, (a, b, c))
    EvaluationDomain.coset_fft_many(pool, (a, b, c))
    a.mul_assign(pool, b)
    del b
    a.sub_assign(pool, c)
    del c
    a.divide_by_z_on_coset(pool)
    a.icoset_fft(pool)
    coeffs = a.into_coeffs()
    coeffs = coeffs[:coeffs.shape[0] - 1]                       # a.truncate(a_len)
    # the multiexp threads use their own streams: the H pipeline (queued on the stream of the TENSORS' device, which need not be
    # the caller's current device) must be complete before they read `coeffs`
    torch.cuda.current_stream(device).synchronize()
    h = submit(params.get_h(coeffs.shape[0]), FullDensity(), coeffs)

    # ---- the assignments (prover.rs:256-298); into_repr is fused into every multiexp
    input_assignment, aux_assignment = prover.input_assignment, prover.aux_assignment
    l = submit(params.get_l(aux_assignment.shape[0]), FullDensity(), aux_assignment)  # noqa: E741
    a_aux_density_total = prover.a_aux_density.get_total_density()
    a_inputs_source, a_aux_source = params.get_a(input_assignment.shape[0], a_aux_density_total)
    a_inputs = submit(a_inputs_source, FullDensity(), input_assignment)
    a_aux = submit(a_aux_source, prover.a_aux_density, aux_assignment)
    b_input_density_total = prover.b_input_density.get_total_density()
    b_aux_density_total = prover.b_aux_density.get_total_density()
    b_g1_inputs_source, b_g1_aux_source = params.get_b_g1(b_input_density_total, b_aux_density_total)
    b_g1_inputs = submit(b_g1_inputs_source, prover.b_input_density, input_assignment)
    b_g1_aux = submit(b_g1_aux_source, prover.b_aux_density, aux_assignment)
    b_g2_inputs_source, b_g2_aux_source = params.get_b_g2(b_input_density_total, b_aux_density_total)
    b_g2_inputs = submit(b_g2_inputs_source, prover.b_input_density, input_assignment)
    b_g2_aux = submit(b_g2_aux_source, prover.b_aux_density, aux_assignment)

    if _is_zero(vk["delta_g1"]) or _is_zero(vk["delta_g2"]):      # prover.rs:300-304: subversion-CRS attack
        raise SynthesisError(SynthesisError.UNEXPECTED_IDENTITY)
    g_a = _add(_mul(vk["delta_g1"], r, device), _from_affine(vk["alpha_g1"]))
    g_b = _add(_mul(vk["delta_g2"], s, device), _from_affine(vk["beta_g2"]))
    g_c = _mul(vk["delta_g1"], r * s % _R_ORDER, device)
    g_c = _add(g_c, _mul(vk["alpha_g1"], s, device))
    g_c = _add(g_c, _mul(vk["beta_g1"], r, device))
    a_answer = _add(a_inputs(), a_aux())
    g_a = _add(g_a, a_answer)
    a_answer = _mul(a_answer, s, device)
    g_c = _add(g_c, a_answer)
    b1_answer = _add(b_g1_inputs(), b_g1_aux())
    b2_answer = _add(b_g2_inputs(), b_g2_aux())
    g_b = _add(g_b, b2_answer)
    b1_answer = _mul(b1_answer, r, device)
    g_c = _add(g_c, b1_answer)
    g_c = _add(g_c, h())
    g_c = _add(g_c, l())
    return _to_affine(g_a), _to_affine(g_b), _to_affine(g_c)
