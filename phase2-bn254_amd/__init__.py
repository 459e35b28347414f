"""mi355zk: MI355X (gfx950) backend for the BN254 MSM / Fr-NTT hot path of kobigurk/phase2-bn254.

Layout:
  csrc/           hand-written HIP kernels + the C ABI (include/mi355zk.h) -> libmi355zk.so
  lib.py          ctypes loader (fails loudly when the library is missing)
  shard.py        multi-GPU point-range sharding + the all-gather/join exchange step
  prover.py       the caller of the path: groth16 create_proof (prover.rs:202-343) over the device library
  ceremony.py     the ceremony-side callers (batch_exp, merge_pairs, QAP evaluation, point FFT, codecs, file containers)
  bellman.py      host-side mirror of the reference's interface for this path:
                  multiexp(), FullDensity, DensityTracker, EvaluationDomain, SynthesisError

The directory name carries a hyphen (it is the reference's name); import it through the
repo-root shim module `phase2_bn254_amd`.
"""
from . import bellman, ceremony, circom, lib, prover, shard  # noqa: F401
from .bellman import (  # noqa: F401
    DensityTracker,
    EvaluationDomain,
    FullDensity,
    MsmTable,
    SynthesisError,
    Worker,
    multiexp,
    pin_bases,
    unpin_bases,
)
