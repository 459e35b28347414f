#!/usr/bin/env python3
"""Differential fuzz of the Fr NTT entry points against the oracle's serial_fft (bit-exact): random sizes 2^1 .. 2^max_log, the four domain
operations, and element patterns that push the lazy-reduction bookkeeping of the stage code to its edges -- all r - 1, alternating 0 / r - 1,
one non-zero element, small values, uniform -- in the canonical Montgomery memory format.
   python tools/fuzz_ntt.py [--cases 120] [--seed 1] [--max-log 18]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=120); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--max-log", type=int, default=18)
a = ap.parse_args()
import bn254_model as M, inputs, oracle_lib as O
import phase2_bn254_amd as zk
worker = zk.Worker(0)
rng = np.random.default_rng(a.seed)
R = M.R_ORDER
mont = lambda v: np.array(M.to_limbs(v * (1 << 256) % R), dtype=np.uint64)   # the memory format: v * 2^256 mod r
bad = 0
for case in range(a.cases):
    log_n = int(rng.integers(1, a.max_log + 1)); n = 1 << log_n
    op = ("fft", "ifft", "coset_fft", "icoset_fft")[int(rng.integers(4))]
    kind = int(rng.integers(6))
    if kind == 0: x = inputs.random_fr_mont(n, seed=int(rng.integers(1 << 30)))
    elif kind == 1: x = np.tile(mont(R - 1), (n, 1))
    elif kind == 2: x = np.tile(mont(R - 1), (n, 1)); x[::2] = 0
    elif kind == 3: x = np.zeros((n, 4), np.uint64); x[int(rng.integers(n))] = mont(int(rng.integers(1, 1 << 62)))
    elif kind == 4: x = np.stack([mont(int(v)) for v in rng.integers(0, 4, size=min(n, 4096))]); x = np.tile(x, (n // len(x), 1))
    else:   # the raw limb pattern just below r (canonical whatever it stands for)
        x = np.tile(np.array(M.to_limbs(R - 1 - int(rng.integers(0, 3))), dtype=np.uint64), (n, 1)); x[rng.random(n) < 0.5] = 0
    want = O.fr_domain_op(x.copy(), log_n, op).reshape(-1, 4)
    dom = zk.EvaluationDomain.from_coeffs(x.copy())
    getattr(dom, op)(worker)
    got = dom.into_coeffs()
    if not np.array_equal(got, want):
        bad += 1
        print(f"MISMATCH case {case}: log_n={log_n} op={op} kind={kind}")
print(f"fuzz_ntt: {a.cases - bad}/{a.cases} ok (seed {a.seed}, sizes up to 2^{a.max_log})")
sys.exit(1 if bad else 0)
