"""Deterministic synthetic inputs for tests and bench.  TEST/BENCH INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np

import bn254_model as M

R_LIMBS = np.array(M.to_limbs(M.R_ORDER), dtype=np.uint64)
Q_LIMBS = np.array(M.to_limbs(M.Q), dtype=np.uint64)


def _lt(a: np.ndarray, mod: np.ndarray) -> np.ndarray:
    """row-wise a < mod on (n,4) little-endian u64 limbs"""
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < mod[i])
        eq &= a[:, i] == mod[i]
    return lt


def random_below(n: int, mod_limbs: np.ndarray, seed: int) -> np.ndarray:
    """(n,4) u64 uniform in [0, mod) by rejection on 254-bit draws."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << 62) - 1)
    bad = ~_lt(out, mod_limbs)
    while bad.any():
        k = int(bad.sum())
        fresh = rng.integers(0, 1 << 64, size=(k, 4), dtype=np.uint64)
        fresh[:, 3] &= np.uint64((1 << 62) - 1)
        out[bad] = fresh
        bad = ~_lt(out, mod_limbs)
    return out


def random_scalars(n: int, seed: int = 1) -> np.ndarray:
    """canonical FrRepr scalars, uniform in [0, r)"""
    return random_below(n, R_LIMBS, seed)


def random_fr_mont(n: int, seed: int = 2) -> np.ndarray:
    """uniform Fr elements in Montgomery form (any value < r is the Montgomery form of a uniform element)"""
    return random_below(n, R_LIMBS, seed)


G1_GEN_RAW = np.array(M.g1_affine_to_raw(M.G1_GEN), dtype=np.uint64)
G2_GEN_RAW = np.array(M.g2_affine_to_raw(M.G2_GEN), dtype=np.uint64)


def bases_cpu(group, n: int, seed: int = 3) -> np.ndarray:
    """n affine raw records P_i = k_i * G (k_i random) via the oracle -- small n only."""
    import oracle_lib as O

    g = O.G1 if group == 1 else O.G2
    gen = G1_GEN_RAW if group == 1 else G2_GEN_RAW
    return g.mul_many_affine(gen, random_scalars(n, seed))


def bases_progression_cpu(group, n: int, seed: int = 4) -> np.ndarray:
    """n distinct affine raw records P_i = k0*G + i*(k1*G): one mixed add per point on the CPU."""
    import oracle_lib as O

    g = O.G1 if group == 1 else O.G2
    gen = G1_GEN_RAW if group == 1 else G2_GEN_RAW
    se = g.mul_many_affine(gen, random_scalars(2, seed))
    return g.arith_progression_affine(se[0], se[1], n)
