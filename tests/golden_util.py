"""Loads tests/golden/*.json into the boundary's byte layouts (numpy uint64 limb arrays)."""
import json
import os

import numpy as np

import bn254_model as M

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def msm_cases(group: int):
    """yields dicts: name, bases (n,8g) u64, scalars (n,4) u64, density (list|None), base_offset, rc, expected affine raw (8g,) u64 | None"""
    key = "g1" if group == 1 else "g2"
    for c in _load("msm_golden.json")[key]:
        def conv(p):
            if p == "inf":
                return None
            v = [int(x, 16) for x in p]
            return (v[0], v[1]) if group == 1 else ((v[0], v[1]), (v[2], v[3]))
        to_raw = M.g1_affine_to_raw if group == 1 else M.g2_affine_to_raw
        bases = np.array([to_raw(conv(p)) for p in c["bases"]], dtype=np.uint64).reshape(-1, 8 * group)
        scalars = np.array([M.to_limbs(int(s, 16)) for s in c["scalars"]], dtype=np.uint64).reshape(-1, 4)
        exp = None if c["rc"] != 0 else np.array(to_raw(conv(c["expected"])), dtype=np.uint64)
        yield {"name": c["name"], "bases": bases, "scalars": scalars, "density": c["density"], "base_offset": c["base_offset"],
               "rc": c["rc"], "expected": exp}


def density_words(bits):
    w = np.zeros((len(bits) + 31) // 32 or 1, dtype=np.uint32)
    for i, b in enumerate(bits):
        if b:
            w[i // 32] |= np.uint32(1 << (i % 32))
    return w


def ntt_cases():
    """yields dicts: log_n, input (n,4) Montgomery u64, omega (4,) Montgomery, and expected per op (n,4) Montgomery"""
    for c in _load("ntt_golden.json"):
        def mont(vals):
            return np.array([M.to_limbs(M.to_mont(int(v, 16), M.R_ORDER)) for v in vals], dtype=np.uint64).reshape(-1, 4)
        out = {"log_n": c["log_n"], "input": mont(c["input"]), "omega": mont([c["omega"]])[0]}
        for op in ("fft", "ifft", "coset_fft", "icoset_fft"):
            out[op] = mont(c[op])
        yield out
