#!/usr/bin/env python3
"""Generates tests/golden/*.json from the independent Python big-int model (tests/bn254_model.py).

The reference holds no BN254 known-answer vectors for multiexp / fft and
cannot be built here, so these vectors come from textbook affine chord-tangent arithmetic and the
O(n^2)/recursive DFT definition -- code that shares nothing with oracle/*.c or the HIP kernels.
Run:  python tests/golden/gen_golden.py     (deterministic; rewrites the JSON files)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bn254_model as M  # noqa: E402

rnd = random.Random(0x3DBE6259)


def hx(v):
    return "%064x" % v


def pt1(p):
    return "inf" if p is None else [hx(p[0]), hx(p[1])]


def pt2(p):
    return "inf" if p is None else [hx(p[0][0]), hx(p[0][1]), hx(p[1][0]), hx(p[1][1])]


def msm_case(name, F, gen, pts, scalars, density=None, offset=0, pt=pt1, expect_rc=0):
    """expected = sum over selected exponents of scalar * base, bases compacted by density (source.rs)."""
    acc = None
    cur = offset
    sel = density if density is not None else [1] * len(scalars)
    for k, d in zip(scalars, sel):
        if not d:
            continue
        if expect_rc == 0:
            acc = M.ec_add(F, acc, M.ec_mul(F, pts[cur], k))
        cur += 1
    return {"name": name, "bases": [pt(p) for p in pts], "scalars": [hx(k) for k in scalars],
            "density": density, "base_offset": offset, "rc": expect_rc, "expected": pt(acc) if expect_rc == 0 else None}


def rand_points(F, gen, n):
    return [M.ec_mul(F, gen, rnd.randrange(1, M.R_ORDER)) for _ in range(n)]


def gen_msm():
    out = {"g1": [], "g2": []}
    for key, F, gen, pt in (("g1", M.FQ_OPS, M.G1_GEN, pt1), ("g2", M.FQ2_OPS, M.G2_GEN, pt2)):
        n = 48 if key == "g1" else 12
        pts = rand_points(F, gen, n)
        sc = [rnd.randrange(M.R_ORDER) for _ in range(n)]
        cases = out[key]
        cases.append(msm_case("random", F, gen, pts, sc, pt=pt))
        cases.append(msm_case("single", F, gen, pts[:1], sc[:1], pt=pt))
        cases.append(msm_case("scalars_0_1_rminus1", F, gen, pts[:6], [0, 1, M.R_ORDER - 1, 1, 0, 2], pt=pt))
        cases.append(msm_case("all_zero_scalars", F, gen, pts[:5], [0] * 5, pt=pt))
        # duplicates: the same point many times with the same scalar -> same bucket, P + P doubling path
        cases.append(msm_case("duplicate_points_same_bucket", F, gen, [pts[0]] * 8, [sc[0]] * 8, pt=pt))
        # P and -P with equal scalars cancel inside one bucket (H == 0 -> infinity, ec.rs:487)
        cases.append(msm_case("p_and_minus_p", F, gen, [pts[1], M.ec_neg(F, pts[1]), pts[2]], [sc[1], sc[1], sc[2]], pt=pt))
        cases.append(msm_case("total_is_infinity", F, gen, [pts[3], M.ec_neg(F, pts[3])], [5, 5], pt=pt))
        # infinity base is fine under a zero scalar (multiexp.rs:95-96) ...
        cases.append(msm_case("infinity_base_zero_scalar", F, gen, [pts[0], None, pts[1]], [sc[0], 0, sc[1]], pt=pt))
        # ... and an error under a non-zero one (source.rs:50-52)
        cases.append(msm_case("infinity_base_nonzero_scalar", F, gen, [pts[0], None, pts[1]], [sc[0], 7, sc[1]], pt=pt, expect_rc=1))
        cases.append(msm_case("infinity_base_scalar_one", F, gen, [pts[0], None], [sc[0], 1], pt=pt, expect_rc=1))
        # density maps: bases are compacted
        dens = [1, 0, 1, 1, 0, 0, 1, 0, 1, 1]
        used = sum(dens)
        cases.append(msm_case("density", F, gen, pts[:used], sc[:10], density=dens, pt=pt))
        cases.append(msm_case("density_with_offset", F, gen, pts[:used + 3], sc[:10], density=dens, offset=3, pt=pt))
        cases.append(msm_case("base_offset", F, gen, pts[:10], sc[:6], offset=4, pt=pt))
        # bases run out (source.rs:46-48)
        cases.append(msm_case("eof", F, gen, pts[:4], sc[:6], pt=pt, expect_rc=2))
        cases.append(msm_case("eof_offset_past_end", F, gen, pts[:2], sc[:2], offset=2, pt=pt, expect_rc=2))
        # small scalars (only the lowest window is populated), powers of two (window boundaries)
        cases.append(msm_case("small_scalars", F, gen, pts[:8], [3, 5, 7, 2, 1 << 13, (1 << 14) - 1, 1 << 16, 65537], pt=pt))
        cases.append(msm_case("powers_of_two", F, gen, pts[:8], [1 << e for e in (0, 15, 16, 31, 32, 63, 64, 253)], pt=pt))
    return out


def gen_ntt():
    cases = []
    for log_n in (0, 1, 2, 3, 4, 6, 10):
        n = 1 << log_n
        a = [rnd.randrange(M.R_ORDER) for _ in range(n)]
        w = M.domain_omega(log_n)
        entry = {"log_n": log_n, "input": [hx(v) for v in a], "omega": hx(w)}
        ref = M.dft(a, w) if n <= 64 else M.fft_recursive(a, w)
        if n <= 64:
            assert ref == M.fft_recursive(a, w)
        entry["fft"] = [hx(v) for v in ref]
        for op in ("ifft", "coset_fft", "icoset_fft"):
            entry[op] = [hx(v) for v in M.domain_op(a, op)]
        cases.append(entry)
    return cases


if __name__ == "__main__":
    with open(os.path.join(HERE, "msm_golden.json"), "w") as f:
        json.dump(gen_msm(), f, indent=0)
    with open(os.path.join(HERE, "ntt_golden.json"), "w") as f:
        json.dump(gen_ntt(), f, indent=0)
    print("wrote msm_golden.json, ntt_golden.json")
