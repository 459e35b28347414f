#!/usr/bin/env python3
"""Differential fuzz of the multiexp entry points against the oracle (run on an MI355X; not part of the pytest suite):
random sizes, densities, source offsets, zero / tiny / maximal scalars, and base pools so small that buckets see the same
point twice (doubling branch) and P next to -P (infinity branch).  The window layout is whatever MI355ZK_MSM_C /
MI355ZK_MSM_RADIX select for this process; tools/fuzz_msm.sh sweeps several."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phase2_bn254_amd as zk, inputs, oracle_lib as O, bn254_model as M

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=40); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=3000)
ap.add_argument("--table", action="store_true", help="evaluate through TABLE MODE (MsmTable of the device-resident vector); MI355ZK_MSM_TABLE_C selects the table's window width")
ap.add_argument("--devices", type=int, default=1, help="k > 1: the single-process multi-GPU mode over k logical devices (all GPU 0), calls cut from 64 exponents on")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
if a.devices > 1:
    os.environ["MI355ZK_MULTI_MIN_LOG"] = "6"
    w = zk.Worker(devices=[0] * a.devices)
else:
    w = zk.Worker(0)
pools = {g: inputs.bases_cpu(g, 24, seed=900 + g) for g in (1, 2)}
for g in (1, 2):  # negatives of the first pool entries: P and -P collide in buckets
    G = O.G1 if g == 1 else O.G2
    for i in range(4):
        neg = G.from_affine(pools[g][i]); y = pools[g][i].copy()
        # -P: negate y in the oracle's field
        ny = [O.fe_sub(O.FQ, np.zeros(4, np.uint64), y[4 * k:4 * k + 4]) for k in range(G.aff // 4 // 2, G.aff // 4)]
        pools[g][12 + i] = np.concatenate([y[:G.aff // 2]] + ny)
bad = 0
for case in range(a.cases):
    g = int(rng.integers(1, 3))
    G = O.G1 if g == 1 else O.G2
    n = int(rng.integers(1, a.max_n if g == 1 else a.max_n // 4 + 2))
    pool = int(rng.choice([2, 5, 24]))
    scal_kind = int(rng.integers(0, 4))
    if scal_kind == 0: sc = inputs.random_scalars(n, seed=int(rng.integers(1 << 30)))
    elif scal_kind == 1: sc = np.zeros((n, 4), np.uint64); sc[:, 0] = rng.integers(0, 4, n)
    elif scal_kind == 2:
        sc = inputs.random_scalars(n, seed=int(rng.integers(1 << 30))); sc[rng.random(n) < 0.3] = np.array(M.to_limbs(M.R_ORDER - 1), np.uint64)
    else:
        base_s = inputs.random_scalars(3, seed=int(rng.integers(1 << 30))); sc = base_s[rng.integers(0, 3, n)]  # identical scalars -> identical digits
    use_density = rng.random() < 0.4
    offset = int(rng.integers(0, 5))
    if use_density:
        bits = rng.random(n) < 0.6
        nb_needed = int(bits.sum())
        dm = zk.DensityTracker.from_bools(bits)
        dens_words = np.zeros((n + 31) // 32, np.uint32)
        for i in np.nonzero(bits)[0]: dens_words[i >> 5] |= np.uint32(1) << np.uint32(i & 31)
    else:
        nb_needed = n; dm = zk.FullDensity(); dens_words = None
    bases = pools[g][rng.integers(0, pool if pool < 24 else 24, offset + nb_needed)]
    if pool == 2: bases = pools[g][np.where(rng.random(offset + nb_needed) < 0.5, 0, 12)]  # P and -P only
    rc_o, want = G.multiexp(bases, sc, density=dens_words, density_bits=n if use_density else None, base_offset=offset, threads=4)
    try:
        if a.table:
            import torch
            dv = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
            got = zk.multiexp(w, (zk.MsmTable(dv(bases)), offset), dm, dv(sc)).wait()
        else:
            got = zk.multiexp(w, (bases, offset), dm, sc).wait()
        ok = rc_o == 0 and np.array_equal(G.to_affine(got), G.to_affine(want))
    except zk.SynthesisError as e:
        ok = rc_o != 0
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: g={g} n={n} pool={pool} scal={scal_kind} density={use_density} offset={offset} rc_o={rc_o}")
print(f"fuzz{' (table mode)' if a.table else ''}{' (%d devices)' % a.devices if a.devices > 1 else ''}: {a.cases - bad}/{a.cases} ok  (C={os.environ.get('MI355ZK_MSM_C')}, RADIX={os.environ.get('MI355ZK_MSM_RADIX')}, TABLE_C={os.environ.get('MI355ZK_MSM_TABLE_C')})")
sys.exit(1 if bad else 0)
