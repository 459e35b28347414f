#!/usr/bin/env python3
"""Differential fuzz of the ceremony-side entry points against the oracle, bit-exact on affine records:
batch_exp (per-point / one scalar, G1 / G2), dense_multiexp and merge_pairs (device and host-buffer forms), the QAP sparse matvec,
the point FFT (against the oracle's group-domain serial FFT where it exists: sizes the oracle does in seconds), and the point codecs
(encode -> oracle decode, oracle encode -> decode, both compressions).  Random sizes (incl. 1, 2, 3 and non-powers of two), infinity
records, zero / one / r - 1 / small scalars.
   python tools/fuzz_rows.py [--cases 60] [--seed 1]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--devices", type=int, default=1, help="k > 1: the host-buffer forms run over k logical devices (all GPU 0)")
a = ap.parse_args()
import bn254_model as M, inputs, oracle_lib as O
import phase2_bn254_amd as zk
zk.Worker(devices=[0] * a.devices) if a.devices > 1 else zk.Worker(0)
rng = np.random.default_rng(a.seed)
R = M.R_ORDER
pools = {g: inputs.bases_cpu(g, 40, seed=4100 + g) for g in (1, 2)}

def twist_points(count):
    """on-twist G2 records that are NOT in the order-r subgroup (x = (c0, 1) with a square root of x^3 + b'; r * P != infinity), and their sums with
    subgroup points: what the reference's decoders admit and its wNAF multiplies exactly (round 5: so do the default paths here)"""
    b = O.g2_coeff_b(); out = []
    mont = lambda v: np.array(M.to_limbs(M.to_mont(v, M.Q)), dtype=np.uint64)
    c0 = 5
    while len(out) < count:
        c0 += 1
        x = np.concatenate([mont(c0), mont(1)])
        rhs = O.fq2_mul(O.fq2_sqr(x), x)
        rhs = np.concatenate([O.fe_add(0, rhs[:4], b[:4]), O.fe_add(0, rhs[4:], b[4:])])
        y = O.fq2_sqrt(rhs)
        if y is None: continue
        pt = np.concatenate([x, y])
        if not O.G2.to_affine(O.G2.mul(O.G2.from_affine(pt), np.array(M.to_limbs(R), dtype=np.uint64))).any(): continue
        out.append(pt)
        out.append(O.G2.to_affine(O.G2.add_mixed(O.G2.from_affine(pools[2][len(out) % 40]), pt)))
    return np.stack(out[:count])
outside = twist_points(8)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
host = lambda t: t.cpu().numpy().view(np.uint64)

def points(g, n, allow_outside=False):
    p = pools[g][rng.integers(0, 40, n)].copy()
    if n > 2 and rng.random() < 0.5: p[rng.integers(0, n, max(1, n // 10))] = 0      # infinity records
    members = True
    if g == 2 and allow_outside and rng.random() < 0.5:                              # G2 records outside the order-r subgroup
        k = max(1, n // 6)
        p[rng.integers(0, n, k)] = outside[rng.integers(0, len(outside), k)]
        members = False
    return (p, members) if allow_outside else p

def scalars(n):
    s = inputs.random_scalars(n, seed=int(rng.integers(1 << 30)))
    kind = rng.integers(0, 5, n)
    s[kind == 0] = 0
    s[kind == 1] = np.array([1, 0, 0, 0], np.uint64)
    s[kind == 2] = np.array(M.to_limbs(R - 1), np.uint64)
    small = kind == 3
    s[small] = 0; s[small, 0] = rng.integers(0, 70000, int(small.sum())).astype(np.uint64)
    return s

def affine(G, xyz): return G.to_affine(np.ascontiguousarray(xyz))

bad = 0
def check(ok, what):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", what)

for case in range(a.cases):
    g = int(rng.integers(1, 3)); G = O.G1 if g == 1 else O.G2
    which = int(rng.integers(0, 6))
    if which == 0:      # batch_exp
        n = int(rng.choice([1, 2, 3, 17, 64, 65, 200, 257])) if g == 1 else int(rng.choice([1, 2, 3, 17, 40]))
        same = bool(rng.integers(0, 2))
        (p, members), s = points(g, n, True), scalars(1 if same else n)
        got = host(zk.ceremony.batch_exp(dev(p), dev(s), same_scalar=same))
        got_h = zk.ceremony.batch_exp_host(p, s, same_scalar=same)
        want = np.stack([G.to_affine(G.mul(G.from_affine(p[i]), s[0 if same else i])) for i in range(n)])
        ok = np.array_equal(got, want) and np.array_equal(got_h, want)
        if members:   # every record in the subgroup: the promise may be given, and the split kernels return the same records
            ok = ok and np.array_equal(host(zk.ceremony.batch_exp(dev(p), dev(s), same_scalar=same, trusted_subgroup=True)), want)
        check(ok, f"batch_exp g{g} n={n} same={same} members={members} (case {case})")
    elif which == 1:    # dense_multiexp
        n = int(rng.choice([1, 2, 5, 33, 100, 1000, 2500])) if g == 1 else int(rng.choice([1, 2, 5, 33, 300]))
        p, s = points(g, n), scalars(n)
        want = affine(G, G.dense_multiexp(p, s))
        check(np.array_equal(affine(G, zk.ceremony.dense_multiexp(dev(p), dev(s))), want) and
              np.array_equal(affine(G, zk.ceremony.dense_multiexp_host(p, s)), want), f"dense_multiexp g{g} n={n} (case {case})")
    elif which == 2:    # merge_pairs / power_pairs
        n = int(rng.choice([1, 2, 5, 33, 100, 1000])) if g == 1 else int(rng.choice([1, 2, 5, 33, 200]))
        v = points(g, n + 1); rho = scalars(n)
        w1, w2 = affine(G, G.dense_multiexp(v[:-1], rho)), affine(G, G.dense_multiexp(v[1:], rho))
        s1, s2 = zk.ceremony.power_pairs(dev(v), dev(rho))
        h1, h2 = zk.ceremony.merge_pairs_host(v[:-1], v[1:], rho)
        check(np.array_equal(affine(G, s1), w1) and np.array_equal(affine(G, s2), w2) and np.array_equal(affine(G, h1), w1) and np.array_equal(affine(G, h2), w2),
              f"merge_pairs g{g} n={n} (case {case})")
    elif which == 3:    # QAP sparse matvec
        nb = int(rng.integers(1, 40)); rows = int(rng.integers(1, 60))
        p, members = points(g, nb, True)
        lens = rng.integers(0, 5, rows); lens[rng.integers(0, rows)] = int(rng.integers(0, 120 if g == 1 else 30))
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32); nnz = int(rp[-1])
        col = rng.integers(0, nb, nnz).astype(np.uint32); cf = scalars(max(nnz, 1))[:nnz]
        d32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).cuda()
        if nnz == 0: continue
        got = host(zk.ceremony.eval_qap(dev(p), d32(rp), d32(col), dev(cf)))
        got_h = zk.ceremony.eval_qap_host(p, rp, col, cf)
        want = np.zeros((rows, G.aff), np.uint64)
        for r in range(rows):
            acc = G.from_affine(np.zeros(G.aff, np.uint64))
            for t in range(rp[r], rp[r + 1]): acc = G.add(acc, G.mul(G.from_affine(p[col[t]]), cf[t]))
            want[r] = G.to_affine(acc)
        ok = np.array_equal(got, want) and np.array_equal(got_h, want)
        if members: ok = ok and np.array_equal(host(zk.ceremony.eval_qap(dev(p), d32(rp), d32(col), dev(cf), trusted_subgroup=True)), want)
        check(ok, f"sparse_matvec g{g} rows={rows} nnz={nnz} members={members} (case {case})")
    elif which == 4:    # codecs
        n = int(rng.choice([1, 2, 31, 100])); comp = bool(rng.integers(0, 2))
        p = points(g, n)
        enc = zk.ceremony.encode_points(dev(p), comp).cpu().numpy()
        want_enc = O.encode_points(g, p, comp)
        rc, _, back = O.decode_points(g, enc, comp, True)
        dec = host(zk.ceremony.decode_points(torch.from_numpy(want_enc).cuda(), g, comp, True))
        check(np.array_equal(enc, want_enc) and rc == 0 and np.array_equal(back, p) and np.array_equal(dec, p), f"codec g{g} n={n} compressed={comp} (case {case})")
    else:               # point FFT round trip + linearity: ifft(fft(v)) == v, and fft(v)[0] == sum v
        log_n = int(rng.integers(0, 7 if g == 1 else 7)); n = 1 << log_n
        p = pools[g][rng.integers(0, 40, n)].copy()
        if g == 2 and n >= 4 and rng.random() < 0.5:
            # records outside the subgroup: omega^n == 1 holds mod r only, so ifft(fft(v)) == v is NOT an identity for them (nor in the reference);
            # the transform itself is still the group law applied in the reference's order: compare with the oracle's Point<G2> FFT directly
            p[rng.integers(0, n, max(1, n // 8))] = outside[rng.integers(0, len(outside), max(1, n // 8))]
            op = ("fft", "ifft")[int(rng.integers(2))]
            got = host((zk.ceremony.point_fft if op == "fft" else zk.ceremony.point_ifft)(dev(p)))
            check(np.array_equal(got, O.point_domain_op(2, p, log_n, op)), f"point_{op} g2 n={n} with records outside the subgroup (case {case})")
            continue
        d = dev(p)
        f = host(zk.ceremony.point_fft(d.clone()))
        back = host(zk.ceremony.point_ifft(dev(f)))
        tot = G.from_affine(np.zeros(G.aff, np.uint64))
        for i in range(n): tot = G.add(tot, G.from_affine(p[i]))
        check(np.array_equal(back, p) and np.array_equal(affine(G, G.from_affine(f[0])), G.to_affine(tot)), f"point_fft g{g} n={n} (case {case})")
print(f"fuzz_rows: {a.cases} cases, {bad} mismatches (seed {a.seed}, {a.devices} logical device(s))")
sys.exit(1 if bad else 0)
