#!/usr/bin/env python3
"""Concurrency stress: T host threads, each on its own HIP stream (and two of them on the SAME default stream), issue a random mix of
transforms (2^8 .. 2^15, all four domain operations), multiexps (device-resident and host-buffer, G1 / G2), batch_exp (per point / one
scalar) and merge_pairs for `--seconds`, every result compared with an expectation computed by the oracle beforehand.  What it
exercises is the library's shared state: table / scratch caches, workspace leases, the launch-order locks, the profiling-event pool.
   python tools/stress_threads.py [--threads 8] [--seconds 20] [--seed 1]"""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--threads", type=int, default=8); ap.add_argument("--seconds", type=float, default=20); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--devices", type=int, default=1); ap.add_argument("--prof", action="store_true", help="with the per-kernel HIP events on (the event pool under concurrent callers)")
a = ap.parse_args()
import bn254_model as M, inputs, oracle_lib as O
import phase2_bn254_amd as zk
worker = zk.Worker(devices=[0] * a.devices) if a.devices > 1 else zk.Worker(0)
rng = np.random.default_rng(a.seed)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
host = lambda t: t.cpu().numpy().view(np.uint64)
jobs = []   # (name, callable returning np array, expected)
for log_n in (8, 10, 11, 13, 15):
    x = inputs.random_fr_mont(1 << log_n, seed=100 + log_n)
    for op in ("fft", "ifft", "coset_fft", "icoset_fft"):
        want = O.fr_domain_op(x.copy(), log_n, op).reshape(-1, 4)
        def f(x=x, op=op):
            dom = zk.EvaluationDomain.from_coeffs(x.copy()); getattr(dom, op)(worker); return dom.into_coeffs()
        jobs.append((f"{op} 2^{log_n}", f, want))
for log_n in (9, 12, 14):       # (round 5) three arrays per call: one launch per pass over all of them (mi355zk_bn254_fr_domain_op_batch_dev)
    xs = [inputs.random_fr_mont(1 << log_n, seed=150 + 10 * log_n + t) for t in range(3)]
    for op in ("ifft", "coset_fft", "icoset_fft"):
        want = np.concatenate([O.fr_domain_op(x.copy(), log_n, op).reshape(-1, 4) for x in xs])
        def fb(xs=xs, op=op, log_n=log_n):
            doms = [zk.EvaluationDomain(dev(x), log_n) for x in xs]
            getattr(zk.EvaluationDomain, op + "_many")(worker, doms)
            return np.concatenate([host(d.coeffs) for d in doms])
        jobs.append((f"{op}_many 2^{log_n} x 3", fb, want))
for g, G in ((1, O.G1), (2, O.G2)):
    for n in ((700, 3000) if g == 1 else (300,)):
        b = inputs.bases_progression_cpu(g, n, seed=300 + g + n); s = inputs.random_scalars(n, seed=400 + n)
        rc, xyz = G.multiexp(b, s); assert rc == 0
        want = G.to_affine(xyz)
        db, ds = dev(b), dev(s)
        def fd(db=db, ds=ds, G=G): return G.to_affine(np.ascontiguousarray(zk.multiexp(worker, (db, 0), zk.FullDensity(), ds).wait()))
        def fh(b=b, s=s, G=G): return G.to_affine(np.ascontiguousarray(zk.multiexp(worker, (b, 0), zk.FullDensity(), s).wait()))
        jobs.append((f"multiexp g{g} n={n} (device)", fd, want)); jobs.append((f"multiexp g{g} n={n} (host)", fh, want))
    n = 96 if g == 1 else 24
    b = inputs.bases_progression_cpu(g, n, seed=500 + g); s = inputs.random_scalars(n, seed=501)
    for same in (False, True):
        sc = s[:1] if same else s
        want = np.stack([G.to_affine(G.mul(G.from_affine(b[i]), sc[0 if same else i])) for i in range(n)])
        db, ds = dev(b), dev(sc)
        def fe(db=db, ds=ds, same=same): return host(zk.ceremony.batch_exp(db, ds, same_scalar=same))
        jobs.append((f"batch_exp g{g} same={same}", fe, want))
    v = inputs.bases_progression_cpu(g, 400 if g == 1 else 100, seed=600 + g); rho = inputs.random_scalars(len(v) - 1, seed=601)
    w1, w2 = G.to_affine(G.dense_multiexp(v[:-1], rho)), G.to_affine(G.dense_multiexp(v[1:], rho))
    dv, dr = dev(v), dev(rho)
    def fm(dv=dv, dr=dr, G=G):
        s1, s2 = zk.ceremony.power_pairs(dv, dr); return np.concatenate([G.to_affine(s1), G.to_affine(s2)])
    jobs.append((f"power_pairs g{g}", fm, np.concatenate([w1, w2])))
print(f"{len(jobs)} kinds of call prepared", flush=True)
if a.prof: zk.lib.load().mi355zk_prof_enable(1)
for _name, _f, _w in jobs: _f()          # every kind once: caches, tables and pools are in place before the memory reading
torch.cuda.synchronize()
free0, res0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
bad, count, lock = [], [0], threading.Lock()
t_end = time.time() + a.seconds
def run(tid):
    r = np.random.default_rng(a.seed * 1000 + tid)
    st = torch.cuda.Stream() if tid >= 2 else None          # threads 0 and 1 share the default stream
    ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        while time.time() < t_end and not bad:
            name, f, want = jobs[int(r.integers(len(jobs)))]
            try:
                got = f()
            except Exception as e:   # noqa: BLE001
                with lock: bad.append(f"{name}: {type(e).__name__}: {e}")
                return
            if not np.array_equal(got, want):
                with lock: bad.append(f"{name}: wrong result (thread {tid})")
                return
            with lock: count[0] += 1
            if a.prof and tid == 0 and count[0] % 500 == 0: zk.lib.load().mi355zk_prof_reset()
ths = [threading.Thread(target=run, args=(i,)) for i in range(a.threads)]
[t.start() for t in ths]; [t.join() for t in ths]
torch.cuda.synchronize()
free1, res1 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
print(f"device memory in use grew by {(free0 - free1) / 2**20:.1f} MiB over the run, {(res1 - res0) / 2**20:.1f} MiB of it in torch's caching allocator (the test's own tensors, per stream); the rest is the library's (per-stream scratch, workspaces of concurrent calls)")
print(f"stress_threads: {count[0]} calls from {a.threads} threads in {a.seconds:.0f} s, {len(bad)} failures (seed {a.seed}, {a.devices} logical device(s))")
for b in bad: print("  FAIL", b)
sys.exit(1 if bad else 0)
