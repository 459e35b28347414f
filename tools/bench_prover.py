#!/usr/bin/env python3
"""The prover-shaped caller (phase2-bn254_amd/prover.py = groth16 create_proof, prover.rs:202-343) on a synthetic instance of
2^log_m constraints, data resident in HBM: 7 NTTs + the elementwise H steps + 8 multiexps (G1 and G2, three density maps,
Montgomery scalars).  Times the eight multiexps issued one after the other against issued from eight host threads on eight
streams (how the reference queues them before its first wait())."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench, oracle_lib as O

ap = argparse.ArgumentParser(); ap.add_argument("--log-m", type=int, default=20); ap.add_argument("--iters", type=int, default=9)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
m = 1 << a.log_m; num_inputs, num_aux = 16, m - 64
rng = np.random.default_rng(1)
def synth(group, n, seed):
    k = bench.gen_scalars(n, seed, dev); p = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    fn = L.mi355zk_bn254_g1_batch_mul_dev if group == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert fn(C.c_void_p(p.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    return p
def witness(n, seed):  # Montgomery Fr with the 0 / 1 mix of a real witness
    v = bench.gen_scalars(n, seed, dev)
    g = torch.Generator(device=dev); g.manual_seed(seed + 1)
    kind = torch.randint(0, 10, (n,), device=dev, generator=g)
    one_m = torch.tensor([v_ - (1 << 64) if v_ >= (1 << 63) else v_ for v_ in [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]], dtype=torch.int64, device=dev)
    v[kind < 3] = 0; v[(kind >= 3) & (kind < 6)] = one_m
    return v
a_bits, bi_bits, ba_bits = rng.random(num_aux) < 0.5, rng.random(num_inputs) < 0.5, rng.random(num_aux) < 0.4
na, nb_ = int(a_bits.sum()), int(bi_bits.sum()) + int(ba_bits.sum())
vk1 = O.G1.mul_many_affine(inputs.G1_GEN_RAW, inputs.random_scalars(3, seed=2)); vk2 = O.G2.mul_many_affine(inputs.G2_GEN_RAW, inputs.random_scalars(2, seed=3))
vk = {"alpha_g1": vk1[0], "beta_g1": vk1[1], "delta_g1": vk1[2], "beta_g2": vk2[0], "delta_g2": vk2[1]}
params = zk.prover.Parameters(vk, synth(1, m - 1, 10), synth(1, num_aux, 11), synth(1, num_inputs + na, 12), synth(1, nb_, 13), synth(2, nb_, 14))
abc = [bench.gen_scalars(m, 20 + i, dev) for i in range(3)]
inp, aux = witness(num_inputs, 30), witness(num_aux, 31)
dens = [zk.DensityTracker.from_bools(x) for x in (a_bits, bi_bits, ba_bits)]
out = {"log_m": a.log_m}
proofs = {}
tabled = params.with_tables()   # window tables for the vectors long enough to gain (G1 >= 2^19, G2 >= 2^16)
for name, conc, pr in (("sequential", False, params), ("eight_threads", True, params), ("eight_threads_tables", True, tabled)):
    def run():
        asg = zk.prover.ProvingAssignment(abc[0].clone(), abc[1].clone(), abc[2].clone(), inp, aux, *dens)
        return zk.prover.create_proof(w, pr, asg, 12345, 67890, concurrent=conc)
    for _ in range(3): proofs[name] = run()      # warm-up: tables, workspace pool, per-thread streams
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.iters):
        t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort(); out[name + "_ms"] = round(ts[len(ts) // 2], 2); out[name + "_min_ms"] = round(ts[0], 2)
out["same_proof"] = all(np.array_equal(x, y) and np.array_equal(x, z) for x, y, z in zip(proofs["sequential"], proofs["eight_threads"], proofs["eight_threads_tables"]))
print(json.dumps(out))
