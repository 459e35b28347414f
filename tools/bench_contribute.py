#!/usr/bin/env python3
"""BASELINE.json config 5 on one MI355X: the device work of phase2 `MPCParameters::contribute` on synthetic
parameters of a ~2^20-constraint circuit (|L| = 2^20 G1 points, |H| = 2^20 - 1 G1 points; phase2/src/parameters.rs:
414-522: L and H times delta^-1 by `batch_exp`, affine out) and of the check `verify_contribution` runs over them
(`merge_pairs` of the before / after vectors, phase2/src/utils.rs:59-105 -> same_ratio against (delta_g2_after,
delta_g2_before)).  Points stay in HBM.  Checks: (1) spot check of L'[i] = delta^-1 * L[i] against the oracle,
(2) the same_ratio statement with the known delta: sum rho_i L[i] == delta * sum rho_i L'[i]."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench, bn254_model as M

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n_l, n_h = 1 << a.log_n, (1 << a.log_n) - 1
gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
def synth(n, seed):
    k = bench.gen_scalars(n, seed, dev); p = torch.empty((n, 8), dtype=torch.int64, device=dev)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(p.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
    return p
l_before, h_before = synth(n_l, 901), synth(n_h, 902)
delta = 0x0123456789ABCDEF0FEDCBA9876543210123456789ABCDEF % M.R_ORDER
delta_inv = torch.from_numpy(np.array([M.to_limbs(pow(delta, -1, M.R_ORDER))], dtype=np.uint64).view(np.int64)).to(dev)
torch.cuda.synchronize()
zk.ceremony.batch_exp(l_before[:1024], delta_inv, same_scalar=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.iters):
    l_after = zk.ceremony.batch_exp(l_before, delta_inv, same_scalar=True)
    h_after = zk.ceremony.batch_exp(h_before, delta_inv, same_scalar=True)
torch.cuda.synchronize(); t_contribute = (time.perf_counter() - t) / a.iters
import oracle_lib as O
hl, hla = l_before[:3].cpu().numpy().view(np.uint64), l_after[:3].cpu().numpy().view(np.uint64)
spot = all(np.array_equal(hla[i], O.G1.to_affine(O.G1.mul(O.G1.from_affine(hl[i]), delta_inv.cpu().numpy().view(np.uint64)[0]))) for i in range(3))
rho_l, rho_h = bench.gen_scalars(n_l, 903, dev), bench.gen_scalars(n_h, 904, dev)
zk.ceremony.merge_pairs(l_before, l_after, rho_l)
t = time.perf_counter()
for _ in range(a.iters):
    s_l, sx_l = zk.ceremony.merge_pairs(l_before, l_after, rho_l)
    s_h, sx_h = zk.ceremony.merge_pairs(h_before, h_after, rho_h)
t_verify = (time.perf_counter() - t) / a.iters
dl = np.array(M.to_limbs(delta), dtype=np.uint64)
ratio = all(np.array_equal(O.G1.to_affine(s), O.G1.to_affine(O.G1.mul(sx, dl))) for s, sx in ((s_l, sx_l), (s_h, sx_h)))
# BASELINE config 1 on the device: powersoftau compute_constrained over a blank accumulator (new_constrained), power 12 and 18
pot = {}
for power in (12, 18):
    n, n1 = 1 << power, (2 << power) - 1
    g1r, g2r = torch.from_numpy(np.ascontiguousarray(inputs.G1_GEN_RAW).view(np.int64)).to(dev), torch.from_numpy(np.ascontiguousarray(inputs.G2_GEN_RAW).view(np.int64)).to(dev)
    blank = {"hash": torch.zeros(64, dtype=torch.uint8, device=dev), "tau_g1": g1r.repeat(n1, 1), "tau_g2": g2r.repeat(n, 1),
             "alpha_g1": g1r.repeat(n, 1), "beta_g1": g1r.repeat(n, 1), "beta_g2": g2r.repeat(1, 1)}
    zk.ceremony.contribute_accumulator(blank, 3, 5, 7); torch.cuda.synchronize()
    t = time.perf_counter()
    acc = zk.ceremony.contribute_accumulator(blank, 0x1234567890ABCDEF, 0xFEDCBA, 0x13579BDF)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter()
    blob = zk.ceremony.write_accumulator(acc, compressed=True); back = zk.ceremony.read_accumulator(blob, power, compressed=True)
    torch.cuda.synchronize(); dt_io = time.perf_counter() - t
    pot[f"powersoftau_compute_2e{power}"] = {"ms": round(dt * 1e3, 2), "write_plus_read_compressed_ms": round(dt_io * 1e3, 2),
                                             "roundtrip_ok": bool(torch.equal(back["tau_g1"], acc["tau_g1"]) and torch.equal(back["tau_g2"], acc["tau_g2"]))}
print(json.dumps({**pot, "config": "phase2 contribute, synthetic params |L| = 2^%d, |H| = 2^%d - 1 (G1), 1 GPU" % (a.log_n, a.log_n),
                  "contribute_batch_exp_ms": round(t_contribute * 1e3, 2), "contribute_Mpoint_per_s": round((n_l + n_h) / t_contribute / 1e6, 2),
                  "verify_merge_pairs_ms": round(t_verify * 1e3, 2), "spot_check_vs_oracle": bool(spot), "same_ratio_with_known_delta": bool(ratio)}))
