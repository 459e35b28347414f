#!/usr/bin/env python3
"""Measurements for the SURVEY 8(f) "next" rows that are built: batch_exp (row 1), merge_pairs (row 2), the QAP sparse matvec (row 3) and the G1 point FFT (row 4).
G2 rows are measured twice: the default (plain windows: the reference's result for every record its decoders admit) and, keys ending in
"_trusted_subgroup", under the promise MI355ZK_G2_TRUSTED_SUBGROUP (psi-split scalars) -- the price of the exact default on honest data."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phase2_bn254_amd as zk, inputs, bench

ap = argparse.ArgumentParser(); ap.add_argument("--log-n", type=int, default=20); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
L = zk.lib.load(); w = zk.Worker(0); dev = torch.device("cuda", 0)
n = 1 << a.log_n
out = {"log_n": a.log_n}
for g, limbs, gen in ((1, 8, inputs.G1_GEN_RAW), (2, 16, inputs.G2_GEN_RAW)):
    k = bench.gen_scalars(n + 1, 31 + g, dev)
    bases = torch.empty((n + 1, limbs), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(gen)
    mul = L.mi355zk_bn254_g1_batch_mul_dev if g == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert mul(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n + 1, None) == 0
    torch.cuda.synchronize()
    sc = bench.gen_scalars(n, 41 + g, dev)
    res = torch.empty((n, limbs), dtype=torch.int64, device=dev)
    bexp = L.mi355zk_bn254_g1_batch_exp_dev if g == 1 else L.mi355zk_bn254_g2_batch_exp_dev
    TRUST = zk.lib.G2_TRUSTED_SUBGROUP
    modes = ((0, ""),) if g == 1 else ((0, ""), (TRUST, "_trusted_subgroup"))
    for same in (0, 1):
      for tr, suffix in modes:
        assert bexp(C.c_void_p(res.data_ptr()), C.c_void_p(bases.data_ptr()), C.c_void_p(sc.data_ptr()), n, same | tr, None) == 0; torch.cuda.synchronize()
        if tr: ref_res = res.clone()
        t = time.perf_counter()
        for _ in range(a.iters): bexp(C.c_void_p(res.data_ptr()), C.c_void_p(bases.data_ptr()), C.c_void_p(sc.data_ptr()), n, same | tr, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / a.iters
        out[f"g{g}_batch_exp_{'same_scalar' if same else 'per_point'}{suffix}"] = {"ms": round(dt * 1e3, 3), "Mpoint_per_s": round(n / dt / 1e6, 2)}
        if tr:
            assert bexp(C.c_void_p(res.data_ptr()), C.c_void_p(bases.data_ptr()), C.c_void_p(sc.data_ptr()), n, same, None) == 0
            out[f"g{g}_batch_exp_{'same_scalar' if same else 'per_point'}{suffix}"]["equals_default_on_subgroup_points"] = bool(torch.equal(res, ref_res))
    mp = L.mi355zk_bn254_g1_merge_pairs_dev if g == 1 else L.mi355zk_bn254_g2_merge_pairs_dev
    s, sx = np.zeros(12 * g, np.uint64), np.zeros(12 * g, np.uint64)
    args = (C.c_void_p(bases.data_ptr()), C.c_void_p(bases.data_ptr() + 64 * g), C.c_void_p(sc.data_ptr()), n, None, s.ctypes.data_as(C.c_void_p), sx.ctypes.data_as(C.c_void_p))
    assert mp(*args) == 0
    t = time.perf_counter()
    for _ in range(a.iters): mp(*args)
    dt = (time.perf_counter() - t) / a.iters
    out[f"g{g}_power_pairs"] = {"ms": round(dt * 1e3, 3), "Mscalar_mul_per_s_both_sums": round(2 * n / dt / 1e6, 2)}
    # row 3: QAP evaluation as a CSR sparse matvec over points: n variables, ~3 terms each, variable 0 ("one") in n/4 terms
    g_ = torch.Generator(device=dev); g_.manual_seed(77 + g)
    lens = torch.randint(0, 6, (n,), device=dev, generator=g_, dtype=torch.int64); lens[0] = n // 4
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(lens, 0)
    nnz = int(rp[-1].item())
    rp32 = rp.to(torch.int32); col = torch.randint(0, n, (nnz,), device=dev, generator=g_, dtype=torch.int32)
    cf = bench.gen_scalars(nnz, 61 + g, dev)
    smv = L.mi355zk_bn254_g1_sparse_matvec_dev if g == 1 else L.mi355zk_bn254_g2_sparse_matvec_dev
    sargs = (C.c_void_p(res.data_ptr()), C.c_void_p(bases.data_ptr()), n, C.c_void_p(rp32.data_ptr()), C.c_void_p(col.data_ptr()), C.c_void_p(cf.data_ptr()), n, nnz, None)
    for tr, suffix in modes:
        assert smv(*sargs, tr) == 0
        t = time.perf_counter()
        for _ in range(a.iters): assert smv(*sargs, tr) == 0
        dt = (time.perf_counter() - t) / a.iters
        out[f"g{g}_qap_sparse_matvec{suffix}"] = {"rows": n, "nnz": nnz, "ms": round(dt * 1e3, 2), "Mterm_per_s": round(nnz / dt / 1e6, 2)}
    # the same matrix with circom-like coefficients: 90 % of the terms +-1 (no scalar multiplication), 10 % general
    kind = torch.randint(0, 20, (nnz,), device=dev, generator=g_)
    cf1 = cf.clone()
    one = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=dev)
    rm1 = torch.tensor([0x43E1F593F0000000, 0x2833E84879B97091, 0xB85045B68181585D - (1 << 64), 0x30644E72E131A029], dtype=torch.int64, device=dev)
    cf1[kind < 9] = one
    cf1[(kind >= 9) & (kind < 18)] = rm1
    sargs1 = sargs[:5] + (C.c_void_p(cf1.data_ptr()),) + sargs[6:]
    for tr, suffix in modes:   # (G2 default: a coefficient r - 1 is a full multiplication -- (r - 1) P == -P only where r P == infinity)
        assert smv(*sargs1, tr) == 0
        t = time.perf_counter()
        for _ in range(a.iters): assert smv(*sargs1, tr) == 0
        dt = (time.perf_counter() - t) / a.iters
        out[f"g{g}_qap_sparse_matvec_90pct_unit_coeffs{suffix}"] = {"rows": n, "nnz": nnz, "ms": round(dt * 1e3, 2), "Mterm_per_s": round(nnz / dt / 1e6, 2)}
# row 4: point FFT (prepare_phase2's Lagrange-basis conversion)
for ln in (12, 16, a.log_n):
    m = 1 << ln
    k = bench.gen_scalars(m, 51, dev)
    pts = torch.empty((m, 8), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), m, None) == 0
    torch.cuda.synchronize()
    ref = pts.clone()
    t = time.perf_counter()
    assert L.mi355zk_bn254_g1_point_fft_dev(C.c_void_p(pts.data_ptr()), ln, 1, None) == 0
    dt = time.perf_counter() - t
    assert L.mi355zk_bn254_g1_point_fft_dev(C.c_void_p(pts.data_ptr()), ln, 0, None) == 0
    out[f"g1_point_ifft_2e{ln}"] = {"ms": round(dt * 1e3, 2), "Mbutterfly_per_s": round(m / 2 * ln / dt / 1e6, 2), "fft_of_ifft_is_identity": bool(torch.equal(pts, ref))}
# row 4: point codecs (accumulator files): encode, then decode (compressed = one sqrt per point)
for g, limbs in ((1, 8), (2, 16)):
    m = n if g == 1 else n // 4
    k = bench.gen_scalars(m, 71 + g, dev)
    pts = torch.empty((m, limbs), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if g == 1 else inputs.G2_GEN_RAW)
    mul = L.mi355zk_bn254_g1_batch_mul_dev if g == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert mul(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), m, None) == 0
    enc_f = L.mi355zk_bn254_g1_encode_dev if g == 1 else L.mi355zk_bn254_g2_encode_dev
    dec_f = L.mi355zk_bn254_g1_decode_dev if g == 1 else L.mi355zk_bn254_g2_decode_dev
    for comp in (0, 1):
        enc = torch.zeros((m, (32 if comp else 64) * g), dtype=torch.uint8, device=dev)
        back = torch.zeros_like(pts)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(a.iters): assert enc_f(C.c_void_p(enc.data_ptr()), C.c_void_p(pts.data_ptr()), m, comp, None) == 0
        torch.cuda.synchronize(); te = (time.perf_counter() - t) / a.iters
        t = time.perf_counter()
        for _ in range(a.iters): assert dec_f(C.c_void_p(back.data_ptr()), C.c_void_p(enc.data_ptr()), m, comp, 1, None, None) == 0
        td = (time.perf_counter() - t) / a.iters
        out[f"g{g}_codec_{'compressed' if comp else 'uncompressed'}"] = {"points": m, "encode_ms": round(te * 1e3, 3), "decode_checked_ms": round(td * 1e3, 3),
                                                                          "decode_Mpoint_per_s": round(m / td / 1e6, 1), "roundtrip_ok": bool(torch.equal(back, pts))}
for ln in (12, min(a.log_n, 18)):  # G2 leg (coeffs_g2)
    m = 1 << ln
    k = bench.gen_scalars(m, 52, dev)
    pts = torch.empty((m, 16), dtype=torch.int64, device=dev)
    gen = np.ascontiguousarray(inputs.G2_GEN_RAW)
    assert L.mi355zk_bn254_g2_batch_mul_dev(C.c_void_p(pts.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), m, None) == 0
    torch.cuda.synchronize()
    ref = pts.clone()
    for tr, suffix in ((0, ""), (zk.lib.G2_TRUSTED_SUBGROUP, "_trusted_subgroup")):
        t = time.perf_counter()
        assert L.mi355zk_bn254_g2_point_fft_dev(C.c_void_p(pts.data_ptr()), ln, 1 | tr, None) == 0
        dt = time.perf_counter() - t
        assert L.mi355zk_bn254_g2_point_fft_dev(C.c_void_p(pts.data_ptr()), ln, 0 | tr, None) == 0
        out[f"g2_point_ifft_2e{ln}{suffix}"] = {"ms": round(dt * 1e3, 2), "Mbutterfly_per_s": round(m / 2 * ln / dt / 1e6, 2), "fft_of_ifft_is_identity": bool(torch.equal(pts, ref))}
print(json.dumps(out))
