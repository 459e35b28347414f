#!/usr/bin/env python3
"""Experiment: can a latency-bound kernel chain (the bucket reduction, the partition's small scans) run BESIDE the
bucket accumulation if a few CUs are taken away from the accumulation on purpose -- hipExtStreamCreateWithCUMask -- instead of hoping
for free wave slots (side streams starve behind msm_accumulate_kernel, with or without stream priorities)?

Proxy built from the library's own entry points, which take a hipStream_t: the BIG call is a 2^24-point G1 multiexp (82 % of it is
msm_accumulate_kernel), the SMALL calls are 2^12-point multiexps (0.5 ms each: a chain of ~25 dependent launches of a few hundred waves,
the shape of a reduce tail).  Measured: each alone on an unmasked stream, each alone on its masked stream (what the mask costs / what
R CUs are worth to the chain), and both at once -- unmasked streams (the starvation baseline) and masked streams.

  python tools/exp_cu_mask.py [--reserve 8] [--big-log-n 24] [--small-log-n 12]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import inputs  # noqa: E402
import phase2_bn254_amd as zk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", type=int, default=8)
    ap.add_argument("--big-log-n", type=int, default=24)
    ap.add_argument("--small-log-n", type=int, default=12)
    ap.add_argument("--small-calls", type=int, default=20)
    args = ap.parse_args()
    L = zk.lib.load()
    zk.Worker(0)
    dev = torch.device("cuda", 0)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count

    def masked_stream(bits):
        words = (n_cu + 31) // 32
        m = (C.c_uint32 * words)()
        for b in bits:
            m[b // 32] |= 1 << (b % 32)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, m)
        assert rc == 0, rc
        return st

    def plain_stream():
        st = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0  # hipStreamNonBlocking
        return st

    def make(log_n, seed):
        n = 1 << log_n
        k = bench.gen_scalars(n, seed, dev)
        b = torch.empty((n, 8), dtype=torch.int64, device=dev)
        gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
        assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(b.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
        return b, bench.gen_scalars(n, seed + 1, dev), n

    big = make(args.big_log_n, 11)
    small = make(args.small_log_n, 21)
    torch.cuda.synchronize()

    def msm(inp, st):
        b, s, n = inp
        out = np.zeros(12, dtype=np.uint64)
        rc = L.mi355zk_bn254_g1_msm_dev(C.c_void_p(b.data_ptr()), n, 0, C.c_void_p(s.data_ptr()), n, None, 0, st, out.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        aff = np.zeros(8, dtype=np.uint64)   # (the Jacobian representative depends on which join path a call took: compare affine)
        L.mi355zk_bn254_g1_to_affine(aff.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return aff

    def time_big(st, reps=3):
        msm(big, st)
        t = time.perf_counter()
        for _ in range(reps):
            r = msm(big, st)
        return (time.perf_counter() - t) / reps * 1e3, r

    def time_small(st, calls):
        msm(small, st)
        t = time.perf_counter()
        for _ in range(calls):
            r = msm(small, st)
        return (time.perf_counter() - t) / calls * 1e3, r

    def both(st_big, st_small):
        res = {}

        def a():
            res["big_ms"], res["big_r"] = time_big(st_big, reps=2)

        def b():
            time.sleep(0.004)  # let the big call reach its accumulation
            res["small_ms"], res["small_r"] = time_small(st_small, args.small_calls)

        ta, tb = threading.Thread(target=a), threading.Thread(target=b)
        ta.start(); tb.start(); ta.join(); tb.join()
        return res

    out = {"n_cu": n_cu, "reserve": args.reserve, "big_log_n": args.big_log_n, "small_log_n": args.small_log_n}
    s_all, s_all2 = plain_stream(), plain_stream()
    t_big, ref_big = time_big(s_all)
    t_small, ref_small = time_small(s_all2, args.small_calls)
    out["alone_unmasked"] = {"big_ms": round(t_big, 3), "small_ms_per_call": round(t_small, 4)}
    r = both(s_all, s_all2)
    out["together_unmasked"] = {"big_ms": round(r["big_ms"], 3), "small_ms_per_call": round(r["small_ms"], 4)}
    assert np.array_equal(r["big_r"], ref_big) and np.array_equal(r["small_r"], ref_small)
    for name, reserved in (("low_bits", list(range(args.reserve))),
                           ("strided", [i * (n_cu // args.reserve) for i in range(args.reserve)])):
        rest = [i for i in range(n_cu) if i not in reserved]
        s_main, s_side = masked_stream(rest), masked_stream(reserved)
        tb, rb = time_big(s_main)
        ts, rs = time_small(s_side, args.small_calls)
        assert np.array_equal(rb, ref_big) and np.array_equal(rs, ref_small)
        r = both(s_main, s_side)
        assert np.array_equal(r["big_r"], ref_big) and np.array_equal(r["small_r"], ref_small)
        out["mask_" + name] = {"reserved_cus": reserved, "alone": {"big_ms_on_%d_cus" % len(rest): round(tb, 3), "small_ms_per_call_on_%d_cus" % len(reserved): round(ts, 4)},
                               "together": {"big_ms": round(r["big_ms"], 3), "small_ms_per_call": round(r["small_ms"], 4)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
