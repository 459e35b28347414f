#!/usr/bin/env python3
"""The single-process multi-GPU mode of the C ABI (mi355zk_init with n_devices > 1) timed end to end: ONE host thread calls
mi355zk_bn254_g1_msm on host buffers, the library cuts the call into one point range per device, streams every range's exponents
over that device's PCIe link and joins the partials on the host.  This is BASELINE.json's metric through the door a Rust ceremony
binary would use (H2D of the scalars INSIDE the call, pinned bases resident on the devices after the first call).

  python tools/bench_multi_device.py [--log-n 26] [--devices 1 2 4 8]

With fewer physical GPUs than a requested count the ids wrap around (logical devices sharing a GPU): a correctness / overhead check
-- the cells then queue on one device -- not a scaling number; "physical_gpus" in the output says which it is."""
import argparse
import ctypes as C
import json
import os
import sys
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
    ap.add_argument("--log-n", type=int, default=26)
    ap.add_argument("--devices", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--no-batch-exp", action="store_true")
    args = ap.parse_args()
    L = zk.lib.load()
    w1 = zk.Worker(0)
    dev = torch.device("cuda", 0)
    n = 1 << args.log_n
    phys = torch.cuda.device_count()
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW)
    bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
    shard = min(n, 1 << 22)
    for s in range(n // shard):
        k = bench.gen_scalars(shard, 7000 + s, dev)
        assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(bases[s * shard:(s + 1) * shard].data_ptr()), gen.ctypes.data_as(C.c_void_p),
                                                C.c_void_p(k.data_ptr()), shard, None) == 0
    scalars = torch.cat([bench.gen_scalars(shard, 8000 + s, dev) for s in range(n // shard)])
    ref = zk.multiexp(w1, (bases, 0), zk.FullDensity(), scalars).wait()
    ref_aff = np.zeros(8, dtype=np.uint64)
    L.mi355zk_bn254_g1_to_affine(ref_aff.ctypes.data_as(C.c_void_p), np.ascontiguousarray(ref).ctypes.data_as(C.c_void_p))
    hb = bases.cpu().numpy().view(np.uint64)
    hs_t = torch.empty(scalars.shape, dtype=torch.int64, pin_memory=True)   # page-locked exponents (what a shim gets from hipHostMalloc)
    hs_t.copy_(scalars)
    hs = hs_t.numpy().view(np.uint64)
    del bases, scalars
    torch.cuda.empty_cache()
    out = {"log_n": args.log_n, "physical_gpus": phys, "runs": []}
    for k in args.devices:
        zk.unpin_bases(None)      # (every device count starts with an empty cache: first_call_ms includes the upload of what the devices keep)
        zk.pin_bases(hb)
        ids = [i % phys for i in range(k)]
        w = zk.Worker(devices=ids) if k > 1 else zk.Worker(0)
        t = time.perf_counter()
        first = zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()      # every device uploads the SLICE of the pinned vector its cell consumes
        t_first = time.perf_counter() - t
        t = time.perf_counter()
        for _ in range(args.iters):
            got = zk.multiexp(w, (hb, 0), zk.FullDensity(), hs).wait()
        dt = (time.perf_counter() - t) / args.iters
        aff = np.zeros(8, dtype=np.uint64)
        L.mi355zk_bn254_g1_to_affine(aff.ctypes.data_as(C.c_void_p), np.ascontiguousarray(got).ctypes.data_as(C.c_void_p))
        cached, tab = C.c_size_t(0), C.c_size_t(0)
        L.mi355zk_bases_cache_info(hb.ctypes.data_as(C.c_void_p), C.byref(cached), C.byref(tab))
        out["runs"].append({"devices": ids, "distinct_gpus": len(set(ids)), "ms_per_call": round(dt * 1e3, 3), "Mscalar_mul_per_s": round(n / dt / 1e6, 1),
                            "first_call_ms": round(t_first * 1e3, 1), "first_call_over_steady": round(t_first / dt, 2),
                            "cached_base_bytes_total": cached.value, "cached_base_bytes_per_logical_device": cached.value // k,
                            "vector_bytes": int(hb.nbytes), "same_point_as_device_resident_call": bool(np.array_equal(aff, ref_aff))})
        del first
    zk.unpin_bases(None)
    if args.no_batch_exp:
        zk.Worker(0)
        print(json.dumps(out))
        return
    # phase2 contribute through the same door: mi355zk_bn254_g1_batch_exp on host buffers, 2^20 points times one scalar
    m = 1 << 20
    pts = hb[:m]
    dinv = np.array([[0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1111111111111111, 0x0222222222222222]], dtype=np.uint64)
    want = None
    out["batch_exp_2e20"] = []
    for k in args.devices:
        ids = [i % phys for i in range(k)]
        zk.Worker(devices=ids) if k > 1 else zk.Worker(0)
        zk.ceremony.batch_exp_host(pts, dinv, same_scalar=True)
        t = time.perf_counter()
        for _ in range(args.iters):
            got = zk.ceremony.batch_exp_host(pts, dinv, same_scalar=True)
        dt = (time.perf_counter() - t) / args.iters
        want = got if want is None else want
        out["batch_exp_2e20"].append({"devices": ids, "ms_per_call_incl_pcie": round(dt * 1e3, 3), "Mpoint_per_s": round(m / dt / 1e6, 1),
                                      "same_records_as_one_device": bool(np.array_equal(got, want))})
    zk.Worker(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
