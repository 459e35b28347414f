"""N>1 path on CPU: world_size-2 gloo run of the exchange step (phase2-bn254_amd/shard.py).
Each rank's partial sum is computed by the ORACLE here (the CPU stand-in for the device multiexp, which
needs a GPU); what is under test is the product's sharding + all-gather + host-side join."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, group, n, with_density, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import golden_util as GU
    import inputs
    import oracle_lib as O
    import phase2_bn254_amd as zk

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = O.G1 if group == 1 else O.G2
    bases = inputs.bases_progression_cpu(group, n, seed=21)
    scalars = inputs.random_scalars(n, seed=22)
    lo, hi = zk.shard.shard_range(n, world, rank)
    if with_density:
        bits = [(i * 7 + 3) % 5 != 0 for i in range(n)]
        off = zk.shard.density_base_offsets(bits, world)[rank]
        rc, part = G.multiexp(bases, scalars[lo:hi], density=GU.density_words(bits[lo:hi]), density_bits=hi - lo, base_offset=off)
        rc_full, full = G.multiexp(bases, scalars, density=GU.density_words(bits), density_bits=n)
    else:
        rc, part = G.multiexp(bases[lo:hi], scalars[lo:hi])
        rc_full, full = G.multiexp(bases, scalars)
    assert rc == 0 and rc_full == 0
    total = zk.shard.allgather_join(part)
    ok = bool(np.array_equal(G.to_affine(total), G.to_affine(full)))
    # the same through the exchange step with the error path; then rank 1 alone fails: BOTH ranks must raise the same error,
    # carrying the GLOBAL exponent index (and nobody may be left waiting in the collective)
    total2 = zk.shard.exchange(zk.bellman._Ready(part), 12 * group, index_offset=lo)
    ok = ok and bool(np.array_equal(G.to_affine(total2), G.to_affine(full)))
    fut = zk.bellman._Ready(part) if rank == 0 else zk.bellman._Ready(error=zk.SynthesisError(zk.SynthesisError.UNEXPECTED_IDENTITY, 5))
    try:
        zk.shard.exchange(fut, 12 * group, index_offset=lo)
        ok = False
    except zk.SynthesisError as e:
        ok = ok and e.kind == zk.SynthesisError.UNEXPECTED_IDENTITY and e.index == zk.shard.shard_range(n, world, 1)[0] + 5
    # Eof on rank 0 at a LOWER global index than an identity on rank 1: Eof wins everywhere
    fut = (zk.bellman._Ready(error=zk.SynthesisError(zk.SynthesisError.IO_UNEXPECTED_EOF, 2)) if rank == 0
           else zk.bellman._Ready(error=zk.SynthesisError(zk.SynthesisError.UNEXPECTED_IDENTITY, 0)))
    try:
        zk.shard.exchange(fut, 12 * group, index_offset=lo)
        ok = False
    except zk.SynthesisError as e:
        ok = ok and e.kind == zk.SynthesisError.IO_UNEXPECTED_EOF and e.index == 2
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("group,n,with_density", [(1, 257, False), (1, 100, True), (2, 33, False)])
def test_two_rank_shard_allgather_join(group, n, with_density):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, group, n, with_density, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, True), (1, True)]


def _worker_batch_exp(rank, world, port, n, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import inputs
    import oracle_lib as O
    import phase2_bn254_amd as zk

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases = inputs.bases_progression_cpu(1, n, seed=31)
    delta_inv = inputs.random_scalars(1, seed=32)

    def stand_in(b, k, same_scalar):  # the oracle as the CPU stand-in of ceremony.batch_exp (which needs a GPU)
        hb = b.numpy().view(np.uint64)
        out = np.stack([O.G1.to_affine(O.G1.mul(O.G1.from_affine(p), k.numpy().view(np.uint64)[0])) for p in hb]) if len(hb) else hb
        return torch.from_numpy(np.ascontiguousarray(out).view(np.int64))

    got = zk.shard.batch_exp_sharded(torch.from_numpy(bases.view(np.int64)), torch.from_numpy(delta_inv.view(np.int64)), same_scalar=True, fn=stand_in)
    want = np.stack([O.G1.to_affine(O.G1.mul(O.G1.from_affine(p), delta_inv[0])) for p in bases])
    q.put((rank, bool(np.array_equal(got.numpy().view(np.uint64), want))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_contribute_by_point_range():
    """config 5's multi-GPU leg (phase2 contribute): point ranges, no exchange but the optional all-gather of the results;
    an odd length, so the ranges differ by one."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batch_exp, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]
