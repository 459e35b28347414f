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


def _worker_plan(rank, world, port, n, q):
    """world = 4 / 8 under shard.plan (1 x 4, 2 x 4): the cell of a rank is (point range of its point group) x (its window group).  The
    oracle has no window-group partial, so window group 0 of every point range contributes the range's whole sum and the other groups the
    identity -- the join is a sum of `world` partials whatever they mean, which is what is under test, together with the error rule."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import inputs
    import oracle_lib as O
    import phase2_bn254_amd as zk

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = O.G1
    bases = inputs.bases_progression_cpu(1, n, seed=41)
    scalars = inputs.random_scalars(n, seed=42)
    pg, pgi, wg, wgi = zk.shard.rank_groups(world, rank)
    assert pg * wg == world and rank == pgi * wg + wgi
    lo, hi = zk.shard.shard_range(n, pg, pgi)
    if wgi == 0:
        rc, part = G.multiexp(bases[lo:hi], scalars[lo:hi])
        assert rc == 0
    else:
        part = G.from_affine(np.zeros(8, np.uint64))
    rc_full, full = G.multiexp(bases, scalars)
    ok = bool(np.array_equal(G.to_affine(zk.shard.exchange(zk.bellman._Ready(part), 12, index_offset=lo)), G.to_affine(full)))
    ok = ok and bool(np.array_equal(G.to_affine(zk.shard.allgather_join(part)), G.to_affine(full)))

    def err(kind, idx):
        return zk.bellman._Ready(error=zk.SynthesisError(kind, idx))

    ID, EOF = zk.SynthesisError.UNEXPECTED_IDENTITY, zk.SynthesisError.IO_UNEXPECTED_EOF
    lo_of = lambda r: zk.shard.shard_range(n, pg, zk.shard.rank_groups(world, r)[1])[0]  # noqa: E731
    # (a) three ranks fail at different places: everybody raises the error at the lowest GLOBAL exponent index
    failing = {1: (ID, 3), 2: (ID, 0), world - 1: (EOF, 1)}
    want = min((lo_of(r) + i, 0 if k == EOF else 1, k) for r, (k, i) in failing.items())
    try:
        zk.shard.exchange(err(*failing[rank]) if rank in failing else zk.bellman._Ready(part), 12, index_offset=lo)
        ok = False
    except zk.SynthesisError as e:
        ok = ok and (e.index, e.kind) == (want[0], want[2])
    # (b) two ranks of ONE point range report the same global index, one Eof and one identity: Eof wins (oracle/tmpl_multiexp.h)
    failing = {0: (ID, 7), wg - 1: (EOF, 7)} if wg > 1 else {0: (EOF, 7)}
    try:
        zk.shard.exchange(err(*failing[rank]) if rank in failing else zk.bellman._Ready(part), 12, index_offset=lo)
        ok = False
    except zk.SynthesisError as e:
        ok = ok and (e.index, e.kind) == (7, EOF)
    # (c) one rank's multiexp dies with something that is not a SynthesisError (a device failure): every rank still reaches the collective
    # and every rank raises -- nobody is left waiting
    class Boom:
        def wait(self):
            raise RuntimeError("device lost")

    try:
        zk.shard.exchange(Boom() if rank == world - 2 else zk.bellman._Ready(part), 12, index_offset=lo)
        ok = False
    except RuntimeError:
        pass
    # ... and the group is still usable afterwards
    ok = ok and bool(np.array_equal(G.to_affine(zk.shard.exchange(zk.bellman._Ready(part), 12, index_offset=lo)), G.to_affine(full)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_plan_cells_and_error_order_with_4_and_8_ranks(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plan, args=(r, world, port, 203, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(r, True) for r in range(world)]
