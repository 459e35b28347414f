"""Multi-GPU sharding of the multiexp: one process per GPU, contiguous point ranges,
(or, shard.plan, a few point ranges x groups of scalar windows: mi355zk_bn254_g*_msm_part_dev),
ONE exchange step -- an all-gather of the Jacobian partial sums (96 B for G1, 192 B for G2) over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests) followed
by world-1 group additions on every rank.

A literal all-reduce(SUM) of limbs is not the group law, so the north-star's "all-reduce of partial
bucket sums" is realised as all-gather + local EC adds.  The reference has no counterpart (it is a
single process): the join is the same `add_assign` that joins windows in multiexp.rs:146-154.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    """[start, end) of rank's contiguous slice of n points; remainders go to the first ranks."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def plan(world: int) -> tuple[int, int]:
    """(point_groups, window_groups) with point_groups * window_groups == world.  Splitting by scalar WINDOWS first keeps every
    rank on the large-n window geometry (fewer windows per scalar than a 1/world-size point range would get) at the price of
    holding more bases per GPU; the window count of a geometry divides by at most 4 in practice (12 windows), beyond that the
    points are split as well: 1 -> (1, 1), 2 -> (1, 2), 4 -> (1, 4), 8 -> (2, 4), 16 -> (4, 4); other sizes -> (world, 1)."""
    if world >= 1 and world & (world - 1) == 0:
        wg = min(world, 4)
        return world // wg, wg
    return world, 1


def rank_groups(world: int, rank: int) -> tuple[int, int, int, int]:
    """(point_groups, point_group, window_groups, window_group) of a rank: ranks that share a point range are adjacent."""
    pg, wg = plan(world)
    return pg, rank // wg, wg, rank % wg


def density_base_offsets(density_bits, world: int) -> list[int]:
    """For a density map, the index of the first base each rank's slice consumes (prefix popcount):
    bases are compacted, so slice d starts at the number of set bits before its first exponent."""
    bits = np.asarray(density_bits, dtype=np.uint8)
    csum = np.concatenate([[0], np.cumsum(bits)])
    return [int(csum[shard_range(len(bits), world, r)[0]]) for r in range(world)]


def join_partials(partials: np.ndarray) -> np.ndarray:
    """sum of (world, 12|24) u64 Jacobian points with the library's host-side add_assign."""
    partials = np.ascontiguousarray(partials, dtype=np.uint64)
    group = {12: 1, 24: 2}[partials.shape[1]]
    add = _lib.load().mi355zk_bn254_g1_add if group == 1 else _lib.load().mi355zk_bn254_g2_add
    acc = partials[0].copy()
    for r in range(1, partials.shape[0]):
        rc = add(acc.ctypes.data_as(C.c_void_p), np.ascontiguousarray(partials[r]).ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"mi355zk add failed rc={rc}")
    return acc


_RC = {"UnexpectedIdentity": 1, "IoError(UnexpectedEof)": 2}

_XCHG = {}  # (device, words, world) -> (pinned send, device send, device recv, pinned recv): allocated once, reused by every step


def _allgather_words(rec: np.ndarray, device, group, world: int) -> np.ndarray:
    """all-gather of one small u64 record per rank -> (world, words).  device given (RCCL): the record goes through a PINNED
    host buffer and the three transfers (H2D, all-gather, D2H) are queued on the current stream without a host wait in
    between -- ONE synchronisation per exchange.  The Jacobian partial lives on the host on both sides of the exchange (the
    window join before it and the EC additions after it are host arithmetic), so the device only relays it over xGMI."""
    import torch
    import torch.distributed as dist

    words = rec.size
    if device is None:
        mine = torch.from_numpy(rec.view(np.int64))
        gathered = torch.empty(world * words, dtype=torch.int64)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        return gathered.numpy().view(np.uint64).reshape(world, -1)
    key = (str(device), words, world)
    if key not in _XCHG:
        _XCHG[key] = (torch.empty(words, dtype=torch.int64, pin_memory=True), torch.empty(words, dtype=torch.int64, device=device),
                      torch.empty(world * words, dtype=torch.int64, device=device), torch.empty(world * words, dtype=torch.int64, pin_memory=True))
    h_send, d_send, d_recv, h_recv = _XCHG[key]
    h_send.numpy()[:] = rec.view(np.int64)
    d_send.copy_(h_send, non_blocking=True)
    dist.all_gather_into_tensor(d_recv, d_send, group=group)
    h_recv.copy_(d_recv, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return h_recv.numpy().view(np.uint64).reshape(world, -1).copy()


def exchange(fut, limbs: int, index_offset: int = 0, device=None, group=None) -> np.ndarray:
    """The exchange step WITH the error path: `fut` is this rank's ready future (bellman.multiexp(...)) of a `limbs`-word
    Jacobian partial (12 for G1, 24 for G2).  Every rank contributes (partial, rc, GLOBAL exponent index = local index +
    index_offset); after the all-gather every rank either returns the same total or raises the SAME SynthesisError -- the one
    at the lowest exponent index, Eof before identity at one index, which is what the unsharded call reports
    (oracle/tmpl_multiexp.h).  Without this a failing rank would raise before the collective and leave the others blocked in it,
    and a window-group partial alone is not a complete identity check (a base is only looked at by the groups whose windows hold
    a non-zero digit of its exponent: the minimum over the groups is)."""
    import torch
    import torch.distributed as dist

    from .bellman import SynthesisError

    rec = np.zeros(limbs + 2, dtype=np.uint64)
    local_exc = None
    try:
        rec[:limbs] = np.ascontiguousarray(fut.wait(), dtype=np.uint64)
    except SynthesisError as e:
        rec[limbs] = _RC[e.kind]
        rec[limbs + 1] = np.uint64(max(e.index, 0) + index_offset)
    except Exception as e:  # noqa: BLE001  (device failure / bad arguments: every rank must still reach the collective)
        rec[limbs] = 3
        local_exc = e
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        allr = rec.reshape(1, -1)
    else:
        allr = _allgather_words(rec, device, group, world)
    rcs = allr[:, limbs].astype(np.int64)
    if (rcs == 3).any():
        raise local_exc if local_exc is not None else RuntimeError("mi355zk: a peer rank failed inside its multiexp")
    bad = [(int(allr[r, limbs + 1]), 0 if rcs[r] == 2 else 1) for r in range(world) if rcs[r] != 0]
    if bad:
        idx, kind = min(bad)
        raise SynthesisError(SynthesisError.IO_UNEXPECTED_EOF if kind == 0 else SynthesisError.UNEXPECTED_IDENTITY, idx)
    return join_partials(allr[:, :limbs])


def allgather_join(partial: np.ndarray, device=None, group=None) -> np.ndarray:
    """The exchange step: every rank contributes its Jacobian partial, every rank gets the total."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return np.ascontiguousarray(partial, dtype=np.uint64)
    return join_partials(_allgather_words(np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1), device, group, world))


def batch_exp_sharded(bases, exps, same_scalar: bool = False, gather: bool = True, group=None, fn=None):
    """BASELINE config 5's "1 vs 8 GPU": the per-point scalar multiplications of phase2 `contribute` (parameters.rs:423-470, every
    point of L and H by delta^-1) and of powersoftau `batch_exp` (batched_accumulator.rs:1130-1181) are independent, so they shard
    by CONTIGUOUS POINT RANGE with no exchange at all: rank r transforms bases[lo:hi) (shard_range).  `bases` / `exps` are the FULL
    vectors on every rank (or anything sliceable: a rank only touches its range); the local result comes back, or -- gather=True,
    what a rank that writes the whole parameter file needs -- the concatenation of all ranges through ONE all-gather of the affine
    records (ranges padded to equal length).  fn: the local kernel, default ceremony.batch_exp (the gloo CPU test passes a stand-in)."""
    import torch
    import torch.distributed as dist

    from . import ceremony

    fn = fn or ceremony.batch_exp
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = bases.shape[0]
    lo, hi = shard_range(n, world, rank)
    local = fn(bases[lo:hi].contiguous(), exps if same_scalar else exps[lo:hi].contiguous(), same_scalar)
    if not gather or world == 1:
        return local
    longest = shard_range(n, world, 0)[1]                      # the first ranks hold the remainders: rank 0 is a longest range
    pad = torch.zeros((longest, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:hi - lo] = local
    allp = torch.empty((world * longest, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(allp, pad, group=group)
    parts = [allp[r * longest:r * longest + (shard_range(n, world, r)[1] - shard_range(n, world, r)[0])] for r in range(world)]
    return torch.cat(parts)
