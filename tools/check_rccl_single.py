#!/usr/bin/env python3
"""Single-rank sanity check that RCCL (torch.distributed backend "nccl") initialises and runs the collectives bench.py uses
(all_gather_into_tensor, barrier, all_reduce MAX) in the same process as libmi355zk.so -- the multi-GPU path itself is run by
the driver; this only guards against runtime / library conflicts on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np, torch, torch.distributed as dist
import phase2_bn254_amd as zk, inputs
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w = zk.Worker(0)
bases = inputs.bases_progression_cpu(1, 64, seed=1); sc = inputs.random_scalars(64, seed=2)
part = zk.multiexp(w, (torch.from_numpy(bases.view(np.int64)).cuda(), 0), zk.FullDensity(), torch.from_numpy(sc.view(np.int64)).cuda()).wait()
# force the collective path even with one rank
mine = torch.from_numpy(np.ascontiguousarray(part).view(np.int64)).to(dev)
allp = torch.empty(mine.numel(), dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(allp, mine)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", torch.equal(allp, mine), float(t.item()))
dist.destroy_process_group()
