"""Child process of test_gpu_msm.py::test_pair_and_lane_kernels_agree: the library reads MI355ZK_G1_PAIR / MI355ZK_G2_PAIR once per
process, so each mode (0 = one lane per bucket at every size, 1 = a pair of lanes per bucket at every size) runs in its own
interpreter.  argv[1] = group (1 / 2).  Prints one JSON line of affine results (hex) for a fixed set of multiexps, checked against
the CPU oracle where the oracle is quick."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import inputs
import oracle_lib as O
import phase2_bn254_amd as zk

group = int(sys.argv[1])
G = O.G1 if group == 1 else O.G2
GEN = inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW
worker = zk.Worker(0)
out = {}
for n in (1, 2, 33, 500, 4096):
    bases = inputs.bases_progression_cpu(group, n, seed=n)
    scalars = inputs.random_scalars(n, seed=11 * n + 1)
    rc, want = G.multiexp(bases, scalars, threads=8)
    got = G.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait())
    assert rc == 0 and np.array_equal(got, G.to_affine(want)), n
    out["oracle_%d" % n] = got.tobytes().hex()
# equal and opposite points meeting in one bucket (the doubling / infinity branches of the mixed addition), and an identity base
n = 4096
rng = np.random.default_rng(5)
p_aff = G.mul_many_affine(GEN, inputs.random_scalars(1, seed=99))[0]
bases = np.ascontiguousarray(np.stack([p_aff] * n))
scalars = np.array([[int(v), 0, 0, 0] for v in rng.integers(1, 4, size=n)], dtype=np.uint64)
rc, want = G.multiexp(bases, scalars, threads=8)
got = G.to_affine(zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait())
assert rc == 0 and np.array_equal(got, G.to_affine(want))
out["collide"] = got.tobytes().hex()
bases = inputs.bases_progression_cpu(group, 300, seed=3)
bases[123] = 0
scalars = inputs.random_scalars(300, seed=4)
try:
    zk.multiexp(worker, (bases, 0), zk.FullDensity(), scalars).wait()
    out["identity"] = "no error"
except zk.SynthesisError as e:
    out["identity"] = str(e)
# a streamed host-buffer call (carried buckets) at 2^17 points and a device-resident one at 2^19 (above the automatic gate)
import torch

import bench

dev = torch.device("cuda", 0)
for log_n in (17, 19):
    n = 1 << log_n
    s = bench.gen_scalars(n, 31 + log_n, dev)
    b = inputs.bases_progression_cpu(group, n, seed=log_n)
    bd = torch.from_numpy(b.view(np.int64)).to(dev)
    out["dev_%d" % log_n] = G.to_affine(zk.multiexp(worker, (bd, 0), zk.FullDensity(), s).wait()).tobytes().hex()
os.environ["MI355ZK_HOST_CHUNK_TEST"] = "20000"
n = 1 << 17
b = inputs.bases_progression_cpu(group, n, seed=17)
s = bench.gen_scalars(n, 31 + 17, dev).cpu().numpy().view(np.uint64)
out["host_17"] = G.to_affine(zk.multiexp(worker, (b, 0), zk.FullDensity(), s).wait()).tobytes().hex()
assert out["host_17"] == out["dev_17"], "streamed host call differs from the device-resident one"
print(json.dumps(out))
