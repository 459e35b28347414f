"""bench.py's N > 1 control flow on ONE GPU: N gloo ranks share the device (BENCH_BACKEND=gloo), every rank takes its
(point range x window group) cell, the 96-byte partials go through the exchange step (shard.exchange: all-gather + error
record + join).  The input is generated in seed-per-shard blocks, so the result must be the SAME group element for every N;
bench.py itself checks the timed sharded result against the unsharded evaluation and full-size linearity.
(2^21 points: the input is generated in 2^20-point shards, and N = 8 splits the points into two ranges.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(n_ranks, bases):
    args = ["bench.py", "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--log-n", "21", "--no-cpu-baseline", "--no-h2d-leg", "--bases", bases]
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n_ranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("bases", ["random", "tau"])
def test_bench_result_is_identical_for_every_rank_count(bases):
    ref = _run(1, bases)
    assert ref["full_size_linearity_check"] and ref["n_gpus"] == 1
    for n in (2, 4, 8) if bases == "random" else (8,):
        got = _run(n, bases)
        assert got["n_gpus"] == n and got["full_size_linearity_check"] and got["sharded_result_matches_unsharded"]
        assert got["result_affine_x_limb0"] == ref["result_affine_x_limb0"], (n, bases)
        assert got["config"]["bases_bytes_per_gpu"] * (n // min(n, 4)) == (1 << 21) * 64
