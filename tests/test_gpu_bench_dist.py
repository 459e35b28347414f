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


def _run(n_ranks, bases, log_n=21):
    args = ["bench.py", "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--log-n", str(log_n), "--no-cpu-baseline", "--no-h2d-leg", "--no-secondary",
            "--bases", bases]
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n_ranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
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
        # the preflight all-gather (before the input generation) and the N x 1 plan timed next to the default one, same point
        assert got["preflight"]["ranks"] == n and got["preflight"]["record_bytes"] == 112
        assert got["alt_plan"]["same_point_as_the_timed_plan"] and got["alt_plan"]["points_per_gpu"] == (1 << 21) // n
        assert got["config"]["bases_bytes_per_gpu"] * (n // min(n, 4)) == (1 << 21) * 64


def test_eight_ranks_at_the_baseline_size():
    """BASELINE config 4 at ITS size with the N = 8 control flow: 8 gloo ranks share the one device (each holds its 2^25-point
    range: 2 GiB of bases, 1 GiB of exponents and its workspace -- 288 GB of HBM hold all eight), every rank runs its (point
    range x window group) cell of the (2 x 4) plan at the real geometry, the partials go through the exchange step, and the
    sum is the group element the single-GPU bench line reports for the same seed-per-shard input (x limb 0x5c2b6af288baf266:
    BENCH_r01 / r02 / r03).  bench.py itself checks the sharded result against the unsharded evaluation and full-size linearity."""
    got = _run(8, "random", log_n=26)
    assert got["n_gpus"] == 8 and got["full_size_linearity_check"] and got["sharded_result_matches_unsharded"]
    assert got["result_affine_x_limb0"] == "0x5c2b6af288baf266"
    assert got["config"]["points_per_gpu"] == 1 << 25 and "2 point range(s) x 4 window group(s)" in got["config"]["parallelism"]
    assert got["alt_plan"]["same_point_as_the_timed_plan"] and got["alt_plan"]["points_per_gpu"] == 1 << 23 and "8 point range(s) x 1" in got["alt_plan"]["parallelism"]


def test_plain_command_launches_its_own_ranks():
    """`python3 bench.py --gpus 4 ...` exactly as typed (no torchrun, WORLD_SIZE unset): bench.py starts the ranks itself and rank 0
    prints the one JSON line -- so the first multi-GPU lease produces a scaling curve instead of an argument error."""
    args = [sys.executable, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--log-n", "21", "--no-cpu-baseline", "--no-h2d-leg", "--no-secondary"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got["n_gpus"] == 4 and got["rccl_ranks"] == 4 and got["backend"] == "gloo" and got["sharded_result_matches_unsharded"]
    ref = _run(1, "random")
    assert got["result_affine_x_limb0"] == ref["result_affine_x_limb0"]
    # without the gloo escape hatch a box with fewer GPUs than ranks is refused with a message, not a crash inside RCCL
    import torch

    if torch.cuda.device_count() < 4:
        env.pop("BENCH_BACKEND")
        out = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 2 and "GPU(s) visible" in out.stderr


def test_single_process_multi_gpu_leg_of_the_bench():
    """When bench.py's process sees several GPUs its secondary block also times the single-process multi-GPU mode of the C ABI
    (mi355zk_init with n > 1); BENCH_MULTI_LOGICAL=4 exercises that leg with four logical devices on the one GPU there is: same point."""
    args = [sys.executable, "bench.py", "--log-n", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-h2d-leg"]
    env = dict(os.environ, BENCH_MULTI_LOGICAL="4")
    out = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    leg = got["secondary"]["single_process_multi_gpu_2e24"]
    assert [len(r["devices"]) for r in leg["runs"]] == [1, 2, 4] and all(r["same_point_as_device_resident_call"] for r in leg["runs"])
