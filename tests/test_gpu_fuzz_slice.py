"""A fixed-seed slice of the differential fuzzers under the driver: tools/fuzz_msm.py (plain, mixed-radix windows with the
streamed call cut into 64-exponent chunks, table mode, three logical devices, one-lane reduce tails), tools/fuzz_ntt.py (the default
wave-local pass with folded tables up to 2^22; the barrier-per-stage-pair kernel with the scale factors as separate products under a table
budget of 32 MiB, so that every large transform evicts) and tools/fuzz_rows.py (batch_exp / sparse matvec / merge_pairs / point FFT / codecs).
Every case is compared with the CPU oracle, bit exact.  Each configuration is its own process -- the window layout and the NTT planner
read their environment once -- and the processes run side by side on the one GPU (which also makes them a concurrency test of the
library's per-device state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = [
    ("msm default layout", ["tools/fuzz_msm.py", "--cases", "40", "--seed", "611"], {}),
    ("msm mixed radix 3,6 + streamed chunks of 64", ["tools/fuzz_msm.py", "--cases", "36", "--seed", "612"], {"MI355ZK_MSM_RADIX": "3,6", "MI355ZK_HOST_CHUNK_TEST": "64"}),
    ("msm c = 11, one-lane reduce tails", ["tools/fuzz_msm.py", "--cases", "36", "--seed", "613"], {"MI355ZK_MSM_C": "11", "MI355ZK_MSM_QUAD": "0"}),
    ("msm table mode, table c = 7", ["tools/fuzz_msm.py", "--table", "--cases", "30", "--seed", "614"], {"MI355ZK_MSM_TABLE_C": "7"}),
    ("msm over 3 logical devices", ["tools/fuzz_msm.py", "--devices", "3", "--cases", "36", "--seed", "615"], {}),
    ("ntt default (wave-local, folded tables) to 2^22", ["tools/fuzz_ntt.py", "--cases", "90", "--seed", "616", "--max-log", "22"], {}),
    ("ntt barrier kernel, no fold, 32 MiB table budget", ["tools/fuzz_ntt.py", "--cases", "70", "--seed", "617", "--max-log", "21"],
     {"MI355ZK_NTT_WAVELOCAL": "0", "MI355ZK_NTT_NO_FOLD": "1", "MI355ZK_NTT_TABLES_GB": "0.03"}),
    ("ntt folded tables under a 32 MiB table budget", ["tools/fuzz_ntt.py", "--cases", "60", "--seed", "618", "--max-log", "21"], {"MI355ZK_NTT_TABLES_GB": "0.03"}),
    ("rows", ["tools/fuzz_rows.py", "--cases", "40", "--seed", "619"], {}),
    ("rows over 3 logical devices", ["tools/fuzz_rows.py", "--cases", "24", "--seed", "620", "--devices", "3"], {}),
]


def test_fixed_seed_fuzz_slice_matches_the_oracle(zk, worker):
    procs = []
    for name, argv, env in CONFIGS:
        e = dict(os.environ)
        for k in [k for k in e if k.startswith("MI355ZK_")]:
            del e[k]
        e.update(env)
        procs.append((name, subprocess.Popen([sys.executable] + argv, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failures = []
    for name, p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            failures.append((name, "TIMEOUT\n" + out[-800:]))
            continue
        if p.returncode != 0 or "MISMATCH" in out:
            failures.append((name, "rc %d\n%s" % (p.returncode, out[-1200:])))
    assert not failures, "\n\n".join("%s: %s" % f for f in failures)
