"""The host side of the library under UndefinedBehaviorSanitizer WITH device work: a slice of the GPU suite -- the multiexp entry points
with their error paths, the streamed and multi-device host-buffer calls (copy threads, pools, caches, per-stream scratch: api.hip +
host_entry.hip), several host threads at once, the prover's eight calls in flight, the ceremony rows and the NTT table cache -- runs once more in a child
process over tools/bin/libmi355zk_ubsan.so (`make ubsan`: every translation unit's host code instrumented, -fno-sanitize-recover, so the first
signed overflow / misaligned access / out-of-range shift / bad enum aborts the child).  The AddressSanitizer build covers the host-only paths in the
CPU suite (tests/test_asan_host.py); it cannot run here: ROCm's ASan runtime intercepts the HSA allocator (Makefile, profiles/r06_asan_on_gpu.txt).
Skipped where the sanitizer library has not been built (__graft_entry__.build() builds it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UBSAN_SO = os.path.join(ROOT, "tools", "bin", "libmi355zk_ubsan.so")
SLICES = [
    ["tests/test_gpu_msm.py", "-k", "golden or error_index or density or concurrent or repeated or over_long or montgomery or matches_oracle or streamed or host_entry or heavy or empty"],
    ["tests/test_gpu_multi_device.py", "tests/test_gpu_prover.py", "tests/test_gpu_ntt.py::test_domain_ops_match_oracle", "tests/test_gpu_ntt.py::test_batched_domain_ops_match_oracle",
     "tests/test_gpu_ceremony.py::test_power_pairs_like_the_reference", "tests/test_gpu_ceremony.py::test_eval_qap_and_dense_multiexp",
     "tests/test_gpu_ceremony.py::test_codec_roundtrip_and_error", "tests/test_gpu_ceremony.py::test_contribute_accumulator_like_compute_constrained"],
]


@pytest.mark.skipif(not os.path.exists(UBSAN_SO), reason="tools/bin/libmi355zk_ubsan.so not built (make ubsan / __graft_entry__.build())")
def test_gpu_slice_passes_under_ubsan(zk, worker):
    probe = ("import sys; sys.path.insert(0, %r); import phase2_bn254_amd as zk; zk.lib.load(); m = open('/proc/self/maps').read(); "
             "assert 'libmi355zk_ubsan.so' in m and 'libclang_rt.ubsan' in m, 'the sanitizer build is not what got loaded'; print('ubsan-loaded')" % ROOT)
    out = subprocess.run([os.path.join(ROOT, "tools", "run_ubsan.sh"), sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ubsan-loaded" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    procs = [subprocess.Popen([os.path.join(ROOT, "tools", "run_ubsan.sh"), sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + sl,
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for sl in SLICES]   # (side by side on the one GPU)
    ran = 0
    for p in procs:
        text, _ = p.communicate(timeout=900)
        tail = text[-3000:]
        assert p.returncode == 0, tail
        assert "runtime error:" not in text, tail
        import re

        m = re.search(r"(\d+) passed", text)
        assert m, tail
        ran += int(m.group(1))
    assert ran >= 60, ran
