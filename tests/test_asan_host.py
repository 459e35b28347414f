"""The sanitizer build of the library (`make asan`: AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side of every translation unit,
include/mi355zk.h's pools, caches, planners and the host arithmetic the kernels share -- field.hpp, fieldu.hpp, curveu.hpp, glv.hpp, the digit
extraction, the host join) under the CPU suite: the host-side test files run once more in a child process that loads
tools/bin/libmi355zk_asan.so with the ASan runtime preloaded and halt_on_error set, so any report -- out-of-bounds, use-after-free,
signed overflow, misaligned access, a shift past the width -- fails the child.  tests/test_gpu_ubsan.py does the same with device work (UBSan only: see there).
Skipped (with the reason) where the sanitizer library has not been built: __graft_entry__.build() builds it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_SO = os.path.join(ROOT, "tools", "bin", "libmi355zk_asan.so")
HOST_TESTS = ["tests/test_abi.py", "tests/test_uform_host.py", "tests/test_glv_host.py", "tests/test_msm_digits_host.py", "tests/test_host_logic.py"]


@pytest.mark.skipif(not os.path.exists(ASAN_SO), reason="tools/bin/libmi355zk_asan.so not built (make asan / __graft_entry__.build())")
def test_host_side_tests_pass_under_asan_and_ubsan():
    probe = ("import os, sys; sys.path.insert(0, %r); import phase2_bn254_amd as zk; zk.lib.load(); m = open('/proc/self/maps').read(); "
             "assert 'libmi355zk_asan.so' in m and 'libclang_rt.asan' in m, 'the sanitizer build is not what got loaded'; print('asan-loaded')" % ROOT)
    out = subprocess.run([os.path.join(ROOT, "tools", "run_asan.sh"), sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan-loaded" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    out = subprocess.run([os.path.join(ROOT, "tools", "run_asan.sh"), sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + HOST_TESTS,
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in out.stdout
