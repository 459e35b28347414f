"""The Rust side of the boundary, machine-checked as far as an image without rustc allows:
  - integration/mi355zk.rs's `extern "C"` block is the one tools/gen_rust_ffi.py derives from include/mi355zk.h, and -- parsed back
    independently -- every item has the name, the argument count and the pointer / integer shape of its C prototype;
  - the symbols the hand-written shim code calls exist in that block;
  - integration/bellman_mi355zk.patch carries exactly that file as bellman/src/mi355zk.rs and applies (`patch --dry-run -p1`) to a copy
    of the reference's bellman/ tree (multiexp.rs:330-355, source.rs:21-34,72-118, domain.rs:263-272).  That last check needs
    /root/reference and is skipped where it is absent (the GPU box)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as gen  # noqa: E402

RS = os.path.join(ROOT, "integration", "mi355zk.rs")
PATCH = os.path.join(ROOT, "integration", "bellman_mi355zk.patch")
REF_BELLMAN = "/root/reference/bellman"


def _rust_items():
    """{name: ([argument types], return type or None)} of the extern block, parsed from the Rust text (not from the generator)"""
    src = open(RS).read()
    m = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S)
    assert m, "no extern block"
    items = {}
    for fm in re.finditer(r"pub fn (\w+)\((.*?)\)( -> ([^;]+))?;", m.group(1), flags=re.S):
        args = re.sub(r"/\*.*?\*/", "", fm.group(2))
        types = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        items[fm.group(1)] = (types, fm.group(4).strip() if fm.group(4) else None)
    return items


def test_extern_block_is_generated_from_the_header():
    assert gen.main.__module__ == "gen_rust_ffi"
    src = open(RS).read()
    i, j = src.index(gen.BEGIN), src.index(gen.END) + len(gen.END)
    assert src[i:j] == gen.block(), "integration/mi355zk.rs is stale: run tools/gen_rust_ffi.py"


def test_every_extern_item_matches_its_c_prototype():
    items = _rust_items()
    protos = {name: (ret, params) for ret, name, params in gen.prototypes()}
    public = [n for n in protos if not n.startswith(("mi355zk_selftest_", "mi355zk_ubench_"))]
    assert sorted(items) == sorted(public)
    for name, (types, ret) in items.items():
        c_ret, c_params = protos[name]
        assert len(types) == len(c_params), name
        assert (ret is None) == (c_ret == "void"), name
        for rt, (ct, _) in zip(types, c_params):
            c_is_ptr = "*" in ct or "[" in ct
            assert rt.startswith(("*const", "*mut")) == c_is_ptr, (name, rt, ct)
            if c_is_ptr:
                assert rt.startswith("*const") == ct.startswith("const ") or ct.replace(" ", "") == "void*const*", (name, rt, ct)
            else:
                assert rt == {"int": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "long long": "c_longlong"}[ct], (name, rt, ct)


def test_the_shim_only_calls_declared_symbols():
    src = open(RS).read()
    body = src[src.index(gen.END):]
    called = set(re.findall(r"\b(mi355zk_\w+)\s*\(", body))
    assert {"mi355zk_bn254_g1_msm", "mi355zk_bn254_g2_msm", "mi355zk_bn254_fr_ntt", "mi355zk_bn254_g1_to_affine", "mi355zk_bn254_g2_to_affine"} <= called
    assert called <= set(_rust_items())


def test_the_patch_carries_this_file():
    patch = open(PATCH).read()
    m = re.search(r"^\+\+\+ b/src/mi355zk\.rs\n@@ -0,0 \+1,(\d+) @@\n((?:\+.*\n)+)", patch, flags=re.M)
    assert m, "the patch does not add src/mi355zk.rs"
    added = "".join(ln[1:] + "\n" for ln in m.group(2).splitlines())
    assert added == open(RS).read(), "integration/bellman_mi355zk.patch is stale: run integration/make_patch.py"
    for f in ("Cargo.toml", "src/lib.rs", "src/source.rs", "src/multiexp.rs", "src/domain.rs"):
        assert "+++ b/%s\n" % f in patch, f
    assert "try_multiexp::<Q, D, G, S>" in patch and "try_best_fft::<E, T>" in patch


@pytest.mark.skipif(not os.path.isdir(REF_BELLMAN) or shutil.which("patch") is None, reason="needs /root/reference/bellman and patch(1)")
def test_the_patch_applies_to_the_reference_tree(tmp_path):
    work = tmp_path / "bellman"
    shutil.copytree(REF_BELLMAN, work)
    r = subprocess.run(["patch", "--dry-run", "-p1", "-i", PATCH], cwd=work, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    r = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (work / "src" / "mi355zk.rs").read_text() == open(RS).read()
    src = (work / "src" / "multiexp.rs").read_text()
    # the early return sits between the query-size assertion and the reference's own call (multiexp.rs:347-354)
    assert src.index("assert!(query_size == exponents.len());") < src.index("crate::mi355zk::try_multiexp") < src.index("multiexp_inner_impl(pool, bases, density_map, exponents, 0, c, true)")
