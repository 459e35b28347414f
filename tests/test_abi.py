"""The C-ABI library loads and exports every symbol include/mi355zk.h declares (no GPU needed), and the
pure-host entry points work without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mi355zk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355zk_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
        assert n in zk.lib.SIGNATURES, f"{n} declared in the header but not bound in lib.py"
    assert sorted(zk.lib.SIGNATURES) == names
    assert lib.mi355zk_version().startswith(b"mi355zk")


def test_bad_arguments_are_rejected_without_a_device():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    out = np.zeros(12, np.uint64)
    assert lib.mi355zk_bn254_g1_msm(None, 4, 0, None, 4, None, 0, out.ctypes.data_as(C.c_void_p)) == zk.lib.ERR_BAD_ARGS
    assert lib.mi355zk_bn254_fr_fft(None, 4) == zk.lib.ERR_BAD_ARGS
    a = np.zeros((2, 4), np.uint64)
    assert lib.mi355zk_bn254_fr_fft(a.ctypes.data_as(C.c_void_p), 29) == zk.lib.ERR_BAD_ARGS  # > Fr::S (domain.rs:75-77)
    assert lib.mi355zk_bn254_fr_domain_op(a.ctypes.data_as(C.c_void_p), 1, 7) == zk.lib.ERR_BAD_ARGS


def test_domain_constants_host_arithmetic_matches_oracle():
    """mi355zk_bn254_fr_domain_constants runs the library's own host-side Fr code (the same source as the
    kernels, compiled for the host): omega / omegainv / geninv / minv of from_coeffs (domain.rs:84-98)."""
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    for log_n in (0, 1, 10, 20, 28):
        outs = [np.zeros(4, np.uint64) for _ in range(4)]
        assert lib.mi355zk_bn254_fr_domain_constants(log_n, *[o.ctypes.data_as(C.c_void_p) for o in outs]) == 0
        for got, want in zip(outs, O.fr_domain(log_n)):
            assert np.array_equal(got, want)
    assert lib.mi355zk_bn254_fr_domain_constants(29, None, None, None, None) == zk.lib.ERR_BAD_ARGS


def test_host_group_helpers_match_oracle():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    for G, add, to_aff, gen in ((O.G1, lib.mi355zk_bn254_g1_add, lib.mi355zk_bn254_g1_to_affine, inputs.G1_GEN_RAW),
                                (O.G2, lib.mi355zk_bn254_g2_add, lib.mi355zk_bn254_g2_to_affine, inputs.G2_GEN_RAW)):
        a = G.mul(G.from_affine(gen), M.to_limbs(12345))
        b = G.mul(G.from_affine(gen), M.to_limbs(M.R_ORDER - 7))
        zero = G.from_affine(np.zeros(G.aff, np.uint64))
        for x, y in ((a, b), (a, a), (a, zero), (zero, b), (a, G.mul(G.from_affine(gen), M.to_limbs(M.R_ORDER - 12345)))):
            acc = x.copy()
            assert add(acc.ctypes.data_as(C.c_void_p), np.ascontiguousarray(y).ctypes.data_as(C.c_void_p)) == 0
            aff = np.zeros(G.aff, np.uint64)
            assert to_aff(aff.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(aff, G.to_affine(G.add(x, y)))


def test_host_scalar_mul_matches_oracle():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    for G, mul, to_aff, gen in ((O.G1, lib.mi355zk_bn254_g1_mul, lib.mi355zk_bn254_g1_to_affine, inputs.G1_GEN_RAW),
                                (O.G2, lib.mi355zk_bn254_g2_mul, lib.mi355zk_bn254_g2_to_affine, inputs.G2_GEN_RAW)):
        p = G.mul(G.from_affine(gen), M.to_limbs(987654321))
        zero = G.from_affine(np.zeros(G.aff, np.uint64))
        for base in (p, zero):
            for k in (0, 1, 2, 3, 0xFFFFFFFFFFFFFFFF, 1 << 64, M.R_ORDER - 1, M.R_ORDER, 0x2B5F3A1C9E7D46820F1E2D3C4B5A69788796A5B4C3D2E1F0123456789ABCDEF % M.R_ORDER):
                acc = base.copy()
                kk = np.array(M.to_limbs(k), dtype=np.uint64)
                assert mul(acc.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p)) == 0
                aff = np.zeros(G.aff, np.uint64)
                assert to_aff(aff.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(aff, G.to_affine(G.mul(base, kk))), (G, k)


def test_window_geometry_is_sane():
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    for n in (1, 100, 1 << 16, 1 << 20, 1 << 26):
        nw = C.c_int()
        c = lib.mi355zk_msm_window_bits(n, C.byref(nw))
        # c = bits of the bucket field: nb <= 2^c - 1 slots per window, digits in base B = 2 nb (a power of two or 3 / 5 times
        # one), so 2^(c-1) <= B <= 2^(c+1): the windows must cover 254 bits without a spare one
        assert 2 <= c <= 24 and nw.value * (c + 1) >= 254 and (nw.value - 1) * (c - 1) < 254 + c


def test_table_geometry_is_sane_and_table_entries_reject_bad_arguments():
    """mi355zk_msm_table_geometry is host arithmetic: power-of-two windows that cover the 254 bits of an exponent, a table index
    (n_windows * n_bases) that fits the 31 bits of an index-list entry up to 2^27 points; the table entry points refuse null
    pointers and unknown flags before they touch a device."""
    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    for group in (1, 2):
        prev_w = 64
        for lg in (0, 4, 10, 16, 18, 19, 20, 22, 23, 24, 26, 27):
            c, w = C.c_uint32(), C.c_uint32()
            assert lib.mi355zk_msm_table_geometry(1 << lg, group, C.byref(c), C.byref(w)) == 0
            assert 4 <= c.value <= 24
            assert (w.value - 1) * c.value + (c.value - 1) >= 254 > (w.value - 2) * c.value + (c.value - 1)   # the smallest window count that covers 254 bits
            assert w.value <= prev_w   # longer vectors never take more windows
            prev_w = w.value
            assert (w.value << lg) < (1 << 31)
    assert lib.mi355zk_msm_table_geometry(16, 3, None, None) == zk.lib.ERR_BAD_ARGS
    out = np.zeros(12, np.uint64)
    assert lib.mi355zk_bn254_g1_msm_table_dev(None, 4, 0, None, 4, None, 0, 0, None, out.ctypes.data_as(C.c_void_p)) == zk.lib.ERR_BAD_ARGS
    assert lib.mi355zk_bn254_g1_msm_table_dev(None, 0, 0, None, 0, None, 0, 2, None, out.ctypes.data_as(C.c_void_p)) == zk.lib.ERR_BAD_ARGS   # unknown flag
    assert lib.mi355zk_bn254_g2_msm_table_build_dev(None, 4, None, 0, None) == zk.lib.ERR_BAD_ARGS
    assert lib.mi355zk_bases_cache_pin_tables(None, 4, 1) == zk.lib.ERR_BAD_ARGS
    d, t = C.c_size_t(1), C.c_size_t(1)
    assert lib.mi355zk_bases_cache_info(out.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(t)) == 0 and d.value == 0 and t.value == 0


def test_device_set_queries_work_without_a_gpu():
    """mi355zk_visible_devices / mi355zk_device_count are plain queries (no device needed); a device set naming a GPU that is not
    there is refused (bad arguments or a device error, never accepted)."""
    import ctypes as C

    import phase2_bn254_amd as zk

    lib = zk.lib.load()
    n = lib.mi355zk_visible_devices()
    assert n >= 0
    assert lib.mi355zk_device_count() >= 1
    ids = (C.c_int * 2)(0, 1 << 20)
    assert lib.mi355zk_init(ids, 2) != 0
    assert lib.mi355zk_init(None, -1) != 0


def test_header_is_plain_c99(tmp_path):
    """include/mi355zk.h is what a C / cgo / bindgen consumer includes: it must compile as C99 with warnings on, without HIP or C++."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "mi355zk.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + inc, "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
