"""ctypes loader for libmi355zk.so (the hand-written HIP library, C ABI in include/mi355zk.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device is
visible when a compute entry point is called, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MI355ZK_SO: another build of the same library, for same-box A/B measurements of a kernel variant -- tools/, never the tests)
SO_PATH = os.environ.get("MI355ZK_SO") or os.path.join(_HERE, "libmi355zk.so")

OK, ERR_UNEXPECTED_IDENTITY, ERR_UNEXPECTED_EOF, ERR_BAD_ARGS, ERR_DEVICE = 0, 1, 2, 3, -1
OP_FFT, OP_IFFT, OP_COSET_FFT, OP_ICOSET_FFT = 0, 1, 2, 3
MSM_SCALARS_MONTGOMERY = 1
ABI_VERSION = 6   # MI355ZK_ABI_VERSION of the include/mi355zk.h this table was written against: load() refuses another library
EXP_SAME_SCALAR, FFT_INVERSE, G2_TRUSTED_SUBGROUP = 1, 1, 2   # mode / flag bits of batch_exp, point_fft, sparse_matvec (include/mi355zk.h)

_vp, _sz, _i, _u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
_u64p, _u32p, _u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)

# every symbol include/mi355zk.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mi355zk_init": (_i, [C.POINTER(C.c_int), _i]),
    "mi355zk_device_count": (_i, []),
    "mi355zk_visible_devices": (_i, []),
    "mi355zk_shutdown": (None, []),
    "mi355zk_version": (C.c_char_p, []),
    "mi355zk_abi_version": (_i, []),
    "mi355zk_bases_cache_pin": (_i, [_vp, _sz, _i]),
    "mi355zk_bases_cache_pin_tables": (_i, [_vp, _sz, _i]),
    "mi355zk_bases_cache_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mi355zk_bases_cache_invalidate": (None, [_vp]),
    "mi355zk_bn254_g1_msm": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp]),
    "mi355zk_bn254_g2_msm": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp]),
    "mi355zk_bn254_g1_msm_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g2_msm_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g1_msm_part_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _u32, _vp, _vp]),
    "mi355zk_bn254_g2_msm_part_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _u32, _vp, _vp]),
    "mi355zk_bn254_g1_msm_ex_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _u32, _u32, _vp, _vp]),
    "mi355zk_bn254_g2_msm_ex_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _u32, _u32, _vp, _vp]),
    "mi355zk_msm_table_geometry": (_i, [_sz, _i, C.POINTER(_u32), C.POINTER(_u32)]),
    "mi355zk_bn254_g1_msm_table_build_dev": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "mi355zk_bn254_g2_msm_table_build_dev": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "mi355zk_bn254_g1_msm_table_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _vp, _vp]),
    "mi355zk_bn254_g2_msm_table_dev": (_i, [_vp, _sz, _sz, _vp, _sz, _vp, _sz, _u32, _vp, _vp]),
    "mi355zk_bn254_g1_dense_multiexp": (_i, [_vp, _vp, _sz, _vp]),
    "mi355zk_bn254_g2_dense_multiexp": (_i, [_vp, _vp, _sz, _vp]),
    "mi355zk_bn254_g1_merge_pairs": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g2_merge_pairs": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g1_dense_multiexp_dev": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g2_dense_multiexp_dev": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "mi355zk_bn254_g1_merge_pairs_dev": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "mi355zk_bn254_g2_merge_pairs_dev": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "mi355zk_last_error_index": (C.c_longlong, []),
    "mi355zk_msm_window_bits": (_i, [_sz, C.POINTER(C.c_int)]),
    "mi355zk_msm_window_bits_groups": (_i, [_sz, _u32, C.POINTER(C.c_int)]),
    "mi355zk_bn254_fr_ntt": (_i, [_vp, _u32, _vp]),
    "mi355zk_bn254_fr_domain_op": (_i, [_vp, _u32, _i]),
    "mi355zk_bn254_fr_fft": (_i, [_vp, _u32]),
    "mi355zk_bn254_fr_ifft": (_i, [_vp, _u32]),
    "mi355zk_bn254_fr_coset_fft": (_i, [_vp, _u32]),
    "mi355zk_bn254_fr_icoset_fft": (_i, [_vp, _u32]),
    "mi355zk_bn254_fr_ntt_dev": (_i, [_vp, _u32, _vp, _vp]),
    "mi355zk_bn254_fr_ntt_scaled_dev": (_i, [_vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "mi355zk_bn254_fr_domain_op_dev": (_i, [_vp, _u32, _i, _vp]),
    "mi355zk_bn254_fr_domain_op_batch_dev": (_i, [C.POINTER(C.c_void_p), _u32, _u32, _i, _vp]),
    "mi355zk_bn254_fr_domain_constants": (_i, [_u32, _vp, _vp, _vp, _vp]),
    "mi355zk_bn254_fr_mul_assign_dev": (_i, [_vp, _vp, _sz, _vp]),
    "mi355zk_bn254_fr_sub_assign_dev": (_i, [_vp, _vp, _sz, _vp]),
    "mi355zk_bn254_fr_into_repr_dev": (_i, [_vp, _vp, _sz, _vp]),
    "mi355zk_bn254_fr_divide_by_z_on_coset_dev": (_i, [_vp, _u32, _vp]),
    "mi355zk_bn254_fr_domain_z": (_i, [_u32, _vp, _vp]),
    "mi355zk_ubench_fp_mul": (_i, [_i, _u32, _u32, _vp, _vp, _vp, C.POINTER(C.c_float)]),
    "mi355zk_selftest_g1_record_sum": (_i, [_i, _vp, _vp, _vp, _sz, _sz, _vp]),
    "mi355zk_selftest_g2_record_sum": (_i, [_i, _vp, _vp, _vp, _sz, _sz, _vp]),
    "mi355zk_selftest_msm_digits": (_i, [_sz, _u32, _vp, _u32, _u32, _i, _vp, _vp]),
    "mi355zk_selftest_glv_split": (_i, [_vp, _vp]),
    "mi355zk_selftest_glv_wnaf5": (_i, [_vp, _vp]),
    "mi355zk_selftest_glv2_split": (_i, [_vp, _vp]),
    "mi355zk_selftest_g2_psi": (_i, [_vp, _vp]),
    "mi355zk_selftest_u_mul": (_i, [_i, _vp, _vp, _vp]),
    "mi355zk_selftest_u_mul_shoup": (_i, [_i, _vp, _vp, _vp, _vp]),
    "mi355zk_selftest_u_sub": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "mi355zk_selftest_u_pack": (_i, [_i, _vp, _vp, _vp, _vp]),
    "mi355zk_selftest_u_reduce32": (_i, [_i, _vp, _vp]),
    "mi355zk_selftest_g1_accumulate": (_i, [_i, _vp, _vp, _sz, _vp]),
    "mi355zk_selftest_g2_scalar_mul_u": (_i, [_vp, _vp, _vp]),
    "mi355zk_selftest_g2_accumulate": (_i, [_i, _vp, _vp, _sz, _vp]),
    "mi355zk_bn254_g1_sparse_matvec": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _i]),
    "mi355zk_bn254_g2_sparse_matvec": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _i]),
    "mi355zk_bn254_g1_sparse_matvec_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _vp, _i]),
    "mi355zk_bn254_g2_sparse_matvec_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _vp, _i]),
    "mi355zk_bn254_g1_decode_dev": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    "mi355zk_bn254_g2_decode_dev": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    "mi355zk_bn254_g1_encode_dev": (_i, [_vp, _vp, _sz, _i, _vp]),
    "mi355zk_bn254_g2_encode_dev": (_i, [_vp, _vp, _sz, _i, _vp]),
    "mi355zk_bn254_g1_point_fft_dev": (_i, [_vp, _u32, _i, _vp]),
    "mi355zk_bn254_g2_point_fft_dev": (_i, [_vp, _u32, _i, _vp]),
    "mi355zk_bn254_g1_batch_mul_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "mi355zk_bn254_g2_batch_mul_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "mi355zk_bn254_g1_batch_exp": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mi355zk_bn254_g2_batch_exp": (_i, [_vp, _vp, _vp, _sz, _i]),
    "mi355zk_bn254_g1_batch_exp_dev": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "mi355zk_bn254_g2_batch_exp_dev": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "mi355zk_bn254_g2_subgroup_check_dev": (_i, [_vp, _sz, _vp, C.POINTER(C.c_longlong)]),
    "mi355zk_selftest_g2_in_subgroup": (_i, [_vp]),
    "mi355zk_bn254_g1_add": (_i, [_vp, _vp]),
    "mi355zk_bn254_g2_add": (_i, [_vp, _vp]),
    "mi355zk_bn254_g1_to_affine": (_i, [_vp, _vp]),
    "mi355zk_bn254_g2_to_affine": (_i, [_vp, _vp]),
    "mi355zk_bn254_g1_mul": (_i, [_vp, _vp]),
    "mi355zk_bn254_g2_mul": (_i, [_vp, _vp]),
    "mi355zk_malloc": (_i, [C.POINTER(_vp), _sz]),
    "mi355zk_free": (_i, [_vp]),
    "mi355zk_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "mi355zk_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "mi355zk_sync": (_i, [_vp]),
    "mi355zk_prof_enable": (None, [_i]),
    "mi355zk_prof_only": (None, [C.c_char_p]),
    "mi355zk_prof_reset": (None, []),
    "mi355zk_prof_get": (_i, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libmi355zk.so and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7)
        # and must be loaded FIRST so that this library's NEEDED libamdhip64.so.7 resolves to the copy
        # torch already mapped; the other order maps two runtimes and the second sees no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: run `make` (or __graft_entry__.build()) -- there is no CPU fallback")
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.mi355zk_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{SO_PATH}: ABI revision {lib.mi355zk_abi_version()}, this binding was written against {ABI_VERSION} "
                               "(argument meanings differ between revisions: include/mi355zk.h)")
        _lib = lib
    return _lib
