//! bellman/src/mi355zk.rs -- binds `libmi355zk.so` (C ABI: include/mi355zk.h of the mi355zk repository) under bellman_ce's two hot
//! functions, `multiexp::multiexp` (multiexp.rs:330-355) and `domain::best_fft` (domain.rs:263-272).
//!
//! Added by integration/bellman_mi355zk.patch together with five small hunks (Cargo.toml: the `mi355zk` feature; lib.rs: this module;
//! source.rs: two defaulted trait methods; multiexp.rs / domain.rs: one early return each).  With the feature off nothing changes; with it
//! on, BN254 G1 / G2 multiexps over `(Arc<Vec<G>>, usize)` sources and BN254 `Scalar` FFTs run on the MI355X and every other engine, source
//! or element type falls through to the existing Rust path.  Return codes: 0 ok; 1 / 2 / 3 are the reference's own errors (`map_err`); a
//! negative code is a DEVICE failure and means "fall through to the CPU path" -- a broken GPU never changes a result.
//!
//! This image has no rustc: the file is kept honest mechanically instead -- the `extern "C"` block below is GENERATED from the header
//! (tools/gen_rust_ffi.py; tests/test_integration_patch.py checks it is current, that every item matches its C prototype in name and
//! argument count, and that the patch applies to the reference's bellman/ tree).
#![allow(dead_code)]

use std::any::TypeId;
use std::io::{self, Write};
use std::mem;
use std::os::raw::{c_char, c_int, c_long, c_longlong, c_void};
use std::ptr;
use std::sync::Arc;

use futures::{future, Future};

use crate::pairing::bn256::{Bn256, Fq, FqRepr, G1Affine, G1Uncompressed, G2Affine, G2Uncompressed, G1, G2};
use crate::pairing::ff::{PrimeField, PrimeFieldRepr, ScalarEngine};
use crate::pairing::{CurveAffine, CurveProjective, EncodedPoint, Engine, RawEncodable};

use crate::group::Group;
use crate::source::{QueryDensity, SourceBuilder};
use crate::SynthesisError;

/// MI355ZK_ABI_VERSION of the header the block below was generated from; `abi_ok()` compares it with the loaded library's.
pub const MI355ZK_ABI_VERSION: c_int = 6;
// mode / flag bits (include/mi355zk.h)
pub const MI355ZK_MSM_SCALARS_MONTGOMERY: u32 = 1; // msm_ex_dev / msm_table_dev `flags`: the exponents are Montgomery Fr (prover.rs:89-129 fused)
pub const MI355ZK_EXP_SAME_SCALAR: c_int = 1; // batch_exp `mode`: ONE scalar for all points (phase2 contribute)
pub const MI355ZK_FFT_INVERSE: c_int = 1; // point_fft `mode`: omega^-1 and the 1/m scaling
pub const MI355ZK_G2_TRUSTED_SUBGROUP: c_int = 2; // batch_exp / point_fft / sparse_matvec: the caller's promise

// ---- BEGIN GENERATED (tools/gen_rust_ffi.py from include/mi355zk.h) ----
#[link(name = "mi355zk")]
extern "C" {
    pub fn mi355zk_init(device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn mi355zk_device_count() -> c_int;
    pub fn mi355zk_visible_devices() -> c_int;
    pub fn mi355zk_shutdown();
    pub fn mi355zk_version() -> *const c_char;
    pub fn mi355zk_abi_version() -> c_int;
    pub fn mi355zk_bn254_g1_msm(bases: *const u8, n_bases: usize, base_offset: usize, scalars: *const u64, n_scalars: usize, density: *const u32, density_bits: usize, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bases_cache_pin(host_bases: *const c_void, n_bases: usize, group: c_int) -> c_int;
    pub fn mi355zk_bases_cache_pin_tables(host_bases: *const c_void, n_bases: usize, group: c_int) -> c_int;
    pub fn mi355zk_bases_cache_invalidate(host_bases: *const c_void);
    pub fn mi355zk_bases_cache_info(host_bases: *const c_void, device_bytes: *mut usize, table_bytes: *mut usize) -> c_int;
    pub fn mi355zk_bn254_g2_msm(bases: *const u8, n_bases: usize, base_offset: usize, scalars: *const u64, n_scalars: usize, density: *const u32, density_bits: usize, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_msm_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, stream: *mut c_void, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_msm_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, stream: *mut c_void, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_dense_multiexp_dev(d_bases: *const c_void, d_scalars: *const c_void, n: usize, stream: *mut c_void, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_dense_multiexp_dev(d_bases: *const c_void, d_scalars: *const c_void, n: usize, stream: *mut c_void, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_merge_pairs_dev(d_v1: *const c_void, d_v2: *const c_void, d_rho: *const c_void, n: usize, stream: *mut c_void, out_s: *mut u64 /* [12] */, out_sx: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_merge_pairs_dev(d_v1: *const c_void, d_v2: *const c_void, d_rho: *const c_void, n: usize, stream: *mut c_void, out_s: *mut u64 /* [24] */, out_sx: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_dense_multiexp(bases: *const u8, scalars: *const u64, n: usize, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_dense_multiexp(bases: *const u8, scalars: *const u64, n: usize, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_merge_pairs(v1: *const u8, v2: *const u8, rho: *const u64, n: usize, out_s: *mut u64 /* [12] */, out_sx: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_merge_pairs(v1: *const u8, v2: *const u8, rho: *const u64, n: usize, out_s: *mut u64 /* [24] */, out_sx: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_msm_part_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, window_groups: u32, window_group: u32, stream: *mut c_void, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_msm_part_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, window_groups: u32, window_group: u32, stream: *mut c_void, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_msm_ex_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, flags: u32, window_groups: u32, window_group: u32, stream: *mut c_void, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_msm_ex_dev(d_bases: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, flags: u32, window_groups: u32, window_group: u32, stream: *mut c_void, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_msm_table_geometry(n_bases: usize, group: c_int, window_bits: *mut u32, n_windows: *mut u32) -> c_int;
    pub fn mi355zk_bn254_g1_msm_table_build_dev(d_bases: *const c_void, n_bases: usize, d_table: *mut c_void, table_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g2_msm_table_build_dev(d_bases: *const c_void, n_bases: usize, d_table: *mut c_void, table_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g1_msm_table_dev(d_table: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, flags: u32, stream: *mut c_void, out_xyz: *mut u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_msm_table_dev(d_table: *const c_void, n_bases: usize, base_offset: usize, d_scalars: *const c_void, n_scalars: usize, density: *const u32, density_bits: usize, flags: u32, stream: *mut c_void, out_xyz: *mut u64 /* [24] */) -> c_int;
    pub fn mi355zk_last_error_index() -> c_longlong;
    pub fn mi355zk_msm_window_bits(n_scalars: usize, n_windows: *mut c_int) -> c_int;
    pub fn mi355zk_msm_window_bits_groups(n_scalars: usize, window_groups: u32, n_windows: *mut c_int) -> c_int;
    pub fn mi355zk_bn254_fr_ntt(a: *mut u64, log_n: u32, omega: *const u64 /* [4] */) -> c_int;
    pub fn mi355zk_bn254_fr_domain_op(a: *mut u64, log_n: u32, op: c_int) -> c_int;
    pub fn mi355zk_bn254_fr_fft(a: *mut u64, log_n: u32) -> c_int;
    pub fn mi355zk_bn254_fr_ifft(a: *mut u64, log_n: u32) -> c_int;
    pub fn mi355zk_bn254_fr_coset_fft(a: *mut u64, log_n: u32) -> c_int;
    pub fn mi355zk_bn254_fr_icoset_fft(a: *mut u64, log_n: u32) -> c_int;
    pub fn mi355zk_bn254_fr_ntt_dev(d_a: *mut c_void, log_n: u32, omega: *const u64 /* [4] */, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_ntt_scaled_dev(d_a: *mut c_void, log_n: u32, omega: *const u64 /* [4] */, pre_g: *const u64 /* [4] */, post_c: *const u64 /* [4] */, post_g: *const u64 /* [4] */, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_domain_op_dev(d_a: *mut c_void, log_n: u32, op: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_domain_op_batch_dev(d_arrays: *const *mut c_void, batch: u32, log_n: u32, op: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_domain_constants(log_n: u32, omega: *mut u64 /* [4] */, omegainv: *mut u64 /* [4] */, geninv: *mut u64 /* [4] */, minv: *mut u64 /* [4] */) -> c_int;
    pub fn mi355zk_bn254_fr_mul_assign_dev(d_a: *mut c_void, d_b: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_sub_assign_dev(d_a: *mut c_void, d_b: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_into_repr_dev(d_out: *mut c_void, d_in: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_divide_by_z_on_coset_dev(d_a: *mut c_void, log_n: u32, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_fr_domain_z(log_n: u32, tau: *const u64 /* [4] */, out: *mut u64 /* [4] */) -> c_int;
    pub fn mi355zk_bn254_g1_sparse_matvec_dev(d_out_affine: *mut c_void, d_bases_affine: *const c_void, n_bases: usize, d_row_ptr: *const u32, d_col: *const u32, d_coeffs: *const c_void, n_rows: usize, nnz: usize, stream: *mut c_void, flags: c_int) -> c_int;
    pub fn mi355zk_bn254_g2_sparse_matvec_dev(d_out_affine: *mut c_void, d_bases_affine: *const c_void, n_bases: usize, d_row_ptr: *const u32, d_col: *const u32, d_coeffs: *const c_void, n_rows: usize, nnz: usize, stream: *mut c_void, flags: c_int) -> c_int;
    pub fn mi355zk_bn254_g1_sparse_matvec(out_affine: *mut u8, bases_affine: *const u8, n_bases: usize, row_ptr: *const u32, col: *const u32, coeffs: *const u64, n_rows: usize, nnz: usize, flags: c_int) -> c_int;
    pub fn mi355zk_bn254_g2_sparse_matvec(out_affine: *mut u8, bases_affine: *const u8, n_bases: usize, row_ptr: *const u32, col: *const u32, coeffs: *const u64, n_rows: usize, nnz: usize, flags: c_int) -> c_int;
    pub fn mi355zk_bn254_g1_decode_dev(d_out_affine: *mut c_void, d_in_bytes: *const c_void, n: usize, compressed: c_int, checked: c_int, stream: *mut c_void, err_index: *mut c_longlong) -> c_int;
    pub fn mi355zk_bn254_g2_decode_dev(d_out_affine: *mut c_void, d_in_bytes: *const c_void, n: usize, compressed: c_int, checked: c_int, stream: *mut c_void, err_index: *mut c_longlong) -> c_int;
    pub fn mi355zk_bn254_g1_encode_dev(d_out_bytes: *mut c_void, d_in_affine: *const c_void, n: usize, compressed: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g2_encode_dev(d_out_bytes: *mut c_void, d_in_affine: *const c_void, n: usize, compressed: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g1_point_fft_dev(d_points_affine: *mut c_void, log_n: u32, mode: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g2_point_fft_dev(d_points_affine: *mut c_void, log_n: u32, mode: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g1_batch_mul_dev(d_out_affine: *mut c_void, base_affine: *const u64 /* [8] */, d_scalars: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g2_batch_mul_dev(d_out_affine: *mut c_void, base_affine: *const u64 /* [16] */, d_scalars: *const c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g1_batch_exp_dev(d_out_affine: *mut c_void, d_bases_affine: *const c_void, d_scalars: *const c_void, n: usize, mode: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g2_batch_exp_dev(d_out_affine: *mut c_void, d_bases_affine: *const c_void, d_scalars: *const c_void, n: usize, mode: c_int, stream: *mut c_void) -> c_int;
    pub fn mi355zk_bn254_g1_batch_exp(out_affine: *mut u8, bases_affine: *const u8, scalars: *const u64, n: usize, mode: c_int) -> c_int;
    pub fn mi355zk_bn254_g2_batch_exp(out_affine: *mut u8, bases_affine: *const u8, scalars: *const u64, n: usize, mode: c_int) -> c_int;
    pub fn mi355zk_bn254_g2_subgroup_check_dev(d_points_affine: *const c_void, n: usize, stream: *mut c_void, bad_index: *mut c_longlong) -> c_int;
    pub fn mi355zk_bn254_g1_add(acc_xyz: *mut u64 /* [12] */, other_xyz: *const u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_add(acc_xyz: *mut u64 /* [24] */, other_xyz: *const u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_to_affine(out_xy: *mut u64 /* [8] */, xyz: *const u64 /* [12] */) -> c_int;
    pub fn mi355zk_bn254_g2_to_affine(out_xy: *mut u64 /* [16] */, xyz: *const u64 /* [24] */) -> c_int;
    pub fn mi355zk_bn254_g1_mul(acc_xyz: *mut u64 /* [12] */, scalar: *const u64 /* [4] */) -> c_int;
    pub fn mi355zk_bn254_g2_mul(acc_xyz: *mut u64 /* [24] */, scalar: *const u64 /* [4] */) -> c_int;
    pub fn mi355zk_malloc(d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mi355zk_free(d_ptr: *mut c_void) -> c_int;
    pub fn mi355zk_memcpy_h2d(d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn mi355zk_memcpy_d2h(h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn mi355zk_sync(stream: *mut c_void) -> c_int;
    pub fn mi355zk_prof_enable(on: c_int);
    pub fn mi355zk_prof_only(kernel: *const c_char);
    pub fn mi355zk_prof_reset();
    pub fn mi355zk_prof_get(kernel: *const c_char, total_ms: *mut f64, count: *mut c_long) -> c_int;
}
// ---- END GENERATED ----

/// rc > 0 are the reference's own errors (source.rs:44-70); rc < 0 is a device failure: `None` = take the CPU path.
pub fn map_err(rc: c_int) -> Option<SynthesisError> {
    match rc {
        1 => Some(SynthesisError::UnexpectedIdentity), // source.rs:50-52
        2 => Some(io::Error::new(io::ErrorKind::UnexpectedEof, "expected more bases when adding from source").into()), // source.rs:46-48
        3 => Some(SynthesisError::IoError(io::Error::new(io::ErrorKind::InvalidInput, "mi355zk: bad arguments"))),
        _ => None,
    }
}

/// The library speaks the revision this file was generated against (argument MEANINGS changed between revisions -- batch_exp's and
/// point_fft's `mode`, sparse_matvec's `flags` -- so a mismatch is a refusal, i.e. the CPU path, never a guess).  Checked once.
pub fn abi_ok() -> bool {
    use std::sync::Once;
    static CHECK: Once = Once::new();
    static mut OK: bool = false;
    unsafe {
        CHECK.call_once(|| OK = mi355zk_abi_version() == MI355ZK_ABI_VERSION);
        OK
    }
}

/// `A` and `B` are the same type (checked): the generic `G` / `G::Projective` of `multiexp` seen as the concrete bn256 type and back.
fn same_type_copy<A: 'static, B: 'static + Copy>(a: &A) -> B {
    assert!(TypeId::of::<A>() == TypeId::of::<B>());
    unsafe { mem::transmute_copy::<A, B>(a) }
}

/// n x 64 B raw records x || y, Montgomery limbs little-endian (RawEncodable::into_raw_uncompressed_le, ec.rs:653-664).  That encoder
/// ignores the `infinity` flag (it writes x = 0, y = R) while the library -- like the reference's raw DECODER, ec.rs:673-675 -- treats
/// exactly the all-zero record as the point at infinity: identities are left zeroed here.
pub fn g1_records(v: &[G1Affine]) -> Vec<u8> {
    let mut raw = vec![0u8; v.len() * 64];
    for (p, dst) in v.iter().zip(raw.chunks_mut(64)) {
        if !p.is_zero() {
            dst.copy_from_slice(p.into_raw_uncompressed_le().as_ref());
        }
    }
    raw
}

/// n x 128 B raw records x.c0 || x.c1 || y.c0 || y.c1 (G2Affine has no RawEncodable impl, pairing/src/lib.rs:236-246 is implemented for
/// G1Affine only: the records are built from get_x() / get_y(), ec.rs:88-94).
pub fn g2_records(v: &[G2Affine]) -> Vec<u8> {
    let mut raw = vec![0u8; v.len() * 128];
    for (p, dst) in v.iter().zip(raw.chunks_mut(128)) {
        if p.is_zero() {
            continue;
        }
        let (x, y) = (p.get_x(), p.get_y());
        let mut w = &mut dst[..];
        for c in [x.c0, x.c1, y.c0, y.c1].iter() {
            c.into_raw_repr().write_le(&mut w).unwrap();
        }
    }
    raw
}

/// Jacobian X, Y, Z (Montgomery limbs; Z = 0 is the identity, ec.rs:229-245) -> the crate's projective type.  The coordinate fields of
/// `G1` are private to pairing_ce, so the point is normalised by the library's host helper (one inversion) and enters through the raw
/// decoder; `PartialEq` of the projective types compares points, not representatives (ec.rs:45-85).
fn g1_from_jacobian(xyz: &[u64]) -> G1 {
    let mut aff = [0u64; 8];
    unsafe { mi355zk_bn254_g1_to_affine(aff.as_mut_ptr(), xyz.as_ptr()) };
    if aff.iter().all(|l| *l == 0) {
        return G1::zero();
    }
    let mut enc = G1Uncompressed::empty();
    {
        let mut w = enc.as_mut();
        for l in aff.iter() {
            w.write_all(&l.to_le_bytes()).unwrap();
        }
    }
    G1Affine::from_raw_uncompressed_le_unchecked(&enc, false).expect("reduced coordinates").into_projective()
}

fn g2_from_jacobian(xyz: &[u64]) -> G2 {
    let mut aff = [0u64; 16];
    unsafe { mi355zk_bn254_g2_to_affine(aff.as_mut_ptr(), xyz.as_ptr()) };
    if aff.iter().all(|l| *l == 0) {
        return G2::zero();
    }
    // G2Uncompressed is x.c1 || x.c0 || y.c1 || y.c0, canonical big-endian (ec.rs:1154-1200); the records are c0 || c1, Montgomery
    let mut enc = G2Uncompressed::empty();
    {
        let mut w = enc.as_mut();
        for i in [1usize, 0, 3, 2].iter() {
            let fe = Fq::from_raw_repr(FqRepr([aff[4 * i], aff[4 * i + 1], aff[4 * i + 2], aff[4 * i + 3]])).expect("reduced coordinate");
            fe.into_repr().write_be(&mut w).unwrap();
        }
    }
    enc.into_affine_unchecked().expect("a point the library returned").into_projective()
}

/// The body of `multiexp` (multiexp.rs:330-355) on the device.  `None`: not a BN254 curve, not a contiguous source, an opaque density
/// map, or a device failure -- the caller continues into `multiexp_inner_impl` unchanged.  A ready future is what
/// `singlecore::Worker::compute` already returns (singlecore.rs:33-47); the library is re-entrant, so the prover's eight calls before
/// the first `wait()` (prover.rs:250-298) may come from any threads.
pub fn try_multiexp<Q, D, G, S>(
    bases: &S,
    density_map: &D,
    exponents: &Arc<Vec<<<G::Engine as ScalarEngine>::Fr as PrimeField>::Repr>>,
) -> Option<Box<dyn Future<Item = <G as CurveAffine>::Projective, Error = SynthesisError>>>
where
    for<'a> &'a Q: QueryDensity,
    D: AsRef<Q>,
    G: CurveAffine,
    S: SourceBuilder<G>,
{
    let is_g1 = TypeId::of::<G>() == TypeId::of::<G1Affine>();
    if (!is_g1 && TypeId::of::<G>() != TypeId::of::<G2Affine>()) || !abi_ok() {
        return None;
    }
    let (slice, offset) = bases.as_contiguous()?;
    let (words, bits) = density_map.as_ref().density_words()?;
    let density = if words.is_empty() { ptr::null() } else { words.as_ptr() }; // FullDensity: NULL
    // FrRepr is `pub struct FrRepr(pub [u64; 4])`: the exponent vector IS the ABI's scalar array
    let scalars = exponents.as_ptr() as *const u64;
    let mut out = [0u64; 24];
    let rc = if is_g1 {
        let raw = g1_records(unsafe { &*(slice as *const [G] as *const [G1Affine]) });
        unsafe { mi355zk_bn254_g1_msm(raw.as_ptr(), slice.len(), offset, scalars, exponents.len(), density, bits, out.as_mut_ptr()) }
    } else {
        let raw = g2_records(unsafe { &*(slice as *const [G] as *const [G2Affine]) });
        unsafe { mi355zk_bn254_g2_msm(raw.as_ptr(), slice.len(), offset, scalars, exponents.len(), density, bits, out.as_mut_ptr()) }
    };
    if rc == 0 {
        let p: <G as CurveAffine>::Projective = if is_g1 { same_type_copy(&g1_from_jacobian(&out)) } else { same_type_copy(&g2_from_jacobian(&out)) };
        return Some(Box::new(future::ok::<_, SynthesisError>(p)));
    }
    map_err(rc).map(|e| Box::new(future::err::<<G as CurveAffine>::Projective, _>(e)) as Box<dyn Future<Item = <G as CurveAffine>::Projective, Error = SynthesisError>>)
}

/// `best_fft` (domain.rs:263-272) for `T = Scalar<Bn256>` (group.rs:53: a transparent wrapper around the four Montgomery limbs of an
/// `Fr`): in place on the device copy, written back only on success -- `false` leaves `a` untouched for serial_fft / parallel_fft.
/// `Point<G>` (96 / 192 B) and other engines return `false` at once.
pub fn try_best_fft<E: Engine, T: Group<E>>(a: &mut [T], omega: &E::Fr, log_n: u32) -> bool {
    if TypeId::of::<E>() != TypeId::of::<Bn256>() || mem::size_of::<T>() != 32 || log_n > 28 || a.len() != 1usize << log_n || !abi_ok() {
        return false;
    }
    unsafe { mi355zk_bn254_fr_ntt(a.as_mut_ptr() as *mut u64, log_n, omega as *const E::Fr as *const u64) == 0 }
}

/// Once per process, e.g. from `Worker::new()` (multicore.rs:24-35): every visible GPU becomes part of the device set, and host-buffer
/// multiexps of >= 2^20 exponents are cut into one point range per device.  Optional: without it the library
/// runs on the process's current HIP device (device 0).
pub fn init_all_devices() -> bool {
    let n = unsafe { mi355zk_visible_devices() };
    if n <= 0 {
        return false;
    }
    let ids: Vec<c_int> = (0..n).collect();
    unsafe { mi355zk_init(ids.as_ptr(), n) == 0 }
}
