/*
 * mi355zk.h -- C ABI of libmi355zk.so: the MI355X (gfx950) backend for the BN254 MSM / Fr-NTT hot
 * path of kobigurk/phase2-bn254.
 *
 * The reference has no FFI seam; the two Rust functions a maintainer would redirect are
 *   bellman/src/multiexp.rs:330-355   pub fn multiexp(pool, bases, density_map, exponents)
 *   bellman/src/domain.rs:263-272     fn best_fft(a, worker, omega, log_n)
 *   (+ the O(m) scalings of EvaluationDomain::{ifft, coset_fft, icoset_fft}, domain.rs:159-203)
 * INTEGRATION.md shows the Rust `extern "C"` block and the two call-site patches.
 *
 * Conventions (all integers little endian):
 *   Fq / Fr element   = 32 bytes = u64[4], least-significant limb first, MONTGOMERY form x*2^256 mod p,
 *                       fully reduced: the in-memory form of the reference's `Fq`/`Fr`, i.e. what
 *                       `into_raw_repr()` exposes (pairing/src/bn256/ec.rs:659-660).
 *   FrRepr (scalar)   = 32 bytes = u64[4], CANONICAL integer < r: the output of `into_repr()`
 *                       (the element type of multiexp's `exponents`, multiexp.rs:334).
 *   G1 affine base    = 64 bytes  x || y            (RawEncodable layout, ec.rs:653-664)
 *   G2 affine base    = 128 bytes x.c0 || x.c1 || y.c0 || y.c1
 *                       the ALL-ZERO record is the point at infinity (ec.rs:673-675); the shim must
 *                       zero the record itself when `is_zero()` (the reference's raw encoder does not).
 *   Jacobian result   = X || Y || Z (12 / 24 u64), Z == 0 <=> infinity (ec.rs:227-246).  Any
 *                       representative of the group element may be returned (projective equality is
 *                       by value, ec.rs:45-85).
 *   density           = NULL for FullDensity (source.rs:80-99); else bit i of the map is
 *                       (density[i/32] >> (i%32)) & 1, `density_bits` bits long (DensityTracker,
 *                       source.rs:101-118).  Bases are COMPACTED: only set bits consume a base.
 *
 * Return codes:
 *   0  ok
 *   1  UnexpectedIdentity: a selected base with a non-zero exponent is infinity   (source.rs:50-52)
 *   2  UnexpectedEof:      the bases ran out                                      (source.rs:46-48,62-64)
 *   3  bad arguments (NULL pointer, log_n > 28 = PolynomialDegreeTooLarge domain.rs:66-79, sizes >= 2^31, an exponent
 *      >= 2^254, i.e. not a canonical FrRepr: mi355zk_last_error_index() names it)
 *  <0  device failure (details on stderr).  There is NO CPU fallback inside the library.
 * When both error kinds are present the one at the lowest exponent index is reported (the reference's
 * answer depends on thread scheduling there; see oracle/tmpl_multiexp.h).
 *
 * Threading: every entry point may be called concurrently from several host threads (the prover
 * queues 8 multiexps before waiting, bellman/src/groth16/prover.rs:250-298); calls on one device are
 * serialised internally.  Entry points are synchronous: the result is complete on return, which a
 * futures-0.1 shim wraps in `future::result` (what singlecore::Worker::compute does, singlecore.rs:33-47).
 */
#ifndef MI355ZK_H
#define MI355ZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355ZK_OK 0
#define MI355ZK_ERR_UNEXPECTED_IDENTITY 1
#define MI355ZK_ERR_UNEXPECTED_EOF 2
#define MI355ZK_ERR_BAD_ARGS 3
#define MI355ZK_ERR_DEVICE (-1)

/* EvaluationDomain operations (domain.rs:154-203) for mi355zk_bn254_fr_domain_op[_dev] */
#define MI355ZK_OP_FFT 0
#define MI355ZK_OP_IFFT 1
#define MI355ZK_OP_COSET_FFT 2
#define MI355ZK_OP_ICOSET_FFT 3

/* ---- lifecycle.  device_ids == NULL / n_devices == 0: use the process's current HIP device.  The LAST call defines the
 * library's device set.
 *   n_devices == 1: one GPU (device_ids[0] is selected with hipSetDevice) -- one rank per GPU under torch.distributed / RCCL
 *                   (shard.py), or a single-GPU process.
 *   n_devices  > 1: SINGLE-PROCESS MULTI-GPU MODE, for the consumer this library is a drop-in for: one Rust process
 *                   (phase2/src/bin/prove.rs -> bellman/src/groth16/prover.rs:250-298 -> multiexp.rs:330-355) driving the 8 GPUs of a
 *                   node.  mi355zk_bn254_g{1,2}_msm (host buffers) with >= 2^20 exponents (env MI355ZK_MULTI_MIN_LOG) is then cut
 *                   into one cell per device -- contiguous point ranges, each evaluated by the single-GPU pipeline on its
 *                   own device from its own host thread, its exponents streamed over that device's PCIe link, its copy of a PINNED
 *                   base vector kept resident there -- and the n_devices Jacobian partials are joined on the host ("D2H of 8
 *                   records": no collective inside one process).  Result, return code and mi355zk_last_error_index are those of
 *                   the single-device call (the error at the lowest exponent index wins).  Shorter calls run whole, on the devices
 *                   of the set in turn, so a prover's eight concurrent multiexps spread over the node.  The calling thread's
 *                   current device is device_ids[0] on return and is left alone by the multiexps.  Device-pointer (`_dev`) entry
 *                   points are unaffected: their buffers live on one device, the caller's current one.  An id may be repeated
 *                   (logical devices sharing a GPU): that is how the mode is tested on a one-GPU box.
 * Returns 3 for an id that is not a visible device, < 0 if a device is not gfx950.  Replaces nothing in the reference (it has no
 * device). */
int mi355zk_init(const int *device_ids, int n_devices);
/* number of (logical) devices host-buffer multiexps are spread over: 1 unless mi355zk_init was given more */
int mi355zk_device_count(void);
/* GPUs visible to the process (hipGetDeviceCount; 0 when there is none or no driver): the ids 0 .. count-1 are what a caller
 * without HIP bindings of its own passes to mi355zk_init */
int mi355zk_visible_devices(void);
void mi355zk_shutdown(void);
const char *mi355zk_version(void);
/* The ABI revision this library was built with; a binding compares it with the MI355ZK_ABI_VERSION of the header it was written against
 * and refuses to run on a mismatch (lib.py and integration/mi355zk.rs do).  Bumped whenever a prototype or the meaning of an argument
 * changes: 6 = round 5's breaks -- batch_exp's `same_scalar` and point_fft's `inverse` became the bit masks `mode` (a legacy "true" of 2
 * would now read as MI355ZK_G2_TRUSTED_SUBGROUP), sparse_matvec[_dev] gained a trailing `flags`. */
#define MI355ZK_ABI_VERSION 6
int mi355zk_abi_version(void);

/* ---- multiexp: host buffers.  Replaces bellman/src/multiexp.rs:330 `multiexp` for
 * S = (Arc<Vec<G1Affine>>, usize) (source.rs:36-70); `base_offset` is that usize. */
int mi355zk_bn254_g1_msm(const uint8_t *bases, size_t n_bases, size_t base_offset,
                         const uint64_t *scalars, size_t n_scalars,
                         const uint32_t *density, size_t density_bits,
                         uint64_t out_xyz[12]);
/* The CRS is an immutable `Arc<Vec<G>>` reused by every proof (groth16/mod.rs:216-238).  A caller that can PROMISE that --
 * the Rust shim keeps a clone of the Arc next to its raw-record vector, so the allocation is neither rewritten nor freed --
 * pins the vector: mi355zk_bases_cache_pin(ptr, n_bases, group 1|2).  Host-buffer calls over exactly (ptr, n_bases) then keep
 * their uploaded copy on the device and only the scalars cross PCIe after the first call (large calls are streamed: chunked
 * upload overlapped with the kernels).  Vectors that were NOT pinned are uploaded on every call: the plain `const uint8_t*`
 * entry is correct whatever the caller does with its buffer between calls.  mi355zk_bases_cache_invalidate(ptr) ends the
 * promise and drops the device copy (NULL: every vector) -- call it before rewriting or freeing a pinned vector.  A
 * fingerprint of sampled records is re-checked on every call as a second line of defence.  LRU-bounded by env
 * MI355ZK_BASES_CACHE_GB (default 64, 0 = off); env MI355ZK_BASES_CACHE_IMPLICIT=1 treats every vector as pinned. */
int mi355zk_bases_cache_pin(const void *host_bases, size_t n_bases, int group);
/* the same promise, and the library may also keep the vector's WINDOW TABLE on the device (table mode, below: n_windows times the
 * vector, inside MI355ZK_BASES_CACHE_GB): host-buffer calls over it that are not cut into chunks (< 2^23 exponents) then run in
 * table mode from the second call on -- what a prover wants for the h / l / a / b vectors of its Parameters. */
int mi355zk_bases_cache_pin_tables(const void *host_bases, size_t n_bases, int group);
void mi355zk_bases_cache_invalidate(const void *host_bases);
/* diagnostics: 1 when device copies of (parts of) the vector at host_bases are cached, else 0; *device_bytes = their total over all devices -- the whole
 * vector on the one device that runs the calls over it, or, after multi-GPU calls (mi355zk_init with n_devices > 1), one slice per device: n / N
 * records each -- and *table_bytes = the size of its window table or 0 */
int mi355zk_bases_cache_info(const void *host_bases, size_t *device_bytes, size_t *table_bytes);
/* Same for G = G2Affine (prover.rs:297-298). */
int mi355zk_bn254_g2_msm(const uint8_t *bases, size_t n_bases, size_t base_offset,
                         const uint64_t *scalars, size_t n_scalars,
                         const uint32_t *density, size_t density_bits,
                         uint64_t out_xyz[24]);

/* ---- multiexp: bases and scalars already resident in HBM (the CRS / tau-table is reused across
 * calls: `Arc<Vec<G>>` inside groth16::Parameters, groth16/mod.rs:216-238).  `density` stays a HOST
 * pointer (it is tiny and the library needs its prefix sums).  `stream` is a hipStream_t (NULL = the
 * default stream); the call returns after the result has been copied back. */
int mi355zk_bn254_g1_msm_dev(const void *d_bases, size_t n_bases, size_t base_offset,
                             const void *d_scalars, size_t n_scalars,
                             const uint32_t *density, size_t density_bits,
                             void *stream, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_msm_dev(const void *d_bases, size_t n_bases, size_t base_offset,
                             const void *d_scalars, size_t n_scalars,
                             const uint32_t *density, size_t density_bits,
                             void *stream, uint64_t out_xyz[24]);
/* ---- verification multiexps of the ceremony code, device-resident inputs:
 * dense_multiexp (powersoftau/src/utils.rs:189-292): sum_i exp_i * base_i with bases.len() == exponents.len();
 * infinity bases add nothing and there are no Source errors.
 * merge_pairs (powersoftau/src/utils.rs:112-128; phase2/src/utils.rs:59-105; power_pairs = merge_pairs(v[0..n-1], v[1..])):
 * s = sum rho_i * v1_i and sx = sum rho_i * v2_i for ONE scalar vector rho -- digit extraction and sorts are
 * shared between the two sums.  rho are canonical FrRepr like every exponent of this ABI. */
int mi355zk_bn254_g1_dense_multiexp_dev(const void *d_bases, const void *d_scalars, size_t n, void *stream, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_dense_multiexp_dev(const void *d_bases, const void *d_scalars, size_t n, void *stream, uint64_t out_xyz[24]);
int mi355zk_bn254_g1_merge_pairs_dev(const void *d_v1, const void *d_v2, const void *d_rho, size_t n, void *stream, uint64_t out_s[12], uint64_t out_sx[12]);
int mi355zk_bn254_g2_merge_pairs_dev(const void *d_v1, const void *d_v2, const void *d_rho, size_t n, void *stream, uint64_t out_s[24], uint64_t out_sx[24]);
/* The same two on HOST buffers, over the device set of mi355zk_init (a single-process caller: powersoftau's verify_transform runs them over
 * 2^21 .. 2^28-point vectors): the sums are linear in the points, so the vectors are cut into pieces of 2^22 points, every piece is one
 * device call and the Jacobian partials are added on the host; the pieces are dealt to two host threads per device (one uploads while the
 * other computes).  Records and scalars as in the _dev forms; v1, v2 may overlap (power_pairs: v2 = v1 + one record).  Synchronous. */
int mi355zk_bn254_g1_dense_multiexp(const uint8_t *bases, const uint64_t *scalars, size_t n, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_dense_multiexp(const uint8_t *bases, const uint64_t *scalars, size_t n, uint64_t out_xyz[24]);
int mi355zk_bn254_g1_merge_pairs(const uint8_t *v1, const uint8_t *v2, const uint64_t *rho, size_t n, uint64_t out_s[12], uint64_t out_sx[12]);
int mi355zk_bn254_g2_merge_pairs(const uint8_t *v1, const uint8_t *v2, const uint64_t *rho, size_t n, uint64_t out_s[24], uint64_t out_sx[24]);
/* ONE WINDOW GROUP of a multiexp, for multi-GPU runs that shard by scalar windows as well as by point range: the windows of
 * the geometry chosen for n_scalars (a window count divisible by window_groups) are dealt out in window_groups equal groups
 * and only group `window_group` is evaluated: out = sum over its windows w of B^w * T_w.  The partials of all groups add up
 * (mi355zk_bn254_g{1,2}_add) to what mi355zk_bn254_g{1,2}_msm_dev returns; errors and their indices are those of the full
 * call.  (1, 0) is the full multiexp. */
int mi355zk_bn254_g1_msm_part_dev(const void *d_bases, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                  const uint32_t *density, size_t density_bits, uint32_t window_groups, uint32_t window_group,
                                  void *stream, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_msm_part_dev(const void *d_bases, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                  const uint32_t *density, size_t density_bits, uint32_t window_groups, uint32_t window_group,
                                  void *stream, uint64_t out_xyz[24]);
/* The general form: flags + window groups.  MI355ZK_MSM_SCALARS_MONTGOMERY: `d_scalars` holds Fr elements in MONTGOMERY form
 * (the prover's `Vec<Scalar<E>>` / `Vec<E::Fr>`) instead of canonical FrRepr -- the O(m) conversion passes
 * scalars_into_representations / field_elements_into_representations (bellman/src/groth16/prover.rs:89-129: `into_repr()` per
 * element) are fused into the digit extraction (one Montgomery reduction per scalar, no extra pass over HBM). */
#define MI355ZK_MSM_SCALARS_MONTGOMERY 1u
int mi355zk_bn254_g1_msm_ex_dev(const void *d_bases, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                const uint32_t *density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void *stream, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_msm_ex_dev(const void *d_bases, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                const uint32_t *density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void *stream, uint64_t out_xyz[24]);
/* ---- TABLE MODE: multiexps over a base vector that does not change between calls -- the `Arc<Vec<G>>` of groth16::Parameters
 * (bellman/src/groth16/mod.rs:216-238: every proof of a circuit queries the same h / l / a / b_g1 / b_g2) -- evaluated against a
 * precomputed WINDOW TABLE of that vector:  table[w * n_bases + i] = 2^(shift of window w) * bases[i]  for the n_windows windows of
 * mi355zk_msm_table_geometry(n_bases).  A digit of any window then belongs to the bucket of its value in ONE bucket set shared
 * by all windows: one reduction instead of one per window, and a wider window (fewer additions) on the 2^19 .. 2^24-point calls a
 * prover makes (nothing to gain at 2^26).  Same contract, result and errors as mi355zk_bn254_g{1,2}_msm_ex_dev
 * with window_groups = 1 (multiexp.rs:330-355; `base_offset` / `density` as there), at n_windows times the base memory.
 *   table_geometry:   window_bits / n_windows for a vector of n_bases points (group 1 = G1, 2 = G2): the table holds
 *                     n_windows * n_bases affine records (64 B / 128 B each).
 *   table_build_dev:  fills d_table (table_bytes >= n_windows * n_bases * record) from d_bases; d_table may start with the
 *                     bases themselves (d_table == d_bases: window 0 is the vector).  Synchronises `stream`.  One-time work.
 *                     Window w + 1 is window w doubled width[w] times (plain doublings, one batched normalisation): exact for
 *                     every point the decoders admit, in the order-r subgroup or not.
 *   msm_table_dev:    the multiexp; `n_bases` is the length of the ORIGINAL vector (the table's window stride). */
int mi355zk_msm_table_geometry(size_t n_bases, int group, uint32_t *window_bits, uint32_t *n_windows);
int mi355zk_bn254_g1_msm_table_build_dev(const void *d_bases, size_t n_bases, void *d_table, size_t table_bytes, void *stream);
int mi355zk_bn254_g2_msm_table_build_dev(const void *d_bases, size_t n_bases, void *d_table, size_t table_bytes, void *stream);
int mi355zk_bn254_g1_msm_table_dev(const void *d_table, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                   const uint32_t *density, size_t density_bits, uint32_t flags, void *stream, uint64_t out_xyz[12]);
int mi355zk_bn254_g2_msm_table_dev(const void *d_table, size_t n_bases, size_t base_offset, const void *d_scalars, size_t n_scalars,
                                   const uint32_t *density, size_t density_bits, uint32_t flags, void *stream, uint64_t out_xyz[24]);
/* exponent index at which the last failing multiexp of this thread raised its error, or -1 */
long long mi355zk_last_error_index(void);
/* bits of the bucket field (c for the power-of-two window layouts, ceil(log2(B/2 + 1)) for the mixed-radix ones) and
 * the window count the library would use for n scalars (diagnostics / DESIGN.md section 4) */
int mi355zk_msm_window_bits(size_t n_scalars, int *n_windows);
/* the same for a run whose windows are dealt out in window_groups groups (the window count is then a multiple of it) */
int mi355zk_msm_window_bits_groups(size_t n_scalars, uint32_t window_groups, int *n_windows);

/* ---- Fr NTT.  Replaces bellman/src/domain.rs:263 `best_fft` for T = Scalar<Bn256>:
 * a[0..2^log_n) in place, natural order in and out, `omega` of order 2^log_n (Montgomery form). */
int mi355zk_bn254_fr_ntt(uint64_t *a, uint32_t log_n, const uint64_t omega[4]);
/* EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} (domain.rs:154-203) with the constants
 * `from_coeffs` derives for m = 2^log_n (domain.rs:84-98: omega, omegainv, geninv = 7^-1, minv). */
int mi355zk_bn254_fr_domain_op(uint64_t *a, uint32_t log_n, int op);
int mi355zk_bn254_fr_fft(uint64_t *a, uint32_t log_n);
int mi355zk_bn254_fr_ifft(uint64_t *a, uint32_t log_n);
int mi355zk_bn254_fr_coset_fft(uint64_t *a, uint32_t log_n);
int mi355zk_bn254_fr_icoset_fft(uint64_t *a, uint32_t log_n);
/* device-resident variants: asynchronous on `stream` (no host synchronisation). */
int mi355zk_bn254_fr_ntt_dev(void *d_a, uint32_t log_n, const uint64_t omega[4], void *stream);
/* (round 5) best_fft with the scalings around it fused in, for ANY factors -- distribute_powers(g) takes any g (domain.rs:176-189); the four
 * domain operations are this call with the domain's constants:
 *     a[i] *= pre_g^i   (if pre_g)        X[k] = sum_i a[i] * omega^(i k)        X[k] *= post_c * post_g^k   (each factor if given)
 * omega: a 2^log_n-th root of unity; all four are Montgomery forms (4 x u64); pre_g / post_c / post_g may be NULL.  Up to 2^20 the
 * factors live in a per-(omega, factors) copy of the inter-pass twiddle table (at most four per omega are kept; see DESIGN.md, NTT). */
int mi355zk_bn254_fr_ntt_scaled_dev(void *d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t pre_g[4], const uint64_t post_c[4],
                                    const uint64_t post_g[4], void *stream);
int mi355zk_bn254_fr_domain_op_dev(void *d_a, uint32_t log_n, int op, void *stream);
/* (round 5) The same EvaluationDomain operation on `batch` (1 .. 64) DISTINCT device arrays of 2^log_n elements each, in place -- what
 * prover.rs:217-241 does to a, b and c one after the other (ifft, then coset_fft).  d_arrays: a HOST array of `batch` device pointers.
 * Results are those of `batch` calls of mi355zk_bn254_fr_domain_op_dev, byte for byte; every pass is ONE launch over the tiles of all
 * the arrays, so that one transform's loads and stores run under another's butterflies (2^20: 0.105 -> 0.087 ms per transform).
 * 3 = a null or repeated pointer, batch out of range, unknown op. */
int mi355zk_bn254_fr_domain_op_batch_dev(void *const *d_arrays, uint32_t batch, uint32_t log_n, int op, void *stream);
/* the domain constants themselves (Montgomery form), for callers that keep their own EvaluationDomain */
int mi355zk_bn254_fr_domain_constants(uint32_t log_n, uint64_t omega[4], uint64_t omegainv[4], uint64_t geninv[4], uint64_t minv[4]);

/* ---- elementwise Fr operations of EvaluationDomain on device-resident data (asynchronous on `stream`):
 * a[i] *= b[i] (mul_assign, domain.rs:236-249) and a[i] -= b[i] (sub_assign, domain.rs:251-260). */
int mi355zk_bn254_fr_mul_assign_dev(void *d_a, const void *d_b, size_t n, void *stream);
int mi355zk_bn254_fr_sub_assign_dev(void *d_a, const void *d_b, size_t n, void *stream);
/* out[i] = into_repr(in[i]): Montgomery Fr -> canonical FrRepr (scalars_into_representations / field_elements_into_representations,
 * prover.rs:89-129) as a standalone pass; d_out may alias d_in.  (The multiexp can take Montgomery scalars directly: msm_ex_dev.) */
int mi355zk_bn254_fr_into_repr_dev(void *d_out, const void *d_in, size_t n, void *stream);
/* EvaluationDomain::divide_by_z_on_coset (domain.rs:207-234): a[i] *= (g^m - 1)^-1 with m = 2^log_n and g = 7 the multiplicative
 * generator (z(tau) = tau^m - 1, domain.rs:207-212) -- the division step of the prover's H polynomial (prover.rs:217-241), so that
 * the whole ifft / coset_fft / mul / sub / divide / icoset_fft pipeline stays in HBM.  Asynchronous on `stream`. */
int mi355zk_bn254_fr_divide_by_z_on_coset_dev(void *d_a, uint32_t log_n, void *stream);
/* EvaluationDomain::z (domain.rs:207-212): out = tau^(2^log_n) - 1, Montgomery in and out (host arithmetic). */
int mi355zk_bn254_fr_domain_z(uint32_t log_n, const uint64_t tau[4], uint64_t out[4]);
/* Measured Montgomery-product rate of this library (the integer-ALU roofline the kernels are priced
 * against): `blocks` x 256 lanes each run 4 independent chains of `iters` products.  which: 0 Fq, 1 Fr.
 * out[0..3] = a*b^iters (Montgomery arithmetic, lane 0, chain 0) for a parity check; *ms = kernel time. */
int mi355zk_ubench_fp_mul(int which, uint32_t blocks, uint32_t iters, const uint64_t a[4], const uint64_t b[4], uint64_t out[16], float *ms);

/* ---- self-test hooks: the kernels' "U-form" arithmetic (29-bit lazy limbs, csrc/fieldu.hpp, curveu.hpp)
 * compiled for the HOST, so that it can be checked against the oracle / big-int model without a GPU. */
int mi355zk_selftest_u_mul(int which, const uint32_t a[9], const uint32_t b[9], uint32_t out[9]);
int mi355zk_selftest_u_mul_shoup(int which, const uint32_t a[9], const uint32_t w_plain[8], uint32_t out[9], uint32_t out_wq[9]);
int mi355zk_selftest_u_sub(int which, int k, int s, const uint32_t a[9], const uint32_t b[9], uint32_t out[9]);
int mi355zk_selftest_u_pack(int which, const uint64_t a_std[4], uint32_t out_u[9], const uint32_t in_u[9], uint64_t out_std[4]);
int mi355zk_selftest_u_reduce32(int which, const uint32_t in_u[9], uint64_t out_std[4]);
int mi355zk_selftest_g1_accumulate(int mode, const uint64_t *affine_pts, const uint8_t *negate, size_t n, uint64_t out_xyzz[16]);
int mi355zk_selftest_g1_record_sum(int mode, const uint64_t *affine_pts, const uint8_t *negate, const uint32_t *group, size_t n, size_t n_groups, uint64_t out_xyzz[16]);
int mi355zk_selftest_g2_record_sum(int mode, const uint64_t *affine_pts, const uint8_t *negate, const uint32_t *group, size_t n, size_t n_groups, uint64_t out_xyzz[32]);
int mi355zk_selftest_msm_digits(size_t n_scalars, uint32_t window_groups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct, int32_t *digits, uint32_t *geom);
int mi355zk_selftest_glv_split(const uint32_t k[8], uint32_t out[12]);
int mi355zk_selftest_glv_wnaf5(const uint32_t m[5], int8_t digits[164]);
int mi355zk_selftest_glv2_split(const uint32_t k[8], uint32_t out[10]);
int mi355zk_selftest_g2_psi(const uint64_t affine_pt[16], uint64_t out_xyz[24]);
int mi355zk_selftest_g2_scalar_mul_u(const uint64_t affine_pt[16], const uint64_t scalar[4], uint64_t out_xyz[24]);
int mi355zk_selftest_g2_accumulate(int mode, const uint64_t *affine_pts, const uint8_t *negate, size_t n, uint64_t out_xyzz[32]);

/* ---- mode / flag bits of the scalar-multiplication entry points below (batch_exp: `mode`; point_fft: `mode`; sparse_matvec: `flags`).
 * batch_exp and sparse_matvec return, by default, the reference's result for EVERY record the reference's decoders admit (point_fft: for
 * G2 records on the twist -- in the subgroup or not -- and G1 records on the curve; its stage kernels do not carry batch_exp's handling of
 * records on no curve, whose sums have no order-independent meaning anyway): the reference's `mul` is a wNAF
 * double-and-add (pairing/src/wnaf.rs:4-71, ec.rs:538-560, 983-997), i.e. the plain group law, and its bn256 decoders test the curve
 * equation at most (ec.rs:133-150, 1136-1344) -- never membership in the order-r subgroup of the twist -- while `compute_constrained`
 * reads its challenge unchecked (powersoftau/src/bin/compute_constrained.rs:16).  The G2 kernels therefore run PLAIN fixed windows over
 * the whole scalar.  MI355ZK_G2_TRUSTED_SUBGROUP is the caller's PROMISE that every G2 record of the call lies in the order-r subgroup
 * (honest ceremony data does; mi355zk_bn254_g2_subgroup_check_dev establishes it for untrusted data): the kernels then split each
 * scalar over the twist's endomorphism psi (k P = k1 P + k2 psi(P), k1, k2 < 2^128; psi(P) = mu P holds in that subgroup only) and
 * run 1.3-1.4 x faster (profiles/r05_g2_exact_cost.json).  A broken promise is not detected: a record with a cofactor component then
 * yields a point that is not k P.  The bit is accepted and ignored by the G1 entry points: E(Fq) has prime order r, so G1's split over
 * phi(x, y) = (beta x, y) is exact for every point ON the curve. */
#define MI355ZK_EXP_SAME_SCALAR 1      /* batch_exp: ONE scalar for all points (phase2 contribute) instead of one per point */
#define MI355ZK_FFT_INVERSE 1          /* point_fft: omega^-1 and the 1/m scaling */
#define MI355ZK_G2_TRUSTED_SUBGROUP 2  /* all three: the promise above */

/* ---- QAP evaluation: the per-variable sparse sums of MPCParameters::new
 * (phase2/src/parameters.rs:225-294: `a_g1[v] += coeffs_g1[lag].mul(coeff)` over the terms of variable v, then
 * batch_normalization) as one CSR-matrix x point-vector product:
 *   out[r] = sum_{t = row_ptr[r]}^{row_ptr[r+1]-1} coeff[t] * bases[col[t]],  r < n_rows,  affine out (all-zero = infinity).
 * row_ptr: u32[n_rows + 1] (row_ptr[n_rows] == nnz), col: u32[nnz], coeff: nnz canonical FrRepr; all device pointers.
 * The index arrays are validated on the device (col[t] < n_bases, row_ptr monotone from 0 to nnz): 3 = bad arguments otherwise.
 * `ext` of the reference (three products added) is one call on the concatenated term lists / bases.
 * flags: 0 or MI355ZK_G2_TRUSTED_SUBGROUP (other bits: 3 = bad arguments).  Synchronises `stream` before returning. */
int mi355zk_bn254_g1_sparse_matvec_dev(void *d_out_affine, const void *d_bases_affine, size_t n_bases, const uint32_t *d_row_ptr,
                                       const uint32_t *d_col, const void *d_coeffs, size_t n_rows, size_t nnz, void *stream, int flags);
int mi355zk_bn254_g2_sparse_matvec_dev(void *d_out_affine, const void *d_bases_affine, size_t n_bases, const uint32_t *d_row_ptr,
                                       const uint32_t *d_col, const void *d_coeffs, size_t n_rows, size_t nnz, void *stream, int flags);

/* The same on HOST buffers, over the device set of mi355zk_init (MPCParameters::new in one process on N GPUs): the rows are independent,
 * so device d evaluates the d-th contiguous row range -- its slice of (col, coeff), the whole base vector -- and writes its rows; no
 * exchange.  Same validation (3 = bad arguments) and output as the _dev form.  Synchronous. */
int mi355zk_bn254_g1_sparse_matvec(uint8_t *out_affine, const uint8_t *bases_affine, size_t n_bases, const uint32_t *row_ptr, const uint32_t *col,
                                   const uint64_t *coeffs, size_t n_rows, size_t nnz, int flags);
int mi355zk_bn254_g2_sparse_matvec(uint8_t *out_affine, const uint8_t *bases_affine, size_t n_bases, const uint32_t *row_ptr, const uint32_t *col,
                                   const uint64_t *coeffs, size_t n_rows, size_t nnz, int flags);

/* ---- point codecs: the reference's wire encodings <-> raw affine records.
 * Replaces EncodedPoint::{into_affine, into_affine_unchecked, from_affine} for G1Uncompressed (64 B), G1Compressed
 * (32 B), G2Uncompressed (128 B), G2Compressed (64 B)  (pairing/src/bn256/ec.rs:763-946, 1136-1344), which
 * powersoftau applies to every element of an accumulator file (batched_accumulator.rs read_points_chunk /
 * write_point): big-endian canonical coordinates, Fq2 as c1 || c0, byte 0 bit 7 = "y is the larger root"
 * (compressed), bit 6 = infinity.  n records, device pointers (byte buffers 4-byte aligned), one record per point;
 * the all-zero raw record is the point at infinity.  checked != 0: into_affine (on-curve test for uncompressed
 * input; decompression is on the curve by construction -- for G2 up to the reference's Fq2::sqrt quirk, which is
 * reproduced).  decode returns 0, or the GroupDecodingError of the FIRST failing record with its index in
 * *err_index (may be NULL): 4 NotOnCurve, 6 CoordinateDecodingError, 7 UnexpectedCompressionMode,
 * 8 UnexpectedInformation; failing records decode to infinity.  decode synchronises `stream`; encode is asynchronous. */
int mi355zk_bn254_g1_decode_dev(void *d_out_affine, const void *d_in_bytes, size_t n, int compressed, int checked, void *stream,
                                long long *err_index);
int mi355zk_bn254_g2_decode_dev(void *d_out_affine, const void *d_in_bytes, size_t n, int compressed, int checked, void *stream,
                                long long *err_index);
int mi355zk_bn254_g1_encode_dev(void *d_out_bytes, const void *d_in_affine, size_t n, int compressed, void *stream);
int mi355zk_bn254_g2_encode_dev(void *d_out_bytes, const void *d_in_affine, size_t n, int compressed, void *stream);

/* ---- FFT over curve points: EvaluationDomain<Point<G1>>::fft / ifft (bellman/src/group.rs:22-51
 * under domain.rs:154-173), the Lagrange-basis conversion of powersoftau/src/bin/prepare_phase2.rs:68-131.  In
 * place on 2^log_n AFFINE raw records (64 B, all-zero = infinity); the output is normalised to affine, i.e. what
 * `batch_normalization` + `into_affine` leave (ec.rs:251-299, 596-629).  mode: MI355ZK_FFT_INVERSE (omega^-1 and the 1/m scaling)
 * | MI355ZK_G2_TRUSTED_SUBGROUP; other bits: 3 = bad arguments.  Synchronises `stream` before returning. */
int mi355zk_bn254_g1_point_fft_dev(void *d_points_affine, uint32_t log_n, int mode, void *stream);
/* the same over G2 (128-byte affine records; `coeffs_g2` of prepare_phase2.rs:102-105): exact for every vector of points of the twist
 * (plain windows) unless the caller promises the subgroup */
int mi355zk_bn254_g2_point_fft_dev(void *d_points_affine, uint32_t log_n, int mode, void *stream);

/* ---- batch fixed-base scalar multiplication out[i] = k[i] * P, affine (all-zero = infinity).
 * Building block of the per-point `batch_exp` path (powersoftau/src/batched_accumulator.rs:1130-1181,
 * SURVEY 8f row 1); used here to synthesise tau-table-like bases on the device. */
int mi355zk_bn254_g1_batch_mul_dev(void *d_out_affine, const uint64_t base_affine[8], const void *d_scalars, size_t n, void *stream);
int mi355zk_bn254_g2_batch_mul_dev(void *d_out_affine, const uint64_t base_affine[16], const void *d_scalars, size_t n, void *stream);
/* ---- per-point batch exponentiation out[i] = k[i] * P[i] (powersoftau `batch_exp`, batched_accumulator.rs:1130-1181) or, with
 * MI355ZK_EXP_SAME_SCALAR in `mode`, out[i] = k[0] * P[i] (phase2 contribute, phase2/src/parameters.rs:423-470), normalised to affine
 * like `batch_normalization` (ec.rs:251-299); the all-zero record is infinity on both sides.  mode: MI355ZK_EXP_SAME_SCALAR |
 * MI355ZK_G2_TRUSTED_SUBGROUP (other bits: 3 = bad arguments).  Asynchronous on `stream`.
 * G2 is the reference's `mul` for every record -- in the subgroup, on the twist outside it, or (checked = 0 decoding) on no curve at
 * all: the plain windows are the group law of y^2 = x^3 + (y0^2 - x0^3), which no formula names.  G1 likewise: the split kernels test
 * y^2 = x^3 + 3 per record and hand an off-curve record (checked = 0 decoding only) to the plain-window kernel, so batch_exp returns the
 * reference's `mul` for those too (tests/test_g2_subgroup.py::test_batch_exp_of_records_that_are_on_no_curve). */
int mi355zk_bn254_g1_batch_exp_dev(void *d_out_affine, const void *d_bases_affine, const void *d_scalars, size_t n, int mode, void *stream);
int mi355zk_bn254_g2_batch_exp_dev(void *d_out_affine, const void *d_bases_affine, const void *d_scalars, size_t n, int mode, void *stream);
/* The same on HOST buffers, spread over the device set of mi355zk_init: what MPCParameters::contribute (parameters.rs:423-470) and
 * powersoftau's batch_exp (batched_accumulator.rs:1130-1181) are to a single-process caller.  The points are independent, so device d takes
 * the d-th contiguous point range -- upload, kernels, download from its own host thread, no exchange -- and with one device the
 * vector is one range.  out / bases: n raw affine records (64 / 128 B, all-zero = infinity; out may not alias bases); scalars: n canonical
 * FrRepr, or ONE with MI355ZK_EXP_SAME_SCALAR.  Synchronous.  mode as above. */
int mi355zk_bn254_g1_batch_exp(uint8_t *out_affine, const uint8_t *bases_affine, const uint64_t *scalars, size_t n, int mode);
int mi355zk_bn254_g2_batch_exp(uint8_t *out_affine, const uint8_t *bases_affine, const uint64_t *scalars, size_t n, int mode);
/* The test that lets a caller give the promise MI355ZK_G2_TRUSTED_SUBGROUP for data it did not produce: *bad_index = the lowest index of a G2 record that is on the twist but NOT in
 * the order-r subgroup (-1: all n records are; the all-zero record is the identity).  [x + 1] P + psi([x] P) + psi^2([x] P) == psi^3([2 x] P)
 * with x the 63-bit BN parameter -- one plain double-and-add (no split), sound and complete for BN254; a record that is not on the twist is not a member.  The reference has no counterpart -- its bn256 decoders do not test membership either -- so this
 * is an addition for callers that handle untrusted G2 data, not a drop-in for anything.  Synchronises `stream`. */
int mi355zk_bn254_g2_subgroup_check_dev(const void *d_points_affine, size_t n, void *stream, long long *bad_index);
int mi355zk_selftest_g2_in_subgroup(const uint64_t affine_pt[16]);   /* host run of the same test: 1 / 0 */

/* ---- host-side group helpers on Jacobian results: acc += other (CurveProjective::add_assign,
 * ec.rs:360-454) -- how per-GPU partial sums are joined after the all-gather -- and into_affine
 * (ec.rs:596-629; infinity -> all-zero record). */
int mi355zk_bn254_g1_add(uint64_t acc_xyz[12], const uint64_t other_xyz[12]);
int mi355zk_bn254_g2_add(uint64_t acc_xyz[24], const uint64_t other_xyz[24]);
int mi355zk_bn254_g1_to_affine(uint64_t out_xy[8], const uint64_t xyz[12]);
int mi355zk_bn254_g2_to_affine(uint64_t out_xy[16], const uint64_t xyz[24]);
/* acc = scalar * acc on the host (CurveProjective::mul_assign, ec.rs:538-560), scalar = 4 canonical u64 limbs: the single-point
 * products of a proof assembly (bellman/src/groth16/prover.rs:300-333). */
int mi355zk_bn254_g1_mul(uint64_t acc_xyz[12], const uint64_t scalar[4]);
int mi355zk_bn254_g2_mul(uint64_t acc_xyz[24], const uint64_t scalar[4]);

/* ---- plain device-memory helpers so that a C / Rust caller needs no HIP bindings of its own */
int mi355zk_malloc(void **d_ptr, size_t bytes);
int mi355zk_free(void *d_ptr);
int mi355zk_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int mi355zk_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
int mi355zk_sync(void *stream);

/* ---- per-kernel timing (HIP events on the launch stream) for bench.py's roofline leg.
 * names: "msm_digits" "msm_sort" "msm_accumulate_heavy" "msm_accumulate" "msm_reduce" "ntt_pass" "ntt_scale"
 * mi355zk_prof_enable: 0 off; 1 every kernel group (~20 event records per multiexp: 0.1-0.2 ms of host time per call); 2 only the
 * group named by mi355zk_prof_only (two records per call: what bench.py leaves on inside its timed region, for the dominant kernel). */
void mi355zk_prof_enable(int on);
void mi355zk_prof_only(const char *kernel);
void mi355zk_prof_reset(void);
int mi355zk_prof_get(const char *kernel, double *total_ms, long *count);

#ifdef __cplusplus
}
#endif
#endif /* MI355ZK_H */
