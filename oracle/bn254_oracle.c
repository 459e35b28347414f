/*
 * oracle/bn254_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's hot path:
 *   - BN254 Fq / Fr / Fq2 arithmetic                     (field.h)
 *   - G1 / G2 Jacobian group law of pairing/src/bn256/ec.rs (tmpl_curve.h)
 *   - bellman_ce multiexp (Pippenger) + Source/Density   (tmpl_multiexp.h)
 *   - bellman_ce EvaluationDomain fft/ifft/coset_fft     (tmpl_fft.h)
 * plus the same templates instantiated over the reference's DummyEngine
 * (bellman/src/tests/dummy_engine.rs: Fr = Z/64513, G1 = G2 = Fr) so that the hard-coded literals of
 * bellman/src/groth16/tests/mod.rs `test_xordemo` pin the ALGORITHM STRUCTURE of this restatement.
 *
 * PARITY STATUS: the Rust reference cannot be built in this image (no cargo/rustc; arithmetic lives
 * in un-vendored ff_ce 0.7.1) and it holds no BN254 known-answer vectors for multiexp / fft.  BN254
 * *values* are therefore pinned only indirectly: reference constants (fq.rs / fr.rs literals),
 * the DummyEngine KAT, and an independent Python big-int model (tests/golden/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "field.h"

/* ------------------------------------------------------------------ G1 over Fq */
#define CNAME(x) g1_##x
#define F fe_t
#define F_ZERO(p) fe_zero(p)
#define F_ONE(p) fe_one(&FQ, p)
#define F_IS_ZERO(p) fe_is_zero(p)
#define F_EQ(a, b) fe_eq(a, b)
#define F_ADD(r, a, b) fe_add(&FQ, r, a, b)
#define F_SUB(r, a, b) fe_sub(&FQ, r, a, b)
#define F_DBL(r, a) fe_double(&FQ, r, a)
#define F_NEG(r, a) fe_neg(&FQ, r, a)
#define F_MUL(r, a, b) fe_mul(&FQ, r, a, b)
#define F_SQR(r, a) fe_sqr(&FQ, r, a)
#define F_INV(r, a) fe_inv(&FQ, r, a)
#include "tmpl_curve.h"
#undef CNAME
#undef F
#undef F_ZERO
#undef F_ONE
#undef F_IS_ZERO
#undef F_EQ
#undef F_ADD
#undef F_SUB
#undef F_DBL
#undef F_NEG
#undef F_MUL
#undef F_SQR
#undef F_INV

/* ------------------------------------------------------------------ G2 over Fq2 */
#define CNAME(x) g2_##x
#define F fe2_t
#define F_ZERO(p) fe2_zero(p)
#define F_ONE(p) fe2_one(p)
#define F_IS_ZERO(p) fe2_is_zero(p)
#define F_EQ(a, b) fe2_eq(a, b)
#define F_ADD(r, a, b) fe2_add(r, a, b)
#define F_SUB(r, a, b) fe2_sub(r, a, b)
#define F_DBL(r, a) fe2_double(r, a)
#define F_NEG(r, a) fe2_neg(r, a)
#define F_MUL(r, a, b) fe2_mul(r, a, b)
#define F_SQR(r, a) fe2_sqr(r, a)
#define F_INV(r, a) fe2_inv(r, a)
#include "tmpl_curve.h"
#undef CNAME
#undef F
#undef F_ZERO
#undef F_ONE
#undef F_IS_ZERO
#undef F_EQ
#undef F_ADD
#undef F_SUB
#undef F_DBL
#undef F_NEG
#undef F_MUL
#undef F_SQR
#undef F_INV

/* bench-only window width override for the BN254 multiexp instances (0 = the reference's rule) */
static uint32_t oracle_window_override = 0;
/* ------------------------------------------------------------------ multiexp: G1, G2 */
#define MNAME(x) g1m_##x
#define M_AFFINE g1_affine_t
#define M_PROJ g1_jac_t
#define M_AFFINE_IS_ZERO(p) g1_affine_is_zero(p)
#define M_SET_ZERO(p) g1_set_zero(p)
#define M_ADD_MIXED(p, a) g1_add_mixed(p, a)
#define M_ADD(p, o) g1_add(p, o)
#define M_DOUBLE(p) g1_double(p)
#define M_SCALAR_LIMBS 4
#define M_NUM_BITS 254
#define ORACLE_WINDOW_OVERRIDE oracle_window_override
#include "tmpl_multiexp.h"
#undef ORACLE_WINDOW_OVERRIDE
#undef MNAME
#undef M_AFFINE
#undef M_PROJ
#undef M_AFFINE_IS_ZERO
#undef M_SET_ZERO
#undef M_ADD_MIXED
#undef M_ADD
#undef M_DOUBLE
#undef M_SCALAR_LIMBS
#undef M_NUM_BITS

#define MNAME(x) g2m_##x
#define M_AFFINE g2_affine_t
#define M_PROJ g2_jac_t
#define M_AFFINE_IS_ZERO(p) g2_affine_is_zero(p)
#define M_SET_ZERO(p) g2_set_zero(p)
#define M_ADD_MIXED(p, a) g2_add_mixed(p, a)
#define M_ADD(p, o) g2_add(p, o)
#define M_DOUBLE(p) g2_double(p)
#define M_SCALAR_LIMBS 4
#define M_NUM_BITS 254
#define ORACLE_WINDOW_OVERRIDE oracle_window_override
#include "tmpl_multiexp.h"
#undef ORACLE_WINDOW_OVERRIDE
#undef MNAME
#undef M_AFFINE
#undef M_PROJ
#undef M_AFFINE_IS_ZERO
#undef M_SET_ZERO
#undef M_ADD_MIXED
#undef M_ADD
#undef M_DOUBLE
#undef M_SCALAR_LIMBS
#undef M_NUM_BITS

/* ------------------------------------------------------------------ DummyEngine (Z/64513) */
/* bellman/src/tests/dummy_engine.rs:25-88 (field), :341-400 (group = the field's additive group) */
#define DUMMY_P 64513u
typedef uint32_t dfe_t;
static inline void dfe_pow(dfe_t *r, const dfe_t *a, uint64_t e) {
  uint64_t res = 1, b = *a;
  int found = 0;
  for (int i = 63; i >= 0; --i) {
    int bit = (int)((e >> i) & 1);
    if (found) res = res * res % DUMMY_P; else found = bit;
    if (bit) res = res * b % DUMMY_P;
  }
  *r = (dfe_t)res;
}
static inline void dfe_inv(dfe_t *r, const dfe_t *a) { dfe_pow(r, a, DUMMY_P - 2); }

#define MNAME(x) dm_##x
#define M_AFFINE dfe_t
#define M_PROJ dfe_t
#define M_AFFINE_IS_ZERO(p) (*(p) == 0)
#define M_SET_ZERO(p) (*(p) = 0)
#define M_ADD_MIXED(p, a) (*(p) = (*(p) + *(a)) % DUMMY_P)
#define M_ADD(p, o) (*(p) = (*(p) + *(o)) % DUMMY_P)
#define M_DOUBLE(p) (*(p) = (*(p) << 1) % DUMMY_P)
#define M_SCALAR_LIMBS 1
#define M_NUM_BITS 16
#include "tmpl_multiexp.h"
#undef MNAME
#undef M_AFFINE
#undef M_PROJ
#undef M_AFFINE_IS_ZERO
#undef M_SET_ZERO
#undef M_ADD_MIXED
#undef M_ADD
#undef M_DOUBLE
#undef M_SCALAR_LIMBS
#undef M_NUM_BITS

#define FFT_UNDEF_ALL
#define TNAME(x) dfft_##x
#define T_FE dfe_t
#define S_FE dfe_t
#define S_ONE(p) (*(p) = 1)
#define T_ZERO(p) (*(p) = 0)
#define T_ADD(r, a, b) (*(r) = (*(a) + *(b)) % DUMMY_P)
#define T_SUB(r, a, b) (*(r) = ((DUMMY_P + *(a)) - *(b)) % DUMMY_P)
#define S_MUL(r, a, b) (*(r) = (dfe_t)((uint64_t)*(a) * *(b) % DUMMY_P))
#define T_MULS(r, a, b) S_MUL(r, a, b)
#define S_INV(r, a) dfe_inv(r, a)
#define S_POW64(r, a, e) dfe_pow(r, a, e)
#define S_ROOT_OF_UNITY(p) (*(p) = 57751)   /* dummy_engine.rs:292-294 */
#define S_GENERATOR(p) (*(p) = 5)           /* dummy_engine.rs:288-290 */
#define S_FROM_U64(p, v) (*(p) = (dfe_t)((v) % DUMMY_P))
#define S_S 10                              /* dummy_engine.rs:258 */
#include "tmpl_fft.h"
#include "tmpl_fft_undef.h"

/* ------------------------------------------------------------------ BN254 Fr FFT */
static inline void fr_pow64(fe_t *r, const fe_t *a, uint64_t e) { fe_pow(&FR, r, a, &e, 1); }
static inline void fr_root_of_unity(fe_t *r) {
  /* ff_derive: ROOT_OF_UNITY = GENERATOR^((r-1)/2^S), S = 28 (fr.rs:5,31-34) */
  uint64_t t[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  /* (r - 1) >> 28 */
  t[0] -= 1;
  for (int i = 0; i < 4; ++i) t[i] = (t[i] >> 28) | (i < 3 ? t[i + 1] << 36 : 0);
  uint64_t seven[4] = {7, 0, 0, 0};
  fe_t g;
  fe_from_canonical(&FR, &g, seven);
  fe_pow(&FR, r, &g, t, 4);
}
static inline void fr_generator(fe_t *r) { uint64_t seven[4] = {7, 0, 0, 0}; fe_from_canonical(&FR, r, seven); }
static inline void fr_from_u64(fe_t *r, uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; fe_from_canonical(&FR, r, c); }

/* scalar field of every BN254 instantiation */
#define S_FE fe_t
#define S_ONE(p) fe_one(&FR, p)
#define S_MUL(r, a, b) fe_mul(&FR, r, a, b)
#define S_INV(r, a) fe_inv(&FR, r, a)
#define S_POW64(r, a, e) fr_pow64(r, a, e)
#define S_ROOT_OF_UNITY(p) fr_root_of_unity(p)
#define S_GENERATOR(p) fr_generator(p)
#define S_FROM_U64(p, v) fr_from_u64(p, v)
#define S_S 28

/* Scalar<Bn256> elements (group.rs:53-82) */
#define TNAME(x) frfft_##x
#define T_FE fe_t
#define T_ZERO(p) fe_zero(p)
#define T_ADD(r, a, b) fe_add(&FR, r, a, b)
#define T_SUB(r, a, b) fe_sub(&FR, r, a, b)
#define T_MULS(r, a, b) fe_mul(&FR, r, a, b)
#include "tmpl_fft.h"
#undef TNAME
#undef T_FE
#undef T_ZERO
#undef T_ADD
#undef T_SUB
#undef T_MULS

/* Point<G1> / Point<G2> elements (group.rs:22-51): group_mul_assign(by) = self.0.mul_assign(by.into_repr()) */
static inline void g1pt_muls(g1_jac_t *r, const g1_jac_t *a, const fe_t *s) { uint64_t k[4]; fe_to_canonical(&FR, k, s); g1_jac_t t = *a; g1_mul(&t, k); *r = t; }
static inline void g1pt_add(g1_jac_t *r, const g1_jac_t *a, const g1_jac_t *b) { g1_jac_t t = *a; g1_add(&t, b); *r = t; }
static inline void g1pt_sub(g1_jac_t *r, const g1_jac_t *a, const g1_jac_t *b) { g1_jac_t t = *a, nb = *b; g1_negate(&nb); g1_add(&t, &nb); *r = t; }
#define TNAME(x) g1fft_##x
#define T_FE g1_jac_t
#define T_ZERO(p) g1_set_zero(p)
#define T_ADD(r, a, b) g1pt_add(r, a, b)
#define T_SUB(r, a, b) g1pt_sub(r, a, b)
#define T_MULS(r, a, b) g1pt_muls(r, a, b)
#include "tmpl_fft.h"
#undef TNAME
#undef T_FE
#undef T_ZERO
#undef T_ADD
#undef T_SUB
#undef T_MULS

static inline void g2pt_muls(g2_jac_t *r, const g2_jac_t *a, const fe_t *s) { uint64_t k[4]; fe_to_canonical(&FR, k, s); g2_jac_t t = *a; g2_mul(&t, k); *r = t; }
static inline void g2pt_add(g2_jac_t *r, const g2_jac_t *a, const g2_jac_t *b) { g2_jac_t t = *a; g2_add(&t, b); *r = t; }
static inline void g2pt_sub(g2_jac_t *r, const g2_jac_t *a, const g2_jac_t *b) { g2_jac_t t = *a, nb = *b; g2_negate(&nb); g2_add(&t, &nb); *r = t; }
#define TNAME(x) g2fft_##x
#define T_FE g2_jac_t
#define T_ZERO(p) g2_set_zero(p)
#define T_ADD(r, a, b) g2pt_add(r, a, b)
#define T_SUB(r, a, b) g2pt_sub(r, a, b)
#define T_MULS(r, a, b) g2pt_muls(r, a, b)
#include "tmpl_fft.h"

#include "codec.h"

/* ================================================================== exported C API (ctypes) */
#define EXPORT __attribute__((visibility("default")))

/* which: 0 = Fq, 1 = Fr */
static const modulus_t *pick(int which) { return which ? &FR : &FQ; }
EXPORT void oracle_fe_mul(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) { fe_mul(pick(which), (fe_t *)r, (const fe_t *)a, (const fe_t *)b); }
EXPORT void oracle_fe_add(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) { fe_add(pick(which), (fe_t *)r, (const fe_t *)a, (const fe_t *)b); }
EXPORT void oracle_fe_sub(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) { fe_sub(pick(which), (fe_t *)r, (const fe_t *)a, (const fe_t *)b); }
EXPORT int oracle_fe_inv(int which, uint64_t r[4], const uint64_t a[4]) { return fe_inv(pick(which), (fe_t *)r, (const fe_t *)a); }
EXPORT void oracle_fe_from_canonical(int which, uint64_t r[4], const uint64_t a[4]) { fe_from_canonical(pick(which), (fe_t *)r, a); }
EXPORT void oracle_fe_to_canonical(int which, uint64_t r[4], const uint64_t a[4]) { fe_to_canonical(pick(which), r, (const fe_t *)a); }
EXPORT void oracle_fe_mul_many(int which, uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n) {
  for (size_t i = 0; i < n; ++i) fe_mul(pick(which), (fe_t *)(r + 4 * i), (const fe_t *)(a + 4 * i), (const fe_t *)(b + 4 * i));
}
EXPORT void oracle_fq2_mul(uint64_t r[8], const uint64_t a[8], const uint64_t b[8]) { fe2_mul((fe2_t *)r, (const fe2_t *)a, (const fe2_t *)b); }
EXPORT void oracle_fq2_sqr(uint64_t r[8], const uint64_t a[8]) { fe2_sqr((fe2_t *)r, (const fe2_t *)a); }
EXPORT int oracle_fq2_inv(uint64_t r[8], const uint64_t a[8]) { return fe2_inv((fe2_t *)r, (const fe2_t *)a); }
EXPORT void oracle_fr_root_of_unity(uint64_t r[4]) { fr_root_of_unity((fe_t *)r); }

/* group ops; points as Jacobian X,Y,Z Montgomery limbs (12 / 24 u64), affine raw records (8 / 16 u64) */
EXPORT void oracle_g1_double(uint64_t p[12]) { g1_double((g1_jac_t *)p); }
EXPORT void oracle_g1_add(uint64_t p[12], const uint64_t o[12]) { g1_add((g1_jac_t *)p, (const g1_jac_t *)o); }
EXPORT void oracle_g1_add_mixed(uint64_t p[12], const uint64_t o[8]) { g1_add_mixed((g1_jac_t *)p, (const g1_affine_t *)o); }
EXPORT void oracle_g1_mul(uint64_t p[12], const uint64_t k[4]) { g1_mul((g1_jac_t *)p, k); }
EXPORT void oracle_g1_to_affine(uint64_t r[8], const uint64_t p[12]) { g1_to_affine((g1_affine_t *)r, (const g1_jac_t *)p); }
EXPORT void oracle_g1_from_affine(uint64_t r[12], const uint64_t p[8]) { g1_from_affine((g1_jac_t *)r, (const g1_affine_t *)p); }
EXPORT int oracle_g1_eq(const uint64_t a[12], const uint64_t b[12]) { return g1_eq((const g1_jac_t *)a, (const g1_jac_t *)b); }
EXPORT void oracle_g1_batch_normalization(uint64_t *v, size_t n) { g1_batch_normalization((g1_jac_t *)v, n); }
EXPORT void oracle_g2_double(uint64_t p[24]) { g2_double((g2_jac_t *)p); }
EXPORT void oracle_g2_add(uint64_t p[24], const uint64_t o[24]) { g2_add((g2_jac_t *)p, (const g2_jac_t *)o); }
EXPORT void oracle_g2_add_mixed(uint64_t p[24], const uint64_t o[16]) { g2_add_mixed((g2_jac_t *)p, (const g2_affine_t *)o); }
EXPORT void oracle_g2_mul(uint64_t p[24], const uint64_t k[4]) { g2_mul((g2_jac_t *)p, k); }
EXPORT void oracle_g2_to_affine(uint64_t r[16], const uint64_t p[24]) { g2_to_affine((g2_affine_t *)r, (const g2_jac_t *)p); }
EXPORT void oracle_g2_from_affine(uint64_t r[24], const uint64_t p[16]) { g2_from_affine((g2_jac_t *)r, (const g2_affine_t *)p); }
EXPORT int oracle_g2_eq(const uint64_t a[24], const uint64_t b[24]) { return g2_eq((const g2_jac_t *)a, (const g2_jac_t *)b); }
EXPORT void oracle_g2_batch_normalization(uint64_t *v, size_t n) { g2_batch_normalization((g2_jac_t *)v, n); }

/* k_i * P for many scalars, affine output (used to synthesise bases: P_i = k_i * G) */
EXPORT void oracle_g1_mul_many_affine(uint64_t *out_affine, const uint64_t base_affine[8], const uint64_t *ks, size_t n) {
  g1_jac_t *v = (g1_jac_t *)malloc(sizeof(g1_jac_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) { g1_from_affine(&v[i], (const g1_affine_t *)base_affine); g1_mul(&v[i], ks + 4 * i); }
  g1_batch_normalization(v, n);
  for (size_t i = 0; i < n; ++i) g1_to_affine((g1_affine_t *)(out_affine + 8 * i), &v[i]);
  free(v);
}
EXPORT void oracle_g2_mul_many_affine(uint64_t *out_affine, const uint64_t base_affine[16], const uint64_t *ks, size_t n) {
  g2_jac_t *v = (g2_jac_t *)malloc(sizeof(g2_jac_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) { g2_from_affine(&v[i], (const g2_affine_t *)base_affine); g2_mul(&v[i], ks + 4 * i); }
  g2_batch_normalization(v, n);
  for (size_t i = 0; i < n; ++i) g2_to_affine((g2_affine_t *)(out_affine + 16 * i), &v[i]);
  free(v);
}
/* cheap synthetic base table: P_0 = start, P_{i+1} = P_i + step (all affine out).  Used for the
 * large bench inputs where n scalar-muls on the CPU would take too long. */
EXPORT void oracle_g1_arith_progression_affine(uint64_t *out_affine, const uint64_t start_affine[8], const uint64_t step_affine[8], size_t n) {
  g1_jac_t *v = (g1_jac_t *)malloc(sizeof(g1_jac_t) * (n ? n : 1));
  g1_jac_t cur; g1_from_affine(&cur, (const g1_affine_t *)start_affine);
  for (size_t i = 0; i < n; ++i) { v[i] = cur; g1_add_mixed(&cur, (const g1_affine_t *)step_affine); }
  g1_batch_normalization(v, n);
  for (size_t i = 0; i < n; ++i) g1_to_affine((g1_affine_t *)(out_affine + 8 * i), &v[i]);
  free(v);
}
EXPORT void oracle_g2_arith_progression_affine(uint64_t *out_affine, const uint64_t start_affine[16], const uint64_t step_affine[16], size_t n) {
  g2_jac_t *v = (g2_jac_t *)malloc(sizeof(g2_jac_t) * (n ? n : 1));
  g2_jac_t cur; g2_from_affine(&cur, (const g2_affine_t *)start_affine);
  for (size_t i = 0; i < n; ++i) { v[i] = cur; g2_add_mixed(&cur, (const g2_affine_t *)step_affine); }
  g2_batch_normalization(v, n);
  for (size_t i = 0; i < n; ++i) g2_to_affine((g2_affine_t *)(out_affine + 16 * i), &v[i]);
  free(v);
}

/* multiexp: same argument meaning as include/mi355zk.h.  `threads` = Worker cpus (window tasks). */
EXPORT int oracle_g1_multiexp(const uint64_t *bases, size_t n_bases, size_t base_offset, const uint64_t *scalars, size_t n_scalars,
                              const uint32_t *density, size_t density_bits, int threads, uint64_t out_xyz[12]) {
  g1_jac_t out;
  int rc = g1m_multiexp((const g1_affine_t *)bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, threads, &out);
  if (rc == 0) memcpy(out_xyz, &out, sizeof out);
  return rc;
}
EXPORT int oracle_g2_multiexp(const uint64_t *bases, size_t n_bases, size_t base_offset, const uint64_t *scalars, size_t n_scalars,
                              const uint32_t *density, size_t density_bits, int threads, uint64_t out_xyz[24]) {
  g2_jac_t out;
  int rc = g2m_multiexp((const g2_affine_t *)bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, threads, &out);
  if (rc == 0) memcpy(out_xyz, &out, sizeof out);
  return rc;
}
EXPORT uint32_t oracle_multiexp_window_bits(size_t n_scalars) { return g1m_choose_c(n_scalars); }
/* BENCH ONLY: force the window width of the BN254 multiexps (0 = back to the reference's rule ceil(ln n), multiexp.rs:341-345) */
EXPORT void oracle_multiexp_set_window_bits(uint32_t c) { oracle_window_override = c; }

/* powersoftau::utils::dense_multiexp (powersoftau/src/utils.rs:189-292); `cpus` = num_cpus::get() */
EXPORT void oracle_g1_dense_multiexp(const uint64_t *bases, const uint64_t *scalars, size_t n, int cpus, uint64_t out_xyz[12]) {
  g1_jac_t out;
  g1m_dense_multiexp((const g1_affine_t *)bases, scalars, n, cpus, &out);
  memcpy(out_xyz, &out, sizeof out);
}
EXPORT void oracle_g2_dense_multiexp(const uint64_t *bases, const uint64_t *scalars, size_t n, int cpus, uint64_t out_xyz[24]) {
  g2_jac_t out;
  g2m_dense_multiexp((const g2_affine_t *)bases, scalars, n, cpus, &out);
  memcpy(out_xyz, &out, sizeof out);
}

/* naive sum_i k_i * P_i via mul_assign (the reference tests' `naive_multiexp`, multiexp.rs:486-499) */
EXPORT void oracle_g1_naive_multiexp(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_xyz[12]) {
  g1_jac_t acc; g1_set_zero(&acc);
  for (size_t i = 0; i < n; ++i) { g1_jac_t t; g1_from_affine(&t, (const g1_affine_t *)(bases + 8 * i)); g1_mul(&t, scalars + 4 * i); g1_add(&acc, &t); }
  memcpy(out_xyz, &acc, sizeof acc);
}
EXPORT void oracle_g2_naive_multiexp(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_xyz[24]) {
  g2_jac_t acc; g2_set_zero(&acc);
  for (size_t i = 0; i < n; ++i) { g2_jac_t t; g2_from_affine(&t, (const g2_affine_t *)(bases + 16 * i)); g2_mul(&t, scalars + 4 * i); g2_add(&acc, &t); }
  memcpy(out_xyz, &acc, sizeof acc);
}

/* Fr FFT family.  log_cpus selects serial (log_n <= log_cpus is serial per best_fft) vs parallel shape. */
EXPORT void oracle_fr_serial_fft(uint64_t *a, uint32_t log_n, const uint64_t omega[4]) { frfft_serial_fft((fe_t *)a, (const fe_t *)omega, log_n); }
EXPORT void oracle_fr_parallel_fft(uint64_t *a, uint32_t log_n, const uint64_t omega[4], uint32_t log_cpus) { frfft_parallel_fft((fe_t *)a, (const fe_t *)omega, log_n, log_cpus); }
EXPORT int oracle_fr_domain(uint32_t log_n, uint64_t omega[4], uint64_t omegainv[4], uint64_t geninv[4], uint64_t minv[4]) {
  frfft_domain_t d;
  if (frfft_domain_init(&d, log_n)) return -1;
  memcpy(omega, &d.omega, 32); memcpy(omegainv, &d.omegainv, 32); memcpy(geninv, &d.geninv, 32); memcpy(minv, &d.minv, 32);
  return 0;
}
/* op: 0 fft, 1 ifft, 2 coset_fft, 3 icoset_fft.  log_cpus >= log_n -> serial_fft (normative). */
EXPORT int oracle_fr_domain_op(uint64_t *a, uint32_t log_n, int op, uint32_t log_cpus) {
  frfft_domain_t d;
  if (frfft_domain_init(&d, log_n)) return -1;
  switch (op) {
    case 0: frfft_fft((fe_t *)a, &d, log_cpus); break;
    case 1: frfft_ifft((fe_t *)a, &d, log_cpus); break;
    case 2: frfft_coset_fft((fe_t *)a, &d, log_cpus); break;
    case 3: frfft_icoset_fft((fe_t *)a, &d, log_cpus); break;
    default: return -2;
  }
  return 0;
}

/* DummyEngine instances */
EXPORT int oracle_dummy_multiexp(const uint32_t *bases, size_t n_bases, size_t base_offset, const uint64_t *scalars, size_t n_scalars,
                                 const uint32_t *density, size_t density_bits, uint32_t *out) {
  return dm_multiexp(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, 1, out);
}
EXPORT int oracle_dummy_domain_op(uint32_t *a, uint32_t log_n, int op, uint32_t log_cpus) {
  dfft_domain_t d;
  if (dfft_domain_init(&d, log_n)) return -1;
  switch (op) {
    case 0: dfft_fft(a, &d, log_cpus); break;
    case 1: dfft_ifft(a, &d, log_cpus); break;
    case 2: dfft_coset_fft(a, &d, log_cpus); break;
    case 3: dfft_icoset_fft(a, &d, log_cpus); break;
    default: return -2;
  }
  return 0;
}
/* EvaluationDomain<Point<G>> ops on Jacobian points, then batch_normalization (what prepare_phase2.rs:102-131 does
 * before it writes the Lagrange-basis points): pts = 2^log_n x (12 | 24) u64, result normalised (z = one or infinity). */
EXPORT int oracle_g1_point_domain_op(uint64_t *pts, uint32_t log_n, int op, uint32_t log_cpus) {
  g1fft_domain_t d;
  if (g1fft_domain_init(&d, log_n)) return -1;
  if (op == 0) g1fft_fft((g1_jac_t *)pts, &d, log_cpus);
  else if (op == 1) g1fft_ifft((g1_jac_t *)pts, &d, log_cpus);
  else return -2;
  g1_batch_normalization((g1_jac_t *)pts, (size_t)1 << log_n);
  return 0;
}
EXPORT int oracle_g2_point_domain_op(uint64_t *pts, uint32_t log_n, int op, uint32_t log_cpus) {
  g2fft_domain_t d;
  if (g2fft_domain_init(&d, log_n)) return -1;
  if (op == 0) g2fft_fft((g2_jac_t *)pts, &d, log_cpus);
  else if (op == 1) g2fft_ifft((g2_jac_t *)pts, &d, log_cpus);
  else return -2;
  g2_batch_normalization((g2_jac_t *)pts, (size_t)1 << log_n);
  return 0;
}
EXPORT uint32_t oracle_dummy_domain_omega(uint32_t log_n) { dfft_domain_t d; if (dfft_domain_init(&d, log_n)) return 0; return d.omega; }

/* ---- point codecs (codec.h): n records; returns the first failing record's code and index, or 0 */
EXPORT int oracle_g1_decode(uint64_t *out_affine, const uint8_t *in, size_t n, int compressed, int checked, long long *err_index) {
  size_t sz = compressed ? 32 : 64;
  for (size_t i = 0; i < n; ++i) {
    int rc = g1_decode((g1_affine_t *)(out_affine + 8 * i), in + sz * i, compressed, checked);
    if (rc) { if (err_index) *err_index = (long long)i; return rc; }
  }
  return 0;
}
EXPORT void oracle_g1_encode(uint8_t *out, const uint64_t *affine, size_t n, int compressed) {
  size_t sz = compressed ? 32 : 64;
  for (size_t i = 0; i < n; ++i) g1_encode(out + sz * i, (const g1_affine_t *)(affine + 8 * i), compressed);
}
EXPORT int oracle_g2_decode(uint64_t *out_affine, const uint8_t *in, size_t n, int compressed, int checked, long long *err_index) {
  size_t sz = compressed ? 64 : 128;
  for (size_t i = 0; i < n; ++i) {
    int rc = g2_decode((g2_affine_t *)(out_affine + 16 * i), in + sz * i, compressed, checked);
    if (rc) { if (err_index) *err_index = (long long)i; return rc; }
  }
  return 0;
}
EXPORT void oracle_g2_encode(uint8_t *out, const uint64_t *affine, size_t n, int compressed) {
  size_t sz = compressed ? 64 : 128;
  for (size_t i = 0; i < n; ++i) g2_encode(out + sz * i, (const g2_affine_t *)(affine + 16 * i), compressed);
}
EXPORT int oracle_fq_sqrt(uint64_t r[4], const uint64_t a[4]) { return fq_sqrt((fe_t *)r, (const fe_t *)a); }
EXPORT int oracle_fq2_sqrt(uint64_t r[8], const uint64_t a[8]) { return fq2_sqrt_ref((fe2_t *)r, (const fe2_t *)a); }
EXPORT void oracle_g2_coeff_b(uint64_t r[8]) { g2_coeff_b((fe2_t *)r); }
