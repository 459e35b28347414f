/*
 * oracle/field.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product library).
 *
 * 4x64-bit Montgomery prime-field arithmetic for BN254 Fq and Fr, restating the semantics of the
 * `#[derive(PrimeField)]` expansion the reference relies on (pairing/src/bn256/fq.rs:4-7,
 * pairing/src/bn256/fr.rs:3-6).  The expansion itself lives in the un-vendored crates
 * ff_ce 0.7.1 / ff_derive_ce 0.5.1 (powersoftau/Cargo.lock:147-164), so this file restates the
 * published algorithm (CIOS Montgomery multiplication, R = 2^256, fully reduced results) and is
 * anchored on the literals the reference does contain: G1_GENERATOR_X == R mod q
 * (pairing/src/bn256/fq.rs:39-44), B_COEFF == 3R mod q (fq.rs:11-16), Fr::S == 28 (fr.rs:31-34).
 * Because elements are always fully reduced, any correct implementation produces identical limbs.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct { uint64_t l[4]; } fe_t; /* little-endian limbs; Montgomery form unless stated */

typedef struct {
  uint64_t p[4];   /* modulus */
  uint64_t r[4];   /* R mod p  (== one()) */
  uint64_t r2[4];  /* R^2 mod p */
  uint64_t inv;    /* -p^{-1} mod 2^64 */
} modulus_t;

static const modulus_t FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    0x87d20782e4866389ULL};

static const modulus_t FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    0xc2e1f593efffffffULL};

static inline int fe_is_zero(const fe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe_t *a, const fe_t *b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline void fe_zero(fe_t *a) { memset(a, 0, sizeof *a); }
static inline void fe_one(const modulus_t *m, fe_t *a) { memcpy(a->l, m->r, 32); }

/* a >= b on raw limbs (PrimeFieldRepr Ord: most-significant limb first) */
static inline int limbs_geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}

static inline uint64_t limbs_sub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)t;
    borrow = (t >> 64) & 1;
  }
  return (uint64_t)borrow;
}

static inline uint64_t limbs_add(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 carry = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a[i] + b[i] + carry;
    r[i] = (uint64_t)t;
    carry = t >> 64;
  }
  return (uint64_t)carry;
}

static inline void fe_add(const modulus_t *m, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t t[4];
  limbs_add(t, a->l, b->l); /* p < 2^254 so no carry out */
  if (limbs_geq(t, m->p)) limbs_sub(t, t, m->p);
  memcpy(r->l, t, 32);
}

static inline void fe_double(const modulus_t *m, fe_t *r, const fe_t *a) { fe_add(m, r, a, a); }

static inline void fe_sub(const modulus_t *m, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t t[4];
  if (limbs_sub(t, a->l, b->l)) limbs_add(t, t, m->p);
  memcpy(r->l, t, 32);
}

static inline void fe_neg(const modulus_t *m, fe_t *r, const fe_t *a) {
  if (fe_is_zero(a)) { fe_zero(r); return; }
  uint64_t t[4];
  limbs_sub(t, m->p, a->l);
  memcpy(r->l, t, 32);
}

/* CIOS Montgomery multiplication: r = a*b*R^{-1} mod p, fully reduced. */
static inline void fe_mul(const modulus_t *m, fe_t *r, const fe_t *a, const fe_t *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t k = t[0] * m->inv;
    c = (u128)k * m->p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)k * m->p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || limbs_geq(t, m->p)) limbs_sub(t, t, m->p);
  memcpy(r->l, t, 32);
}

static inline void fe_sqr(const modulus_t *m, fe_t *r, const fe_t *a) { fe_mul(m, r, a, a); }

/* canonical integer (< p) -> Montgomery form  (PrimeField::from_repr, Appendix A) */
static inline void fe_from_canonical(const modulus_t *m, fe_t *r, const uint64_t c[4]) {
  fe_t a, r2;
  memcpy(a.l, c, 32);
  memcpy(r2.l, m->r2, 32);
  fe_mul(m, r, &a, &r2);
}

/* Montgomery form -> canonical integer  (PrimeField::into_repr) */
static inline void fe_to_canonical(const modulus_t *m, uint64_t c[4], const fe_t *a) {
  fe_t one = {{1, 0, 0, 0}}, t;
  fe_mul(m, &t, a, &one);
  memcpy(c, t.l, 32);
}

/* a^e with e given as limbs, MSB-first square-and-multiply (Field::pow, Appendix A) */
static inline void fe_pow(const modulus_t *m, fe_t *r, const fe_t *a, const uint64_t *e, int nlimbs) {
  fe_t res;
  fe_one(m, &res);
  int found_one = 0;
  for (int i = nlimbs * 64 - 1; i >= 0; --i) {
    int bit = (int)((e[i / 64] >> (i % 64)) & 1);
    if (found_one) fe_sqr(m, &res, &res); else found_one = bit;
    if (bit) fe_mul(m, &res, &res, a);
  }
  *r = res;
}

/* inverse via Fermat (a^(p-2)); returns 0 for a == 0 (Field::inverse -> None) */
static inline int fe_inv(const modulus_t *m, fe_t *r, const fe_t *a) {
  if (fe_is_zero(a)) return 0;
  uint64_t e[4], two[4] = {2, 0, 0, 0};
  limbs_sub(e, m->p, two);
  fe_pow(m, r, a, e, 4);
  return 1;
}

/* ---- Fq2 = Fq[u]/(u^2+1), pairing/src/bn256/fq2.rs ---- */
typedef struct { fe_t c0, c1; } fe2_t;

static inline int fe2_is_zero(const fe2_t *a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static inline int fe2_eq(const fe2_t *a, const fe2_t *b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static inline void fe2_zero(fe2_t *a) { memset(a, 0, sizeof *a); }
static inline void fe2_one(fe2_t *a) { fe_one(&FQ, &a->c0); fe_zero(&a->c1); } /* fq2.rs:120-125 */
static inline void fe2_add(fe2_t *r, const fe2_t *a, const fe2_t *b) { fe_add(&FQ, &r->c0, &a->c0, &b->c0); fe_add(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void fe2_sub(fe2_t *r, const fe2_t *a, const fe2_t *b) { fe_sub(&FQ, &r->c0, &a->c0, &b->c0); fe_sub(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void fe2_double(fe2_t *r, const fe2_t *a) { fe2_add(r, a, a); }
static inline void fe2_neg(fe2_t *r, const fe2_t *a) { fe_neg(&FQ, &r->c0, &a->c0); fe_neg(&FQ, &r->c1, &a->c1); }

/* Karatsuba, fq2.rs:167-180 */
static inline void fe2_mul(fe2_t *r, const fe2_t *a, const fe2_t *b) {
  fe_t aa, bb, o, s;
  fe_mul(&FQ, &aa, &a->c0, &b->c0);
  fe_mul(&FQ, &bb, &a->c1, &b->c1);
  fe_add(&FQ, &o, &b->c0, &b->c1);
  fe_add(&FQ, &s, &a->c1, &a->c0);
  fe_mul(&FQ, &s, &s, &o);
  fe_sub(&FQ, &s, &s, &aa);
  fe_sub(&FQ, &s, &s, &bb);
  r->c1 = s;
  fe_sub(&FQ, &r->c0, &aa, &bb);
}

/* complex squaring, fq2.rs:131-145 */
static inline void fe2_sqr(fe2_t *r, const fe2_t *a) {
  fe_t ab, c0c1, c0;
  fe_mul(&FQ, &ab, &a->c0, &a->c1);
  fe_add(&FQ, &c0c1, &a->c0, &a->c1);
  fe_neg(&FQ, &c0, &a->c1);
  fe_add(&FQ, &c0, &c0, &a->c0);
  fe_mul(&FQ, &c0, &c0, &c0c1);
  fe_sub(&FQ, &c0, &c0, &ab);
  fe_add(&FQ, &r->c1, &ab, &ab);
  fe_add(&FQ, &r->c0, &c0, &ab);
}

/* fq2.rs:182-199 */
static inline int fe2_inv(fe2_t *r, const fe2_t *a) {
  fe_t t0, t1, t;
  fe_sqr(&FQ, &t1, &a->c1);
  fe_sqr(&FQ, &t0, &a->c0);
  fe_add(&FQ, &t0, &t0, &t1);
  if (!fe_inv(&FQ, &t, &t0)) return 0;
  fe_mul(&FQ, &r->c0, &a->c0, &t);
  fe_mul(&FQ, &t1, &a->c1, &t);
  fe_neg(&FQ, &r->c1, &t1);
  return 1;
}

#endif
