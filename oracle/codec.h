/*
 * oracle/codec.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The point encodings of the reference ("point codecs"): pairing/src/bn256/ec.rs
 *   G1Uncompressed :763-845   G1Compressed :867-946   G2Uncompressed :1136-1229   G2Compressed :1255-1344
 *   get_point_from_x :110-131   is_on_curve :133-148
 * and the square roots they rest on:  Fq::sqrt (ff_derive_ce 0.5.1, q = 3 mod 4: a1 = a^((q-3)/4), a0 = a1^2 a,
 * a0 == -1 -> None, else a1 a)  and  Fq2::sqrt pairing/src/bn256/fq2.rs:211-261 (Algorithm 9 of eprint 2012/685).
 *
 * REFERENCE QUIRK, reproduced on purpose: fq2.rs:231-241 compares against `NEGATIVE_ONE`
 * (fq.rs:434-439), which is -(2^256 mod r) for the SCALAR modulus r, not -(2^256 mod q); so Fq2::sqrt of the
 * reference never answers None and never takes its "alpha == -1" branch.  The constant is derived below from FR.
 *
 * Wire format: big-endian canonical coordinates (x, then y; for Fq2 c1 before c0); the two top bits of byte 0 are
 * flags: bit 7 = "y is the lexicographically larger root" (compressed only), bit 6 = point at infinity.
 * Decoded points are raw affine records (Montgomery limbs, all-zero = infinity), the layout every kernel uses.
 * Return codes mirror GroupDecodingError (pairing/src/lib.rs): 0 ok, 4 NotOnCurve, 6 CoordinateDecodingError,
 * 7 UnexpectedCompressionMode, 8 UnexpectedInformation.
 */
#ifndef ORACLE_CODEC_H
#define ORACLE_CODEC_H

enum { DEC_OK = 0, DEC_NOT_ON_CURVE = 4, DEC_COORD = 6, DEC_COMPRESSION_MODE = 7, DEC_UNEXPECTED_INFO = 8 };

static void be_read(uint64_t l[4], const uint8_t *b) { /* FqRepr::read_be: most significant limb first */
  for (int i = 0; i < 4; ++i) {
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v = (v << 8) | b[8 * i + k];
    l[3 - i] = v;
  }
}
static void be_write(uint8_t *b, const uint64_t l[4]) {
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) b[8 * i + k] = (uint8_t)(l[3 - i] >> (56 - 8 * k));
}
/* Fq::from_repr: Err for >= q (ec.rs:817-822) */
static int fq_from_be(fe_t *r, const uint8_t *b) {
  uint64_t c[4];
  be_read(c, b);
  if (limbs_geq(c, FQ.p)) return 0;
  fe_from_canonical(&FQ, r, c);
  return 1;
}
static void fq_to_be(uint8_t *b, const fe_t *a) {
  uint64_t c[4];
  fe_to_canonical(&FQ, c, a);
  be_write(b, c);
}
/* Ord for Fq = order of into_repr(); for Fq2: c1 first (fq2.rs:20-31) */
static int fq_cmp(const fe_t *a, const fe_t *b) {
  uint64_t x[4], y[4];
  fe_to_canonical(&FQ, x, a);
  fe_to_canonical(&FQ, y, b);
  for (int i = 3; i >= 0; --i)
    if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}
static int fq2_cmp(const fe2_t *a, const fe2_t *b) {
  int c = fq_cmp(&a->c1, &b->c1);
  return c ? c : fq_cmp(&a->c0, &b->c0);
}

static void q_exponent(uint64_t e[4], unsigned sub, unsigned shift) { /* (q - sub) >> shift */
  uint64_t s[4] = {sub, 0, 0, 0};
  limbs_sub(e, FQ.p, s);
  for (unsigned k = 0; k < shift; ++k)
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
}

/* ff_derive sqrt for q = 3 mod 4 */
static int fq_sqrt(fe_t *r, const fe_t *a) {
  uint64_t e[4];
  q_exponent(e, 3, 2);
  fe_t a1, a0, one, neg1;
  fe_pow(&FQ, &a1, a, e, 4);
  fe_sqr(&FQ, &a0, &a1);
  fe_mul(&FQ, &a0, &a0, a);
  fe_one(&FQ, &one);
  fe_neg(&FQ, &neg1, &one);
  if (fe_eq(&a0, &neg1)) return 0;
  fe_mul(&FQ, r, &a1, a);
  return 1;
}

static void fe2_pow(fe2_t *r, const fe2_t *a, const uint64_t *e, int nlimbs) { /* MSB first, like Field::pow */
  fe2_t res;
  fe2_one(&res);
  for (int i = nlimbs * 64 - 1; i >= 0; --i) {
    fe2_sqr(&res, &res);
    if ((e[i >> 6] >> (i & 63)) & 1) fe2_mul(&res, &res, a);
  }
  *r = res;
}

/* fq2.rs:211-261, with the reference's NEGATIVE_ONE (see header) */
static int fq2_sqrt_ref(fe2_t *r, const fe2_t *a) {
  if (fe2_is_zero(a)) { fe2_zero(r); return 1; }
  uint64_t e[4];
  fe2_t a1, alpha, a0, neg1;
  q_exponent(e, 3, 2);
  fe2_pow(&a1, a, e, 4);
  fe2_sqr(&alpha, &a1);
  fe2_mul(&alpha, &alpha, a);
  a0 = alpha;
  fe_neg(&FQ, &a0.c1, &a0.c1); /* frobenius_map(1): c1 *= (-1)^((q-1)/2) = -1   (fq2.rs:201-203, fq.rs:96-103) */
  fe2_mul(&a0, &a0, &alpha);
  fe_t one_r;
  fe_one(&FR, &one_r);
  fe_neg(&FR, &neg1.c0, &one_r); /* the quirk: -(2^256 mod r) mod r used as Fq limbs */
  fe_zero(&neg1.c1);
  if (fe2_eq(&a0, &neg1)) return 0;
  fe2_mul(&a1, &a1, a);
  if (fe2_eq(&alpha, &neg1)) {
    fe2_t u;
    fe_zero(&u.c0);
    fe_one(&FQ, &u.c1);
    fe2_mul(&a1, &a1, &u);
  } else {
    fe2_t one;
    fe2_one(&one);
    fe2_add(&alpha, &alpha, &one);
    q_exponent(e, 1, 1);
    fe2_pow(&alpha, &alpha, e, 4);
    fe2_mul(&a1, &a1, &alpha);
  }
  *r = a1;
  return 1;
}

static void g1_coeff_b(fe_t *b) { /* 3 (fq.rs:11-16) */
  fe_t one;
  fe_one(&FQ, &one);
  fe_add(&FQ, b, &one, &one);
  fe_add(&FQ, b, b, &one);
}
static void g2_coeff_b(fe2_t *b) { /* 3 / (9 + u)  (fq.rs:18-31 holds the value; derived here) */
  fe_t one, three, nine;
  fe_one(&FQ, &one);
  g1_coeff_b(&three);
  fe_add(&FQ, &nine, &three, &three);
  fe_add(&FQ, &nine, &nine, &three);
  fe2_t xi = {nine, one}, inv;
  fe2_inv(&inv, &xi);
  fe2_t t = {three, {{0, 0, 0, 0}}};
  fe2_mul(b, &t, &inv);
}

static int g1_on_curve(const g1_affine_t *p) { /* ec.rs:133-148 */
  if (g1_affine_is_zero(p)) return 1;
  fe_t y2, x3b, b;
  fe_sqr(&FQ, &y2, &p->y);
  fe_sqr(&FQ, &x3b, &p->x);
  fe_mul(&FQ, &x3b, &x3b, &p->x);
  g1_coeff_b(&b);
  fe_add(&FQ, &x3b, &x3b, &b);
  return fe_eq(&y2, &x3b);
}
static int g2_on_curve(const g2_affine_t *p) {
  if (g2_affine_is_zero(p)) return 1;
  fe2_t y2, x3b, b;
  fe2_sqr(&y2, &p->y);
  fe2_sqr(&x3b, &p->x);
  fe2_mul(&x3b, &x3b, &p->x);
  g2_coeff_b(&b);
  fe2_add(&x3b, &x3b, &b);
  return fe2_eq(&y2, &x3b);
}

/* ec.rs:110-131 */
static int g1_point_from_x(g1_affine_t *p, const fe_t *x, int greatest) {
  fe_t x3b, b, y, negy;
  fe_sqr(&FQ, &x3b, x);
  fe_mul(&FQ, &x3b, &x3b, x);
  g1_coeff_b(&b);
  fe_add(&FQ, &x3b, &x3b, &b);
  if (!fq_sqrt(&y, &x3b)) return 0;
  fe_neg(&FQ, &negy, &y);
  p->x = *x;
  p->y = ((fq_cmp(&y, &negy) < 0) ^ (greatest != 0)) ? y : negy;
  return 1;
}
static int g2_point_from_x(g2_affine_t *p, const fe2_t *x, int greatest) {
  fe2_t x3b, b, y, negy;
  fe2_sqr(&x3b, x);
  fe2_mul(&x3b, &x3b, x);
  g2_coeff_b(&b);
  fe2_add(&x3b, &x3b, &b);
  if (!fq2_sqrt_ref(&y, &x3b)) return 0;
  fe2_neg(&negy, &y);
  p->x = *x;
  p->y = ((fq2_cmp(&y, &negy) < 0) ^ (greatest != 0)) ? y : negy;
  return 1;
}

static int all_zero_after_mask(const uint8_t *b, size_t n) {
  if (b[0] & 0x3f) return 0;
  for (size_t i = 1; i < n; ++i)
    if (b[i]) return 0;
  return 1;
}

/* into_affine (checked != 0) / into_affine_unchecked */
static int g1_decode(g1_affine_t *out, const uint8_t *in, int compressed, int checked) {
  uint8_t b[64];
  size_t n = compressed ? 32 : 64;
  memcpy(b, in, n);
  memset(out, 0, sizeof *out);
  if (b[0] & 0x40) return all_zero_after_mask(b, n) ? DEC_OK : DEC_UNEXPECTED_INFO;
  int greatest = (b[0] & 0x80) != 0;
  if (!compressed && greatest) return DEC_UNEXPECTED_INFO; /* ec.rs:797-801 */
  b[0] &= 0x3f;
  fe_t x, y;
  if (!fq_from_be(&x, b)) return DEC_COORD;
  if (compressed) {
    g1_affine_t p;
    if (!g1_point_from_x(&p, &x, greatest)) return DEC_NOT_ON_CURVE;
    *out = p;
    return DEC_OK;
  }
  if (!fq_from_be(&y, b + 32)) return DEC_COORD;
  g1_affine_t p = {x, y};
  if (checked && !g1_on_curve(&p)) return DEC_NOT_ON_CURVE;
  *out = p;
  return DEC_OK;
}
static void g1_encode(uint8_t *out, const g1_affine_t *p, int compressed) {
  size_t n = compressed ? 32 : 64;
  memset(out, 0, n);
  if (g1_affine_is_zero(p)) { out[0] |= 0x40; return; }
  fq_to_be(out, &p->x);
  if (!compressed) { fq_to_be(out + 32, &p->y); return; }
  fe_t negy;
  fe_neg(&FQ, &negy, &p->y);
  if (fq_cmp(&p->y, &negy) > 0) out[0] |= 0x80;
}
static int g2_decode(g2_affine_t *out, const uint8_t *in, int compressed, int checked) {
  uint8_t b[128];
  size_t n = compressed ? 64 : 128;
  memcpy(b, in, n);
  memset(out, 0, sizeof *out);
  if (!compressed && (b[0] & 0x80)) return DEC_COMPRESSION_MODE; /* ec.rs:1158-1161, before the infinity flag */
  if (b[0] & 0x40) return all_zero_after_mask(b, n) ? DEC_OK : DEC_UNEXPECTED_INFO;
  int greatest = (b[0] & 0x80) != 0;
  b[0] &= 0x3f;
  fe2_t x, y;
  if (!fq_from_be(&x.c0, b + 32)) return DEC_COORD; /* c1 is first on the wire; c0 is converted first (ec.rs:1195-1202) */
  if (!fq_from_be(&x.c1, b)) return DEC_COORD;
  if (compressed) {
    g2_affine_t p;
    if (!g2_point_from_x(&p, &x, greatest)) return DEC_NOT_ON_CURVE;
    *out = p;
    return DEC_OK;
  }
  if (!fq_from_be(&y.c0, b + 96)) return DEC_COORD;
  if (!fq_from_be(&y.c1, b + 64)) return DEC_COORD;
  g2_affine_t p = {x, y};
  if (checked && !g2_on_curve(&p)) return DEC_NOT_ON_CURVE;
  *out = p;
  return DEC_OK;
}
static void g2_encode(uint8_t *out, const g2_affine_t *p, int compressed) {
  size_t n = compressed ? 64 : 128;
  memset(out, 0, n);
  if (g2_affine_is_zero(p)) { out[0] |= 0x40; return; }
  fq_to_be(out, &p->x.c1);
  fq_to_be(out + 32, &p->x.c0);
  if (!compressed) { fq_to_be(out + 64, &p->y.c1); fq_to_be(out + 96, &p->y.c0); return; }
  fe2_t negy;
  fe2_neg(&negy, &p->y);
  if (fq2_cmp(&p->y, &negy) > 0) out[0] |= 0x80;
}

#endif
