// GLV scalar decomposition for BN254 G1, used by the per-point scalar multiplications (batch_exp and what is built on it: the
// terms of the QAP sums, the butterflies of the point FFT).  The curve y^2 = x^3 + 3 over Fq has the endomorphism
//     phi(x, y) = (beta x, y) = lambda (x, y),      beta^3 = 1 in Fq,  lambda^2 + lambda + 1 = 0 mod r,
// so k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^128: half the doublings of a 254-bit ladder.  For a point of the order-r group the
// group element -- hence the affine output the reference's batch_exp / batch_normalization leave (ec.rs:251-299) -- does not depend on
// the chain that produced it.  G1: E(Fq) has prime order, every on-curve point qualifies.  G2: the twist has a cofactor, psi(P) =
// mu P holds in the order-r subgroup ONLY, and neither the reference's decoders nor ours test subgroup membership: for an on-curve
// point with a cofactor component the split chain does NOT give k P (include/mi355zk.h states the precondition).
//   lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd,  beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
//   lattice basis of {(x, y): x + y lambda = 0 mod r}:  v1 = (a1, -|b1|),  v2 = (a2, b2)
//   k1 = k - c1 a1 - c2 a2,   k2 = c1 |b1| - c2 b2,   c1 = floor(k g1 / 2^256),  c2 = floor(k g2 / 2^256),
//   g1 = floor(2^256 b2 / r),  g2 = floor(2^256 |b1| / r).
// ANY integers c1, c2 give k1 + k2 lambda = k (mod r) exactly; the floors (instead of roundings) only cost a bit of size:
// |k1|, |k2| < 2^128 (tests/test_glv_host.py: identity and size on random and edge scalars against big integers).
#pragma once

#include "field.hpp"

namespace zk {

struct GlvSplit {
  uint32_t k1[5], k2[5];  // magnitudes, < 2^128 (the fifth limb is headroom)
  bool neg1, neg2;
};

namespace glv_detail {
// out[0 .. NA+NB) = a * b   (32-bit limbs, little endian)
template <int NA, int NB>
ZK_HD void mul_limbs(const uint32_t* a, const uint32_t* b, uint32_t* out) {
#pragma unroll
  for (int i = 0; i < NA + NB; ++i) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    out[i + NB] = (uint32_t)carry;
  }
}
// r = a - b on N limbs (two's complement); returns the sign (1: negative) and leaves |a - b| in r
template <int N>
ZK_HD bool sub_abs(const uint32_t* a, const uint32_t* b, uint32_t* r) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const uint64_t t = (uint64_t)a[i] - b[i] - borrow;
    r[i] = (uint32_t)t;
    borrow = (t >> 32) & 1u;
  }
  const bool neg = borrow != 0;
  if (neg) {
    uint64_t carry = 1;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint64_t t = (uint64_t)(~r[i]) + carry;
      r[i] = (uint32_t)t;
      carry = t >> 32;
    }
  }
  return neg;
}
}  // namespace glv_detail

// beta in the memory format (beta * 2^256 mod q)
ZK_HD Fq glv_beta() {
  Fq b;
  const uint32_t l[8] = {0xd782e155u, 0x71930c11u, 0xffbe3323u, 0xa6bb947cu, 0xd4741444u, 0xaa303344u, 0x26594943u, 0x2c3b3f0du};
#pragma unroll
  for (int i = 0; i < 8; ++i) b.l[i] = l[i];
  return b;
}

// k: canonical scalar (< r), 8 x 32-bit limbs
ZK_HD GlvSplit glv_split(const uint32_t k[8]) {
  using namespace glv_detail;
  const uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
  const uint32_t G2[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
  const uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};                                  // = b2
  const uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
  const uint32_t B1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};        // |b1|
  uint32_t t11[11], t13[13];
  mul_limbs<8, 3>(k, G1, t11);
  mul_limbs<8, 5>(k, G2, t13);
  const uint32_t* c1 = t11 + 8;   // 3 limbs
  const uint32_t* c2 = t13 + 8;   // 5 limbs
  // k1 = k - (c1 a1 + c2 a2) on 9 limbs
  uint32_t p1[5], p2[9], sum[9], kk[9], d[9];
  mul_limbs<3, 2>(c1, A1, p1);
  mul_limbs<5, 4>(c2, A2, p2);
  {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const uint64_t t = (uint64_t)p2[i] + (i < 5 ? p1[i] : 0u) + carry;
      sum[i] = (uint32_t)t;
      carry = t >> 32;
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) kk[i] = i < 8 ? k[i] : 0u;
  GlvSplit s;
  s.neg1 = sub_abs<9>(kk, sum, d);
#pragma unroll
  for (int i = 0; i < 5; ++i) s.k1[i] = d[i];
  // k2 = c1 |b1| - c2 b2 on 8 limbs
  uint32_t q1[7], q2[7], e1[8], e2[8], e[8];
  mul_limbs<3, 4>(c1, B1, q1);
  mul_limbs<5, 2>(c2, A1, q2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    e1[i] = i < 7 ? q1[i] : 0u;
    e2[i] = i < 7 ? q2[i] : 0u;
  }
  s.neg2 = sub_abs<8>(e1, e2, e);
#pragma unroll
  for (int i = 0; i < 5; ++i) s.k2[i] = e[i];
  return s;
}

// ---- G2: the twist has psi = (untwist) o Frobenius o (twist),  psi(x, y) = (cx conj(x), cy conj(y)),  cx = xi^((q-1)/3),
// cy = xi^((q-1)/2), xi = 9 + u, which acts on the order-r subgroup as multiplication by  mu = q mod r = 6 x^2 (127 bits; x the BN
// parameter).  mu^2 ~ r, so the split is a division:  k = k1 + k2 mu,  k2 = floor(k g / 2^256) with g = floor(2^256 / mu) (the true
// quotient or one less),  k1 = k - k2 mu:  both NON-NEGATIVE and < 2^128 (tests/test_glv_host.py).
struct Glv2Split {
  uint32_t k1[5], k2[5];
};
ZK_HD Glv2Split glv2_split(const uint32_t k[8]) {
  using namespace glv_detail;
  const uint32_t MU[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};
  const uint32_t G[5] = {0xc8e01941u, 0x2cb62031u, 0xa773d2d5u, 0x4ccef014u, 0x00000002u};
  uint32_t t13[13], prod[9], kk[9], d[9];
  mul_limbs<8, 5>(k, G, t13);
  Glv2Split s;
#pragma unroll
  for (int i = 0; i < 5; ++i) s.k2[i] = t13[8 + i];
  mul_limbs<5, 4>(s.k2, MU, prod);
#pragma unroll
  for (int i = 0; i < 9; ++i) kk[i] = i < 8 ? k[i] : 0u;
  (void)sub_abs<9>(kk, prod, d);   // never negative: k2 <= floor(k / mu)
#pragma unroll
  for (int i = 0; i < 5; ++i) s.k1[i] = d[i];
  return s;
}
// the constants of psi in the memory format (c * 2^256 mod q, per Fq2 component)
ZK_HD Fq2 glv2_cx() {
  const uint32_t a[8] = {0x4563ab30u, 0xb5773b10u, 0xa9aa6454u, 0x347f91c8u, 0x242e0991u, 0x7a007127u, 0x118214ecu, 0x1956bcd8u};
  const uint32_t b[8] = {0xa0aa4757u, 0x6e849f1eu, 0x89f89141u, 0xaa1c7b6du, 0xfae0ca3au, 0xb6e713cdu, 0x4e82ebc3u, 0x26694fbbu};
  Fq2 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r.c0.l[i] = a[i]; r.c1.l[i] = b[i]; }
  return r;
}
ZK_HD Fq2 glv2_cy() {
  const uint32_t a[8] = {0x2936b629u, 0xe4bbdd0cu, 0xe133bacbu, 0xbb30f162u, 0xf9645366u, 0x31a9d1b6u, 0xa500f8ddu, 0x253570beu};
  const uint32_t b[8] = {0x5ffe77c7u, 0xa1d77ce4u, 0x7826d1dbu, 0x07affd11u, 0xbb7edc6bu, 0x6d16bd27u, 0x85defeccu, 0x2c872002u};
  Fq2 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r.c0.l[i] = a[i]; r.c1.l[i] = b[i]; }
  return r;
}
// the twist's coefficient b' = 3 / (9 + u) in the memory format (pairing/src/bn256/fq.rs:18-31 B_COEFF_FQ2, the same limbs)
ZK_HD Fq2 g2_coeff_b() {
  const uint32_t a[8] = {0x77b802a8u, 0x3bf938e3u, 0x3633535du, 0x020b1b27u, 0x49755260u, 0x26b7edf0u, 0x4384a86du, 0x2514c632u};
  const uint32_t b[8] = {0xd1dcff67u, 0x38e7ecccu, 0x93ce0d3eu, 0x65f0b37du, 0x22ac00aau, 0xd749d0ddu, 0x4a688d4du, 0x0141b9ceu};
  Fq2 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) { r.c0.l[i] = a[i]; r.c1.l[i] = b[i]; }
  return r;
}

// non-adjacent form of a magnitude m < 2^159 (5 limbs): digit j = bit_{j+1}(3m) - bit_{j+1}(m);  pos / neg: 5 limbs + 1 bit (6 words)
ZK_HD void glv_naf(const uint32_t m[5], uint32_t pos[6], uint32_t neg[6]) {
  uint32_t m3[7], mm[7];
  uint64_t carry = 0;
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    const uint64_t t = (uint64_t)(l < 5 ? m[l] : 0u) * 3u + carry;
    m3[l] = (uint32_t)t;
    carry = t >> 32;
    mm[l] = l < 5 ? m[l] : 0u;
  }
  m3[6] = 0;
  mm[6] = 0;
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    const uint32_t a = (m3[l] >> 1) | (m3[l + 1] << 31);
    const uint32_t b = (mm[l] >> 1) | (mm[l + 1] << 31);
    pos[l] = a & ~b;
    neg[l] = b & ~a;
  }
}

// width-5 non-adjacent form of a magnitude m < 2^159 (5 limbs): m = sum d_j 2^j with every non-zero d_j odd, |d_j| <= 15, and at least four
// zeros after each -- one addition per six bits on average against the NAF's three, over a table of the eight odd multiples.  digits[j],
// j < GLV_WNAF_LEN (the form of an n-bit number has at most n + 1 digits); returns the index of the top non-zero digit, -1 for m == 0.
constexpr int GLV_WNAF_LEN = 164;
ZK_HD int glv_wnaf5(const uint32_t m_in[5], int8_t digits[GLV_WNAF_LEN]) {
  uint32_t m[6] = {m_in[0], m_in[1], m_in[2], m_in[3], m_in[4], 0u};
  int top = -1;
  for (int j = 0; j < GLV_WNAF_LEN; ++j) {
    int d = 0;
    if (m[0] & 1u) {
      d = (int)(m[0] & 31u);
      if (d >= 16) d -= 32;
      // m -= d: clears the low five bits (d > 0) or carries into bit 5 (d < 0)
      if (d > 0) {
        m[0] -= (uint32_t)d;
      } else {
        uint64_t c = (uint64_t)m[0] + (uint32_t)(-d);
        m[0] = (uint32_t)c;
        c >>= 32;
        for (int l = 1; l < 6; ++l) {
          c += m[l];
          m[l] = (uint32_t)c;
          c >>= 32;
        }
      }
      top = j;
    }
    digits[j] = (int8_t)d;
    for (int l = 0; l < 5; ++l) m[l] = (m[l] >> 1) | (m[l + 1] << 31);
    m[5] >>= 1;
  }
  return top;
}

// fixed signed 4-bit windows of a magnitude given as NIN 32-bit words: m = sum_j d_j 16^j with d_j in [-8, 8] -- mag holds |d_j| one nibble
// each (NW >= NIN words: the carry out of the last input nibble lands in nibble 8 NIN when NW > NIN; with NW == NIN the caller knows the
// top nibble leaves room, e.g. the < 2^128 halves of a split in five words), sgn bit j = (d_j < 0).
template <int NW, int NIN>
ZK_HD void signed_nibbles(const uint32_t* m, uint32_t mag[NW], uint32_t sgn[(NW + 3) / 4]) {
#pragma unroll
  for (int w = 0; w < (NW + 3) / 4; ++w) sgn[w] = 0;
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t word = w < NIN ? m[w] : 0u;
    uint32_t o = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t d = ((word >> (4 * q)) & 15u) + carry;
      carry = d > 8u ? 1u : 0u;
      if (carry) {
        d = 16u - d;
        sgn[w >> 2] |= 1u << (8 * (w & 3) + q);
      }
      o |= d << (4 * q);
    }
    mag[w] = o;
  }
}

}  // namespace zk
