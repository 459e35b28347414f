// BN254 prime-field arithmetic on 8 x 32-bit Montgomery limbs (R = 2^256) for gfx950 and for the
// host side of the library (same source, compiled twice).
//
// Representation contract with the reference: an element is the 32 bytes of
// `PrimeField::into_raw_repr()` -- 4 little-endian u64 limbs of x*2^256 mod p, fully reduced
// (pairing/src/bn256/fq.rs:39-50 pins R = 2^256; pairing/src/bn256/ec.rs:653-664 is the raw
// encoder).  A little-endian u64[4] and a little-endian u32[8] are the same bytes, so no
// conversion happens at the boundary.  The arithmetic itself replaces ff_ce 0.7.1's derive
// expansion (not vendored in the reference): every result is fully reduced into [0, p), hence
// bit-identical to the reference's limbs.
//
// 32-bit limbs because the CDNA4 VALU multiplier is v_mad_u64_u32 (32x32+64 -> 64).  Measured on
// MI355X (tools/ubench_valu.hip): v_mad_u64_u32 ~ 2x the issue cost of v_add_u32 and about equal to
// a carry-propagating v_addc_co_u32, so the multiplier below keeps carries inside the 64-bit addend
// of the mad wherever it can.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

struct FqParams {
  // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  (fq.rs:5)
  static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};   // 2^256 mod q == G1_GENERATOR_X (fq.rs:39-44)
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};  // 2^512 mod q
  static constexpr uint32_t INV = 0xe4866389u;  // -q^{-1} mod 2^32
};

struct FrParams {
  // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617  (fr.rs:4)
  static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t INV = 0xefffffffu;  // -r^{-1} mod 2^32
};

template <class PR>
struct Fp {
  uint32_t l[8];

  ZK_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = 0;
    return r;
  }
  ZK_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = PR::R[i];
    return r;
  }
  ZK_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= l[i];
    return o == 0;
  }
  ZK_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
#define ZK_HOST64 1
// Host side (the window join of every multiexp, the g*_add / to_affine helpers): the same 8 x u32 limbs viewed as
// 4 x u64 with carries through unsigned __int128 -- about 2x the 32-bit loops below, which the device keeps.
namespace host64 {
typedef unsigned __int128 u128;
// add / subtract with carry (clang has builtins that map to adc / sbb; elsewhere through 128-bit arithmetic)
inline uint64_t adc(uint64_t a, uint64_t b, unsigned long long& c) {
#if defined(__clang__)
  return __builtin_addcll(a, b, c, &c);
#else
  const u128 t = (u128)a + b + c;
  c = (unsigned long long)(t >> 64);
  return (uint64_t)t;
#endif
}
inline uint64_t sbb(uint64_t a, uint64_t b, unsigned long long& c) {
#if defined(__clang__)
  return __builtin_subcll(a, b, c, &c);
#else
  const u128 t = (u128)a - b - c;
  c = (unsigned long long)((t >> 64) & 1);
  return (uint64_t)t;
#endif
}
template <class PR>
inline void load(const Fp<PR>& a, uint64_t o[4]) {
  for (int i = 0; i < 4; ++i) o[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
}
template <class PR>
inline Fp<PR> store(const uint64_t t[4]) {
  Fp<PR> r;
  for (int i = 0; i < 4; ++i) {
    r.l[2 * i] = (uint32_t)t[i];
    r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
  }
  return r;
}
template <class PR>
inline void modulus(uint64_t P[4]) {
  for (int i = 0; i < 4; ++i) P[i] = (uint64_t)PR::P[2 * i] | ((uint64_t)PR::P[2 * i + 1] << 32);
}
// t = t - p if t >= p  (t < 2p).  Branch-free: whether a sum wraps is a coin toss to the branch predictor, and the join of a multiexp
// (254 doublings, 14 of these each) was paying a misprediction on every other one (round 5: G1 jac_double 0.72 -> 0.3 us on the build box).
template <class PR>
inline void reduce_once(uint64_t t[4]) {
  uint64_t P[4], d[4];
  modulus<PR>(P);
  unsigned long long b = 0;
  d[0] = sbb(t[0], P[0], b);
  d[1] = sbb(t[1], P[1], b);
  d[2] = sbb(t[2], P[2], b);
  d[3] = sbb(t[3], P[3], b);
  const uint64_t keep = 0 - (uint64_t)b;   // all ones: t < p, keep t
  for (int i = 0; i < 4; ++i) t[i] = (t[i] & keep) | (d[i] & ~keep);
}
}  // namespace host64
#endif

// r = a - p if a >= p else a   (a < 2p)
template <class PR>
ZK_HD Fp<PR> reduce_once(const Fp<PR>& a) {
#ifdef ZK_HOST64
  uint64_t t64[4];
  host64::load(a, t64);
  host64::reduce_once<PR>(t64);
  return host64::store<PR>(t64);
#endif
  Fp<PR> t;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t d = (uint64_t)a.l[i] - PR::P[i] - borrow;
    t.l[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = borrow ? a.l[i] : t.l[i];
  return r;
}

template <class PR>
ZK_HD Fp<PR> add(const Fp<PR>& a, const Fp<PR>& b) {
#ifdef ZK_HOST64
  uint64_t A[4], B[4];
  host64::load(a, A);
  host64::load(b, B);
  unsigned long long cy = 0;
  A[0] = host64::adc(A[0], B[0], cy);
  A[1] = host64::adc(A[1], B[1], cy);
  A[2] = host64::adc(A[2], B[2], cy);
  A[3] = host64::adc(A[3], B[3], cy);   // (p < 2^254: no carry out)
  host64::reduce_once<PR>(A);
  return host64::store<PR>(A);
#endif
  Fp<PR> t;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.l[i] + b.l[i];
    t.l[i] = (uint32_t)c;
    c >>= 32;
  }
  return reduce_once(t);  // p < 2^254: a + b < 2^255, no carry out of limb 7
}

template <class PR>
ZK_HD Fp<PR> dbl(const Fp<PR>& a) {
#ifdef ZK_HOST64
  uint64_t A[4];
  host64::load(a, A);
  A[3] = (A[3] << 1) | (A[2] >> 63);
  A[2] = (A[2] << 1) | (A[1] >> 63);
  A[1] = (A[1] << 1) | (A[0] >> 63);
  A[0] <<= 1;
  host64::reduce_once<PR>(A);
  return host64::store<PR>(A);
#endif
  Fp<PR> t;
#pragma unroll
  for (int i = 7; i > 0; --i) t.l[i] = (a.l[i] << 1) | (a.l[i - 1] >> 31);
  t.l[0] = a.l[0] << 1;
  return reduce_once(t);
}

template <class PR>
ZK_HD Fp<PR> sub(const Fp<PR>& a, const Fp<PR>& b) {
#ifdef ZK_HOST64
  uint64_t A[4], B[4], P[4];
  host64::load(a, A);
  host64::load(b, B);
  host64::modulus<PR>(P);
  unsigned long long bw = 0, cy = 0;
  A[0] = host64::sbb(A[0], B[0], bw);
  A[1] = host64::sbb(A[1], B[1], bw);
  A[2] = host64::sbb(A[2], B[2], bw);
  A[3] = host64::sbb(A[3], B[3], bw);
  const uint64_t m = 0 - (uint64_t)bw;             // borrowed: add p back (branch-free, see reduce_once)
  A[0] = host64::adc(A[0], P[0] & m, cy);
  A[1] = host64::adc(A[1], P[1] & m, cy);
  A[2] = host64::adc(A[2], P[2] & m, cy);
  A[3] = host64::adc(A[3], P[3] & m, cy);
  return host64::store<PR>(A);
#endif
  Fp<PR> t;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
    t.l[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  uint32_t mask = borrow ? 0xffffffffu : 0u;
  Fp<PR> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)t.l[i] + (PR::P[i] & mask);
    r.l[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class PR>
ZK_HD Fp<PR> neg(const Fp<PR>& a) {
  if (a.is_zero()) return a;
  Fp<PR> r;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t d = (uint64_t)PR::P[i] - a.l[i] - borrow;
    r.l[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  return r;
}

// Montgomery product a*b*2^-256 mod p, fully reduced.
//   device (gfx950): finely-integrated product scanning in inline asm (mont_mul_gfx950.inc, generated by
//     tools/gen_mont_mul.py): per 32x32 partial product one v_mad_u64_u32 (64-bit accumulate) + one
//     v_addc_co_u32 into the third accumulator word, no moves: 136 mad + 136 addc + 8 v_mul_lo_u32.
//   host: CIOS on 4 x 64-bit limbs with unsigned __int128 (portable 32-bit-limb CIOS kept as the fallback).
template <class PR>
ZK_HD Fp<PR> mul(const Fp<PR>& a, const Fp<PR>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_PORTABLE_MUL)
#include "mont_mul_gfx950.inc"
  return reduce_once(r);
#elif !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
  // host: CIOS on 4 x 64-bit limbs (the same bytes viewed as u64) -- the window join of a multiexp runs a few
  // hundred doublings on the host.  p < 2^254 leaves two spare bits, so the running value stays below 2^257 and the
  // fifth word is a single carry word.
  typedef unsigned __int128 u128;
  uint64_t A[4], B[4], P[4], t[5] = {0, 0, 0, 0, 0};
  host64::load(a, A);
  host64::load(b, B);
  host64::modulus<PR>(P);
  // -p^-1 mod 2^64 from -p^-1 mod 2^32 by one Newton step: x' = x * (2 + p0 * x)  (for x = -p0^-1)
  const uint64_t x32 = PR::INV;
  const uint64_t inv64 = x32 * (2 + P[0] * x32);
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)A[j] * B[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    uint64_t t4 = t[4] + (uint64_t)c;  // < 2^64: the value is < 2p * 2^64 ... see above
    uint64_t k = t[0] * inv64;
    c = (u128)k * P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)k * P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t4;
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  // a, b < p  =>  t < 2p < 2^255: t[4] == 0
  host64::reduce_once<PR>(t);
  return host64::store<PR>(t);
#else
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a.l[j] * b.l[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    uint32_t k = t[0] * PR::INV;
    c = (uint64_t)k * PR::P[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c += (uint64_t)k * PR::P[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
  // a, b < p < 2^254  =>  t < 2p < 2^255, so t[8] == 0 here
  Fp<PR> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i];
  return reduce_once(r);
#endif
}

template <class PR>
ZK_HD Fp<PR> sqr(const Fp<PR>& a) {
  return mul(a, a);
}

// a^e, e given as `n` u32 limbs (little endian); square-and-multiply MSB first.
template <class PR>
ZK_HD Fp<PR> pow_limbs(const Fp<PR>& a, const uint32_t* e, int n) {
  Fp<PR> res = Fp<PR>::one();
  bool found = false;
  for (int i = n * 32 - 1; i >= 0; --i) {
    bool bit = (e[i >> 5] >> (i & 31)) & 1;
    if (found) res = sqr(res);
    else found = bit;
    if (bit) res = mul(res, a);
  }
  return res;
}

template <class PR>
ZK_HD Fp<PR> pow_u64(const Fp<PR>& a, uint64_t e) {
  uint32_t l[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
  return pow_limbs(a, l, 2);
}

// Fermat inverse a^(p-2); inverse of zero returns zero (callers test is_zero first).
template <class PR>
ZK_HD Fp<PR> inv(const Fp<PR>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = PR::P[i];
  e[0] -= 2;  // p is odd and P[0] >= 2 for both moduli
  return pow_limbs(a, e, 8);
}

// canonical integer (< p) -> Montgomery form
template <class PR>
ZK_HD Fp<PR> from_canonical(const Fp<PR>& c) {
  Fp<PR> r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.l[i] = PR::R2[i];
  return mul(c, r2);
}

// Montgomery form -> canonical integer
template <class PR>
ZK_HD Fp<PR> to_canonical(const Fp<PR>& a) {
  Fp<PR> o = Fp<PR>::zero();
  o.l[0] = 1;
  return mul(a, o);
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1); element layout c0 || c1 (64 bytes), as pairing/src/bn256/fq2.rs:9-12.
struct Fq2 {
  Fq c0, c1;
  ZK_HD static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  ZK_HD static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  ZK_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
};

ZK_HD Fq2 add(const Fq2& a, const Fq2& b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_HD Fq2 sub(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZK_HD Fq2 dbl(const Fq2& a) { return Fq2{dbl(a.c0), dbl(a.c1)}; }
ZK_HD Fq2 neg(const Fq2& a) { return Fq2{neg(a.c0), neg(a.c1)}; }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
ZK_HD Fq2 mul(const Fq2& a, const Fq2& b) {
  Fq aa = mul(a.c0, b.c0);
  Fq bb = mul(a.c1, b.c1);
  Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return Fq2{sub(aa, bb), sub(sub(s, aa), bb)};
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u
ZK_HD Fq2 sqr(const Fq2& a) {
  Fq ab = mul(a.c0, a.c1);
  Fq c0 = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
  return Fq2{c0, dbl(ab)};
}
ZK_HD Fq2 inv(const Fq2& a) {
  Fq t = inv(add(sqr(a.c0), sqr(a.c1)));
  return Fq2{mul(a.c0, t), neg(mul(a.c1, t))};
}

}  // namespace zk
