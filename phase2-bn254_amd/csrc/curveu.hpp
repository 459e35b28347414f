// XYZZ bucket accumulator on U-form (29-bit lazy) field elements: the inner loop of the G1 multiexp.
//
// Same group law as curve.hpp's xyzz_add_mixed (madd-2008-s, with the P+P / P+(-P) / infinity cases the
// reference handles in ec.rs:456-536), restated on fieldu.hpp arithmetic.  Two things are tracked by hand:
//
// (1) DOMAINS.  u_mul divides by 2^261 while the memory format carries 2^256, so instead of converting the
//     loaded affine point (one extra product per coordinate) the accumulator keeps its coordinates in the
//     powers of two that make every formula line close:
//         x2, y2 (loaded)   : value * 2^256        X, Y   : value * 2^261        ZZ, ZZZ : value * 2^266
//     e.g. U2 = u_mul(x2, ZZ) = x2*zz * 2^(256+266-261) = (x2*zz) * 2^261, the domain of X.  Only the first
//     point of a bucket (and the rare doubling) pays a product by the constant 2^266 mod p.
//
// (2) BOUNDS.  Every value is annotated "< k p" and every limb vector "N" (limbs 0..7 < 2^29) or "< s*2^29";
//     they are exactly the preconditions of u_mul / u_sub in fieldu.hpp.  c = p/2^261 < 0.0060 is the factor
//     by which a product of two values (in units of p) shrinks:  u_mul(a, b) < (a/p)(b/p) * 0.006 p + p.
//     Accumulator invariant:  X < 6p,  Y < 2p,  ZZ < 2p,  ZZZ < 2p,  all N-form.
//     Infinity is ZZ == literal zero limbs.
#pragma once

#include "curve.hpp"
#include "fieldu.hpp"

namespace zk {

template <class PR>
struct XYZZU {
  FpU<PR> x, y, zz, zzz;
  ZK_HD static XYZZU zero() { return XYZZU{FpU<PR>::zero(), FpU<PR>::zero(), FpU<PR>::zero(), FpU<PR>::zero()}; }
  ZK_HD bool is_zero() const { return zz.limbs_all_zero(); }
};

// value == 0 mod p for an N-form value < 8p (rare path only)
template <class PR>
ZK_HD bool u_is_zero_lt8p(const FpU<PR>& a) {
  bool z = a.limbs_all_zero();
  for_limbs<7>([&](auto kc) {
    constexpr int k = decltype(kc)::value + 1;
    uint32_t e = 0;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr uint32_t c = USubConst<PR, k, 0>::limb(i);  // k*p on normalised limbs
      e |= a.l[i] ^ c;
    });
    z = z || (e == 0);
  });
  return z;
}

// acc = 2 * (x2, y2), x2, y2 loaded affine coordinates (canonical, 2^256 domain)   [mdbl-2008-s-1]
template <class PR>
ZK_HD XYZZU<PR> xyzzu_double_affine(const FpU<PR>& x2, const FpU<PR>& y2) {
  const FpU<PR> C = UPow2<PR, 266>::get();
  FpU<PR> x = u_mul(x2, C);                   // x * 2^261, < 2p, N
  FpU<PR> y = u_mul(y2, C);                   // y * 2^261, < 2p, N
  FpU<PR> u = u_dbl(y);                       // < 4p, limbs < 2^30
  FpU<PR> v = u_mul(u, u);                    // < 1.1p   (u is not N-form: plain product)
  FpU<PR> w = u_mul(u, v);                    // < 1.1p
  FpU<PR> s = u_mul(x, v);                    // < 1.1p
  FpU<PR> xx = u_sqr(x);                      // < 1.1p
  FpU<PR> m = u_carry(u_add(u_dbl(xx), xx));  // 3*xx < 3.3p, N after the carry
  FpU<PR> mm = u_sqr(m);                      // < 1.1p
  XYZZU<PR> r;
  r.x = u_sub<4, 2>(mm, u_dbl(s));            // 2s < 2.2p <= 4p, limbs < 2^30;  X < 5.1p
  FpU<PR> d = u_sub<8, 1>(s, r.x);            // < 9.1p
  FpU<PR> ny = u_sub<2, 1>(FpU<PR>::zero(), y);  // 2p - y, N
  r.y = u_mul2(m, d, w, ny);                  // M*D - W*y: (3.3*9.1 + 1.1*2) c + 1 < 1.2p  (invariant Y < 2p)
  r.zz = u_mul(v, C);                         // v * 2^266, < 2p
  r.zzz = u_mul(w, C);
  return r;
}

// acc = 2 * acc   [dbl-2008-s-1]; infinity stays infinity.  Same domains / invariants as the mixed add.
template <class PR>
ZK_HD XYZZU<PR> xyzzu_double(const XYZZU<PR>& a) {
  if (a.is_zero()) return a;
  FpU<PR> u = u_dbl(a.y);                       // < 4p, limbs < 2^30
  FpU<PR> v = u_mul(u, u);                      // < 1.1p
  FpU<PR> w = u_mul(u, v);                      // < 1.03p
  FpU<PR> s = u_mul(a.x, v);                    // X < 6p: < 1.04p
  FpU<PR> xx = u_sqr(a.x);                      // < 1.22p
  FpU<PR> m = u_carry(u_add(u_dbl(xx), xx));    // 3*xx < 3.7p, N
  FpU<PR> mm = u_sqr(m);                        // < 1.09p
  XYZZU<PR> r;
  r.x = u_sub<4, 2>(mm, u_dbl(s));              // < 5.1p
  FpU<PR> d = u_sub<8, 1>(s, r.x);              // < 9.1p
  FpU<PR> ny = u_sub<2, 1>(FpU<PR>::zero(), a.y);
  r.y = u_mul2(m, d, w, ny);                    // M*D - W*Y1 < 1.22p
  r.zz = u_mul(v, a.zz);                        // 261 + 266 - 261 = 266 domain
  r.zzz = u_mul(w, a.zzz);
  return r;
}

// acc += (+/-)(x2, y2);  (x2, y2) != infinity, canonical memory-format coordinates.
template <class PR>
ZK_HD void xyzzu_add_mixed(XYZZU<PR>& acc, const Fp<PR>& x2s, const Fp<PR>& y2s, bool negate) {
  const FpU<PR> x2 = u_from_std(x2s);                      // < p, N
  FpU<PR> y2 = u_from_std(y2s);                            // < p, N
  {
    FpU<PR> ny = u_sub<1, 1>(FpU<PR>::zero(), y2);          // p - y2 in (0, p], N
#pragma unroll
    for (int i = 0; i < 9; ++i) y2.l[i] = negate ? ny.l[i] : y2.l[i];
  }
  if (acc.is_zero()) {
    const FpU<PR> C = UPow2<PR, 266>::get();
    acc.x = u_mul(x2, C);                                   // x * 2^261 < 2p
    acc.y = u_mul(y2, C);
    acc.zz = C;                                             // 1 * 2^266
    acc.zzz = C;
    return;
  }
  FpU<PR> u2 = u_mul(x2, acc.zz);                           // < 2p
  FpU<PR> s2 = u_mul(y2, acc.zzz);                          // < 2p
  FpU<PR> p = u_sub<8, 1>(u2, acc.x);                       // X < 6p <= 8p;   P < 10p, N
  FpU<PR> r = u_sub<2, 1>(s2, acc.y);                       // Y < 2p;         R < 4p, N
  FpU<PR> pp = u_sqr(p);                                    // 100c + 1 < 1.6p
  FpU<PR> ppp = u_mul(p, pp);                               // < 1.1p
  FpU<PR> q = u_mul(acc.x, pp);                             // < 1.06p
  FpU<PR> rr = u_sqr(r);                                    // 16c + 1 < 1.1p
  FpU<PR> t = u_add(ppp, u_dbl(q));                         // < 3.3p <= 4p, limbs < 3 * 2^29
  FpU<PR> x3 = u_sub<4, 3>(rr, t);                          // < 5.1p  (invariant X < 6p)
  FpU<PR> d = u_sub<8, 1>(q, x3);                           // < 9.1p
  FpU<PR> ny1 = u_sub<2, 1>(FpU<PR>::zero(), acc.y);        // 2p - Y1 in (0, 2p], N
  FpU<PR> y3 = u_mul2(r, d, ny1, ppp);                      // R*D - Y1*PPP: (4*9.1 + 2*1.1) c + 1 < 1.24p  (invariant Y < 2p)
  FpU<PR> zz3 = u_mul(acc.zz, pp);                          // < 2p
  FpU<PR> zzz3 = u_mul(acc.zzz, ppp);                       // < 2p
  if (u_is_zero_lt2p(zz3)) {
    // P == 0 (ZZ1 != 0): the points have the same x.  Same point -> double (ec.rs:483-485); opposite -> infinity (ec.rs:487).
    if (u_is_zero_lt8p(r)) acc = xyzzu_double_affine(x2, y2);
    else acc = XYZZU<PR>::zero();
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.zz = zz3;
  acc.zzz = zzz3;
}

// accumulator -> memory-format XYZZ (canonical coordinates, 2^256 domain); infinity -> all zero
template <class PR>
ZK_HD XYZZ<Fp<PR>> xyzzu_to_std(const XYZZU<PR>& a) {
  if (a.is_zero()) return XYZZ<Fp<PR>>::zero();
  const FpU<PR> c256 = UPow2<PR, 256>::get();  // X*2^261 * 2^256 / 2^261 = X * 2^256
  const FpU<PR> c251 = UPow2<PR, 251>::get();  // ZZ*2^266 * 2^251 / 2^261 = ZZ * 2^256
  XYZZ<Fp<PR>> r;
  r.x = u_to_std_lt2p(u_mul(a.x, c256));
  r.y = u_to_std_lt2p(u_mul(a.y, c256));
  r.zz = u_to_std_lt2p(u_mul(a.zz, c251));
  r.zzz = u_to_std_lt2p(u_mul(a.zzz, c251));
  return r;
}

// =================================================================================================
// XYZZ RECORDS IN THE R DOMAIN: what the bucket accumulation hands to the bucket reduction (msm_impl.hpp: the running sums, the
// LDS trees, the heavy-bucket combine).  A record is the 128-byte XYZZ<Fp> of curve.hpp whose four coordinates are CANONICAL
// (< p) values in the 2^261 domain -- Montgomery form for u_mul -- instead of the memory format's 2^256.  A full addition then
// closes on U-form arithmetic without a single domain-fixing product (with the accumulator's mixed domains it would take four),
// records load and store by re-packing bits only, and the accumulator saves two of its four closing products.  Because all four
// coordinates carry the same extra factor 2^5, x = X/ZZ and y = Y/ZZZ are unchanged: xyzz_to_affine works on a record as it is;
// anything else (the host join: xyzz_to_jacobian) converts with xyzzr_to_std first.  The all-zero record is infinity.
//   Register form (XYZZU, all coordinates 2^261 domain): X < 6p, Y < 2p, ZZ < 2p, ZZZ < 2p, N-form; infinity: ZZ literal zero.

// bucket accumulator (X, Y: 2^261; ZZ, ZZZ: 2^266) -> record
template <class PR>
ZK_HD XYZZ<Fp<PR>> xyzzu_to_r(const XYZZU<PR>& a) {
  if (a.is_zero()) return XYZZ<Fp<PR>>::zero();
  const FpU<PR> c256 = UPow2<PR, 256>::get();  // ZZ*2^266 * 2^256 / 2^261 = ZZ * 2^261
  XYZZ<Fp<PR>> r;
  r.x = u_to_std_lt32p(a.x);                   // < 6p
  r.y = u_to_std_lt2p(a.y);
  r.zz = u_to_std_lt2p(u_mul(a.zz, c256));
  r.zzz = u_to_std_lt2p(u_mul(a.zzz, c256));
  return r;
}

// record -> bucket accumulator (X, Y: 2^261; ZZ, ZZZ: 2^266): a bucket sum carried from one chunk of a streamed multiexp to the
// next (msm_impl.hpp: msm_accumulate_kernel<.., CARRY>) is taken up again at the price of two products
template <class PR>
ZK_HD XYZZU<PR> xyzzu_from_r(const XYZZ<Fp<PR>>& s) {
  if (s.is_zero()) return XYZZU<PR>::zero();
  const FpU<PR> c266 = UPow2<PR, 266>::get();  // v*2^261 * 2^266 / 2^261 = v * 2^266
  return XYZZU<PR>{u_from_std(s.x), u_from_std(s.y), u_mul(u_from_std(s.zz), c266), u_mul(u_from_std(s.zzz), c266)};  // < p, < p, < 2p, < 2p
}

template <class PR>
ZK_HD XYZZU<PR> xyzzr_load(const XYZZ<Fp<PR>>& s) {  // canonical coordinates: < p, N-form; the zero record gives ZZ == 0 limbs
  return XYZZU<PR>{u_from_std(s.x), u_from_std(s.y), u_from_std(s.zz), u_from_std(s.zzz)};
}

template <class PR>
ZK_HD XYZZ<Fp<PR>> xyzzr_store(const XYZZU<PR>& a) {
  if (a.is_zero()) return XYZZ<Fp<PR>>::zero();
  XYZZ<Fp<PR>> r;
  r.x = u_to_std_lt32p(a.x);                   // < 6p
  r.y = u_to_std_lt2p(a.y);
  r.zz = u_to_std_lt2p(a.zz);
  r.zzz = u_to_std_lt2p(a.zzz);
  return r;
}

// record -> memory-format XYZZ (2^256 domain)
template <class PR>
ZK_HD XYZZ<Fp<PR>> xyzzr_to_std(const XYZZ<Fp<PR>>& s) {
  if (s.is_zero()) return s;
  const FpU<PR> c256 = UPow2<PR, 256>::get();  // v*2^261 * 2^256 / 2^261 = v * 2^256
  XYZZ<Fp<PR>> r;
  r.x = u_to_std_lt2p(u_mul(u_from_std(s.x), c256));
  r.y = u_to_std_lt2p(u_mul(u_from_std(s.y), c256));
  r.zz = u_to_std_lt2p(u_mul(u_from_std(s.zz), c256));
  r.zzz = u_to_std_lt2p(u_mul(u_from_std(s.zzz), c256));
  return r;
}

// acc += o, both in register form  [add-2008-s, with the P + P / P + (-P) / infinity cases of ec.rs:360-454]
template <class PR>
ZK_HD void xyzzr_add(XYZZU<PR>& acc, const XYZZU<PR>& o) {
  if (o.is_zero()) return;
  if (acc.is_zero()) {
    acc = o;
    return;
  }
  FpU<PR> u1 = u_mul(acc.x, o.zz);                          // 6*2 c + 1 < 1.08p
  FpU<PR> u2 = u_mul(o.x, acc.zz);                          // < 1.08p
  FpU<PR> s1 = u_mul(acc.y, o.zzz);                         // < 1.03p
  FpU<PR> s2 = u_mul(o.y, acc.zzz);                         // < 1.03p
  FpU<PR> p = u_sub<2, 1>(u2, u1);                          // U1 < 2p;  P < 4p, N
  FpU<PR> r = u_sub<2, 1>(s2, s1);                          // S1 < 2p;  R < 4p, N
  FpU<PR> pp = u_sqr(p);                                    // 16c + 1 < 1.1p
  FpU<PR> ppp = u_mul(p, pp);                               // < 1.03p
  FpU<PR> q = u_mul(u1, pp);                                // < 1.01p
  FpU<PR> rr = u_sqr(r);                                    // < 1.1p
  FpU<PR> t = u_add(ppp, u_dbl(q));                         // < 3.1p <= 4p, limbs < 3 * 2^29
  FpU<PR> x3 = u_sub<4, 3>(rr, t);                          // < 5.1p  (invariant X < 6p)
  FpU<PR> d = u_sub<8, 1>(q, x3);                           // < 9.1p
  FpU<PR> ns1 = u_sub<2, 1>(FpU<PR>::zero(), s1);           // 2p - S1 in (0, 2p], N
  FpU<PR> y3 = u_mul2(r, d, ns1, ppp);                      // R*D - S1*PPP: (4*9.1 + 2*1.03) c + 1 < 1.24p  (invariant Y < 2p)
  FpU<PR> zz3 = u_mul(u_mul(acc.zz, o.zz), pp);             // < 2p
  FpU<PR> zzz3 = u_mul(u_mul(acc.zzz, o.zzz), ppp);         // < 2p
  if (u_is_zero_lt2p(zz3)) {
    // P == 0 (ZZ1, ZZ2 != 0): the points have the same x.  Same point -> double; opposite -> infinity.
    if (u_is_zero_lt8p(r)) acc = xyzzu_double(acc);         // (domain-agnostic in ZZ / ZZZ: they only pass through products)
    else acc = XYZZU<PR>::zero();
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.zz = zz3;
  acc.zzz = zzz3;
}

#if defined(__HIPCC__)
// ---- the same addition by FOUR cooperating lanes (a "quad": lanes 4q .. 4q+3 of a wave), for the parallelism-starved tails of the
// bucket reduction (msm_impl.hpp: the later running-sum levels, the last rounds of the tree kernels).  There a few thousand lanes
// each walk a chain of dependent additions, and a lane's addition is 14 field products one after the other (~3300 VALU instructions,
// ~7 us) however idle the device is.  The products of an addition are mostly independent:
//     round 1:  U1 = X1 ZZ2   U2 = X2 ZZ1   S1 = Y1 ZZZ2   S2 = Y2 ZZZ1
//     round 2:  ZZ1 ZZ2       ZZZ1 ZZZ2     PP = P^2       RR = R^2             (P = U2 - U1, R = S2 - S1)
//     round 3:  PPP = P PP    Q = U1 PP     ZZ3 = ZZ1 ZZ2 PP
//     round 4:  Y3 = R (Q - X3) - S1 PPP  (two products, one reduction)         ZZZ3 = ZZZ1 ZZZ2 PPP
// Every lane of the quad holds BOTH operands (replicated), picks its own pair of factors by its role (v_cndmask), runs the one
// product routine of the round, and the four results travel to all four lanes with DPP quad permutes (v_mov_b32 quad_perm, full
// rate, no LDS): three u_mul + one u_mul2 in sequence instead of 14 products, ~1400 instructions per lane.  The result is again
// replicated.  Same values, same bounds as xyzzr_add (u_sqr(a) and u_mul(a, a) form the same column sums), so the canonical records
// written downstream are bit-identical.  The special cases (an operand at infinity, equal or opposite points) are uniform over the
// quad -- its lanes hold the same data -- and take the one-lane code, executed redundantly.
// ALL FOUR lanes of a quad must be active when this is called (DPP reads its source lanes).
template <int K>
__device__ __forceinline__ uint32_t quad_get(uint32_t v) {   // v of lane K of the caller's quad
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, true);
#else
  return v;   // (host pass of hipcc: never executed)
#endif
}
template <int K, class PR>
__device__ __forceinline__ FpU<PR> quad_get(const FpU<PR>& a) {
  FpU<PR> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = quad_get<K>(a.l[i]);
  return r;
}
template <class PR>
__device__ __forceinline__ FpU<PR> quad_pick(uint32_t role, const FpU<PR>& a0, const FpU<PR>& a1, const FpU<PR>& a2, const FpU<PR>& a3) {
  // two levels of two-way selects on VALUES (a four-way `?:` chain over the limb arrays became a table in scratch memory indexed by
  // the role: hipcc selects the pointer, not the value)
  FpU<PR> r;
  const bool odd = (role & 1u) != 0, high = (role & 2u) != 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t v0 = a0.l[i], v1 = a1.l[i], v2 = a2.l[i], v3 = a3.l[i];
    const uint32_t lo = odd ? v1 : v0, hi = odd ? v3 : v2;
    r.l[i] = high ? hi : lo;
  }
  return r;
}
template <int K, class PR>
__device__ __forceinline__ XYZZU<PR> quad_fetch(const XYZZU<PR>& a) {   // the accumulator lane K of the quad holds, in all four lanes
  return XYZZU<PR>{quad_get<K>(a.x), quad_get<K>(a.y), quad_get<K>(a.zz), quad_get<K>(a.zzz)};
}
template <class PR>
__device__ __forceinline__ XYZZU<PR> xyzzr_add_quad(const XYZZU<PR> acc, const XYZZU<PR> o, uint32_t role) {   // by value in, by value out, ONE exit (hipcc keeps by-reference / multiply-returned sums in scratch)
  XYZZU<PR> res = acc;
  if (!o.is_zero()) {
    res = o;
    if (!acc.is_zero()) {
      FpU<PR> t = u_mul(quad_pick(role, acc.x, o.x, acc.y, o.y), quad_pick(role, o.zz, acc.zz, o.zzz, acc.zzz));
      const FpU<PR> u1 = quad_get<0>(t), u2 = quad_get<1>(t), s1 = quad_get<2>(t), s2 = quad_get<3>(t);   // < 1.08p, < 1.08p, < 1.03p, < 1.03p
      const FpU<PR> p = u_sub<2, 1>(u2, u1);                     // U1 < 2p;  P < 4p, N
      const FpU<PR> r = u_sub<2, 1>(s2, s1);                     // S1 < 2p;  R < 4p, N
      t = u_mul(quad_pick(role, acc.zz, acc.zzz, p, r), quad_pick(role, o.zz, o.zzz, p, r));
      const FpU<PR> zz12 = quad_get<0>(t), zzz12 = quad_get<1>(t), pp = quad_get<2>(t), rr = quad_get<3>(t);   // < 1.03p, < 1.03p, < 1.1p, < 1.1p
      t = u_mul(quad_pick(role, p, u1, zz12, zz12), pp);
      const FpU<PR> ppp = quad_get<0>(t), q = quad_get<1>(t), zz3 = quad_get<2>(t);   // < 1.03p, < 1.01p, < 2p
      const FpU<PR> x3 = u_sub<4, 3>(rr, u_add(ppp, u_dbl(q)));  // < 5.1p  (invariant X < 6p)
      const FpU<PR> d = u_sub<8, 1>(q, x3);                      // < 9.1p
      const FpU<PR> ns1 = u_sub<2, 1>(FpU<PR>::zero(), s1);      // 2p - S1 in (0, 2p], N
      const bool lead = role == 0;
      FpU<PR> fa, fb, fc;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        fa.l[i] = lead ? r.l[i] : zzz12.l[i];
        fb.l[i] = lead ? d.l[i] : ppp.l[i];
        fc.l[i] = lead ? ns1.l[i] : 0u;
      }
      t = u_mul2(fa, fb, fc, ppp);                               // lane 0: R*D - S1*PPP < 1.24p;  lanes 1..3: ZZZ1 ZZZ2 PPP < 2p
      res = XYZZU<PR>{x3, quad_get<0>(t), zz3, quad_get<1>(t)};
      if (u_is_zero_lt2p(zz3)) {
        // P == 0 (ZZ1, ZZ2 != 0): the points have the same x.  Same point -> double; opposite -> infinity.  (uniform over the quad)
        res = XYZZU<PR>::zero();
        if (u_is_zero_lt8p(r)) res = xyzzu_double(acc);
      }
    }
  }
  return res;
}
#endif  // __HIPCC__

// =================================================================================================
// Jacobian accumulator on U-form elements, for SCALAR MULTIPLICATION (batch_exp): a doubling is cheaper in
// Jacobian coordinates (dbl-2009-l: 4 squarings + 3 products here, 1071 mads) than in XYZZ (1467), and a scalar
// multiplication is doublings first of all.  Every coordinate lives in the same 2^261 domain (the base point is
// converted once per scalar multiplication), so formulas are the textbook ones.
//   Invariants (N-form):  X < 6p,  Y < 2p,  Z < 2p;   infinity is Z == literal zero limbs.
//   c = p / 2^261 < 0.006:  u_mul(a, b) < (a/p)(b/p) c p + p.
template <class PR>
struct JacU {
  FpU<PR> x, y, z;
  ZK_HD static JacU zero() { return JacU{FpU<PR>::zero(), FpU<PR>::zero(), FpU<PR>::zero()}; }
  ZK_HD bool is_zero() const { return z.limbs_all_zero(); }
};

// 2 * a   [dbl-2009-l with D = 4 X B taken as a product: the textbook (X+B)^2 - A - C costs a squaring instead, but its
// lazy subtractions would let X grow past the invariant]
template <class PR>
ZK_HD JacU<PR> jacu_double(const JacU<PR>& a) {
  if (a.is_zero()) return a;
  FpU<PR> A = u_sqr(a.x);                                   // 36c + 1 < 1.22p
  FpU<PR> B = u_sqr(a.y);                                   // 4c + 1 < 1.03p
  FpU<PR> C = u_sqr(B);                                     // < 1.01p
  FpU<PR> D = u_mul(u_dbl(u_dbl(a.x)), B);                  // 4 X B: limbs of 4X < 2^31;  4*6*1.03 c + 1 < 1.15p
  FpU<PR> E = u_carry(u_add(u_dbl(A), A));                  // 3A < 3.66p, N
  FpU<PR> F = u_sqr(E);                                     // < 1.08p
  JacU<PR> r;
  r.x = u_sub<4, 2>(F, u_dbl(D));                           // 2D < 2.3p <= 4p, limbs < 2^30;  X3 < 5.08p
  FpU<PR> dmx = u_sub<8, 1>(D, r.x);                        // < 9.15p, N
  FpU<PR> negc = u_sub<2, 1>(FpU<PR>::zero(), C);           // 2p - C in (0, 2p], N
  r.y = u_mul2(E, dmx, negc, UPow2<PR, 264>::get());        // E (D - X3) - 8 C: (3.66*9.15 + 2) c + 1 < 1.22p
  r.z = u_mul(u_dbl(a.y), a.z);                             // 2 Y Z: 8c + 1 < 1.05p
  return r;
}

// acc += (+/-)(x2, y2);  x2, y2: the base point in the 2^261 domain, N-form, < 2p, not infinity   [madd-2007-bl]
template <class PR>
ZK_HD void jacu_add_mixed(JacU<PR>& acc, const FpU<PR>& x2, const FpU<PR>& y2in, bool negate) {
  FpU<PR> y2 = y2in;
  {
    FpU<PR> ny = u_sub<2, 1>(FpU<PR>::zero(), y2in);        // 2p - y2 in (0, 2p], N
#pragma unroll
    for (int i = 0; i < 9; ++i) y2.l[i] = negate ? ny.l[i] : y2in.l[i];
  }
  if (acc.is_zero()) {
    acc.x = x2;
    acc.y = y2;
    acc.z = UPow2<PR, 261>::get();                          // one
    return;
  }
  FpU<PR> z1z1 = u_sqr(acc.z);                              // 4c + 1 < 1.03p
  FpU<PR> u2 = u_mul(x2, z1z1);                             // < 1.02p
  FpU<PR> s2 = u_mul(y2, u_mul(acc.z, z1z1));               // < 1.02p
  FpU<PR> h = u_sub<8, 1>(u2, acc.x);                       // X < 6p <= 8p;  H < 9.02p, N
  FpU<PR> hh = u_sqr(h);                                    // 81.4c + 1 < 1.49p
  FpU<PR> i4 = u_dbl(u_dbl(hh));                            // I = 4 HH < 5.94p, limbs < 2^31
  FpU<PR> j = u_mul(h, i4);                                 // < 1.32p
  FpU<PR> r = u_carry(u_dbl(u_sub<2, 1>(s2, acc.y)));       // 2 (S2 - Y1 + 2p) < 6.04p, N
  FpU<PR> v = u_mul(acc.x, i4);                             // < 1.22p
  FpU<PR> rr = u_sqr(r);                                    // 36.5c + 1 < 1.22p
  FpU<PR> x3 = u_sub<4, 3>(rr, u_add(j, u_dbl(v)));         // J + 2V < 3.76p <= 4p, limbs < 3 * 2^29;  X3 < 5.22p
  FpU<PR> vmx = u_sub<8, 1>(v, x3);                         // < 9.22p, N
  FpU<PR> ny2 = u_sub<4, 2>(FpU<PR>::zero(), u_dbl(acc.y)); // 4p - 2 Y1 in (0, 4p], N
  FpU<PR> y3 = u_mul2(r, vmx, ny2, j);                      // r (V - X3) - 2 Y1 J: (6.04*9.22 + 4*1.32) c + 1 < 1.37p
  FpU<PR> z3 = u_mul(u_dbl(acc.z), h);                      // (Z1 + H)^2 - Z1Z1 - HH = 2 Z1 H: 36.1c + 1 < 1.22p
  if (u_is_zero_lt2p(z3)) {
    // H == 0 (Z1 != 0): the points have the same x.  Same point -> double (ec.rs:483-485); opposite -> infinity (ec.rs:487).
    if (u_is_zero_lt8p(r)) {
      JacU<PR> b{x2, y2, UPow2<PR, 261>::get()};
      acc = jacu_double(b);
    } else {
      acc = JacU<PR>::zero();
    }
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.z = z3;
}

// A table entry for windowed scalar multiplication: a Jacobian point together with Z^2 and Z^3, so that adding it costs
// 3 squarings + 9 products + 1 fused pair (2079 mads) instead of add-2007-bl's 4 + 10 + 1.  48 words = 192 bytes.
// Invariants as JacU, ZZ < 2p, ZZZ < 2p.
template <class PR>
struct alignas(16) JacTabU {
  FpU<PR> x, y, z, zz, zzz;
  uint32_t pad[3];
};
template <class PR>
ZK_HD JacTabU<PR> jacu_tab_entry(const JacU<PR>& q) {
  JacTabU<PR> t;
  t.x = q.x;
  t.y = q.y;
  t.z = q.z;
  t.zz = u_sqr(q.z);                                        // 4c + 1 < 1.03p
  t.zzz = u_mul(q.z, t.zz);                                 // < 1.02p
  t.pad[0] = t.pad[1] = t.pad[2] = 0;
  return t;
}

// acc += (+/-) t;  t != infinity   [add-2007-bl with the entry's Z^2, Z^3 given]
template <class PR>
ZK_HD void jacu_add_tab(JacU<PR>& acc, const JacTabU<PR>& t, bool negate) {
  FpU<PR> y2 = t.y;
  {
    FpU<PR> ny = u_sub<2, 1>(FpU<PR>::zero(), t.y);         // 2p - Y2 in (0, 2p], N
#pragma unroll
    for (int i = 0; i < 9; ++i) y2.l[i] = negate ? ny.l[i] : t.y.l[i];
  }
  if (acc.is_zero()) {
    acc.x = t.x;
    acc.y = y2;
    acc.z = t.z;
    return;
  }
  FpU<PR> z1z1 = u_sqr(acc.z);                              // < 1.03p
  FpU<PR> u1 = u_mul(acc.x, t.zz);                          // 6*2c + 1 < 1.08p
  FpU<PR> u2 = u_mul(t.x, z1z1);                            // 6*1.03c + 1 < 1.04p
  FpU<PR> s1 = u_mul(acc.y, t.zzz);                         // < 1.03p
  FpU<PR> s2 = u_mul(y2, u_mul(acc.z, z1z1));               // < 1.02p
  FpU<PR> h = u_sub<2, 1>(u2, u1);                          // < 3.04p, N
  FpU<PR> hh = u_sqr(h);                                    // 9.3c + 1 < 1.06p
  FpU<PR> i4 = u_dbl(u_dbl(hh));                            // I = 4 HH < 4.24p, limbs < 2^31
  FpU<PR> j = u_mul(h, i4);                                 // < 1.08p
  FpU<PR> r = u_carry(u_dbl(u_sub<2, 1>(s2, s1)));          // 2 (S2 - S1 + 2p) < 6.04p, N
  FpU<PR> v = u_mul(u1, i4);                                // < 1.03p
  FpU<PR> rr = u_sqr(r);                                    // < 1.22p
  FpU<PR> x3 = u_sub<4, 3>(rr, u_add(j, u_dbl(v)));         // J + 2V < 3.14p <= 4p, limbs < 3 * 2^29;  X3 < 5.22p
  FpU<PR> vmx = u_sub<8, 1>(v, x3);                         // < 9.03p, N
  FpU<PR> n2s1 = u_sub<4, 2>(FpU<PR>::zero(), u_dbl(s1));   // 4p - 2 S1 in (0, 4p], N
  FpU<PR> y3 = u_mul2(r, vmx, n2s1, j);                     // r (V - X3) - 2 S1 J: (6.04*9.03 + 4*1.08) c + 1 < 1.36p
  FpU<PR> z3 = u_mul(u_mul(u_dbl(acc.z), t.z), h);          // 2 Z1 Z2 H: (8c + 1) * 3.04 c + 1 < 1.02p
  if (u_is_zero_lt2p(z3)) {
    // H == 0: same x.  Same point -> double; opposite -> infinity (ec.rs:398-408).
    if (u_is_zero_lt8p(r)) {
      JacU<PR> b{t.x, y2, t.z};
      acc = jacu_double(b);
    } else {
      acc = JacU<PR>::zero();
    }
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.z = z3;
}

// accumulator -> memory-format Jacobian (canonical coordinates, 2^256 domain); infinity -> z == 0 (x, y zero too)
template <class PR>
ZK_HD Jacobian<Fp<PR>> jacu_to_std(const JacU<PR>& a) {
  Jacobian<Fp<PR>> r{Fp<PR>::zero(), Fp<PR>::zero(), Fp<PR>::zero()};
  if (a.is_zero()) return r;
  const FpU<PR> c256 = UPow2<PR, 256>::get();               // v*2^261 * 2^256 / 2^261 = v * 2^256
  r.x = u_to_std_lt2p(u_mul(a.x, c256));
  r.y = u_to_std_lt2p(u_mul(a.y, c256));
  r.z = u_to_std_lt2p(u_mul(a.z, c256));
  return r;
}

// =================================================================================================
// G2: the same accumulator over Fq2 = Fq[u]/(u^2+1) with U-form components (Fq2U, fieldu.hpp).
// An Fq2 product is two sums of two Fq products, each with ONE Montgomery reduction (u_mul2):
//     (a0 + a1 u)(b0 + b1 u) = (a0 b0 + a1 (-b1)) + (a0 b1 + a1 b0) u          486 mads, like Karatsuba's 3 x 162,
// but with no subtraction afterwards, so every component stays < 2p.  Domains and invariants are those of
// the G1 accumulator, componentwise:  X < 6p, Y < 2p, ZZ < 2p, ZZZ < 2p, N-form.
struct XYZZU2 {
  Fq2U x, y, zz, zzz;
  ZK_HD static XYZZU2 zero() { return XYZZU2{Fq2U::zero(), Fq2U::zero(), Fq2U::zero(), Fq2U::zero()}; }
  ZK_HD bool is_zero() const { return zz.limbs_all_zero(); }
};

// a * b; K bounds value(b.c1) <= K p.  Components: c0 < (va0 vb0 + va1 K) c + 1, c1 < (va0 vb1 + va1 vb0) c + 1.
template <int K>
ZK_HD Fq2U f2u_mul(const Fq2U& a, const Fq2U& b) {
  FqU nb1 = u_sub<K, 1>(FqU::zero(), b.c1);
  return Fq2U{u_mul2(a.c0, b.c0, a.c1, nb1), u_mul2(a.c0, b.c1, a.c1, b.c0)};
}
template <int K>
ZK_HD Fq2U f2u_sub(const Fq2U& a, const Fq2U& b) { return Fq2U{u_sub<K, 1>(a.c0, b.c0), u_sub<K, 1>(a.c1, b.c1)}; }

ZK_HD Fq2U f2u_from_std(const Fq2& a) { return Fq2U{u_from_std(a.c0), u_from_std(a.c1)}; }

ZK_HD Fq2U f2u_carry(const Fq2U& a) { return Fq2U{u_carry(a.c0), u_carry(a.c1)}; }
ZK_HD Fq2U f2u_dbl(const Fq2U& a) { return Fq2U{u_dbl(a.c0), u_dbl(a.c1)}; }
ZK_HD Fq2U f2u_add(const Fq2U& a, const Fq2U& b) { return Fq2U{u_add(a.c0, b.c0), u_add(a.c1, b.c1)}; }
// a^2; a N-form, K bounds value(a.c1) <= K p.   c0 < (va0 + va1)(va0 + K) c + 1,  c1 < 2 va0 va1 c + 1.
template <int K>
ZK_HD Fq2U f2u_sqr(const Fq2U& a) {
  return Fq2U{u_mul(u_carry(u_add(a.c0, a.c1)), u_sub<K, 1>(a.c0, a.c1)), u_mul(u_dbl(a.c0), a.c1)};
}

// 2 * (x2, y2) over Fq2 in U-form, x2, y2 canonical memory-format values re-packed (u_from_std)   [mdbl-2008-s-1, as xyzzu_double_affine]
ZK_HD XYZZU2 xyzzu2_double_affine(const Fq2U& x2, const Fq2U& y2) {
  const FqU C = UPow2<FqParams, 266>::get();
  const FqU zero = FqU::zero();
  const Fq2U x = Fq2U{u_mul(x2.c0, C), u_mul(x2.c1, C)};     // * 2^261, < 2p, N
  const Fq2U y = Fq2U{u_mul(y2.c0, C), u_mul(y2.c1, C)};
  const Fq2U u = f2u_carry(f2u_dbl(y));                      // < 4p, N
  const Fq2U v = f2u_sqr<4>(u);                              // c0 < 8 * 8 c + 1 < 1.38p,  c1 < 32 c + 1 < 1.19p
  const Fq2U w = f2u_mul<2>(u, v);                           // c0 < (4 * 1.38 + 4 * 2) c + 1 < 1.09p,  c1 < (4 * 1.19 + 4 * 1.38) c + 1 < 1.07p
  const Fq2U s = f2u_mul<2>(x, v);                           // < 1.05p
  const Fq2U xx = f2u_sqr<2>(x);                             // c0 < 4 * 4 c + 1 < 1.1p,  c1 < 8 c + 1 < 1.05p
  const Fq2U m = f2u_carry(f2u_add(f2u_dbl(xx), xx));        // 3 xx < 3.3p, N
  const Fq2U mm = f2u_sqr<4>(m);                             // c0 < 6.6 * 7.3 c + 1 < 1.29p,  c1 < 21.8 c + 1 < 1.13p
  XYZZU2 r;
  r.x = Fq2U{u_sub<4, 2>(mm.c0, u_dbl(s.c0)), u_sub<4, 2>(mm.c1, u_dbl(s.c1))};   // 2s < 2.1p <= 4p, limbs < 2^30;  X < 5.3p
  const Fq2U d = f2u_sub<8>(s, r.x);                         // < 9.05p
  const FqU nd1 = u_sub<16, 1>(zero, d.c1), ny0 = u_sub<2, 1>(zero, y.c0), ny1 = u_sub<2, 1>(zero, y.c1);
  // M D - W y:  c0 = M0 D0 + M1 (16p - D1) + W0 (2p - y0) + W1 y1;  c1 = M0 D1 + M1 D0 + W0 (2p - y1) + W1 (2p - y0)
  r.y.c0 = u_mul4(m.c0, d.c0, m.c1, nd1, w.c0, ny0, w.c1, y.c1);   // (29.9 + 52.8 + 2.2 + 2.2) c + 1 < 1.52p
  r.y.c1 = u_mul4(m.c0, d.c1, m.c1, d.c0, w.c0, ny1, w.c1, ny0);   // (29.9 + 29.9 + 2.2 + 2.2) c + 1 < 1.38p
  r.zz = Fq2U{u_mul(v.c0, C), u_mul(v.c1, C)};               // * 2^266, < 2p
  r.zzz = Fq2U{u_mul(w.c0, C), u_mul(w.c1, C)};
  return r;
}

// 2 * a over Fq2 in U-form   [dbl-2008-s-1, as xyzzu_double: X, Y in one domain, ZZ / ZZZ pass through products -- the bucket accumulator
// (ZZ, ZZZ in the 2^266 domain) and the register form of a record alike].  a: X < 6p, Y < 2p, ZZ, ZZZ < 2p, N-form; result: the same.
ZK_HD XYZZU2 xyzzu2_double(const XYZZU2& a) {
  if (a.is_zero()) return a;
  const FqU zero = FqU::zero();
  const Fq2U u = f2u_carry(f2u_dbl(a.y));                    // < 4p, N
  const Fq2U v = f2u_sqr<4>(u);                              // c0 < 8 * 8 c + 1 < 1.38p,  c1 < 32 c + 1 < 1.19p
  const Fq2U w = f2u_mul<2>(u, v);                           // < 1.09p, < 1.07p
  const Fq2U s = f2u_mul<2>(a.x, v);                         // c0 < (6 * 1.38 + 6 * 2) c + 1 < 1.13p,  c1 < (6 * 1.19 + 6 * 1.38) c + 1 < 1.1p
  const Fq2U xx = f2u_sqr<6>(a.x);                           // c0 < 12 * 12 c + 1 < 1.86p,  c1 < 72 c + 1 < 1.43p
  const Fq2U m = f2u_carry(f2u_add(f2u_dbl(xx), xx));        // 3 xx < 5.6p, N
  const Fq2U mm = f2u_sqr<6>(m);                             // c0 < 11.2 * 11.6 c + 1 < 1.77p,  c1 < 63 c + 1 < 1.38p
  XYZZU2 r;
  r.x = Fq2U{u_sub<4, 2>(mm.c0, u_dbl(s.c0)), u_sub<4, 2>(mm.c1, u_dbl(s.c1))};   // 2s < 2.3p <= 4p, limbs < 2^30;  X < 5.8p
  const Fq2U d = f2u_sub<8>(s, r.x);                         // < 9.2p
  const FqU nd1 = u_sub<16, 1>(zero, d.c1), ny0 = u_sub<2, 1>(zero, a.y.c0), ny1 = u_sub<2, 1>(zero, a.y.c1);
  r.y.c0 = u_mul4(m.c0, d.c0, m.c1, nd1, w.c0, ny0, w.c1, a.y.c1);   // (5.6 * 9.2 + 5.6 * 16 + 2.2 + 2.2) c + 1 < 1.87p
  r.y.c1 = u_mul4(m.c0, d.c1, m.c1, d.c0, w.c0, ny1, w.c1, ny0);     // (51.6 + 51.6 + 2.2 + 2.2) c + 1 < 1.64p
  r.zz = f2u_mul<2>(v, a.zz);                                // < 1.04p
  r.zzz = f2u_mul<2>(w, a.zzz);
  return r;
}


// std XYZZ over Fq2 (every coordinate in the 2^256 domain) -> accumulator domains (rare paths only)
ZK_HD XYZZU2 xyzzu2_from_std(const XYZZ<Fq2>& s) {
  if (s.is_zero()) return XYZZU2::zero();
  const FqU c266 = UPow2<FqParams, 266>::get();  // v*2^256 * 2^266 / 2^261 = v * 2^261
  const FqU c271 = UPow2<FqParams, 271>::get();  // v*2^256 * 2^271 / 2^261 = v * 2^266
  auto cv = [&](const Fq2& v, const FqU& c) { return Fq2U{u_mul(u_from_std(v.c0), c), u_mul(u_from_std(v.c1), c)}; };
  return XYZZU2{cv(s.x, c266), cv(s.y, c266), cv(s.zz, c271), cv(s.zzz, c271)};
}

ZK_HD XYZZ<Fq2> xyzzu2_to_std(const XYZZU2& a) {
  if (a.is_zero()) return XYZZ<Fq2>::zero();
  const FqU c256 = UPow2<FqParams, 256>::get();
  const FqU c251 = UPow2<FqParams, 251>::get();
  auto cv = [&](const Fq2U& v, const FqU& c) { return Fq2{u_to_std_lt2p(u_mul(v.c0, c)), u_to_std_lt2p(u_mul(v.c1, c))}; };
  return XYZZ<Fq2>{cv(a.x, c256), cv(a.y, c256), cv(a.zz, c251), cv(a.zzz, c251)};
}

// acc += (+/-)(x2, y2);  (x2, y2) != infinity, canonical memory-format coordinates.
ZK_HD void xyzzu2_add_mixed(XYZZU2& acc, const Fq2& x2s, const Fq2& y2s, bool negate) {
  const Fq2U x2 = f2u_from_std(x2s);                        // < p
  Fq2U y2 = f2u_from_std(y2s);
  {
    FqU n0 = u_sub<1, 1>(FqU::zero(), y2.c0), n1 = u_sub<1, 1>(FqU::zero(), y2.c1);  // p - y, N
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      y2.c0.l[i] = negate ? n0.l[i] : y2.c0.l[i];
      y2.c1.l[i] = negate ? n1.l[i] : y2.c1.l[i];
    }
  }
  if (acc.is_zero()) {
    const FqU C = UPow2<FqParams, 266>::get();
    acc.x = Fq2U{u_mul(x2.c0, C), u_mul(x2.c1, C)};         // * 2^261, < 2p
    acc.y = Fq2U{u_mul(y2.c0, C), u_mul(y2.c1, C)};
    acc.zz = Fq2U{C, FqU::zero()};                          // (1, 0) * 2^266
    acc.zzz = acc.zz;
    return;
  }
  Fq2U u2 = f2u_mul<2>(x2, acc.zz);                         // (1*2 + 1*2) c + 1 < 1.03p
  Fq2U s2 = f2u_mul<2>(y2, acc.zzz);                        // < 1.03p
  Fq2U p = f2u_sub<8>(u2, acc.x);                           // X < 6p;  P < 10p
  Fq2U r = f2u_sub<2>(s2, acc.y);                           // Y < 2p;  R < 4p
  Fq2U pp;                                                  // P^2 = (P0^2 - P1^2) + 2 P0 P1 u
  pp.c0 = u_mul2(p.c0, p.c0, p.c1, u_sub<16, 1>(FqU::zero(), p.c1));  // (100 + 10*16) c + 1 < 2.56p
  pp.c1 = u_mul(u_dbl(p.c0), p.c1);                         // 200 c + 1 < 2.2p
  Fq2U ppp = f2u_mul<4>(p, pp);                             // c0 < (25.6 + 40) c + 1 < 1.4p, c1 < (22 + 25.6) c + 1 < 1.29p
  Fq2U q = f2u_mul<4>(acc.x, pp);                           // < 1.24p, < 1.17p
  Fq2U rr;                                                  // R^2 = (R0 + R1)(R0 - R1) + 2 R0 R1 u
  rr.c0 = u_mul(u_carry(u_add(r.c0, r.c1)), u_sub<4, 1>(r.c0, r.c1));  // 8 * 8 c + 1 < 1.39p
  rr.c1 = u_mul(u_dbl(r.c0), r.c1);                         // 32 c + 1 < 1.2p
  Fq2U x3, d;
  x3.c0 = u_sub<4, 3>(rr.c0, u_add(ppp.c0, u_dbl(q.c0)));   // PPP + 2Q < 3.9p <= 4p, limbs < 3 * 2^29;  X3 < 5.4p
  x3.c1 = u_sub<4, 3>(rr.c1, u_add(ppp.c1, u_dbl(q.c1)));
  d = f2u_sub<8>(q, x3);                                    // < 9.3p
  FqU ny0 = u_sub<2, 1>(FqU::zero(), acc.y.c0);             // 2p - Y0
  FqU ny1 = u_sub<2, 1>(FqU::zero(), acc.y.c1);
  FqU nd1 = u_sub<16, 1>(FqU::zero(), d.c1);                // 16p - D1
  Fq2U y3;                                                  // R*D - Y1*PPP
  y3.c0 = u_mul4(r.c0, d.c0, r.c1, nd1, ny0, ppp.c0, acc.y.c1, ppp.c1);  // (37 + 64 + 2.8 + 2.6) c + 1 < 1.65p
  y3.c1 = u_mul4(r.c0, d.c1, r.c1, d.c0, ny0, ppp.c1, ny1, ppp.c0);      // (37 + 37 + 2.6 + 2.8) c + 1 < 1.48p
  Fq2U zz3 = f2u_mul<4>(acc.zz, pp);                        // < 1.08p
  Fq2U zzz3 = f2u_mul<2>(acc.zzz, ppp);                     // PPP1 < 1.29p <= 2p;  < 1.03p
  if (u_is_zero_lt2p(zz3.c0) && u_is_zero_lt2p(zz3.c1)) {
    // P == 0: same x.  Same point -> double (ec.rs:483-485); opposite -> infinity (ec.rs:487).
    if (u_is_zero_lt8p(r.c0) && u_is_zero_lt8p(r.c1)) {
      acc = xyzzu2_double_affine(x2, y2);                     // (y2 carries the sign)
    } else {
      acc = XYZZU2::zero();
    }
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.zz = zz3;
  acc.zzz = zzz3;
}

// ---- G2 records in the R domain (see the G1 section "XYZZ RECORDS IN THE R DOMAIN"): componentwise the same thing.
//   Register form (XYZZU2, every component 2^261 domain): X < 6p, Y < 2p, ZZ < 2p, ZZZ < 2p, N-form.
ZK_HD XYZZ<Fq2> xyzzu_to_r(const XYZZU2& a) {          // bucket accumulator (X, Y: 2^261; ZZ, ZZZ: 2^266) -> record
  if (a.is_zero()) return XYZZ<Fq2>::zero();
  const FqU c256 = UPow2<FqParams, 256>::get();
  XYZZ<Fq2> r;
  r.x = Fq2{u_to_std_lt32p(a.x.c0), u_to_std_lt32p(a.x.c1)};
  r.y = Fq2{u_to_std_lt2p(a.y.c0), u_to_std_lt2p(a.y.c1)};
  r.zz = Fq2{u_to_std_lt2p(u_mul(a.zz.c0, c256)), u_to_std_lt2p(u_mul(a.zz.c1, c256))};
  r.zzz = Fq2{u_to_std_lt2p(u_mul(a.zzz.c0, c256)), u_to_std_lt2p(u_mul(a.zzz.c1, c256))};
  return r;
}
ZK_HD XYZZU2 xyzzu_from_r(const XYZZ<Fq2>& s) {          // record -> bucket accumulator (see the G1 form)
  if (s.is_zero()) return XYZZU2::zero();
  const FqU c266 = UPow2<FqParams, 266>::get();
  return XYZZU2{f2u_from_std(s.x), f2u_from_std(s.y), Fq2U{u_mul(u_from_std(s.zz.c0), c266), u_mul(u_from_std(s.zz.c1), c266)},
                Fq2U{u_mul(u_from_std(s.zzz.c0), c266), u_mul(u_from_std(s.zzz.c1), c266)}};
}
ZK_HD XYZZU2 xyzzr_load(const XYZZ<Fq2>& s) {
  return XYZZU2{f2u_from_std(s.x), f2u_from_std(s.y), f2u_from_std(s.zz), f2u_from_std(s.zzz)};
}
ZK_HD XYZZ<Fq2> xyzzr_store(const XYZZU2& a) {
  if (a.is_zero()) return XYZZ<Fq2>::zero();
  XYZZ<Fq2> r;
  r.x = Fq2{u_to_std_lt32p(a.x.c0), u_to_std_lt32p(a.x.c1)};
  r.y = Fq2{u_to_std_lt2p(a.y.c0), u_to_std_lt2p(a.y.c1)};
  r.zz = Fq2{u_to_std_lt2p(a.zz.c0), u_to_std_lt2p(a.zz.c1)};
  r.zzz = Fq2{u_to_std_lt2p(a.zzz.c0), u_to_std_lt2p(a.zzz.c1)};
  return r;
}
ZK_HD XYZZ<Fq2> xyzzr_to_std(const XYZZ<Fq2>& s) {      // record -> memory-format XYZZ (2^256 domain)
  if (s.is_zero()) return s;
  const FqU c256 = UPow2<FqParams, 256>::get();
  auto cv = [&](const Fq2& v) { return Fq2{u_to_std_lt2p(u_mul(u_from_std(v.c0), c256)), u_to_std_lt2p(u_mul(u_from_std(v.c1), c256))}; };
  return XYZZ<Fq2>{cv(s.x), cv(s.y), cv(s.zz), cv(s.zzz)};
}
ZK_HD XYZZU2 xyzzr2_from_std(const XYZZ<Fq2>& s) {      // memory-format XYZZ -> register form (rare paths only)
  if (s.is_zero()) return XYZZU2::zero();
  const FqU c266 = UPow2<FqParams, 266>::get();           // v*2^256 * 2^266 / 2^261 = v * 2^261
  auto cv = [&](const Fq2& v) { return Fq2U{u_mul(u_from_std(v.c0), c266), u_mul(u_from_std(v.c1), c266)}; };
  return XYZZU2{cv(s.x), cv(s.y), cv(s.zz), cv(s.zzz)};
}

// acc += o, both in register form  [add-2008-s over Fq2; the component bounds follow xyzzu2_add_mixed]
ZK_HD void xyzzr_add(XYZZU2& acc, const XYZZU2& o) {
  if (o.is_zero()) return;
  if (acc.is_zero()) {
    acc = o;
    return;
  }
  Fq2U u1 = f2u_mul<2>(acc.x, o.zz);                        // (6*2 + 6*2) c + 1 < 1.15p
  Fq2U u2 = f2u_mul<2>(o.x, acc.zz);                        // < 1.15p
  Fq2U s1 = f2u_mul<2>(acc.y, o.zzz);                       // (2*2 + 2*2) c + 1 < 1.05p
  Fq2U s2 = f2u_mul<2>(o.y, acc.zzz);                       // < 1.05p
  Fq2U p = f2u_sub<2>(u2, u1);                              // U1 < 2p;  P < 3.15p <= 4p, N
  Fq2U r = f2u_sub<2>(s2, s1);                              // S1 < 2p;  R < 3.05p <= 4p, N
  Fq2U pp;                                                  // P^2 = (P0^2 - P1^2) + 2 P0 P1 u
  pp.c0 = u_mul2(p.c0, p.c0, p.c1, u_sub<4, 1>(FqU::zero(), p.c1));   // (16 + 4*4) c + 1 < 1.2p
  pp.c1 = u_mul(u_dbl(p.c0), p.c1);                         // 32 c + 1 < 1.2p
  Fq2U ppp = f2u_mul<2>(p, pp);                             // c0 < (4.8 + 8) c + 1 < 1.08p, c1 < 9.6 c + 1 < 1.06p
  Fq2U q = f2u_mul<2>(u1, pp);                              // < 1.03p
  Fq2U rr;                                                  // R^2 = (R0 + R1)(R0 - R1) + 2 R0 R1 u
  rr.c0 = u_mul(u_carry(u_add(r.c0, r.c1)), u_sub<4, 1>(r.c0, r.c1));  // 8 * 8 c + 1 < 1.39p
  rr.c1 = u_mul(u_dbl(r.c0), r.c1);                         // 32 c + 1 < 1.2p
  Fq2U x3, d;
  x3.c0 = u_sub<4, 3>(rr.c0, u_add(ppp.c0, u_dbl(q.c0)));   // PPP + 2Q < 3.2p <= 4p, limbs < 3 * 2^29;  X3 < 5.4p
  x3.c1 = u_sub<4, 3>(rr.c1, u_add(ppp.c1, u_dbl(q.c1)));
  d = f2u_sub<8>(q, x3);                                    // < 9.1p
  FqU ns0 = u_sub<2, 1>(FqU::zero(), s1.c0);                // 2p - S1_0
  FqU ns1 = u_sub<2, 1>(FqU::zero(), s1.c1);
  FqU nd1 = u_sub<16, 1>(FqU::zero(), d.c1);                // 16p - D1
  Fq2U y3;                                                  // R*D - S1*PPP
  y3.c0 = u_mul4(r.c0, d.c0, r.c1, nd1, ns0, ppp.c0, s1.c1, ppp.c1);   // (36.4 + 64 + 2.2 + 1.2) c + 1 < 1.63p
  y3.c1 = u_mul4(r.c0, d.c1, r.c1, d.c0, ns0, ppp.c1, ns1, ppp.c0);    // (36.4 + 36.4 + 2.2 + 2.2) c + 1 < 1.47p
  Fq2U zz3 = f2u_mul<2>(f2u_mul<2>(acc.zz, o.zz), pp);      // inner < 1.05p;  < 1.03p
  Fq2U zzz3 = f2u_mul<2>(f2u_mul<2>(acc.zzz, o.zzz), ppp);  // < 1.03p
  if (u_is_zero_lt2p(zz3.c0) && u_is_zero_lt2p(zz3.c1)) {
    // P == 0: same x.  Same point -> double; opposite -> infinity.  Not rare in the bucket reduction of a SHORT call: a running sum
    // whose first non-empty bucket is followed by an empty one adds that bucket's sum to itself (2^12 points over 2^11 buckets per
    // window: 13 % of the buckets are empty) -- on the saturated-limb formulas of curve.hpp, through the memory format's domain and
    // back, the doubling cost four additions and the G2 reduce of a 2^12-point call 0.29 ms instead of 0.17.
    if (u_is_zero_lt8p(r.c0) && u_is_zero_lt8p(r.c1)) acc = xyzzu2_double(acc);
    else acc = XYZZU2::zero();
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.zz = zz3;
  acc.zzz = zzz3;
}

#if defined(__HIPCC__)
// ---- the G2 addition by a QUAD of lanes (see xyzzr_add_quad above).  An Fq2 product is two independent u_mul2 (its components), and
// the products of an addition come in independent pairs, so a round is FOUR u_mul2 side by side: role bit 0 = the component, role bit 1
// = which of the round's two Fq2 products.
//     round 1: U1 = X1 ZZ2, U2 = X2 ZZ1      round 2: S1 = Y1 ZZZ2, S2 = Y2 ZZZ1     round 3: ZZ1 ZZ2, ZZZ1 ZZZ2
//     round 4: PP = P^2, RR = R^2 (four single / double products)                    round 5: PPP = P PP, Q = U1 PP
//     round 6: ZZ3 = (ZZ1 ZZ2) PP, ZZZ3 = (ZZZ1 ZZZ2) PPP                            round 7: Y3 (one u_mul4 per component: two lanes)
// six u_mul2 and one u_mul4 in sequence instead of 28 + 2 of them.  Operands, bounds and results are those of xyzzr_add(XYZZU2&, ..).
template <int K>
__device__ __forceinline__ Fq2U quad_get2(const FqU& t) { return Fq2U{quad_get<K>(t), quad_get<K + 1>(t)}; }
__device__ __forceinline__ FqU pick2(bool second, const FqU& a, const FqU& b) {
  FqU r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t va = a.l[i], vb = b.l[i];
    r.l[i] = second ? vb : va;
  }
  return r;
}
// lanes 0, 1: the components of a1 * b1;  lanes 2, 3: those of a2 * b2   (nb1 / nb2: K p - b.c1, what f2u_mul<K> multiplies a.c1 by)
__device__ __forceinline__ FqU quad_f2u_mul_pair(uint32_t role, const Fq2U& a1, const Fq2U& b1, const FqU& nb1, const Fq2U& a2, const Fq2U& b2,
                                                  const FqU& nb2) {
  const bool odd = (role & 1u) != 0, high = (role & 2u) != 0;
  const FqU a0 = pick2(high, a1.c0, a2.c0), a1c = pick2(high, a1.c1, a2.c1);
  const FqU b0 = pick2(high, b1.c0, b2.c0), b1c = pick2(high, b1.c1, b2.c1), nb = pick2(high, nb1, nb2);
  return u_mul2(a0, pick2(odd, b0, b1c), a1c, pick2(odd, nb, b0));   // c0 = a0 b0 + a1 (-b1);  c1 = a0 b1 + a1 b0
}
template <int K>
__device__ __forceinline__ XYZZU2 quad_fetch(const XYZZU2& a) {
  return XYZZU2{Fq2U{quad_get<K>(a.x.c0), quad_get<K>(a.x.c1)}, Fq2U{quad_get<K>(a.y.c0), quad_get<K>(a.y.c1)},
                Fq2U{quad_get<K>(a.zz.c0), quad_get<K>(a.zz.c1)}, Fq2U{quad_get<K>(a.zzz.c0), quad_get<K>(a.zzz.c1)}};
}
__device__ __forceinline__ XYZZU2 xyzzr_add_quad(const XYZZU2 acc, const XYZZU2 o, uint32_t role) {
  XYZZU2 res = acc;
  if (!o.is_zero()) {
    res = o;
    if (!acc.is_zero()) {
      const bool odd = (role & 1u) != 0, high = (role & 2u) != 0;
      const FqU zero = FqU::zero();
      FqU t = quad_f2u_mul_pair(role, acc.x, o.zz, u_sub<2, 1>(zero, o.zz.c1), o.x, acc.zz, u_sub<2, 1>(zero, acc.zz.c1));
      const Fq2U u1 = quad_get2<0>(t), u2 = quad_get2<2>(t);   // < 1.15p
      t = quad_f2u_mul_pair(role, acc.y, o.zzz, u_sub<2, 1>(zero, o.zzz.c1), o.y, acc.zzz, u_sub<2, 1>(zero, acc.zzz.c1));
      const Fq2U s1 = quad_get2<0>(t), s2 = quad_get2<2>(t);   // < 1.05p
      const Fq2U p = f2u_sub<2>(u2, u1);                        // U1 < 2p;  P < 3.15p <= 4p, N
      const Fq2U r = f2u_sub<2>(s2, s1);                        // S1 < 2p;  R < 3.05p <= 4p, N
      t = quad_f2u_mul_pair(role, acc.zz, o.zz, u_sub<2, 1>(zero, o.zz.c1), acc.zzz, o.zzz, u_sub<2, 1>(zero, o.zzz.c1));
      const Fq2U zz12 = quad_get2<0>(t), zzz12 = quad_get2<2>(t);   // < 1.05p
      // round 4: lane 0: P0 P0 + P1 (4p - P1);  lane 1: (2 P0) P1;  lane 2: (R0 + R1)(R0 - R1 + 4p);  lane 3: (2 R0) R1
      {
        const FqU f1 = quad_pick(role, p.c0, u_dbl(p.c0), u_carry(u_add(r.c0, r.c1)), u_dbl(r.c0));
        const FqU f2 = quad_pick(role, p.c0, p.c1, u_sub<4, 1>(r.c0, r.c1), r.c1);
        const FqU np1 = u_sub<4, 1>(zero, p.c1);
        FqU f3, f4;
        const bool first = role == 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          f3.l[i] = first ? p.c1.l[i] : 0u;
          f4.l[i] = first ? np1.l[i] : 0u;
        }
        t = u_mul2(f1, f2, f3, f4);   // (the doubled operands have limbs < 2^30 and a zero second pair: u_mul's column bound)
      }
      const Fq2U pp = quad_get2<0>(t), rr = quad_get2<2>(t);       // PP < 1.2p, RR.c0 < 1.39p, RR.c1 < 1.2p
      const FqU npp1 = u_sub<2, 1>(zero, pp.c1);
      t = quad_f2u_mul_pair(role, p, pp, npp1, u1, pp, npp1);
      const Fq2U ppp = quad_get2<0>(t), q = quad_get2<2>(t);       // PPP < 1.08p / 1.06p, Q < 1.03p
      t = quad_f2u_mul_pair(role, zz12, pp, npp1, zzz12, ppp, u_sub<2, 1>(zero, ppp.c1));
      const Fq2U zz3 = quad_get2<0>(t), zzz3 = quad_get2<2>(t);    // < 1.03p
      Fq2U x3;
      x3.c0 = u_sub<4, 3>(rr.c0, u_add(ppp.c0, u_dbl(q.c0)));      // PPP + 2Q < 3.2p <= 4p, limbs < 3 * 2^29;  X3 < 5.4p
      x3.c1 = u_sub<4, 3>(rr.c1, u_add(ppp.c1, u_dbl(q.c1)));
      const Fq2U d = f2u_sub<8>(q, x3);                            // < 9.1p
      const FqU ns0 = u_sub<2, 1>(zero, s1.c0), ns1 = u_sub<2, 1>(zero, s1.c1), nd1 = u_sub<16, 1>(zero, d.c1);
      // round 7: even lanes  R0 D0 + R1 (16p - D1) + (2p - S1_0) PPP0 + S1_1 PPP1;  odd lanes  R0 D1 + R1 D0 + (2p - S1_0) PPP1 + (2p - S1_1) PPP0
      t = u_mul4(r.c0, pick2(odd, d.c0, d.c1), r.c1, pick2(odd, nd1, d.c0), ns0, pick2(odd, ppp.c0, ppp.c1), pick2(odd, s1.c1, ns1),
                 pick2(odd, ppp.c1, ppp.c0));
      (void)high;
      res = XYZZU2{x3, quad_get2<0>(t), zz3, zzz3};
      if (u_is_zero_lt2p(zz3.c0) && u_is_zero_lt2p(zz3.c1)) {
        // P == 0: same x.  Same point -> double; opposite -> infinity (uniform over the quad; the doubling as in xyzzr_add)
        res = XYZZU2::zero();
        if (u_is_zero_lt8p(r.c0) && u_is_zero_lt8p(r.c1)) res = xyzzu2_double(acc);
      }
    }
  }
  return res;
}
#endif  // __HIPCC__

// =================================================================================================
// G2 Jacobian accumulator on U-form Fq2, for SCALAR MULTIPLICATION (batch_exp, the terms of the QAP sums): the G1 design above,
// componentwise.  Fq2 squaring is (a0 + a1)(a0 - a1) + 2 a0 a1 u (two products), an Fq2 product two fused pairs (f2u_mul), and
// the two "r (V - X3) - 2 S1 J"-shaped lines take ONE Montgomery reduction per component (u_mul3 / u_mul4).
//   Invariants (N-form, per component):  X < 7p,  Y <= 3p,  Z < 3p;   infinity is Z == literal zero limbs.
//   Every coordinate lives in the 2^261 domain.  c = p / 2^261 < 0.006:  u_mul(a, b) < (a/p)(b/p) c p + p.
struct JacU2 {
  Fq2U x, y, z;
  ZK_HD static JacU2 zero() { return JacU2{Fq2U::zero(), Fq2U::zero(), Fq2U::zero()}; }
  ZK_HD bool is_zero() const { return z.limbs_all_zero(); }
};


// 2 * a   [dbl-2009-l, D = 4 X B as a product like jacu_double]
ZK_HD JacU2 jacu2_double(const JacU2& a) {
  if (a.is_zero()) return a;
  const Fq2U A = f2u_sqr<8>(a.x);                            // X1 < 7 <= 8:  c0 < 14*15c + 1 < 2.26p,  c1 < 98c + 1 < 1.59p
  const Fq2U B = f2u_sqr<4>(a.y);                            // Y1 <= 3 <= 4:  c0 < 6*7c + 1 < 1.26p,  c1 < 18c + 1 < 1.11p
  const Fq2U C = f2u_sqr<2>(B);                              // < 1.05p
  const Fq2U X4 = f2u_carry(f2u_dbl(f2u_dbl(a.x)));          // 4X < 28p, N
  const Fq2U D = f2u_mul<2>(X4, B);                          // B1 < 1.11 <= 2:  c0 < (28*1.26 + 28*2)c + 1 < 1.55p,  c1 < 1.40p
  const Fq2U E = f2u_carry(f2u_add(f2u_dbl(A), A));          // 3A < 6.8p, N
  const Fq2U F = f2u_sqr<8>(E);                              // c0 < 13.6*14.8c + 1 < 2.21p,  c1 < 1.56p
  JacU2 r;
  r.x = Fq2U{u_sub<4, 2>(F.c0, u_dbl(D.c0)), u_sub<4, 2>(F.c1, u_dbl(D.c1))};   // 2D < 3.1p <= 4p, limbs < 2^30;  X3 < 6.21p
  const Fq2U dmx = f2u_sub<8>(D, r.x);                       // X3 <= 8p;  < 9.55p, N
  const Fq2U negc = f2u_sub<2>(Fq2U::zero(), C);             // 2p - C in (0, 2p], N
  const FqU nd1 = u_sub<16, 1>(FqU::zero(), dmx.c1);         // 16p - d1
  const FqU c8 = UPow2<FqParams, 264>::get();                // 8 * 2^261:  negc * c8 / 2^261 = -8C
  r.y.c0 = u_mul3(E.c0, dmx.c0, E.c1, nd1, negc.c0, c8);     // E (D - X3) - 8C:  (64.9 + 108.8 + 2)c + 1 < 2.06p
  r.y.c1 = u_mul3(E.c0, dmx.c1, E.c1, dmx.c0, negc.c1, c8);  // (64.9 + 64.9 + 2)c + 1 < 1.80p
  r.z = f2u_mul<4>(f2u_carry(f2u_dbl(a.y)), a.z);            // 2 Y Z:  2Y <= 6p, Z1 < 3 <= 4:  (18 + 24)c + 1 < 1.26p
  return r;
}

// A table entry for windowed scalar multiplication: the point with Z^2 and Z^3 (80 words used of 92: 368 bytes).
struct alignas(16) JacTabU2 {
  Fq2U x, y, z, zz, zzz;
  uint32_t pad[2];
};
ZK_HD JacTabU2 jacu2_tab_entry(const JacU2& q) {
  JacTabU2 t;
  t.x = q.x;
  t.y = q.y;
  t.z = q.z;
  t.zz = f2u_sqr<4>(q.z);                                    // Z < 3 <= 4:  < 1.26p, < 1.11p
  t.zzz = f2u_mul<2>(q.z, t.zz);                             // ZZ1 < 1.11 <= 2:  (3*1.26 + 3*2)c + 1 < 1.06p
  t.pad[0] = t.pad[1] = 0;
  return t;
}

// acc += (+/-) t;  t != infinity   [add-2007-bl with the entry's Z^2, Z^3 given]
ZK_HD void jacu2_add_tab(JacU2& acc, const JacTabU2& t, bool negate) {
  Fq2U y2 = t.y;
  {
    const Fq2U ny = f2u_sub<3>(Fq2U::zero(), t.y);           // 3p - Y2 in [0, 3p], N   (Y2 <= 3p)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      y2.c0.l[i] = negate ? ny.c0.l[i] : t.y.c0.l[i];
      y2.c1.l[i] = negate ? ny.c1.l[i] : t.y.c1.l[i];
    }
  }
  if (acc.is_zero()) {
    acc.x = t.x;
    acc.y = y2;
    acc.z = t.z;
    return;
  }
  const Fq2U z1z1 = f2u_sqr<4>(acc.z);                       // < 1.26p, < 1.11p
  const Fq2U u1 = f2u_mul<2>(acc.x, t.zz);                   // (7*1.26 + 7*2)c + 1 < 1.14p
  const Fq2U u2 = f2u_mul<2>(t.x, z1z1);                     // < 1.14p
  const Fq2U s1 = f2u_mul<2>(acc.y, t.zzz);                  // (3*1.06 + 3*2)c + 1 < 1.06p
  const Fq2U s2 = f2u_mul<2>(y2, f2u_mul<2>(acc.z, z1z1));   // Z1^3 < 1.06p;  < 1.06p
  const Fq2U h = f2u_sub<2>(u2, u1);                         // < 3.14p, N
  const Fq2U hh = f2u_sqr<4>(h);                             // H1 < 3.14 <= 4:  c0 < 6.28*7.14c + 1 < 1.27p,  c1 < 1.12p
  const Fq2U i4 = f2u_carry(f2u_dbl(f2u_dbl(hh)));           // I = 4 HH < 5.1p, N
  const Fq2U j = f2u_mul<8>(h, i4);                          // I1 < 4.5 <= 8:  (3.14*5.1 + 3.14*8)c + 1 < 1.25p
  const Fq2U r = f2u_carry(f2u_dbl(f2u_sub<2>(s2, s1)));     // 2 (S2 - S1 + 2p) < 6.12p, N
  const Fq2U v = f2u_mul<8>(u1, i4);                         // < 1.09p
  const Fq2U rr = f2u_sqr<8>(r);                             // r1 < 6.12 <= 8:  c0 < 12.24*14.12c + 1 < 2.04p,  c1 < 1.45p
  Fq2U x3;                                                   // r^2 - J - 2V:  J + 2V < 3.43p <= 4p, limbs < 3 * 2^29;  X3 < 6.04p
  x3.c0 = u_sub<4, 3>(rr.c0, u_add(j.c0, u_dbl(v.c0)));
  x3.c1 = u_sub<4, 3>(rr.c1, u_add(j.c1, u_dbl(v.c1)));
  const Fq2U vmx = f2u_sub<8>(v, x3);                        // < 9.09p, N
  const FqU nvmx1 = u_sub<16, 1>(FqU::zero(), vmx.c1);       // 16p - (V - X3)_1
  const FqU n2s0 = u_sub<4, 2>(FqU::zero(), u_dbl(s1.c0));   // 4p - 2 S1_0 in (0, 4p], N
  const FqU n2s1 = u_sub<4, 2>(FqU::zero(), u_dbl(s1.c1));
  const FqU p2s1 = u_carry(u_dbl(s1.c1));                    // 2 S1_1 < 2.12p, N
  Fq2U y3;                                                   // r (V - X3) - 2 S1 J
  y3.c0 = u_mul4(r.c0, vmx.c0, r.c1, nvmx1, n2s0, j.c0, p2s1, j.c1);   // (55.6 + 97.9 + 5 + 2.7)c + 1 < 1.97p
  y3.c1 = u_mul4(r.c0, vmx.c1, r.c1, vmx.c0, n2s0, j.c1, n2s1, j.c0);  // (55.6 + 55.6 + 5 + 5)c + 1 < 1.73p
  const Fq2U z3 = f2u_mul<4>(f2u_mul<4>(f2u_carry(f2u_dbl(acc.z)), t.z), h);   // 2 Z1 Z2 < 1.26p;  times H (H1 < 3.14 <= 4):  < 1.06p
  if (u_is_zero_lt2p(z3.c0) && u_is_zero_lt2p(z3.c1)) {
    // H == 0: same x.  Same point -> double; opposite -> infinity (ec.rs:398-408).
    if (u_is_zero_lt8p(r.c0) && u_is_zero_lt8p(r.c1)) acc = jacu2_double(JacU2{t.x, y2, t.z});
    else acc = JacU2::zero();
    return;
  }
  acc.x = x3;
  acc.y = y3;
  acc.z = z3;
}

// psi of a table entry (glv.hpp): (cx conj(X), cy conj(Y), conj(Z), conj(Z^2), conj(Z^3)) -- conj negates the u-component.
// cx, cy: the constants in U-form, 2^261 domain, components < 2p.  Result within the entry invariants
// (X < 1.2p, Y < 1.1p, Z_1 <= 3p, ZZ_1, ZZZ_1 <= 2p).
ZK_HD JacTabU2 jacu2_tab_psi(const JacTabU2& e, const Fq2U& cx, const Fq2U& cy) {
  JacTabU2 t;
  t.x = f2u_mul<2>(Fq2U{e.x.c0, u_sub<8, 1>(FqU::zero(), e.x.c1)}, cx);    // X < 7p:  (7*2 + 8*2)c + 1 < 1.2p
  t.y = f2u_mul<2>(Fq2U{e.y.c0, u_sub<4, 1>(FqU::zero(), e.y.c1)}, cy);    // Y <= 3p: (3*2 + 4*2)c + 1 < 1.1p
  t.z = Fq2U{e.z.c0, u_sub<3, 1>(FqU::zero(), e.z.c1)};                    // Z < 3p
  t.zz = Fq2U{e.zz.c0, u_sub<2, 1>(FqU::zero(), e.zz.c1)};                 // ZZ, ZZZ < 2p
  t.zzz = Fq2U{e.zzz.c0, u_sub<2, 1>(FqU::zero(), e.zzz.c1)};
  t.pad[0] = t.pad[1] = 0;
  return t;
}

// raw affine point (memory format, not infinity) -> table entry of 1 * P in the 2^261 domain
ZK_HD JacTabU2 jacu2_tab_from_affine(const Fq2& x, const Fq2& y) {
  const FqU C = UPow2<FqParams, 266>::get();                 // x*2^256 * 2^266 / 2^261 = x * 2^261
  JacU2 q;
  q.x = Fq2U{u_mul(u_from_std(x.c0), C), u_mul(u_from_std(x.c1), C)};   // < 2p
  q.y = Fq2U{u_mul(u_from_std(y.c0), C), u_mul(u_from_std(y.c1), C)};
  q.z = Fq2U{UPow2<FqParams, 261>::get(), FqU::zero()};      // one
  return jacu2_tab_entry(q);
}

// accumulator -> memory-format Jacobian (canonical coordinates, 2^256 domain); infinity -> z == 0 (x, y zero too)
ZK_HD Jacobian<Fq2> jacu2_to_std(const JacU2& a) {
  Jacobian<Fq2> r{Fq2::zero(), Fq2::zero(), Fq2::zero()};
  if (a.is_zero()) return r;
  const FqU c256 = UPow2<FqParams, 256>::get();              // v*2^261 * 2^256 / 2^261 = v * 2^256
  auto cv = [&](const Fq2U& v) { return Fq2{u_to_std_lt2p(u_mul(v.c0, c256)), u_to_std_lt2p(u_mul(v.c1, c256))}; };
  r.x = cv(a.x);
  r.y = cv(a.y);
  r.z = cv(a.z);
  return r;
}

#if defined(__HIPCC__)
// ---- the G2 bucket accumulator held by a PAIR of lanes (lanes 2k, 2k + 1 of a wave; msm_impl.hpp: msm_accumulate_pair_kernel).
// The one-lane Fq2 accumulation needs the accumulator (72 registers), the operands and the temporaries of 8M + 2S over Fq2 at
// once: 256 VGPRs + ~150 AGPRs, ONE wave per SIMD, and its multipliers run at 61 % of their rate (G1, four waves: 76 - 83 %).
// Here the EVEN lane keeps (X, ZZ) and the ODD lane (Y, ZZZ) -- 36 registers each -- and the mixed addition is five rounds in
// which both lanes run the SAME product routine on their own operands (no divergence), all nine products of the addition
// exactly once over the pair:
//     round 1  f2u_mul    even: U2 = x2 ZZ            odd: S2 = y2 ZZZ          then  P = U2 - X  |  R = S2 - Y
//     round 2  square     even: PP = P^2              odd: RR = R^2
//     round 3  f2u_mul    even: Q = X PP              odd: PPP = P PP           then  X3 = RR - PPP - 2Q,  D = Q - X3  (both lanes)
//     round 4  f2u_mul    even: ZZ3 = ZZ PP           odd: ZZZ3 = ZZZ PPP
//     round 5  u_mul4     even: Y3.c0                 odd: Y3.c1                (R D - Y PPP, one reduction per component)
// 18 Fq products + 9 reductions per lane, 36 + 18 per pair (one lane: 37 + 18).  Values cross between the two lanes by DPP
// quad permutes (v_mov_b32 quad_perm, full rate, no LDS): P, R, PP, RR, Q, PPP, Y and Y3.c0 -- ~170 moves and ~110 selects per
// addition against ~2400 multiplier instructions per lane.  Each lane gathers only ITS coordinate of the base (64 B).
// Bounds are those of xyzzu2_add_mixed except where noted.  BOTH lanes of a pair must be active wherever this is called.
constexpr int PAIR_SWAP = 0xB1, PAIR_EVEN = 0xA0, PAIR_ODD = 0xF5;   // quad_perm [1,0,3,2] / [0,0,2,2] / [1,1,3,3]
template <int CTRL>
__device__ __forceinline__ uint32_t pair_dpp(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
#else
  return v;   // (host pass of hipcc: never executed)
#endif
}
template <int CTRL>
__device__ __forceinline__ FqU pair_dpp(const FqU& a) {
  FqU r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = pair_dpp<CTRL>(a.l[i]);
  return r;
}
template <int CTRL>
__device__ __forceinline__ Fq2U pair_dpp(const Fq2U& a) { return Fq2U{pair_dpp<CTRL>(a.c0), pair_dpp<CTRL>(a.c1)}; }
__device__ __forceinline__ Fq2U pick2(bool second, const Fq2U& a, const Fq2U& b) { return Fq2U{pick2(second, a.c0, b.c0), pick2(second, a.c1, b.c1)}; }
// a + k p - b with k = KE on the even lane and KO on the odd one (u_sub's preconditions with the lane's k; S as there)
template <int KE, int KO, int S>
__device__ __forceinline__ FqU u_sub_role(bool odd, const FqU& a, const FqU& b) {
  FqU t;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t ke = USubConst<FqParams, KE, S>::limb(i), ko = USubConst<FqParams, KO, S>::limb(i);
    t.l[i] = a.l[i] + (odd ? ko : ke) - b.l[i];
  });
  return u_carry(t);
}

#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_PAIR_ROUND() __builtin_amdgcn_sched_barrier(0)
#else
#define ZK_PAIR_ROUND() ((void)0)
#endif

struct PairAcc2 {
  Fq2U a, z;   // even lane: (X, ZZ), odd lane: (Y, ZZZ);  domains and bounds of XYZZU2.  Infinity: z == literal zeros in BOTH lanes
  __device__ __forceinline__ static PairAcc2 zero() { return PairAcc2{Fq2U::zero(), Fq2U::zero()}; }
  __device__ __forceinline__ bool is_zero() const { return z.limbs_all_zero(); }
};

// ---- the same split for G1 (short calls: msm_accumulate_pair_g1_kernel): U2 | S2, PP | RR, Q | PPP, ZZ3 | ZZZ3 as single products, then
// Y3 = R D - Y PPP (one u_mul2) in both lanes: 3 u_mul + 1 u_sqr + 1 u_mul2 deep (891 multiplier instructions) instead of the
// 1467 of the one-lane addition.  No throughput to gain (the pair issues 1782) -- it is for launches that last as long as their
// lanes' chains of dependent additions.  Bounds of xyzzu_add_mixed.
struct PairAcc1 {
  FqU a, z;   // even lane: (X, ZZ), odd lane: (Y, ZZZ);  domains and bounds of XYZZU.  Infinity: z == literal zeros in BOTH lanes
  __device__ __forceinline__ static PairAcc1 zero() { return PairAcc1{FqU::zero(), FqU::zero()}; }
  __device__ __forceinline__ bool is_zero() const { return z.limbs_all_zero(); }
};
__device__ __forceinline__ PairAcc1 pair_add_mixed(PairAcc1 acc, const Fq& c2s, bool negate, bool odd) {
  const FqU zero = FqU::zero();
  FqU c2 = u_from_std(c2s);                                  // < p
  c2 = pick2(negate && odd, c2, u_sub<1, 1>(zero, c2));      // odd lane: (+/-) y2
  if (acc.is_zero()) {                                       // (uniform over the pair)
    const FqU C = UPow2<FqParams, 266>::get();
    acc.a = u_mul(c2, C);                                    // * 2^261, < 2p
    acc.z = C;                                               // 1 * 2^266
    return acc;
  }
  const FqU m = u_mul(c2, acc.z);                            // even: U2, odd: S2;  < 2p
  const FqU d1 = u_sub_role<8, 2, 1>(odd, m, acc.a);         // even: P < 10p (X < 6p);  odd: R < 4p (Y < 2p)
  const FqU sq = u_sqr(d1);                                  // even: PP < 100 c + 1 < 1.6p;  odd: RR < 16 c + 1 < 1.1p
  const FqU pp = pair_dpp<PAIR_EVEN>(sq), rr = pair_dpp<PAIR_ODD>(sq);
  const FqU p = pair_dpp<PAIR_EVEN>(d1), r = pair_dpp<PAIR_ODD>(d1);
  const FqU m3 = u_mul(pick2(odd, acc.a, p), pp);            // even: Q = X PP < 1.06p;  odd: PPP = P PP < 1.1p
  const FqU q = pair_dpp<PAIR_EVEN>(m3), ppp = pair_dpp<PAIR_ODD>(m3);
  const FqU z3 = u_mul(acc.z, pick2(odd, pp, ppp));          // even: ZZ3 = ZZ PP;  odd: ZZZ3 = ZZZ PPP;  < 2p
  const FqU x3 = u_sub<4, 3>(rr, u_add(ppp, u_dbl(q)));      // PPP + 2Q < 3.3p <= 4p, limbs < 3 * 2^29;  X3 < 5.1p
  const FqU d = u_sub<8, 1>(q, x3);                          // < 9.1p
  const FqU y = pair_dpp<PAIR_ODD>(acc.a);
  const FqU y3 = u_mul2(r, d, u_sub<2, 1>(zero, y), ppp);    // R D - Y PPP < 1.24p  (both lanes; the odd one keeps it)
  if (u_is_zero_lt2p(z3)) {
    // P == 0 (uniform over the pair, see the G2 form): same point -> double (ec.rs:483-485), opposite -> infinity (ec.rs:487)
    PairAcc1 res = PairAcc1::zero();
    if (u_is_zero_lt8p(r)) {
      const FqU o2 = pair_dpp<PAIR_SWAP>(c2);
      const XYZZU<FqParams> dbl = xyzzu_double_affine(pick2(odd, c2, o2), pick2(odd, o2, c2));
      res = PairAcc1{pick2(odd, dbl.x, dbl.y), pick2(odd, dbl.zz, dbl.zzz)};
    }
    return res;
  }
  acc.a = pick2(odd, x3, y3);
  acc.z = z3;
  return acc;
}

// acc += (+/-)(x2, y2);  c2s = this lane's coordinate of the base (even: x2, odd: y2; canonical memory format), not infinity.
__device__ __forceinline__ PairAcc2 pair_add_mixed(PairAcc2 acc, const Fq2& c2s, bool negate, bool odd) {
  const FqU zero = FqU::zero();
  Fq2U c2 = f2u_from_std(c2s);                               // < p
  {
    const FqU n0 = u_sub<1, 1>(zero, c2.c0), n1 = u_sub<1, 1>(zero, c2.c1);   // p - y, N
    const bool ng = negate && odd;
    c2 = Fq2U{pick2(ng, c2.c0, n0), pick2(ng, c2.c1, n1)};
  }
  if (acc.is_zero()) {                                       // (uniform over the pair)
    const FqU C = UPow2<FqParams, 266>::get();
    acc.a = Fq2U{u_mul(c2.c0, C), u_mul(c2.c1, C)};          // * 2^261, < 2p
    acc.z = Fq2U{C, zero};                                   // (1, 0) * 2^266
    return acc;
  }
  const Fq2U m = f2u_mul<2>(c2, acc.z);                      // even: U2, odd: S2;  < 1.03p
  const Fq2U d1 = Fq2U{u_sub_role<8, 2, 1>(odd, m.c0, acc.a.c0), u_sub_role<8, 2, 1>(odd, m.c1, acc.a.c1)};   // even: P < 10p (X < 6p);  odd: R < 4p (Y < 2p)
  ZK_PAIR_ROUND();
  Fq2U sq;                                                   // (v0 + v1)(v0 - v1) + 2 v0 v1 u
  sq.c0 = u_mul(u_carry(u_add(d1.c0, d1.c1)), u_sub_role<10, 4, 1>(odd, d1.c0, d1.c1));   // even: 20 * 20 c + 1 < 3.37p;  odd: 8 * 8 c + 1 < 1.39p
  sq.c1 = u_mul(u_dbl(d1.c0), d1.c1);                        // even: 200 c + 1 < 2.19p;  odd: 32 c + 1 < 1.2p
  ZK_PAIR_ROUND();
  const Fq2U pp = pair_dpp<PAIR_EVEN>(sq), rr = pair_dpp<PAIR_ODD>(sq);
  const Fq2U p = pair_dpp<PAIR_EVEN>(d1), r = pair_dpp<PAIR_ODD>(d1);
  const Fq2U m3 = f2u_mul<4>(pick2(odd, acc.a, p), pp);      // even: Q = X PP: (6 * 3.37 + 6 * 4) c + 1 < 1.27p, (6 * 2.19 + 6 * 3.37) c + 1 < 1.2p
                                                             // odd: PPP = P PP: (33.7 + 40) c + 1 < 1.44p, (21.9 + 33.7) c + 1 < 1.33p
  ZK_PAIR_ROUND();
  const Fq2U q = pair_dpp<PAIR_EVEN>(m3), ppp = pair_dpp<PAIR_ODD>(m3);
  const Fq2U z3 = f2u_mul<4>(acc.z, pick2(odd, pp, ppp));    // even: ZZ3 = ZZ PP < 1.09p;  odd: ZZZ3 = ZZZ PPP < 1.07p
  ZK_PAIR_ROUND();
  Fq2U x3;
  x3.c0 = u_sub<4, 3>(rr.c0, u_add(ppp.c0, u_dbl(q.c0)));    // PPP + 2Q < 1.44p + 2.53p < 4p, limbs < 3 * 2^29;  X3 < 5.4p
  x3.c1 = u_sub<4, 3>(rr.c1, u_add(ppp.c1, u_dbl(q.c1)));
  const Fq2U d = f2u_sub<8>(q, x3);                          // < 9.3p
  const Fq2U y = pair_dpp<PAIR_ODD>(acc.a);
  const FqU ny0 = u_sub<2, 1>(zero, y.c0), ny1 = u_sub<2, 1>(zero, y.c1), nd1 = u_sub<16, 1>(zero, d.c1);
  // even: R0 D0 + R1 (16p - D1) + (2p - Y0) PPP0 + Y1 PPP1 < 1.65p;   odd: R0 D1 + R1 D0 + (2p - Y0) PPP1 + (2p - Y1) PPP0 < 1.48p
  const FqU t = u_mul4(r.c0, pick2(odd, d.c0, d.c1), r.c1, pick2(odd, nd1, d.c0), ny0, pick2(odd, ppp.c0, ppp.c1), pick2(odd, y.c1, ny1),
                       pick2(odd, ppp.c1, ppp.c0));
  ZK_PAIR_ROUND();
  const FqU t0 = pair_dpp<PAIR_EVEN>(t);
  if (u_is_zero_lt2p(z3.c0) && u_is_zero_lt2p(z3.c1)) {
    // P == 0 (ZZ3 = ZZ P^2 and ZZZ3 = ZZZ P^3 vanish together: uniform over the pair): same x.  Same point -> double
    // (ec.rs:483-485), opposite -> infinity (ec.rs:487).  Rare: the whole doubling in U-form, in both lanes.
    PairAcc2 res = PairAcc2::zero();
    if (u_is_zero_lt8p(r.c0) && u_is_zero_lt8p(r.c1)) {
      const Fq2U o2 = pair_dpp<PAIR_SWAP>(c2);               // the other lane's coordinate (y already carries the sign)
      const XYZZU2 dbl = xyzzu2_double_affine(pick2(odd, c2, o2), pick2(odd, o2, c2));
      res = PairAcc2{pick2(odd, dbl.x, dbl.y), pick2(odd, dbl.zz, dbl.zzz)};
    }
    return res;
  }
  acc.a = Fq2U{pick2(odd, x3.c0, t0), pick2(odd, x3.c1, t)};
  acc.z = z3;
  return acc;
}
// record <-> pair accumulator, coordinate by coordinate (xyzzu_from_r / xyzzu_to_r): ra / oa = this lane's x or y, rz / oz = its zz or zzz
__device__ __forceinline__ PairAcc1 pair_from_r(const Fq& ra, const Fq& rz) {     // (the zero record gives zero limbs)
  return PairAcc1{u_from_std(ra), u_mul(u_from_std(rz), UPow2<FqParams, 266>::get())};
}
__device__ __forceinline__ PairAcc2 pair_from_r(const Fq2& ra, const Fq2& rz) {
  const FqU c266 = UPow2<FqParams, 266>::get();
  return PairAcc2{f2u_from_std(ra), Fq2U{u_mul(u_from_std(rz.c0), c266), u_mul(u_from_std(rz.c1), c266)}};
}
__device__ __forceinline__ void pair_to_r(const PairAcc1& acc, Fq& oa, Fq& oz) {
  oa = Fq::zero();
  oz = Fq::zero();
  if (!acc.is_zero()) {
    oa = u_to_std_lt32p(acc.a);                              // X < 6p / Y < 2p
    oz = u_to_std_lt2p(u_mul(acc.z, UPow2<FqParams, 256>::get()));
  }
}
__device__ __forceinline__ void pair_to_r(const PairAcc2& acc, Fq2& oa, Fq2& oz) {
  oa = Fq2::zero();
  oz = Fq2::zero();
  if (!acc.is_zero()) {
    const FqU c256 = UPow2<FqParams, 256>::get();
    oa = Fq2{u_to_std_lt32p(acc.a.c0), u_to_std_lt32p(acc.a.c1)};
    oz = Fq2{u_to_std_lt2p(u_mul(acc.z.c0, c256)), u_to_std_lt2p(u_mul(acc.z.c1, c256))};
  }
}
__device__ __forceinline__ uint32_t coord_or(const Fq& c) {
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) o |= c.l[k];
  return o;
}
__device__ __forceinline__ uint32_t coord_or(const Fq2& c) { return coord_or(c.c0) | coord_or(c.c1); }
template <class F> struct PairAccOf { using type = PairAcc1; };
template <> struct PairAccOf<Fq2> { using type = PairAcc2; };
#endif  // __HIPCC__

}  // namespace zk
