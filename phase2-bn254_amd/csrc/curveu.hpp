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

}  // namespace zk
