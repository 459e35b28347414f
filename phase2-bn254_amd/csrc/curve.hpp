// Short-Weierstrass (a = 0) group arithmetic for BN254 G1 (over Fq) and G2 (over Fq2), host+device.
//
// The reference accumulates buckets in Jacobian coordinates (pairing/src/bn256/ec.rs:360-536:
// add-2007-bl / madd-2007-bl / dbl-2009-l).  What a caller observes is the GROUP ELEMENT
// (projective `==` is cross-multiplied, ec.rs:45-85; `into_affine` normalises, ec.rs:596-629), so
// the kernels are free to pick the cheapest complete-enough coordinate system:
//
//   XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0)
//     mixed add  8M + 2S   (madd-2008-s)      vs 7M + 4S Jacobian
//     full add  12M + 2S   (add-2008-s)       vs 11M + 5S Jacobian
//   with explicit handling of the cases the reference's formulas cover implicitly or by branch:
//   acc == infinity, P + P (-> doubling, ec.rs:483-485), P + (-P) (-> infinity, ec.rs:487).
//
// Affine inputs use the reference's raw layout (ec.rs:653-706): x || y Montgomery limbs, the
// all-zero record is the point at infinity.
#pragma once

#include "field.hpp"

namespace zk {

template <class F>
struct Affine {
  F x, y;
  ZK_HD bool is_zero() const { return x.is_zero() && y.is_zero(); }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ZK_HD static XYZZ zero() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZK_HD bool is_zero() const { return zz.is_zero(); }
};

// Jacobian point, same meaning as the reference's projective struct (ec.rs:20-24): z == 0 <=> infinity.
template <class F>
struct Jacobian {
  F x, y, z;
  ZK_HD static Jacobian zero() { return Jacobian{F::zero(), F::one(), F::zero()}; }  // ec.rs:229-235
  ZK_HD bool is_zero() const { return z.is_zero(); }
};

// acc = 2 * (affine p), p != infinity   (mdbl-2008-s-1)
template <class F>
ZK_HD XYZZ<F> xyzz_double_affine(const F& x, const F& y) {
  F u = dbl(y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(x, v);
  F xx = sqr(x);
  F m = add(dbl(xx), xx);
  XYZZ<F> r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// acc = 2 * acc  (dbl-2008-s-1); infinity stays infinity (ZZ3 = V*ZZ1 = 0)
template <class F>
ZK_HD XYZZ<F> xyzz_double(const XYZZ<F>& p) {
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = add(dbl(xx), xx);
  XYZZ<F> r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = mul(v, p.zz);
  r.zzz = mul(w, p.zzz);
  return r;
}

// acc += (x2, y2) affine, (x2, y2) != infinity.   `negate` adds -(x2, y2) instead (signed digits).
template <class F>
ZK_HD void xyzz_add_mixed(XYZZ<F>& acc, const F& x2, const F& y2in, bool negate) {
  F y2 = negate ? neg(y2in) : y2in;
  if (acc.is_zero()) {
    acc.x = x2;
    acc.y = y2;
    acc.zz = F::one();
    acc.zzz = F::one();
    return;
  }
  F u2 = mul(x2, acc.zz);
  F s2 = mul(y2, acc.zzz);
  F p = sub(u2, acc.x);
  F r = sub(s2, acc.y);
  if (p.is_zero()) {
    if (r.is_zero()) acc = xyzz_double_affine(x2, y2);  // same point: double (ec.rs:483-485)
    else acc = XYZZ<F>::zero();                          // opposite points: infinity (ec.rs:487)
    return;
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F q = mul(acc.x, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(q));
  F y3 = sub(mul(r, sub(q, x3)), mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(acc.zz, pp);
  acc.zzz = mul(acc.zzz, ppp);
}

// acc += o  (add-2008-s), complete for all inputs
template <class F>
ZK_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& o) {
  if (o.is_zero()) return;
  if (acc.is_zero()) {
    acc = o;
    return;
  }
  F u1 = mul(acc.x, o.zz);
  F u2 = mul(o.x, acc.zz);
  F s1 = mul(acc.y, o.zzz);
  F s2 = mul(o.y, acc.zzz);
  F p = sub(u2, u1);
  F r = sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) acc = xyzz_double(acc);
    else acc = XYZZ<F>::zero();
    return;
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F q = mul(u1, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(q));
  F y3 = sub(mul(r, sub(q, x3)), mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(mul(acc.zz, o.zz), pp);
  acc.zzz = mul(mul(acc.zzz, o.zzz), ppp);
}

// XYZZ -> Jacobian without an inversion: Z = ZZZ, X' = X*ZZ^2... see DESIGN.md.
//   x = X/ZZ, y = Y/ZZZ and ZZ^3 = ZZZ^2.  Take Z := ZZZ*ZZ^-1?  (needs inverse) -- instead use
//   Z := ZZZ: Z^2 = ZZZ^2 = ZZ^3  => X' = x*Z^2 = X*ZZ^2 ;  Z^3 = ZZZ^3 => Y' = y*Z^3 = Y*ZZZ^2.
template <class F>
ZK_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F>& p) {
  if (p.is_zero()) return Jacobian<F>::zero();
  Jacobian<F> r;
  r.x = mul(p.x, sqr(p.zz));
  r.y = mul(p.y, sqr(p.zzz));
  r.z = p.zzz;
  return r;
}

// XYZZ -> affine (one inversion); infinity -> all-zero record
template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
  if (p.is_zero()) return Affine<F>{F::zero(), F::zero()};
  // 1/ZZ = ZZ^2 * (1/ZZZ)^2 ... simpler: i = 1/(ZZ*ZZZ); 1/ZZ = i*ZZZ; 1/ZZZ = i*ZZ
  F i = inv(mul(p.zz, p.zzz));
  return Affine<F>{mul(p.x, mul(i, p.zzz)), mul(p.y, mul(i, p.zz))};
}

// ---- Jacobian ops for the host-side join of windows (multiexp.rs:146-154) and output format.
// dbl-2009-l (ec.rs:301-358)
template <class F>
ZK_HD void jac_double(Jacobian<F>& p) {
  if (p.is_zero()) return;
  F a = sqr(p.x);
  F b = sqr(p.y);
  F c = sqr(b);
  F d = dbl(sub(sub(sqr(add(p.x, b)), a), c));
  F e = add(dbl(a), a);
  F f = sqr(e);
  F z3 = dbl(mul(p.z, p.y));
  F x3 = sub(sub(f, d), d);
  F y3 = sub(mul(e, sub(d, x3)), dbl(dbl(dbl(c))));
  p.x = x3;
  p.y = y3;
  p.z = z3;
}

// add-2007-bl with the reference's special cases (ec.rs:360-454)
template <class F>
ZK_HD void jac_add(Jacobian<F>& p, const Jacobian<F>& o) {
  if (p.is_zero()) {
    p = o;
    return;
  }
  if (o.is_zero()) return;
  F z1z1 = sqr(p.z);
  F z2z2 = sqr(o.z);
  F u1 = mul(p.x, z2z2);
  F u2 = mul(o.x, z1z1);
  F s1 = mul(mul(p.y, o.z), z2z2);
  F s2 = mul(mul(o.y, p.z), z1z1);
  if (u1 == u2) {
    if (s1 == s2) jac_double(p);
    else p = Jacobian<F>::zero();
    return;
  }
  F h = sub(u2, u1);
  F i = sqr(dbl(h));
  F j = mul(h, i);
  F r = dbl(sub(s2, s1));
  F v = mul(u1, i);
  F x3 = sub(sub(sub(sqr(r), j), v), v);
  F y3 = sub(mul(r, sub(v, x3)), dbl(mul(s1, j)));
  F z3 = mul(sub(sub(sqr(add(p.z, o.z)), z1z1), z2z2), h);
  p.x = x3;
  p.y = y3;
  p.z = z3;
}

template <class F>
ZK_HD Jacobian<F> affine_to_jacobian(const Affine<F>& a) {
  if (a.is_zero()) return Jacobian<F>::zero();
  return Jacobian<F>{a.x, a.y, F::one()};
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;
using G1Jacobian = Jacobian<Fq>;
using G2Jacobian = Jacobian<Fq2>;

}  // namespace zk
