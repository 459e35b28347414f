// Scalar multiplication on the device: the per-point `batch_exp` of the ceremony code with its normalisation to affine, the fixed-base
// batch_mul, the window-table build of table mode and the G2 subgroup-membership test (include/mi355zk.h).  Split out of
// api.hip in round 6 (the C ABI wrappers stay there); the interface to the other translation units is api_internal.hpp.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <string>
#include <type_traits>
#include <algorithm>
#include <vector>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "glv.hpp"
#include "device_util.hpp"

#include "api_internal.hpp"

namespace zk {
// ------------------------------------------------------------------------------------------------
// batch scalar multiplication out[i] = k[i or 0] * P[i or 0], affine out (infinity -> all-zero record):
// the per-point `batch_exp` of the ceremony code (powersoftau/src/batched_accumulator.rs:1130-1181: point i
// by its own tau-power; phase2/src/parameters.rs:423-470: every point by the same delta^-1) followed by the
// normalisation to affine that `batch_normalization` performs there (ec.rs:251-299).  The reference uses
// wNAF-4 and one inversion per chunk; for points of the order-r group the group element, hence the affine output, does not depend on
// the chain (G2: the psi split below REQUIRES the subgroup -- glv.hpp, include/mi355zk.h).
//   G1: signed binary (NAF: one addition per three bits instead of two) on the U-form JACOBIAN accumulator of
//       curveu.hpp (a doubling is 1071 mads against 1467 in XYZZ), X and Y parked in the output record and Z in a
//       scratch array, then batch_normalize_kernel: 16 points per lane share one inversion (Montgomery's trick),
//       which is what batch_normalization does with one inversion per CPU chunk.
//   G2: MSB-first double-and-add on the memory-format XYZZ formulas, one inversion per point.
// y^2 == x^3 + 3 (ec.rs:133-148): the G1 kernels below split their scalar over phi(x, y) = (beta x, y), which is multiplication by lambda on
// E(Fq) -- a group of PRIME order r, so on every point of the curve -- and on nothing else: a record that is on no curve (`checked = 0`
// decoding, compute_constrained.rs:16) is handed to the plain-window kernel instead (`defer`), whose doublings and additions are the
// group law of y^2 = x^3 + (y0^2 - x0^3) -- what the reference's wNAF computes for it (wnaf.rs:4-71; no formula names b).
ZK_HD bool g1_on_curve(const Affine<Fq>& p) {
  const Fq one = Fq::one();
  return sqr(p.y) == add(mul(sqr(p.x), p.x), add(add(one, one), one));
}
__device__ __forceinline__ bool g1_defer(const Affine<Fq>& base, uint64_t i, uint32_t* __restrict__ defer_list, uint32_t* __restrict__ defer_count) {
  if (g1_on_curve(base)) return false;
  defer_list[atomicAdd(defer_count, 1u)] = (uint32_t)i;
  return true;
}

template <class F>
__global__ void __launch_bounds__(256) batch_exp_kernel(Affine<F>* __restrict__ out, const Affine<F>* __restrict__ bases, int same_base,
                                                       const uint32_t* __restrict__ scalars, int same_scalar, uint64_t n,
                                                       const uint32_t* __restrict__ base_index, F* __restrict__ zbuf,
                                                       uint32_t* __restrict__ defer_list, uint32_t* __restrict__ defer_count) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  const uint32_t* sp = scalars + (same_scalar ? 0 : i * 8);
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = sp[l];
  const Affine<F> base = bases[same_base ? 0 : (base_index ? base_index[i] : i)];
  if constexpr (std::is_same<F, Fq>::value) {
    JacU<FqParams> acc = JacU<FqParams>::zero();
    if (!base.is_zero()) {
      if (g1_defer(base, i, defer_list, defer_count)) return;
      // GLV (glv.hpp): k P = k1 P + k2 phi(P), |k1|, |k2| < 2^128, phi(x, y) = (beta x, y): 129 doublings instead of 254.  Both
      // halves in non-adjacent form (one addition per three bits each): digit j = bit_{j+1}(3m) - bit_{j+1}(m).
      const GlvSplit g = glv_split(s);
      uint32_t p1[6], n1[6], p2[6], n2[6];
      glv_naf(g.k1, p1, n1);
      glv_naf(g.k2, p2, n2);
      const FqU C = UPow2<FqParams, 266>::get();           // x*2^256 * 2^266 / 2^261 = x * 2^261
      const FqU x2 = u_mul(u_from_std(base.x), C);          // < 2p, N
      const FqU y2 = u_mul(u_from_std(base.y), C);
      const FqU xb = u_mul(x2, u_mul(u_from_std(glv_beta()), C));   // beta x, 2^261 domain, < 2p
      bool found = false;
      for (int bit = 160; bit >= 0; --bit) {   // (canonical scalars end at bit 128; the leading zeros cost nothing: nothing is doubled before the first digit)
        const bool a1 = (p1[bit >> 5] >> (bit & 31)) & 1, m1 = (n1[bit >> 5] >> (bit & 31)) & 1;
        const bool a2 = (p2[bit >> 5] >> (bit & 31)) & 1, m2 = (n2[bit >> 5] >> (bit & 31)) & 1;
        if (found) acc = jacu_double(acc);
        if (a1 | m1) jacu_add_mixed(acc, x2, y2, m1 != g.neg1);
        if (a2 | m2) jacu_add_mixed(acc, xb, y2, m2 != g.neg2);
        found = found | a1 | m1 | a2 | m2;
      }
    }
    const Jacobian<F> r = jacu_to_std(acc);
    out[i] = Affine<F>{r.x, r.y};
    zbuf[i] = r.z;
  } else {
    XYZZ<F> res = XYZZ<F>::zero();
    if (!base.is_zero()) {
      bool found = false;
      for (int bit = 255; bit >= 0; --bit) {
        bool b = (s[bit >> 5] >> (bit & 31)) & 1;
        if (found) res = xyzz_double(res);
        else found = b;
        if (b) xyzz_add_mixed(res, base.x, base.y, false);
      }
    }
    out[i] = xyzz_to_affine(res);
  }
}

// G1, per-point scalars: NAF gives every LANE an addition on a third of the bits, but a WAVE then adds on nearly every
// bit (some lane always has a non-zero digit).  With fixed signed 4-bit windows all lanes add at the same places (64 for a
// 254-bit scalar; 2 x 33 after the GLV split, which halves the doublings):
// each lane builds its own table {1..8} * P (Jacobian + Z^2, Z^3: JacTabU, 192 B) in a scratch array laid out
// [entry][lane], then runs 4 doublings + one table addition per window.  254 x 1071 + 60 x 2079 + table ~ 408k mads per
// scalar against 254 x (1071 + 1593) on the NAF path when lanes diverge.
// SPLIT = false: the plain form for the records the split kernels defer (off the curve): 65 windows over the whole scalar, a doubling that
// lands on Z == 0 (a point of order two: such curves have them) made the literal infinity, infinite table entries skipped.
constexpr int EXP_TAB = 8;
template <bool SPLIT>
__global__ void __launch_bounds__(256) batch_exp_win_kernel(Affine<Fq>* __restrict__ out, const Affine<Fq>* __restrict__ bases, int same_base,
                                                           const uint32_t* __restrict__ scalars, int same_scalar, uint64_t i0, uint64_t n_chunk,
                                                           const uint32_t* __restrict__ base_index, Fq* __restrict__ zbuf,
                                                           JacTabU<FqParams>* __restrict__ tab, const uint32_t* __restrict__ term_list,
                                                           const uint32_t* __restrict__ term_count, uint32_t* __restrict__ defer_list,
                                                           uint32_t* __restrict__ defer_count) {
  // term_list != nullptr: only the listed elements are worked on (lane t of the launch <-> term_list[i0 + t], up to *term_count)
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunk) return;
  uint64_t i = i0 + t;
  if (term_list != nullptr) {
    if (i >= *term_count) return;
    i = term_list[i];
  }
  uint32_t s[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = scalars[(same_scalar ? 0 : i * 8) + l];
  const Affine<Fq> base = bases[same_base ? 0 : (base_index ? base_index[i] : i)];
  JacU<FqParams> acc = JacU<FqParams>::zero();
  if (!base.is_zero()) {
    if constexpr (SPLIT)
      if (g1_defer(base, i, defer_list, defer_count)) return;
    auto canon = [](JacU<FqParams>& q) {                    // plain form: 2 Y Z == 0 is infinity
      if constexpr (!SPLIT)
        if (u_is_zero_lt2p(q.z)) q = JacU<FqParams>::zero();
    };
    const FqU C = UPow2<FqParams, 266>::get();             // x*2^256 * 2^266 / 2^261 = x * 2^261
    const FqU x2 = u_mul(u_from_std(base.x), C);            // < 2p, N
    const FqU y2 = u_mul(u_from_std(base.y), C);
    tab[t] = jacu_tab_entry(JacU<FqParams>{x2, y2, UPow2<FqParams, 261>::get()});
#pragma unroll 1
    for (int e = 2; e <= EXP_TAB; ++e) {                    // e*P = 2 * (e/2)*P  or  (e-1)*P + P
      const JacTabU<FqParams> src = tab[(uint64_t)((e & 1) ? e - 2 : e / 2 - 1) * n_chunk + t];
      JacU<FqParams> q{src.x, src.y, src.z};
      if (e & 1) {
        jacu_add_mixed(q, x2, y2, false);
      } else {
        q = jacu_double(q);
        canon(q);
      }
      tab[(uint64_t)(e - 1) * n_chunk + t] = jacu_tab_entry(q);
    }
    if constexpr (SPLIT) {
      // GLV (glv.hpp): k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^128 -- 33 windows of 4 doublings instead of 64; phi of a table
      // entry is the entry with X multiplied by beta (Y, Z, Z^2, Z^3 unchanged), one more product per addition.
      // signed digits d_j in [-8, 8] of both halves: m = sum d_j 16^j
      const GlvSplit g = glv_split(s);
      uint32_t mag1[5], mag2[5], sgn1[2], sgn2[2];
      signed_nibbles<5, 5>(g.k1, mag1, sgn1);   // (magnitudes < 2^128: the carry out of nibble 31 lands in nibble 32, nothing beyond)
      signed_nibbles<5, 5>(g.k2, mag2, sgn2);
      const FqU betaU = u_mul(u_from_std(glv_beta()), C);     // beta, 2^261 domain
#pragma unroll 1
      for (int j = 39; j >= 0; --j) {   // all 40 nibbles of the five limbs: canonical scalars use 33, and doubling infinity returns at once
#pragma unroll 1
        for (int rep = 0; rep < 4; ++rep) acc = jacu_double(acc);
        const uint32_t d1 = (mag1[j >> 3] >> (4 * (j & 7))) & 15u;
        if (d1) jacu_add_tab(acc, tab[(uint64_t)(d1 - 1) * n_chunk + t], (((sgn1[j >> 5] >> (j & 31)) & 1u) != 0) != g.neg1);
        const uint32_t d2 = (mag2[j >> 3] >> (4 * (j & 7))) & 15u;
        if (d2) {
          JacTabU<FqParams> e = tab[(uint64_t)(d2 - 1) * n_chunk + t];
          e.x = u_mul(e.x, betaU);                            // X < 6p: < 1.08p
          jacu_add_tab(acc, e, (((sgn2[j >> 5] >> (j & 31)) & 1u) != 0) != g.neg2);
        }
      }
    } else {
      uint32_t mag[9], sgn[3];
      signed_nibbles<9, 8>(s, mag, sgn);
#pragma unroll 1
      for (int j = 64; j >= 0; --j) {   // the 64 nibbles and the carry out of the last
#pragma unroll 1
        for (int rep = 0; rep < 4; ++rep) {
          acc = jacu_double(acc);
          canon(acc);
        }
        const uint32_t d = (mag[j >> 3] >> (4 * (j & 7))) & 15u;
        if (d) {
          const JacTabU<FqParams> e = tab[(uint64_t)(d - 1) * n_chunk + t];
          if (!e.z.limbs_all_zero()) {
            jacu_add_tab(acc, e, ((sgn[j >> 5] >> (j & 31)) & 1u) != 0);
            canon(acc);                                       // (the addition doubles when acc == e)
          }
        }
      }
    }
  }
  const Jacobian<Fq> r = jacu_to_std(acc);
  out[i] = Affine<Fq>{r.x, r.y};
  zbuf[i] = r.z;
}

// G1, ONE scalar for every point (phase2 contribute: all of L and H times delta^-1, parameters.rs:423-470): the digit string is the
// same in every lane, so a sliding window costs no divergence.  Both GLV halves in width-5 non-adjacent form (glv_wnaf5) over a
// per-lane table of the eight odd multiples P, 3P .. 15P (JacTabU, [entry][lane] like the windowed kernel's): 127 doublings + ~42
// table additions (2079 mads) + the table (one doubling, one mixed and six table additions) ~ 245k mads per point against the
// ~271k of the plain NAF's 85 mixed additions (1593) -- measured on one box 85.0 -> 91.6 Mpoint/s (contribute on |L| = 2^20: 24.65 ->
// 22.9 ms); 198 VGPRs = two waves per SIMD, and forcing three or four (amdgpu_waves_per_eu, 140 / 336 B of spill) changes nothing: the
// kernel runs at the multiplier's rate.  The digits are made once per call by a one-lane kernel (the scalar lives on the device) and
// read through uniform (scalar) loads.  MI355ZK_EXP_SAME_NAF=1 runs the plain-NAF kernel for the comparison.
struct SameDigits {
  int8_t d1[GLV_WNAF_LEN], d2[GLV_WNAF_LEN];   // digits of |k1|, |k2| with the signs of the split folded in
  int32_t top;                                   // highest index with a non-zero digit in either string, -1: the scalar is zero
};
__global__ void batch_exp_same_digits_kernel(const uint32_t* __restrict__ scalar, SameDigits* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint32_t s[8];
  for (int l = 0; l < 8; ++l) s[l] = scalar[l];
  const GlvSplit g = glv_split(s);
  int8_t a[GLV_WNAF_LEN], b[GLV_WNAF_LEN];
  const int t1 = glv_wnaf5(g.k1, a), t2 = glv_wnaf5(g.k2, b);
  for (int j = 0; j < GLV_WNAF_LEN; ++j) {
    out->d1[j] = g.neg1 ? (int8_t)-a[j] : a[j];
    out->d2[j] = g.neg2 ? (int8_t)-b[j] : b[j];
  }
  out->top = t1 > t2 ? t1 : t2;
}

__global__ void __launch_bounds__(256) batch_exp_same_kernel(Affine<Fq>* __restrict__ out, const Affine<Fq>* __restrict__ bases, int same_base,
                                                            uint64_t i0, uint64_t n_chunk, const uint32_t* __restrict__ base_index,
                                                            Fq* __restrict__ zbuf, JacTabU<FqParams>* __restrict__ tab,
                                                            const SameDigits* __restrict__ dig, uint32_t* __restrict__ defer_list,
                                                            uint32_t* __restrict__ defer_count) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunk) return;
  const uint64_t i = i0 + t;
  const Affine<Fq> base = bases[same_base ? 0 : (base_index ? base_index[i] : i)];
  JacU<FqParams> acc = JacU<FqParams>::zero();
  const int top = dig->top;
  if (!base.is_zero() && top >= 0) {
    if (g1_defer(base, i, defer_list, defer_count)) return;
    const FqU C = UPow2<FqParams, 266>::get();             // x*2^256 * 2^266 / 2^261 = x * 2^261
    const FqU x2 = u_mul(u_from_std(base.x), C);            // < 2p, N
    const FqU y2 = u_mul(u_from_std(base.y), C);
    {
      JacU<FqParams> q{x2, y2, UPow2<FqParams, 261>::get()};
      tab[t] = jacu_tab_entry(q);                           // P
      q = jacu_double(q);
      const JacTabU<FqParams> twice = jacu_tab_entry(q);    // 2P, added six times
      jacu_add_mixed(q, x2, y2, false);                     // 3P
      tab[n_chunk + t] = jacu_tab_entry(q);
#pragma unroll 1
      for (int e = 2; e < EXP_TAB; ++e) {                   // 5P .. 15P   (a point of the prime-order group: no sum here is the identity)
        jacu_add_tab(q, twice, false);
        tab[(uint64_t)e * n_chunk + t] = jacu_tab_entry(q);
      }
    }
    const FqU betaU = u_mul(u_from_std(glv_beta()), C);     // beta, 2^261 domain
#pragma unroll 1
    for (int j = top; j >= 0; --j) {
      acc = jacu_double(acc);                               // (infinity returns at once)
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {                // ONE inlined jacu_add_tab for both halves
        const int d = half ? dig->d2[j] : dig->d1[j];
        if (d == 0) continue;
        const int mag = d < 0 ? -d : d;
        JacTabU<FqParams> e = tab[(uint64_t)(mag >> 1) * n_chunk + t];
        if (half) e.x = u_mul(e.x, betaU);                  // phi of the entry: X * beta (X < 6p: < 1.08p)
        jacu_add_tab(acc, e, d < 0);
      }
    }
  }
  const Jacobian<Fq> r = jacu_to_std(acc);
  out[i] = Affine<Fq>{r.x, r.y};
  zbuf[i] = r.z;
}

// G2: the same fixed signed 4-bit windows on the U-form Fq2 Jacobian accumulator of curveu.hpp (JacU2: 29-bit lazy limbs, one
// v_mad_u64_u32 per partial product, shared Montgomery reductions) -- round 1 ran this on memory-format Fq2 at 9 Mpoint/s.
// Table build and main loop run through ONE loop with a single inlined jacu2_double and a single inlined jacu2_add_tab: the Fq2
// group law is > 100 KB of code per copy.
//   step = (load entry, double?, add entry, store entry); entries 1..8 hold 1P..8P (with Z^2, Z^3), 0 = none.
__device__ __forceinline__ JacTabU2 tabu2_load(const JacTabU2* p) {
  JacTabU2 r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(JacTabU2) / 16); ++i) d[i] = q[i];
  return r;
}
__device__ __forceinline__ void tabu2_store(JacTabU2* p, const JacTabU2& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(JacTabU2) / 16); ++i) d[i] = s[i];
}

// SPLIT = true: the scalar goes over the twist's endomorphism psi (glv.hpp: k P = k1 P + k2 psi(P), k1, k2 < 2^128 -- 33 windows of four
// doublings instead of 64).  psi(P) = mu P holds in the order-r subgroup ONLY, so this form runs only under the caller's promise
// MI355ZK_G2_TRUSTED_SUBGROUP.  SPLIT = false (the default): 65 plain windows over the whole 256-bit scalar -- the group law and nothing
// else, hence the reference's wNAF answer (pairing/src/wnaf.rs:4-71) for EVERY record its decoders admit (ec.rs:1136-1344 test the curve
// equation at most): points of the twist outside the subgroup, and -- none of the formulas uses the curve's b -- records that are on no
// curve at all (`checked = 0` decoding), whose multiples live on y^2 = x^3 + (y0^2 - x0^3) where small orders exist: a doubling that
// lands on Z == 0 (a point of order two) is made the literal infinity, and an infinite table entry is skipped.
template <bool SPLIT>
__global__ void __launch_bounds__(256) batch_exp_win_u2_kernel(Affine<Fq2>* __restrict__ out, const Affine<Fq2>* __restrict__ bases, int same_base,
                                                              const uint32_t* __restrict__ scalars, int same_scalar, uint64_t i0,
                                                              uint64_t n_chunk, const uint32_t* __restrict__ base_index,
                                                              Fq2* __restrict__ zbuf, JacTabU2* __restrict__ tab,
                                                              const uint32_t* __restrict__ term_list, const uint32_t* __restrict__ term_count) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunk) return;
  uint64_t i = i0 + t;
  if (term_list != nullptr) {
    if (i >= *term_count) return;
    i = term_list[i];
  }
  uint32_t s[8];
  const uint32_t* sp = scalars + (same_scalar ? 0 : i * 8);
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = sp[l];
  const Affine<Fq2> base = bases[same_base ? 0 : (base_index ? base_index[i] : i)];
  JacU2 acc = JacU2::zero();
  if (!base.is_zero()) {
    tabu2_store(tab + t, jacu2_tab_from_affine(base.x, base.y));
    // signed digits d_j in [-8, 8]: m = sum d_j 16^j.  SPLIT: of both halves (five words each); plain: of the scalar (eight words and the carry)
    constexpr int NW = SPLIT ? 5 : 9;
    uint32_t mag1[NW], mag2[SPLIT ? 5 : 1], sgn1[(NW + 3) / 4], sgn2[2];
    Fq2U cxU, cyU;
    if constexpr (SPLIT) {
      const Glv2Split g = glv2_split(s);
      signed_nibbles<5, 5>(g.k1, mag1, sgn1);
      signed_nibbles<5, 5>(g.k2, mag2, sgn2);
      const FqU C266 = UPow2<FqParams, 266>::get();
      const Fq2 cxs = glv2_cx(), cys = glv2_cy();
      cxU = Fq2U{u_mul(u_from_std(cxs.c0), C266), u_mul(u_from_std(cxs.c1), C266)};   // 2^261 domain, < 2p
      cyU = Fq2U{u_mul(u_from_std(cys.c0), C266), u_mul(u_from_std(cys.c1), C266)};
    } else {
      signed_nibbles<9, 8>(s, mag1, sgn1);
    }
    // table program, one nibble per field (load, double, add, store):  2P = 2*1P, 3P = 2P + 1P, 4P = 2*2P, 5P = 4P + 1P, ...
    constexpr uint32_t PROG[7] = {0x1102, 0x0013, 0x2104, 0x0015, 0x3106, 0x0017, 0x4108};
    // SPLIT: all nibbles of the five limbs (canonical scalars use 33); plain: the 64 nibbles and the carry out of the last (doubling infinity returns at once)
    constexpr int WINDOWS = SPLIT ? 40 : 65, PER = SPLIT ? 5 : 4;
#pragma unroll 1
    for (int step = 0; step < 7 + PER * WINDOWS; ++step) {
      uint32_t load = 0, dbl_it = 0, add = 0, store = 0, negate = 0, psi = 0;
      if (step < 7) {
        const uint32_t pr = PROG[step];
        load = pr >> 12;
        dbl_it = (pr >> 8) & 15u;
        add = (pr >> 4) & 15u;
        store = pr & 15u;
      } else {
        const int m = step - 7;        // per window: four doublings (the fourth adds the k1 digit), then (SPLIT) the k2 digit through psi
        if (m == 0) acc = JacU2::zero();
        const int win = m / PER, sub = m - PER * win, j = WINDOWS - 1 - win;
        if (sub < 4) {
          dbl_it = 1;
          if (sub == 3) {
            add = (mag1[j >> 3] >> (4 * (j & 7))) & 15u;
            negate = (sgn1[j >> 5] >> (j & 31)) & 1u;
          }
        } else {
          add = (mag2[j >> 3] >> (4 * (j & 7))) & 15u;
          negate = (sgn2[j >> 5] >> (j & 31)) & 1u;
          psi = 1;
        }
      }
      if (load) {
        const JacTabU2 e = tabu2_load(tab + (uint64_t)(load - 1) * n_chunk + t);
        acc = JacU2{e.x, e.y, e.z};
      }
      if (dbl_it) {
        acc = jacu2_double(acc);
        if constexpr (!SPLIT)
          if (u_is_zero_lt2p(acc.z.c0) && u_is_zero_lt2p(acc.z.c1)) acc = JacU2::zero();   // 2 Y Z == 0: Y == 0, a point of order two
      }
      if (add) {
        JacTabU2 e = tabu2_load(tab + (uint64_t)(add - 1) * n_chunk + t);
        if constexpr (SPLIT) {
          if (psi) e = jacu2_tab_psi(e, cxU, cyU);
          jacu2_add_tab(acc, e, negate != 0);
        } else {
          if (!e.z.limbs_all_zero()) {                        // (d P == infinity for a small d: only off the twist)
            jacu2_add_tab(acc, e, negate != 0);
            if (u_is_zero_lt2p(acc.z.c0) && u_is_zero_lt2p(acc.z.c1)) acc = JacU2::zero();   // (the addition doubles when acc == e)
          }
        }
      }
      if (store) tabu2_store(tab + (uint64_t)(store - 1) * n_chunk + t, jacu2_tab_entry(acc));
    }
  }
  const Jacobian<Fq2> r = jacu2_to_std(acc);
  out[i] = Affine<Fq2>{r.x, r.y};
  zbuf[i] = r.z;
}

// io[i] = (X, Y) of a Jacobian point whose Z is z[i]  ->  the affine record (X / Z^2, Y / Z^3); Z == 0 -> all-zero record.
// K consecutive points per lane share ONE inversion (prefix products, ec.rs:251-299's scheme).
template <class F, int K>
__global__ void __launch_bounds__(256) batch_normalize_kernel(Affine<F>* __restrict__ io, const F* __restrict__ z, uint64_t n) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * K;
  if (i0 >= n) return;
  F pre[K];
  F run = F::one();
  // for_limbs: the index is a compile-time constant, which keeps pre[] in registers (a "#pragma unroll" over these bodies is refused)
  for_limbs<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    pre[k] = run;
    if (i0 + k < n) {
      const F zk = z[i0 + k];
      if (!zk.is_zero()) run = mul(run, zk);
    }
  });
  F inv_run = inv(run);
  for_limbs<K>([&](auto kc) {
    constexpr int k = K - 1 - decltype(kc)::value;
    if (i0 + k < n) {
      const F zk = z[i0 + k];
      Affine<F> p{F::zero(), F::zero()};
      if (!zk.is_zero()) {
        const F zi = mul(inv_run, pre[k]);
        inv_run = mul(inv_run, zk);
        const F zi2 = sqr(zi);
        const Affine<F> xy = io[i0 + k];
        p.x = mul(xy.x, zi2);
        p.y = mul(xy.y, mul(zi2, zi));
      }
      io[i0 + k] = p;
    }
  });
}

int batch_normalize_g1(void* d_io_affine, const void* d_z, uint64_t n, hipStream_t st) {
  if (n == 0) return ZK_OK;
  constexpr int K = 16;
  const uint64_t lanes = (n + K - 1) / K;
  hipLaunchKernelGGL((batch_normalize_kernel<Fq, K>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_io_affine,
                     (const Fq*)d_z, n);
  ZK_HIP(hipGetLastError());
  return ZK_OK;
}

int batch_normalize_g2(void* d_io_affine, const void* d_z, uint64_t n, hipStream_t st) {
  if (n == 0) return ZK_OK;
  constexpr int K = 8;
  const uint64_t lanes = (n + K - 1) / K;
  hipLaunchKernelGGL((batch_normalize_kernel<Fq2, K>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, (Affine<Fq2>*)d_io_affine,
                     (const Fq2*)d_z, n);
  ZK_HIP(hipGetLastError());
  return ZK_OK;
}

// per (device, stream) scratch for the Z coordinates between the two kernels (grow-only; freed at shutdown)
struct ExpScratch {
  void* p = nullptr;
  size_t bytes = 0;
};
static std::mutex g_exp_mu;
static std::map<std::pair<int, void*>, ExpScratch> g_exp_scratch;

static int exp_scratch(size_t bytes, void* stream, void** out) {
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_exp_mu);
  // (one buffer per stream a caller has ever used: bounded.  Past 16 streams on this device everything is dropped once the device
  // is idle -- the callers hold g_exp_launch_mu, so no other scalar-multiplication kernels are being enqueued meanwhile.)
  if (g_exp_scratch.find(std::make_pair(dev, stream)) == g_exp_scratch.end()) {
    size_t mine = 0;
    for (auto& kv : g_exp_scratch) mine += kv.first.first == dev ? 1 : 0;
    if (mine >= 16) {
      ZK_HIP(hipDeviceSynchronize());
      for (auto it = g_exp_scratch.begin(); it != g_exp_scratch.end();) {
        if (it->first.first != dev) { ++it; continue; }
        (void)hipFree(it->second.p);
        it = g_exp_scratch.erase(it);
      }
    }
  }
  ExpScratch& sb = g_exp_scratch[std::make_pair(dev, stream)];
  if (sb.bytes < bytes) {
    if (sb.p) {
      ZK_HIP(hipStreamSynchronize((hipStream_t)stream));  // earlier launches on this stream may still use the old buffer
      ZK_HIP(hipFree(sb.p));
    }
    sb.p = nullptr;
    sb.bytes = 0;
    ZK_HIP(hipMalloc(&sb.p, bytes));
    sb.bytes = bytes;
  }
  *out = sb.p;
  return ZK_OK;
}
void exp_scratch_release_all() {
  std::lock_guard<std::mutex> lk(g_exp_mu);
  for (auto& kv : g_exp_scratch) {
    (void)hipSetDevice(kv.first.first);
    (void)hipFree(kv.second.p);
  }
  g_exp_scratch.clear();
}

// QAP coefficients are mostly +-1 (circom R1CS): a term with coefficient 1 / r - 1 / 0 is the base itself / its negative (where the
// base is known to have order r: see allow_minus_one) / nothing, no scalar multiplication.  Those terms are written directly (Z = one resp. 0 for the normalisation pass that follows);
// the indices of the others are appended to `list` and only they run the windowed multiplication, as full waves.
template <class F>
__global__ void __launch_bounds__(256) exp_classify_kernel(Affine<F>* __restrict__ out, F* __restrict__ zbuf, const Affine<F>* __restrict__ bases,
                                                          const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ base_index, uint64_t n,
                                                          uint32_t* __restrict__ list, uint32_t* __restrict__ count, int order_r,
                                                          const uint8_t* __restrict__ member, uint32_t* __restrict__ list_out,
                                                          uint32_t* __restrict__ count_out) {
  // order_r: 1 = every base has order r (G2 under the caller's promise; G1, where a record ON the curve has), 0 = none is known to,
  // 2 = member[b] says so per base (the G2 membership test was run over the bases).  A term whose base is not known to have order r
  // goes to list_out (the plain-window kernel) when that list is given, and gets no r - 1 shortcut.
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t bi = base_index ? base_index[i] : i;
  const bool in_group = order_r == 1 || (order_r == 2 && member[bi] != 0);
  uint32_t s[8];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  const uint4 s0 = sp[0], s1 = sp[1];
  s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
  bool hi_zero = true, is_rm1 = true;
#pragma unroll
  for (int l = 1; l < 8; ++l) {
    hi_zero = hi_zero && s[l] == 0;
    is_rm1 = is_rm1 && s[l] == FrParams::P[l];
  }
  is_rm1 = is_rm1 && s[0] == FrParams::P[0] - 1u;
  const bool is_zero = hi_zero && s[0] == 0, is_one = hi_zero && s[0] == 1;
  // (r - 1) P == -P needs r P == infinity: true in the order-r group only
  if (!in_group) is_rm1 = false;
  if (!(is_zero || is_one || is_rm1)) {
    if (in_group || list_out == nullptr) list[atomicAdd(count, 1u)] = (uint32_t)i;
    else list_out[atomicAdd(count_out, 1u)] = (uint32_t)i;
    return;
  }
  Affine<F> p = bases[bi];
  if constexpr (std::is_same<F, Fq>::value)
    if (is_rm1 && !p.is_zero() && !g1_on_curve(p)) {
      list[atomicAdd(count, 1u)] = (uint32_t)i;
      return;
    }
  if (is_zero || p.is_zero()) {
    p = Affine<F>{F::zero(), F::zero()};
    zbuf[i] = F::zero();
  } else {
    if (is_rm1) p.y = neg(p.y);
    zbuf[i] = F::one();
  }
  out[i] = p;
}

// The scratch (Z coordinates, window tables) is per (device, stream) and the two kernels of one call must reach the stream
// back to back: several host threads may share a stream (the default one above all), and A.exp, B.exp, A.normalize would
// let A normalise with B's Z.  Held while ENQUEUEING only; the stream orders the kernels.
static std::mutex g_exp_launch_mu;

template <class F>
int batch_exp(void* d_out, const void* d_bases, int same_base, const void* d_scalars, int same_scalar, size_t n, void* stream,
              const uint32_t* d_base_index, bool shortcut_unit_scalars, bool g2_trusted, const uint8_t* d_g2_member) {
  // g2_trusted (G2 only): the caller's promise that every base lies in the order-r subgroup -- the psi-split kernel; otherwise the plain one,
  // or (d_g2_member: one byte per base from g2_subgroup_flags, with shortcut_unit_scalars) each term by its base's membership
  if (!d_out || !d_bases || !d_scalars) return n ? ZK_ERR_BAD_ARGS : ZK_OK;
  if (n == 0) return ZK_OK;
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> launch_lk(g_exp_launch_mu);
  if constexpr (std::is_same<F, Fq>::value) {
    const bool windowed = !same_scalar;                     // per-point scalars: fixed windows (see batch_exp_win_kernel)
    const size_t chunk = n < ((size_t)1 << 18) ? n : ((size_t)1 << 18);
    const size_t z_bytes = (n * sizeof(Fq) + 255) & ~(size_t)255;
    const bool shortcut = shortcut_unit_scalars && windowed && !same_base;
    const size_t list_bytes = shortcut ? ((n + 1) * 4 + 255) & ~(size_t)255 : 0;
    const size_t defer_bytes = ((n + 1) * 4 + 255) & ~(size_t)255;   // [0] = count, then the records that are on no curve (g1_defer)
    void* p = nullptr;
    static const bool same_naf = std::getenv("MI355ZK_EXP_SAME_NAF") != nullptr;   // (the plain-NAF kernel of rounds 2-3, for the comparison)
    const bool same_win = !windowed && !same_naf;          // one scalar for all points: the sliding-window kernel
    const size_t tab_bytes = (size_t)EXP_TAB * chunk * sizeof(JacTabU<FqParams>);
    int rc = exp_scratch(z_bytes + list_bytes + defer_bytes + tab_bytes + (same_win ? 512 : 0), stream, &p);
    if (rc) return rc;
    Fq* zbuf = (Fq*)p;
    uint32_t* list = shortcut ? (uint32_t*)((char*)p + z_bytes) : nullptr;   // [0] = count, then the general terms
    uint32_t* defer = (uint32_t*)((char*)p + z_bytes + list_bytes);
    JacTabU<FqParams>* tab = (JacTabU<FqParams>*)((char*)p + z_bytes + list_bytes + defer_bytes);
    ZK_HIP(hipMemsetAsync(defer, 0, 4, st));
    if (shortcut) {
      ZK_HIP(hipMemsetAsync(list, 0, 4, st));
      hipLaunchKernelGGL(exp_classify_kernel<Fq>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_out, zbuf, (const Affine<Fq>*)d_bases,
                         (const uint32_t*)d_scalars, d_base_index, (uint64_t)n, list + 1, list, /*order_r=*/1, (const uint8_t*)nullptr,
                         (uint32_t*)nullptr, (uint32_t*)nullptr);
    }
    if (windowed) {
      for (size_t i0 = 0; i0 < n; i0 += chunk) {
        const size_t m = n - i0 < chunk ? n - i0 : chunk;
        hipLaunchKernelGGL(batch_exp_win_kernel<true>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_out, (const Affine<Fq>*)d_bases,
                           same_base, (const uint32_t*)d_scalars, 0, (uint64_t)i0, (uint64_t)m, d_base_index, zbuf, tab,
                           shortcut ? list + 1 : (const uint32_t*)nullptr, shortcut ? list : (const uint32_t*)nullptr, defer + 1, defer);
      }
    } else if (same_win) {
      SameDigits* dig = (SameDigits*)((char*)p + z_bytes + list_bytes + defer_bytes + tab_bytes);
      static_assert(sizeof(SameDigits) <= 512, "digit buffer");
      hipLaunchKernelGGL(batch_exp_same_digits_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)d_scalars, dig);
      for (size_t i0 = 0; i0 < n; i0 += chunk) {
        const size_t m = n - i0 < chunk ? n - i0 : chunk;
        hipLaunchKernelGGL(batch_exp_same_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_out, (const Affine<Fq>*)d_bases,
                           same_base, (uint64_t)i0, (uint64_t)m, d_base_index, zbuf, tab, (const SameDigits*)dig, defer + 1, defer);
      }
    } else {
      hipLaunchKernelGGL(batch_exp_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Affine<F>*)d_out, (const Affine<F>*)d_bases,
                         same_base, (const uint32_t*)d_scalars, same_scalar, (uint64_t)n, d_base_index, zbuf, defer + 1, defer);
    }
    // the deferred records (none on honest data: every lane of these launches reads the count and leaves) through the plain windows
    for (size_t i0 = 0; i0 < n; i0 += chunk) {
      const size_t m = n - i0 < chunk ? n - i0 : chunk;
      hipLaunchKernelGGL(batch_exp_win_kernel<false>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_out, (const Affine<Fq>*)d_bases,
                         same_base, (const uint32_t*)d_scalars, same_scalar, (uint64_t)i0, (uint64_t)m, d_base_index, zbuf, tab,
                         (const uint32_t*)(defer + 1), (const uint32_t*)defer, (uint32_t*)nullptr, (uint32_t*)nullptr);
    }
    ZK_HIP(hipGetLastError());
    constexpr int K = 16;
    const uint64_t lanes = (n + K - 1) / K;
    hipLaunchKernelGGL((batch_normalize_kernel<Fq, K>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, (Affine<Fq>*)d_out, (const Fq*)zbuf,
                       (uint64_t)n);
    ZK_HIP(hipGetLastError());
  } else {
    const size_t chunk = n < ((size_t)1 << 18) ? n : ((size_t)1 << 18);
    const size_t z_bytes = (n * sizeof(F) + 255) & ~(size_t)255;
    const bool shortcut = shortcut_unit_scalars && !same_scalar && !same_base;
    const bool by_member = shortcut && !g2_trusted && d_g2_member != nullptr;
    const size_t list_bytes = shortcut ? ((n + 1) * 4 + 255) & ~(size_t)255 : 0;
    void* p = nullptr;
    int rc = exp_scratch(z_bytes + 2 * list_bytes + (size_t)EXP_TAB * chunk * sizeof(JacTabU2), stream, &p);
    if (rc) return rc;
    F* zbuf = (F*)p;
    uint32_t* list = shortcut ? (uint32_t*)((char*)p + z_bytes) : nullptr;                  // terms for the split kernel (or: all general terms)
    uint32_t* list_out = shortcut ? (uint32_t*)((char*)p + z_bytes + list_bytes) : nullptr;  // by_member: terms whose base is outside the subgroup
    if (shortcut) {
      ZK_HIP(hipMemsetAsync(list, 0, 4, st));
      ZK_HIP(hipMemsetAsync(list_out, 0, 4, st));
      hipLaunchKernelGGL(exp_classify_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Affine<F>*)d_out, zbuf, (const Affine<F>*)d_bases,
                         (const uint32_t*)d_scalars, d_base_index, (uint64_t)n, list + 1, list, /*order_r=*/g2_trusted ? 1 : (by_member ? 2 : 0), d_g2_member,
                         by_member ? list_out + 1 : (uint32_t*)nullptr, by_member ? list_out : (uint32_t*)nullptr);
    }
    JacTabU2* tab = (JacTabU2*)((char*)p + z_bytes + 2 * list_bytes);
    for (int pass = 0; pass < (by_member ? 2 : 1); ++pass) {
      const bool split = g2_trusted || (by_member && pass == 0);
      const uint32_t* tl = !shortcut ? nullptr : (by_member && pass == 1 ? list_out : list);
      for (size_t i0 = 0; i0 < n; i0 += chunk) {
        const size_t m = n - i0 < chunk ? n - i0 : chunk;
        if (split)
          hipLaunchKernelGGL(batch_exp_win_u2_kernel<true>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (Affine<Fq2>*)d_out,
                             (const Affine<Fq2>*)d_bases, same_base, (const uint32_t*)d_scalars, same_scalar, (uint64_t)i0, (uint64_t)m, d_base_index,
                             (Fq2*)zbuf, tab, tl ? tl + 1 : (const uint32_t*)nullptr, tl);
        else
          hipLaunchKernelGGL(batch_exp_win_u2_kernel<false>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (Affine<Fq2>*)d_out,
                             (const Affine<Fq2>*)d_bases, same_base, (const uint32_t*)d_scalars, same_scalar, (uint64_t)i0, (uint64_t)m, d_base_index,
                             (Fq2*)zbuf, tab, tl ? tl + 1 : (const uint32_t*)nullptr, tl);
      }
    }
    ZK_HIP(hipGetLastError());
    constexpr int K = 8;
    const uint64_t lanes = (n + K - 1) / K;
    hipLaunchKernelGGL((batch_normalize_kernel<F, K>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, (Affine<F>*)d_out, (const F*)zbuf,
                       (uint64_t)n);
    ZK_HIP(hipGetLastError());
  }
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// G2 subgroup membership.  The psi-split kernels are exact in the order-r subgroup ONLY (glv.hpp); neither the reference's decoders nor
// ours test membership (ec.rs:1136-1344 check the curve equation), so the default paths above either avoid the split or run THIS test
// first.  For a point of the twist, with x the BN parameter (63 bits) and psi the twist's Frobenius endomorphism:
//     P in G2   <=>   [x + 1] P + psi([x] P) + psi^2([x] P) == psi^3([2 x] P)
// "=>": psi acts on G2 as q, and (x + 1) + x q + x q^2 - 2 x q^3 == 0 mod r (a short vector of the BN lattice).  "<=": psi satisfies
// chi(X) = X^2 - t X + q on ALL of E'(Fq2), so a point killed by f(psi), f = (x + 1) + x X + x X^2 - 2 x X^3, is killed by the integer
// Res(f, chi); its order divides gcd(Res(f, chi), #E'(Fq2)) = gcd(Res, r (2 q - r)), and for BN254 that gcd is r exactly (computed:
// r | Res, gcd(Res, 2 q - r) = 1 -- tests/test_g2_subgroup.py recomputes it).  Rounds 3-4 tested psi(P) == [6 x^2] P (127 doublings + 68
// additions, sound by the same argument); this form is ONE multiplication by x in non-adjacent form -- 62 doublings + 23 additions -- plus
// three psi, four additions and a doubling: 2^20 points in ~17 ms against 40.  A record that is not on the twist is not a member.
static int mul_slot(void* stream, void** out);
ZK_HD bool g2_in_subgroup(const Affine<Fq2>& p) {
  if (p.is_zero()) return true;  // the identity
  if (sqr(p.y) != add(mul(sqr(p.x), p.x), g2_coeff_b())) return false;
  const FqU C266 = UPow2<FqParams, 266>::get();
  const Fq2 cxs = glv2_cx(), cys = glv2_cy();
  const Fq2U cxU{u_mul(u_from_std(cxs.c0), C266), u_mul(u_from_std(cxs.c1), C266)};
  const Fq2U cyU{u_mul(u_from_std(cys.c0), C266), u_mul(u_from_std(cys.c1), C266)};
  const JacTabU2 e = jacu2_tab_from_affine(p.x, p.y);
  // x = 0x44e992b44a6909f1 = POS - NEG (non-adjacent form, 24 digits, top bit 62)
  const uint64_t POS = 0x450a14044a890a01ull, NEG = 0x0020815000200010ull;
  JacU2 a = JacU2::zero();
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    a = jacu2_double(a);
    const bool pos = (POS >> bit) & 1ull, neg = (NEG >> bit) & 1ull;
    if (pos | neg) jacu2_add_tab(a, e, neg);
  }
  if (a.is_zero()) return false;  // [x] P == infinity for P != infinity: the order divides x, not r
  const JacTabU2 b1 = jacu2_tab_psi(jacu2_tab_entry(a), cxU, cyU);   // psi([x] P)
  const JacTabU2 b2 = jacu2_tab_psi(b1, cxU, cyU);                    // psi^2([x] P)
  const JacTabU2 b3 = jacu2_tab_psi(b2, cxU, cyU);                    // psi^3([x] P)
  JacU2 d = jacu2_double(JacU2{b3.x, b3.y, b3.z});                    // psi^3([2 x] P)
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {                                       // a := [x] P + P + psi + psi^2, then d -= a   (ONE inlined addition)
    if (k < 3) {
      jacu2_add_tab(a, k == 0 ? e : (k == 1 ? b1 : b2), false);
    } else {
      if (a.is_zero()) break;
      jacu2_add_tab(d, jacu2_tab_entry(a), true);
    }
  }
  return d.is_zero();
}

__global__ void __launch_bounds__(256) g2_subgroup_check_kernel(const Affine<Fq2>* __restrict__ pts, uint64_t n, unsigned long long* __restrict__ bad,
                                                               uint8_t* __restrict__ member) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool in = g2_in_subgroup(pts[i]);
  if (member) member[i] = in ? 1 : 0;
  if (!in && bad) atomicMin(bad, (unsigned long long)i);
}
// member[i] = 1 iff record i is in the order-r subgroup (asynchronous on `stream`): the G2 sparse product multiplies a member's terms
// through the psi split and everything else through the plain windows
int g2_subgroup_flags(const void* d_points, size_t n, void* stream, uint8_t* d_member) {
  if (n == 0) return ZK_OK;
  hipLaunchKernelGGL(g2_subgroup_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const Affine<Fq2>*)d_points, (uint64_t)n,
                     (unsigned long long*)nullptr, d_member);
  ZK_HIP(hipGetLastError());
  return ZK_OK;
}

int g2_subgroup_check(const void* d_points, size_t n, void* stream, long long* bad_index) {
  if (!bad_index || (!d_points && n)) return ZK_ERR_BAD_ARGS;
  *bad_index = -1;
  if (n == 0) return ZK_OK;
  hipStream_t st = (hipStream_t)stream;
  void* d_bad = nullptr;
  int rc = mul_slot(stream, &d_bad);  // (a 256-byte device slot from the per-stream ring below)
  if (rc) return rc;
  ZK_HIP(hipMemsetAsync(d_bad, 0xff, 8, st));
  hipLaunchKernelGGL(g2_subgroup_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const Affine<Fq2>*)d_points, (uint64_t)n,
                     (unsigned long long*)d_bad, (uint8_t*)nullptr);
  ZK_HIP(hipGetLastError());
  unsigned long long h = 0;
  ZK_HIP(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, st));
  ZK_HIP(hipStreamSynchronize(st));
  if (h != ~0ull) *bad_index = (long long)h;
  return ZK_OK;
}

// fixed base given by value on the host (input synthesis: P_i = k_i * G).  The 64 / 128-byte device copy of the base comes from a
// per-(device, stream) ring of slots allocated once: hipMalloc / hipFree per call synchronise the whole device, and this entry is
// the building block of per-point batch_exp synthesis (256 calls per bench input).  A slot is in flight only until its call's
// closing stream synchronisation; MUL_SLOTS concurrent calls on ONE stream is more than any caller here issues.
constexpr int MUL_SLOTS = 16;
struct MulSlots {
  void* p = nullptr;
  unsigned next = 0;
};
static std::mutex g_mul_mu;
static std::map<std::pair<int, void*>, MulSlots> g_mul_slots;
static int mul_slot(void* stream, void** out) {
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mul_mu);
  MulSlots& m = g_mul_slots[std::make_pair(dev, stream)];
  if (m.p == nullptr) ZK_HIP(hipMalloc(&m.p, (size_t)MUL_SLOTS * 256));
  *out = (char*)m.p + (size_t)(m.next++ % MUL_SLOTS) * 256;
  return ZK_OK;
}
void mul_slots_release_all() {
  std::lock_guard<std::mutex> lk(g_mul_mu);
  for (auto& kv : g_mul_slots) {
    (void)hipSetDevice(kv.first.first);
    (void)hipFree(kv.second.p);
  }
  g_mul_slots.clear();
}
template <class F>
int batch_mul(void* d_out, const uint64_t* base_raw, const void* d_scalars, size_t n, void* stream) {
  static_assert(sizeof(Affine<F>) <= 256, "slot size");
  if (!d_out || !base_raw || (!d_scalars && n)) return ZK_ERR_BAD_ARGS;
  if (n == 0) return ZK_OK;
  void* d_base = nullptr;
  int rc = mul_slot(stream, &d_base);
  if (rc) return rc;
  ZK_HIP(hipMemcpyAsync(d_base, base_raw, sizeof(Affine<F>), hipMemcpyHostToDevice, (hipStream_t)stream));
  // G2: ONE base, so its membership in the order-r subgroup is decided here, on the host (psi(P) == mu P, ~200 group operations),
  // and the psi-split kernel runs only for a member; any other record of the twist goes through the plain windows.
  bool member = false;
  if constexpr (std::is_same<F, Fq2>::value) {
    Affine<Fq2> b;
    std::memcpy(&b, base_raw, sizeof b);
    member = g2_in_subgroup(b);
  }
  rc = batch_exp<F>(d_out, d_base, 1, d_scalars, 0, n, stream, nullptr, false, member);
  if (rc != ZK_OK) return rc;
  ZK_HIP(hipStreamSynchronize((hipStream_t)stream));  // base_raw is the caller's (pageable) memory; the result is ready on return
  return ZK_OK;
}

// Window table of a base vector for table-mode multiexps (msm_impl.hpp: msm_device with table_stride != 0):
//   table[w * n + i] = 2^(width[0] + .. + width[w-1]) * bases[i],  w < W,  affine records (the identity stays the identity).
// Window w + 1 is window w doubled width[w] times: PLAIN doublings on the U-form Jacobian accumulator (X, Y parked in the output
// plane, Z in scratch), then one batched normalisation (one inversion per 16 / 8 points).  Exact for EVERY point the decoders admit:
// a doubling is the group law itself, whereas the shared-scalar batch_exp this used to call splits 2^k over psi, which is a
// multiplication by mu on the order-r subgroup of the twist only -- a G2 record with a cofactor component (nothing in the reference
// or here tests membership) got a table that disagreed with the plain bucket call and the reference.  It is also cheaper: width[w]
// ~ 20 doublings against the ~128 doublings + additions of a split multiplication.  One-time work per pinned parameter vector.
template <class F>
__global__ void __launch_bounds__(256) table_double_kernel(Affine<F>* __restrict__ out, const Affine<F>* __restrict__ in, uint64_t n, uint32_t doublings,
                                                          F* __restrict__ zbuf) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> p = in[i];
  if constexpr (std::is_same<F, Fq>::value) {
    JacU<FqParams> acc = JacU<FqParams>::zero();
    if (!p.is_zero()) {
      const FqU C = UPow2<FqParams, 266>::get();            // x*2^256 * 2^266 / 2^261 = x * 2^261
      acc = JacU<FqParams>{u_mul(u_from_std(p.x), C), u_mul(u_from_std(p.y), C), UPow2<FqParams, 261>::get()};
#pragma unroll 1
      for (uint32_t k = 0; k < doublings; ++k) acc = jacu_double(acc);   // (a point of order 2 does not exist on y^2 = x^3 + b over Fq: r is odd)
    }
    const Jacobian<F> r = jacu_to_std(acc);
    out[i] = Affine<F>{r.x, r.y};
    zbuf[i] = r.z;
  } else {
    JacU2 acc = JacU2::zero();
    if (!p.is_zero()) {
      const JacTabU2 e = jacu2_tab_from_affine(p.x, p.y);
      acc = JacU2{e.x, e.y, e.z};
#pragma unroll 1
      for (uint32_t k = 0; k < doublings; ++k) acc = jacu2_double(acc);  // (no 2-torsion on the twist either: #E'(Fq2) = r (2q - r) is odd)
    }
    const Jacobian<F> r = jacu2_to_std(acc);
    out[i] = Affine<F>{r.x, r.y};
    zbuf[i] = r.z;
  }
}

template <int GROUP>
int msm_table_build(const void* d_bases, size_t n, void* d_table, size_t table_bytes, void* stream) {
  using F = typename std::conditional<GROUP == 1, Fq, Fq2>::type;
  if (n == 0) return ZK_OK;
  if (!d_bases || !d_table || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  uint32_t c = 0, W = 0;
  uint8_t width[64];
  msm_table_geometry(n, GROUP, &c, &W, width);
  if ((uint64_t)W * n > 0x7fffffffull || table_bytes < (size_t)W * n * sizeof(Affine<F>)) return ZK_ERR_BAD_ARGS;
  hipStream_t st = (hipStream_t)stream;
  char* t = (char*)d_table;
  const size_t plane = n * sizeof(Affine<F>);
  if ((const void*)t != d_bases) ZK_HIP(hipMemcpyAsync(t, d_bases, plane, hipMemcpyDeviceToDevice, st));
  std::lock_guard<std::mutex> launch_lk(g_exp_launch_mu);  // (the Z scratch is per (device, stream): see batch_exp)
  void* zbuf = nullptr;
  int rc = exp_scratch((n * sizeof(F) + 255) & ~(size_t)255, stream, &zbuf);
  if (rc) return rc;
  for (uint32_t w = 0; w + 1 < W; ++w) {
    hipLaunchKernelGGL(table_double_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (Affine<F>*)(t + (size_t)(w + 1) * plane),
                       (const Affine<F>*)(t + (size_t)w * plane), (uint64_t)n, (uint32_t)width[w], (F*)zbuf);
    ZK_HIP(hipGetLastError());
    rc = GROUP == 1 ? batch_normalize_g1(t + (size_t)(w + 1) * plane, zbuf, n, st) : batch_normalize_g2(t + (size_t)(w + 1) * plane, zbuf, n, st);
    if (rc) return rc;
  }
  ZK_HIP(hipStreamSynchronize(st));
  return ZK_OK;
}

bool g2_in_subgroup_host(const Affine<Fq2>& p) { return g2_in_subgroup(p); }

template int batch_exp<Fq>(void*, const void*, int, const void*, int, size_t, void*, const uint32_t*, bool, bool, const uint8_t*);
template int batch_exp<Fq2>(void*, const void*, int, const void*, int, size_t, void*, const uint32_t*, bool, bool, const uint8_t*);
template int batch_mul<Fq>(void*, const uint64_t*, const void*, size_t, void*);
template int batch_mul<Fq2>(void*, const uint64_t*, const void*, size_t, void*);
template int msm_table_build<1>(const void*, size_t, void*, size_t, void*);
template int msm_table_build<2>(const void*, size_t, void*, size_t, void*);
}  // namespace zk
