// "U-form": BN254 field elements on NINE UNSATURATED 29-bit limbs with lazy reduction -- the arithmetic
// the hot kernels run on between their loads and stores.
//
// Why (measured on MI355X, tools/ubench_valu.hip + DESIGN.md section 2): v_mad_u64_u32 accumulates a 64-bit
// addend for free but has no carry-in, and every carry-flag instruction (v_addc_co_u32) costs as much as
// the mad itself.  On saturated 32-bit limbs a column sum needs a third word, i.e. mad + addc per partial
// product.  With 29-bit limbs a column of 9 products (< 2^58 each) plus the Montgomery terms fits a 64-bit
// accumulator outright: ONE v_mad_u64_u32 per partial product, no carry flags anywhere, additions are
// nine independent v_add_u32, and reduction is deferred (7 spare bits: 9 x 29 = 261 > 254).
//
// Conventions
//   value(a) = sum_i a.l[i] * 2^(29 i).  Limbs are u32; "N-form" means l[0..7] < 2^29 (l[8] is whatever the
//   value needs).  Values are only kept congruent mod p and bounded by a small multiple of p.
//   mulu(a, b) = a * b * 2^-261 mod p   (Montgomery with R' = 2^261), result N-form and
//                < a*b/2^261 + p   (so < 2p whenever a*b < 2^261 * p, e.g. a, b < 10p).
//   The 32-byte memory format (x * 2^256 mod p on 8 x 32 bits, field.hpp) is only re-packed, never
//   multiplied, on the way in and out: callers track which power of two a value carries ("domain"),
//   see curveu.hpp and ntt.hip.
// Every precondition on limb / value size is stated at the function; the callers' bound bookkeeping is
// written next to each call.
#pragma once

#include <type_traits>
#include <utility>

#include "field.hpp"

namespace zk {

// for_limbs<N>(f): calls f(std::integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{}).  The index is a
// constant expression inside f, which forces every modulus-derived constant to be folded at compile time.
template <class F, int... I>
ZK_HD void for_limbs_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
ZK_HD void for_limbs(F&& f) {
  for_limbs_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr uint32_t U_BITS = 29;
constexpr uint32_t U_MASK = (1u << U_BITS) - 1u;

// limb i (29 bits) of the 256-bit number given as 8 x 32-bit words
constexpr uint32_t u_limb_of(const uint32_t* w, int i) {
  int bit = 29 * i;
  int word = bit >> 5, off = bit & 31;
  uint64_t two = (uint64_t)w[word] | (word + 1 < 8 ? (uint64_t)w[word + 1] << 32 : 0ull);
  return (uint32_t)(two >> off) & U_MASK;
}

// -p^-1 mod 2^29 from -p^-1 mod 2^32
constexpr uint32_t u_inv29(uint32_t inv32) { return inv32 & U_MASK; }

template <class PR>
struct UParams {
  static constexpr uint32_t P(int i) { return u_limb_of(PR::P, i); }
  static constexpr uint32_t INV = u_inv29(PR::INV);
};

template <class PR>
struct FpU {
  uint32_t l[9];
  ZK_HD static FpU zero() {
    FpU r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
  }
  // exact zero limbs (used for the infinity marker ZZ == 0, which is stored as literal zeros)
  ZK_HD bool limbs_all_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= l[i];
    return o == 0;
  }
};

// ---- re-packing between the memory format (8 x 32) and U limbs; the integer value is unchanged ----
template <class PR>
ZK_HD FpU<PR> u_from_std(const Fp<PR>& a) {
  FpU<PR> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, word = bit >> 5, off = bit & 31;
    uint64_t two = (uint64_t)a.l[word] | (word + 1 < 8 ? (uint64_t)a.l[word + 1] << 32 : 0ull);
    r.l[i] = (uint32_t)(two >> off) & U_MASK;
  }
  return r;
}

// carry propagation: afterwards l[0..7] < 2^29.  Precondition: every l[i] + carry-in < 2^32 (true for all
// limb bounds used in this library: limbs < 2^32 - 2^4).
template <class PR>
ZK_HD FpU<PR> u_carry(const FpU<PR>& a) {
  FpU<PR> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t t = a.l[i] + c;
    r.l[i] = t & U_MASK;
    c = t >> U_BITS;
  }
  r.l[8] = a.l[8] + c;
  return r;
}

// N-form value < 2p  ->  canonical [0, p) in the 8 x 32 memory format
template <class PR>
ZK_HD Fp<PR> u_to_std_lt2p(const FpU<PR>& a) {
  // d = a - p with borrow propagation on 29-bit limbs
  uint32_t d[9];
  uint32_t borrow = 0;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t pi = UParams<PR>::P(i);
    uint32_t t = a.l[i] - pi - borrow;
    borrow = t >> 31;  // limbs < 2^30: a wrapped difference has its top bit set
    d[i] = (i < 8) ? (t & U_MASK) : t;
  });
  uint32_t v[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = borrow ? a.l[i] : d[i];
  Fp<PR> r;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int bit = 32 * w, i = bit / 29, off = bit % 29;
    uint64_t acc = (uint64_t)v[i] >> off;
    int have = 29 - off;
    acc |= (uint64_t)v[i + 1] << have;
    have += 29;
    if (have < 32 && i + 2 < 9) acc |= (uint64_t)v[i + 2] << have;
    r.l[w] = (uint32_t)acc;
  }
  return r;
}

// N-form value < 32p  ->  canonical [0, p) in the 8 x 32 memory format, without a multiplication by one:
// q = floor(v / p) is estimated from the top limb (v >> 232 against p >> 232, an underestimate by at most 2),
// v - q p is formed limb by limb with a signed carry, and two conditional subtractions finish.  ~70 instructions
// against the ~230 of a Montgomery product.
template <class PR>
ZK_HD Fp<PR> u_to_std_lt32p(const FpU<PR>& a) {
  constexpr uint32_t p_top = UParams<PR>::P(8);                         // p >> 232 (22 bits)
  constexpr uint32_t magic = (uint32_t)(0x100000000ull / (p_top + 1));  // floor(2^32 / (p_top + 1))
  const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * magic) >> 32);      // <= floor(v / p), >= floor(v / p) - 2; a.l[8] < 2^28
  FpU<PR> r;
  int64_t carry = 0;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t pi = UParams<PR>::P(i);
    const int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * pi) + carry;
    if constexpr (i < 8) {
      r.l[i] = (uint32_t)t & U_MASK;
      carry = t >> U_BITS;  // arithmetic shift: floor division
    } else {
      r.l[i] = (uint32_t)t;  // the total is non-negative and < 3p
    }
  });
  // r < 3p: one conditional subtraction brings it below 2p, u_to_std_lt2p does the last one
  uint32_t d[9];
  uint32_t borrow = 0;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t pi = UParams<PR>::P(i);
    const uint32_t t = r.l[i] - pi - borrow;
    borrow = t >> 31;
    d[i] = (i < 8) ? (t & U_MASK) : t;
  });
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = borrow ? r.l[i] : d[i];
  return u_to_std_lt2p(r);
}

// limbwise sum; limb bounds add, value bounds add.  No normalisation.
template <class PR>
ZK_HD FpU<PR> u_add(const FpU<PR>& a, const FpU<PR>& b) {
  FpU<PR> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}

template <class PR>
ZK_HD FpU<PR> u_dbl(const FpU<PR>& a) {
  FpU<PR> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] << 1;
  return r;
}

// K = k * p in a redundant limb form whose limbs 0..7 are >= s * 2^29, so that  a + K - b  has
// non-negative limbs 0..7 for every b with limbs 0..7 < s * 2^29.  Limb 8 may wrap in u32 arithmetic; it
// is exact again after u_carry because the total value a + K - b is non-negative (needs value(b) <= k*p).
template <class PR, int K, int S>
struct USubConst {
  static constexpr uint32_t limb(int i) {
    // k*p on 9 normalised limbs
    uint64_t carry = 0;
    uint32_t c = 0;
    for (int j = 0; j <= i; ++j) {
      uint64_t t = (uint64_t)UParams<PR>::P(j) * (uint64_t)K + carry;
      c = (uint32_t)(t & U_MASK);
      carry = t >> U_BITS;
      if (j == 8) c = (uint32_t)t;  // top limb keeps everything
    }
    uint32_t add = (i < 8) ? (uint32_t)S << U_BITS : 0u;
    uint32_t take = (i > 0) ? (uint32_t)S : 0u;
    return c + add - take;
  }
};

// r = a + k*p - b, carried to N-form.
//   preconditions: limbs 0..7 of b < S * 2^29;  value(b) <= K * p;  limbs 0..7 of a < 2^32 - (S+1) * 2^29 - 2^4.
//   result: N-form, value = value(a) + K*p - value(b)  (< value(a) + K*p).
template <int K, int S, class PR>
ZK_HD FpU<PR> u_sub(const FpU<PR>& a, const FpU<PR>& b) {
  FpU<PR> t;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t kc = USubConst<PR, K, S>::limb(i);
    t.l[i] = a.l[i] + kc - b.l[i];
  });
  return u_carry(t);
}

// acc += a * b.  With ZK_CHAIN_MAD (defined by a translation unit before it includes this header) every accumulation
// on the device is followed by an EMPTY asm statement that pins `acc` in a VGPR pair: hipcc cannot reassociate the
// column sum across it, so each column is ONE dependent chain of v_mad_u64_u32 seeded with the carry of the previous
// column.  Left alone, hipcc splits every column into two chains (latency) and merges them with a 64-bit add
// (v_lshl_add_u64) -- 144 extra instructions per mixed addition.  Whether that pays depends on the kernel: with
// 4 waves per SIMD the chain latency is hidden (msm_accumulate_kernel<Fq> 61.3 -> 59.0 ms, ntt_pass_kernel -2 %), the
// register-heavier kernels (Fq2 accumulation, the windowed scalar multiplications) lose.  The s_nop hipcc pads after
// each asm statement issues on the scalar port and costs next to nothing.
ZK_HD void u_mad(uint64_t& acc, uint32_t a, uint32_t b) {
  acc += (uint64_t)a * b;
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_CHAIN_MAD)
  asm("" : "+v"(acc));
#endif
}

// Montgomery product on 29-bit limbs: a * b * 2^-261 mod p.
//   preconditions: max_limb(a) * max_limb(b) < 2^60.5 (e.g. both < 2^30, or one < 2^29 and the other < 2^31.5),
//                  so that every column sum (9 products + 9 m_i*p_j terms + carry) stays below 2^64.
//   result: N-form (l[8] < 2^29), value < value(a)*value(b)/2^261 + p.
// Plain C++ on purpose: there are no carries to steer, hipcc maps every term to one v_mad_u64_u32 and is
// free to schedule the 162 of them.
template <class PR>
ZK_HD FpU<PR> u_mul(const FpU<PR>& a, const FpU<PR>& b) {
  uint32_t m[9];
  FpU<PR> r;
  uint64_t acc = 0;
  for_limbs<17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) u_mad(acc, a.l[i], b.l[j]);
    });
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9 && (k >= 9 || i < k)) {
        constexpr uint32_t pj = UParams<PR>::P(j);
        u_mad(acc, m[i], pj);
      }
    });
    if constexpr (k < 9) {
      constexpr uint32_t p0 = UParams<PR>::P(0);
      m[k] = ((uint32_t)acc * UParams<PR>::INV) & U_MASK;
      u_mad(acc, m[k], p0);  // low 29 bits are now zero
    } else {
      r.l[k - 9] = (uint32_t)acc & U_MASK;
    }
    acc >>= U_BITS;
  });
  r.l[8] = (uint32_t)acc;
  return r;
}

// Montgomery reduction tail shared by u_mul / u_sqr / u_mul2: the callers add their product terms to
// `acc` for column k through the callback and this adds the m_i * p_j terms and extracts the limb.
template <class PR, class ProductTerms>
ZK_HD FpU<PR> u_montgomery_columns(ProductTerms&& terms) {
  uint32_t m[9];
  FpU<PR> r;
  uint64_t acc = 0;
  for_limbs<17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    terms(kc, acc);
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9 && (k >= 9 || i < k)) {
        constexpr uint32_t pj = UParams<PR>::P(j);
        u_mad(acc, m[i], pj);
      }
    });
    if constexpr (k < 9) {
      constexpr uint32_t p0 = UParams<PR>::P(0);
      m[k] = ((uint32_t)acc * UParams<PR>::INV) & U_MASK;
      u_mad(acc, m[k], p0);  // low 29 bits are now zero
    } else {
      r.l[k - 9] = (uint32_t)acc & U_MASK;
    }
    acc >>= U_BITS;
  });
  r.l[8] = (uint32_t)acc;
  return r;
}

// a^2 * 2^-261 mod p with the 36 cross products taken once against the doubled operand: 45 + 81 mads
// instead of 162.   precondition: a N-form (limbs < 2^29; doubled limbs < 2^30, 4 cross terms + 1 square +
// the Montgomery terms per column stay far below 2^64).   result as u_mul.
template <class PR>
ZK_HD FpU<PR> u_sqr(const FpU<PR>& a) {
  uint32_t a2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
  return u_montgomery_columns<PR>([&](auto kc, uint64_t& acc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) {
        if constexpr (i < j) u_mad(acc, a2[i], a.l[j]);
        else if constexpr (i == j) u_mad(acc, a.l[i], a.l[i]);
      }
    });
  });
}

// (a*b + c*d) * 2^-261 mod p with ONE Montgomery reduction: 162 + 81 mads instead of 324.
//   preconditions: all four operands N-form (18 products < 2^58 per column);
//   result: N-form, value < (value(a) value(b) + value(c) value(d)) / 2^261 + p.
template <class PR>
ZK_HD FpU<PR> u_mul2(const FpU<PR>& a, const FpU<PR>& b, const FpU<PR>& c, const FpU<PR>& d) {
  return u_montgomery_columns<PR>([&](auto kc, uint64_t& acc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) {
        u_mad(acc, a.l[i], b.l[j]);
        u_mad(acc, c.l[i], d.l[j]);
      }
    });
  });
}

// (a*b + c*d + e*f) * 2^-261 mod p with ONE Montgomery reduction.
//   preconditions: all six operands N-form (27 products < 2^58 plus the Montgomery terms < 2^63.3 per column);
//   result: N-form, value < (sum of the three value products) / 2^261 + p.
template <class PR>
ZK_HD FpU<PR> u_mul3(const FpU<PR>& a, const FpU<PR>& b, const FpU<PR>& c, const FpU<PR>& d, const FpU<PR>& e, const FpU<PR>& f) {
  return u_montgomery_columns<PR>([&](auto kc, uint64_t& acc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) {
        u_mad(acc, a.l[i], b.l[j]);
        u_mad(acc, c.l[i], d.l[j]);
        u_mad(acc, e.l[i], f.l[j]);
      }
    });
  });
}

// (a*b + c*d + e*f + g*h) * 2^-261 mod p with ONE Montgomery reduction (Fq2 products of sums).
//   preconditions: all eight operands N-form: 36 products < 2^58 plus the Montgomery terms < 2^63.6 per column;
//   result: N-form, value < (sum of the four value products) / 2^261 + p.
template <class PR>
ZK_HD FpU<PR> u_mul4(const FpU<PR>& a, const FpU<PR>& b, const FpU<PR>& c, const FpU<PR>& d, const FpU<PR>& e, const FpU<PR>& f,
                     const FpU<PR>& g, const FpU<PR>& h) {
  return u_montgomery_columns<PR>([&](auto kc, uint64_t& acc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) {
        u_mad(acc, a.l[i], b.l[j]);
        u_mad(acc, c.l[i], d.l[j]);
        u_mad(acc, e.l[i], f.l[j]);
        u_mad(acc, g.l[i], h.l[j]);
      }
    });
  });
}

// ---- product by a CONSTANT with a precomputed quotient (Shoup / Barrett with the quotient of the constant) ----
// For a constant w < p let wq = floor(w * 2^261 / p).  Then q = floor(a * wq / 2^261) is at most 1 + a / 2^261 (+ 1 for the
// truncation below) short of floor(a * w / p), and  r = a * w - q * p  needs only the LOW nine limbs of the two products
// (r < 2^261): 45 + 45 mads, plus 53 for the high half of a * wq -- 143 v_mad_u64_u32 against the 171 + 9 v_mul_lo_u32 of the
// Montgomery product, no serial m_i chain, and no factor 2^-261: the value is a * w itself, so data in the memory format's 2^256
// domain stays there with PLAIN constants.  The NTT multiplies by table twiddles only (ntt.hip).
//   PB = 2^261 - p, so that  a * w - q * p == a * w + q * PB  (mod 2^261)  is one accumulation without signs.
template <class PR>
struct UShoup {
  static constexpr uint32_t PB(int i) { return i == 0 ? (1u << U_BITS) - UParams<PR>::P(0) : U_MASK - UParams<PR>::P(i); }   // P(0) is odd: no carry out of limb 0
};

// a * w mod p (no Montgomery factor) for a table constant (w, wq).
//   preconditions: w canonical (< p) in N-form, wq = floor(w * 2^261 / p) in N-form with l[8] < 2^29;  limbs of a < 2^31
//                  (columns: 9 * 2^60 + 9 * 2^58 + carry < 2^64);  value(a) < 160 p (< 0.94 * 2^261).
//   result: N-form (l[8] < 2^23), value = a * w - q * p with floor(a w / p) - 1 - a / 2^261 - 2^-20 < q <= floor(a w / p):
//           0 <= value < (2 + a / 2^261) p, and since value and q are integers with value == a w (mod p):  value < 2p whenever a < 160p.
//   (the columns 0 .. 6 of a * wq are dropped: their sum is < 7 * 2^60 * 2^174 * 1.01 < 2^238, i.e. below 2^-23 of 2^261)
template <class PR>
ZK_HD FpU<PR> u_mul_shoup(const FpU<PR>& a, const FpU<PR>& w, const FpU<PR>& wq) {
  uint32_t q[9];
  uint64_t acc = 0;
  for_limbs<10>([&](auto kc) {
    constexpr int k = 7 + decltype(kc)::value;                              // columns 7 .. 16 of a * wq
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) u_mad(acc, a.l[i], wq.l[j]);
    });
    if constexpr (k >= 9) q[k - 9] = (uint32_t)acc & U_MASK;
    acc >>= U_BITS;
  });
  q[8] = (uint32_t)acc;                                                     // column 17: the carry (a * wq < 2^261 * 2^261)
  FpU<PR> r;
  acc = 0;
  for_limbs<9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) u_mad(acc, a.l[i], w.l[j]);
    });
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = k - i;
      if constexpr (j >= 0 && j < 9) {
        constexpr uint32_t pb = UShoup<PR>::PB(j);
        u_mad(acc, q[i], pb);
      }
    });
    r.l[k] = (uint32_t)acc & U_MASK;                                        // (k == 8: mod 2^261; the value is < 2p < 2^255)
    acc >>= U_BITS;
  });
  return r;
}

// wq = floor(w * 2^261 / p) for a canonical w given as the plain integer on 8 x 32-bit words (NOT a Montgomery form): restoring
// division, one quotient bit per round.  Table builders and the host only (~10^4 instructions).
template <class PR>
ZK_HD FpU<PR> u_shoup_quotient(const uint32_t w_plain[8]) {
  uint32_t rem[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rem[i] = w_plain[i];
  FpU<PR> q;
  for_limbs<9>([&](auto lc) {                                               // bits 260 .. 0, limb by limb (static limb index)
    constexpr int limb = 8 - decltype(lc)::value;
    uint32_t ql = 0;
    for (int b = 28; b >= 0; --b) {
      uint32_t carry = 0;                                                   // rem = 2 rem  (rem < p < 2^254: no carry out)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t nc = rem[i] >> 31;
        rem[i] = (rem[i] << 1) | carry;
        carry = nc;
      }
      uint32_t d[8];
      uint32_t borrow = 0;
      for_limbs<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr uint32_t pi = PR::P[i];
        const uint64_t t = (uint64_t)rem[i] - pi - borrow;
        d[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 32) & 1u;
      });
      if (!borrow) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rem[i] = d[i];
        ql |= 1u << b;
      }
    }
    q.l[limb] = ql;
  });
  return q;
}

// value == 0 mod p for an N-form value < 2p  (i.e. value in {0, p})
template <class PR>
ZK_HD bool u_is_zero_lt2p(const FpU<PR>& a) {
  uint32_t z = 0, e = 0;
  for_limbs<9>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t pi = UParams<PR>::P(i);
    z |= a.l[i];
    e |= a.l[i] ^ pi;
  });
  return z == 0 || e == 0;
}

// compile-time constant c = 2^e mod p as an N-form U element (e < 512)
template <class PR, int E>
struct UPow2 {
  // computed by repeated doubling mod p on 8 x 32-bit words at compile time
  struct Words {
    uint32_t w[8];
  };
  static constexpr Words compute() {
    Words x{};
    x.w[0] = 1;
    for (int s = 0; s < E; ++s) {
      // x = 2x
      uint32_t carry = 0;
      for (int i = 0; i < 8; ++i) {
        uint32_t nc = x.w[i] >> 31;
        x.w[i] = (x.w[i] << 1) | carry;
        carry = nc;
      }
      // if x >= p: x -= p   (x < 2p < 2^255, no carry out of the top word)
      bool ge = true;
      for (int i = 7; i >= 0; --i) {
        if (x.w[i] > PR::P[i]) { ge = true; break; }
        if (x.w[i] < PR::P[i]) { ge = false; break; }
      }
      if (ge) {
        uint64_t borrow = 0;
        for (int i = 0; i < 8; ++i) {
          uint64_t d = (uint64_t)x.w[i] - PR::P[i] - borrow;
          x.w[i] = (uint32_t)d;
          borrow = (d >> 32) & 1;
        }
      }
    }
    return x;
  }
  static constexpr uint32_t limb(int i) { return u_limb_of(compute().w, i); }
  ZK_HD static FpU<PR> get() {
    FpU<PR> r;
    for_limbs<9>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr uint32_t v = limb(i);
      r.l[i] = v;
    });
    return r;
  }
};

using FqU = FpU<FqParams>;
using FrU = FpU<FrParams>;

// ---- Fq2 in U-form: c0 + c1 u, u^2 = -1 ----
struct Fq2U {
  FqU c0, c1;
  ZK_HD static Fq2U zero() { return Fq2U{FqU::zero(), FqU::zero()}; }
  ZK_HD bool limbs_all_zero() const { return c0.limbs_all_zero() && c1.limbs_all_zero(); }
};

}  // namespace zk
