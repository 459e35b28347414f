// Field-product throughput of the Fq representations on gfx950: the memory format (8 x 32-bit limbs, field.hpp: column-wise
// v_mad_u64_u32 + v_addc_co_u32 carry word, inline asm) against the U-form (9 x 29-bit lazy limbs, fieldu.hpp: one v_mad_u64_u32
// per partial product, no carry flags).  Every lane runs CHAINS independent chains x = x * y of ITERS products; 1024 SIMDs are
// filled at 1, 2 and 4 waves per SIMD.  Output: products per second for the whole device and cycles per wave-product per SIMD.
//
// Third row (round 5): a Montgomery product on the DOUBLE-PRECISION multiplier -- five 52-bit limbs held as doubles,
// R = 2^260.  A partial product a_i b_j (< 2^104) is split exactly by two v_fma_f64 under round-toward-zero:
//     h = fma(a, b, 2^104)              = 2^104 + hi 2^52      (the binade [2^104, 2^105) pins the ulp at 2^52: hi = floor(a b / 2^52))
//     l = fma(a, b, (2^104 + 2^52) - h) = 2^52 + lo            (exact: lo = a b - hi 2^52 < 2^52)
// and hi / lo are the MANTISSA BITS of h / l, so the column sums are 64-bit integer additions of the raw bit patterns (the constant
// exponent words are folded into the columns' start values).  Per partial product: 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer adds, for
// 25 (a b) + 5 (q_i = t_i * (-p^-1) mod 2^52) + 25 (q p) partial products.  A wider-limb variant that keeps the column sums in the f64
// addend does not exist: the hi parts of a column can be chained through the FMA addend only when 2^(ulp + 52) leaves room for the sum
// (limbs <= 44 bits: 78 partial products), and the lo part of each product still needs ITS hi individually (one more v_add_f64 each) --
// four double-precision instructions per partial product either way, against ONE v_mad_u64_u32 per 29 x 29-bit partial product that
// multiplies AND accumulates.  v_fma_f64 issues at the integer mad's rate on this chip (5.3 against 5.55 cycles per wave instruction,
// tools/ubench_valu.hip), so the arithmetic says ~0.6-0.7 x the U-form; the measurement is the row below (profiles/r05_ubench_fieldmul.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../phase2-bn254_amd/csrc/field.hpp"
#include "../phase2-bn254_amd/csrc/fieldu.hpp"
using namespace zk;
constexpr int ITERS = 2000;
template <int CHAINS>
__global__ void __launch_bounds__(256) k_std(Fq* io) {
  Fq x[CHAINS], y = io[1];
  for (int c = 0; c < CHAINS; ++c) { x[c] = io[0]; x[c].l[0] += threadIdx.x + c; }
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = mul(x[c], y);
  Fq s = x[0];
  for (int c = 1; c < CHAINS; ++c) s = add(s, x[c]);
  if (s.l[7] == 0x12345678u) io[2] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_u(Fq* io) {
  FpU<FqParams> x[CHAINS], y = u_from_std(io[1]);
  for (int c = 0; c < CHAINS; ++c) { Fq t = io[0]; t.l[0] += threadIdx.x + c; x[c] = u_from_std(t); }
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = u_mul(x[c], y);
  FpU<FqParams> s = x[0];
  for (int c = 1; c < CHAINS; ++c) s = u_add(s, x[c]);
  Fq r = u_to_std_lt2p(u_mul(s, y));
  if (r.l[7] == 0x12345678u) io[2] = r;
}

// ---- the f64 product (see the header).  A value is 5 doubles holding integers < 2^52, lazily reduced: < 2p in, < 2p out (R = 2^260 > 4p).
struct F5 { double l[5]; };
__device__ __forceinline__ void f64_round_toward_zero() {   // MODE.FP_ROUND[3:2] (f64 / f16) := 3   (hwreg id 1 = MODE, offset 2, width 2)
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
}
constexpr uint64_t F5_MASK = (1ull << 52) - 1, BITS_2E52 = 0x4330000000000000ull, BITS_2E104 = 0x4670000000000000ull;
// The double-precision instructions are written as asm: hipcc's mode-register pass knows that LLVM's fma / fadd mean round-to-nearest and
// puts "s_setreg FP_ROUND = 0" in front of every compiled f64 instruction that follows an explicit mode change (first build of this file:
// the product loop ran in round-to-nearest and the parity line said NO).
__device__ __forceinline__ double f64_fma(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double f64_sub(double a, double b) {   // exact wherever it is used
  double r;
  asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// t_lo += lo(a b), t_hi += hi(a b), as raw bit patterns: the caller's start values hold -(bits(2^52)) / -(bits(2^104)) per expected term
__device__ __forceinline__ void f5_mac(double a, double b, uint64_t& t_lo, uint64_t& t_hi) {
  const double c1 = __longlong_as_double((long long)BITS_2E104), c2 = __longlong_as_double((long long)(BITS_2E104 + 1));   // 2^104, 2^104 + 2^52
  const double h = f64_fma(a, b, c1);
  const double l = f64_fma(a, b, f64_sub(c2, h));
  t_hi += (uint64_t)__double_as_longlong(h);
  t_lo += (uint64_t)__double_as_longlong(l);
}
__device__ __forceinline__ double f5_to_double(uint64_t v52) { return f64_sub(__longlong_as_double((long long)(BITS_2E52 | v52)), __longlong_as_double((long long)BITS_2E52)); }
__device__ __forceinline__ F5 f5_mul(const F5& a, const F5& b) {
  // q in 52-bit limbs, and -q^-1 mod 2^52
  const double P[5] = {(double)0x8c16d87cfd47ull, (double)0x916871ca8d3c2ull, (double)0x181585d97816aull, (double)0xa029b85045b68ull, (double)0x30644e72e131ull};
  const double NP0 = (double)0x20782e4866389ull;
  // column k receives: lo parts of a_i b_j (i + j = k) and q_i p_j (i + j = k), hi parts of both with i + j = k - 1, one lo of t_k * NP0 (k < 5:
  // added to a scratch column, not here).  Start value: minus the exponent words those terms carry.
  uint64_t t[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    int n_lo = 0, n_hi = 0;
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) {
        if (i + j == k) n_lo += 2;
        if (i + j == k - 1) n_hi += 2;
      }
    t[k] = 0ull - (uint64_t)n_lo * BITS_2E52 - (uint64_t)n_hi * BITS_2E104;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) f5_mac(a.l[i], b.l[j], t[i + j], t[i + j + 1]);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    // the q p rows 0 .. i-1 have been added: t[i] is complete but for the exponent words of terms still to come (row i's own lo, the hi
    // of row i's j - 1 ... none for column i: row i adds lo to t[i], hi to t[i+1]) -- so add row i's pending word back for the read
    const uint64_t ti = (t[i] + BITS_2E52) & F5_MASK;        // (t[i] lacks exactly ONE lo term: q_i p_0)
    uint64_t q_lo = 0ull - BITS_2E52, q_hi = 0ull - BITS_2E104;
    f5_mac(f5_to_double(ti), NP0, q_lo, q_hi);               // q_i = ti * (-p^-1) mod 2^52
    const double q = f5_to_double(q_lo & F5_MASK);
#pragma unroll
    for (int j = 0; j < 5; ++j) f5_mac(q, P[j], t[i + j], t[i + j + 1]);
    t[i + 1] += t[i] >> 52;                                   // t[i] == 0 mod 2^52 now
  }
  F5 r;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    r.l[k] = f5_to_double(t[5 + k] & F5_MASK);
    t[6 + k] += t[5 + k] >> 52;
  }
  return r;
}
__device__ __forceinline__ F5 f5_from_std(const Fq& x) {     // 8 x 32-bit limbs -> 5 x 52-bit
  F5 r;
  uint64_t w[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 5; ++k)
    for (int bit = 0; bit < 52; ++bit) {
      const int g = 52 * k + bit;
      if (g < 256) w[k] |= (uint64_t)((x.l[g >> 5] >> (g & 31)) & 1u) << bit;
    }
  for (int k = 0; k < 5; ++k) r.l[k] = (double)w[k];
  return r;
}
__device__ __forceinline__ Fq f5_to_std(const F5& x) {       // (limbs < 2^52, value < 2^256)
  Fq r = Fq::zero();
  for (int k = 0; k < 5; ++k) {
    const uint64_t w = (uint64_t)x.l[k];
    for (int bit = 0; bit < 52; ++bit) {
      const int g = 52 * k + bit;
      if (g < 256) r.l[g >> 5] |= (uint32_t)((w >> bit) & 1ull) << (g & 31);
    }
  }
  return r;
}
__device__ __forceinline__ void f5_pin(F5& x) {   // the conversions that made x are complete here (they may not sink below the mode change)
#pragma unroll
  for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(x.l[k]));
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_f64(Fq* io) {
  F5 x[CHAINS], y = f5_from_std(io[1]);
  for (int c = 0; c < CHAINS; ++c) { Fq t = io[0]; t.l[0] += threadIdx.x + c; x[c] = f5_from_std(t); }
  f5_pin(y);
  for (int c = 0; c < CHAINS; ++c) f5_pin(x[c]);
  f64_round_toward_zero();   // (after the integer -> double conversions: the compiler's mode-register pass resets FP_ROUND around those)
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = f5_mul(x[c], y);
  uint64_t s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int k = 0; k < 5; ++k) s += (uint64_t)__double_as_longlong(x[c].l[k]);
  if (s == 12345ull) io[2] = f5_to_std(x[0]);
}
// parity: x y^n 2^(-260 n) by the f64 product against the same through the memory-format product (2^(-256 n)) times 16^(-n)
__global__ void k_f64_check(const Fq* io, Fq* out, int n) {
  F5 x = f5_from_std(io[0]);
  const F5 y = f5_from_std(io[1]);
  Fq m = io[0];
  const Fq c = inv(from_canonical(Fq{{16, 0, 0, 0, 0, 0, 0, 0}}));   // 16^-1, Montgomery form: mul(t, c) = t / 16
  for (int i = 0; i < n; ++i) m = mul(mul(m, io[1]), c);
  F5 yy = y;
  f5_pin(x);
  f5_pin(yy);
  f64_round_toward_zero();
  for (int i = 0; i < n; ++i) x = f5_mul(x, yy);
  out[0] = reduce_once(f5_to_std(x));   // (< 2p -> canonical)
  out[1] = m;
}

// ---- (round 6) LOWER BOUND of a product whose Montgomery reduction runs on the int8 matrix cores.  The reduction is two products by
// CONSTANTS -- m = (t mod 2^261) * (-p^-1) mod 2^261 and m * p -- i.e., over the 64 elements of a wave, two GEMMs against constant Toeplitz matrices of the
// constants' bytes (V_MFMA_I32_16X16X64_I8: a 64 x 33 by 33 x 33 and a 64 x 33 by 33 x 66 product of signed bytes, i32 column sums < 33 * 2^14).  That would take
// 81 of the U-form's 162 v_mad_u64_u32 off the VALU.  What stays ON the VALU whatever the MFMA costs: the 81 mads of the variable product, re-limbing its low
// half from 29-bit limbs to packed signed bytes, carry-propagating the 33 i32 columns of m into signed bytes for the second GEMM, carry-propagating its 66 columns
// into words, re-limbing those to 29 bits and adding them to the upper half.  This kernel runs exactly that and NOTHING of the matrix part -- no MFMA, none of the
// three lane <-> operand-layout transposes (an element lives in one lane; the MFMA operands spread an element's bytes over four lanes and return 4 columns of 16
// elements per lane: through LDS that is 33 + 33 + 66 dword stores and loads per element more).  The column values the MFMA would deliver are stand-ins aliased to
// live registers (zero instructions), so the figure is a strict lower bound of the real thing's time.  Gate: >= 1.15 x the U-form's rate.
template <int CHAINS>
__global__ void __launch_bounds__(256) k_mfma_glue(Fq* io) {
  FpU<FqParams> x[CHAINS], y = u_from_std(io[1]);
  for (int c = 0; c < CHAINS; ++c) { Fq t = io[0]; t.l[0] += threadIdx.x + c; x[c] = u_from_std(t); }
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      const FpU<FqParams> a = x[c];
      // (1) the variable product: 81 mads, 17 columns -> 18 limbs of 29 bits
      uint32_t t[18];
      uint64_t acc = 0;
#pragma unroll
      for (int k = 0; k < 17; ++k) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int j = k - i;
          if (j >= 0 && j < 9) u_mad(acc, a.l[i], y.l[j]);
        }
        t[k] = (uint32_t)acc & U_MASK;
        acc >>= U_BITS;
      }
      t[17] = (uint32_t)acc;
      // (2) low half: 9 x 29 bits -> 9 packed words of 4 bytes (33 bytes), then SIGNED bytes: (w + 0x80..80 with the carries between words) ^ 0x80..80
      uint32_t w[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int bit = 32 * k, lo = bit / 29, sh = bit % 29;
        uint64_t v = (uint64_t)t[lo] >> sh;
        if (lo + 1 < 9) v |= (uint64_t)t[lo + 1] << (29 - sh);
        if (lo + 2 < 9 && 58 - sh < 32) v |= (uint64_t)t[lo + 2] << (58 - sh);
        w[k] = (uint32_t)v;
      }
      uint64_t cy = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        cy += (uint64_t)w[k] + 0x80808080u;
        w[k] = (uint32_t)cy ^ 0x80808080u;
        cy >>= 32;
      }
      // (3) [MFMA 1: 33 i32 columns of m] stand-ins c1[i] = w[i % 9];  carry-propagate them into 33 bytes, packed, signed again
      uint32_t mw[9] = {};
      int32_t sgn = 0;
#pragma unroll
      for (int i = 0; i < 33; ++i) {
        sgn += (int32_t)w[i % 9];
        mw[i / 4] |= ((uint32_t)sgn & 0xffu) << (8 * (i % 4));
        sgn >>= 8;
      }
      cy = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        cy += (uint64_t)mw[k] + 0x80808080u;
        mw[k] = (uint32_t)cy ^ 0x80808080u;
        cy >>= 32;
      }
      // (4) [MFMA 2: 66 i32 columns of m * p] stand-ins c2[i] = mw[i % 9];  carry-propagate into 17 words
      uint32_t q[17] = {};
      sgn = 0;
#pragma unroll
      for (int i = 0; i < 66; ++i) {
        sgn += (int32_t)mw[i % 9];
        q[i / 4] |= ((uint32_t)sgn & 0xffu) << (8 * (i % 4));
        sgn >>= 8;
      }
      // (5) the upper half of t + m p: re-limb q's bits 261 .. 521 to 29-bit limbs and add (lazy limbs: no carries)
      FpU<FqParams> r;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int bit = 261 + 29 * k, lo = bit / 32, sh = bit % 32;
        uint64_t v = (uint64_t)q[lo] >> sh;
        if (lo + 1 < 17) v |= (uint64_t)q[lo + 1] << (32 - sh);
        r.l[k] = (((uint32_t)v) & U_MASK) + t[9 + k];
      }
      r.l[0] += (uint32_t)sgn & 1u;
      x[c] = r;
    }
  FpU<FqParams> s = x[0];
  for (int c = 1; c < CHAINS; ++c) s = u_add(s, x[c]);
  uint32_t o = 0;
  for (int k = 0; k < 9; ++k) o ^= s.l[k];
  if (o == 0x12345678u) io[2] = u_to_std_lt2p(u_mul(s, y));
}
template <class K>
static void run(const char* name, K kern, int chains, Fq* d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;  // 256 CUs x 4 SIMDs x wps waves = blocks x 4 waves
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double prods = (double)blocks * 256 * chains * ITERS;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * chains * ITERS);
    std::printf("%-22s chains=%d waves/SIMD=%d  %8.3f ms  %8.1f G products/s  %7.1f cycles per wave-product per SIMD\n", name, chains, wps, ms,
                prods / ms * 1e-6, cyc);
  }
}
int main() {
  Fq h[3] = {};
  for (int i = 0; i < 8; ++i) { h[0].l[i] = 0x01234567u * (i + 1); h[1].l[i] = 0x089abcdeu * (i + 3); }
  h[0].l[7] &= 0x0fffffffu; h[1].l[7] &= 0x0fffffffu;
  Fq* d = nullptr;
  hipMalloc(&d, sizeof h);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  run("memory format (asm)", k_std<1>, 1, d);
  run("memory format (asm)", k_std<2>, 2, d);
  run("U-form", k_u<1>, 1, d);
  run("U-form", k_u<2>, 2, d);
  {
    Fq* o = nullptr;
    hipMalloc(&o, 2 * sizeof(Fq));
    bool ok = true;
    for (int n : {1, 2, 7, 100}) {
      hipLaunchKernelGGL(k_f64_check, dim3(1), dim3(1), 0, 0, d, o, n);
      Fq r[2];
      hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
      ok = ok && r[0] == r[1];
    }
    std::printf("f64 (5 x 52-bit, v_fma_f64) product chain == memory-format chain / 16^n for n = 1, 2, 7, 100: %s\n", ok ? "yes" : "NO");
    hipFree(o);
  }
  run("f64 5x52 (v_fma_f64)", k_f64<1>, 1, d);
  run("f64 5x52 (v_fma_f64)", k_f64<2>, 2, d);
  std::printf("-- lower bound of a product with its Montgomery reduction on V_MFMA_I32_*_I8: 81 mads + the VALU glue, no MFMA, no transposes (garbage values) --\n");
  run("MFMA-reduce glue bound", k_mfma_glue<1>, 1, d);
  run("MFMA-reduce glue bound", k_mfma_glue<2>, 2, d);
  hipFree(d);
  return 0;
}
