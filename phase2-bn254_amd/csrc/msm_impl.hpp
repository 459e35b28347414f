#pragma once
// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/mi355zk.h, the reference's
//   bellman/src/multiexp.rs:330-355  multiexp()          (window choice, density contract)
//   bellman/src/multiexp.rs:53-157   multiexp_inner()    (bucket fill, summation by parts, window join)
//   bellman/src/source.rs:36-70      (Arc<Vec<G>>, usize) Source: cursor + error semantics
// and the group law underneath (pairing/src/bn256/ec.rs:301-536).
//
// MI355X design (DESIGN.md "MSM"), not the reference's one-thread-per-window scan:
//   1. msm_digits_plain_kernel  every scalar -> W signed digits (power-of-two or mixed-radix windows), written window-major as
//      msm_tile_hist_kernel     4-byte keys; per (window, super-tile) a histogram of the keys over the coarse bins.
//   2. partition (hand-written, no library sort): column scans of the tile histograms give every (super-tile, window, bin) run
//      its exact position; msm_scatter_kernel moves (key, base index) pairs into their bin through LDS; msm_bucket_kernel sorts
//      every bin by bucket in registers and writes the per-bucket index lists (bucket starts aligned to 4 entries) and bounds;
//      msm_bigbin_* handle bins that skewed inputs overfill.  HBM-bound: 28 B per (point, window).
//   3. msm_size_*_kernel        counting sort of the buckets by size (wave-level load balance, heavy buckets first).
//   4. msm_accumulate_kernel    ONE LANE PER BUCKET for all W * nb buckets at once, in size order: gathers its affine bases and
//                               folds them into a U-form XYZZ accumulator held in VGPRs (8M+2S per point) -- the dominant
//                               kernel; the identity-base check (source.rs:50-52) is fused in;
//      msm_accumulate_heavy / msm_heavy_combine   segment-parallel path for buckets that one lane would walk too long.
//   5. msm_reduce_level_kernel  sum_k k*B_k per window by chunked running sums (levels), then
//      msm_tree_kernel          pairwise trees: plain sums of the levels' A[] and the bit decomposition of the rest;
//                               both on R-domain XYZZ records (curveu.hpp: U-form full additions).
//   6. host                     ONE Horner pass over all partial sums, grouped by their power of two (multiexp.rs:146-154).
// The result is a group element; the reference compares/normalises projective points by value
// (ec.rs:45-85, 596-629), so parity is defined on the affine normalisation.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include <type_traits>
#include <atomic>

#include "curveu.hpp"
#include "device_util.hpp"

namespace zk {

// multiexps inside msm_device per device, G1 and G2 together (defined in msm_g1.hip): a call that has the device to itself may
// take every register of a SIMD (the two-wave G2 accumulation); one of the prover's eight concurrent calls leaves room for the others
extern std::atomic<int> g_msm_inflight[16];

namespace {  // one copy per translation unit (msm_g1.hip / msm_g2.hip): compiled in parallel

constexpr uint32_t SIGN_BIT = 0x80000000u;

// Window layout.  The 254 scalar bits are split into W windows of (nearly) EQUAL width instead of W-1 full
// c-bit windows and a short remainder: a short top window would put its n digits into very few buckets
// (2^26 points, c = 20: 16383 buckets of 4096 entries against 128 everywhere else).  Windows 0..W-2 use signed
// digits (d in [-2^(width-1), 2^(width-1)], width <= c); the TOP window is at most c-1 bits wide and keeps its
// digits unsigned (value + carry <= 2^(c-1) = nb), which is also what absorbs the final carry.
struct MsmGeom {
  uint32_t c;        // bits of the sort field: bucket slots 0..nb-1, nb itself = "no bucket"
  uint32_t W;        // windows
  uint32_t nb;       // bucket slots per window (power-of-two layout: 2^(c-1); narrower windows leave their upper slots empty)
  uint8_t width[64]; // power-of-two layout: bits of window w (c >= 4: at most 64 windows)
  uint8_t shift[64]; // power-of-two layout: first bit of window w
  // MIXED-RADIX layout (rmul != 1): digits in base B = rmul * 2^rshift instead of a power of two, so that the window
  // count is not tied to whole bits -- 254 bits in 12 windows need 21.2 bits each: B = 5 * 2^19 takes 12 windows of
  // 1.31 M buckets where c = 20 takes 13 (one accumulation pass and one sort-pass share less) and c = 22 would pay
  // 2.1 M buckets per window in the reduction.  k = sum_w d_w B^w, d_w in (-B/2, B/2], top digit unsigned <= nb = B/2.
  uint32_t rmul;     // 1 (power-of-two layout) or an odd multiplier 3..15
  uint32_t rshift;
};

template <class F>
__device__ __forceinline__ Affine<F> load_affine(const Affine<F>* p) {
  // sizeof(Affine<F>) is 64 (G1) or 128 (G2): 4 / 8 x 16-byte loads
  Affine<F> r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(Affine<F>) / 16); ++i) d[i] = q[i];
  return r;
}

template <class T>
__device__ __forceinline__ void store_vec(T* p, const T& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) d[i] = s[i];
}
template <class T>
__device__ __forceinline__ T load_vec(const T* p) {
  T r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) d[i] = q[i];
  return r;
}

// ------------------------------------------------------------------------------------------------
// 1. digits.  density == nullptr: FullDensity (source.rs:80-99): base of exponent i is base_offset+i.
//    Otherwise bit i of `density` selects exponent i and the bases are compacted (source.rs:101-118):
//    rank(i) = dprefix[i/32] + popc(density[i/32] & ((1<<i%32)-1)).
// q = q div M, returns q mod M   (q on 8 words of which only words 0..top can be non-zero; M a small constant).  `top` is
// uniform over the launch (it follows from the window number), so the skipped word steps are skipped by scalar branches: the
// dividend loses ~42 bits with every pair of digits taken off, and the multiword work is what the digit extraction costs.
template <uint32_t M>
__host__ __device__ __forceinline__ uint32_t msm_divmod_small(uint32_t q[8], int top) {
  uint32_t rem = 0;
#pragma unroll
  for (int l = 7; l >= 0; --l) {
    if (l > top) continue;
    const uint64_t cur = ((uint64_t)rem << 32) | q[l];
    q[l] = (uint32_t)(cur / M);
    rem = (uint32_t)(cur % M);
  }
  return rem;
}

// two mixed-radix digits at once: (r0 + B r1) = q mod B^2, q = q div B^2, B = M * 2^sh (sh <= 22).  One multiword division by
// M^2 instead of two by M.
template <uint32_t M>
__host__ __device__ __forceinline__ void msm_two_digits(uint32_t q[8], uint32_t sh, uint32_t& r0, uint32_t& r1, int top) {
  const uint64_t low = (((uint64_t)q[1] << 32) | q[0]) & ((1ull << (2 * sh)) - 1ull);
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l <= top) q[l] = (q[l] >> sh) | (l < 7 ? q[l + 1] << (32 - sh) : 0u);
  }
  const uint32_t rem = msm_divmod_small<M * M>(q, top);
  const uint64_t pv = low + ((uint64_t)rem << (2 * sh));  // < B^2 < 2^52
  const uint64_t hi = (pv >> sh) / M;                      // pv div B
  r1 = (uint32_t)hi;
  r0 = (uint32_t)(pv - hi * ((uint64_t)M << sh));
}

// q = q div M^k (M a small odd constant, k uniform over the launch), in steps of M^4 and M (two instantiations per multiplier: the
// digit kernels carry seven multipliers and must stay inside the instruction cache); nbits: q < 2^nbits.
template <uint32_t M>
__host__ __device__ __forceinline__ void msm_div_pow(uint32_t q[8], uint32_t k, int& nbits, int flog_m) {
  constexpr uint32_t M4 = M * M * M * M;
#pragma unroll 1
  for (; k >= 4; k -= 4) {
    (void)msm_divmod_small<M4>(q, nbits > 0 ? (nbits - 1) >> 5 : 0);
    nbits -= 4 * flog_m;
  }
#pragma unroll 1
  for (; k >= 1; k -= 1) {
    (void)msm_divmod_small<M>(q, nbits > 0 ? (nbits - 1) >> 5 : 0);
    nbits -= flog_m;
  }
}

// Digits of one scalar (canonical limbs s[0..7], s[8] = 0) from window w_first on: emit(w, d, neg) for w_first <= w < w_stop,
// d = |digit| (0 = no bucket), neg = SIGN_BIT for a negative digit.  The carry chain of the signed digits starts at w_first
// with carry 0: exact for w_first == 0; for w_first > 0 the carry OUT of window w_first is exact unless that window's raw digit
// sits on the boundary (then it depends on the carry in, which was not computed): the function returns false BEFORE emitting
// anything, and the caller starts again from window 0.  (The digit of window w_first itself may be off by the missing carry:
// callers that start above 0 do not use it.)
// RM: the geometry's multiplier G.rmul as a compile-time constant (1 = power-of-two windows): every digit kernel is instantiated
// per multiplier, so that it carries ONE set of constant divisions instead of seven behind a switch (instruction cache).
template <uint32_t RM, class Emit>
__host__ __device__ __forceinline__ bool msm_scalar_digits_from(const uint32_t s[9], const MsmGeom& G, uint32_t w_first, uint32_t w_stop, Emit emit) {
  uint32_t carry = 0;
  if constexpr (RM != 1) {
    // mixed radix: repeatedly  low = q mod 2^rshift;  q >>= rshift;  (q, r) = divmod(q, rmul);  digit = low + 2^rshift * r
    uint32_t q[8];
    const uint32_t sh = G.rshift, B = RM << sh;
    constexpr int flog_m = RM >= 8 ? 3 : RM >= 4 ? 2 : 1;  // floor(log2 rmul): a digit takes at least sh + flog_m bits off q
    int nbits = 254;                                           // q < 2^nbits (exponents are < r < 2^254)
    if (w_first == 0) {
#pragma unroll
      for (int l = 0; l < 8; ++l) q[l] = s[l];
    } else {
      // q = s div B^w_first = (s >> (sh * w_first)) div rmul^w_first: one multiword shift and one or two multiword divisions
      // instead of w_first digit steps (uniform over the launch: scalar branches, static register indices)
      const uint32_t bits = sh * w_first, ls = bits >> 5, bs = bits & 31;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (ls == (uint32_t)c) {
            lo = l + c < 8 ? s[l + c < 8 ? l + c : 0] : 0u;
            hi = l + c + 1 < 8 ? s[l + c + 1 < 8 ? l + c + 1 : 0] : 0u;
          }
        q[l] = bs ? (lo >> bs) | (hi << (32 - bs)) : lo;
      }
      nbits -= (int)bits;
      msm_div_pow<RM>(q, w_first, nbits, flog_m);
    }
    uint32_t pending = 0;
    bool have_pending = false;
    for (uint32_t w = w_first; w < w_stop; ++w) {
      const int top = nbits > 0 ? (nbits - 1) >> 5 : 0;
      uint32_t d, neg = 0;
      if (w + 1 < G.W) {
        uint32_t raw;
        if (have_pending) {
          raw = pending;
          have_pending = false;
        } else if (w + 2 < G.W) {  // this window and the next one, neither of them the top window
          msm_two_digits<RM>(q, sh, raw, pending, top);  // division by compile-time constants (multiply-high)
          have_pending = true;
          nbits -= 2 * (int)(sh + flog_m);
        } else {
          const uint32_t low = q[0] & ((1u << sh) - 1u);
#pragma unroll
          for (int l = 0; l < 8; ++l) q[l] = (q[l] >> sh) | (l < 7 ? q[l + 1] << (32 - sh) : 0u);
          nbits -= (int)(sh + flog_m);
          const uint32_t rem = msm_divmod_small<RM>(q, top);
          raw = low + (rem << sh);
        }
        if (w == w_first && w_first != 0 && raw == G.nb) return false;  // carry out = carry in: not known here
        d = raw + carry;
        carry = 0;
        if (d > G.nb) {          // d in (B/2, B]  ->  d - B in (-B/2, 0]
          d = B - d;
          neg = d ? SIGN_BIT : 0;
          carry = 1;
        }
      } else {
        d = q[0] + carry;        // top digit, unsigned: <= nb by the choice of B (make_geom_radix)
      }
      emit(w, d, neg);
    }
    return true;
  } else {
  for (uint32_t w = w_first; w < w_stop; ++w) {
    const uint32_t width = G.width[w], bit = G.shift[w];
    const uint32_t limb = bit >> 5, off = bit & 31;
    // (static indices under a uniform condition: a dynamic s[limb] would move the whole array, in every path of the
    // kernel, from registers to scratch memory)
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (limb == (uint32_t)l) { lo = s[l]; hi = s[l + 1]; }
    const uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
    const uint32_t raw = (uint32_t)(two >> off) & ((1u << width) - 1u);
    if (w == w_first && w_first != 0 && w + 1 < G.W && raw == (1u << (width - 1))) return false;  // carry out = carry in
    uint32_t d = raw + carry;
    uint32_t neg = 0;
    carry = 0;
    if (w + 1 < G.W && d > (1u << (width - 1))) {  // d in (2^(width-1), 2^width]  ->  d - 2^width in (-2^(width-1), 0]
      d = (1u << width) - d;
      neg = (d != 0) ? SIGN_BIT : 0;
      carry = 1;
    }
    emit(w, d, neg);
  }
  return true;
  }
}

// The digits of windows w_start <= w < w_stop (a multi-GPU rank that owns a group of windows; the whole range on one GPU): the
// chain starts ONE window below w_start -- the carry into w_start is all the lower windows contribute -- and only the rare
// scalar whose digit there sits exactly on the sign boundary (one in B) walks the chain from window 0.
template <uint32_t RM, class Emit>
__host__ __device__ __forceinline__ void msm_scalar_digits(const uint32_t s[9], const MsmGeom& G, uint32_t w_start, uint32_t w_stop, Emit emit) {
  if (w_start == 0) {  // the whole range (one GPU): its own copy of the chain, specialised for a start at window 0
    (void)msm_scalar_digits_from<RM>(s, G, 0, w_stop, emit);
    return;
  }
  const uint32_t w_first = w_start >= 2 ? w_start - 1 : 0;
  auto windowed = [&](uint32_t w, uint32_t d, uint32_t neg) {
    if (w >= w_start) emit(w, d, neg);
  };
  // (one inlined copy of the chain for every start above 0: the second trip only runs for the boundary scalars)
  uint32_t from = w_first;
#pragma unroll 1
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (msm_scalar_digits_from<RM>(s, G, from, w_stop, windowed)) return;
    from = 0;
  }
}

// run `stmt` with RM bound to G.rmul as a compile-time constant
#define ZK_DISPATCH_RMUL(rmul, stmt)                      \
  switch (rmul) {                                          \
    case 1: { constexpr uint32_t RM = 1; stmt; } break;    \
    case 3: { constexpr uint32_t RM = 3; stmt; } break;    \
    case 5: { constexpr uint32_t RM = 5; stmt; } break;    \
    case 7: { constexpr uint32_t RM = 7; stmt; } break;    \
    case 9: { constexpr uint32_t RM = 9; stmt; } break;    \
    case 11: { constexpr uint32_t RM = 11; stmt; } break;  \
    case 13: { constexpr uint32_t RM = 13; stmt; } break;  \
    default: { constexpr uint32_t RM = 15; stmt; } break;  \
  }

// ------------------------------------------------------------------------------------------------
// 2. PARTITION: the (window, bucket) grouping of the n * WL digits, hand-written (no library sort on the path).
//    Only GROUPING by bucket is needed (bucket membership of multiexp.rs:104-117; the order inside a bucket is irrelevant to
//    the sum), so instead of a full LSD radix sort the digits take ONE coarse and ONE fine step, both staged through LDS:
//      pass A  msm_digits_hist_kernel  scalars -> W signed digits, written window-major as 4-byte keys (bucket | sign); the
//              workgroup keeps an LDS histogram over (window, coarse bin = bucket >> lo_bits) and dumps it once per
//              super-tile of ST scalars: tile_hist[super-tile][window][bin] (u16).
//      scan    msm_colsum / msm_binscan / msm_tileoff: column-wise exclusive prefix sums of tile_hist -> for every
//              (super-tile, window, bin) the exact position of its run inside the bin's region: no atomics, no look-back.
//      pass B  msm_scatter_kernel  one workgroup per (window, super-tile): keys -> LDS histogram ranks -> the tile's
//              elements reordered by bin in LDS -> written out as contiguous runs of (key, base index) pairs.
//      pass C  msm_bucket_kernel   one workgroup per (window, coarse bin) (~2^14 elements, held in registers): LDS histogram
//              over its 2^lo_bits buckets -> bucket bounds first[] / last[] (bucket starts aligned to 4 entries so that the
//              accumulation can read its index list with 16-byte loads) -> indices placed through an LDS staging buffer and
//              written out coalesced.  Bins that outgrow registers / LDS (skewed scalars) take a two-read path.
//    HBM traffic per element: 4 B (keys) written + read, 8 B (pairs) written + read, 4 B (indices) written = 28 B, against
//    3 x 16 B for a three-pass pair sort.
constexpr uint32_t MSM_SIZE_BINS = 4096;   // bins of the counting sort that orders the buckets by size (3b)
constexpr uint32_t PART_THREADS = 1024;
constexpr uint32_t PART_MAX_ST = 16384;       // super-tile: scalars per pass-A histogram dump = elements per pass-B workgroup
constexpr uint32_t PART_MAX_EB = PART_MAX_ST / PART_THREADS;
constexpr uint32_t PART_LO_MAX = 12;          // at most 4096 buckets per coarse bin
constexpr uint32_t PART_EC = 32;              // pass C: elements per lane held in registers
constexpr uint32_t PART_LDS_A = 76 * 1024;    // pass A histogram budget (16-bit counters; two workgroups per CU)
constexpr uint32_t PART_LDS_MAX = 160 * 1024 - 512;
constexpr uint32_t KEY_NONE_MASK = 0x00ffffffu;  // key = bucket (or nb = none) in the low 24 bits | SIGN_BIT

struct PartGeom {
  uint32_t lo_bits;   // fine bits: buckets per coarse bin = 2^lo_bits
  uint32_t nbin;      // coarse bins per window
  uint32_t st;        // super-tile size (multiple of PART_THREADS)
  uint32_t n_st;      // super-tiles
  uint32_t n_chunk;   // row chunks of the column scans
  uint32_t rows_per_chunk;
};

// exclusive prefix sums over len <= 4 * PART_THREADS values: value(idx) -> out(idx, exclusive prefix); returns the total.
// `scratch`: 17 words of LDS.  Every thread of the 1024-thread workgroup must call it.
template <class In, class Out>
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t len, uint32_t* scratch, In value, Out out) {
  constexpr uint32_t PER = 4;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    const uint32_t idx = tid * PER + k;
    v[k] = idx < len ? value(idx) : 0u;
    sum += v[k];
  }
  uint32_t inc = sum;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  __syncthreads();  // scratch may still be read from a previous call
  if (lane == 63) scratch[wv] = inc;
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t w = 0; w < PART_THREADS / 64; ++w) {
      const uint32_t t = scratch[w];
      scratch[w] = run;
      run += t;
    }
    scratch[16] = run;
  }
  __syncthreads();
  uint32_t run = scratch[wv] + inc - sum;
#pragma unroll
  for (uint32_t k = 0; k < PER; ++k) {
    const uint32_t idx = tid * PER + k;
    if (idx < len) out(idx, run);
    run += v[k];
  }
  return scratch[16];
}

// the same over any length: chunks of 4 * PART_THREADS values with a running carry
template <class In, class Out>
__device__ __forceinline__ uint32_t block_scan_long(uint32_t len, uint32_t* scratch, In value, Out out) {
  uint32_t carry = 0;
  for (uint32_t base = 0; base < len; base += 4 * PART_THREADS) {
    const uint32_t m = len - base < 4 * PART_THREADS ? len - base : 4 * PART_THREADS;
    carry += block_scan_1024(m, scratch, [&](uint32_t x) { return value(base + x); }, [&](uint32_t x, uint32_t ex) { out(base + x, carry + ex); });
  }
  return carry;
}

// pass A.  density == nullptr: FullDensity (source.rs:80-99).  Otherwise bit i of `density` selects exponent i
// (source.rs:101-118).  scalars_mont != 0: the exponents are Fr elements in Montgomery form (what the prover holds before
// scalars_into_representations, prover.rs:89-129): the conversion into_repr() is one Montgomery reduction, fused here.
template <uint32_t RM>
__global__ void __launch_bounds__(PART_THREADS) msm_digits_hist_kernel(const uint32_t* __restrict__ scalars, uint64_t n,
                                                                       const uint32_t* __restrict__ density, MsmGeom G, uint32_t w_lo,
                                                                       uint32_t w_hi, int scalars_mont, PartGeom P, uint64_t kstride,
                                                                       uint32_t* __restrict__ keys, uint16_t* __restrict__ tile_hist,
                                                                       unsigned long long* __restrict__ err_scalar, uint64_t i_bias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // counters are 16 bits wide (a super-tile has at most 16384 scalars), two per LDS word: half the LDS, two workgroups per CU,
  // and the dump below is a plain copy (little endian: even cell = low half)
  uint32_t* lh = reinterpret_cast<uint32_t*>(smem);
  const uint32_t WL = w_hi - w_lo, ncell = WL * P.nbin, nword = (ncell + 1) / 2;
  for (uint32_t st = blockIdx.x; st < P.n_st; st += gridDim.x) {
    for (uint32_t t = threadIdx.x; t < nword; t += PART_THREADS) lh[t] = 0;
    __syncthreads();
    const uint64_t i_end = (uint64_t)(st + 1) * P.st < n ? (uint64_t)(st + 1) * P.st : n;
    // the next scalar of the lane is requested before the current one is worked on (~10^3 instructions): the load latency is
    // then hidden inside the lane itself, not only by the other waves
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
    {
      const uint64_t i_first = (uint64_t)st * P.st + threadIdx.x;
      if (i_first < i_end) {
        const uint4* sp = reinterpret_cast<const uint4*>(scalars + i_first * 8);
        n0 = sp[0];
        n1 = sp[1];
      }
    }
    for (uint64_t i = (uint64_t)st * P.st + threadIdx.x; i < i_end; i += PART_THREADS) {
      bool active = true;
      if (density != nullptr) active = (density[i >> 5] >> (i & 31)) & 1;
      uint32_t s[9];
      const uint4 s0 = n0, s1 = n1;
      if (i + PART_THREADS < i_end) {
        const uint4* sp = reinterpret_cast<const uint4*>(scalars + (i + PART_THREADS) * 8);
        n0 = sp[0];
        n1 = sp[1];
      }
      s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w; s[8] = 0;
      if (scalars_mont) {
        Fr f;
#pragma unroll
        for (int l = 0; l < 8; ++l) f.l[l] = s[l];
        f = to_canonical(f);
#pragma unroll
        for (int l = 0; l < 8; ++l) s[l] = f.l[l];
      }
      const uint32_t any = s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7];
      if (active && (s[7] >> 30)) {
        // not a canonical FrRepr (r < 2^254): its top digit would overflow the bucket field of the key.  Reported as bad
        // arguments (the lowest such exponent index), the exponent is skipped.
        atomicMin(err_scalar, (unsigned long long)(i + i_bias));
        active = false;
      }
      if (!active || any == 0) {  // multiexp.rs:93-96: zero exponent skips its base without looking at it
        for (uint32_t wl = 0; wl < WL; ++wl) keys[(uint64_t)wl * kstride + i] = G.nb;
        continue;
      }
      // (a selected base with a non-zero exponent must not be the identity, source.rs:50-52: checked where the base is
      // loaded anyway, in accumulate_run)
      msm_scalar_digits<RM>(s, G, w_lo, w_hi < G.W ? w_hi : G.W, [&](uint32_t w, uint32_t d, uint32_t neg) {
        if (w >= w_lo && w < w_hi) {
          const uint32_t wl = w - w_lo;
          keys[(uint64_t)wl * kstride + i] = d ? ((d - 1) | neg) : G.nb;
          if (d) {
            const uint32_t cell = wl * P.nbin + ((d - 1) >> P.lo_bits);
            atomicAdd(&lh[cell >> 1], 1u << (16u * (cell & 1u)));
          }
        }
      });
    }
    __syncthreads();
    // rows are padded to an even cell count (ncell_pad) so that every row starts on a word
    uint32_t* row = reinterpret_cast<uint32_t*>(tile_hist) + (uint64_t)st * nword;
    for (uint32_t t = threadIdx.x; t < nword; t += PART_THREADS) row[t] = lh[t];
    __syncthreads();
  }
}

// pass A in two kernels (the default): a plain streaming digit kernel -- one scalar per lane, no LDS, any number of workgroups
// in flight -- and the tile histograms taken from the keys it wrote.  The fused kernel above saves the 4 bytes per (point,
// window) the histogram pass reads again, but its 1024-lane workgroups with a 61 KiB LDS histogram stream the scalars at under
// half the rate of the plain kernel (2.75 ms against 1.1 + 0.7 ms at 2^26), and a multi-GPU rank that owns a few windows pays
// the slow scalar stream in full.
template <uint32_t RM>
__global__ void __launch_bounds__(256) msm_digits_plain_kernel(const uint32_t* __restrict__ scalars, uint64_t n, const uint32_t* __restrict__ density,
                                                              MsmGeom G, uint32_t w_lo, uint32_t w_hi, int scalars_mont, uint64_t kstride,
                                                              uint32_t* __restrict__ keys, unsigned long long* __restrict__ err_scalar,
                                                              uint64_t i_bias /* index of exponent 0 of this chunk in the whole call */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t WL = w_hi - w_lo;
  if (i >= n) {
    // the (at most three) slots between the key planes: "no bucket", so that a table-mode call can read the planes as ONE array
    if (i < kstride) for (uint32_t wl = 0; wl < WL; ++wl) keys[(uint64_t)wl * kstride + i] = G.nb;
    return;
  }
  bool active = true;
  if (density != nullptr) active = (density[i >> 5] >> (i & 31)) & 1;
  uint32_t s[9];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  const uint4 s0 = sp[0], s1 = sp[1];
  s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w; s[8] = 0;
  if (scalars_mont) {
    Fr f;
#pragma unroll
    for (int l = 0; l < 8; ++l) f.l[l] = s[l];
    f = to_canonical(f);
#pragma unroll
    for (int l = 0; l < 8; ++l) s[l] = f.l[l];
  }
  const uint32_t any = s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7];
  if (active && (s[7] >> 30)) {   // not a canonical FrRepr: see msm_digits_hist_kernel
    atomicMin(err_scalar, (unsigned long long)(i + i_bias));
    active = false;
  }
  if (!active || any == 0) {
    for (uint32_t wl = 0; wl < WL; ++wl) keys[(uint64_t)wl * kstride + i] = G.nb;
    return;
  }
  msm_scalar_digits<RM>(s, G, w_lo, w_hi < G.W ? w_hi : G.W, [&](uint32_t w, uint32_t d, uint32_t neg) {
    if (w >= w_lo && w < w_hi) keys[(uint64_t)(w - w_lo) * kstride + i] = d ? ((d - 1) | neg) : G.nb;
  });
}

// tile_hist[st][wl][bin] from the keys of (window wl, super-tile st): one workgroup each
__global__ void __launch_bounds__(PART_THREADS) msm_tile_hist_kernel(const uint32_t* __restrict__ keys, uint64_t n, uint64_t kstride, uint32_t nb,
                                                                     uint32_t WL, PartGeom P, uint16_t* __restrict__ tile_hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* lh = reinterpret_cast<uint32_t*>(smem);
  const uint32_t st = blockIdx.x % P.n_st, wl = blockIdx.x / P.n_st;
  const uint64_t i0 = (uint64_t)st * P.st;
  const uint32_t cnt = (uint32_t)(n - i0 < P.st ? n - i0 : P.st);
  for (uint32_t t = threadIdx.x; t < P.nbin; t += PART_THREADS) lh[t] = 0;
  __syncthreads();
  const uint32_t* kp = keys + (uint64_t)wl * kstride + i0;
  for (uint32_t idx = 4u * threadIdx.x; idx < cnt; idx += 4u * PART_THREADS) {
    uint32_t v[4] = {nb, nb, nb, nb};
    if (idx + 4 <= cnt) {
      const uint4 q = *reinterpret_cast<const uint4*>(kp + idx);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      for (uint32_t e = 0; e < 4; ++e)
        if (idx + e < cnt) v[e] = kp[idx + e];
    }
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      const uint32_t b = v[e] & KEY_NONE_MASK;
      if (b < nb) atomicAdd(&lh[b >> P.lo_bits], 1u);
    }
  }
  __syncthreads();
  const uint32_t ncell = WL * P.nbin, stride = (ncell + 1u) & ~1u;
  uint16_t* row = tile_hist + (uint64_t)st * stride + (uint64_t)wl * P.nbin;
  for (uint32_t t = threadIdx.x; t < P.nbin; t += PART_THREADS) row[t] = (uint16_t)lh[t];
}

// column sums of tile_hist over one chunk of rows: csum[chunk][col]
__global__ void __launch_bounds__(256) msm_colsum_kernel(const uint16_t* __restrict__ tile_hist, PartGeom P, uint32_t ncell,
                                                        uint32_t* __restrict__ csum) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x, chunk = blockIdx.y;
  if (col >= ncell) return;
  const uint32_t stride = (ncell + 1u) & ~1u;
  const uint32_t r0 = chunk * P.rows_per_chunk, r1 = r0 + P.rows_per_chunk < P.n_st ? r0 + P.rows_per_chunk : P.n_st;
  uint32_t s = 0;
#pragma unroll 8
  for (uint32_t r = r0; r < r1; ++r) s += tile_hist[(uint64_t)r * stride + col];
  csum[(uint64_t)chunk * ncell + col] = s;
}

// per column: csum[chunk][col] -> exclusive prefix over the chunks, total[col] = the column sum (coalesced across columns)
__global__ void __launch_bounds__(256) msm_colscan_kernel(uint32_t* __restrict__ csum, PartGeom P, uint32_t ncell, uint32_t* __restrict__ total) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncell) return;
  uint32_t run = 0;
  for (uint32_t ch = 0; ch < P.n_chunk; ++ch) {
    const uint32_t v = csum[(uint64_t)ch * ncell + col];
    csum[(uint64_t)ch * ncell + col] = run;
    run += v;
  }
  total[col] = run;
}

// one workgroup: bin_start[col] = elements before (window, bin) `col` in the pair array (exclusive scan of total[]),
// out_start[col] = first slot of the bin's region in the index array (every bucket start is padded to a multiple of 4
// entries: + up to 3 per bucket).
__global__ void __launch_bounds__(PART_THREADS) msm_binscan_kernel(const uint32_t* __restrict__ total, PartGeom P, uint32_t ncell, uint32_t nb,
                                                                   uint32_t* __restrict__ bin_start, uint32_t* __restrict__ out_start) {
  __shared__ uint32_t scratch[32];
  auto padded = [&](uint32_t col) {
    const uint32_t bin = col % P.nbin;
    const uint32_t nf = ((bin + 1) << P.lo_bits) <= nb ? 1u << P.lo_bits : nb - (bin << P.lo_bits);
    return (total[col] + 3u * nf + 3u) & ~3u;
  };
  const uint32_t t0 = block_scan_long(ncell, scratch, [&](uint32_t c) { return total[c]; }, [&](uint32_t c, uint32_t ex) { bin_start[c] = ex; });
  const uint32_t t1 = block_scan_long(ncell, scratch, padded, [&](uint32_t c, uint32_t ex) { out_start[c] = ex; });
  if (threadIdx.x == 0) {
    bin_start[ncell] = t0;
    out_start[ncell] = t1;
  }
}

// tile_off[row][col] = position in the pair array at which super-tile `row` writes its run of (window, bin) `col`
__global__ void __launch_bounds__(256) msm_tileoff_kernel(const uint16_t* __restrict__ tile_hist, const uint32_t* __restrict__ cbase,
                                                         const uint32_t* __restrict__ bin_start, PartGeom P, uint32_t ncell,
                                                         uint32_t* __restrict__ tile_off) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x, chunk = blockIdx.y;
  if (col >= ncell) return;
  const uint32_t stride = (ncell + 1u) & ~1u;
  const uint32_t r0 = chunk * P.rows_per_chunk, r1 = r0 + P.rows_per_chunk < P.n_st ? r0 + P.rows_per_chunk : P.n_st;
  uint32_t run = bin_start[col] + cbase[(uint64_t)chunk * ncell + col];
#pragma unroll 8
  for (uint32_t r = r0; r < r1; ++r) {
    tile_off[(uint64_t)r * ncell + col] = run;
    run += tile_hist[(uint64_t)r * stride + col];
  }
}

// pass B.  One workgroup per (window wl, super-tile st): the tile's keys are ranked inside their coarse bin by LDS atomics,
// reordered by bin in LDS and written out as one contiguous run per bin at tile_off[st][wl][bin].  The pair carries the
// key (bucket | sign) and the BASE index of the exponent: base_offset + i under FullDensity, base_offset + rank(i) for a
// density map (source.rs:101-118: rank(i) = dprefix[i/32] + popc(density[i/32] & ((1 << i%32) - 1))).
__global__ void __launch_bounds__(PART_THREADS) msm_scatter_kernel(const uint32_t* __restrict__ keys, uint64_t n, uint64_t kstride, uint64_t base_offset,
                                                                   const uint32_t* __restrict__ density, const uint32_t* __restrict__ dprefix,
                                                                   uint32_t nb, uint32_t WL, PartGeom P, uint32_t xcds,
                                                                   const uint32_t* __restrict__ tile_off, uint2* __restrict__ pairs,
                                                                   uint64_t flat_stride, uint64_t table_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t nbin4 = (P.nbin + 3u) & ~3u;
  uint32_t* loff = reinterpret_cast<uint32_t*>(smem);             // nbin: counts, then exclusive offsets inside the tile
  uint32_t* delta = loff + nbin4;                                 // nbin: tile_off[bin] - loff[bin] (mod 2^32)
  uint32_t* scratch = delta + nbin4;                              // 32 words
  uint2* staging = reinterpret_cast<uint2*>(scratch + 32);        // st pairs
  // Workgroups are dealt to the XCDs round-robin by block id; runs of one bin written by CONSECUTIVE super-tiles are adjacent
  // in memory, so consecutive super-tiles are given to the same XCD (block id b -> XCD b % xcds takes a contiguous range of
  // super-tiles) and their partial lines merge in that XCD's L2 instead of leaving it one by one.
  uint32_t st, wl;
  {
    const uint32_t per = P.n_st / xcds;  // super-tiles per XCD in the remapped part
    const uint32_t body = per * xcds;    // the first `body` super-tiles are remapped, the remainder keeps the plain order
    const uint32_t b = blockIdx.x;
    if (per != 0 && b < body * WL) {
      const uint32_t x = b % xcds, j = b / xcds;
      st = x * per + j % per;
      wl = j / per;
    } else {
      const uint32_t r = b - body * WL, rem = P.n_st - body;
      st = body + r % rem;
      wl = r / rem;
    }
  }
  const uint64_t i0 = (uint64_t)st * P.st;
  const uint32_t cnt = (uint32_t)(n - i0 < P.st ? n - i0 : P.st);
  for (uint32_t t = threadIdx.x; t < P.nbin; t += PART_THREADS) loff[t] = 0;
  __syncthreads();
  // lane t holds keys 4 * (k * 1024 + t) .. + 3: one 16-byte load (the key planes are kstride = 4 * ceil(n / 4) apart, tiles are
  // multiples of 1024: every tile starts on 16 bytes)
  uint32_t key[PART_MAX_EB], rank[PART_MAX_EB];
  const uint32_t* kp = keys + (uint64_t)wl * kstride + i0;
  const bool wide = true;
#pragma unroll
  for (uint32_t k4 = 0; k4 < PART_MAX_EB / 4; ++k4) {
    const uint32_t idx = 4u * (k4 * PART_THREADS + threadIdx.x);
    uint32_t v[4] = {nb, nb, nb, nb};
    if (wide && idx + 4 <= cnt) {
      const uint4 q = *reinterpret_cast<const uint4*>(kp + idx);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (uint32_t e = 0; e < 4; ++e)
        if (idx + e < cnt) v[e] = kp[idx + e];
    }
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      key[4 * k4 + e] = v[e];
      rank[4 * k4 + e] = 0;
      const uint32_t b = v[e] & KEY_NONE_MASK;
      if (b < nb) rank[4 * k4 + e] = atomicAdd(&loff[b >> P.lo_bits], 1u);
    }
  }
  __syncthreads();
  const uint32_t* toff = tile_off + ((uint64_t)st * WL + wl) * P.nbin;
  const uint32_t total = block_scan_1024(P.nbin, scratch, [&](uint32_t x) { return loff[x]; },
                                         [&](uint32_t x, uint32_t ex) { loff[x] = ex; delta[x] = toff[x] - ex; });
  __syncthreads();
#pragma unroll
  for (uint32_t k4 = 0; k4 < PART_MAX_EB / 4; ++k4) {
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
      const uint32_t k = 4 * k4 + e;
      const uint32_t idx = 4u * (k4 * PART_THREADS + threadIdx.x) + e;
      const uint32_t b = key[k] & KEY_NONE_MASK;
      if (idx < cnt && b < nb) {
        uint64_t i = i0 + idx, tw = 0;
        if (flat_stride != 0) {  // table mode: position = window * flat_stride + exponent; the window's copy of the bases starts at window * table_stride
          tw = i / flat_stride;
          i -= tw * flat_stride;
          tw *= table_stride;
        }
        uint64_t bi = base_offset + i;
        if (density != nullptr) {
          const uint32_t wd = density[i >> 5];
          bi = base_offset + dprefix[i >> 5] + __popc(wd & ((1u << (i & 31)) - 1u));
        }
        bi += tw;
        staging[loff[b >> P.lo_bits] + rank[k]] = make_uint2(key[k], (uint32_t)bi);
      }
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < total; p += PART_THREADS) {
    const uint2 e = staging[p];
    pairs[(uint64_t)(p + delta[(e.x & KEY_NONE_MASK) >> P.lo_bits])] = e;
  }
}

// pass C.  One workgroup per (window wl, coarse bin): its pairs [bin_start[col], bin_start[col + 1]) are grouped by bucket.
//   first[id] / last[id] (id = wl * nb + bucket) delimit the bucket's index list inside `vals`; every list starts at a
//   multiple of 4 entries (16-byte loads in the accumulation; the padding slots are never consumed).
//   vals entry = base index | SIGN_BIT for a negative digit.
__global__ void __launch_bounds__(PART_THREADS) msm_bucket_kernel(const uint2* __restrict__ pairs, const uint32_t* __restrict__ bin_start,
                                                                  const uint32_t* __restrict__ out_start, uint32_t nb, PartGeom P,
                                                                  uint32_t stage_cap, uint32_t* __restrict__ first, uint32_t* __restrict__ last,
                                                                  uint32_t* __restrict__ vals) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t nfmax = 1u << P.lo_bits;
  const uint32_t nfl = nfmax < 4 ? 4 : nfmax;            // (keeps `staging` 16-byte aligned)
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);   // nfmax
  uint32_t* start = hist + nfl;                          // nfmax
  uint32_t* scratch = start + nfl;                       // 32
  uint32_t* staging = scratch + 32;                      // stage_cap
  const uint32_t col = blockIdx.x, wl = col / P.nbin, bin = col % P.nbin;
  const uint32_t beg = bin_start[col], cnt = bin_start[col + 1] - beg;
  const uint32_t ob = out_start[col];
  const uint32_t nf = (bin + 1) << P.lo_bits <= nb ? nfmax : nb - (bin << P.lo_bits);
  const uint32_t fmask = nfmax - 1u;
  for (uint32_t t = threadIdx.x; t < nfmax; t += PART_THREADS) hist[t] = 0;
  __syncthreads();
  if (cnt > PART_EC * PART_THREADS) return;  // a BIG bin (skewed exponents, or very large n): msm_bigbin_* kernels, many workgroups
  uint32_t val[PART_EC], fr[PART_EC];  // fr = fine bucket | rank << PART_LO_MAX
#pragma unroll
  for (uint32_t k = 0; k < PART_EC; ++k) {
    const uint32_t idx = k * PART_THREADS + threadIdx.x;
    val[k] = 0;
    fr[k] = 0;
    if (idx < cnt) {
      const uint2 e = pairs[(uint64_t)beg + idx];
      const uint32_t f = e.x & fmask;
      val[k] = e.y | (e.x & SIGN_BIT);
      fr[k] = f | (atomicAdd(&hist[f], 1u) << PART_LO_MAX);
    }
  }
  __syncthreads();
  const uint32_t padded = block_scan_1024(nf, scratch, [&](uint32_t x) { return (hist[x] + 3u) & ~3u; },
                                          [&](uint32_t x, uint32_t ex) { start[x] = ex; });
  __syncthreads();
  const uint32_t id0 = wl * nb + (bin << P.lo_bits);
  for (uint32_t t = threadIdx.x; t < nf; t += PART_THREADS) {
    first[id0 + t] = ob + start[t];
    last[id0 + t] = ob + start[t] + hist[t];
  }
  if (padded <= stage_cap) {
#pragma unroll
    for (uint32_t k = 0; k < PART_EC; ++k) {
      const uint32_t idx = k * PART_THREADS + threadIdx.x;
      if (idx < cnt) staging[start[fr[k] & ((1u << PART_LO_MAX) - 1u)] + (fr[k] >> PART_LO_MAX)] = val[k];
    }
    __syncthreads();
    // 16-byte stores; `ob` and `padded` are multiples of 4
    uint4* dst = reinterpret_cast<uint4*>(vals + ob);
    const uint4* src = reinterpret_cast<const uint4*>(staging);
    for (uint32_t q = threadIdx.x; q < padded / 4; q += PART_THREADS) dst[q] = src[q];
  } else {
#pragma unroll
    for (uint32_t k = 0; k < PART_EC; ++k) {
      const uint32_t idx = k * PART_THREADS + threadIdx.x;
      if (idx < cnt) vals[(uint64_t)ob + start[fr[k] & ((1u << PART_LO_MAX) - 1u)] + (fr[k] >> PART_LO_MAX)] = val[k];
    }
  }
}

// BIG bins: more elements than one workgroup holds in registers.  Prover-like exponents do this -- a Groth16 witness is full of
// 0 / 1 / small values, so window 0 sends a large share of ALL points into the first few buckets, i.e. into one bin -- and so does
// a uniform input at n > 2^26.  Such a bin is cut into SEGMENTS of PART_EC * 1024 elements, one workgroup each:
//   msm_bigbin_plan_kernel   (one workgroup) the list of big bins and the prefix of their segment counts
//   msm_bigbin_count_kernel  every segment adds its LDS histogram over the bin's buckets to gcnt[bucket id]
//   msm_bigbin_place_kernel  every segment derives the bucket starts from gcnt (scan, 4-entry aligned like the small bins),
//                            reserves its share of each bucket with ONE atomic per non-empty bucket (gcur) and writes its
//                            indices; segment 0 of a bin also writes first[] / last[].
// A fixed grid strides over the segments -- the host never reads the plan back -- and idle workgroups exit at once.
constexpr uint32_t BIG_SEG = PART_EC * PART_THREADS;

struct BigPlan {         // device-side
  uint32_t n_big;        // big bins
  uint32_t total_seg;    // their segments
};

__global__ void __launch_bounds__(PART_THREADS) msm_bigbin_plan_kernel(const uint32_t* __restrict__ bin_start, uint32_t ncell,
                                                                       uint32_t* __restrict__ big_col, uint32_t* __restrict__ big_seg_off,
                                                                       BigPlan* __restrict__ plan) {
  __shared__ uint32_t scratch[32];
  auto cnt_of = [&](uint32_t col) { return bin_start[col + 1] - bin_start[col]; };
  // compact the big bins, then the prefix of their segment counts
  const uint32_t n_big = block_scan_long(ncell, scratch, [&](uint32_t c) { return cnt_of(c) > BIG_SEG ? 1u : 0u; },
                                         [&](uint32_t c, uint32_t ex) { if (cnt_of(c) > BIG_SEG) big_col[ex] = c; });
  __syncthreads();
  const uint32_t n_seg = block_scan_long(n_big, scratch, [&](uint32_t k) { return (cnt_of(big_col[k]) + BIG_SEG - 1) / BIG_SEG; },
                                         [&](uint32_t k, uint32_t ex) { big_seg_off[k] = ex; });
  if (threadIdx.x == 0) {
    plan->n_big = n_big;
    plan->total_seg = n_seg;
  }
}

// The four scans above in ONE single-workgroup launch, for SHORT calls (n_st * ncell small: a 2^16-point call has 16 super-tiles x ~2000
// columns): a call of 0.4 - 0.6 ms pays ~5 us for every launch, however little it does.  Also clears the size histogram of the
// bucket order (one memset less) and writes the (empty-or-not) big-bin plan.
__global__ void __launch_bounds__(PART_THREADS) msm_scan_small_kernel(const uint16_t* __restrict__ tile_hist, PartGeom P, uint32_t ncell, uint32_t nb,
                                                                      uint32_t* __restrict__ col_total, uint32_t* __restrict__ bin_start,
                                                                      uint32_t* __restrict__ out_start, uint32_t* __restrict__ tile_off,
                                                                      uint32_t* __restrict__ size_hist, uint32_t* __restrict__ big_col,
                                                                      uint32_t* __restrict__ big_seg_off, BigPlan* __restrict__ plan) {
  __shared__ uint32_t scratch[32];
  const uint32_t stride = (ncell + 1u) & ~1u;
  for (uint32_t t = threadIdx.x; t < MSM_SIZE_BINS; t += PART_THREADS) size_hist[t] = 0;
  for (uint32_t col = threadIdx.x; col < ncell; col += PART_THREADS) {
    uint32_t s = 0;
    for (uint32_t r = 0; r < P.n_st; ++r) s += tile_hist[(uint64_t)r * stride + col];
    col_total[col] = s;
  }
  __syncthreads();   // (global writes of this workgroup are visible to it after the barrier)
  auto padded = [&](uint32_t col) {
    const uint32_t bin = col % P.nbin;
    const uint32_t nf = ((bin + 1) << P.lo_bits) <= nb ? 1u << P.lo_bits : nb - (bin << P.lo_bits);
    return (col_total[col] + 3u * nf + 3u) & ~3u;
  };
  const uint32_t t0 = block_scan_long(ncell, scratch, [&](uint32_t c) { return col_total[c]; }, [&](uint32_t c, uint32_t ex) { bin_start[c] = ex; });
  const uint32_t t1 = block_scan_long(ncell, scratch, padded, [&](uint32_t c, uint32_t ex) { out_start[c] = ex; });
  if (threadIdx.x == 0) {
    bin_start[ncell] = t0;
    out_start[ncell] = t1;
  }
  __syncthreads();
  for (uint32_t col = threadIdx.x; col < ncell; col += PART_THREADS) {
    uint32_t run = bin_start[col];
    for (uint32_t r = 0; r < P.n_st; ++r) {
      tile_off[(uint64_t)r * ncell + col] = run;
      run += tile_hist[(uint64_t)r * stride + col];
    }
  }
  // the big-bin plan (msm_bigbin_plan_kernel)
  auto cnt_of = [&](uint32_t col) { return bin_start[col + 1] - bin_start[col]; };
  const uint32_t n_big = block_scan_long(ncell, scratch, [&](uint32_t c) { return cnt_of(c) > BIG_SEG ? 1u : 0u; },
                                         [&](uint32_t c, uint32_t ex) { if (cnt_of(c) > BIG_SEG) big_col[ex] = c; });
  __syncthreads();
  const uint32_t n_seg = block_scan_long(n_big, scratch, [&](uint32_t k) { return (cnt_of(big_col[k]) + BIG_SEG - 1) / (BIG_SEG); },
                                         [&](uint32_t k, uint32_t ex) { big_seg_off[k] = ex; });
  if (threadIdx.x == 0) {
    plan->n_big = n_big;
    plan->total_seg = n_seg;
  }
}

// (big bin, segment) of workgroup b: the last k with big_seg_off[k] <= b
__device__ __forceinline__ bool bigbin_locate(const BigPlan* plan, const uint32_t* big_col, const uint32_t* big_seg_off, uint32_t b,
                                              uint32_t* col, uint32_t* seg) {
  if (b >= plan->total_seg) return false;
  uint32_t lo = 0, hi = plan->n_big;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (big_seg_off[mid] <= b) lo = mid;
    else hi = mid;
  }
  *col = big_col[lo];
  *seg = b - big_seg_off[lo];
  return true;
}

__global__ void __launch_bounds__(PART_THREADS) msm_bigbin_count_kernel(const uint2* __restrict__ pairs, const uint32_t* __restrict__ bin_start,
                                                                        const BigPlan* __restrict__ plan, const uint32_t* __restrict__ big_col,
                                                                        const uint32_t* __restrict__ big_seg_off, uint32_t nb, PartGeom P,
                                                                        uint32_t* __restrict__ gcnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  for (uint32_t b = blockIdx.x;; b += gridDim.x) {   // a fixed grid strides over the segments (their number is only known on the device)
    uint32_t col, seg;
    if (!bigbin_locate(plan, big_col, big_seg_off, b, &col, &seg)) return;
    const uint32_t nfmax = 1u << P.lo_bits, fmask = nfmax - 1u;
    const uint32_t wl = col / P.nbin, bin = col % P.nbin;
    const uint32_t beg = bin_start[col] + seg * BIG_SEG, end = bin_start[col + 1];
    const uint32_t cnt = end - beg < BIG_SEG ? end - beg : BIG_SEG;
    for (uint32_t t = threadIdx.x; t < nfmax; t += PART_THREADS) hist[t] = 0;
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < cnt; idx += PART_THREADS) atomicAdd(&hist[pairs[(uint64_t)beg + idx].x & fmask], 1u);
    __syncthreads();
    const uint32_t id0 = wl * nb + (bin << P.lo_bits);
    for (uint32_t t = threadIdx.x; t < nfmax; t += PART_THREADS)
      if (hist[t]) atomicAdd(&gcnt[id0 + t], hist[t]);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(PART_THREADS) msm_bigbin_place_kernel(const uint2* __restrict__ pairs, const uint32_t* __restrict__ bin_start,
                                                                        const uint32_t* __restrict__ out_start, const BigPlan* __restrict__ plan,
                                                                        const uint32_t* __restrict__ big_col, const uint32_t* __restrict__ big_seg_off,
                                                                        uint32_t nb, PartGeom P, int staged, const uint32_t* __restrict__ gcnt,
                                                                        uint32_t* __restrict__ gcur, uint32_t* __restrict__ first,
                                                                        uint32_t* __restrict__ last, uint32_t* __restrict__ vals) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {
  uint32_t col, seg;
  if (!bigbin_locate(plan, big_col, big_seg_off, blk, &col, &seg)) return;
  const uint32_t nfmax = 1u << P.lo_bits, fmask = nfmax - 1u;
  const uint32_t nfl = nfmax < 4 ? 4 : nfmax;
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);   // this segment's count per bucket, then its reserved base inside the bucket
  uint32_t* start = hist + nfl;                          // the bucket's first slot relative to the bin's region
  uint32_t* scratch = start + nfl;                       // 32
  uint32_t* lstart = scratch + 32;                       // staged: the bucket's first slot inside this segment's staging area, then its length
  uint32_t* llen = lstart + nfl;
  uint32_t* staging = llen + nfl;                        // staged: BIG_SEG indices
  const uint32_t wl = col / P.nbin, bin = col % P.nbin;
  const uint32_t beg = bin_start[col] + seg * BIG_SEG, end = bin_start[col + 1];
  const uint32_t cnt = end - beg < BIG_SEG ? end - beg : BIG_SEG;
  const uint32_t ob = out_start[col];
  const uint32_t nf = (bin + 1) << P.lo_bits <= nb ? nfmax : nb - (bin << P.lo_bits);
  const uint32_t id0 = wl * nb + (bin << P.lo_bits);
  for (uint32_t t = threadIdx.x; t < nfmax; t += PART_THREADS) hist[t] = 0;
  __syncthreads();
  uint32_t val[PART_EC], fr[PART_EC];
#pragma unroll
  for (uint32_t k = 0; k < PART_EC; ++k) {
    const uint32_t idx = k * PART_THREADS + threadIdx.x;
    val[k] = 0;
    fr[k] = 0;
    if (idx < cnt) {
      const uint2 e = pairs[(uint64_t)beg + idx];
      const uint32_t f = e.x & fmask;
      val[k] = e.y | (e.x & SIGN_BIT);
      fr[k] = f | (atomicAdd(&hist[f], 1u) << PART_LO_MAX);   // rank < 2^15 (a segment has 2^15 elements): fits above the 12 bucket bits
    }
  }
  __syncthreads();
  block_scan_1024(nf, scratch, [&](uint32_t x) { return (gcnt[id0 + x] + 3u) & ~3u; }, [&](uint32_t x, uint32_t ex) { start[x] = ex; });
  if (staged) block_scan_1024(nf, scratch, [&](uint32_t x) { return hist[x]; }, [&](uint32_t x, uint32_t ex) { lstart[x] = ex; });
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < nf; t += PART_THREADS) {
    if (seg == 0) {
      first[id0 + t] = ob + start[t];
      last[id0 + t] = ob + start[t] + gcnt[id0 + t];
    }
    const uint32_t mine = hist[t];
    if (staged) llen[t] = mine;
    hist[t] = mine ? atomicAdd(&gcur[id0 + t], mine) : 0u;   // this segment's base inside bucket t
  }
  __syncthreads();
  if (staged) {
    // the segment's indices grouped by bucket in LDS, then every bucket's run written by one wave: contiguous stores
    // (64 lanes x 4 B) instead of one 4-byte store per lane all over the bin's region
#pragma unroll
    for (uint32_t k = 0; k < PART_EC; ++k) {
      const uint32_t idx = k * PART_THREADS + threadIdx.x;
      if (idx < cnt) staging[lstart[fr[k] & ((1u << PART_LO_MAX) - 1u)] + (fr[k] >> PART_LO_MAX)] = val[k];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t f = wv; f < nf; f += PART_THREADS / 64) {
      const uint32_t len = llen[f];
      const uint32_t* src = staging + lstart[f];
      uint32_t* dst = vals + (uint64_t)ob + start[f] + hist[f];
      for (uint32_t q = lane; q < len; q += 64) dst[q] = src[q];
    }
    __syncthreads();
    continue;
  }
#pragma unroll
  for (uint32_t k = 0; k < PART_EC; ++k) {
    const uint32_t idx = k * PART_THREADS + threadIdx.x;
    if (idx < cnt) {
      const uint32_t f = fr[k] & ((1u << PART_LO_MAX) - 1u);
      vals[(uint64_t)ob + start[f] + hist[f] + (fr[k] >> PART_LO_MAX)] = val[k];
    }
  }
  __syncthreads();
  }
}

// partition geometry for n scalars, WL windows of nb bucket slots each
inline PartGeom choose_part(uint64_t n, uint32_t WL, uint32_t nb) {
  static const char* env_lo = std::getenv("MI355ZK_PART_LO");
  static const char* env_st = std::getenv("MI355ZK_PART_ST");
  PartGeom P{};
  uint32_t lo_cap = 0;
  while ((1u << lo_cap) < nb && lo_cap < PART_LO_MAX) ++lo_cap;  // one bin holds everything, or 2^PART_LO_MAX buckets
  auto nbin_of = [&](uint32_t lo) { return (uint32_t)(((uint64_t)nb + (1ull << lo) - 1) >> lo); };
  // as fine as the pass-A histogram (WL * nbin words of LDS) allows, but no finer than ~8192 elements per bin need
  // pass A keeps WL * nbin 16-bit counters in LDS (two workgroups per CU while they fit PART_LDS_A, one up to twice that);
  // pass B scans nbin words with 1024 lanes x 4
  auto fits_lds = [&](uint32_t lo, uint64_t budget) { return (uint64_t)WL * nbin_of(lo) * 2 <= budget && nbin_of(lo) <= 4 * PART_THREADS; };
  auto fits = [&](uint32_t lo) { return fits_lds(lo, 2 * PART_LDS_A); };
  auto pop = [&](uint32_t lo) { return (uint64_t)n * (1ull << lo) / nb; };  // mean elements per (window, bin)
  uint32_t lo = lo_cap;
  const uint64_t pop_target = 12288;
  const uint64_t pop_cap = (uint64_t)PART_EC * PART_THREADS * 85 / 100;     // pass C holds a bin in registers: stay clear of the cliff
  while (lo > 0 && pop(lo) > pop_target && fits_lds(lo - 1, PART_LDS_A)) --lo;
  while (lo > 0 && pop(lo) > pop_cap && fits(lo - 1)) --lo;
  while (!fits(lo) && lo < PART_LO_MAX) ++lo;
  if (env_lo) {
    const int v = std::atoi(env_lo);
    if (v >= 0 && v <= (int)lo_cap && fits((uint32_t)v)) lo = (uint32_t)v;
  }
  P.lo_bits = lo;
  P.nbin = nbin_of(lo);
  // Super-tiles as large as LDS allows: a pass-B workgroup pays its scans and barriers once, whatever it moves (2^20 exponents, 16
  // windows: 0.150 ms with 2048-element tiles, 0.049 ms with 16384; round 2 shrank the tiles until there were 512 of them PER WINDOW,
  // which at 16 windows is 8192 workgroups of two elements per lane).  Smaller only while a launch would not even give every CU one
  // workgroup (n_st * WL < 256: 2^16 exponents run 4096-element tiles).
  uint32_t st = PART_MAX_ST;
  while (st > PART_THREADS && (n / st) * WL < 256) st >>= 1;
  if (env_st) {
    const int v = std::atoi(env_st);
    if (v >= (int)PART_THREADS && v <= (int)PART_MAX_ST && v % (int)PART_THREADS == 0) st = (uint32_t)v;
  }
  // pass B holds the tile (8 B per element) and nbin words in LDS
  while (st > PART_THREADS && (uint64_t)st * 8 + (uint64_t)P.nbin * 8 + 256 > PART_LDS_MAX) st -= PART_THREADS;
  if ((uint64_t)st * 8 + (uint64_t)P.nbin * 8 + 256 > PART_LDS_MAX) st = 0;  // cannot happen: nbin <= 4096
  P.st = st;
  P.n_st = (uint32_t)((n + st - 1) / st);
  P.rows_per_chunk = P.n_st > 64 ? (P.n_st + 63) / 64 : 1;
  P.n_chunk = (P.n_st + P.rows_per_chunk - 1) / P.rows_per_chunk;
  return P;
}

// 3b. buckets ordered by size (descending) so that the 64 lanes of a wave own buckets of (nearly) equal length --
//     bucket sizes are Poisson distributed and a wave runs as long as its longest lane -- and so that the few very
//     long buckets of a skewed input come first.  A counting sort on the size (exact below 2048, then in steps of
//     2048): histogram, suffix scan, scatter; the order inside a bin is irrelevant.
__device__ __forceinline__ uint32_t msm_size_bin(uint32_t sz) {
  uint32_t hi = sz >> 11;
  return sz < 2048 ? sz : 2048 + (hi < 2047 ? hi : 2047);
}
__global__ void __launch_bounds__(1024) msm_size_hist_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                            uint32_t n_buckets, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[MSM_SIZE_BINS];
  for (uint32_t t = threadIdx.x; t < MSM_SIZE_BINS; t += blockDim.x) lh[t] = 0;
  __syncthreads();
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += gridDim.x * blockDim.x)
    atomicAdd(&lh[msm_size_bin(last[b] - first[b])], 1u);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < MSM_SIZE_BINS; t += blockDim.x)
    if (lh[t]) atomicAdd(&hist[t], lh[t]);
}
// offs[bin] = number of buckets in larger bins (in place over hist); one workgroup
__global__ void __launch_bounds__(1024) msm_size_scan_kernel(uint32_t* __restrict__ hist) {
  __shared__ uint32_t part[1024];
  constexpr uint32_t PER = MSM_SIZE_BINS / 1024;
  uint32_t v[PER], sum = 0;
  for (uint32_t k = 0; k < PER; ++k) { v[k] = hist[threadIdx.x * PER + k]; sum += v[k]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // inclusive suffix sums
    uint32_t add = threadIdx.x + d < 1024 ? part[threadIdx.x + d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;  // buckets in the bins of higher threads
  for (int k = (int)PER - 1; k >= 0; --k) { hist[threadIdx.x * PER + k] = run; run += v[k]; }
}
constexpr uint32_t MSM_SCATTER_PER = 4;  // buckets per lane
__global__ void __launch_bounds__(1024) msm_size_scatter_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                               uint32_t n_buckets, uint32_t* __restrict__ offs, uint32_t* __restrict__ order,
                                                               uint32_t* __restrict__ sizes_sorted) {
  __shared__ uint32_t cnt[MSM_SIZE_BINS];
  __shared__ uint32_t base[MSM_SIZE_BINS];
  for (uint32_t t = threadIdx.x; t < MSM_SIZE_BINS; t += blockDim.x) cnt[t] = 0;
  __syncthreads();
  uint32_t sz[MSM_SCATTER_PER], rank[MSM_SCATTER_PER];
  const uint32_t b0 = blockIdx.x * (blockDim.x * MSM_SCATTER_PER) + threadIdx.x;
#pragma unroll
  for (uint32_t k = 0; k < MSM_SCATTER_PER; ++k) {
    uint32_t b = b0 + k * blockDim.x;
    if (b < n_buckets) {
      sz[k] = last[b] - first[b];
      rank[k] = atomicAdd(&cnt[msm_size_bin(sz[k])], 1u);
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < MSM_SIZE_BINS; t += blockDim.x)
    if (cnt[t]) base[t] = atomicAdd(&offs[t], cnt[t]);
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < MSM_SCATTER_PER; ++k) {
    uint32_t b = b0 + k * blockDim.x;
    if (b < n_buckets) {
      uint32_t pos = base[msm_size_bin(sz[k])] + rank[k];
      order[pos] = b;
      sizes_sorted[pos] = sz[k];
    }
  }
}
// hist: MSM_SIZE_BINS words, zeroed by the caller
inline void msm_order_by_size(const uint32_t* first, const uint32_t* last, uint32_t n_buckets, uint32_t* hist, uint32_t* order,
                              uint32_t* sizes_sorted, hipStream_t st) {
  uint32_t hb = (n_buckets + 4095) / 4096;
  hipLaunchKernelGGL(msm_size_hist_kernel, dim3(hb < 1024 ? hb : 1024), dim3(1024), 0, st, first, last, n_buckets, hist);
  hipLaunchKernelGGL(msm_size_scan_kernel, dim3(1), dim3(1024), 0, st, hist);
  hipLaunchKernelGGL(msm_size_scatter_kernel, dim3((n_buckets + 1024 * MSM_SCATTER_PER - 1) / (1024 * MSM_SCATTER_PER)), dim3(1024), 0, st, first,
                     last, n_buckets, hist, order, sizes_sorted);
}

constexpr uint32_t MSM_HEAVY_BLOCKS = 65536;  // at most this many buckets take the segment-parallel path (the rest of a pathological input runs one lane per bucket)

// Bucket sums and everything derived from them (segment sums, running sums, tree sums, window sums) are XYZZ records in the R
// domain of curveu.hpp (canonical coordinates in the 2^261 domain, 128 B for G1 and 256 B for G2): additions run on U-form
// arithmetic -- 1.33 x the product rate of the memory format (tools/ubench_fieldmul.hip) and one reduction per Fq2 component.
template <class F>
__host__ __device__ __forceinline__ void rec_add(XYZZ<F>& a, const XYZZ<F>& b) {
  if (b.is_zero()) return;
  if (a.is_zero()) { a = b; return; }
  auto ua = xyzzr_load(a);
  xyzzr_add(ua, xyzzr_load(b));
  a = xyzzr_store(ua);
}
template <class F>
__host__ __device__ __forceinline__ XYZZ<F> rec_to_std(const XYZZ<F>& r) { return xyzzr_to_std(r); }
// R-domain record -> Jacobian in the memory format, for the host join.  The record's limbs, READ in the memory format's 2^256
// domain, are the coordinates times f = 2^5 -- all four by the same f -- and  x = X / ZZ,  y = Y / ZZZ  do not see a common factor
// (which is why xyzz_to_affine takes a record as it is).  A Jacobian triple with the same property, for ANY such quadruple:
//     z = ZZ * ZZZ,   x_j = X * ZZ * ZZZ^2,   y_j = Y * ZZ^3 * ZZZ^2        (x_j / z^2 = X / ZZ,  y_j / z^3 = Y / ZZZ)
// -- 6 products + 2 squarings on the host's 4 x 64-bit Montgomery arithmetic, instead of xyzzr_to_std's four U-form products (plain
// C on nine 29-bit limbs: ~4 x slower per product on a CPU) followed by xyzz_to_jacobian's four: the ~200 window sums of a
// multiexp cost 143 -> ~90 us to join (G2: three times that).
template <class F>
inline Jacobian<F> rec_to_jacobian(const XYZZ<F>& r) {
  if (r.is_zero()) return Jacobian<F>::zero();
  const F t = sqr(r.zzz);
  Jacobian<F> j;
  j.z = mul(r.zz, r.zzz);
  j.x = mul(mul(r.x, r.zz), t);
  j.y = mul(mul(r.y, mul(sqr(r.zz), r.zz)), t);
  return j;
}

// A4: the index list starts on a multiple of 4 entries, is walked with stride 1 and its padding slots are readable (the
// lists the partition writes): a lane reads its indices FOUR AT A TIME with one 16-byte load.  Read one by
// one, a lane touches each 128-byte line of its list 32 times, ~10^4 instructions apart, and by then the line has usually
// left the caches (64 lanes x 16 waves x 32 CUs share an L2 that 52 GB of bases stream through): 8 touches instead of 32.
template <class F>
struct BucketAcc { using type = XYZZU<FqParams>; };
template <>
struct BucketAcc<Fq2> { using type = XYZZU2; };

// acc += the signed points of one index list.  The accumulator is the caller's: zero for a fresh bucket, xyzzu_from_r(record)
// for a bucket carried from an earlier chunk of a streamed multiexp (msm_accumulate_kernel<.., CARRY>).
template <class F, bool A4 = false>
__device__ __forceinline__ typename BucketAcc<F>::type accumulate_run(typename BucketAcc<F>::type acc, const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ vals,
                                               uint32_t j, uint32_t e, uint32_t stride, bool skip_zero, unsigned long long* __restrict__ err_base) {
  // the all-zero record is the point at infinity (no curve point has y == 0).  skip_zero (dense mode, powersoftau's
  // dense_multiexp): it adds nothing.  Otherwise it is the reference's UnexpectedIdentity (source.rs:50-52: a selected base
  // with a non-zero exponent): the lowest such BASE index is reported and the record skipped.
  // (precondition: j < e -- the callers deal with an empty list before they set an accumulator up)
  // The gather of point k+1 is issued, as back-to-back 16-byte loads, before the ~2200 (G2: ~7000) ALU instructions of
  // addition k: the loads of a record then hit the same 128-byte line while it is still in cache (left to the
  // scheduler they drift apart to their uses and the line is fetched more than once: +22 % HBM traffic).
  // A4: this group of four indices, rotated so that q.x is the current one.  (A second group loaded one step ahead was
  // measured too: it costs four more VGPRs -- 129, one over the 128 that allow four waves per SIMD.)
  uint4 q = make_uint4(0, 0, 0, 0);
  uint32_t v;
  if constexpr (A4) {
    q = *reinterpret_cast<const uint4*>(vals + j);
    v = q.x;
  } else {
    v = vals[j];
  }
  Affine<F> p = load_affine(bases + (v & ~SIGN_BIT));
  for (;;) {
    const uint32_t jn = j + stride;
    const bool more = jn < e;
    uint32_t vn = 0;
    Affine<F> pn = p;
    if constexpr (A4) {
      if ((jn & 3u) == 0) {  // uniform over the wave: every lane started on a multiple of 4
        if (more) q = *reinterpret_cast<const uint4*>(vals + jn);
      } else {
        q.x = q.y; q.y = q.z; q.z = q.w;
      }
      vn = q.x;
      if (more) pn = load_affine(bases + (vn & ~SIGN_BIT));
    } else if (more) {
      vn = vals[jn];
      pn = load_affine(bases + (vn & ~SIGN_BIT));
    }
    if (!p.y.is_zero()) {
      if constexpr (std::is_same<F, Fq>::value) xyzzu_add_mixed(acc, p.x, p.y, (v & SIGN_BIT) != 0);
      else xyzzu2_add_mixed(acc, p.x, p.y, (v & SIGN_BIT) != 0);
    } else if (!skip_zero) {
      atomicMin(err_base, (unsigned long long)(v & ~SIGN_BIT));
    }
    if (!more) break;
    v = vn;
    p = pn;
    j = jn;
  }
  return acc;
}

// 4a. heavy buckets (longer than `heavy`: skewed scalars such as the many 0/1 witnesses of a Groth16 prover --
//     every scalar equal to 1 lands in bucket 1 of window 0 -- and the short top window).  A heavy bucket is cut
//     into segments of MSM_HEAVY_SEG entries; every segment is one wave (64 strided partial sums + an
//     LDS tree) and a second kernel adds the segment sums of each bucket, so even a bucket holding a third
//     of all points is spread over thousands of workgroups.  Because `order` is sorted by size, the heavy
//     buckets are order[0..H): msm_heavy_plan_kernel scans ceil(size / SEG) over that prefix.
constexpr uint32_t MSM_HEAVY_SEG = 4096;    // entries per segment at size; short calls cut finer (heavy_seg_for)
constexpr uint32_t MSM_HEAVY_LANES = 64;   // one wave per segment: 64 strided partial sums of <= 64 points, then a 6-level tree

// item_off: hb + 2 words.  [0 .. H]: exclusive scan of the segment counts over the H leading entries of the size order that can be
// heavy, [H] = the number of segments; [hb + 1] = H.  The size order is a counting sort over msm_size_bin -- descending in the BIN, not
// inside a bin -- so H = the entries whose bin is at least the bin of heavy + 1 (a binary search), and the exact test runs on those
// only: uniform exponents have H = 0 and the three launches of the heavy path cost their start-up (0.02 ms instead of the 0.06 ms of
// a scan over all hb entries).
__global__ void __launch_bounds__(1024) msm_heavy_plan_kernel(const uint32_t* __restrict__ sizes_sorted, uint32_t hb, uint32_t heavy, uint32_t seg,
                                                             uint32_t* __restrict__ item_off /* hb + 2 */) {
  __shared__ uint32_t scratch[32];
  const uint32_t bin_min = msm_size_bin(heavy == 0xffffffffu ? heavy : heavy + 1);
  uint32_t lo = 0, hi = hb;  // first index whose bin is below bin_min (uniform over the workgroup: every lane walks the same path)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (msm_size_bin(sizes_sorted[mid]) >= bin_min) lo = mid + 1;
    else hi = mid;
  }
  const uint32_t H = lo;
  const uint32_t tot = block_scan_long(H, scratch, [&](uint32_t i) {
    const uint32_t sz = sizes_sorted[i];
    return sz > heavy ? (sz + seg - 1) / seg : 0u;
  }, [&](uint32_t i, uint32_t ex) { item_off[i] = ex; });
  if (threadIdx.x == 0) {
    item_off[H] = tot;
    item_off[hb + 1] = H;
  }
}

// Sum of the 64 lanes' partial sums of a one-wave workgroup, by QUAD additions (curveu.hpp: xyzzr_add_quad): the sums go to LDS once,
// then quad q = lane / 4 adds the pairs (e, e + s) for e = q, q + 16, ..: 2 + 1 + 1 + 1 + 1 + 1 quad additions in sequence instead of
// six one-lane additions, each 2.4 x (G2: 3 x) shorter.  Every lane of quad 0 -- lane 0 among them -- returns the total.
// `sh`: 64 entries; all 64 lanes must call it (it synchronises the workgroup).
template <class U>
__device__ __forceinline__ U wave_sum_quads(U a, U* sh) {
  sh[threadIdx.x] = a;
  __syncthreads();
  const uint32_t q = threadIdx.x >> 2, role = threadIdx.x & 3u;
  for (uint32_t s = 32; s > 0; s >>= 1) {
    for (uint32_t e = q; e < s; e += 16) {       // (uniform over a quad)
      const U x = xyzzr_add_quad(sh[e], sh[e + s], role);
      if (role == 0) sh[e] = x;
    }
    __syncthreads();
  }
  return sh[0];
}

template <class F>
__global__ void __launch_bounds__(MSM_HEAVY_LANES) msm_accumulate_heavy_kernel(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                                  const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                                  const uint32_t* __restrict__ order, const uint32_t* __restrict__ item_off,
                                                                  uint32_t hb, uint32_t seg, XYZZ<F>* __restrict__ seg_sums, int skip_zero,
                                                                  unsigned long long* __restrict__ err_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using U = typename BucketAcc<F>::type;   // the tree runs on register-form sums (R domain): no re-packing per round
  U* sh = reinterpret_cast<U*>(smem);
  hb = item_off[hb + 1];  // the leading entries that can be heavy (msm_heavy_plan_kernel)
  const uint32_t total = item_off[hb];
  // a fixed-size grid strides over the segments: dispatching one (mostly empty) workgroup per POSSIBLE segment
  // costs milliseconds at 2^26 points
  for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
    // bucket of this segment: last i with item_off[i] <= item
    uint32_t lo = 0, hi = hb;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (item_off[mid] <= item) lo = mid;
      else hi = mid;
    }
    const uint32_t b = order[lo];
    const uint32_t j0 = first[b] + (item - item_off[lo]) * seg;
    const uint32_t e = j0 + seg < last[b] ? j0 + seg : last[b];
    U a = U::zero();
    if (j0 + threadIdx.x < e)   // an R-domain record (curveu.hpp) is what every consumer of bucket sums works on: its register form here
      a = xyzzr_load(xyzzu_to_r(accumulate_run<F>(U::zero(), bases, vals, j0 + threadIdx.x, e, blockDim.x, skip_zero != 0, err_base)));
    a = wave_sum_quads(a, sh);   // (round 4: the 6-level tree on quad additions)
    if (threadIdx.x == 0) store_vec(seg_sums + item, xyzzr_store(a));
    __syncthreads();
  }
}

// bucket = sum of its segment sums (one workgroup per heavy bucket)
template <class F>
__global__ void __launch_bounds__(64) msm_heavy_combine_kernel(const XYZZ<F>* __restrict__ seg_sums, const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ item_off, uint32_t hb, XYZZ<F>* __restrict__ buckets,
                                                              int carry) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using U = typename BucketAcc<F>::type;
  U* sh = reinterpret_cast<U*>(smem);
  hb = item_off[hb + 1];  // the leading entries that can be heavy (msm_heavy_plan_kernel)
  for (uint32_t i = blockIdx.x; i < hb; i += gridDim.x) {
    const uint32_t lo = item_off[i], hi = item_off[i + 1];
    if (hi == lo) continue;  // not heavy (uniform per workgroup)
    U a = U::zero();
    for (uint32_t k = lo + threadIdx.x; k < hi; k += blockDim.x) xyzzr_add(a, xyzzr_load(load_vec(seg_sums + k)));
    a = wave_sum_quads(a, sh);
    if (threadIdx.x == 0) {
      if (carry) xyzzr_add(a, xyzzr_load(load_vec(buckets + order[i])));  // the bucket's sum over the earlier chunks
      store_vec(buckets + order[i], xyzzr_store(a));
    }
    __syncthreads();
  }
}

// 4b. one lane per bucket, buckets taken in size order.  Both groups run on U-form arithmetic (curveu.hpp:
//     29-bit lazy limbs, one v_mad_u64_u32 per partial product, no carry flags).
//     CARRY: the launch continues buckets that an earlier chunk of the same (streamed) multiexp has written: a bucket without
//     entries in this chunk is left alone, the others take their record up again (xyzzu_from_r: two products).
template <class F, bool A4, bool CARRY>
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                            const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                                                            const uint32_t* __restrict__ order, uint32_t heavy, uint32_t hb, uint32_t n_buckets,
                                                            XYZZ<F>* __restrict__ buckets, int skip_zero, unsigned long long* __restrict__ err_base) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_buckets) return;
  const uint32_t b = order[i];
  const uint32_t j = first[b], e = last[b];
  if (i < hb && e - j > heavy) return;  // done by msm_accumulate_heavy_kernel
  if (j >= e) {
    if constexpr (!CARRY) store_vec(buckets + b, XYZZ<F>::zero());
    return;
  }
  typename BucketAcc<F>::type acc = BucketAcc<F>::type::zero();
  // (r6, profiles/r06_carry_cost.txt: what a carried launch costs over a fresh one is its first addition -- a full mixed addition where the
  //  fresh launch copies the point into an empty accumulator -- i.e. one addition per TOUCHED BUCKET per extra chunk; the record's load and
  //  its two products are not measurable.  Adding the record at the END by a full addition instead was 1.2 x slower per launch: ~3000
  //  instructions executed once per wave run from a cold instruction cache.)
  if constexpr (CARRY) acc = xyzzu_from_r(load_vec(buckets + b));
  store_vec(buckets + b, xyzzu_to_r(accumulate_run<F, A4>(acc, bases, vals, j, e, 1, skip_zero != 0, err_base)));
}
// G2 with >= 2^19 buckets: the same kernel at TWO waves per SIMD.  With the rare doubling in U-form (xyzzu2_double_affine) the Fq2 loop
// needs 262 registers -- it took 256 + 147 with the saturated-limb doubling inlined -- and at a budget of 256 the seven that do not
// fit (32 B of scratch per lane) cost less than the second wave brings.  (A kernel of its own, so that the G1 instantiation of the
// template above keeps its code.)
template <bool A4, bool CARRY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_accumulate_g2w2_kernel(
    const Affine<Fq2>* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
    const uint32_t* __restrict__ order, uint32_t heavy, uint32_t hb, uint32_t n_buckets, XYZZ<Fq2>* __restrict__ buckets, int skip_zero,
    unsigned long long* __restrict__ err_base) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_buckets) return;
  const uint32_t b = order[i];
  const uint32_t j = first[b], e = last[b];
  if (i < hb && e - j > heavy) return;  // done by msm_accumulate_heavy_kernel
  if (j >= e) {
    if constexpr (!CARRY) store_vec(buckets + b, XYZZ<Fq2>::zero());
    return;
  }
  XYZZU2 acc = XYZZU2::zero();
  if constexpr (CARRY) acc = xyzzu_from_r(load_vec(buckets + b));
  store_vec(buckets + b, xyzzu_to_r(accumulate_run<Fq2, A4>(acc, bases, vals, j, e, 1, skip_zero != 0, err_base)));
}

// 4b'. G2: one PAIR of lanes per bucket (curveu.hpp: PairAcc2 / pair_add_mixed -- the even lane keeps (X, ZZ) and gathers the
//     base's x, the odd lane keeps (Y, ZZZ) and gathers y; the same products as the one-lane addition, split evenly, at half the
//     registers per lane).  Same lists, same order, same records as msm_accumulate_kernel<Fq2>: bit-identical bucket sums.
constexpr uint32_t MSM_PAIR_MAX_BUCKETS = 3u << 17;   // (2^18 points: 17 windows x 2^14 buckets = 278 k -> pairs; 2^19: 16 x 2^15 = 524 k -> lanes)
template <class F, bool A4>
__device__ __forceinline__ typename PairAccOf<F>::type accumulate_run_pair(typename PairAccOf<F>::type acc, const Affine<F>* __restrict__ bases,
                                                                           const uint32_t* __restrict__ vals, uint32_t j, uint32_t e, bool odd, bool skip_zero,
                                                                           unsigned long long* __restrict__ err_base) {
  // (accumulate_run with stride 1; both lanes of the pair walk the same list and stay together)
  uint4 q = make_uint4(0, 0, 0, 0);
  uint32_t v;
  if constexpr (A4) {
    q = *reinterpret_cast<const uint4*>(vals + j);
    v = q.x;
  } else {
    v = vals[j];
  }
  const uint32_t co = odd ? 1u : 0u;
  F p = load_vec(reinterpret_cast<const F*>(bases + (v & ~SIGN_BIT)) + co);
  for (;;) {
    const uint32_t jn = j + 1;
    const bool more = jn < e;
    uint32_t vn = 0;
    F pn = p;
    if constexpr (A4) {
      if ((jn & 3u) == 0) {
        if (more) q = *reinterpret_cast<const uint4*>(vals + jn);
      } else {
        q.x = q.y; q.y = q.z; q.z = q.w;
      }
      vn = q.x;
      if (more) pn = load_vec(reinterpret_cast<const F*>(bases + (vn & ~SIGN_BIT)) + co);
    } else if (more) {
      vn = vals[jn];
      pn = load_vec(reinterpret_cast<const F*>(bases + (vn & ~SIGN_BIT)) + co);
    }
    const uint32_t ynz = pair_dpp<PAIR_ODD>(coord_or(p));   // the all-zero record is the point at infinity: y == 0 (the odd lane's coordinate)
    if (ynz != 0) {
      acc = pair_add_mixed(acc, p, (v & SIGN_BIT) != 0, odd);
    } else if (!skip_zero && odd) {
      atomicMin(err_base, (unsigned long long)(v & ~SIGN_BIT));
    }
    if (!more) break;
    v = vn;
    p = pn;
    j = jn;
  }
  return acc;
}

template <class F, bool A4, bool CARRY>
__device__ __forceinline__ void accumulate_pair_body(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first,
                                                     const uint32_t* __restrict__ last, const uint32_t* __restrict__ order, uint32_t heavy, uint32_t hb,
                                                     uint32_t n_buckets, XYZZ<F>* __restrict__ buckets, int skip_zero, unsigned long long* __restrict__ err_base) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, i = gt >> 1;
  const bool odd = (gt & 1u) != 0;
  if (i >= n_buckets) return;                          // (whole pairs leave: every test up to the loop is on i)
  const uint32_t b = order[i];
  const uint32_t j = first[b], e = last[b];
  if (i < hb && e - j > heavy) return;
  F* rec = reinterpret_cast<F*>(buckets + b);          // {x, y, zz, zzz}: this lane's coordinates are rec[odd] and rec[2 + odd]
  const uint32_t co = odd ? 1u : 0u;
  if (j >= e) {
    if constexpr (!CARRY) {
      store_vec(rec + co, F::zero());
      store_vec(rec + 2 + co, F::zero());
    }
    return;
  }
  typename PairAccOf<F>::type acc = PairAccOf<F>::type::zero();
  if constexpr (CARRY) acc = pair_from_r(load_vec(rec + co), load_vec(rec + 2 + co));
  acc = accumulate_run_pair<F, A4>(acc, bases, vals, j, e, odd, skip_zero != 0, err_base);
  F oa, oz;
  pair_to_r(acc, oa, oz);
  store_vec(rec + co, oa);
  store_vec(rec + 2 + co, oz);
}
template <bool A4, bool CARRY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) msm_accumulate_pair_kernel(
    const Affine<Fq2>* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
    const uint32_t* __restrict__ order, uint32_t heavy, uint32_t hb, uint32_t n_buckets, XYZZ<Fq2>* __restrict__ buckets, int skip_zero,
    unsigned long long* __restrict__ err_base) {
  accumulate_pair_body<Fq2, A4, CARRY>(bases, vals, first, last, order, heavy, hb, n_buckets, buckets, skip_zero, err_base);
}
template <bool A4, bool CARRY>
__global__ void __launch_bounds__(256) msm_accumulate_pair_g1_kernel(
    const Affine<Fq>* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
    const uint32_t* __restrict__ order, uint32_t heavy, uint32_t hb, uint32_t n_buckets, XYZZ<Fq>* __restrict__ buckets, int skip_zero,
    unsigned long long* __restrict__ err_base) {
  accumulate_pair_body<Fq, A4, CARRY>(bases, vals, first, last, order, heavy, hb, n_buckets, buckets, skip_zero, err_base);
}

// 4c. SHORT calls: the launch above lasts as long as its LONGEST bucket -- a lane adds a point every ~8 us however idle the device is,
//     and among 2^18 buckets of 4 points on average some hold 14 (2^16 points: 139 us of accumulation where the additions themselves
//     are 80 us of the device's time).  Buckets longer than `split_t` (and not heavy) are therefore taken out of the lane-per-bucket
//     launch and walked by a QUAD of lanes each: lane r of the quad adds entries r, r + 4, ... (its own accumulator), the four partial
//     sums are brought to all four lanes by DPP and joined by three quad additions (curveu.hpp: xyzzr_add_quad).  A 14-entry bucket is
//     then 4 mixed additions + the join instead of 14.  The buckets are order[0 .. hb) (size order): the main kernel skips them with
//     its `heavy` test set to split_t.  First chunk / unchunked calls only (no carried record).
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_accumulate_split_kernel(
    const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
    const uint32_t* __restrict__ order, uint32_t split_t, uint32_t heavy, uint32_t heavy_hb, uint32_t hb, XYZZ<F>* __restrict__ buckets,
    int skip_zero, unsigned long long* __restrict__ err_base) {
  using U = typename BucketAcc<F>::type;
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, i = gt >> 2, role = gt & 3u;
  if (i >= hb) return;                                 // (whole quads leave: i is the same for the four lanes)
  const uint32_t b = order[i];
  const uint32_t j = first[b], e = last[b];
  // the lane-per-bucket launch has the short ones, the segment-parallel path the heavy ones AMONG order[0 .. heavy_hb): a heavy bucket
  // past that reach (more than MSM_HEAVY_BLOCKS over-long buckets: skewed exponents) is walked here -- the main kernel skips every
  // bucket of order[0 .. max(hb, heavy_hb)) longer than split_t, so nobody else would take it
  if (e - j <= split_t || (e - j > heavy && i < heavy_hb)) return;
  U part = xyzzr_load(XYZZ<F>::zero());
  if (j + role < e) part = xyzzr_load(xyzzu_to_r(accumulate_run<F, false>(U::zero(), bases, vals, j + role, e, 4, skip_zero != 0, err_base)));
  // (split_t >= 4: every lane has at least one entry; the test above is for safety)
  const U p0 = quad_fetch<0>(part), p1 = quad_fetch<1>(part), p2 = quad_fetch<2>(part), p3 = quad_fetch<3>(part);
  const U sum = xyzzr_add_quad(xyzzr_add_quad(p0, p1, role), xyzzr_add_quad(p2, p3, role), role);
  if (role == 0) store_vec(buckets + b, xyzzr_store(sum));
}

// 5. bucket reduction  T_w = sum_{k=1..nb} k * B_k  per window, without scalar multiplications:
//    split the index x = ch*L + y:   sum_x (x+off) B[x] = sum_ch A[ch] + L * sum_ch ch * S[ch]
//    with A[ch] = sum_y (y+off) B[ch*L+y] (running sums) and S[ch] = sum_y B[ch*L+y]; the second term
//    is the same problem on the L-times shorter array S with off = 0.  Each level is one launch of
//    msm_reduce_level_kernel (a lane per chunk) plus a per-window sum of its A[]; the host applies
//    the powers of L (log2 L doublings per level) while it joins the windows.
//    The two running-sum additions share ONE inlined xyzz_add (selected operands): these kernels are
//    not hot, and a single call site keeps the gfx950 code size / compile time down (Fq2 above all).
template <class F>
__global__ void __launch_bounds__(256) msm_reduce_level_kernel(const XYZZ<F>* __restrict__ in, uint32_t count, uint32_t L, uint32_t off,
                                                              uint32_t W, XYZZ<F>* __restrict__ outA, XYZZ<F>* __restrict__ outS) {
  uint32_t chunks = (count + L - 1) / L;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= chunks * W) return;
  uint32_t w = t / chunks, ch = t % chunks;
  uint32_t lo = ch * L, hi = lo + L < count ? lo + L : count;
  const XYZZ<F>* B = in + (uint64_t)w * count;
  uint32_t steps = 2 * (hi - lo);
  // R-domain records: both running sums stay in U-form registers, a bucket is re-packed (no product) when it is loaded
  auto run = xyzzr_load(XYZZ<F>::zero()), acc = run;
  for (uint32_t it = 0; it < steps; ++it) {
    uint32_t x = hi - 1 - (it >> 1);
    // off == 1: run += B[x]; acc += run   (weights y+1)      off == 0: acc += run; run += B[x]   (weights y)
    bool do_acc = ((it & 1) != 0) == (off != 0);
    auto a = do_acc ? acc : run;
    auto b = do_acc ? run : xyzzr_load(load_vec(B + x));
    xyzzr_add(a, b);
    if (do_acc) acc = a;
    else run = a;
  }
  store_vec(outA + t, xyzzr_store(acc));
  store_vec(outS + t, xyzzr_store(run));
}

// The same level with a QUAD of lanes per chunk (curveu.hpp: xyzzr_add_quad -- four lanes share the products of every addition):
// for the levels that are left with too few chunks to fill the device, where a lane's 2L dependent additions are what the launch
// lasts.  Lanes 4t .. 4t+3 hold the same running sums.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_reduce_level_quad_kernel(const XYZZ<F>* __restrict__ in, uint32_t count, uint32_t L, uint32_t off,
                                                                   uint32_t W, XYZZ<F>* __restrict__ outA, XYZZ<F>* __restrict__ outS) {
  const uint32_t chunks = (count + L - 1) / L;
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, t = gt >> 2, role = gt & 3u;
  if (t >= chunks * W) return;   // (whole quads leave: t is the same for the four lanes)
  const uint32_t w = t / chunks, ch = t % chunks;
  const uint32_t lo = ch * L, hi = lo + L < count ? lo + L : count;
  const XYZZ<F>* B = in + (uint64_t)w * count;
  // run_x = B[x] + .. + B[hi-1];  acc = sum of run_x over x (off == 1: weights y + 1) or over x > lo (off == 0: weights y).
  // (two call sites and no selected operands: with the lane kernel's `a = do_acc ? acc : run` hipcc kept the sums in scratch here)
  auto run = xyzzr_load(XYZZ<F>::zero()), acc = run;
  for (uint32_t x = hi; x-- > lo;) {
    run = xyzzr_add_quad(run, xyzzr_load(load_vec(B + x)), role);
    if (off != 0 || x > lo) acc = xyzzr_add_quad(acc, run, role);
  }
  if (role == 0) store_vec(outA + t, xyzzr_store(acc));
  if (role == 1) store_vec(outS + t, xyzzr_store(run));
}

// 5b/5c. the tail of the reduction, by trees.  A running-sum level costs 2L dependent additions however few
//     elements are left, so once at most MSM_FINAL_MAX elements per window remain the weighted sum of the last S[]
//     is finished by BIT DECOMPOSITION:  sum_x (x+off) S[x] = sum_j 2^j * (sum over x with bit j of (x+off) set of S[x]),
//     and the plain sums of every level's A[] ride in the same launch.  One workgroup sums a slice of MSM_TREE_SLICE
//     elements of one job (A of a level, or one bit) of one window: one pair per lane, then an LDS tree (depth 9);
//     further launches of the same kernel sum the slice sums.  The host applies the powers of two and of L.
//     The grid is 1-D over the NON-EMPTY (window, job, slice) triples: jobs of one launch differ in length, and a
//     (slices, jobs, windows) grid put every short job's only workgroup on the same XCD (block id = multiple of 8).
constexpr uint32_t MSM_FINAL_MAX = 1024;  // (2048 and 8192 measured in round 2: the bit-decomposition trees cost more than the level they replace)
constexpr uint32_t MSM_TREE_SLICE = 512;
constexpr uint32_t MSM_MAX_LEVELS = 8;
constexpr uint32_t MSM_MAX_JOBS = MSM_MAX_LEVELS + 24;
template <class F>
struct TreeJobs {
  // job j sums the elements x < cnt[j] of  in[j] + w * stride[j] + rep * rep_stride[j]  taken elem_stride[j] apart (bit[j] < 0), or
  // only the x with bit bit[j] of (x + off[j]) set; a job is a FAMILY of reps[j] such sums (the rows / the columns of the 2-D tail);
  // an element exists while rep * rep_stride + x * elem_stride < limit[j]
  const XYZZ<F>* in[MSM_MAX_JOBS];
  XYZZ<F>* out[MSM_MAX_JOBS];        // slice sum of (w, rep, slice) -> out[j][(w * out_w[j] + rep) * slices(j) + slice]
  uint32_t cnt[MSM_MAX_JOBS];
  uint32_t stride[MSM_MAX_JOBS];
  uint32_t reps[MSM_MAX_JOBS], rep_stride[MSM_MAX_JOBS], elem_stride[MSM_MAX_JOBS], limit[MSM_MAX_JOBS];
  uint32_t out_w[MSM_MAX_JOBS];
  uint32_t off[MSM_MAX_JOBS];
  int32_t bit[MSM_MAX_JOBS];
  uint32_t first_block[MSM_MAX_JOBS + 1];  // prefix sums of reps * slices: blocks per window = first_block[n_jobs]
  uint32_t n_jobs;
  uint32_t quad;                           // 1: the last rounds of a slice's tree run four lanes per addition (G1)
};
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_tree_kernel(const TreeJobs<F> J) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using U = typename BucketAcc<F>::type;   // sums travel between the rounds in REGISTER form (U-form limbs): no re-packing per round
  U* sh = reinterpret_cast<U*>(smem);
  const uint32_t per_w = J.first_block[J.n_jobs];
  const uint32_t w = blockIdx.x / per_w, r = blockIdx.x % per_w;
  uint32_t job = 0;
  while (job + 1 < J.n_jobs && J.first_block[job + 1] <= r) ++job;
  const uint32_t count = J.cnt[job];
  const uint32_t slices = (count + MSM_TREE_SLICE - 1) / MSM_TREE_SLICE;
  const uint32_t idx = r - J.first_block[job];
  const uint32_t rp = idx / slices, slice = idx % slices;
  const int32_t bit = J.bit[job];
  const uint32_t off = J.off[job], es = J.elem_stride[job];
  const uint32_t rbase = rp * J.rep_stride[job], limit = J.limit[job];
  const XYZZ<F>* P = J.in[job] + (uint64_t)w * J.stride[job] + rbase;
  const uint32_t x0 = slice * MSM_TREE_SLICE + threadIdx.x, x1 = x0 + 256;
  XYZZ<F> r0 = XYZZ<F>::zero(), r1 = XYZZ<F>::zero();
  if (x0 < count && (uint64_t)rbase + (uint64_t)x0 * es < limit && (bit < 0 || (((x0 + off) >> bit) & 1))) r0 = load_vec(P + (uint64_t)x0 * es);
  if (x1 < count && (uint64_t)rbase + (uint64_t)x1 * es < limit && (bit < 0 || (((x1 + off) >> bit) & 1))) r1 = load_vec(P + (uint64_t)x1 * es);
  U acc = xyzzr_load(r0), other = xyzzr_load(r1);   // (the zero record loads as the zero accumulator: ZZ == 0 limbs)
  // ONE inlined xyzz_add for the pair and for every tree level (code size).  Lanes [s, 2s) publish, lanes [0, s)
  // consume; the regions written in consecutive rounds are disjoint from the ones still being read, so one
  // barrier per round.
  // Round 4: once at most 64 additions are left in a round, FOUR lanes share each of them (curveu.hpp: xyzzr_add_quad): the
  // last seven rounds are a chain of seven additions on an otherwise idle workgroup, and a quad's addition is three products +
  // one double product deep instead of fourteen.  J.quad == 0 keeps the one-lane rounds (the comparison).
  const uint32_t quad_from = J.quad ? 128u : 0u;   // the value of s after which the quads take over
  for (uint32_t s = 256;;) {
    xyzzr_add(acc, other);   // (its results keep the invariants its operands need: the running-sum levels chain it the same way)
    s >>= 1;
    if (s == 0 || s < quad_from) break;
    if (threadIdx.x >= s && threadIdx.x < 2 * s) sh[threadIdx.x] = acc;
    __syncthreads();
    other = threadIdx.x < s ? sh[threadIdx.x + s] : U::zero();
  }
  {
    if (quad_from) {
      // lanes 0 .. 127 hold the sums of the round s = 128; quad q = lane / 4 continues element q
      if (threadIdx.x < 128) sh[threadIdx.x] = acc;
      __syncthreads();
      const uint32_t q = threadIdx.x >> 2, role = threadIdx.x & 3u;
      acc = sh[q];
      for (uint32_t s = 64; s > 0; s >>= 1) {
        if (q < s) {                       // (uniform over a quad)
          acc = xyzzr_add_quad(acc, sh[q + s], role);
          if (role == 0) sh[q] = acc;      // read by quad q - s/2 in the next round (when q >= s/2)
        }
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0) store_vec(J.out[job] + ((uint64_t)w * J.out_w[job] + rp) * slices + slice, xyzzr_store(acc));
}

// Table mode, error path only: the accumulation reports the lowest TABLE index that held the identity, which orders by window
// first; the reference reports the lowest EXPONENT (source.rs:50-52).  One pass over the exponents and their bases (the table's first
// window) finds it: the lowest base index that is the identity under a selected, non-zero exponent.
template <class F>
__global__ void __launch_bounds__(256) msm_identity_scan_kernel(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ scalars, uint64_t n,
                                                               uint64_t base_offset, const uint32_t* __restrict__ density,
                                                               const uint32_t* __restrict__ dprefix, unsigned long long* __restrict__ err_base) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t bi = base_offset + i;
  if (density != nullptr) {
    const uint32_t wd = density[i >> 5];
    if (!((wd >> (i & 31)) & 1u)) return;
    bi = base_offset + dprefix[i >> 5] + __popc(wd & ((1u << (i & 31)) - 1u));
  }
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
  const uint4 s0 = sp[0], s1 = sp[1];
  if ((s0.x | s0.y | s0.z | s0.w | s1.x | s1.y | s1.z | s1.w) == 0) return;
  if (load_affine(bases + bi).y.is_zero()) atomicMin(err_base, (unsigned long long)bi);
}

// ------------------------------------------------------------------------------------------------
// host side

struct Workspace {
  void* p = nullptr;
  size_t bytes = 0;
  std::mutex mu;                    // serialises the MSM calls that share this device's workspace
};
// per device: a process may drive several GPUs (mi355zk_init with n_devices > 1 runs one cell of a multiexp per device, each from
// its own host thread), and calls on different devices must not wait for each other
std::mutex g_ws_reg_mu;             // guards the map only
std::map<int, Workspace*> g_ws;
Workspace& ws_of(int dev) {
  std::lock_guard<std::mutex> lk(g_ws_reg_mu);
  Workspace*& w = g_ws[dev];
  if (w == nullptr) w = new Workspace();
  return *w;
}

int ws_reserve(Workspace& w, size_t bytes, void** out) {  // under w.mu
  if (w.bytes < bytes) {
    if (w.p) ZK_HIP(hipFree(w.p));
    w.p = nullptr;
    w.bytes = 0;
    ZK_HIP(hipMalloc(&w.p, bytes));
    w.bytes = bytes;
  }
  *out = w.p;
  return 0;
}

// Calls whose workspace is small (<= WS_SMALL: up to ~2^21 points) do not share the device-wide workspace and its lock: each
// leases a buffer from a pool for its duration.  The prover queues eight multiexps from eight threads (prover.rs:250-298) -- the
// short ones (inputs, B_G1 ...) then run concurrently on their callers' streams instead of waiting behind the long ones.  (A pool
// rather than a buffer per thread: callers come and go -- a thread pool per proof -- and their buffers must not pile up.)
constexpr size_t WS_SMALL = (size_t)1 << 30;
struct SmallWs {
  int dev = -1;
  void* p = nullptr;
  size_t bytes = 0;
  bool busy = false;
};
std::mutex g_tws_mu;
std::vector<SmallWs*> g_tws;  // the pool: as many entries as there have been concurrent calls

// Pinned host buffers for the one copy that ends a multiexp (the window sums and the error words): into pageable memory the runtime
// stages each copy (~20 us apiece at this size, twice per call); leased per call from a pool like the workspaces, grow-only.
struct PinBuf {
  int dev = -1;              // the device that was current when the buffer was allocated (a process may drive several)
  void* p = nullptr;
  size_t bytes = 0;
  bool busy = false;
};
std::mutex g_pin_mu;
std::vector<PinBuf*> g_pin;
struct PinLease {
  PinBuf* b = nullptr;
  ~PinLease() {
    if (b == nullptr) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    b->busy = false;
  }
};
int pin_acquire(int dev, size_t bytes, PinLease* lease) {
  PinBuf* pick = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinBuf* b : g_pin)
      if (!b->busy && b->dev == dev && (pick == nullptr || b->bytes > pick->bytes)) pick = b;
    if (pick == nullptr) {
      pick = new PinBuf();
      pick->dev = dev;
      g_pin.push_back(pick);
    }
    pick->busy = true;
  }
  lease->b = pick;
  if (pick->bytes < bytes) {
    if (pick->p) (void)hipHostFree(pick->p);
    pick->p = nullptr;
    pick->bytes = 0;
    size_t want = 65536;
    while (want < bytes) want <<= 1;
    ZK_HIP(hipHostMalloc(&pick->p, want, hipHostMallocDefault));
    pick->bytes = want;
  }
  return ZK_OK;
}
struct SmallWsLease {
  SmallWs* w = nullptr;
  hipStream_t st = nullptr;
  bool idle = false;  // set once the caller has synchronised the stream after its last use of the buffer
  ~SmallWsLease() {
    if (w == nullptr) return;
    if (!idle) (void)hipStreamSynchronize(st);  // an error path: kernels using the buffer may still be queued
    std::lock_guard<std::mutex> lk(g_tws_mu);
    w->busy = false;
  }
};
// Buffers come in power-of-two sizes (>= 16 MiB) and are never regrown: hipFree / hipMalloc synchronise the device, and eight
// concurrent calls of eight different sizes would otherwise keep trading buffers.  Idle buffers are only given back when the pool
// exceeds TWS_POOL_CAP.
constexpr size_t TWS_POOL_CAP = (size_t)12 << 30;
int tws_acquire(int dev, size_t bytes, hipStream_t st, SmallWsLease* lease, void** out) {
  size_t cls = (size_t)16 << 20;
  while (cls < bytes) cls <<= 1;
  SmallWs* pick = nullptr;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lk(g_tws_mu);
    size_t pool = 0;
    for (SmallWs* w : g_tws) {  // the smallest idle buffer that fits
      pool += w->bytes;
      if (w->busy || w->dev != dev || w->p == nullptr || w->bytes < cls) continue;
      if (pick == nullptr || w->bytes < pick->bytes) pick = w;
    }
    if (pick == nullptr) {
      for (SmallWs* w : g_tws) {  // an empty slot, and room under the cap
        if (w->busy) continue;
        if (w->p == nullptr) { if (pick == nullptr) pick = w; continue; }
        if (pool + cls > TWS_POOL_CAP && w->dev == dev) {
          drop.push_back(w->p);
          pool -= w->bytes;
          w->p = nullptr;
          w->bytes = 0;
          if (pick == nullptr) pick = w;
        }
      }
      if (pick == nullptr) {
        pick = new SmallWs();
        g_tws.push_back(pick);
      }
      pick->dev = dev;
    }
    pick->busy = true;
  }
  lease->w = pick;
  lease->st = st;
  lease->idle = true;  // nothing queued on it yet
  for (void* d : drop) (void)hipFree(d);  // idle: their last users synchronised before releasing them
  void* p = pick->p;   // (ours: busy was set under the lock)
  if (p == nullptr) {
    ZK_HIP(hipMalloc(&p, cls));
    // published under the lock: another thread's scan sums `bytes` over ALL slots, busy ones included (r6: ThreadSanitizer on the GPU box
    // reported this write against that read, profiles/r06_tsan.txt -- the only report inside this library)
    std::lock_guard<std::mutex> lk(g_tws_mu);
    pick->p = p;
    pick->bytes = cls;
  }
  lease->idle = false;
  *out = p;
  return 0;
}

void ws_release_all() {
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (PinBuf* b : g_pin) {
      if (b->busy) continue;
      (void)hipSetDevice(b->dev);
      if (b->p) (void)hipHostFree(b->p);
      b->p = nullptr;
      b->bytes = 0;
    }
  }
  {
    std::lock_guard<std::mutex> lk(g_tws_mu);
    for (SmallWs* t : g_tws) {
      if (t->p) {
        (void)hipSetDevice(t->dev);
        (void)hipFree(t->p);
      }
      t->p = nullptr;
      t->bytes = 0;
    }
  }
  std::lock_guard<std::mutex> lk(g_ws_reg_mu);
  for (auto& kv : g_ws) {
    std::lock_guard<std::mutex> wl(kv.second->mu);
    (void)hipSetDevice(kv.first);
    (void)hipFree(kv.second->p);
    kv.second->p = nullptr;
    kv.second->bytes = 0;
  }
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Window size.  The reference uses c = ceil(ln n) (multiexp.rs:341-345), sized for its
// one-thread-per-window scan.  Here every bucket of every window is a lane, so c trades
// W*n mixed adds (10 mul each) against W*2^(c-1) buckets to reduce (~35 mul each) while keeping
// enough buckets to fill 256 CUs.  Override: env MI355ZK_MSM_C.
// widths for a maximum window size c: top window c-1 bits (unsigned), the rest as even as possible, all <= c
MsmGeom make_geom(uint32_t c) {
  MsmGeom G{};
  G.rmul = 1;
  G.c = c;
  uint32_t W = 1;
  while ((W - 1) * c + (c - 1) < 254) ++W;      // smallest W with (W-1) windows of <= c bits + a top window of <= c-1 bits
  G.W = W;
  G.nb = 1u << (c - 1);
  uint32_t top = c - 1;
  if (W == 1) top = 254 < top ? 254 : top;
  uint32_t rest = 254 > top ? 254 - top : 0;    // bits for windows 0..W-2
  uint32_t base = W > 1 ? rest / (W - 1) : 0, rem = W > 1 ? rest % (W - 1) : 0;
  uint32_t bit = 0;
  for (uint32_t w = 0; w + 1 < W; ++w) {
    G.width[w] = (uint8_t)(base + (w < rem ? 1 : 0));
    G.shift[w] = (uint8_t)bit;
    bit += G.width[w];
  }
  G.width[W - 1] = (uint8_t)(254 - bit);        // == top when rest was spread exactly (always: base*(W-1)+rem == rest)
  G.shift[W - 1] = (uint8_t)bit;
  return G;
}

// mixed-radix layout: B = rmul * 2^rshift, nb = B/2, the smallest W with B^(W-1) * nb >= 2^254 (scalars are < r < 2^254)
MsmGeom make_geom_radix(uint32_t rmul, uint32_t rshift) {
  MsmGeom G{};
  G.rmul = rmul;
  G.rshift = rshift;
  const double B = std::ldexp((double)rmul, (int)rshift);
  G.nb = (rmul << rshift) / 2;
  G.c = 1;
  while ((1u << G.c) <= G.nb) ++G.c;           // field values 0..nb
  uint32_t W = 2;
  while ((W - 1) * std::log2(B) + std::log2((double)G.nb) < 254.001) ++W;
  G.W = W;
  return G;
}

// top_values: the number of digit values the TOP window can take (2^254 / B^(W-1), or 2^width): a layout whose top window is
// much narrower than the others sends n / top_values points into each of its buckets.
double geom_cost(double W, double nbk, double field_bits, uint64_t n, double top_values) {
  // per (point, window): one mixed add (10 units) + one radix-sort pass per 8 key bits (0.7 units each,
  // measured); per bucket: ~45 units of reduction
  double cost = W * ((10.0 + 0.7 * std::ceil(field_bits / 8.0)) * (double)n + 45.0 * nbk);
  // occupancy term: fewer than ~2^17 bucket lanes leaves CUs idle during accumulation
  double lanes = W * nbk;
  if (lanes < 131072.0) cost *= (1.0 + 0.5 * (131072.0 / lanes - 1.0));
  // A top window whose buckets pass the heavy threshold (msm_device: max(64, 2 * mean + 16) for a short call) takes the
  // segment-parallel path IN FRONT of the accumulation: ~0.18 ms of a 1-ms call at 2^16 points (B = 13 * 2^10: 169 top values, 388
  // points per top bucket: profiles/r03_msm16_timeline.txt).  A constant the size of that detour: decisive for short calls, nothing
  // at 2^20 and beyond (where the segments also run at throughput).
  // Below that threshold a lane still walks the top bucket alone, ~8 us per point whatever the rest of the launch does (2^15
  // points, B = 3 * 2^12: 46 points per top bucket among buckets of 5: the launch lasts 0.44 ms instead of 0.2): ~5e5 units per
  // point a top bucket holds beyond what the ordinary buckets' tail reaches anyway.
  const double mean = (double)n / nbk;
  const double heavy = std::max(64.0, 2.0 * mean + 16.0);
  const double top_len = (double)n / top_values;
  if (top_len > heavy) cost += 1.2e7;
  else if (top_len > 2.0 * mean + 16.0) cost += (top_len - (2.0 * mean + 16.0)) * 5e5;
  return cost;
}

MsmGeom choose_geom(uint64_t n, int group, uint32_t wgroups = 1) {
  static const char* env = std::getenv("MI355ZK_MSM_C");
  static const char* env_radix = std::getenv("MI355ZK_MSM_RADIX");  // "0": power-of-two layouts only; "m,s": force B = m * 2^s (m odd, 3..15)
  (void)group;
  if (env_radix && wgroups == 1) {
    int rm = 0, rs = 0;
    if (std::sscanf(env_radix, "%d,%d", &rm, &rs) == 2 && rm >= 3 && rm <= 15 && (rm & 1) && rs >= 2 && rs <= 22) return make_geom_radix((uint32_t)rm, (uint32_t)rs);
  }
  if (env && wgroups == 1) {
    int v = std::atoi(env);
    if (v >= 2 && v <= 24) return make_geom((uint32_t)v);
  }
  // Short calls (n < 2^20, all windows on one GPU): MEASURED choice.  The launch no longer fills the device there and the model
  // below -- throughput of additions, 45 units per bucket -- misses what a call costs: the reduce is a chain of dependent additions
  // whose length depends on c alone (c = 10: 0.12 ms ... 16: 0.40, 17: 0.62) and the accumulation reaches its throughput only from
  // ~2^19 bucket lanes on.  profiles/r03_small_n_window_sweep.txt: every c at every size; the table is its minimum (8 - 15 % per call
  // against the model's choice; power-of-two windows, whose top window is as wide as the others).
  if (wgroups == 1 && n < (1ull << 20)) {
    uint32_t lg = 0;
    while ((1ull << lg) < n) ++lg;
    uint32_t c = lg <= 10 ? 10u : lg <= 12 ? 11u : lg == 13 ? 12u : lg <= 15 ? 13u : lg <= 17 ? 15u : 16u;
    // G2 (profiles/r03_small_n_window_sweep.txt, second half): an addition costs three times G1's and so does every step of the
    // reduce chain -- one bit narrower between 2^14 and 2^18 (2^14: 1.43 -> 1.33 ms, 2^16: 1.71 -> 1.61, 2^18: 2.54 -> 2.49)
    if (group == 2) c = lg <= 10 ? 10u : lg == 11 ? 11u : lg <= 15 ? 12u : lg <= 17 ? 14u : lg == 18 ? 15u : 16u;
    // (end of round 4, with the pair-per-bucket accumulation: profiles/r04_small_n_sweep_pair.txt -- G1 2^15: c = 15 -> 13, 0.475 -> 0.45 ms;
    // G2 2^15: 14 -> 12, 1.08 -> 0.99 ms; G2 2^17: 15 -> 14, 1.52 -> 1.46 ms; every other entry stayed the minimum)
    return make_geom(c);
  }
  // wgroups > 1: the windows are dealt out to that many ranks, so W must divide evenly (per-rank cost ~ total / wgroups)
  uint32_t best_c = 0;
  double best = 1e300;
  for (uint32_t c = 4; c <= 24; ++c) {
    double W = std::ceil((254.0 + 1.0) / c);  // (W-1)*c + (c-1) >= 254
    if ((uint32_t)W % wgroups) continue;
    // (make_geom: the top window keeps what the W - 1 equal windows leave of the 254 bits, at most c - 1)
    const MsmGeom Gc = make_geom(c);
    double cost = geom_cost(W, std::ldexp(1.0, (int)c - 1), c, n, std::ldexp(1.0, (int)Gc.width[Gc.W - 1]));
    if (cost < best) { best = cost; best_c = c; }
  }
  MsmGeom G{};
  if (best_c) G = make_geom(best_c);
  if (env_radix && env_radix[0] == '0' && best_c) return G;
  // a mixed-radix layout must win by 1.5 % to be taken (its host join is slightly longer)
  for (uint32_t rmul = 3; rmul <= 15; rmul += 2)
    for (uint32_t rshift = 4; rshift <= 22; ++rshift) {
      MsmGeom R = make_geom_radix(rmul, rshift);
      if (R.W > 64 || R.c > 24 || R.W % wgroups) continue;
      const double top_values = std::exp2(254.0 - (R.W - 1.0) * std::log2(std::ldexp((double)rmul, (int)rshift)));
      double cost = geom_cost(R.W, R.nb, R.c, n, top_values);
      if (cost < 0.985 * best) { best = cost / 0.985; G = R; }
    }
  return G;
}

// Window width of a TABLE-MODE call (msm_device, table_stride != 0) over a base vector of n_bases points: one bucket set serves all
// windows, so the reduction is paid once and the window may be wider than choose_geom's (fewer windows = fewer additions); what
// limits it is the length of the one reduce chain and the bucket lists getting short.  Measured (profiles/r03_table_mode.txt);
// override: env MI355ZK_MSM_TABLE_C.  Power-of-two windows only: table[w] = 2^shift_w * P.
uint32_t table_window_bits(uint64_t n_bases, int group) {
  static const char* env = std::getenv("MI355ZK_MSM_TABLE_C");
  if (env) {
    int v = std::atoi(env);
    if (v >= 4 && v <= 24) return (uint32_t)v;
  }
  uint32_t lg = 0;
  while ((1ull << lg) < n_bases) ++lg;
  // Only the smallest c of every window count matters (17: 15 windows, 19: 14, 20: 13, 22: 12, 24: 11).  Every bucket of the one set
  // is populated (the top window's unsigned digits reach all of them), so the reduce runs at its full length: c = 22 costs 1.0 ms,
  // 23: 1.6, 24: 2.8.  Short vectors need bucket LANES before anything else (2^16 points at the plain call's c = 15: 16 k lanes, 1.18 ms
  // against 0.72 at c = 17) -- and gain nothing over the plain call below 2^19.
  if (group == 2) return lg <= 18 ? 17u : lg == 19 ? 19u : 20u;   // (a G2 reduce step costs three G1 steps: 15 - 25 % over the plain call from 2^16 on)
  return lg <= 16 ? 17u : lg <= 22 ? 20u : 22u;
}

// the partition kernels use up to the whole 160 KiB of LDS (dynamic): raise the limit once per device
std::mutex g_part_cfg_mu;
std::map<std::pair<int, int>, int> g_part_cfg;
template <class F>
int part_configure(int dev) {
  std::lock_guard<std::mutex> lk(g_part_cfg_mu);
  const std::pair<int, int> key(dev, (int)sizeof(F));
  auto it = g_part_cfg.find(key);
  if (it != g_part_cfg.end()) return it->second;
  int rc = ZK_OK;
  // (the tree kernel keeps 256 register-form sums in LDS: 72 KiB for G2, above the 64 KiB a kernel gets without asking)
  std::vector<const void*> fns = {reinterpret_cast<const void*>(msm_scatter_kernel), reinterpret_cast<const void*>(msm_bucket_kernel),
                                  reinterpret_cast<const void*>(msm_bigbin_place_kernel), reinterpret_cast<const void*>(msm_tree_kernel<F>)};
  for (uint32_t rmul : {1u, 3u, 5u, 7u, 9u, 11u, 13u, 15u}) ZK_DISPATCH_RMUL(rmul, fns.push_back(reinterpret_cast<const void*>(msm_digits_hist_kernel<RM>)));
  for (const void* fn : fns) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      std::fprintf(stderr, "[mi355zk] hipFuncSetAttribute(partition kernel, 160 KiB LDS) failed: %s\n", hipGetErrorString(e));
      rc = ZK_ERR_DEVICE;
    }
  }
  g_part_cfg[key] = rc;
  return rc;
}

template <class F>
int msm_device(const Affine<F>* d_bases, uint64_t n_bases, uint64_t base_offset, const uint32_t* d_scalars, uint64_t n,
               const uint32_t* d_density, const uint32_t* d_dprefix, hipStream_t st, Jacobian<F>* out, long long* err_index_out,
               bool dense = false, const Affine<F>* d_bases2 = nullptr, Jacobian<F>* out2 = nullptr, uint32_t wgroups = 1,
               uint32_t wgroup = 0, bool scalars_mont = false, MsmChunks* chunks = nullptr, uint64_t table_stride = 0, uint32_t table_c = 0) {
  // table_stride != 0: TABLE MODE.  d_bases is a window table of the base vector -- table[w * table_stride + i] = 2^shift_w * bases[i]
  // for the windows of make_geom(table_c) (msm_table_build) -- so a digit of ANY window goes to the bucket of its value in ONE
  // bucket set shared by all windows: the W key planes the digit kernel writes are read as ONE array of W * kstride (digit, table
  // index) pairs, partitioned as a single window, accumulated into 2^(c-1) buckets and reduced once; the join is the window sum
  // itself.  What it buys: one window's reduction instead of W, and with it a wider window (fewer additions) on short calls.
  // wgroups > 1: only window group `wgroup` of `wgroups` equal groups is evaluated -- the partial  sum_{w in group} B^w T_w
  // of this point set; the partials of all groups (and of all point ranges) add up to the multiexp (shard.py).
  // dense == true: powersoftau's dense_multiexp contract (infinity bases add nothing, no Source errors);
  // d_bases2 != nullptr: a second base vector evaluated with the SAME exponents (merge_pairs), sharing the
  // digit extraction and the sorts.
  // chunks != nullptr: a STREAMED multiexp (the host-buffer entry point uploads the exponents while the kernels run).  The
  // exponents [cuts[c], cuts[c+1]) of chunk c are taken from the pointer chunks->acquire(c) hands out; geometry and bucket array
  // are those of the WHOLE call: every chunk runs digits -> partition -> accumulate into the SAME buckets (the first chunk
  // writes them, the others carry them on), and the reduction and the join run once.  d_scalars is ignored, d_density /
  // d_dprefix cover the whole call.
  *out = Jacobian<F>::zero();
  if (out2) *out2 = Jacobian<F>::zero();
  *err_index_out = -1;
  if (n == 0) return ZK_OK;
  if (n_bases > 0x7fffffffull || n > 0x7fffffffull) return ZK_ERR_BAD_ARGS;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  struct Inflight {
    std::atomic<int>& c;
    explicit Inflight(std::atomic<int>& x) : c(x) { c.fetch_add(1, std::memory_order_relaxed); }
    ~Inflight() { c.fetch_sub(1, std::memory_order_relaxed); }
  } inflight(g_msm_inflight[dev & 15]);
  if (wgroups == 0 || wgroup >= wgroups) return ZK_ERR_BAD_ARGS;
  const bool tmode = table_stride != 0;
  if (tmode && (wgroups != 1 || (chunks != nullptr && chunks->n_chunks != 1) || d_bases2 != nullptr || table_c < 4 || table_c > 24 || base_offset > table_stride)) return ZK_ERR_BAD_ARGS;
  const uint64_t whole[2] = {0, n};
  const uint32_t n_chunks = chunks ? chunks->n_chunks : 1u;
  const uint64_t* cuts = chunks ? chunks->cuts : whole;
  if (n_chunks == 0 || cuts[0] != 0 || cuts[n_chunks] != n) return ZK_ERR_BAD_ARGS;
  if (n_chunks > 1 && d_bases2 != nullptr) return ZK_ERR_BAD_ARGS;
  uint64_t n_max = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    if (cuts[c + 1] <= cuts[c] || (c > 0 && (cuts[c] & 31))) return ZK_ERR_BAD_ARGS;  // (density words are not shared between chunks)
    if (cuts[c + 1] - cuts[c] > n_max) n_max = cuts[c + 1] - cuts[c];
  }
  const MsmGeom G = tmode ? make_geom(table_c) : choose_geom(n, (int)(sizeof(F) / sizeof(Fq)), wgroups);
  if (G.W == 0 || G.W % wgroups) return ZK_ERR_BAD_ARGS;
  const uint32_t WD = G.W / wgroups, w_lo = wgroup * WD, w_hi = w_lo + WD;  // this call's windows (digit planes)
  const uint32_t WL = tmode ? 1u : WD;                                      // bucket sets: one per window, or ONE in table mode
  if (tmode && (uint64_t)G.W * table_stride > 0x7fffffffull) return ZK_ERR_BAD_ARGS;  // (an index-list entry is a 31-bit base index + sign)
  const uint64_t m_max = tmode ? ((n_max + 3) & ~3ull) * WD : n_max * WL;
  if (m_max > 0xfffffff0ull) return ZK_ERR_BAD_ARGS;  // pair positions are u32
  const uint32_t n_buckets = WL * G.nb;
  // reduction: running-sum levels (msm_reduce_level_kernel, chunk length L) while more than MSM_FINAL_MAX
  // elements per window are left, then the bit-decomposition stage (msm_tree_kernel)
  // chunk length per level: 2L serial additions per lane, so shorter chunks once lanes are scarce
  uint32_t lvl_cnt[MSM_MAX_LEVELS + 1], lvl_chunks[MSM_MAX_LEVELS + 1], lvl_logl[MSM_MAX_LEVELS + 1], n_levels = 0;
  uint64_t total_chunks = 1;
  uint32_t final_cnt = G.nb;
  const char* env_fmax = std::getenv("MI355ZK_MSM_FINAL_MAX");
  const uint32_t final_max = env_fmax && std::atoi(env_fmax) >= 64 ? (uint32_t)std::atoi(env_fmax) : MSM_FINAL_MAX;
  // 2-D TAIL (round 3): once at most 2^17 elements are left over all windows (and at most 2^18 per window), the weighted sum of the
  // last array S is finished in TWO tree launches instead of further levels and a bit decomposition of what they leave:
  //   x = r * cols + c:   sum_x (x + off) S[x] = cols * sum_r r R_r + sum_c (c + off) C_c,   R_r / C_c = the row / column sums,
  // launch 1 = the rows, the columns (<= 512 elements each) and the levels' A[] as plain tree sums, launch 2 = the bit decompositions of
  // R and C (and the A[] slice sums).  A running-sum level costs 2L dependent additions and a launch however few lanes it has left;
  // the tail is a chain: table mode (ONE window of 2^19 buckets) 0.545 -> 0.424 ms at 2^20.
  // env MI355ZK_MSM_NO_TAIL2D restores the levels-to-1024 schedule for the comparison.
  static const bool no_tail2d = std::getenv("MI355ZK_MSM_NO_TAIL2D") != nullptr;
  while (final_cnt > final_max && n_levels < MSM_MAX_LEVELS) {
    // (rows and columns of at least 128 elements: a tree workgroup spends nine rounds on its slice however few elements it holds, so
    // 16 windows x 8192 elements as 64 x 128 made the 2^20 reduce SLOWER, 0.40 -> 0.71 ms; G1 only: a G2 tree round costs three G1
    // rounds and the two launches gained nothing over the levels, 1.36 -> 1.38 ms)
    if (!no_tail2d && sizeof(F) == sizeof(Fq) && (uint64_t)final_cnt * WL <= (1ull << 17) && final_cnt >= (1u << 15) && final_cnt <= (1u << 18)) break;
    uint32_t logl = (uint64_t)final_cnt * WL >= (1ull << 20) ? 3 : 2;
    {
      // (experiments: MI355ZK_MSM_LOGL = "3,2,2" forces the chunk length 2^k of the first levels -- read per call)
      const char* env_l = std::getenv("MI355ZK_MSM_LOGL");
      if (env_l) {
        uint32_t k = 0;
        const char* q = env_l;
        while (k < n_levels && *q) { if (*q == ',') ++k; ++q; }
        if (k == n_levels && *q >= '1' && *q <= '5') logl = (uint32_t)(*q - '0');
      }
    }
    lvl_cnt[n_levels] = final_cnt;
    lvl_logl[n_levels] = logl;
    lvl_chunks[n_levels] = (final_cnt + (1u << logl) - 1) >> logl;
    total_chunks += lvl_chunks[n_levels];
    final_cnt = lvl_chunks[n_levels];
    ++n_levels;
  }
  const uint32_t final_off = n_levels == 0 ? 1u : 0u;  // bucket x of a window has weight x + 1; chunk sums have weight ch
  const bool tail2d = !no_tail2d && sizeof(F) == sizeof(Fq) && final_cnt >= (1u << 15) && final_cnt <= (1u << 18) && (uint64_t)final_cnt * WL <= (1ull << 17);
  uint32_t cols_log = 0;
  if (tail2d) {
    uint32_t lg = 0;
    while ((1u << lg) < final_cnt) ++lg;
    cols_log = (lg + 1) / 2;
  }
  const uint32_t t_cols = tail2d ? 1u << cols_log : final_cnt, t_rows = tail2d ? (final_cnt + t_cols - 1) / t_cols : 0;
  uint32_t final_bits = 1;                                   // bits of the column index (of the whole index without the 2-D tail)
  while ((1u << final_bits) <= t_cols - 1 + final_off) ++final_bits;
  uint32_t row_bits = 0;                                     // bits of the row index
  while (t_rows > 1 && (1u << row_bits) <= t_rows - 1) ++row_bits;

  // per chunk: its partition geometry, and what one lane may walk before its bucket counts as heavy
  struct ChunkPlan {
    uint64_t lo, n, m, np;  // np: length of the array the partition sees (table mode: all key planes as one)
    PartGeom P;
    uint32_t ncell, heavy, heavy_seg, hb, max_items;
    bool small_scan;              // short calls: the four column scans + the big-bin plan in one single-workgroup launch
    uint32_t split_t, split_hb;   // short calls: buckets longer than split_t take the quad-per-bucket launch (0: none)
  };
  std::vector<ChunkPlan> plan(n_chunks);
  uint64_t keys_cap = 0, tile_hist_b = 0, tile_off_b = 0, csum_b = 0;
  uint32_t ncell_max = 0, hb_max = 0, items_max = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    ChunkPlan& C = plan[c];
    C.lo = cuts[c];
    C.n = cuts[c + 1] - cuts[c];
    C.np = tmode ? ((C.n + 3) & ~3ull) * WD : C.n;
    C.m = C.np * WL;
    C.P = choose_part(C.np, WL, G.nb);
    if (C.P.st == 0) return ZK_ERR_BAD_ARGS;
    C.ncell = WL * C.P.nbin;
    {
      static const bool no_small = std::getenv("MI355ZK_MSM_NO_SMALL_SCAN") != nullptr;
      C.small_scan = !no_small && (uint64_t)C.P.n_st * C.ncell <= (1u << 16);
    }
    // index lists: every bucket start is padded to a multiple of 4 entries (<= 3 per bucket), every bin region to 4
    const uint64_t vals_cap = C.m + 3ull * n_buckets + 4ull * C.ncell + 4;
    if (vals_cap > 0xfffffff0ull) return ZK_ERR_BAD_ARGS;
    keys_cap = std::max(keys_cap, vals_cap);
    tile_hist_b = std::max<uint64_t>(tile_hist_b, (uint64_t)C.P.n_st * ((C.ncell + 1) & ~1u) * 2);
    tile_off_b = std::max<uint64_t>(tile_off_b, (uint64_t)C.P.n_st * C.ncell * 4);
    csum_b = std::max<uint64_t>(csum_b, (uint64_t)C.P.n_chunk * C.ncell * 4);
    ncell_max = std::max(ncell_max, C.ncell);
    // a bucket is "heavy" when it is far longer than the mean; at most m / heavy buckets can be
    // ... and, more to the point, when ONE lane walking it would outlast the whole launch: the lanes of a launch share ~2^18 lane
    // slots (256 CUs x 4 SIMDs x 4 waves x 64), so a launch lasts about m / 2^18 additions per slot; a longer bucket is a straggler
    // (it starts first -- buckets run in size order -- but finishes alone).  Prover-like exponents produce such buckets by the
    // hundred (every byte-sized witness value lands in one of 255 buckets of window 0).
    // Short calls are latency-bound instead: a lane adds a point to its bucket every ~8 us whatever else the device does (ten
    // dependent field products), so a bucket of 100 entries among buckets of 6 holds the launch for 0.8 ms (measured at 2^18
    // prover-like exponents: the 255 byte-valued buckets of window 0).  Hence a floor of 64, a margin of 16 over twice the mean,
    // and segments short enough (heavy_seg) that a segment's 64 lanes add a handful of points each before the tree.
    const uint64_t mean_len = C.m / n_buckets + 1;
    uint64_t heavy64 = mean_len * 8 + 1024;
    const uint64_t heavy_cap = (C.m >> 17) > 64 ? (C.m >> 17) : 64;  // 8 us per entry against ~2^-17 x m x 8 us for the launch at full throughput
    if (heavy64 > heavy_cap) heavy64 = heavy_cap;
    if (heavy64 < 2 * mean_len + 16) heavy64 = 2 * mean_len + 16;   // never the ordinary buckets
    C.heavy = (uint32_t)(heavy64 > 0xffffffffull ? 0xffffffffull : heavy64);
    C.heavy_seg = 128;
    while (C.heavy_seg < MSM_HEAVY_SEG && ((uint64_t)C.heavy_seg << 14) < C.m) C.heavy_seg <<= 1;
    C.hb = n_buckets < MSM_HEAVY_BLOCKS ? n_buckets : MSM_HEAVY_BLOCKS;
    if ((uint64_t)C.hb > C.m / C.heavy + 1) C.hb = (uint32_t)(C.m / C.heavy + 1);
    C.max_items = (uint32_t)(C.m / C.heavy_seg) + C.hb;  // every heavy bucket adds at most one partial segment
    hb_max = std::max(hb_max, C.hb);
    items_max = std::max(items_max, C.max_items);
    // Quad-per-bucket launch for the long buckets of a SHORT, unchunked call (msm_accumulate_split_kernel): while the lane-per-bucket
    // launch fits the device about once (<= 2^18 bucket lanes), its duration is its longest bucket.  Threshold: the mean length plus
    // one standard deviation of a Poisson count (~10 % of the buckets of uniform exponents), at least 4 (a lane per entry of a quad).
    // The two launches run one after the other (same stream): what is gained is the difference between the long buckets' chains.
    // env MI355ZK_MSM_SPLIT=0 disables, =t forces the threshold.
    C.split_t = 0;
    C.split_hb = 0;
    {
      const char* env_split = std::getenv("MI355ZK_MSM_SPLIT");
      const bool off = env_split && env_split[0] == '0' && env_split[1] == 0;
      // (measured, tools/ab_split.sh: G1 2^10 .. 2^14 points -4 .. -9 % per call, nothing at 2^15 .. 2^17, +3 .. 5 % from 2^18 on; G2: -2 % at 2^12, +2 % at 2^16)
      if (!off && n_chunks == 1 && n_buckets <= (sizeof(F) == sizeof(Fq) ? 1u << 18 : 1u << 16) && !tmode) {
        const double mean = (double)C.m / n_buckets;
        uint32_t t = (uint32_t)std::ceil(mean + std::sqrt(mean + 1.0));
        if (env_split && std::atoi(env_split) > 0) t = (uint32_t)std::atoi(env_split);
        if (t < 4) t = 4;
        if (t < C.heavy) {
          C.split_t = t;
          const uint64_t cap = C.m / (t + 1) + 1;    // buckets longer than t
          C.split_hb = (uint32_t)(cap < n_buckets ? cap : n_buckets);
        }
      }
    }
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  // the window-major keys of pass A are dead once pass B has run; pass C writes the index lists over them
  size_t o_keys = take((size_t)keys_cap * 4), o_pairs = take((size_t)m_max * 8);
  size_t o_tile_hist = take((size_t)tile_hist_b), o_tile_off = take((size_t)tile_off_b);
  size_t o_csum = take((size_t)csum_b), o_total = take((size_t)ncell_max * 4);
  size_t o_bin_start = take((size_t)(ncell_max + 1) * 4), o_out_start = take((size_t)(ncell_max + 1) * 4);
  // big bins (msm_bigbin_*): per-bucket counts and cursors (contiguous: one memset), the list of big bins, the plan
  size_t o_gcnt = take((size_t)n_buckets * 4), o_gcur = take((size_t)n_buckets * 4);
  size_t o_big_col = take((size_t)ncell_max * 4), o_big_seg = take((size_t)ncell_max * 4), o_big_plan = take(sizeof(BigPlan));
  size_t o_first = take((size_t)(n_buckets + 1) * 4), o_last = take((size_t)(n_buckets + 1) * 4), o_hist = take(MSM_SIZE_BINS * 4);
  size_t o_sizes_b = take((size_t)n_buckets * 4), o_ids_b = take((size_t)n_buckets * 4);
  size_t o_item_off = take((size_t)(hb_max + 2) * 4);
  size_t o_seg_sums = take((size_t)items_max * sizeof(XYZZ<F>));
  size_t o_buckets = take((size_t)n_buckets * sizeof(XYZZ<F>));
  size_t o_partA = take((size_t)WL * total_chunks * sizeof(XYZZ<F>));
  size_t o_partS = take((size_t)WL * total_chunks * sizeof(XYZZ<F>));
  const uint32_t n_out = n_levels + final_bits + row_bits;  // per window: one A-sum per level, then one sum per (column) bit, then one per row bit
  if (n_out + 2 > MSM_MAX_JOBS) return ZK_ERR_BAD_ARGS;
  size_t o_rc = take((size_t)WL * (t_rows + t_cols) * sizeof(XYZZ<F>));  // 2-D tail: row sums, then column sums, per window
  size_t o_wsums = take((size_t)WL * n_out * sizeof(XYZZ<F>));
  size_t o_err = take(16);  // [0] lowest identity base index, [1] lowest index of a non-canonical exponent -- right behind the window sums: ONE copy brings both back
  // slice sums of msm_tree_kernel (two ping-pong halves): n_out jobs per window, slices of the longest job
  const uint32_t tree_cnt = n_levels ? lvl_chunks[0] : final_cnt;
  const uint64_t tree_tmp = (uint64_t)n_out * ((tree_cnt + MSM_TREE_SLICE - 1) / MSM_TREE_SLICE);
  size_t o_sumtmp = take((size_t)WL * tree_tmp * 2 * sizeof(XYZZ<F>));

  int rc = part_configure<F>(dev);
  if (rc) return rc;
  const bool small_ws = off <= WS_SMALL;
  Workspace& dev_ws = ws_of(dev);
  std::unique_lock<std::mutex> lk(dev_ws.mu, std::defer_lock);
  void* base = nullptr;
  SmallWsLease lease;
  if (small_ws) {
    rc = tws_acquire(dev, off, st, &lease, &base);
  } else {
    lk.lock();
    rc = ws_reserve(dev_ws, off, &base);
  }
  if (rc) return rc;
  char* ws = (char*)base;
  uint32_t* keys = (uint32_t*)(ws + o_keys);
  uint32_t* vals_b = keys;  // (aliases the keys: see above)
  uint2* pairs = (uint2*)(ws + o_pairs);
  uint16_t* tile_hist = (uint16_t*)(ws + o_tile_hist);
  uint32_t* tile_off = (uint32_t*)(ws + o_tile_off);
  uint32_t* csum = (uint32_t*)(ws + o_csum);
  uint32_t* col_total = (uint32_t*)(ws + o_total);
  uint32_t* bin_start = (uint32_t*)(ws + o_bin_start);
  uint32_t* out_start = (uint32_t*)(ws + o_out_start);
  uint32_t* gcnt = (uint32_t*)(ws + o_gcnt);
  uint32_t* gcur = (uint32_t*)(ws + o_gcur);
  uint32_t* big_col = (uint32_t*)(ws + o_big_col);
  uint32_t* big_seg = (uint32_t*)(ws + o_big_seg);
  BigPlan* big_plan = (BigPlan*)(ws + o_big_plan);
  uint32_t* first = (uint32_t*)(ws + o_first);
  uint32_t* last = (uint32_t*)(ws + o_last);
  uint32_t* size_hist = (uint32_t*)(ws + o_hist);
  uint32_t* sizes_b = (uint32_t*)(ws + o_sizes_b);
  uint32_t* order = (uint32_t*)(ws + o_ids_b);
  uint32_t* item_off = (uint32_t*)(ws + o_item_off);
  XYZZ<F>* seg_sums = (XYZZ<F>*)(ws + o_seg_sums);
  XYZZ<F>* buckets = (XYZZ<F>*)(ws + o_buckets);
  XYZZ<F>* partA = (XYZZ<F>*)(ws + o_partA);
  XYZZ<F>* partS = (XYZZ<F>*)(ws + o_partS);
  XYZZ<F>* wsums = (XYZZ<F>*)(ws + o_wsums);
  XYZZ<F>* sumtmp = (XYZZ<F>*)(ws + o_sumtmp);
  unsigned long long* d_err = (unsigned long long*)(ws + o_err);

  ZK_HIP(hipMemsetAsync(d_err, 0xff, 16, st));
  lease.idle = false;

  static const bool debug = std::getenv("MI355ZK_DEBUG") != nullptr;
  auto checkpoint = [&](const char* what, const ChunkPlan& C) -> int {
    if (!debug) return 0;
    ZK_HIP(hipStreamSynchronize(st));
    std::fprintf(stderr, "[mi355zk] msm<%d> n=%llu chunk@%llu+%llu c=%u W=%u buckets=%u levels=%u part(lo=%u nbin=%u st=%u): %s done\n",
                 (int)(sizeof(F) / sizeof(Fq)), (unsigned long long)n, (unsigned long long)C.lo, (unsigned long long)C.n, G.c, WL, n_buckets,
                 n_levels, C.P.lo_bits, C.P.nbin, C.P.st, what);
    return 0;
  };
  static const int slot_digits = prof_slot("msm_digits"), slot_scan = prof_slot("msm_part_scan"), slot_scatter = prof_slot("msm_scatter"),
                   slot_bucket = prof_slot("msm_bucket"), slot_sort = prof_slot("msm_sort"), slot_acc = prof_slot("msm_accumulate"),
                   slot_heavy = prof_slot("msm_accumulate_heavy"), slot_red = prof_slot("msm_reduce");

  // ---- digits + partition of one chunk: index lists per (window, bucket) in vals_b, bounds first[] / last[], size order
  auto partition_chunk = [&](const ChunkPlan& C, const uint32_t* d_sc) -> int {
    const PartGeom& P = C.P;
    const uint64_t nc = C.n, kstride = (nc + 3) & ~3ull;  // distance between the key planes of two windows
    const uint32_t ncell = C.ncell;
    // density words and prefix ranks of this chunk's exponents (cuts are multiples of 32); under FullDensity exponent i of the
    // chunk owns base base_offset + lo + i
    const uint32_t* dens = d_density ? d_density + (C.lo >> 5) : nullptr;
    if (!C.small_scan) ZK_HIP(hipMemsetAsync(size_hist, 0, MSM_SIZE_BINS * 4, st));   // (msm_scan_small_kernel clears it)
    if (C.np > BIG_SEG) ZK_HIP(hipMemsetAsync(gcnt, 0, o_big_col - o_gcnt, st));  // gcnt and gcur (big bins only: see the bucket pass)
    // "msm_sort" spans the whole partition after the digits (scan + scatter + bucket + size order), as it did for the library sort
    prof_begin(slot_digits, st);
    static const bool fused_a = std::getenv("MI355ZK_PART_FUSED_A") != nullptr;  // (the one-kernel pass A, kept for the comparison in DESIGN.md)
    if (!fused_a) {
      ZK_DISPATCH_RMUL(G.rmul, hipLaunchKernelGGL(msm_digits_plain_kernel<RM>, dim3((unsigned)((kstride + 255) / 256)), dim3(256), 0, st, d_sc, nc, dens, G,
                                                  w_lo, w_hi, scalars_mont ? 1 : 0, kstride, keys, d_err + 1, C.lo));
      hipLaunchKernelGGL(msm_tile_hist_kernel, dim3(P.n_st * WL), dim3(PART_THREADS), (size_t)P.nbin * 4, st, keys, C.np, tmode ? C.np : kstride, G.nb, WL, P,
                         tile_hist);
    } else {
      if (tmode) return (int)ZK_ERR_BAD_ARGS;
      int cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      const uint32_t per_cu = (size_t)((ncell + 1) / 2) * 4 <= PART_LDS_A ? 2u : 1u;     // 1024-lane workgroups per CU (LDS histograms)
      const uint32_t grid = P.n_st < per_cu * (uint32_t)cus ? P.n_st : per_cu * (uint32_t)cus;
      ZK_DISPATCH_RMUL(G.rmul, hipLaunchKernelGGL(msm_digits_hist_kernel<RM>, dim3(grid), dim3(PART_THREADS), (size_t)((ncell + 1) / 2) * 4, st, d_sc, nc,
                                                  dens, G, w_lo, w_hi, scalars_mont ? 1 : 0, P, kstride, keys, tile_hist, d_err + 1, C.lo));
    }
    ZK_HIP(hipGetLastError());
    prof_end(slot_digits, st);
    if (checkpoint("digits", C)) return ZK_ERR_DEVICE;
    return ZK_OK;
  };
  auto partition_rest = [&](const ChunkPlan& C) -> int {
    const PartGeom& P = C.P;
    const uint64_t nc = C.n, kstride = (nc + 3) & ~3ull;
    const uint32_t ncell = C.ncell;
    const uint32_t* dens = d_density ? d_density + (C.lo >> 5) : nullptr;
    const uint32_t* dpre = d_dprefix ? d_dprefix + (C.lo >> 5) : nullptr;
    const uint64_t boff = base_offset + (d_density ? 0 : C.lo);
    prof_begin(slot_sort, st);
    prof_begin(slot_scan, st);
    if (C.small_scan) {
      hipLaunchKernelGGL(msm_scan_small_kernel, dim3(1), dim3(PART_THREADS), 0, st, tile_hist, P, ncell, G.nb, col_total, bin_start, out_start, tile_off,
                         size_hist, big_col, big_seg, big_plan);
    } else {
    hipLaunchKernelGGL(msm_colsum_kernel, dim3((ncell + 255) / 256, P.n_chunk), dim3(256), 0, st, tile_hist, P, ncell, csum);
    hipLaunchKernelGGL(msm_colscan_kernel, dim3((ncell + 255) / 256), dim3(256), 0, st, csum, P, ncell, col_total);
    hipLaunchKernelGGL(msm_binscan_kernel, dim3(1), dim3(PART_THREADS), 0, st, col_total, P, ncell, G.nb, bin_start, out_start);
    hipLaunchKernelGGL(msm_tileoff_kernel, dim3((ncell + 255) / 256, P.n_chunk), dim3(256), 0, st, tile_hist, csum, bin_start, P, ncell, tile_off);
    }
    ZK_HIP(hipGetLastError());
    prof_end(slot_scan, st);
    prof_begin(slot_scatter, st);
    {
      static const char* env_x = std::getenv("MI355ZK_PART_XCDS");  // 1 disables the XCD-aware tile order
      const uint32_t xcds = env_x && std::atoi(env_x) >= 1 ? (uint32_t)std::atoi(env_x) : 8u;
      hipLaunchKernelGGL(msm_scatter_kernel, dim3(P.n_st * WL), dim3(PART_THREADS), (size_t)(2 * ((P.nbin + 3u) & ~3u) + 32) * 4 + (size_t)P.st * 8, st,
                         keys, C.np, tmode ? C.np : kstride, boff, dens, dpre, G.nb, WL, P, xcds, tile_off, pairs, tmode ? kstride : 0ull, table_stride);
    }
    ZK_HIP(hipGetLastError());
    prof_end(slot_scatter, st);
    if (checkpoint("scatter", C)) return ZK_ERR_DEVICE;
    prof_begin(slot_bucket, st);
    {
      const uint32_t nfl = (1u << P.lo_bits) < 4 ? 4 : (1u << P.lo_bits);
      const size_t fixed = (size_t)(2 * nfl + 32) * 4;
      // staging for the expected bin population with slack, at most what the CU has
      uint64_t want = (uint64_t)(C.np / P.nbin) * 5 / 4 + 3ull * nfl + 4096;
      const uint64_t cap_max = (PART_LDS_MAX - fixed) / 4;
      if (want > cap_max) want = cap_max;
      if (want > (uint64_t)PART_EC * PART_THREADS + 3ull * nfl) want = (uint64_t)PART_EC * PART_THREADS + 3ull * nfl;
      const uint32_t stage_cap = (uint32_t)want & ~3u;
      hipLaunchKernelGGL(msm_bucket_kernel, dim3(ncell), dim3(PART_THREADS), fixed + (size_t)stage_cap * 4, st, pairs, bin_start, out_start, G.nb, P,
                         stage_cap, first, last, vals_b);
      // the big bins (none for uniform exponents up to 2^26 points: the surplus workgroups of these launches exit at once).  A
      // chunk with at most BIG_SEG elements per window cannot have one at all -- the bucket kernel took every bin -- so a short call
      // does not pay for three idle launches (~5 us each of a 0.4-ms call at 2^10 .. 2^15 points)
      if (C.np > BIG_SEG) {
      const uint32_t max_seg = (uint32_t)(2 * (C.m / BIG_SEG) + 2);
      if (!C.small_scan) hipLaunchKernelGGL(msm_bigbin_plan_kernel, dim3(1), dim3(PART_THREADS), 0, st, bin_start, ncell, big_col, big_seg, big_plan);
      // (small grids: when there is no big bin -- uniform exponents -- the launches only cost their workgroups' start-up, and the
      // place kernel's LDS allows one workgroup per CU anyway)
      int n_cu = 256;
      (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
      const uint32_t big_grid = max_seg < 2u * (uint32_t)n_cu ? max_seg : 2u * (uint32_t)n_cu;
      const uint32_t place_grid = max_seg < (uint32_t)n_cu ? max_seg : (uint32_t)n_cu;
      hipLaunchKernelGGL(msm_bigbin_count_kernel, dim3(big_grid), dim3(PART_THREADS), (size_t)nfl * 4, st, pairs, bin_start, big_plan, big_col, big_seg,
                         G.nb, P, gcnt);
      const size_t place_fixed = (size_t)(4 * nfl + 32) * 4, place_staged = place_fixed + (size_t)BIG_SEG * 4;
      const int staged = place_staged <= PART_LDS_MAX ? 1 : 0;
      hipLaunchKernelGGL(msm_bigbin_place_kernel, dim3(staged ? place_grid : big_grid), dim3(PART_THREADS), staged ? place_staged : place_fixed, st, pairs, bin_start, out_start,
                         big_plan, big_col, big_seg, G.nb, P, staged, gcnt, gcur, first, last, vals_b);
      }
    }
    ZK_HIP(hipGetLastError());
    prof_end(slot_bucket, st);
    if (checkpoint("bucket", C)) return ZK_ERR_DEVICE;
    msm_order_by_size(first, last, n_buckets, size_hist, order, sizes_b, st);
    ZK_HIP(hipGetLastError());
    prof_end(slot_sort, st);
    if (checkpoint("partition", C)) return ZK_ERR_DEVICE;
    return ZK_OK;
  };

  // ---- bucket accumulation of the partitioned chunk over one base vector; carry: the buckets continue an earlier chunk
  auto accumulate_chunk = [&](const ChunkPlan& C, const Affine<F>* bases_set, bool carry) -> int {
    prof_begin(slot_heavy, st);
    hipLaunchKernelGGL(msm_heavy_plan_kernel, dim3(1), dim3(1024), 0, st, sizes_b, C.hb, C.heavy, C.heavy_seg, item_off);
    ZK_HIP(hipGetLastError());
    // (grid-stride over the segments that exist; a short call must not pay for thousands of empty workgroups)
    uint32_t heavy_grid = (uint32_t)((C.m >> 12) < 1024 ? 1024 : (C.m >> 12) > 16384 ? 16384 : (C.m >> 12));
    if (heavy_grid > C.max_items) heavy_grid = C.max_items;
    hipLaunchKernelGGL(msm_accumulate_heavy_kernel<F>, dim3(heavy_grid), dim3(MSM_HEAVY_LANES), MSM_HEAVY_LANES * sizeof(typename BucketAcc<F>::type), st,
                       bases_set, vals_b, first, last, order, item_off, C.hb, C.heavy_seg, seg_sums, dense ? 1 : 0, d_err);
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(msm_heavy_combine_kernel<F>, dim3(C.hb < 2048 ? C.hb : 2048), dim3(64), 64 * sizeof(typename BucketAcc<F>::type), st, seg_sums, order, item_off,
                       C.hb, buckets, carry ? 1 : 0);
    ZK_HIP(hipGetLastError());
    prof_end(slot_heavy, st);
    prof_begin(slot_acc, st);
    static const bool a4 = std::getenv("MI355ZK_ACC_NARROW") == nullptr;  // (the 4-byte index walk, kept for the traffic comparison in profiles/)
    const dim3 grid((n_buckets + 255) / 256), block(256);
    // the lane-per-bucket launch leaves out what another launch does: the heavy buckets (> C.heavy, among order[0 .. C.hb)), and for a
    // short call also the long ones (> C.split_t, among order[0 .. C.split_hb)), which a quad each walks
    const bool split = C.split_t != 0 && !carry;
    const uint32_t skip_len = split ? C.split_t : C.heavy, skip_hb = split ? std::max(C.split_hb, C.hb) : C.hb;
    if (split) {
      hipLaunchKernelGGL(msm_accumulate_split_kernel<F>, dim3((4 * C.split_hb + 255) / 256), dim3(256), 0, st, bases_set, vals_b, first, last, order,
                         C.split_t, C.heavy, C.hb, C.split_hb, buckets, dense ? 1 : 0, d_err);
      ZK_HIP(hipGetLastError());
    }
    bool pair_done = false;
    {
      // A PAIR of lanes per bucket (msm_accumulate_pair_kernel / _g1_kernel) while the bucket lanes do not fill the device several times
      // over: there a launch lasts as long as its lanes' chains of dependent additions, and the pair's chain is half as long.
      // G2 (<= 3 * 2^17 buckets, i.e. <= 2^18 points): 2^16 points accumulate 0.416 -> 0.312 ms, the call 1.27 -> 1.17; 2^12: 0.845 -> 0.82;
      // 2^18: 2.09 -> 2.04.  With >= 2^19 buckets both forms run at the multiplier's rate and the pair pays its ~280 moves / selects per
      // addition (2^20: 3.37 -> 3.48 ms, 2^22: 13.1 -> 13.7): tools/ab_g2_pair.sh, profiles/r04_ab_g2_pair.txt.
      // G1 (same gate: <= 2^17 points): the pair does 1782 multiplier instructions where the lane does 1467, but the chain is 891 deep:
      // accumulate 2^12 0.097 -> 0.079 ms, 2^15 0.085 -> 0.053, 2^16 0.139 -> 0.095, 2^17 0.192 -> 0.171; 2^18 (lanes) 0.272 vs 0.305,
      // 2^20 1.07 vs 1.27: tools/ab_g1_pair.sh, profiles/r04_ab_g1_pair.txt.
      // MI355ZK_G2_PAIR / MI355ZK_G1_PAIR = 0 / 1: never / always.
      constexpr bool g2 = std::is_same<F, Fq2>::value;
      static const int pair_mode = [] { const char* s = std::getenv(g2 ? "MI355ZK_G2_PAIR" : "MI355ZK_G1_PAIR"); return !s ? -1 : s[0] == '0' ? 0 : 1; }();
      const bool pair = pair_mode < 0 ? n_buckets <= MSM_PAIR_MAX_BUCKETS : pair_mode != 0;
      if (pair) {
        const dim3 pgrid((uint32_t)((2ull * n_buckets + 255) / 256));
        auto go = [&](auto kern) {
          hipLaunchKernelGGL(kern, pgrid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets, buckets, dense ? 1 : 0, d_err);
        };
        if constexpr (g2) {
          if (carry) { if (a4) go(msm_accumulate_pair_kernel<true, true>); else go(msm_accumulate_pair_kernel<false, true>); }
          else { if (a4) go(msm_accumulate_pair_kernel<true, false>); else go(msm_accumulate_pair_kernel<false, false>); }
        } else {
          if (carry) { if (a4) go(msm_accumulate_pair_g1_kernel<true, true>); else go(msm_accumulate_pair_g1_kernel<false, true>); }
          else { if (a4) go(msm_accumulate_pair_g1_kernel<true, false>); else go(msm_accumulate_pair_g1_kernel<false, false>); }
        }
        pair_done = true;
      }
    }
    if constexpr (std::is_same<F, Fq2>::value) {
      // the one-lane G2 kernel at two waves per SIMD -- for a call that is ALONE on its device: beside the prover's other seven
      // multiexps the two-wave kernel fills every register of the SIMDs it runs on and their waves cannot share them (eight threads with
      // window tables at 2^20: 6.3 - 6.6 ms with one wave, 6.6 - 7.0 with two; alone: accumulate 3.36 -> 3.21 ms).
      // MI355ZK_G2_WAVES = 1 / 2: always one / always two.
      static const int w_mode = [] { const char* s = std::getenv("MI355ZK_G2_WAVES"); return !s ? 0 : s[0] == '1' ? 1 : 2; }();
      const bool w2 = w_mode == 0 ? g_msm_inflight[dev & 15].load(std::memory_order_relaxed) <= 1 : w_mode == 2;
      if (!pair_done && w2) {
        auto go = [&](auto kern) {
          hipLaunchKernelGGL(kern, grid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets, buckets, dense ? 1 : 0, d_err);
        };
        if (carry) { if (a4) go(msm_accumulate_g2w2_kernel<true, true>); else go(msm_accumulate_g2w2_kernel<false, true>); }
        else { if (a4) go(msm_accumulate_g2w2_kernel<true, false>); else go(msm_accumulate_g2w2_kernel<false, false>); }
        pair_done = true;
      }
    }
    if (pair_done) {
    } else if (carry) {
      if (a4)
        hipLaunchKernelGGL((msm_accumulate_kernel<F, true, true>), grid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets,
                           buckets, dense ? 1 : 0, d_err);
      else
        hipLaunchKernelGGL((msm_accumulate_kernel<F, false, true>), grid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets,
                           buckets, dense ? 1 : 0, d_err);
    } else {
      if (a4)
        hipLaunchKernelGGL((msm_accumulate_kernel<F, true, false>), grid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets,
                           buckets, dense ? 1 : 0, d_err);
      else
        hipLaunchKernelGGL((msm_accumulate_kernel<F, false, false>), grid, block, 0, st, bases_set, vals_b, first, last, order, skip_len, skip_hb, n_buckets,
                           buckets, dense ? 1 : 0, d_err);
    }
    ZK_HIP(hipGetLastError());
    prof_end(slot_acc, st);
    if (checkpoint("accumulate", C)) return (int)ZK_ERR_DEVICE;
    return ZK_OK;
  };

  const uint32_t* d_sc_first = d_scalars;
  // quad additions in the parallelism-starved parts of the reduction (G1; env MI355ZK_MSM_QUAD=0 for the comparison,
  // MI355ZK_MSM_QUAD_MAX = the largest chunk count x windows a level may have to run four lanes per chunk)
  static const char* env_quad = std::getenv("MI355ZK_MSM_QUAD");
  static const char* env_quad_max = std::getenv("MI355ZK_MSM_QUAD_MAX");
  const bool quad_tail = !(env_quad && env_quad[0] == '0');
  const uint32_t quad_max_chunks = env_quad_max ? (uint32_t)std::atoi(env_quad_max) : 65536u;
  // ---- bucket reduction, the copy back and the host join: once per base vector
  auto finish_set = [&](Jacobian<F>* result, bool last_set) -> int {
    const size_t back_bytes = (o_err - o_wsums) + 16;  // the window sums, their alignment padding, the two error words
    PinLease pin;
    if (int prc = pin_acquire(dev, back_bytes, &pin)) return prc;
    prof_begin(slot_red, st);
    {
      // wsums[w * n_out + k]:  k < n_levels: sum of A of level k;  k >= n_levels: bit sum j = k - n_levels of the last array
      TreeJobs<F> J{};
      J.quad = quad_tail ? 1u : 0u;
      const XYZZ<F>* in = buckets;
      uint64_t o = 0;
      for (uint32_t lv = 0; lv < n_levels; ++lv) {
        uint32_t threads = lvl_chunks[lv] * WL;
        XYZZ<F>* A = partA + o * WL;
        XYZZ<F>* S = partS + o * WL;
        // a level with few chunks is a chain of 2L dependent additions per lane on a mostly idle device: four lanes per chunk
        // then (msm_reduce_level_quad_kernel, G1); a level that fills the device keeps the lane per chunk (less work in total)
        bool quad_level = false;
        {
          if (quad_tail && threads <= quad_max_chunks) {
            quad_level = true;
            hipLaunchKernelGGL(msm_reduce_level_quad_kernel<F>, dim3((4 * threads + 255) / 256), dim3(256), 0, st, in, lvl_cnt[lv], 1u << lvl_logl[lv],
                               lv == 0 ? 1u : 0u, WL, A, S);
          }
        }
        if (!quad_level)
          hipLaunchKernelGGL(msm_reduce_level_kernel<F>, dim3((threads + 255) / 256), dim3(256), 0, st, in, lvl_cnt[lv], 1u << lvl_logl[lv],
                             lv == 0 ? 1u : 0u, WL, A, S);
        ZK_HIP(hipGetLastError());
        J.in[lv] = A;
        J.cnt[lv] = J.stride[lv] = lvl_chunks[lv];
        J.bit[lv] = -1;
        in = S;
        o += lvl_chunks[lv];
      }
      // jobs of the first tree launch: the levels' A[] (plain sums), then either the bit decomposition of the last array S (`in`)
      // or, with the 2-D tail, its row and column sums; the second launch then carries the bit decompositions of those
      XYZZ<F>* rc = (XYZZ<F>*)(ws + o_rc);
      auto plain = [&](uint32_t j) { J.reps[j] = 1; J.rep_stride[j] = 0; J.elem_stride[j] = 1; J.limit[j] = 0xffffffffu; J.off[j] = 0; };
      for (uint32_t lv = 0; lv < n_levels; ++lv) plain(lv);
      uint32_t n_jobs = n_levels;
      if (!tail2d) {
        for (uint32_t j = 0; j < final_bits; ++j, ++n_jobs) {
          plain(n_jobs);
          J.in[n_jobs] = in;
          J.cnt[n_jobs] = J.stride[n_jobs] = final_cnt;
          J.bit[n_jobs] = (int32_t)j;
          J.off[n_jobs] = final_off;
        }
      } else {
        // rows: t_rows sums of t_cols consecutive elements; columns: t_cols sums of t_rows elements t_cols apart
        J.in[n_jobs] = in; J.cnt[n_jobs] = t_cols; J.stride[n_jobs] = final_cnt; J.bit[n_jobs] = -1; J.off[n_jobs] = 0;
        J.reps[n_jobs] = t_rows; J.rep_stride[n_jobs] = t_cols; J.elem_stride[n_jobs] = 1; J.limit[n_jobs] = final_cnt;
        ++n_jobs;
        J.in[n_jobs] = in; J.cnt[n_jobs] = t_rows; J.stride[n_jobs] = final_cnt; J.bit[n_jobs] = -1; J.off[n_jobs] = 0;
        J.reps[n_jobs] = t_cols; J.rep_stride[n_jobs] = 1; J.elem_stride[n_jobs] = t_cols; J.limit[n_jobs] = final_cnt;
        ++n_jobs;
      }
      const size_t lds = 256 * sizeof(typename BucketAcc<F>::type);
      XYZZ<F>* dst = sumtmp;
      for (uint32_t launch = 0;; ++launch) {
        const bool families = tail2d && launch == 0;  // (this launch leaves R and C behind: another one must follow)
        uint32_t left = 1;  // longest row of slice sums this launch leaves
        uint64_t o_dst = 0;
        for (uint32_t j = 0; j < n_jobs; ++j) {
          const uint32_t sl = (J.cnt[j] + MSM_TREE_SLICE - 1) / MSM_TREE_SLICE;
          J.first_block[j + 1] = J.first_block[j] + J.reps[j] * sl;
          if (sl > left) left = sl;
        }
        const bool last = left == 1 && !families;
        for (uint32_t j = 0; j < n_jobs; ++j) {
          const uint32_t sl = (J.cnt[j] + MSM_TREE_SLICE - 1) / MSM_TREE_SLICE;
          if (families && j >= n_levels) {            // rows -> rc[w][0 .. t_rows), columns -> rc[w][t_rows .. t_rows + t_cols)
            J.out[j] = rc + (j == n_levels ? 0 : t_rows);
            J.out_w[j] = t_rows + t_cols;
          } else if (last) {
            J.out[j] = wsums + j;
            J.out_w[j] = n_out;
          } else {
            J.out[j] = dst + o_dst;                   // WL x sl slice sums of this job
            J.out_w[j] = 1;
            o_dst += (uint64_t)WL * sl;
          }
        }
        J.n_jobs = n_jobs;
        hipLaunchKernelGGL(msm_tree_kernel<F>, dim3(J.first_block[n_jobs] * WL), dim3(256), lds, st, J);
        ZK_HIP(hipGetLastError());
        if (last) break;
        // next launch: plain sums of the rows of slice sums ...
        for (uint32_t j = 0; j < (families ? n_levels : n_jobs); ++j) {
          const uint32_t sl = (J.cnt[j] + MSM_TREE_SLICE - 1) / MSM_TREE_SLICE;
          J.in[j] = J.out[j];
          J.cnt[j] = J.stride[j] = sl;
          J.bit[j] = -1;
          plain(j);
        }
        if (families) {
          // ... and the bit decompositions of the column sums (weights c + off, 2^e) and of the row sums (weights r, 2^(cols_log + e))
          n_jobs = n_levels;
          for (uint32_t j = 0; j < final_bits; ++j, ++n_jobs) {
            plain(n_jobs);
            J.in[n_jobs] = rc + t_rows;
            J.cnt[n_jobs] = t_cols;
            J.stride[n_jobs] = t_rows + t_cols;
            J.bit[n_jobs] = (int32_t)j;
            J.off[n_jobs] = final_off;
          }
          for (uint32_t j = 0; j < row_bits; ++j, ++n_jobs) {
            plain(n_jobs);
            J.in[n_jobs] = rc;
            J.cnt[n_jobs] = t_rows;
            J.stride[n_jobs] = t_rows + t_cols;
            J.bit[n_jobs] = (int32_t)j;
          }
        }
        dst = dst == sumtmp ? sumtmp + (uint64_t)WL * tree_tmp : sumtmp;
      }
    }
    prof_end(slot_red, st);
    if (checkpoint("reduce", plan[0])) return (int)ZK_ERR_DEVICE;

    ZK_HIP(hipMemcpyAsync(pin.b->p, wsums, back_bytes, hipMemcpyDeviceToHost, st));
    // (parking on an event recorded in front of the reduction and polling the stream from there was measured in round 4: 1.790 against
    // 1.805 ms at 2^20, nothing at 2^12 .. 2^22 or for the prover's eight threads -- the runtime's own wait is not what a short call waits for)
    ZK_HIP(hipStreamSynchronize(st));
    const XYZZ<F>* h_wsums = reinterpret_cast<const XYZZ<F>*>(pin.b->p);
    unsigned long long h_errs[2];
    std::memcpy(h_errs, (const char*)pin.b->p + (o_err - o_wsums), 16);
    lease.idle = true;
    static const bool trace_join = std::getenv("MI355ZK_TRACE_MSM") != nullptr;
    const auto t_join0 = std::chrono::steady_clock::now();
    unsigned long long h_err = h_errs[0];
    if (tmode && h_err != ~0ull && h_errs[1] == ~0ull) {
      // (error path) the lowest identity BASE index, by exponent order: see msm_identity_scan_kernel
      lease.idle = false;
      ZK_HIP(hipMemsetAsync(d_err, 0xff, 8, st));
      hipLaunchKernelGGL(msm_identity_scan_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_bases, d_sc_first, n, base_offset, d_density,
                         d_dprefix, d_err);
      ZK_HIP(hipGetLastError());
      ZK_HIP(hipMemcpyAsync(pin.b->p, d_err, 8, hipMemcpyDeviceToHost, st));
      ZK_HIP(hipStreamSynchronize(st));
      std::memcpy(&h_err, pin.b->p, 8);
      lease.idle = true;
    }
    // the device is done with the workspace: let the next multiexp (another host thread -- the prover keeps 8 in
    // flight, prover.rs:250-298) start while this thread joins its partial sums
    if (last_set && lk.owns_lock()) lk.unlock();
    if (h_errs[1] != ~0ull) {
      *err_index_out = (long long)h_errs[1];
      return ZK_ERR_BAD_ARGS;
    }
    if (h_err != ~0ull) {
      *err_index_out = (long long)h_err;
      return ZK_ERR_UNEXPECTED_IDENTITY;
    }
    // result = sum_w 2^shift_w * T_w,  T_w = A_0 + L_0*(A_1 + L_1*(... + sum_j 2^j Bits_j)):  every partial sum
    // P[w][k] carries a power of two 2^(shift_w + e_k).  Terms are collected per exponent and ONE Horner pass
    // (a doubling per bit, multiexp.rs:146-154) joins everything -- ~270 doublings instead of W * (c + e_max).
    std::vector<uint32_t> e_k(n_out);
    uint32_t e_lv = 0;
    for (uint32_t lv = 0; lv < n_levels; ++lv) {
      e_k[lv] = e_lv;
      e_lv += lvl_logl[lv];
    }
    for (uint32_t j = 0; j < final_bits; ++j) e_k[n_levels + j] = e_lv + j;
    for (uint32_t j = 0; j < row_bits; ++j) e_k[n_levels + final_bits + j] = e_lv + cols_log + j;   // (2-D tail: the row index weighs cols = 2^cols_log)
    uint32_t e_max = 0;
    for (uint32_t k = 0; k < n_out; ++k) e_max = std::max(e_max, e_k[k]);
    auto horner = [](std::vector<Jacobian<F>>& by_exp) {  // sum_t 2^t by_exp[t]
      Jacobian<F> acc = by_exp.back();
      for (int t = (int)by_exp.size() - 2; t >= 0; --t) {
        jac_double(acc);
        if (!by_exp[t].is_zero()) jac_add(acc, by_exp[t]);
      }
      return acc;
    };
    Jacobian<F> acc;
    // The window sums T_w are independent: with the helper threads free (JoinPool: a single caller -- the prover's eight
    // concurrent joins take the single-threaded paths below instead), every T_w is joined from its n_out terms in parallel and
    // only the chain over the windows stays serial: G2 at 2^20 0.48 -> 0.26 ms of host time, G1 0.13 -> 0.08 ms.
    static JoinPool join_pool;
    static const bool join_serial = std::getenv("MI355ZK_MSM_JOIN_SERIAL") != nullptr;
    std::vector<Jacobian<F>> T(WL);
    const bool parallel = !join_serial && WL >= 4 && join_pool.run(WL, [&](uint32_t wl) {
      std::vector<Jacobian<F>> by_exp((size_t)e_max + 1, Jacobian<F>::zero());
      for (uint32_t k = 0; k < n_out; ++k) {
        const XYZZ<F>& pt = h_wsums[(size_t)wl * n_out + k];
        if (!pt.is_zero()) jac_add(by_exp[e_k[k]], rec_to_jacobian(pt));
      }
      T[wl] = horner(by_exp);
    });
    auto wshift = [&](uint32_t w) -> uint32_t { return tmode ? 0u : G.shift[w]; };  // (table mode: the table carries the shifts)
    if (parallel && G.rmul == 1) {
      // sum_w 2^shift_w T_w: Horner over the windows from the top one down, then the shift of the group's lowest window
      acc = T[WL - 1];
      for (int wl = (int)WL - 2; wl >= 0; --wl) {
        for (uint32_t r = G.shift[w_lo + wl]; r < G.shift[w_lo + wl + 1]; ++r) jac_double(acc);
        if (!T[wl].is_zero()) jac_add(acc, T[wl]);
      }
      for (uint32_t r = 0; r < G.shift[w_lo]; ++r) jac_double(acc);
    } else if (G.rmul == 1) {
      std::vector<Jacobian<F>> by_exp((size_t)wshift(w_lo + WL - 1) + e_max + 1, Jacobian<F>::zero());
      for (uint32_t wl = 0; wl < WL; ++wl)
        for (uint32_t k = 0; k < n_out; ++k) {
          const XYZZ<F>& pt = h_wsums[(size_t)wl * n_out + k];
          if (!pt.is_zero()) jac_add(by_exp[wshift(w_lo + wl) + e_k[k]], rec_to_jacobian(pt));
        }
      acc = horner(by_exp);
    } else {
      // mixed radix: T_w by its own Horner pass, then  acc = B * acc + T_w  with  B = rmul * 2^rshift; the windows below this
      // call's group contribute nothing here, only their powers of B
      acc = Jacobian<F>::zero();
      for (int w = (int)w_hi - 1; w >= 0; --w) {
        const Jacobian<F> one_acc = acc;            // rmul * acc by double-and-add over the bits of rmul (<= 15)
        int top = 3;
        while (!((G.rmul >> top) & 1u)) --top;
        for (int bit = top - 1; bit >= 0; --bit) {
          jac_double(acc);
          if ((G.rmul >> bit) & 1u) jac_add(acc, one_acc);
        }
        for (uint32_t r = 0; r < G.rshift; ++r) jac_double(acc);
        if (w < (int)w_lo) continue;
        if (parallel) {
          jac_add(acc, T[w - (int)w_lo]);
          continue;
        }
        std::vector<Jacobian<F>> by_exp((size_t)e_max + 1, Jacobian<F>::zero());
        for (uint32_t k = 0; k < n_out; ++k) {
          const XYZZ<F>& pt = h_wsums[(size_t)(w - (int)w_lo) * n_out + k];
          if (!pt.is_zero()) jac_add(by_exp[e_k[k]], rec_to_jacobian(pt));
        }
        jac_add(acc, horner(by_exp));
      }
    }
    *result = acc;
    if (trace_join)
      std::fprintf(stderr, "[mi355zk] msm n=%llu: host join of %u window sums: %.1f us\n", (unsigned long long)n, WL * n_out,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_join0).count());
    return (int)ZK_OK;
  };

  for (uint32_t c = 0; c < n_chunks; ++c) {
    const ChunkPlan& C = plan[c];
    const uint32_t* d_sc = d_scalars;
    if (chunks) {
      const void* p = nullptr;
      rc = chunks->acquire(c, st, &p);  // (makes `st` wait for the chunk's upload)
      if (rc) return rc;
      d_sc = (const uint32_t*)p;
    }
    d_sc_first = d_sc;  // (table mode runs a single chunk: its exponents, for the error path's rescan)
    rc = partition_chunk(C, d_sc);
    if (rc == ZK_OK && chunks) rc = chunks->digits_enqueued(c, st);  // the digit kernel is the only reader of the exponents
    if (rc == ZK_OK) rc = partition_rest(C);
    if (rc == ZK_OK) rc = accumulate_chunk(C, d_bases, c > 0);
    if (rc) return rc;
  }
  const bool two_sets = d_bases2 != nullptr && out2 != nullptr;
  int rc_set = finish_set(out, !two_sets);
  if (rc_set != ZK_OK) return rc_set;
  if (two_sets) {
    lease.idle = false;
    rc = accumulate_chunk(plan[0], d_bases2, false);
    if (rc) return rc;
    return finish_set(out2, true);
  }
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// Segmented sum of affine points: out[r] = sum of points[row_ptr[r] .. row_ptr[r+1]), normalised to affine.
// This is the bucket accumulation above with the rows of a CSR matrix as the "buckets" (size-ordered lanes,
// segment-parallel path for long rows), used by the QAP evaluation of phase2/src/parameters.rs:225-294
// (per variable: sum of coeff * Lagrange-basis point, then batch_normalization).
__global__ void __launch_bounds__(256) msm_iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
template <class F>
__global__ void __launch_bounds__(256) msm_to_affine_kernel(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = xyzz_to_affine(load_vec(in + i));
}

template <class F>
int segsum_device(const Affine<F>* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, Affine<F>* d_out) {
  if (n_rows == 0) return ZK_OK;
  if (nnz >= 0x7fffffffull) return ZK_ERR_BAD_ARGS;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  const uint32_t* first = d_row_ptr;
  const uint32_t* last = d_row_ptr + 1;
  const uint64_t mean_len = nnz / n_rows + 1;
  const uint32_t heavy = (uint32_t)(mean_len * 8 + 1024 > 0xffffffffull ? 0xffffffffull : mean_len * 8 + 1024);
  uint32_t hb = n_rows < MSM_HEAVY_BLOCKS ? n_rows : MSM_HEAVY_BLOCKS;
  if ((uint64_t)hb > nnz / heavy + 1) hb = (uint32_t)(nnz / heavy + 1);
  const uint32_t max_items = (uint32_t)(nnz / MSM_HEAVY_SEG) + hb;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  size_t o_vals = take((size_t)(nnz ? nnz : 1) * 4);
  size_t o_hist = take(MSM_SIZE_BINS * 4), o_sizes_b = take((size_t)n_rows * 4), o_order = take((size_t)n_rows * 4);
  size_t o_item_off = take((size_t)(hb + 2) * 4);
  size_t o_seg = take((size_t)max_items * sizeof(XYZZ<F>));
  size_t o_buckets = take((size_t)n_rows * sizeof(XYZZ<F>));
  Workspace& dev_ws = ws_of(dev);
  std::lock_guard<std::mutex> lk(dev_ws.mu);
  void* base = nullptr;
  int rc = ws_reserve(dev_ws, off, &base);
  if (rc) return rc;
  char* ws = (char*)base;
  uint32_t* vals = (uint32_t*)(ws + o_vals);
  uint32_t* size_hist = (uint32_t*)(ws + o_hist);
  uint32_t* sizes_b = (uint32_t*)(ws + o_sizes_b);
  uint32_t* order = (uint32_t*)(ws + o_order);
  uint32_t* item_off = (uint32_t*)(ws + o_item_off);
  XYZZ<F>* seg_sums = (XYZZ<F>*)(ws + o_seg);
  XYZZ<F>* buckets = (XYZZ<F>*)(ws + o_buckets);
  if (nnz) hipLaunchKernelGGL(msm_iota_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, vals, (uint32_t)nnz);
  ZK_HIP(hipMemsetAsync(size_hist, 0, MSM_SIZE_BINS * 4, st));
  msm_order_by_size(first, last, n_rows, size_hist, order, sizes_b, st);
  ZK_HIP(hipGetLastError());
  hipLaunchKernelGGL(msm_heavy_plan_kernel, dim3(1), dim3(1024), 0, st, sizes_b, hb, heavy, MSM_HEAVY_SEG, item_off);
  const uint32_t heavy_grid = max_items < 16384 ? max_items : 16384;
  hipLaunchKernelGGL(msm_accumulate_heavy_kernel<F>, dim3(heavy_grid), dim3(MSM_HEAVY_LANES), MSM_HEAVY_LANES * sizeof(typename BucketAcc<F>::type), st, d_points,
                     vals, first, last, order, item_off, hb, MSM_HEAVY_SEG, seg_sums, 1, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(msm_heavy_combine_kernel<F>, dim3(hb < 2048 ? hb : 2048), dim3(64), 64 * sizeof(typename BucketAcc<F>::type), st, seg_sums, order, item_off, hb,
                     buckets, 0);
  hipLaunchKernelGGL((msm_accumulate_kernel<F, false, false>), dim3((n_rows + 255) / 256), dim3(256), 0, st, d_points, vals, first, last, order, heavy, hb,
                     n_rows, buckets, 1, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(msm_to_affine_kernel<F>, dim3((n_rows + 255) / 256), dim3(256), 0, st, buckets, d_out, n_rows);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipStreamSynchronize(st));  // the workspace is shared: finish before releasing the lock
  return ZK_OK;
}

}  // namespace

}  // namespace zk
