// Radix-2 number-theoretic transform over BN254 Fr for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/mi355zk.h, the reference's
//   bellman/src/domain.rs:263-376  best_fft / serial_fft / parallel_fft   (-> ntt_run)
//   bellman/src/domain.rs:159-203  ifft scaling, distribute_powers, coset_fft, icoset_fft (fused in)
// for `Scalar<Bn256>` elements (bellman/src/group.rs:53-82): 32-byte Montgomery Fr limbs, natural
// order in, natural order out, in place.
//
// Algorithm (not the reference's): a size-N transform is factored N = N_1 * ... * N_R (R <= 3,
// N_p <= 4096, normally 1024) Cooley-Tukey style.  Pass p transforms digit p of the index for G adjacent
// "columns" at once: a workgroup stages a G x N_p tile (<= 4096 elements) in LDS on nine 29-bit limbs per
// element (U-form, fieldu.hpp; element-major, see lds_load), runs log2(N_p) DIT stages
// there, multiplies by the inter-pass twiddle omega^(T_p*k_p*rest) and writes back in the 32-byte
// memory format.  The last pass writes straight to the natural-order position (fused digit reversal),
// with the G tile rows chosen so that both its loads and its stores are >= 128-byte contiguous.
// distribute_powers (coset) is fused into the first pass's load, the 1/m and g^-k scalings into the
// last pass's store.  HBM traffic: 64 B per element per pass (R passes) -- DESIGN.md "NTT".
//
// U-form bookkeeping (fieldu.hpp): data stays in the memory format's 2^256 domain the whole time -- every
// product is by a table twiddle, kept as the PLAIN integer w with its quotient floor(w 2^261 / p), and
// u_mul_shoup(data, w, wq) returns data*w itself, below 2p (round 4; rounds 1-3: Montgomery products by
// twiddles in the 2^261 domain, 180 multiplier instructions against 143) -- so loads and stores only
// re-pack bits.  DIT butterflies (a + w b, a - w b) grow values linearly: V_s <= V_0 + 2 s p <= 24p after
// 10 stages (30p with the skipped twiddle-one products), far below the 160p the product admits; limbs
// are carried every 4th stage (the bound of each call is noted at the call).
// The mathematical result X[k] = sum_i a[i] w^(ik) is unique and the stored elements are fully
// reduced, so the output bytes are identical to serial_fft's.
#define ZK_CHAIN_MAD 1  // fieldu.hpp u_mad: one dependent mad chain per column (measured faster in this TU)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "fieldu.hpp"
#include "device_util.hpp"

namespace zk {

namespace {

constexpr int NTT_MAX_LOG_NP = 12;   // longest sub-transform a tile row can be (kernel instantiations, root tables)
constexpr int NTT_LOG_NP = 10;       // the pass planner's default: 1024-point rows (longer ones for 2^21 .. 2^23; env MI355ZK_NTT_LOGNP = 10 / 11 / 12 forces)
constexpr int NTT_TILE_ELEMS = 4096; // G * N_p
constexpr int NTT_THREADS = 1024;

struct NttPassParams {
  uint32_t log_np;       // log2(N_p)
  uint32_t g;            // rows (batch) per tile
  uint64_t in_xs, in_gs; // element strides (in elements) of transform index / batch index on load
  uint64_t out_xs, out_gs;
  // tile -> base offsets: tile id = hi * tiles_lo + lo
  uint64_t tiles_lo;
  uint64_t in_hi_stride, in_lo_stride;
  uint64_t out_hi_stride, out_lo_stride;
  uint32_t load_x_fastest;  // lane order on load: 1 = transform index fastest (last pass)
  // inter-pass twiddle  w^(tw_mul * k * (lo*g + gidx)) ; tw_mul == 0 -> none (last pass)
  uint64_t tw_mul;
  uint32_t tw_h;            // two-level split: w^e = A[e >> h] * B[e & (2^h - 1)]
  uint32_t tw_full;         // 1: the first pass of a two-pass transform reads its twiddle w^(k * col) from a table indexed by the OUTPUT position
  uint32_t pre;             // first pass of coset_fft: element i *= g^i   (1: preA/preB, split pre_h; round 5, folded tables: 2: row position x *= preA[x];
                            // 3: butterfly twiddles from the stage table preA (wave-local kernel); 4: that, and *= preB[col])
  uint32_t pre_h;
  uint32_t post;            // last pass: 1 = multiply by post_c; 2 = by post_c * ginv^k (postA/postB, split post_h); 3 = by nothing; 4 = row output k *= postA[k]; 5 = that, and *= postB[first output index of the row]
  uint32_t post_h;
  uint32_t xcd_pair;        // 1: tiles 2j and 2j + 1 run on the same XCD, one dispatch round apart (see the kernel)
  // (round 5) batch > 1: ONE launch runs this pass of `batch` independent transforms of the same size and kind: workgroup ids
  // [t * tiles, (t + 1) * tiles) belong to transform t, whose arrays are NttBatch::in[t] / out[t]
  uint32_t batch;
  uint64_t tiles;
};
constexpr uint32_t NTT_MAX_BATCH = 8;
struct NttBatch {
  const Fr* in[NTT_MAX_BATCH];
  Fr* out[NTT_MAX_BATCH];
};

// table entry (round 4): a twiddle as the PLAIN canonical integer w on nine 29-bit limbs followed by wq = floor(w * 2^261 / p) --
// the constant and its quotient of fieldu.hpp's u_mul_shoup -- 72 B, 8-byte aligned (padding the entry to 80 B for aligned 16-byte
// loads measured 0.5-1 % slower: tools/ab_ntt_shoup.sh).  (Rounds 1-3: w * 2^261 mod p for the Montgomery product, 48 B.)
struct alignas(8) UTab {
  uint32_t l[18];
};
struct TwU {
  FrU w, q;
};

__device__ __forceinline__ TwU tab_load(const UTab* p) {
  const uint2* s = reinterpret_cast<const uint2*>(p);   // (hipcc merges neighbours into 16-byte loads where the address allows)
  uint32_t v[18];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint2 t = s[i];
    v[2 * i] = t.x;
    v[2 * i + 1] = t.y;
  }
  TwU r;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r.w.l[i] = v[i]; r.q.l[i] = v[9 + i]; }
  return r;
}
// x * (the table constant): the value itself (no Montgomery factor), < 2p for x < 160p with limbs < 2^31
__device__ __forceinline__ FrU tw_mul(const FrU& x, const TwU& t) { return u_mul_shoup(x, t.w, t.q); }

// entry for the plain canonical integer c (8 x 32-bit words)
ZK_HD TwU tw_make(const Fr& c_plain) {
  TwU t;
  t.w = u_from_std(c_plain);
  t.q = u_shoup_quotient<FrParams>(c_plain.l);
  return t;
}
__device__ __forceinline__ void tab_store(UTab* p, const TwU& t) {
  UTab e;
#pragma unroll
  for (int i = 0; i < 9; ++i) { e.l[i] = t.w.l[i]; e.l[9 + i] = t.q.l[i]; }
  *p = e;
}

// LDS layout of a tile (round 4): ELEMENT-major, nine consecutive words per element.  The stride of 9 words is odd, so lanes with
// consecutive (or swz-permuted) element indices still sit on different banks -- bank = (9 idx + l) mod 32 is idx mod 32 up to a
// bijection -- and the nine limbs of an element are at FIXED offsets from ONE address: ds_read / ds_write take them as immediates
// (hipcc pairs them into ds_read2_b32), where the limb-PLANE layout of rounds 1-3 (lds[l * plane + idx], plane a run-time value)
// cost one VALU add and one address register per limb: 36 address registers per radix-4 group, ~85 VALU instructions per
// element and pass.  -DZK_NTT_LDS_PLANES restores the planes for the comparison.
__device__ __forceinline__ FrU lds_load(const uint32_t* lds, uint32_t plane, uint32_t idx) {
  FrU r;
#ifdef ZK_NTT_LDS_PLANES
#pragma unroll
  for (int l = 0; l < 9; ++l) r.l[l] = lds[l * plane + idx];
#else
  (void)plane;
  const uint32_t* e = lds + idx * 9u;
#pragma unroll
  for (int l = 0; l < 9; ++l) r.l[l] = e[l];
#endif
  return r;
}
__device__ __forceinline__ void lds_store(uint32_t* lds, uint32_t plane, uint32_t idx, const FrU& v) {
#ifdef ZK_NTT_LDS_PLANES
#pragma unroll
  for (int l = 0; l < 9; ++l) lds[l * plane + idx] = v.l[l];
#else
  (void)plane;
  uint32_t* e = lds + idx * 9u;
#pragma unroll
  for (int l = 0; l < 9; ++l) e[l] = v.l[l];
#endif
}
__device__ __forceinline__ Fr gload(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void gstore(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// Row-local LDS position of element x: the low five bits (the bank, up to the odd element stride 9) are XOR-ed with a mix of the bits
// above them.  Round 4 XOR-ed bits 5..9 in unchanged and padded the rows to np + 1 elements: tools/ntt_lds_model.py (a bank model of every
// access site of this kernel; it reproduces the counter) showed where its 20 % conflict cycles on the 2^20 pass came from -- the stage pair
// with m = 16 (lanes 16 apart hold x and x + 64, and "+ 64" moved the position by 2, inside the same 16 banks) and the closing read of a
// G >= 2 tile (neighbouring lanes alternate rows at +9 banks per row against +9 per element) -- and 33 - 58 % on the tiles of 2^21 .. 2^26.
// Now: bits 5..9 times 25 (so that x + 32, + 64, + 128 ... each land in another part of the 32 banks), bits 10, 11 of the long rows folded
// in, rows at the plain pitch np and told apart by a row term (lds_pos).  Modelled conflict cycles: 2^20 pass 20 % -> 0, 2^21 .. 2^23
// 26 - 33 % -> 0, 2^24 58 % -> 20 %, 2^25 / 2^26 45 % -> 1 %; measured: profiles/r05_ntt20_pass_sq_pmc.txt.
__device__ __forceinline__ uint32_t swz(uint32_t x) { return x ^ ((((x >> 5) * 25u) ^ (x >> 10)) & 31u); }
// position of element x of row g in a tile of rows of np >= 32 elements: `rowx` = (g * a) & 31 with a = 21 (a = 20 for eight-row tiles)
// moves neighbouring rows to other banks where lanes alternate rows (tile load and store); shorter rows are not permuted (swz is the identity there)
__device__ __forceinline__ uint32_t row_term(uint32_t g, uint32_t G, uint32_t np) { return np >= 32 ? (g * (G == 8 ? 20u : 21u)) & 31u : 0u; }

// One pass over one tile.  roots[x] = w_p^x for x < N_p/2 (w_p = omega^(N/N_p)).
template <uint32_t LOG_NP, bool R4>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, NttPassParams P,
                                                              const UTab* __restrict__ roots, const UTab* __restrict__ twA,
                                                              const UTab* __restrict__ twB, const UTab* __restrict__ preA,
                                                              const UTab* __restrict__ preB, const UTab* __restrict__ postA,
                                                              const UTab* __restrict__ postB, TwU post_c, const UTab* __restrict__ twF, NttBatch BP) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr uint32_t np = 1u << LOG_NP;
  // twiddle-one products are skipped in stages 0 .. SKIP_MAX: rows longer than 2^10 give up stage 2 (a skipped stage doubles the
  // value bound instead of adding 2p: 16p + 2 (LOG_NP - 3) p would pass the 32p the closing reduction allows; 10p + 2 (LOG_NP - 3) p does not)
  constexpr uint32_t SKIP_MAX = LOG_NP > 10 ? 1 : 2;
  constexpr uint32_t pitch = np;
  const uint32_t plane = P.g * pitch;
  const uint32_t elems = P.g * np;
  // One- and two-row tiles move 32- / 64-byte runs at a large stride: a fraction of every DRAM burst and page, the rest belonging to
  // the neighbouring tiles.  Workgroup ids go round the eight XCDs, so neighbours would sit behind different L2s; instead the 32 tiles
  // an XCD runs at a time (ids b + 8k) are made NEIGHBOURS, so that the XCD's L2 sees kilobyte runs.
  uint64_t tile = blockIdx.x;
  if (P.batch > 1) {                                   // (uniform: scalar loads of the transform's two pointers from the kernel arguments)
    const uint32_t bt = (uint32_t)(tile / P.tiles);
    tile -= (uint64_t)bt * P.tiles;
    in = BP.in[bt];
    out = BP.out[bt];
  }
  if (P.xcd_pair == 1) tile = (tile & ~15ull) | ((tile & 7ull) << 1) | ((tile >> 3) & 1ull);
  else if (P.xcd_pair == 5) tile = (tile & ~255ull) | ((tile & 7ull) << 5) | ((tile >> 3) & 31ull);  // 32 neighbours per XCD: what it runs at a time
  const uint64_t hi = tile / P.tiles_lo, lo = tile % P.tiles_lo;
  const uint64_t in_base = hi * P.in_hi_stride + lo * P.in_lo_stride;
  const uint64_t out_base = hi * P.out_hi_stride + lo * P.out_lo_stride;

  // load (bit-reversed placement inside each row: the DIT stages below then finish in natural order)
  for (uint32_t e = threadIdx.x; e < elems; e += blockDim.x) {
    uint32_t x, g;
    if (P.load_x_fastest) { x = e & (np - 1); g = e >> LOG_NP; }
    else { g = e % P.g; x = e / P.g; }
    const uint64_t gi = in_base + x * P.in_xs + g * P.in_gs;
    FrU v = u_from_std(gload(in + gi));                                   // < p, N
    if (P.pre == 1) {                                                     // distribute_powers (domain.rs:176-189)
      v = tw_mul(v, tab_load(preA + (gi >> P.pre_h)));                    // g^i = A[i >> h] * B[i & mask]: two products by constants
      v = tw_mul(v, tab_load(preB + (gi & ((1ull << P.pre_h) - 1))));     // < 2p, N
    } else if (P.pre == 2) {
      v = tw_mul(v, tab_load(preA + x));                                  // (g^S)^x: the column's g^col sits in the folded inter-pass table
    }
    lds_store(lds, plane, g * pitch + (swz(bitrev(x, LOG_NP)) ^ row_term(g, P.g, np)), v);
  }
  __syncthreads();

  // DIT stages.  Entering stage s every element has limbs < ((s & 3) + 1) * 2^29 and is < (4 + 2s) p, or < 16p + 2(s - 3)p when
  // the twiddle-one products of stages 1 and 2 are skipped (see j_slow).
  if constexpr (R4) {
  // (LOG_NP is a template parameter and the stage loops are static: the stage number, the skip / carry decisions and the shift amounts
  // of the index arithmetic are compile-time constants in every stage)
  // One butterfly of stage ST on registers: (u, t) -> (u + w t, u - w t); skip: w = 1 (stage 0; twiddle index 0 of stages 1 and 2).
  // Bounds with skipping: a skipped product leaves t as large as u, so values DOUBLE on that path: V_1 < 4p, V_2 < 8p, V_3 < 16p, and
  // with + 2p for each of the stages 3..9: < 30p at the end (u_to_std_lt32p / the closing product allow < 32p); the subtraction
  // constant follows (u_sub<4,1> / <8,1>).  Stage 3 is not skipped: it would take the bound past 32p.
  auto bf = [&](auto stc, FrU& u, FrU& t, const TwU& w, bool skip) {
    constexpr uint32_t ST = (uint32_t) decltype(stc)::value;
    // (t < 24p with limbs < 4*2^29 either way: the skipped product only leaves t as large as u may be)
    if (!skip) t = tw_mul(t, w);                                           // limbs < 4*2^29, < 30p: ok; < 2p, N
    else t = u_carry(t);
    FrU sum = u_add(u, t);                                                 // limbs grow by 2^29, value by 2p
    if constexpr ((ST & 3) == 3) sum = u_carry(sum);
    FrU dif;                                                               // t N; u limbs < 4*2^29 < 2^32 - 2^30 - 16: ok.  N out
    if constexpr (ST == 1) dif = skip ? u_sub<4, 1>(u, t) : u_sub<2, 1>(u, t);        // skipped: t < 4p
    else if constexpr (ST == 2) dif = skip ? u_sub<8, 1>(u, t) : u_sub<2, 1>(u, t);   // skipped: t < 8p
    else dif = u_sub<2, 1>(u, t);                                          // t < 2p
    u = sum;
    t = dif;
  };
  constexpr uint32_t S0 = LOG_NP & 1;  // an odd number of stages: stage 0 alone (all twiddles one), then pairs
  if constexpr (S0 == 1) {
    const uint32_t half = elems >> 1;
    for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
      const uint32_t g = b >> (LOG_NP - 1), x0 = (b & ((np >> 1) - 1)) << 1;
      const uint32_t rx = row_term(g, P.g, np);
      const uint32_t i0 = g * pitch + (swz(x0) ^ rx), i1 = g * pitch + (swz(x0 + 1) ^ rx);
      FrU u = lds_load(lds, plane, i0), t = lds_load(lds, plane, i1);
      bf(std::integral_constant<int, 0>{}, u, t, TwU{u, u}, true);
      lds_store(lds, plane, i0, u);
      lds_store(lds, plane, i1, t);
    }
    __syncthreads();
  }
  // Two stages per LDS round trip: a lane holds the four elements x0 + {0, m, 2m, 3m} (m = 2^s), runs the two butterflies of stage s
  // (same twiddle w^j) and the two of stage s + 1 (twiddle indices j and j + m) on registers.  For the pair that starts at stage 1
  // or 2 the groups of a row are dealt to the lanes with the twiddle index j SLOWEST (lane = j * blocks + block), so j is uniform
  // over a wave and the waves with j == 0 -- twiddle one -- skip those products; later pairs: j fastest (unit-stride LDS).
  const uint32_t quarter = elems >> 2;
  for_limbs<(int)(LOG_NP / 2)>([&](auto pc) {
    constexpr uint32_t s = S0 + 2u * (uint32_t) decltype(pc)::value;
    constexpr uint32_t m = 1u << s;
    constexpr bool j_slow = (s == 1 || s == 2) && (np >> (s + 2)) >= 64;
    for (uint32_t q = threadIdx.x; q < quarter; q += blockDim.x) {
      const uint32_t g = q >> (LOG_NP - 2);
      const uint32_t qf = q & ((np >> 2) - 1);
      uint32_t j, x0;
      if constexpr (j_slow) {
        constexpr uint32_t blocks_log = LOG_NP - 2 - s;                    // blocks of 4m per row
        j = qf >> blocks_log;
        x0 = ((qf & ((1u << blocks_log) - 1u)) << (s + 2)) + j;
      } else {
        j = qf & (m - 1);
        x0 = ((qf >> s) << (s + 2)) + j;
      }
      const uint32_t row = g * pitch, rx = row_term(g, P.g, np);
      const uint32_t ia = row + (swz(x0) ^ rx), ib = row + (swz(x0 + m) ^ rx), ic = row + (swz(x0 + 2 * m) ^ rx), id = row + (swz(x0 + 3 * m) ^ rx);
      FrU a = lds_load(lds, plane, ia), b = lds_load(lds, plane, ib), c = lds_load(lds, plane, ic), d = lds_load(lds, plane, id);
      const bool one = s <= SKIP_MAX && j == 0;                            // stage s (s == 0: j == 0 always)
      {
        TwU w1{a, a};
        if (!one) w1 = tab_load(roots + ((uint64_t)j << (LOG_NP - 1 - s)));
        bf(std::integral_constant<int, (int)s>{}, a, b, w1, one);
        bf(std::integral_constant<int, (int)s>{}, c, d, w1, one);
      }
      const bool one2 = s + 1 <= SKIP_MAX && j == 0;                       // stage s + 1, pair (a, c): index j
      {
        TwU w2{a, a};
        if (!one2) w2 = tab_load(roots + ((uint64_t)j << (LOG_NP - 2 - s)));
        bf(std::integral_constant<int, (int)s + 1>{}, a, c, w2, one2);
      }
      {
        const TwU w3 = tab_load(roots + ((uint64_t)(j + m) << (LOG_NP - 2 - s)));   // pair (b, d): index j + m, never zero
        bf(std::integral_constant<int, (int)s + 1>{}, b, d, w3, false);
      }
      lds_store(lds, plane, ia, a);
      lds_store(lds, plane, ib, b);
      lds_store(lds, plane, ic, c);
      lds_store(lds, plane, id, d);
    }
    __syncthreads();
  });
  } else {
  // radix-2 stages (short transforms run narrow tiles, at least 256 of them: a pass is latency-bound there, and two independent
  // butterflies per lane and stage beat one group of four on half the lanes: 2^16 0.048 ms against 0.053)
  const uint32_t half = elems >> 1;
  // (LOG_NP is a template parameter and the stage loop a static one: the stage number, the skip / carry decisions and the shift
  // amounts of the index arithmetic are compile-time constants in every stage)
  for_limbs<(int)LOG_NP>([&](auto sc) {
    constexpr uint32_t s = (uint32_t) decltype(sc)::value;
    constexpr uint32_t m = 1u << s;
    constexpr bool carry_now = (s & 3) == 3;
    // Stages 1 and 2 (when a row has at least 64 blocks of 2m): the butterflies of a row are dealt to the lanes with the
    // twiddle index j SLOWEST (lane = j * blocks + block), so j is uniform over a wave and the waves with j == 0 -- twiddle
    // one: 1/2 and 1/4 of the butterflies of these stages -- skip the product.  Later stages: j fastest (unit-stride LDS).
    // Bounds: a skipped product leaves t as large as u, so values DOUBLE on that path: V_1 < 4p, V_2 < 8p, V_3 < 16p, and
    // with + 2p for each of the stages 3..9: < 30p at the end (u_to_std_lt32p / the closing product allow < 32p); the
    // subtraction constant follows (u_sub<4,1> / <8,1>).  Stage 3 is not skipped: it would take the bound past 32p.
    constexpr bool j_slow = s >= 1 && s <= SKIP_MAX && (np >> (s + 1)) >= 64;
    for (uint32_t b = threadIdx.x; b < half; b += blockDim.x) {
      uint32_t g = b >> (LOG_NP - 1);
      uint32_t bf = b & ((np >> 1) - 1);
      uint32_t j, x0;
      if (j_slow) {
        const uint32_t blocks_log = LOG_NP - 1 - s;                     // blocks of 2m per row
        j = bf >> blocks_log;
        x0 = ((bf & ((1u << blocks_log) - 1u)) << (s + 1)) + j;
      } else {
        j = bf & (m - 1);
        x0 = ((bf >> s) << (s + 1)) + j;
      }
      const uint32_t rx = row_term(g, P.g, np);
      uint32_t i0 = g * pitch + (swz(x0) ^ rx);
      uint32_t i1 = g * pitch + (swz(x0 + m) ^ rx);
      FrU u = lds_load(lds, plane, i0);
      FrU t = lds_load(lds, plane, i1);
      // (t < 24p with limbs < 4*2^29 either way: the skipped product only leaves t as large as u may be)
      if (s != 0 && !(j_slow && j == 0)) t = tw_mul(t, tab_load(roots + ((uint64_t)j << (LOG_NP - 1 - s))));  // limbs < 4*2^29, < 30p: ok; < 2p, N
      else t = u_carry(t);                                                // w = 1 (stage 0; j == 0): no product
      FrU sum = u_add(u, t);                                              // limbs grow by 2^29, value by 2p
      if (carry_now) sum = u_carry(sum);
      FrU dif;                                                            // t N; u limbs < 4*2^29 < 2^32 - 2^30 - 16: ok.  N out
      if (j_slow && j == 0) dif = s == 1 ? u_sub<4, 1>(u, t) : u_sub<8, 1>(u, t);   // t < 4p (stage 1), < 8p (stage 2)
      else dif = u_sub<2, 1>(u, t);                                       // t < 2p
      lds_store(lds, plane, i0, sum);
      lds_store(lds, plane, i1, dif);
    }
    __syncthreads();
  });
  }

  // store: one more product brings the value below 2p (inter-pass twiddle, or the post scale / one on the last pass)
  for (uint32_t e = threadIdx.x; e < elems; e += blockDim.x) {
    uint32_t g = e % P.g, k = e / P.g;
    FrU v = lds_load(lds, plane, g * pitch + (swz(k) ^ row_term(g, P.g, np)));   // < 30p, limbs < 4*2^29
    const uint64_t go = out_base + k * P.out_xs + g * P.out_gs;
    TwU w;
    if (P.tw_full) {
      w = tab_load(twF + go);                                             // w^(k * col), streamed: one product instead of two
    } else if (P.tw_mul != 0) {
      uint64_t ex = P.tw_mul * k * (lo * P.g + g);
      v = tw_mul(v, tab_load(twA + (ex >> P.tw_h)));                      // w^ex = A[ex >> h] * B[ex & mask]
      w = tab_load(twB + (ex & ((1ull << P.tw_h) - 1)));
    } else if (P.post == 2) {                                             // minv * ginv^k (icoset_fft, domain.rs:197-203)
      v = tw_mul(v, tab_load(postA + (go >> P.post_h)));
      v = tw_mul(v, tab_load(postB + (go & ((1ull << P.post_h) - 1))));
      w = post_c;
    } else if (P.post == 3) {                                             // plain fft / coset_fft: nothing to multiply by --
      gstore(out + go, u_to_std_lt32p(u_carry(v)));                       // reduce the < 24p value directly
      continue;
    } else if (P.post == 4) {
      w = tab_load(postA + k);                                            // (ginv^N1)^k; minv * ginv^k1 sits in the folded inter-pass table
    } else {
      w = post_c;                                                         // minv (ifft, domain.rs:163-173)
    }
    const FrU prod = tw_mul(v, w);                                         // v < 30p, limbs < 4*2^29: < 2p
    gstore(out + go, u_to_std_lt2p(prod));
  }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the radix-4 pass with WAVE-LOCAL stage pairs (full tiles: G * np = 2048 or 4096 elements, a lane per group of four).
// Same arithmetic per element as ntt_pass_kernel<LOG_NP, true> above (same butterflies, same skipped twiddle-one products, hence the
// same value bounds and the same bytes); what changes is WHICH lane runs which group and how the lanes wait for one another:
//   * lane q takes row g = q mod G and group index qf = q / G in EVERY stage pair (twiddle index j = qf mod m fastest).  A wave
//     therefore owns 64 / G consecutive groups of every row, i.e. an aligned block of 256 / G elements per row, for as long as the
//     groups of a stage pair span no more than that block: the hand-over from pair s to pair s + 2 then stays inside the wave -- the
//     LDS executes a wave's writes and reads in order -- and needs NO workgroup barrier.  For the 2 x 1024 tiles of a 2^20 transform
//     that removes the barriers behind pairs 0 and 2; with the two below, 2 barriers are left of 6.  (Measured before the rewrite, by
//     deleting those barriers from the old kernel -- wrong results, right cost: 61.0 -> 58.6 us per pass.)
//   * column passes load every lane's first group straight into registers: the elements of a group of pair 0 sit at bit-reversed
//     positions 4 qf + k, i.e. they are the elements brev(k) * np / 4 + brev(qf) of the row -- rows of a column pass are strided in
//     memory anyway, so these loads coalesce exactly like the old tile load (G adjacent 32-byte records per element index) -- and
//     the tile's first LDS round trip and the barrier behind the load disappear.  The last pass (rows contiguous in memory) keeps
//     the coalesced tile load through LDS.
//   * the last stage pair leaves the lane with the elements j + r * np / 4 of its row in natural order: they are multiplied by the
//     inter-pass twiddle / scale and stored from registers (G adjacent records per element index, as before): no closing LDS round
//     trip and barrier either.
// The twiddle-one skip of the old kernel's pair 1 / 2 relied on dealing the groups j-slowest; here a lane whose j is 0 still takes
// the skipping branch (same arithmetic), its wave just does not save the time.
// LDS positions: swizzle multiplier 13 and per-G row terms chosen with the bank model for THIS lane order (tools/ntt_lds_model.py).
__device__ __forceinline__ uint32_t swz_wl(uint32_t x) { return x ^ ((((x >> 5) * 13u) ^ (x >> 10)) & 31u); }
__device__ __forceinline__ uint32_t row_term_wl(uint32_t g, uint32_t G) { return (g * (G == 2 ? 26u : G == 4 ? 21u : 25u)) & 31u; }
__device__ __forceinline__ void wave_handover() {
  // the wave's LDS writes of this stage pair before its reads of the next: nothing for the hardware (in order per wave), an order for the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <uint32_t LOG_NP>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass_wl_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, NttPassParams P,
                                                                 const UTab* __restrict__ roots, const UTab* __restrict__ twA,
                                                                 const UTab* __restrict__ twB, const UTab* __restrict__ preA,
                                                                 const UTab* __restrict__ preB, const UTab* __restrict__ postA,
                                                                 const UTab* __restrict__ postB, TwU post_c, const UTab* __restrict__ twF, NttBatch BP) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  static_assert(LOG_NP >= 8, "full tiles of rows of at least 256 elements");
  constexpr uint32_t np = 1u << LOG_NP;
  constexpr uint32_t SKIP_MAX = LOG_NP > 10 ? 1 : 2;   // as in ntt_pass_kernel
  constexpr uint32_t pitch = np;
  constexpr uint32_t plane = 0;                         // (element-major tiles only)
  const uint32_t G = P.g, log_g = 31u - (uint32_t)__clz(G);
  const uint32_t elems = G * np;                        // == 4 * blockDim.x
  uint64_t tile = blockIdx.x;
  if (P.batch > 1) {
    const uint32_t bt = (uint32_t)(tile / P.tiles);
    tile -= (uint64_t)bt * P.tiles;
    in = BP.in[bt];
    out = BP.out[bt];
  }
  if (P.xcd_pair == 1) tile = (tile & ~15ull) | ((tile & 7ull) << 1) | ((tile >> 3) & 1ull);
  else if (P.xcd_pair == 5) tile = (tile & ~255ull) | ((tile & 7ull) << 5) | ((tile >> 3) & 31ull);
  const uint64_t hi = tile / P.tiles_lo, lo = tile % P.tiles_lo;
  const uint64_t in_base = hi * P.in_hi_stride + lo * P.in_lo_stride;
  const uint64_t out_base = hi * P.out_hi_stride + lo * P.out_lo_stride;
  const uint32_t q = threadIdx.x;
  // (a wave whose lanes share ONE row would own 256 elements of it and need one barrier less -- but its loads and stores are then lone
  // 32-byte records, the neighbour column's record going to another wave: measured 0.117 -> 0.139 ms at 2^20, 1.96 -> 2.66 ms at 2^24)
  const uint32_t g = q & (G - 1u), qf = q >> log_g;    // row, group index (qf < np / 4)
  const uint32_t own = 256u / G;                        // elements of a row a wave owns
  const uint32_t row = g * pitch, rx = row_term_wl(g, G);
  // P.pre == 3 (folded tables): the row twist of a coset transform sits in the butterfly twiddles -- a stage-major table (ntt_stage_table_kernel)
  // in preA instead of `roots`, and no twiddle is one
  const bool twisted = P.pre >= 3;
  const UTab* stage_tab = twisted ? preA : roots;
  auto at = [&](uint32_t x) __attribute__((always_inline)) { return row + (swz_wl(x) ^ rx); };

  auto fetch = [&](uint64_t gi, uint32_t x) __attribute__((always_inline)) {           // element x of a row from memory: < 2p, N
    FrU v = u_from_std(gload(in + gi));
    if (P.pre == 4) v = tw_mul(v, tab_load(preB + (gi - (uint64_t)x * P.in_xs)));     // transforms without a full table: g^col (a first pass: gi = x * S + col)
    if (P.pre == 1) {                                   // distribute_powers (domain.rs:176-189)
      v = tw_mul(v, tab_load(preA + (gi >> P.pre_h)));
      v = tw_mul(v, tab_load(preB + (gi & ((1ull << P.pre_h) - 1))));
    } else if (P.pre == 2) {
      v = tw_mul(v, tab_load(preA + x));                // folded tables: (g^S)^x here, g^col in the inter-pass table
    }
    return v;
  };
  // (the butterfly of ntt_pass_kernel, bounds noted there)
  auto bf = [&](auto stc, FrU& u, FrU& t, const TwU& w, bool skip) __attribute__((always_inline)) {
    constexpr uint32_t ST = (uint32_t) decltype(stc)::value;
    if (!skip) t = tw_mul(t, w);
    else t = u_carry(t);
    FrU sum = u_add(u, t);
    if constexpr ((ST & 3) == 3) sum = u_carry(sum);
    FrU dif;
    if constexpr (ST == 1) dif = skip ? u_sub<4, 1>(u, t) : u_sub<2, 1>(u, t);
    else if constexpr (ST == 2) dif = skip ? u_sub<8, 1>(u, t) : u_sub<2, 1>(u, t);
    else dif = u_sub<2, 1>(u, t);
    u = sum;
    t = dif;
  };
  // element k of row g leaves the tile: the closing product (inter-pass twiddle, or the post scale) and the store
  auto emit = [&](uint32_t k, const FrU& v_in) __attribute__((always_inline)) {
    FrU v = v_in;                                       // < 30p, limbs < 4*2^29
    const uint64_t go = out_base + k * P.out_xs + g * P.out_gs;
    TwU w;
    if (P.tw_full) {
      w = tab_load(twF + go);
    } else if (P.tw_mul != 0) {
      const uint64_t ex = P.tw_mul * k * (lo * G + g);
      v = tw_mul(v, tab_load(twA + (ex >> P.tw_h)));
      w = tab_load(twB + (ex & ((1ull << P.tw_h) - 1)));
    } else if (P.post == 2) {
      v = tw_mul(v, tab_load(postA + (go >> P.post_h)));
      v = tw_mul(v, tab_load(postB + (go & ((1ull << P.post_h) - 1))));
      w = post_c;
    } else if (P.post == 3) {
      gstore(out + go, u_to_std_lt32p(u_carry(v)));
      return;
    } else if (P.post == 4) {
      w = tab_load(postA + k);
    } else if (P.post == 5) {                           // transforms without a full table: (minv * ginv^(row's first output index)) * (ginv^stride)^k
      v = tw_mul(v, tab_load(postB + (go - (uint64_t)k * P.out_xs)));
      w = tab_load(postA + k);
    } else {
      w = post_c;
    }
    gstore(out + go, u_to_std_lt2p(tw_mul(v, w)));
  };

  constexpr uint32_t S0 = LOG_NP & 1;
  FrU a, b, c, d;
  const bool direct = S0 == 0 && !P.load_x_fastest;    // (uniform)
  if (direct) {
    const uint32_t xr = bitrev(qf, LOG_NP - 2);         // element index = brev2(k) * np / 4 + brev(qf)
    a = fetch(in_base + (uint64_t)(xr) * P.in_xs + g * P.in_gs, xr);
    b = fetch(in_base + (uint64_t)(xr + 2u * (np >> 2)) * P.in_xs + g * P.in_gs, xr + 2u * (np >> 2));   // k = 1 -> brev2 = 2
    c = fetch(in_base + (uint64_t)(xr + 1u * (np >> 2)) * P.in_xs + g * P.in_gs, xr + 1u * (np >> 2));   // k = 2 -> brev2 = 1
    d = fetch(in_base + (uint64_t)(xr + 3u * (np >> 2)) * P.in_xs + g * P.in_gs, xr + 3u * (np >> 2));
  } else {
    for (uint32_t e = threadIdx.x; e < elems; e += blockDim.x) {
      uint32_t x, gg;
      if (P.load_x_fastest) { x = e & (np - 1); gg = e >> LOG_NP; }
      else { gg = e & (G - 1u); x = e >> log_g; }
      const uint64_t gi = in_base + x * P.in_xs + gg * P.in_gs;
      lds_store(lds, plane, gg * pitch + (swz_wl(bitrev(x, LOG_NP)) ^ row_term_wl(gg, G)), fetch(gi, x));
    }
    __syncthreads();
    if constexpr (S0 == 1) {                           // stage 0 alone (all twiddles one): two butterflies per lane, neighbours in the row
      const uint32_t x0 = qf << 2;
      a = lds_load(lds, plane, at(x0)); b = lds_load(lds, plane, at(x0 + 1)); c = lds_load(lds, plane, at(x0 + 2)); d = lds_load(lds, plane, at(x0 + 3));
      TwU w0{a, a};
      if (twisted) w0 = tab_load(stage_tab + 1);
      bf(std::integral_constant<int, 0>{}, a, b, w0, !twisted);
      bf(std::integral_constant<int, 0>{}, c, d, w0, !twisted);
      lds_store(lds, plane, at(x0), a); lds_store(lds, plane, at(x0 + 1), b); lds_store(lds, plane, at(x0 + 2), c); lds_store(lds, plane, at(x0 + 3), d);
      // pair 1 takes groups of 8 elements = two of these lanes' quadruples: inside the wave while it owns >= 8 elements per row
      if (8u <= own) wave_handover(); else __syncthreads();
    }
  }
  for_limbs<(int)(LOG_NP / 2)>([&](auto pc) {
    constexpr uint32_t s = S0 + 2u * (uint32_t) decltype(pc)::value;
    constexpr uint32_t m = 1u << s;
    constexpr bool first = decltype(pc)::value == 0, last = s + 2 == LOG_NP;
    const uint32_t j = qf & (m - 1), x0 = ((qf >> s) << (s + 2)) + j;
    const uint32_t ia = at(x0), ib = at(x0 + m), ic = at(x0 + 2 * m), id = at(x0 + 3 * m);
    if (!(first && direct)) { a = lds_load(lds, plane, ia); b = lds_load(lds, plane, ib); c = lds_load(lds, plane, ic); d = lds_load(lds, plane, id); }
    const bool one = !twisted && s <= SKIP_MAX && j == 0;
    {
      TwU w1{a, a};
      if (!one) w1 = tab_load(stage_tab + (twisted ? m + j : j << (LOG_NP - 1 - s)));   // (requesting it before the hand-over, to run under the wait: measured, nothing)
      bf(std::integral_constant<int, (int)s>{}, a, b, w1, one);
      bf(std::integral_constant<int, (int)s>{}, c, d, w1, one);
    }
    const bool one2 = !twisted && s + 1 <= SKIP_MAX && j == 0;
    {
      TwU w2{a, a};
      if (!one2) w2 = tab_load(stage_tab + (twisted ? 2u * m + j : j << (LOG_NP - 2 - s)));
      bf(std::integral_constant<int, (int)s + 1>{}, a, c, w2, one2);
    }
    {
      const TwU w3 = tab_load(stage_tab + (twisted ? 3u * m + j : (j + m) << (LOG_NP - 2 - s)));
      bf(std::integral_constant<int, (int)s + 1>{}, b, d, w3, false);
    }
    if constexpr (!last) {
      lds_store(lds, plane, ia, a); lds_store(lds, plane, ib, b); lds_store(lds, plane, ic, c); lds_store(lds, plane, id, d);
      // the next pair's groups span 16 m elements of a row; the wave owns 256 / G of them
      if (16u * m <= own) wave_handover(); else __syncthreads();
    } else {
      emit(x0, a); emit(x0 + m, b); emit(x0 + 2 * m, c); emit(x0 + 3 * m, d);   // (last pair: x0 == j == qf)
    }
  });
}

// tab[j] = base^(j * step): the plain integer and its quotient (UTab)
__global__ void ntt_pow_table_kernel(UTab* tab, Fr base, uint64_t step, uint64_t count) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  tab_store(tab + j, tw_make(to_canonical(pow_u64(base, j * step))));
}

// tab[j] = c * base^j (c a Montgomery form)
__global__ void ntt_pow_scaled_table_kernel(UTab* tab, Fr base, Fr c, uint64_t count) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  tab_store(tab + j, tw_make(to_canonical(mul(pow_u64(base, j), c))));
}

// The inter-pass twiddles of a two-pass transform N = N_1 * S, laid out like the first pass's OUTPUT: full[k * S + col] = w^(k * col)
// (k < N_1, col < S), entries like every table's (UTab) -- from the two-level table w^e = A[e >> h] * B[e & mask].
__global__ void ntt_full_twiddle_kernel(UTab* __restrict__ full, const UTab* __restrict__ A, const UTab* __restrict__ B, uint32_t h,
                                        uint32_t log_s, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t k = i >> log_s, col = i & ((1ull << log_s) - 1), ex = k * col;
  const FrU w = tw_mul(tab_load(A + (ex >> h)).w, tab_load(B + (ex & ((1ull << h) - 1))));  // < 2p, N
  tab_store(full + i, tw_make(u_to_std_lt2p(w)));                                           // canonical, and its quotient
}

// (round 5) The same table with the scale factors of a coset / inverse transform FOLDED in:
//   full[k * S + col] = w^(k * col) * pre^col * post_c * post^k
// The input twist pre^i of coset_fft (i = x * S + col: domain.rs:176-189) splits into pre^col -- constant over the first pass's
// sub-transform of column col, so it commutes with it and lands here -- and (pre^S)^x, ONE product by a table of N_1 entries at the
// load.  The output scale post_c * post^K of ifft / icoset_fft (K = k + N_1 * k2: domain.rs:163-173, 197-203) splits into
// post_c * post^k -- constant over the second pass's sub-transform of row k: here -- and (post^N_1)^k2, ONE product by a table of S
// entries at the last store.  coset_fft's first pass: two products less one; ifft's last pass: one less; icoset_fft's: three less one.
__global__ void ntt_full_folded_kernel(UTab* __restrict__ full, const UTab* __restrict__ A, const UTab* __restrict__ B, uint32_t h,
                                       uint32_t log_s, uint64_t count, const UTab* __restrict__ preA, const UTab* __restrict__ preB,
                                       uint32_t pre_h, const UTab* __restrict__ postA, const UTab* __restrict__ postB, uint32_t post_h,
                                       TwU post_c, int has_post_c) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t k = i >> log_s, col = i & ((1ull << log_s) - 1), ex = k * col;
  FrU w = tw_mul(tab_load(A + (ex >> h)).w, tab_load(B + (ex & ((1ull << h) - 1))));        // < 2p, N (and so after every product below)
  if (preA != nullptr) w = tw_mul(tw_mul(w, tab_load(preA + (col >> pre_h))), tab_load(preB + (col & ((1ull << pre_h) - 1))));
  if (postA != nullptr) w = tw_mul(tw_mul(w, tab_load(postA + (k >> post_h))), tab_load(postB + (k & ((1ull << post_h) - 1))));
  if (has_post_c) w = tw_mul(w, post_c);
  tab_store(full + i, tw_make(u_to_std_lt2p(w)));
}

// (round 5) The row twist (pre^S)^x of a coset transform's first pass folded into the butterflies: a DIT block of 2m = 2^(s+1) outputs at
// stage s is the transform of the row's samples at stride sigma = N_p / 2m, and  sum_t x_t h^(sigma t) w_2m^(t k)  =  E(k) + h^sigma w_2m^k O(k)
// with E, O the same sums over the even / odd samples (stride 2 sigma) -- so the twisted transform is the plain one with stage s's twiddle
// w_2m^j replaced by h^sigma w_2m^j, and no product at the load at all.  Stage-major table: tab[m + j] = h^sigma * w_p^(j sigma), j < m = 2^s
// (tab[0] unused).  The twiddle-one products of the early stages are no longer skipped (h^sigma != 1).
__global__ void ntt_stage_table_kernel(UTab* tab, Fr pre, uint64_t s_cols, Fr omega, uint64_t step0, uint32_t log_np) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_np)) return;
  if (i == 0) { tab_store(tab, tw_make(to_canonical(Fr::one()))); return; }
  const uint32_t s = 31u - (uint32_t)__clz(i), m = 1u << s, j = i - m;
  const uint64_t sigma = (1ull << log_np) >> (s + 1);
  tab_store(tab + i, tw_make(to_canonical(mul(pow_u64(pre, s_cols * sigma), pow_u64(omega, step0 * j * sigma)))));
}

// a[i] *= c * gA[i >> h] * gB[i & mask]   (gA == nullptr: a[i] *= c);  c in the memory format
__global__ void ntt_scale_kernel(Fr* a, uint64_t n, Fr c, const UTab* __restrict__ gA, const UTab* __restrict__ gB, uint32_t h) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FrU v = u_from_std(gload(a + i));
  v = u_mul(v, u_mul(u_from_std(c), UPow2<FrParams, 266>::get()));      // (c is a Montgomery form: c * 2^261, then the Montgomery product) < 2p
  if (gA != nullptr) v = tw_mul(tw_mul(v, tab_load(gA + (i >> h))), tab_load(gB + (i & ((1ull << h) - 1))));
  gstore(a + i, u_to_std_lt2p(v));
}

struct Key {
  int dev;
  uint32_t log_n;
  uint32_t w[8];
  bool operator<(const Key& o) const {
    if (dev != o.dev) return dev < o.dev;
    if (log_n != o.log_n) return log_n < o.log_n;
    for (int i = 0; i < 8; ++i)
      if (w[i] != o.w[i]) return w[i] < o.w[i];
    return false;
  }
};

// Per (device, log_n, omega) tables; built once and kept (the prover reuses one domain size for
// every fft of a proof: bellman/src/groth16/prover.rs:217-241).
// hipMalloc that adds to the owner's byte count (the table cache is bounded by BYTES per device: tables_make_room)
template <class T>
static hipError_t tab_malloc(T** p, size_t bytes, size_t* account) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) *account += bytes;
  return e;
}
struct PowTables {
  size_t bytes = 0;        // device bytes of A, B, roots, full (folded tables count in their own `bytes`)
  uint32_t h = 0;          // w^e = A[e >> h] * B[e & (2^h-1)], e < 2^log_n
  UTab* A = nullptr;
  UTab* B = nullptr;
  UTab* roots[NTT_MAX_LOG_NP + 1] = {};  // roots[b][x] = (w^(N/2^b))^x, x < 2^(b-1)
  UTab* full = nullptr;    // two-pass transforms up to NTT_FULL_TW_MAX_LOG: w^(k * col) at the first pass's output position (72 B per element)
  uint32_t full_log_s = 0;
  // (round 5) `full` with a transform's scale factors folded in (ntt_full_folded_kernel), one per (pre_g, post_c, post_g) this root has
  // been used with -- a domain uses two roots with two each: (fft, coset_fft) and (ifft, icoset_fft)
  struct Folded {
    size_t bytes = 0;
    bool has_pre = false, has_post_c = false, has_post_g = false;
    Fr pre{}, post_c{}, post_g{};
    uint32_t log_s = 0;
    UTab* full = nullptr;
    UTab* pre_rows = nullptr;   // (pre^S)^x, x < N_1
    UTab* pre_stages = nullptr; // the first pass's butterfly twiddles times the row twist (ntt_stage_table_kernel), N_1 entries
    UTab* post_rows = nullptr;  // (post^N_1)^k2, k2 < S  (in general: (post^(N / N_last))^k, k < N_last)
    // transforms without a full table (2^21 and up):
    bool big = false;
    UTab* pre_cols = nullptr;    // pre^col, col < N / N_first
    UTab* post_rowc = nullptr;   // post_c * post^rb, rb < N / N_last: the factor shared by the outputs of one row of the last pass
    UTab* tw_b_scaled = nullptr; // ifft: the low table B of the two-level twiddle times post_c -- the pass before the last multiplies it in
    void free_tabs() {
      (void)hipFree(full); (void)hipFree(pre_rows); (void)hipFree(pre_stages); (void)hipFree(post_rows);
      (void)hipFree(pre_cols); (void)hipFree(post_rowc); (void)hipFree(tw_b_scaled);
    }
  };
  std::vector<Folded> folded;
  void free_all() {
    (void)hipFree(A);
    (void)hipFree(B);
    (void)hipFree(full);
    for (auto* r : roots) (void)hipFree(r);
    for (auto& f : folded) f.free_tabs();
    folded.clear();
  }
};
constexpr size_t NTT_FOLDED_MAX = 4;   // per root: a caller cycling through coset generators must not grow device memory without limit
// Measured (round 3): 2^20 fft 0.1507 -> 0.1456 ms with the table (one product less per element of the first pass, 50 MB more to
// stream); at 2^22 the 192 MiB table makes the transform SLOWER (0.564 -> 0.580 ms): the pass is VALU-bound only while its streams stay
// inside the L2 / Infinity Cache.  Hence two-pass transforms up to 2^20 only.
constexpr uint32_t NTT_FULL_TW_MAX_LOG = 20;

std::mutex g_mu;
std::map<Key, PowTables> g_tables;

// The cache is keyed on arbitrary roots (best_fft takes a caller-supplied omega): bounded per device, so that a caller cycling through
// roots cannot grow device memory without limit.  Called ONCE at the start of a transform (under g_run_mu), before any of its up to
// three table lookups: a drop between two lookups of one call would free the tables the first lookup has just returned (found by the
// NTT fuzz: 64 + entries in one process).  Dropping this device's entries is safe once the device is idle.
constexpr size_t NTT_TABLES_MAX = 64;
// ... and by BYTES: an entry is small up to 2^20 except for its full tables -- 72 B x 2^log_n each, up to 1 + NTT_FOLDED_MAX per
// root -- so 64 entries could hold ~18 GiB.  Default budget 4 GiB per device (env MI355ZK_NTT_TABLES_GB): a prover's two roots with their
// folded tables at 2^20 are 0.3 GiB.
static size_t tables_byte_budget() {
  static const size_t v = [] {
    const char* e = std::getenv("MI355ZK_NTT_TABLES_GB");
    const double gb = e ? std::atof(e) : 4.0;
    return (size_t)((gb > 0.03125 ? gb : 0.03125) * 1073741824.0);
  }();
  return v;
}
int tables_make_room(size_t need) {
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(g_mu);
    size_t mine = 0, bytes = 0;
    for (auto& kv : g_tables) {
      if (kv.first.dev != dev) continue;
      ++mine;
      bytes += kv.second.bytes;
      for (auto& f : kv.second.folded) bytes += f.bytes;
    }
    if (mine + need <= NTT_TABLES_MAX && bytes <= tables_byte_budget()) return ZK_OK;
  }
  // the device drains OUTSIDE g_mu (callers on other devices keep looking their tables up); g_run_mu, held by the caller, keeps new
  // launches of THIS cache's tables from starting meanwhile
  ZK_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_tables.begin(); it != g_tables.end();) {
    if (it->first.dev != dev) { ++it; continue; }
    it->second.free_all();
    it = g_tables.erase(it);
  }
  return ZK_OK;
}

int build_pow_tables(hipStream_t st, uint32_t log_n, const Fr& w, bool want_roots, const uint32_t* bs, int nb, PowTables** out,
                     uint32_t full_log_s = 0) {
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  Key key;
  key.dev = dev;
  key.log_n = log_n;
  for (int i = 0; i < 8; ++i) key.w[i] = w.l[i];
  std::lock_guard<std::mutex> lk(g_mu);
  PowTables& T = g_tables[key];
  bool built = false;
  // on any failure the entry is removed again: a half-built entry (A set, B or a roots table missing) would be taken for
  // complete by the next call
  auto fail = [&](hipError_t e, const char* what) {
    std::fprintf(stderr, "[mi355zk] NTT table build failed (%s): %s\n", what, hipGetErrorString(e));
    (void)hipStreamSynchronize(st);
    T.free_all();
    g_tables.erase(key);
    return (int)ZK_ERR_DEVICE;
  };
  hipError_t e = hipSuccess;
  if (T.A == nullptr) {
    built = true;
    T.h = (log_n + 1) / 2;
    uint64_t nB = 1ull << T.h, nA = 1ull << (log_n - T.h);
    if ((e = tab_malloc(&T.A, nA * sizeof(UTab), &T.bytes)) != hipSuccess) return fail(e, "A");
    if ((e = tab_malloc(&T.B, nB * sizeof(UTab), &T.bytes)) != hipSuccess) return fail(e, "B");
    hipLaunchKernelGGL(ntt_pow_table_kernel, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, st, T.A, w, nB, nA);
    hipLaunchKernelGGL(ntt_pow_table_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, st, T.B, w, 1ull, nB);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
  }
  if (want_roots) {
    for (int p = 0; p < nb; ++p) {
      uint32_t b = bs[p];
      if (b == 0 || T.roots[b] != nullptr) continue;
      built = true;
      uint64_t cnt = 1ull << (b - 1);
      if ((e = tab_malloc(&T.roots[b], cnt * sizeof(UTab), &T.bytes)) != hipSuccess) return fail(e, "roots");
      hipLaunchKernelGGL(ntt_pow_table_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, T.roots[b], w, 1ull << (log_n - b), cnt);
      if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
    }
  }
  if (full_log_s != 0 && (T.full == nullptr || T.full_log_s != full_log_s)) {
    if (T.full) {  // (another split of the same size: only when MI355ZK_NTT_LOGNP changes between calls)
      if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "sync");
      (void)hipFree(T.full);
      T.full = nullptr;
      T.bytes -= sizeof(UTab) << log_n;
    }
    built = true;
    const uint64_t cnt = 1ull << log_n;
    if ((e = tab_malloc(&T.full, cnt * sizeof(UTab), &T.bytes)) != hipSuccess) return fail(e, "full");
    T.full_log_s = full_log_s;
    hipLaunchKernelGGL(ntt_full_twiddle_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, T.full, T.A, T.B, T.h, full_log_s, cnt);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
  }
  if (built && (e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");  // one-time: tables may be used from other streams later
  *out = &T;
  return 0;
}

// The folded tables of (T's root, pre_g, post_c, post_g) (any of the three may be null, not all): found or built.  Called under g_run_mu;
// the device pointers are copied out before anybody can evict the entry.  with_full: the two-pass transform's full inter-pass table with
// everything that is constant per column / per row folded in (ntt_full_folded_kernel); otherwise (2^21 and up, no full table) the small
// tables of the same split: the first pass's stage table and pre^col, the last pass's (post^stride)^k and post_c * post^(row's first index),
// and for a transform scaled by post_c alone the low table of the two-level twiddle times post_c.
// bits[0 .. R): the passes' row lengths (log2), first pass first.
int build_folded(hipStream_t st, uint32_t log_n, const Fr& omega, PowTables* T, const PowTables* Tpre, const PowTables* Tpost, const Fr* pre_g, const Fr* post_c,
                 const TwU& post_cu, const Fr* post_g, const uint32_t* bits, int R, bool with_full, PowTables::Folded* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  const uint32_t log_first = bits[0], log_last = bits[R - 1];
  const uint32_t log_s = log_n - log_first;             // columns of the first pass
  for (const auto& f : T->folded) {
    if (f.log_s != log_s || f.big != !with_full || f.has_pre != (pre_g != nullptr) || f.has_post_c != (post_c != nullptr) || f.has_post_g != (post_g != nullptr)) continue;
    if (pre_g && std::memcmp(&f.pre, pre_g, sizeof(Fr)) != 0) continue;
    if (post_c && std::memcmp(&f.post_c, post_c, sizeof(Fr)) != 0) continue;
    if (post_g && std::memcmp(&f.post_g, post_g, sizeof(Fr)) != 0) continue;
    *out = f;
    return 0;
  }
  if (T->folded.size() >= NTT_FOLDED_MAX) {   // the oldest goes, once nothing on the device can still be reading it
    ZK_HIP(hipDeviceSynchronize());
    T->folded.front().free_tabs();
    T->folded.erase(T->folded.begin());
  }
  PowTables::Folded f;
  f.log_s = log_s;
  f.big = !with_full;
  if (pre_g) { f.has_pre = true; f.pre = *pre_g; }
  if (post_c) { f.has_post_c = true; f.post_c = *post_c; }
  if (post_g) { f.has_post_g = true; f.post_g = *post_g; }
  const uint64_t cnt = 1ull << log_n, n_first = 1ull << log_first, n_cols = 1ull << log_s, n_last = 1ull << log_last, n_rows_last = cnt >> log_last;
  hipError_t e = hipSuccess;
  auto fail = [&](hipError_t err, const char* what) {
    std::fprintf(stderr, "[mi355zk] NTT folded-table build failed (%s): %s\n", what, hipGetErrorString(err));
    (void)hipStreamSynchronize(st);
    f.free_tabs();
    return (int)ZK_ERR_DEVICE;
  };
  auto grid = [](uint64_t c) { return dim3((unsigned)((c + 255) / 256)); };
  if (pre_g) {
    // i = x * n_cols + col:  pre^i = (pre^n_cols)^x * pre^col
    if ((e = tab_malloc(&f.pre_rows, n_first * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "pre rows");
    hipLaunchKernelGGL(ntt_pow_table_kernel, grid(n_first), dim3(256), 0, st, f.pre_rows, *pre_g, n_cols, n_first);
    if ((e = tab_malloc(&f.pre_stages, n_first * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "pre stages");
    hipLaunchKernelGGL(ntt_stage_table_kernel, grid(n_first), dim3(256), 0, st, f.pre_stages, *pre_g, n_cols, omega, n_cols, log_first);
    if (!with_full) {
      if ((e = tab_malloc(&f.pre_cols, n_cols * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "pre cols");
      hipLaunchKernelGGL(ntt_pow_table_kernel, grid(n_cols), dim3(256), 0, st, f.pre_cols, *pre_g, 1ull, n_cols);
    }
  }
  if (post_g) {
    // K = rb + n_rows_last * k (rb < n_rows_last the row's first output index):  post^K = post^rb * (post^n_rows_last)^k
    if ((e = tab_malloc(&f.post_rows, n_last * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "post rows");
    hipLaunchKernelGGL(ntt_pow_table_kernel, grid(n_last), dim3(256), 0, st, f.post_rows, *post_g, n_rows_last, n_last);
    if (!with_full) {
      if ((e = tab_malloc(&f.post_rowc, n_rows_last * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "post row constants");
      hipLaunchKernelGGL(ntt_pow_scaled_table_kernel, grid(n_rows_last), dim3(256), 0, st, f.post_rowc, *post_g, post_c ? *post_c : Fr::one(), n_rows_last);
    }
  } else if (post_c && !with_full) {
    const uint64_t nB = 1ull << T->h;
    if ((e = tab_malloc(&f.tw_b_scaled, nB * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "scaled twiddles");
    hipLaunchKernelGGL(ntt_pow_scaled_table_kernel, grid(nB), dim3(256), 0, st, f.tw_b_scaled, omega, *post_c, nB);
  }
  if (with_full) {
    if ((e = tab_malloc(&f.full, cnt * sizeof(UTab), &f.bytes)) != hipSuccess) return fail(e, "full");
    hipLaunchKernelGGL(ntt_full_folded_kernel, grid(cnt), dim3(256), 0, st, f.full, T->A, T->B, T->h, log_s, cnt,
                       Tpre ? Tpre->A : nullptr, Tpre ? Tpre->B : nullptr, Tpre ? Tpre->h : 0u, Tpost ? Tpost->A : nullptr,
                       Tpost ? Tpost->B : nullptr, Tpost ? Tpost->h : 0u, post_cu, post_c != nullptr ? 1 : 0);
  }
  if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "sync");   // one-time: the tables may be used from other streams later
  T->folded.push_back(f);
  *out = f;
  return 0;
}

struct ScratchBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
// one scratch array per (device, stream): calls on one stream are ordered by the stream itself, calls on
// different streams must not share a buffer.  Guarded by g_run_mu (held only while launching).
std::map<std::pair<int, hipStream_t>, ScratchBuf> g_scratch;
std::mutex g_run_mu;

std::mutex g_cfg_mu;
std::map<int, int> g_cfg;  // device -> rc of its one-time kernel configuration

}  // namespace

// the tile kernel stages up to 4096 x 36 B = 144 KiB in dynamic LDS (gfx950: 160 KiB per CU)
int ntt_configure() {
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_cfg_mu);  // the attribute is per device: once for every device this process drives
  auto it = g_cfg.find(dev);
  if (it != g_cfg.end()) return it->second;
  int rc = ZK_OK;
  const void* fns[2 * NTT_MAX_LOG_NP + 2] = {};
#define ZK_NTT_FN(L) fns[L] = reinterpret_cast<const void*>(ntt_pass_kernel<L, false>); fns[NTT_MAX_LOG_NP + L] = reinterpret_cast<const void*>(ntt_pass_kernel<L, true>);
  ZK_NTT_FN(1) ZK_NTT_FN(2) ZK_NTT_FN(3) ZK_NTT_FN(4) ZK_NTT_FN(5) ZK_NTT_FN(6) ZK_NTT_FN(7) ZK_NTT_FN(8) ZK_NTT_FN(9) ZK_NTT_FN(10)
  ZK_NTT_FN(11) ZK_NTT_FN(12)
#undef ZK_NTT_FN
  const void* wl[5] = {reinterpret_cast<const void*>(ntt_pass_wl_kernel<8>), reinterpret_cast<const void*>(ntt_pass_wl_kernel<9>),
                       reinterpret_cast<const void*>(ntt_pass_wl_kernel<10>), reinterpret_cast<const void*>(ntt_pass_wl_kernel<11>),
                       reinterpret_cast<const void*>(ntt_pass_wl_kernel<12>)};
  for (const void* f : wl)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) rc = ZK_ERR_DEVICE;
  for (int l = 1; l <= 2 * NTT_MAX_LOG_NP; ++l) {
    hipError_t e = hipFuncSetAttribute(fns[l], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      std::fprintf(stderr, "[mi355zk] hipFuncSetAttribute(ntt_pass_kernel, 160 KiB LDS) failed: %s\n", hipGetErrorString(e));
      rc = ZK_ERR_DEVICE;
    }
  }
  g_cfg[dev] = rc;
  return rc;
}

void ntt_release_all() {
  std::lock_guard<std::mutex> lk2(g_run_mu);  // (same order as the run path: g_run_mu, then g_mu)
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_tables) {
    (void)hipSetDevice(kv.first.dev);
    kv.second.free_all();
  }
  g_tables.clear();
  for (auto& kv : g_scratch) {
    (void)hipSetDevice(kv.first.first);
    (void)hipFree(kv.second.p);
  }
  g_scratch.clear();
}

int ntt_scale(Fr* d_a, uint32_t log_n, const Fr& c, const Fr* g, hipStream_t st);

// host: a Montgomery form -> the plain integer and its quotient.  The quotient is a 261-round division (~10 us on the host): the few
// scale factors a process uses (1/m per domain size) are kept.  Called under g_run_mu.
static TwU to_tw(const Fr& x) {
  static std::vector<std::pair<Fr, TwU>> memo;
  for (const auto& e : memo)
    if (std::memcmp(&e.first, &x, sizeof(Fr)) == 0) return e.second;
  const TwU t = tw_make(to_canonical(x));
  if (memo.size() >= 64) memo.clear();
  memo.emplace_back(x, t);
  return t;
}

// d_a: 2^log_n Fr elements on the current device, in place:
//   a[i] *= pre_g^i (if pre_g)  ->  X[k] = sum_i a[i] * omega^(i*k)  ->  X[k] *= post_c * post_g^k (if given).
int ntt_run_batch(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st);
int ntt_run_scaled(Fr* d_a, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st) {
  return ntt_run_batch(&d_a, 1, log_n, omega, pre_g, post_c, post_g, st);
}

// (round 5) `batch` (1 .. NTT_MAX_BATCH) independent transforms of the same size and kind (prover.rs:217-241 runs ifft and coset_fft on a, b and
// c), every pass ONE launch over all their tiles.  Why: a 2^20 pass is 512 workgroups on 512 workgroup slots -- every CU loads, then computes,
// then stores, and the load and store phases (~15 of a pass's 47 - 55 us) hide behind nothing.  With the tiles of the next transform queued in
// the same launch a CU starts loading them while its other workgroup still computes: 2^20 fft 0.105 -> 0.087 ms per transform for batch = 2 .. 3
// (tools/exp_ntt_streams.py measured the same with one stream per transform: profiles/r05_ntt_batch.txt).
int ntt_run_batch(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st) {
  if (batch == 0 || batch > NTT_MAX_BATCH || d_arrays == nullptr) return ZK_ERR_BAD_ARGS;
  for (uint32_t t = 0; t < batch; ++t)
    if (d_arrays[t] == nullptr) return ZK_ERR_BAD_ARGS;
  if (log_n == 0) {
    // single element: X[0] = a[0] * post_c
    if (post_c == nullptr) return 0;
    for (uint32_t t = 0; t < batch; ++t) {
      int rc0 = ntt_scale(d_arrays[t], 0, *post_c, nullptr, st);
      if (rc0) return rc0;
    }
    return 0;
  }
  Fr* const d_a = d_arrays[0];
  if (log_n > 30) return ZK_ERR_BAD_ARGS;
  const uint64_t n = 1ull << log_n;
  // factor the index: R passes of b[p] bits, b[0] most significant digit (DESIGN.md "NTT")
  uint32_t b[3];
  static const char* lognp_env = std::getenv("MI355ZK_NTT_LOGNP");
  // rows of 2^10 by default; 2^11 / 2^12 where that saves a whole pass: 2^21 and 2^22 in two passes (0.372 -> 0.294 ms, 0.715 ->
  // 0.57 ms), 2^23 as 12 + 11 (1.36 -> 1.18 ms).  The one- and two-row tiles of those passes move 32- / 64-byte runs; the kernel's
  // XCD grouping of neighbouring tiles is what makes them pay.  2^24 ran as 12 + 12 in round 2 (2.43 ms); with two 2048-element
  // workgroups per CU three passes of 2^8-point rows are faster (2.32 ms) than two passes whose 4096-point rows own a CU each.
  int row_bits = NTT_LOG_NP;
  if (log_n == 21 || log_n == 22) row_bits = 11;
  if (log_n == 23) row_bits = 12;
  if (lognp_env && std::atoi(lognp_env) >= 10 && std::atoi(lognp_env) <= 12) row_bits = std::atoi(lognp_env);
  int R = (int)((log_n + row_bits - 1) / row_bits);
  for (int p = 0; p < R; ++p) b[p] = log_n / R + ((uint32_t)p < log_n % R ? 1 : 0);

  int rc = ntt_configure();
  if (rc) return rc;
  // held from the table lookup to the last launch: the table cache may be dropped (when full) only while nobody is between
  // "got a table pointer" and "enqueued the kernels that read it"
  std::lock_guard<std::mutex> run_lk(g_run_mu);
  rc = tables_make_room(3);
  if (rc) return rc;
  PowTables* T = nullptr;
  static const bool no_full = std::getenv("MI355ZK_NTT_NO_FULL_TW") != nullptr;  // (the two-level product, kept for the comparison in DESIGN.md)
  const bool full_tw = R == 2 && log_n <= NTT_FULL_TW_MAX_LOG && !no_full;
  // (round 5) a scaled two-pass transform with a full table takes that table with its scale factors folded in (ntt_full_folded_kernel);
  // env MI355ZK_NTT_NO_FOLD: the separate products of rounds 1-4, for the A/B
  static const bool no_fold = std::getenv("MI355ZK_NTT_NO_FOLD") != nullptr;
  const bool fold = full_tw && !no_fold && (pre_g || post_c || post_g);
  // ... and from 2^21 on (no full table; every pass a full tile of the wave-local kernel) the small tables of the same split
  const bool fold_big = !full_tw && R >= 2 && log_n >= 21 && !no_fold && (pre_g || post_c || post_g);
  rc = build_pow_tables(st, log_n, omega, true, b, R, &T, (full_tw && !fold) ? b[1] : 0);
  if (rc) return rc;
  PowTables* Tpre = nullptr;
  PowTables* Tpost = nullptr;
  if (pre_g) { rc = build_pow_tables(st, log_n, *pre_g, false, nullptr, 0, &Tpre); if (rc) return rc; }
  if (post_g) { rc = build_pow_tables(st, log_n, *post_g, false, nullptr, 0, &Tpost); if (rc) return rc; }
  const TwU post_cu = post_c ? to_tw(*post_c) : (post_g ? to_tw(Fr::one()) : TwU{FrU::zero(), FrU::zero()});   // (neither: post == 3 multiplies by nothing)
  PowTables::Folded F;
  if (fold || fold_big) { rc = build_folded(st, log_n, omega, T, Tpre, Tpost, pre_g, post_c, post_cu, post_g, b, R, fold, &F); if (rc) return rc; }
  const UTab* k_full = fold ? F.full : T->full;
  const UTab* k_preA = fold ? F.pre_rows : (Tpre ? Tpre->A : nullptr);
  const UTab* k_preB = (!fold && Tpre) ? Tpre->B : nullptr;
  const UTab* k_postA = fold ? F.post_rows : (Tpost ? Tpost->A : nullptr);
  const UTab* k_postB = (!fold && Tpost) ? Tpost->B : nullptr;
  static const int slot_pass = prof_slot("ntt_pass");

  Fr* scratch = nullptr;
  if (R > 1) {
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    // (one buffer per stream a caller has ever used: bounded -- a caller that makes a stream per call must not pin a buffer per
    // stream for ever.  Past 16 streams on this device everything is dropped once the device is idle; g_run_mu keeps other
    // transforms from being enqueued meanwhile.)
    if (g_scratch.find(std::make_pair(dev, st)) == g_scratch.end()) {
      size_t mine = 0;
      for (auto& kv : g_scratch) mine += kv.first.first == dev ? 1 : 0;
      if (mine >= 16) {
        ZK_HIP(hipDeviceSynchronize());
        for (auto it = g_scratch.begin(); it != g_scratch.end();) {
          if (it->first.first != dev) { ++it; continue; }
          (void)hipFree(it->second.p);
          it = g_scratch.erase(it);
        }
      }
    }
    ScratchBuf& sb = g_scratch[std::make_pair(dev, st)];
    const size_t scratch_bytes = n * sizeof(Fr) * batch;
    if (sb.bytes < scratch_bytes) {
      if (sb.p) {
        ZK_HIP(hipStreamSynchronize(st));  // earlier passes on this stream may still read the old buffer
        ZK_HIP(hipFree(sb.p));
      }
      sb.p = nullptr;
      sb.bytes = 0;
      ZK_HIP(hipMalloc(&sb.p, scratch_bytes));
      sb.bytes = scratch_bytes;
    }
    scratch = (Fr*)sb.p;
  }

  // tile size: 2048 elements (72 KiB of LDS, 512 lanes with a group of four each: TWO workgroups per CU, whose barriers tie eight
  // waves instead of sixteen and whose load / compute / store phases may drift apart) for transforms of 2^20 and more; rows of 2^12
  // are a tile of their own.  Round 2 measured 2048-element tiles 2-5 % SLOWER -- but with radix-2 stages on 1024-lane workgroups,
  // of which the registers (125 VGPRs) admit one per CU: that was never two workgroups per CU.  With a lane per group of four:
  // 2^20 0.1474 -> 0.1443 ms (ifft 0.1418 -> 0.1386), 2^22 0.569 -> 0.544 (ifft 0.536 -> 0.498).  env MI355ZK_NTT_TILE = 4096 / 2048 / 1024.
  static const char* env_tile = std::getenv("MI355ZK_NTT_TILE");
  uint64_t tile_elems = log_n >= 20 ? 2048 : NTT_TILE_ELEMS;
  if (env_tile && (std::atoi(env_tile) == 2048 || std::atoi(env_tile) == 4096 || std::atoi(env_tile) == 1024)) tile_elems = (uint64_t)std::atoi(env_tile);
  // S[p] = prod_{q>p} N_q ; Tm[p] = prod_{q<p} N_q
  uint64_t S[3], Tm[3];
  for (int p = 0; p < R; ++p) {
    S[p] = 1;
    Tm[p] = 1;
    for (int q = p + 1; q < R; ++q) S[p] <<= b[q];
    for (int q = 0; q < p; ++q) Tm[p] <<= b[q];
  }

  for (int p = 0; p < R; ++p) {
    NttPassParams P{};
    P.log_np = b[p];
    const uint64_t np = 1ull << b[p];
    const Fr* src;
    Fr* dst;
    if (R == 1) { src = d_a; dst = d_a; }
    else if (p == 0) { src = d_a; dst = scratch; }
    else if (p == R - 1) { src = scratch; dst = d_a; }
    else { src = scratch; dst = scratch; }
    NttBatch BP{};
    for (uint32_t t = 0; t < batch; ++t) {
      BP.in[t] = (src == d_a) ? d_arrays[t] : scratch + (uint64_t)t * n;
      BP.out[t] = (dst == d_a) ? d_arrays[t] : scratch + (uint64_t)t * n;
    }
    uint64_t tiles;
    if (p < R - 1 || R == 1) {
      // columns: G adjacent low positions share a tile
      uint64_t G = tile_elems / np;
      if (G < 1) G = 1;
      if (G > S[p]) G = S[p];
      while (G > 1 && n / (np * G) < 256) G >>= 1;  // small transforms: prefer >= 256 tiles (one per CU) over wide tiles
      P.g = (uint32_t)G;
      P.in_xs = P.out_xs = S[p];
      P.in_gs = P.out_gs = 1;
      P.tiles_lo = S[p] / G;
      P.in_hi_stride = P.out_hi_stride = np * S[p];
      P.in_lo_stride = P.out_lo_stride = G;
      P.load_x_fastest = (G == 1);
      P.tw_mul = (R == 1) ? 0 : Tm[p];
      P.tw_h = T->h;
      P.tw_full = (full_tw && p == 0) ? 1u : 0u;  // (p == 0 of R == 2: Tm = 1, hi = 0, so the output position is k * S + col)
      tiles = Tm[p] * P.tiles_lo;
    } else {
      // last pass: G rows with adjacent k_1; hi = k_1 group, lo = middle digit (R == 3) else 0
      uint64_t N1 = 1ull << b[0];
      uint64_t G = tile_elems / np;
      if (G < 1) G = 1;
      if (G > N1) G = N1;
      while (G > 1 && n / (np * G) < 256) G >>= 1;
      P.g = (uint32_t)G;
      P.in_xs = 1;
      P.in_gs = S[0];
      P.out_xs = n >> b[p];
      P.out_gs = 1;
      uint64_t mid = (R == 3) ? (1ull << b[1]) : 1;
      P.tiles_lo = mid;
      P.in_hi_stride = G * S[0];
      P.in_lo_stride = np;      // middle digit k_2 sits at stride S[1] = N_3 = np
      P.out_hi_stride = G;
      P.out_lo_stride = N1;     // k_2 * T_2 = k_2 * N_1
      P.load_x_fastest = 1;
      P.tw_mul = 0;
      tiles = (N1 / G) * mid;
    }
    P.batch = batch;
    P.tiles = tiles;
    static const bool no_pair = std::getenv("MI355ZK_NTT_NOPAIR") != nullptr;
    static const char* pair_env = std::getenv("MI355ZK_NTT_PAIR");
    static const bool pair_all = std::getenv("MI355ZK_NTT_PAIR_ALL") != nullptr;
    // (narrow tiles only: with 128-byte runs and more the grouping is neutral -- measured with MI355ZK_NTT_PAIR_ALL)
    P.xcd_pair = ((P.g <= 2 || pair_all) && tiles % 256 == 0 && !no_pair) ? (pair_env ? (uint32_t)std::atoi(pair_env) : 5u) : 0u;
    if (p == 0 && Tpre) { P.pre = fold ? 2 : 1; P.pre_h = Tpre->h; }   // (fold + the wave-local kernel: 3, below)
    if (p == R - 1) {
      if (fold) P.post = Tpost ? 4 : 3;    // post_c (and post_g^k1) sit in the folded table
      else P.post = Tpost ? 2 : (post_c ? 1 : 3);
      P.post_h = Tpost ? Tpost->h : 0;
    }
    uint32_t pitch = (uint32_t)np;
    size_t lds_bytes = (size_t)P.g * pitch * 36;
    uint32_t threads = (uint32_t)((P.g * np) / 2);
    if (threads > NTT_THREADS) threads = NTT_THREADS;
    if (threads < 64) threads = 64;
    prof_begin(slot_pass, st);
    // (one instantiation per row length: static stage loops.  Radix-4 register butterflies for full tiles (transforms of 2^20 and
    // more), radix-2 for the narrow tiles of short transforms, whose passes are latency-bound and want two butterflies per lane
    // rather than half the lanes idle; env MI355ZK_NTT_RADIX = 2 / 4 forces one)
    static const char* radix_env = std::getenv("MI355ZK_NTT_RADIX");
    // a full tile (>= 2048 elements): a lane per group of four
    const bool r4 = radix_env ? std::atoi(radix_env) == 4 : (uint64_t)P.g * np >= 2048;
    if (r4) {
      threads = (uint32_t)((P.g * np) / 4);
      if (threads > NTT_THREADS) threads = NTT_THREADS;
      if (threads < 64) threads = 64;
    }
    // full tiles of rows of >= 256 elements: the wave-local kernel (round 5; env MI355ZK_NTT_WAVELOCAL=0: the barrier-per-pair kernel, for the A/B)
    static const bool no_wl = std::getenv("MI355ZK_NTT_WAVELOCAL") != nullptr && std::getenv("MI355ZK_NTT_WAVELOCAL")[0] == '0';
    const bool wl_kernel = r4 && !no_wl && b[p] >= 8 && (uint64_t)P.g * np == 4ull * threads;
    static const bool no_stage_fold = std::getenv("MI355ZK_NTT_NO_STAGE_FOLD") != nullptr;
    const UTab* k_preA_p = k_preA;
    const UTab* k_preB_p = k_preB;
    const UTab* k_postA_p = k_postA;
    const UTab* k_postB_p = k_postB;
    const UTab* k_twB_p = T->B;
    if (p == 0 && Tpre && fold && wl_kernel && !no_stage_fold) { P.pre = 3; k_preA_p = F.pre_stages; }
    if (fold_big && wl_kernel) {
      if (p == 0 && Tpre) { P.pre = 4; k_preA_p = F.pre_stages; k_preB_p = F.pre_cols; }
      if (p == R - 1 && Tpost) { P.post = 5; k_postA_p = F.post_rows; k_postB_p = F.post_rowc; }
    }
    // a transform scaled by post_c alone: the pass before the last multiplies it in with its twiddle, the last pass by nothing
    // (decided for both passes together: the last pass must be able to drop the product whichever kernel runs it -- post == 3 is in both)
    if (fold_big && post_c && !Tpost) {
      if (p == R - 2) k_twB_p = F.tw_b_scaled;
      if (p == R - 1) P.post = 3;
    }
#define ZK_NTT_LAUNCH_WL(L)                                                                                                                \
  case L:                                                                                                                                  \
    hipLaunchKernelGGL((ntt_pass_wl_kernel<L>), dim3((unsigned)(tiles * batch)), dim3(threads), lds_bytes, st, src, dst, P, T->roots[b[p]], T->A,      \
                       k_twB_p, k_preA_p, k_preB_p, k_postA_p, k_postB_p,    \
                       post_cu, k_full, BP);                                                                                                  \
    break;
    if (wl_kernel) {
      switch (b[p]) {
        ZK_NTT_LAUNCH_WL(8) ZK_NTT_LAUNCH_WL(9) ZK_NTT_LAUNCH_WL(10) ZK_NTT_LAUNCH_WL(11) ZK_NTT_LAUNCH_WL(12)
        default: return ZK_ERR_BAD_ARGS;
      }
    } else {
#define ZK_NTT_LAUNCH(L)                                                                                                                   \
  case L:                                                                                                                                  \
    if (r4)                                                                                                                                \
      hipLaunchKernelGGL((ntt_pass_kernel<L, true>), dim3((unsigned)(tiles * batch)), dim3(threads), lds_bytes, st, src, dst, P, T->roots[b[p]], T->A, \
                         k_twB_p, k_preA, k_preB, k_postA, k_postB,  \
                         post_cu, k_full, BP);                                                                                                \
    else                                                                                                                                   \
      hipLaunchKernelGGL((ntt_pass_kernel<L, false>), dim3((unsigned)(tiles * batch)), dim3(threads), lds_bytes, st, src, dst, P, T->roots[b[p]], T->A, \
                         k_twB_p, k_preA, k_preB, k_postA, k_postB,  \
                         post_cu, k_full, BP);                                                                                                \
    break;
    switch (b[p]) {
      ZK_NTT_LAUNCH(1) ZK_NTT_LAUNCH(2) ZK_NTT_LAUNCH(3) ZK_NTT_LAUNCH(4) ZK_NTT_LAUNCH(5) ZK_NTT_LAUNCH(6) ZK_NTT_LAUNCH(7) ZK_NTT_LAUNCH(8)
      ZK_NTT_LAUNCH(9) ZK_NTT_LAUNCH(10) ZK_NTT_LAUNCH(11) ZK_NTT_LAUNCH(12)
      default: return ZK_ERR_BAD_ARGS;
    }
    }
#undef ZK_NTT_LAUNCH
#undef ZK_NTT_LAUNCH_WL
    ZK_HIP(hipGetLastError());
    prof_end(slot_pass, st);
  }
  return 0;
}

int ntt_run(Fr* d_a, uint32_t log_n, const Fr& omega, hipStream_t st) { return ntt_run_scaled(d_a, log_n, omega, nullptr, nullptr, nullptr, st); }

// a[i] *= c * g^i  (g == nullptr: a[i] *= c): standalone elementwise form of distribute_powers
// (domain.rs:176-189) / the ifft scaling (domain.rs:163-173); the domain ops use the fused forms above.
int ntt_scale(Fr* d_a, uint32_t log_n, const Fr& c, const Fr* g, hipStream_t st) {
  const uint64_t n = 1ull << log_n;
  const UTab* A = nullptr;
  const UTab* B = nullptr;
  uint32_t h = 0;
  std::lock_guard<std::mutex> run_lk(g_run_mu);
  if (g != nullptr && log_n > 0) {
    PowTables* T = nullptr;
    int rc = tables_make_room(1);
    if (rc) return rc;
    rc = build_pow_tables(st, log_n, *g, false, nullptr, 0, &T);
    if (rc) return rc;
    A = T->A;
    B = T->B;
    h = T->h;
  }
  static const int slot_scale = prof_slot("ntt_scale");
  prof_begin(slot_scale, st);
  hipLaunchKernelGGL(ntt_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_a, n, c, A, B, h);
  ZK_HIP(hipGetLastError());
  prof_end(slot_scale, st);
  return 0;
}

}  // namespace zk
