// C ABI of libmi355zk.so (include/mi355zk.h): argument checking, the Source/Density contract of
// bellman/src/source.rs, H2D/D2H staging for the host-buffer entry points, domain constants of
// bellman/src/domain.rs:52-99, and the kernel-timing hooks used by bench.py.
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

namespace zk {
// ntt.hip
int ntt_run(Fr* d_a, uint32_t log_n, const Fr& omega, hipStream_t st);
int ntt_run_scaled(Fr* d_a, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st);
int ntt_run_batch(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st);
int ntt_scale(Fr* d_a, uint32_t log_n, const Fr& c, const Fr* g, hipStream_t st);
void ntt_release_all();
int ntt_configure();
// msm.hip
int msm_g1_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[12], long long* err_index, uint32_t wgroups, uint32_t wgroup, bool scalars_mont,
                  MsmChunks* chunks, uint64_t table_stride = 0, uint32_t table_c = 0);
void msm_table_geometry(uint64_t n_bases, int group, uint32_t* c, uint32_t* W, uint8_t width[64]);
int msm_g2_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[24], long long* err_index, uint32_t wgroups, uint32_t wgroup, bool scalars_mont,
                  MsmChunks* chunks, uint64_t table_stride = 0, uint32_t table_c = 0);
int msm_g1_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz, uint64_t* out2_xyz);
int msm_g2_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz, uint64_t* out2_xyz);
// point_fft.hip
int point_fft_g1(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st);
// codec.hip
int codec_decode(int group, void* d_out, const void* d_in, size_t n, int compressed, int checked, hipStream_t st, long long* err_index);
int codec_encode(int group, void* d_out, const void* d_in, size_t n, int compressed, hipStream_t st);
// point_fft_g2.hip
int point_fft_g2(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st, bool trusted_subgroup);
int segsum_g1_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out);
int segsum_g2_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out);
void msm_release_g1();
void msm_release_g2();
void msm_geometry(uint64_t n, uint32_t wgroups, uint32_t* c, uint32_t* W);
int msm_selftest_digits(uint64_t n, uint32_t wgroups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct, int32_t* digits,
                        uint32_t* geom);

// ------------------------------------------------------------------------------------------------
// profiling hooks
namespace {
struct ProfSlot {
  std::string name;
  double total_ms = 0;
  long count = 0;
  struct Pair {
    int dev;
    hipEvent_t a, b;
  };
  std::vector<Pair> pending;
  // the begin events waiting for their end, keyed by the STREAM they were recorded on: several host threads run msm_device at once (the
  // cells of the multi-GPU mode, the prover's eight multiexps), each on its own stream -- one `open` per slot paired thread A's end with
  // thread B's begin
  struct Open {
    hipStream_t st;
    hipEvent_t e;
    int dev;
  };
  std::vector<Open> open;
};
std::mutex g_prof_mu;
std::vector<ProfSlot> g_prof;
int g_prof_on = 0;                     // 0 off, 1 every slot, 2 only the slot named by prof_only (the dominant kernel: two events per call)
int g_prof_only = -1;
std::map<int, std::vector<hipEvent_t>> g_prof_free;   // per device; events are reused: creating one costs several microseconds on the launching thread
hipEvent_t prof_event(int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  *dev_out = dev;
  auto& pool = g_prof_free[dev];
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void prof_enable(int on) { g_prof_on = on; }
bool prof_enabled() { return g_prof_on != 0; }
int prof_slot(const char* name) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (size_t i = 0; i < g_prof.size(); ++i)
    if (g_prof[i].name == name) return (int)i;
  g_prof.emplace_back();
  g_prof.back().name = name;
  return (int)g_prof.size() - 1;
}
void prof_only(const char* name) { g_prof_only = name ? prof_slot(name) : -1; }
void prof_begin(int slot, hipStream_t st) {
  if (g_prof_on == 0 || (g_prof_on == 2 && slot != g_prof_only)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int dev = 0;
  hipEvent_t e = prof_event(&dev);
  if (e == nullptr) return;
  (void)hipEventRecord(e, st);
  for (auto& o : g_prof[slot].open)
    if (o.st == st && o.dev == dev) {     // (a begin without its end on this (device, stream): an error path.  Keyed by BOTH: two host threads on
                                          //  different devices may each use their device's NULL stream -- ADVICE r5)
      g_prof_free[o.dev].push_back(o.e);
      o.e = e;
      o.dev = dev;
      return;
    }
  g_prof[slot].open.push_back(ProfSlot::Open{st, e, dev});
}
void prof_end(int slot, hipStream_t st) {
  if (g_prof_on == 0 || (g_prof_on == 2 && slot != g_prof_only)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  auto& open = g_prof[slot].open;
  int dev = 0;
  hipEvent_t e = prof_event(&dev);
  if (e == nullptr) return;
  size_t k = 0;
  while (k < open.size() && !(open[k].st == st && open[k].dev == dev)) ++k;
  if (k == open.size()) {               // no begin on this (device, stream)
    g_prof_free[dev].push_back(e);
    return;
  }
  (void)hipEventRecord(e, st);
  g_prof[slot].pending.push_back(ProfSlot::Pair{dev, open[k].e, e});
  open.erase(open.begin() + (long)k);
}
void prof_collect() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof) {
    for (auto& pr : s.pending) {
      if (hipEventSynchronize(pr.b) == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
          s.total_ms += ms;
          s.count += 1;
        }
      }
      g_prof_free[pr.dev].push_back(pr.a);
      g_prof_free[pr.dev].push_back(pr.b);
    }
    s.pending.clear();
  }
}
bool prof_get(const char* name, double* total_ms, long* count) {
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof)
    if (s.name == name) {
      *total_ms = s.total_ms;
      *count = s.count;
      return true;
    }
  return false;
}
void prof_release_all() {   // mi355zk_shutdown: the pooled events go back to their devices
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& kv : g_prof_free) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    for (hipEvent_t e : kv.second) (void)hipEventDestroy(e);
  }
  g_prof_free.clear();
}
void prof_reset() {
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof) {
    s.total_ms = 0;
    s.count = 0;
  }
}

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
              const uint32_t* d_base_index = nullptr, bool shortcut_unit_scalars = false, bool g2_trusted = false,
              const uint8_t* d_g2_member = nullptr) {
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

// ------------------------------------------------------------------------------------------------
// domain constants (host arithmetic, same field code as the kernels)
namespace {

Fr fr_from_u64(uint64_t v) {
  Fr c = Fr::zero();
  c.l[0] = (uint32_t)v;
  c.l[1] = (uint32_t)(v >> 32);
  return from_canonical(c);
}

// ff_derive: ROOT_OF_UNITY = GENERATOR^((r-1) >> S) with GENERATOR = 7, S = 28 (fr.rs:3-6,31-34)
Fr fr_root_of_unity() {
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) e[i] = FrParams::P[i];
  e[0] -= 1;
  for (int i = 0; i < 8; ++i) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
  return pow_limbs(fr_from_u64(7), e, 8);
}

struct DomainConsts {
  Fr omega, omegainv, geninv, minv, gen;
};

// EvaluationDomain::from_coeffs for m = 2^exp (domain.rs:61-98)
int domain_consts(uint32_t exp, DomainConsts* d) {
  if (exp > 28) return ZK_ERR_BAD_ARGS;  // PolynomialDegreeTooLarge (domain.rs:75-77)
  static const Fr rou = fr_root_of_unity();
  Fr w = rou;
  for (uint32_t i = exp; i < 28; ++i) w = sqr(w);
  d->omega = w;
  d->omegainv = inv(w);
  d->gen = fr_from_u64(7);
  d->geninv = inv(d->gen);
  d->minv = inv(fr_from_u64(1ull << exp));
  return ZK_OK;
}

int domain_op_dev(Fr* d_a, uint32_t log_n, int op, hipStream_t st) {
  DomainConsts D;
  int rc = domain_consts(log_n, &D);
  if (rc) return rc;
  switch (op) {
    case MI355ZK_OP_FFT:  // domain.rs:154-157
      return ntt_run(d_a, log_n, D.omega, st);
    case MI355ZK_OP_IFFT:  // domain.rs:159-174: best_fft(omegainv), then *= minv (fused into the last pass)
      return ntt_run_scaled(d_a, log_n, D.omegainv, nullptr, &D.minv, nullptr, st);
    case MI355ZK_OP_COSET_FFT:  // domain.rs:191-195: distribute_powers(g) (fused into the first pass), then fft
      return ntt_run_scaled(d_a, log_n, D.omega, &D.gen, nullptr, nullptr, st);
    case MI355ZK_OP_ICOSET_FFT:  // domain.rs:197-203: ifft, then distribute_powers(geninv) (both fused into the last pass)
      return ntt_run_scaled(d_a, log_n, D.omegainv, nullptr, &D.minv, &D.geninv, st);
    default:
      return ZK_ERR_BAD_ARGS;
  }
}

// the same operation on `batch` arrays of one size: one launch per pass over all of them (ntt.hip: ntt_run_batch), eight arrays at a time
int domain_op_batch_dev(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, int op, hipStream_t st) {
  DomainConsts D;
  int rc = domain_consts(log_n, &D);
  if (rc) return rc;
  // from 2^21 on a pass has more tiles than the device has workgroup slots and pipelines by itself; a joint launch only makes the
  // transforms compete for the Infinity Cache (2^22: 0.459 -> 0.465 ms per transform): one at a time there
  const uint32_t per_launch = log_n <= 20 ? 8u : 1u;
  for (uint32_t done = 0; done < batch; done += per_launch) {
    const uint32_t k = batch - done < per_launch ? batch - done : per_launch;
    switch (op) {
      case MI355ZK_OP_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omega, nullptr, nullptr, nullptr, st); break;
      case MI355ZK_OP_IFFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omegainv, nullptr, &D.minv, nullptr, st); break;
      case MI355ZK_OP_COSET_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omega, &D.gen, nullptr, nullptr, st); break;
      case MI355ZK_OP_ICOSET_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omegainv, nullptr, &D.minv, &D.geninv, st); break;
      default: return ZK_ERR_BAD_ARGS;
    }
    if (rc) return rc;
  }
  return ZK_OK;
}

thread_local long long t_last_err_index = -1;

// Source / QueryDensity contract (source.rs:36-118, multiexp.rs:92): returns the number of exponents
// to process, the exponent index of the first UnexpectedEof (or -1) and, for a density map, the
// per-word exclusive prefix popcounts.
struct DensityPlan {
  uint64_t n = 0;
  long long eof_index = -1;
  std::vector<uint32_t> prefix;
};

int plan_density(size_t n_bases, size_t base_offset, size_t n_scalars, const uint32_t* density, size_t density_bits, DensityPlan* P) {
  uint64_t n = n_scalars;
  if (density != nullptr && density_bits < n) n = density_bits;  // zip() stops at the shorter (multiexp.rs:92)
  P->n = n;
  uint64_t avail = base_offset < n_bases ? n_bases - base_offset : 0;
  if (density == nullptr) {
    if (n > avail) P->eof_index = (long long)avail;
    return ZK_OK;
  }
  uint64_t words = (n + 31) / 32;
  P->prefix.resize(words ? words : 1);
  uint64_t used = 0;
  for (uint64_t w = 0; w < words; ++w) {
    P->prefix[w] = (uint32_t)used;
    uint32_t v = density[w];
    if (w == words - 1 && (n & 31)) v &= (1u << (n & 31)) - 1u;
    uint32_t pc = (uint32_t)__builtin_popcount(v);
    if (P->eof_index < 0 && used + pc > avail) {
      // the (avail - used + 1)-th set bit of this word is the first exponent without a base
      uint64_t need = avail - used;
      for (uint32_t b = 0; b < 32; ++b)
        if ((v >> b) & 1) {
          if (need == 0) { P->eof_index = (long long)(w * 32 + b); break; }
          --need;
        }
    }
    used += pc;
  }
  return ZK_OK;
}

// device copies of density maps (words + prefix popcounts): grow-only buffers, leased per call
struct DensityPool {
  struct Buf {
    int dev = -1;
    void* p = nullptr;
    size_t bytes = 0;
    bool busy = false;
  };
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::vector<Buf*>& all() { static std::vector<Buf*> v; return v; }
  struct Lease {
    Buf* b = nullptr;
    hipStream_t st = nullptr;
    int acquire(int dev, size_t bytes, hipStream_t stream) {
      st = stream;
      {
        std::lock_guard<std::mutex> lk(mu());
        for (Buf* x : all())  // the smallest idle buffer that fits, else the largest idle one (regrown below)
          if (!x->busy && x->dev == dev) {
            if (b == nullptr) { b = x; continue; }
            const bool fits = x->bytes >= bytes, bfits = b->bytes >= bytes;
            if (fits ? (!bfits || x->bytes < b->bytes) : (!bfits && x->bytes > b->bytes)) b = x;
          }
        if (b == nullptr) {
          b = new Buf();
          b->dev = dev;
          all().push_back(b);
        }
        b->busy = true;
      }
      if (b->bytes < bytes) {
        if (b->p) ZK_HIP(hipFree(b->p));  // idle: its last user's stream was synchronised before the release
        b->p = nullptr;
        b->bytes = 0;
        ZK_HIP(hipMalloc(&b->p, bytes));
        b->bytes = bytes;
      }
      return ZK_OK;
    }
    ~Lease() {
      if (b == nullptr) return;
      (void)hipStreamSynchronize(st);  // (idle already after a completed call: the result came back over this stream)
      std::lock_guard<std::mutex> lk(mu());
      b->busy = false;
    }
  };
};

template <int GROUP>
int msm_dev_entry(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                  const uint32_t* density, size_t density_bits, void* stream, uint64_t* out_xyz, uint32_t wgroups = 1, uint32_t wgroup = 0,
                  uint32_t flags = 0, MsmChunks* chunks = nullptr, bool table = false) {
  // table: d_bases is the window table msm_table_build made of a vector of n_bases points (table mode, msm_impl.hpp)
  // chunks != nullptr: the exponents are handed over chunk by chunk while the call runs (msm_host_entry); d_scalars is unused
  t_last_err_index = -1;
  if (!out_xyz || (n_scalars && !d_scalars && !chunks) || (n_bases && !d_bases)) return ZK_ERR_BAD_ARGS;
  if (n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  hipStream_t st = (hipStream_t)stream;
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;  // exponents before the first Eof
  uint32_t* d_density = nullptr;
  uint32_t* d_prefix = nullptr;
  DensityPool::Lease density_lease;
  if (density != nullptr && n > 0) {
    // leased from a small pool for the duration of the call: hipMalloc / hipFree per call would synchronise the whole device and
    // with it every other thread's multiexp, and a buffer per host thread would outlive short-lived caller threads
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    size_t words = (n + 31) / 32;
    rc = density_lease.acquire(dev, words * 8, st);
    if (rc) return rc;
    DensityPool::Buf& buf = *density_lease.b;
    d_density = (uint32_t*)buf.p;
    d_prefix = d_density + words;
    ZK_HIP(hipMemcpyAsync(d_density, density, words * 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_prefix, P.prefix.data(), words * 4, hipMemcpyHostToDevice, st));
  }
  long long err_index = -1;
  const bool mont = (flags & MI355ZK_MSM_SCALARS_MONTGOMERY) != 0;
  if (chunks && (chunks->n_chunks == 0 || chunks->cuts[chunks->n_chunks] != n)) return ZK_ERR_BAD_ARGS;
  uint32_t tc = 0, tW = 0;
  if (table) msm_table_geometry(n_bases, GROUP, &tc, &tW, nullptr);
  const uint64_t tstride = table ? (uint64_t)n_bases : 0;
  if (table && n_bases == 0 && n > 0) return ZK_ERR_BAD_ARGS;
  if (GROUP == 1) rc = msm_g1_device(d_bases, n_bases, base_offset, d_scalars, n, d_density, d_prefix, st, out_xyz, &err_index, wgroups, wgroup, mont, chunks, tstride, tc);
  else rc = msm_g2_device(d_bases, n_bases, base_offset, d_scalars, n, d_density, d_prefix, st, out_xyz, &err_index, wgroups, wgroup, mont, chunks, tstride, tc);
  if (rc == ZK_ERR_UNEXPECTED_IDENTITY) {
    // the kernels report the lowest BASE index that was the identity under a non-zero exponent; the exponent that owns it
    // is the (index - base_offset)-th selected one (source.rs:101-118): itself under FullDensity
    long long rank = err_index - (long long)base_offset;
    if (density != nullptr) {
      size_t w = 0;
      const size_t words = (n + 31) / 32;
      while (w + 1 < words && (long long)P.prefix[w + 1] <= rank) ++w;
      uint32_t word = density[w];
      if ((w + 1) * 32 > n) word &= (n & 31) ? ((1u << (n & 31)) - 1u) : 0xffffffffu;
      long long need = rank - (long long)P.prefix[w];
      uint32_t b = 0;
      for (; b < 32; ++b)
        if ((word >> b) & 1u) { if (need == 0) break; --need; }
      rank = (long long)(w * 32 + b);
    }
    t_last_err_index = rank;
    return rc;
  }
  if (rc == ZK_ERR_BAD_ARGS) t_last_err_index = err_index;  // a non-canonical exponent (>= 2^254): its index
  if (rc != ZK_OK) return rc;
  if (P.eof_index >= 0) { t_last_err_index = P.eof_index; return ZK_ERR_UNEXPECTED_EOF; }
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-buffer entry points (SURVEY 8b "Ownership"): the caller's bases and scalars live in (pageable) host memory.
//   * BASES CACHE: the CRS / tau-table is reused across calls (`Arc<Vec<G>>` inside groth16::Parameters, groth16/mod.rs:216-238),
//     so the device copy of a base vector the caller has PINNED (mi355zk_bases_cache_pin: "this host vector is immutable until
//     I invalidate it" -- the shim holds a clone of the Arc, so the allocation can neither be rewritten nor freed and reused)
//     stays on the device, keyed by (host pointer, length, group) with a fingerprint of sampled records as a safety net;
//     LRU-bounded (env MI355ZK_BASES_CACHE_GB, default 64; 0 disables).  Vectors that were not pinned are uploaded on every call
//     (env MI355ZK_BASES_CACHE_IMPLICIT=1 restores round 2's behaviour: every vector is treated as pinned).
//   * STREAMED UPLOAD: a large call is cut into chunks of ~2^24 exponents; a copy thread uploads chunk i + 1 (its scalars into
//     one of two staging buffers, its bases -- when they are not cached yet -- straight into the cache entry) on a copy stream
//     while the calling thread runs the multiexp of chunk i on a compute stream; the Jacobian partials are added on the host.
//     PCIe and the kernels overlap; the first call is bound by the link (96 B per exponent), later calls by the kernels.

struct BasesEntry {
  const void* host = nullptr;   // first record of what is cached: the pinned vector itself, or the SLICE of it a multi-GPU cell consumes
  const void* owner = nullptr;  // the pinned vector the records belong to (== host unless a slice): what invalidate / info are asked about
  size_t n = 0;
  int group = 0, dev = 0;
  uint64_t fp = 0;
  void* d = nullptr;
  size_t bytes = 0;
  uint64_t tick = 0;
  bool ready = false;      // fully uploaded
  std::mutex fill_mu;      // held by the call that uploads it
  // the vector's WINDOW TABLE (table mode, msm_impl.hpp), for vectors pinned with mi355zk_bases_cache_pin_tables: built by the first
  // call that finds the entry ready, counted against the cache's capacity, freed with the entry
  bool want_table = false, table_failed = false;
  void* table = nullptr;
  size_t table_bytes = 0;
  size_t table_reserved = 0;  // bytes set aside under g_bc_mu while the table is being built (concurrent builds cannot overbook the cache)
  std::mutex table_mu;
};
std::mutex g_bc_mu;
std::vector<std::shared_ptr<BasesEntry>> g_bc;
uint64_t g_bc_tick = 0;

uint64_t fnv1a(uint64_t h, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
// first / last 4 KiB and 4096 records spread over the array: cheap (~0.3 MB hashed), and a different CRS at the same address
// is caught; a few records rewritten IN PLACE are not -- which is why caching is OPT-IN: only vectors the caller pinned (declared
// immutable) are served from the device copy, the fingerprint is a second line of defence, not the contract
uint64_t bases_fingerprint(const uint8_t* p, size_t bytes, size_t rec) {
  uint64_t h = 0xcbf29ce484222325ull;
  const size_t edge = bytes < 4096 ? bytes : 4096;
  h = fnv1a(h, p, edge);
  h = fnv1a(h, p + bytes - edge, edge);
  const size_t nrec = bytes / rec;
  for (size_t k = 1; k <= 4096 && nrec > 0; ++k) h = fnv1a(h, p + (nrec * k / 4097) * rec, rec);
  return h;
}
size_t bases_cache_cap() {
  static const char* env = std::getenv("MI355ZK_BASES_CACHE_GB");
  const double gb = env ? std::atof(env) : 64.0;
  return gb <= 0 ? 0 : (size_t)(gb * 1073741824.0);
}
// the vectors the caller declared immutable (mi355zk_bases_cache_pin)
struct BasesPin {
  const void* host;
  size_t n;
  int group;
  bool tables;
};
std::vector<BasesPin> g_bc_pins;  // under g_bc_mu
bool bases_cache_implicit() {
  static const char* env = std::getenv("MI355ZK_BASES_CACHE_IMPLICIT");
  return env && env[0] == '1';
}
int bases_cache_pin(const void* host, size_t n, int group, bool tables = false) {
  if (!host || n == 0 || (group != 1 && group != 2)) return ZK_ERR_BAD_ARGS;
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (auto& p : g_bc_pins)
    if (p.host == host && p.n == n && p.group == group) {
      p.tables = p.tables || tables;
      return ZK_OK;
    }
  g_bc_pins.push_back(BasesPin{host, n, group, tables});
  return ZK_OK;
}
// returns the entry (locked for filling when *fill == true: the caller uploads and then sets ready) or nullptr (cache off / not
// pinned / no room)
// the vector a multi-GPU cell's slice was cut from (set by the cell's thread around its msm_host_run): the owner of an IMPLICITLY cached
// slice, so that mi355zk_bases_cache_invalidate(vector) reaches the slices on every device (a pinned vector's slices find
// their owner in the pin list)
thread_local const void* t_bases_parent = nullptr;
std::shared_ptr<BasesEntry> bases_lookup(const void* host, size_t n, int group, size_t bytes, int dev, bool* fill) {
  *fill = false;
  const size_t cap = bases_cache_cap();
  if (cap == 0 || bytes > cap) return nullptr;
  bool want_table = false;
  const void* owner = t_bases_parent ? t_bases_parent : host;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    // pinned: the vector itself, or a record range INSIDE a pinned vector (the single-process multi-GPU mode caches on each device only
    // the slice its cell consumes: SURVEY 8e "the tau-table slice stays resident on its GPU")
    bool pinned = false;
    const size_t rec = group == 1 ? 64 : 128;
    for (auto& p : g_bc_pins) {
      if (p.group != group) continue;
      const char* lo = (const char*)p.host;
      if ((const char*)host >= lo && (const char*)host + n * rec <= lo + p.n * rec) {
        pinned = true;
        owner = p.host;
        want_table = want_table || (p.tables && p.host == host && p.n == n);   // (tables for whole vectors only: a slice's calls are cells)
      }
    }
    if (!pinned && !bases_cache_implicit()) return nullptr;
  }
  const uint64_t fp = bases_fingerprint((const uint8_t*)host, bytes, group == 1 ? 64 : 128);
  std::shared_ptr<BasesEntry> hit;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc)
      if (e->host == host && e->n == n && e->group == group && e->dev == dev && e->fp == fp) { hit = e; break; }
    if (hit) { hit->tick = ++g_bc_tick; hit->want_table = hit->want_table || want_table; }
  }
  if (hit) {
    std::lock_guard<std::mutex> wait_fill(hit->fill_mu);  // another thread may still be uploading it
    if (hit->ready) return hit;
    return nullptr;                                       // its upload failed: go uncached
  }
  auto e = std::make_shared<BasesEntry>();
  e->host = host; e->owner = owner; e->n = n; e->group = group; e->dev = dev; e->fp = fp; e->bytes = bytes; e->want_table = want_table;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    // the capacity is PER DEVICE (a process may drive several: mi355zk_init with n_devices > 1 keeps a copy of a pinned vector on
    // every device that evaluates cells over it)
    size_t used = 0;
    for (auto& x : g_bc)
      if (x->dev == dev) used += x->bytes + x->table_bytes + x->table_reserved;
    while (used + bytes > cap && !g_bc.empty()) {           // evict least recently used entries nobody is filling
      size_t victim = g_bc.size();
      for (size_t i = 0; i < g_bc.size(); ++i)
        if (g_bc[i]->dev == dev && g_bc[i]->ready && g_bc[i].use_count() == 1 && g_bc[i]->table_reserved == 0 &&
            (victim == g_bc.size() || g_bc[i]->tick < g_bc[victim]->tick))
          victim = i;
      if (victim == g_bc.size()) break;
      (void)hipFree(g_bc[victim]->d);
      (void)hipFree(g_bc[victim]->table);
      used -= g_bc[victim]->bytes + g_bc[victim]->table_bytes;
      g_bc.erase(g_bc.begin() + (long)victim);
    }
    if (used + bytes > cap) return nullptr;
    if (hipMalloc(&e->d, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e->tick = ++g_bc_tick;
    e->fill_mu.lock();
    g_bc.push_back(e);
  }
  *fill = true;
  return e;
}
void bases_drop(const std::shared_ptr<BasesEntry>& e) {  // a failed upload
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (size_t i = 0; i < g_bc.size(); ++i)
    if (g_bc[i] == e) { g_bc.erase(g_bc.begin() + (long)i); break; }
  (void)hipFree(e->d);
  (void)hipFree(e->table);
  e->d = e->table = nullptr;
  e->table_bytes = 0;
}

// two staging buffers for scalar chunks, a bases buffer for uncached calls, the two streams.  Leased from a pool for the duration
// of a call (callers come and go -- the prover queues its multiexps from short-lived threads -- and their buffers must not pile up)
struct HostStage {
  int dev = -1;
  bool busy = false;
  void* sc[2] = {nullptr, nullptr};
  size_t sc_bytes = 0;
  void* bases = nullptr;
  size_t bases_bytes = 0;
  hipStream_t copy = nullptr, compute = nullptr;
};
std::mutex g_stage_mu;
std::vector<HostStage*> g_stages;  // the pool: as many stages as there have been concurrent host-buffer calls
struct StageLease {
  HostStage* s = nullptr;
  ~StageLease() {
    if (s == nullptr) return;
    // every exit of msm_host_entry has joined its copy thread; the compute stream is idle after the last chunk's result came back,
    // except on an error path
    (void)hipStreamSynchronize(s->compute);
    (void)hipStreamSynchronize(s->copy);
    std::lock_guard<std::mutex> lk(g_stage_mu);
    s->busy = false;
  }
};
HostStage* host_stage(int dev, StageLease* lease) {
  HostStage* mine = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (HostStage* s : g_stages)  // the idle stage of this device with the largest staging buffers
      if (!s->busy && s->dev == dev && (mine == nullptr || s->sc_bytes > mine->sc_bytes)) mine = s;
    if (mine) mine->busy = true;
  }
  if (mine == nullptr) {
    mine = new HostStage();
    mine->dev = dev;
    mine->busy = true;
    if (hipStreamCreateWithFlags(&mine->copy, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&mine->compute, hipStreamNonBlocking) != hipSuccess) {
      delete mine;
      return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_stage_mu);
    g_stages.push_back(mine);
  }
  lease->s = mine;
  return mine;
}
int stage_reserve(void** p, size_t* have, size_t want) {
  if (*have >= want) return ZK_OK;
  if (*p) ZK_HIP(hipFree(*p));
  *p = nullptr;
  *have = 0;
  ZK_HIP(hipMalloc(p, want));
  *have = want;
  return ZK_OK;
}

// forget the device copies of the base vector at `host` (nullptr: of every vector); entries in use stay until their call ends
void bases_cache_invalidate(const void* host) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (size_t i = 0; i < g_bc_pins.size();) {  // the promise of immutability ends here
    if (host == nullptr || g_bc_pins[i].host == host) g_bc_pins.erase(g_bc_pins.begin() + (long)i);
    else ++i;
  }
  for (size_t i = 0; i < g_bc.size();) {
    if ((host == nullptr || g_bc[i]->host == host || g_bc[i]->owner == host) && g_bc[i]->ready && g_bc[i].use_count() == 1) {
      (void)hipSetDevice(g_bc[i]->dev);
      (void)hipFree(g_bc[i]->d);
      (void)hipFree(g_bc[i]->table);
      g_bc.erase(g_bc.begin() + (long)i);
    } else {
      if (host == nullptr || g_bc[i]->host == host || g_bc[i]->owner == host) g_bc[i]->fp ^= 0x9e3779b97f4a7c15ull;  // in use: never matched again
      ++i;
    }
  }
  (void)hipSetDevice(cur);
}

void host_entry_release_all() {
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc) { (void)hipSetDevice(e->dev); (void)hipFree(e->d); (void)hipFree(e->table); }
    g_bc.clear();
  }
  std::lock_guard<std::mutex> lk(g_stage_mu);
  for (HostStage* s : g_stages) {
    (void)hipSetDevice(s->dev);
    (void)hipFree(s->sc[0]); (void)hipFree(s->sc[1]); (void)hipFree(s->bases);
    s->sc[0] = s->sc[1] = s->bases = nullptr;
    s->sc_bytes = s->bases_bytes = 0;
  }
  std::lock_guard<std::mutex> dl(DensityPool::mu());
  for (DensityPool::Buf* b : DensityPool::all()) {
    (void)hipSetDevice(b->dev);
    (void)hipFree(b->p);
    b->p = nullptr;
    b->bytes = 0;
  }
}

// the window table of a ready cache entry whose vector was pinned with tables: built by the first call that asks (the others wait on
// table_mu), inside the cache's capacity (no eviction for it: a table that does not fit is not built and the calls stay plain)
template <int GROUP>
const void* bases_table(const std::shared_ptr<BasesEntry>& e, hipStream_t st) {
  if (!e || !e->want_table || !e->ready) return nullptr;
  std::lock_guard<std::mutex> lk(e->table_mu);
  if (e->table) return e->table;
  if (e->table_failed) return nullptr;
  uint32_t c = 0, W = 0;
  msm_table_geometry(e->n, GROUP, &c, &W, nullptr);
  const size_t bytes = (size_t)W * e->bytes;
  if ((uint64_t)W * e->n > 0x7fffffffull) { e->table_failed = true; return nullptr; }
  {
    // reserve the room before the build: the prover's eight threads build the tables of different vectors at the same time.  A
    // cache that is full NOW is not a failure of this vector -- the next call asks again, after evictions may have made room.
    std::lock_guard<std::mutex> g(g_bc_mu);
    size_t used = 0;
    for (auto& x : g_bc)
      if (x->dev == e->dev) used += x->bytes + x->table_bytes + x->table_reserved;
    if (used + bytes > bases_cache_cap()) return nullptr;
    e->table_reserved = bytes;
  }
  void* t = nullptr;
  bool ok = hipMalloc(&t, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  if (ok && msm_table_build<GROUP>(e->d, e->n, t, bytes, (void*)st) != ZK_OK) { (void)hipFree(t); ok = false; }
  std::lock_guard<std::mutex> g(g_bc_mu);
  e->table_reserved = 0;
  if (!ok) { e->table_failed = true; return nullptr; }  // allocation or build failed: not tried again for this entry
  e->table = t;
  e->table_bytes = bytes;
  return t;
}

constexpr uint64_t HOST_CHUNK_UPLOAD = 1ull << 23;  // exponents per chunk of a streamed call whose bases travel too (link-bound)
constexpr uint64_t HOST_CHUNK_MIN = 1ull << 21;     // smallest first chunk of a call whose bases are on the device; below 4 of these the call is not cut

// One host-buffer multiexp on the calling thread's CURRENT device.  (wgroups, wgroup): only that group of scalar windows (a cell of
// the single-process multi-GPU mode below; (1, 0) is the whole multiexp).
template <int GROUP>
int msm_host_run(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                 const uint32_t* density, size_t density_bits, uint64_t* out_xyz, uint32_t wgroups = 1, uint32_t wgroup = 0) {
  t_last_err_index = -1;
  if (!out_xyz || (n_scalars && !scalars) || (n_bases && !bases)) return ZK_ERR_BAD_ARGS;
  if (n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  constexpr size_t bsz = GROUP == 1 ? 64 : 128;
  constexpr size_t jac_words = GROUP == 1 ? 12 : 24;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  StageLease stage_lease;
  HostStage* S = host_stage(dev, &stage_lease);
  if (S == nullptr) return ZK_ERR_DEVICE;

  // the exponents this call evaluates and the bases they consume (source.rs:36-118)
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  const uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;   // exponents before the first Eof
  auto rank_of = [&](uint64_t i) -> uint64_t {                          // bases consumed by exponents [0, i)
    if (density == nullptr) return i;
    if (i == 0) return 0;
    const uint64_t w = i >> 5;
    uint64_t r = w < P.prefix.size() ? P.prefix[w] : 0;
    if (w >= P.prefix.size()) {  // i == n on a word boundary past the last planned word
      const uint64_t lw = P.prefix.size() - 1;
      uint32_t v = density[lw];
      if ((lw + 1) * 32 > P.n) v &= (P.n & 31) ? ((1u << (P.n & 31)) - 1u) : 0xffffffffu;
      return P.prefix[lw] + (uint32_t)__builtin_popcount(v);
    }
    if (i & 31) r += (uint32_t)__builtin_popcount(density[w] & ((1u << (i & 31)) - 1u));
    return r;
  };

  // ---- bases: cached, being cached by this call, or (not pinned / cache off / full) the leased stage's buffer
  bool fill = false;
  std::shared_ptr<BasesEntry> entry = n_bases ? bases_lookup(bases, n_bases, GROUP, n_bases * bsz, dev, &fill) : nullptr;
  void* d_bases = entry ? entry->d : nullptr;
  bool upload_bases = fill;
  if (!entry && n_bases) {
    rc = stage_reserve(&S->bases, &S->bases_bytes, n_bases * bsz);
    if (rc) return rc;
    d_bases = S->bases;
    upload_bases = true;
  }
  struct FillGuard {  // whatever happens, the entry is either ready or gone when this call returns
    std::shared_ptr<BasesEntry> e;
    bool fill, ok = false;
    ~FillGuard() {
      if (!fill) return;
      e->ready = ok;
      e->fill_mu.unlock();
      if (!ok) bases_drop(e);
    }
  } guard{entry, fill};

  // ---- chunks (cut at multiples of 32 exponents, so that density words are not shared between chunks).  Every chunk runs digits ->
  // partition -> accumulate into the ONE bucket array of the call (msm_device, MsmChunks); what a chunk costs on top of its share
  // of the work is the re-partition of the bucket bounds and one read + write of every bucket record it touches (~1.3 ms at 2^26).
  std::vector<uint64_t> cuts{0};
  const char* env_grow = std::getenv("MI355ZK_HOST_CHUNK_GROWTH");  // percent (read per call: tools/exp_host_chunks.py sweeps it in one process)
  const char* env_first = std::getenv("MI355ZK_HOST_CHUNK_FIRST");  // log2 of the first chunk (bases on the device)
  // (test hook, read on every call: MI355ZK_HOST_CHUNK_TEST = exponents per chunk, a multiple of 32 -- cuts calls of ANY size, so
  // that the chunked path can be held against the CPU oracle at sizes the oracle finishes in seconds)
  const char* env_test = std::getenv("MI355ZK_HOST_CHUNK_TEST");
  const uint64_t test_chunk = env_test ? (uint64_t)std::strtoull(env_test, nullptr, 10) & ~31ull : 0;
  if (test_chunk >= 32) {
    for (uint64_t lo = test_chunk; lo < n; lo += test_chunk) cuts.push_back(lo);
  } else if (n >= 4 * HOST_CHUNK_MIN) {
    if (upload_bases) {
      // Bases travelling too (96 B per exponent): the link is the bottleneck and the kernels of a chunk finish long before the
      // next one has arrived; even chunks, small enough that the last one's kernels are a short tail behind the last byte.
      uint64_t k = (n + HOST_CHUNK_UPLOAD - 1) / HOST_CHUNK_UPLOAD;
      if (k < 2) k = 2;
      const uint64_t per = ((n + k - 1) / k + 31) & ~31ull;
      for (uint64_t lo = per; lo < n; lo += per) cuts.push_back(lo);
    } else {
      // Bases on the device: the kernels are the bottleneck (~1 G exponents/s against ~1.7 G/s of link).  Only the FIRST chunk's
      // upload is exposed, so it is small; each following chunk may be ~1.8 x the previous one and still arrive before the
      // kernels of its predecessor are done.
      const double grow = env_grow && std::atoi(env_grow) >= 100 ? std::atoi(env_grow) / 100.0 : 1.8;
      uint64_t sz = n / 20 > HOST_CHUNK_MIN ? n / 20 : HOST_CHUNK_MIN;
      if (env_first && std::atoi(env_first) >= 16 && std::atoi(env_first) <= 30) sz = 1ull << std::atoi(env_first);
      sz = (sz + 31) & ~31ull;
      uint64_t lo = 0;
      while (n - lo > sz + sz / 2) {  // the last chunk takes what is left, up to 1.5 x the next size
        lo += sz;
        cuts.push_back(lo);
        sz = ((uint64_t)((double)sz * grow) + 31) & ~31ull;
      }
    }
  }
  if (n) cuts.push_back(n);
  const uint64_t n_chunks = cuts.size() - 1;
  uint64_t max_chunk = 0;
  for (uint64_t c = 0; c < n_chunks; ++c) max_chunk = std::max(max_chunk, cuts[c + 1] - cuts[c]);
  const size_t sc_bytes = (size_t)max_chunk * 32;
  if (n) {
    for (int k = 0; k < 2; ++k) {
      size_t have = S->sc_bytes;
      rc = stage_reserve(&S->sc[k], &have, sc_bytes);
      if (rc) { S->sc_bytes = 0; return rc; }
    }
    if (S->sc_bytes < sc_bytes) S->sc_bytes = sc_bytes;
  }

  // The copy thread: for chunk c, scalars -> staging[c & 1] and (when uploading) the bases the chunk consumes; afterwards the
  // bases outside the consumed range, so that a cache entry is complete.  staging[c & 1] is free again once the DIGIT kernel of
  // chunk c - 2 -- the only reader of a chunk's exponents -- has run: the compute side records an event behind it.
  struct Feed : MsmChunks {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t copied = 0, digits = 0;          // chunks uploaded / chunks whose digit kernel has been enqueued
    bool copy_failed = false, abort_copy = false;
    void* sc[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;               // ev[c]: recorded behind chunk c's digit kernel
    int acquire(uint32_t c, hipStream_t, const void** d) override {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return copy_failed || copied > c; });  // (a host-side wait: the earlier chunks' kernels are already queued)
      if (copy_failed) return ZK_ERR_DEVICE;
      *d = sc[c & 1];
      return ZK_OK;
    }
    int digits_enqueued(uint32_t c, hipStream_t st) override {
      ZK_HIP(hipEventRecord(ev[c], st));
      std::lock_guard<std::mutex> lk(mu);
      digits = c + 1;
      cv.notify_all();
      return ZK_OK;
    }
    ~Feed() override {
      for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
  } feed;
  feed.n_chunks = (uint32_t)n_chunks;
  feed.cuts = cuts.data();
  feed.sc[0] = S->sc[0];
  feed.sc[1] = S->sc[1];
  feed.ev.reserve(n_chunks);
  for (uint64_t c = 0; c < n_chunks; ++c) {
    hipEvent_t e;
    ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    feed.ev.push_back(e);
  }
  const uint64_t b_lo = base_offset < n_bases ? base_offset : n_bases;
  static const bool trace = std::getenv("MI355ZK_TRACE_HOST") != nullptr;  // timeline of the streamed call on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  auto copy_fn = [&]() {
    auto fail = [&] { std::lock_guard<std::mutex> lk(feed.mu); feed.copy_failed = true; feed.cv.notify_all(); };
    if (hipSetDevice(dev) != hipSuccess) { fail(); return; }
    uint64_t b_done = b_lo;  // bases [b_lo, b_done) are on the device
    for (uint64_t c = 0; c < n_chunks; ++c) {
      if (c >= 2) {
        {
          std::unique_lock<std::mutex> lk(feed.mu);
          feed.cv.wait(lk, [&] { return feed.abort_copy || feed.digits >= c - 1; });
          if (feed.abort_copy) return;
        }
        if (hipEventSynchronize(feed.ev[c - 2]) != hipSuccess) { fail(); return; }
      } else {
        std::lock_guard<std::mutex> lk(feed.mu);
        if (feed.abort_copy) return;
      }
      const uint64_t lo = cuts[c], hi = cuts[c + 1];
      hipError_t e = hipMemcpyAsync(S->sc[c & 1], scalars + lo * 4, (hi - lo) * 32, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess && upload_bases) {
        uint64_t b_hi = base_offset + rank_of(hi);
        if (b_hi > n_bases) b_hi = n_bases;
        if (b_hi > b_done) {
          e = hipMemcpyAsync((char*)d_bases + b_done * bsz, bases + b_done * bsz, (b_hi - b_done) * bsz, hipMemcpyHostToDevice, S->copy);
          b_done = b_hi;
        }
      }
      if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
      if (e != hipSuccess) { fail(); return; }
      if (trace) std::fprintf(stderr, "[mi355zk] host entry: chunk %llu (%llu exponents) uploaded at %.2f ms\n", (unsigned long long)c, (unsigned long long)(hi - lo), ms_now());
      std::lock_guard<std::mutex> lk(feed.mu);
      feed.copied = c + 1;
      feed.cv.notify_all();
    }
    if (upload_bases && entry) {  // the rest of the vector (not needed by this call) completes the cache entry
      hipError_t e = hipSuccess;
      if (b_lo > 0) e = hipMemcpyAsync(d_bases, bases, b_lo * bsz, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess && b_done < n_bases)
        e = hipMemcpyAsync((char*)d_bases + b_done * bsz, bases + b_done * bsz, (n_bases - b_done) * bsz, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
      if (e != hipSuccess) fail();
    }
  };

  uint64_t result_xyz[jac_words];
  {
    // a call that evaluates no exponent returns the reference's Projective::zero() = (0, 1, 0) (ec.rs:229-235), as msm_device does
    using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
    static_assert(sizeof(J) == sizeof result_xyz, "Jacobian layout");
    const J zero = J::zero();
    std::memcpy(result_xyz, &zero, sizeof result_xyz);
  }
  int result = ZK_OK;
  long long err_idx = -1;
  bool aborted = false;
  if (n_chunks > 0) {
    std::thread copier(copy_fn);
    // a vector pinned WITH TABLES, already on the device, in a call that is not cut: table mode
    // (not for a handful of exponents over a long vector -- the prover's input multiexps over its 2^20-point a / b queries: the
    // table's window width comes from the VECTOR's length, and zeroing + reducing 2^19 buckets for a few points costs more than the
    // plain call, which picks its window from n)
    const bool table_pays = n * 8 >= n_bases;
    const void* d_table = (n_chunks == 1 && entry && !fill && wgroups == 1 && table_pays) ? bases_table<GROUP>(entry, S->compute) : nullptr;
    result = msm_dev_entry<GROUP>(d_table ? d_table : d_bases, n_bases, base_offset, nullptr, n_scalars, density, density_bits, (void*)S->compute, result_xyz,
                                  wgroups, wgroup, 0, &feed, d_table != nullptr);
    err_idx = t_last_err_index;
    if (trace) std::fprintf(stderr, "[mi355zk] host entry: result at %.2f ms (%llu chunks)\n", ms_now(), (unsigned long long)n_chunks);
    {
      std::lock_guard<std::mutex> lk(feed.mu);
      // a call that failed before it had taken every chunk leaves the copy thread waiting: release it
      aborted = result != ZK_OK && result != ZK_ERR_UNEXPECTED_EOF && feed.digits < n_chunks;
      feed.abort_copy = aborted;
      feed.cv.notify_all();
    }
    copier.join();
    if (feed.copy_failed) result = ZK_ERR_DEVICE;
  } else if (upload_bases && entry && n_bases) {
    // nothing to evaluate, but the entry was created: fill it
    hipError_t e = hipMemcpyAsync(d_bases, bases, n_bases * bsz, hipMemcpyHostToDevice, S->copy);
    if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
    if (e != hipSuccess) result = ZK_ERR_DEVICE;
    if (result == ZK_OK && P.eof_index >= 0) { result = ZK_ERR_UNEXPECTED_EOF; err_idx = P.eof_index; }
  } else if (P.eof_index >= 0) {
    result = ZK_ERR_UNEXPECTED_EOF;
    err_idx = P.eof_index;
  }
  guard.ok = result != ZK_ERR_DEVICE && !feed.copy_failed && !(fill && aborted);  // an aborted streamed upload is incomplete
  t_last_err_index = err_idx;
  if (result != ZK_OK && result != ZK_ERR_UNEXPECTED_EOF) return result;
  std::memcpy(out_xyz, result_xyz, sizeof result_xyz);
  return result;
}

// ------------------------------------------------------------------------------------------------
// SINGLE-PROCESS MULTI-GPU MODE.  The consumer this library is a drop-in for is ONE Rust process (phase2/src/bin/prove.rs ->
// bellman/src/groth16/prover.rs:250-298 -> multiexp.rs:330-355), so the 8 GPUs of a node must be reachable through the C ABI, not
// only through one rank per GPU (shard.py).  mi355zk_init(ids, n > 1) records a DEVICE SET; a host-buffer multiexp of at least
// 2^MI355ZK_MULTI_MIN_LOG exponents (default 20) is then cut into cells -- contiguous POINT RANGES (cut at multiples of 32
// exponents so that density words are not shared), optionally x groups of scalar windows -- and every cell is one msm_host_run on
// its own device from its own host thread: its exponents cross ITS PCIe link while its kernels run (the streamed upload above), its
// base vector is cached on that device when the caller pinned it.  The N Jacobian partials (96 / 192 B) come back to the host --
// SURVEY 8e's "or D2H of 8 records": inside one process there is nothing for RCCL to do -- and are joined there with the rule
// shard.exchange defines: a failing cell's error carries its GLOBAL exponent index, the lowest index wins, Eof (planned for the whole
// call) before identity at one index.  Smaller calls run whole, on the devices of the set in turn (the prover's eight concurrent
// multiexps spread over the node).
// Why point ranges and not shard.py's window groups: a rank of shard.py holds its exponents in HBM; here every cell uploads its own,
// and a window-group cell would upload ALL exponents of its range over its link (2^26 on 8 devices: 1 GiB per device against 256 MiB).
// MI355ZK_MULTI_PLAN="PxW" forces P point ranges x W window groups (P * W <= devices) for experiments and for the tests.
std::mutex g_devset_mu;
std::vector<int> g_devset;                 // HIP device ids of the set (a test may repeat one id: logical devices sharing a GPU)
std::atomic<unsigned> g_devset_turn{0};

std::vector<int> devset_snapshot() {
  std::lock_guard<std::mutex> lk(g_devset_mu);
  return g_devset;
}

// a cell / range / worker body run so that nothing is thrown out of a host thread or across the C ABI (std::bad_alloc from a
// staging vector, a std::system_error from a lock): the unit fails as a device error
template <class Fn>
void run_guarded(int& rc, Fn&& fn) noexcept {
  try {
    fn();
  } catch (...) {
    rc = ZK_ERR_DEVICE;
  }
}

struct DeviceGuard {  // the calling thread's current device is its own business: restore it
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// bases consumed by exponents [0, i) of a planned call (prefix popcount of the density map; i itself under FullDensity)
uint64_t density_rank(const DensityPlan& P, const uint32_t* density, uint64_t i) {
  if (density == nullptr || i == 0) return density == nullptr ? i : 0;
  const uint64_t w = i >> 5;
  if (w >= P.prefix.size()) {  // i == n on a word boundary past the last planned word
    const uint64_t lw = P.prefix.size() - 1;
    uint32_t v = density[lw];
    if ((lw + 1) * 32 > P.n) v &= (P.n & 31) ? ((1u << (P.n & 31)) - 1u) : 0xffffffffu;
    return P.prefix[lw] + (uint32_t)__builtin_popcount(v);
  }
  uint64_t r = P.prefix[w];
  if (i & 31) r += (uint32_t)__builtin_popcount(density[w] & ((1u << (i & 31)) - 1u));
  return r;
}

template <int GROUP>
int msm_host_multi(const std::vector<int>& devs, const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars,
                   size_t n_scalars, const uint32_t* density, size_t density_bits, uint64_t* out_xyz) {
  using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
  t_last_err_index = -1;
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  const uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;  // exponents before the first Eof
  // ---- the plan: point ranges x window groups
  uint32_t pg = (uint32_t)devs.size(), wg = 1;
  if (const char* env = std::getenv("MI355ZK_MULTI_PLAN")) {
    unsigned a = 0, b = 0;
    if (std::sscanf(env, "%ux%u", &a, &b) == 2 && a >= 1 && b >= 1 && (size_t)a * b <= devs.size()) { pg = a; wg = b; }
  }
  if (wg > 1) {  // the window count of the range's geometry must divide (choose_geom takes care of that; W == 0: no such layout)
    uint32_t c = 0, W = 0;
    msm_geometry((n + pg - 1) / pg, wg, &c, &W);
    if (W == 0 || W % wg) wg = 1;
  }
  while (pg > 1 && n / pg < 32) --pg;
  std::vector<uint64_t> cut(pg + 1, 0);
  for (uint32_t r = 1; r < pg; ++r) cut[r] = ((n * r / pg) + 31) & ~31ull;
  cut[pg] = n;
  struct Cell {
    int dev = 0;
    uint64_t lo = 0, hi = 0;
    uint32_t wgi = 0;
    int rc = ZK_OK;
    long long err = -1;
    J part;
  };
  std::vector<Cell> cells;
  for (uint32_t r = 0; r < pg; ++r)
    for (uint32_t g = 0; g < wg; ++g) {
      if (cut[r + 1] <= cut[r]) continue;
      Cell c;
      c.dev = devs[cells.size() % devs.size()];
      c.lo = cut[r];
      c.hi = cut[r + 1];
      c.wgi = g;
      c.part = J::zero();
      cells.push_back(c);
    }
  static const bool trace = std::getenv("MI355ZK_TRACE_HOST") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto run_cell = [&](Cell& c) {
    if (hipSetDevice(c.dev) != hipSuccess) { c.rc = ZK_ERR_DEVICE; return; }
    // The cell sees only the SLICE of the base vector its exponents consume -- [boff, boff + used) -- as a vector of its own: that is what
    // its device allocates, uploads and (inside a pinned vector) keeps: 2^26 G1 points on 8 devices are 512 MiB per device, not 4 GiB
    // (SURVEY 8e).  The ranges end before the first exponent without a base (the Eof is planned above for the whole call), so the slice
    // holds every base the cell asks for.
    constexpr size_t bsz = GROUP == 1 ? 64 : 128;
    const uint64_t boff = base_offset + density_rank(P, density, c.lo);
    const uint64_t used = density_rank(P, density, c.hi) - density_rank(P, density, c.lo);
    struct ParentScope {
      explicit ParentScope(const void* p) { t_bases_parent = p; }
      ~ParentScope() { t_bases_parent = nullptr; }
    } parent_scope(bases);
    c.rc = msm_host_run<GROUP>(bases + boff * bsz, used, 0, scalars + c.lo * 4, c.hi - c.lo, density ? density + (c.lo >> 5) : nullptr,
                               density ? c.hi - c.lo : 0, reinterpret_cast<uint64_t*>(&c.part), wg, c.wgi);
    c.err = t_last_err_index;
    if (trace)
      std::fprintf(stderr, "[mi355zk] multi: cell [%llu, %llu) window group %u/%u on device %d: rc %d at %.2f ms\n", (unsigned long long)c.lo,
                   (unsigned long long)c.hi, c.wgi, wg, c.dev, c.rc,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < cells.size(); ++started) th.emplace_back([&, started] { run_guarded(cells[started].rc, [&] { run_cell(cells[started]); }); });
    } catch (const std::exception&) {
      // (no more host threads to be had: the cells that did not get one run here, one after the other -- nothing is thrown across the C ABI)
    }
    if (!cells.empty()) run_guarded(cells[0].rc, [&] { run_cell(cells[0]); });
    for (size_t i = started; i < cells.size(); ++i) run_guarded(cells[i].rc, [&] { run_cell(cells[i]); });
    for (auto& t : th) t.join();
  }
  // ---- the join.  Device failures first, then a non-canonical exponent (bad arguments: the single-device call reports it before
  // anything else too), then the Source errors by global exponent index.
  J total = J::zero();
  long long bad_idx = -1, ident_idx = -1;
  for (Cell& c : cells) {
    if (c.rc < 0) return c.rc;
    if (c.rc == ZK_ERR_BAD_ARGS) {
      const long long g = c.err >= 0 ? c.err + (long long)c.lo : -1;
      if (bad_idx < 0 || (g >= 0 && g < bad_idx)) bad_idx = g >= 0 ? g : bad_idx;
      if (g < 0) { t_last_err_index = -1; return ZK_ERR_BAD_ARGS; }
    } else if (c.rc == ZK_ERR_UNEXPECTED_IDENTITY) {
      const long long g = c.err + (long long)c.lo;
      if (ident_idx < 0 || g < ident_idx) ident_idx = g;
    } else if (c.rc != ZK_OK) {
      return ZK_ERR_DEVICE;  // (a cell never reports Eof: the ranges end before the first exponent without a base)
    }
  }
  if (bad_idx >= 0) { t_last_err_index = bad_idx; return ZK_ERR_BAD_ARGS; }
  if (ident_idx >= 0) { t_last_err_index = ident_idx; return ZK_ERR_UNEXPECTED_IDENTITY; }  // (every exponent before the Eof has a lower index)
  for (Cell& c : cells) jac_add(total, c.part);
  std::memcpy(out_xyz, &total, sizeof total);
  if (P.eof_index >= 0) { t_last_err_index = P.eof_index; return ZK_ERR_UNEXPECTED_EOF; }
  return ZK_OK;
}

// the host-buffer entry points: whole on one device, or cut into cells over the device set
template <int GROUP>
int msm_host_entry(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                   const uint32_t* density, size_t density_bits, uint64_t* out_xyz) {
  const std::vector<int> devs = devset_snapshot();
  if (devs.size() <= 1) return msm_host_run<GROUP>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  if (!out_xyz || (n_scalars && !scalars) || (n_bases && !bases) || n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) {
    t_last_err_index = -1;
    return ZK_ERR_BAD_ARGS;
  }
  const char* env = std::getenv("MI355ZK_MULTI_MIN_LOG");  // (read per call: the tests lower it)
  const int min_log = env ? std::atoi(env) : 20;
  if (n_scalars >= (1ull << (min_log < 0 ? 0 : min_log > 30 ? 30 : min_log)) && n_scalars >= 64)
    return msm_host_multi<GROUP>(devs, bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  // A short call runs whole on ONE device: the one that already holds the (pinned) vector if there is one -- the device copy of a
  // parameter vector is then made once per process, not once per device of the set (8 x 4 GiB for a 2^26-point CRS) -- else the next in
  // turn, so that the prover's concurrent multiexps over its different vectors spread over the node on first touch.
  int pick = -1;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc)
      if (e->host == bases && e->n == n_bases && e->group == GROUP && std::find(devs.begin(), devs.end(), e->dev) != devs.end()) { pick = e->dev; break; }
  }
  if (pick < 0) pick = devs[g_devset_turn.fetch_add(1) % devs.size()];
  DeviceGuard guard;
  ZK_HIP(hipSetDevice(pick));
  return msm_host_run<GROUP>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
}

// ------------------------------------------------------------------------------------------------
// batch_exp on HOST buffers, over the device set: what `MPCParameters::contribute` (phase2/src/parameters.rs:423-470: every point of L
// and H times delta^-1) and powersoftau's `batch_exp` (batched_accumulator.rs:1130-1181) are to a single-process caller.  The points are
// independent, so they shard by CONTIGUOUS POINT RANGE with no exchange at all (shard.batch_exp_sharded is the
// one-process-per-GPU form): device d of mi355zk_init's set takes range d -- upload, the batch_exp kernels, download -- from its own
// host thread; with one device the whole vector is one range.  Ranges are worked off in pieces of 2^18 points (a piece's buffers
// come from the grow-only pool, so a 2^26-point vector does not allocate 10 GiB).  g2_trusted: the promise flag of batch_exp_dev.
template <class F>
int batch_exp_host(uint8_t* out, const uint8_t* bases, const uint64_t* scalars, size_t n, int same_scalar, bool g2_trusted) {
  if ((n && (!out || !bases)) || !scalars) return ZK_ERR_BAD_ARGS;
  if (n == 0) return ZK_OK;
  if (n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  constexpr size_t rec = sizeof(Affine<F>);
  std::vector<int> devs = devset_snapshot();
  if (devs.empty()) {
    int cur = 0;
    ZK_HIP(hipGetDevice(&cur));
    devs.push_back(cur);
  }
  size_t parts = devs.size();
  while (parts > 1 && n / parts < 1024) --parts;
  std::vector<int> rcs(parts, ZK_OK);
  auto run_range = [&](size_t d) {
    const size_t lo = n * d / parts, hi = n * (d + 1) / parts;
    if (hipSetDevice(devs[d]) != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    StageLease stage_lease;
    HostStage* S = host_stage(devs[d], &stage_lease);
    if (S == nullptr) { rcs[d] = ZK_ERR_DEVICE; return; }
    // pieces of 2^18 points, double-buffered
    const size_t piece = (size_t)1 << 18;
    const size_t m_max = hi - lo < piece ? hi - lo : piece;
    const size_t in_bytes = (m_max * rec + 255) & ~(size_t)255;
    const size_t sc_bytes = same_scalar ? 256 : ((m_max * 32 + 255) & ~(size_t)255);
    DensityPool::Lease buf;   // (the grow-only device buffer pool of the host-buffer entry points)
    int rc = buf.acquire(devs[d], 4 * in_bytes + 2 * sc_bytes, S->compute);
    if (rc) { rcs[d] = rc; return; }
    char* base = (char*)buf.b->p;
    char* d_in[2] = {base, base + in_bytes};
    char* d_out[2] = {base + 2 * in_bytes, base + 3 * in_bytes};
    char* d_sc[2] = {base + 4 * in_bytes, base + 4 * in_bytes + sc_bytes};
    hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    auto cleanup = [&] {
      (void)hipStreamSynchronize(S->copy);
      (void)hipStreamSynchronize(S->compute);
      for (int k = 0; k < 2; ++k) { if (up[k]) (void)hipEventDestroy(up[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    };
    auto fail = [&](hipError_t e) {
      std::fprintf(stderr, "[mi355zk] batch_exp (host buffers): HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
      rcs[d] = ZK_ERR_DEVICE;
      cleanup();
    };
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2; ++k)
      if ((e = hipEventCreateWithFlags(&up[k], hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming)) != hipSuccess) return fail(e);
    if (same_scalar && (e = hipMemcpyAsync(d_sc[0], scalars, 32, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return fail(e);
    // ONE host thread keeps the device busy although copies from / to PAGEABLE host memory block it: the kernels of piece i + 1 are
    // always queued before the thread waits for piece i's download, and piece i + 2 is uploaded (into the buffer piece i's kernels
    // have finished with: its download has just returned) while piece i + 1 computes.
    const size_t n_pieces = (hi - lo + piece - 1) / piece;
    auto upload_and_launch = [&](size_t i) -> bool {
      const size_t p0 = lo + i * piece, m = hi - p0 < piece ? hi - p0 : piece;
      const int k = (int)(i & 1);
      if ((e = hipMemcpyAsync(d_in[k], bases + p0 * rec, m * rec, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return false;
      if (!same_scalar && (e = hipMemcpyAsync(d_sc[k], scalars + p0 * 4, m * 32, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return false;
      if ((e = hipEventRecord(up[k], S->copy)) != hipSuccess || (e = hipStreamWaitEvent(S->compute, up[k], 0)) != hipSuccess) return false;
      rc = batch_exp<F>(d_out[k], d_in[k], 0, same_scalar ? d_sc[0] : d_sc[k], same_scalar, m, (void*)S->compute, nullptr, false, g2_trusted);
      if (rc) return false;
      return (e = hipEventRecord(done[k], S->compute)) == hipSuccess;
    };
    auto bail = [&] {
      if (rc) { rcs[d] = rc; cleanup(); }
      else fail(e);
    };
    for (size_t i = 0; i < 2 && i < n_pieces; ++i)
      if (!upload_and_launch(i)) return bail();
    for (size_t i = 0; i < n_pieces; ++i) {
      const size_t p0 = lo + i * piece, m = hi - p0 < piece ? hi - p0 : piece;
      const int k = (int)(i & 1);
      if ((e = hipStreamWaitEvent(S->copy, done[k], 0)) != hipSuccess) return fail(e);
      if ((e = hipMemcpyAsync(out + p0 * rec, d_out[k], m * rec, hipMemcpyDeviceToHost, S->copy)) != hipSuccess) return fail(e);
      if ((e = hipStreamSynchronize(S->copy)) != hipSuccess) return fail(e);   // piece i is on the host; its buffers are free
      if (i + 2 < n_pieces && !upload_and_launch(i + 2)) return bail();
    }
    if ((e = hipStreamSynchronize(S->compute)) != hipSuccess) return fail(e);
    cleanup();
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < parts; ++started) th.emplace_back([&, started] { run_guarded(rcs[started], [&] { run_range(started); }); });
    } catch (const std::exception&) {
    }
    run_guarded(rcs[0], [&] { run_range(0); });
    for (size_t d = started; d < parts; ++d) run_guarded(rcs[d], [&] { run_range(d); });
    for (auto& t : th) t.join();
  }
  for (int rc : rcs)
    if (rc != ZK_OK) return rc;
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// dense_multiexp / merge_pairs on HOST buffers, over the device set: the verification multiexps of the ceremony code (SURVEY 8f row 2:
// powersoftau/src/utils.rs:112-135, 189-292; phase2/src/utils.rs:59-105) for a single-process caller.  sum_i rho_i * v_i is linear in the
// points, so the vectors are cut into pieces of 2^22 points, every piece is one device call (msm_g*_dense_device: digits and partition
// shared by the two sums of merge_pairs) and the Jacobian partials are added on the host.  The pieces are dealt to TWO host threads per
// device of mi355zk_init's set (one piece uploads -- pageable copies block their thread -- while the other computes); v2 == nullptr:
// dense_multiexp.  No Source errors (infinity bases add nothing: the reference's dense contract).
template <int GROUP>
int dense_host(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t* out_s, uint64_t* out_sx) {
  using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
  constexpr size_t rec = GROUP == 1 ? 64 : 128;
  if (!out_s || (v2 && !out_sx) || (n && (!v1 || !rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  J total = J::zero(), total2 = J::zero();
  if (n > 0) {
    std::vector<int> devs = devset_snapshot();
    if (devs.empty()) {
      int cur = 0;
      ZK_HIP(hipGetDevice(&cur));
      devs.push_back(cur);
    }
    size_t piece = (size_t)1 << 22;
    if (const char* env = std::getenv("MI355ZK_DENSE_PIECE_TEST")) {   // (test hook, read per call: points per piece, so that the cut can be held against the oracle)
      const size_t v = (size_t)std::strtoull(env, nullptr, 10);
      if (v >= 16) piece = v;
    }
    const size_t n_pieces = (n + piece - 1) / piece;
    size_t workers = 2 * devs.size();
    if (workers > n_pieces) workers = n_pieces;
    struct Part { int rc = ZK_OK; J s, sx; };
    std::vector<Part> parts(workers);
    for (auto& pt : parts) { pt.s = J::zero(); pt.sx = J::zero(); }
    std::atomic<size_t> next{0};
    auto work = [&](size_t wk) {
      Part& P = parts[wk];
      const int dev = devs[wk % devs.size()];
      if (hipSetDevice(dev) != hipSuccess) { P.rc = ZK_ERR_DEVICE; return; }
      StageLease stage_lease;
      HostStage* S = host_stage(dev, &stage_lease);
      if (S == nullptr) { P.rc = ZK_ERR_DEVICE; return; }
      const size_t m_max = n < piece ? n : piece;
      const size_t vb = ((m_max + 16) * rec + 255) & ~(size_t)255;
      DensityPool::Lease buf;
      if (int rc = buf.acquire(dev, (v2 ? 2 : 1) * vb + m_max * 32, S->compute)) { P.rc = rc; return; }
      char* d_v1 = (char*)buf.b->p;
      // power_pairs (utils.rs:133-135) is merge_pairs(v[0 .. n-1], v[1 .. n]): the two vectors are ONE array seen at two offsets, and
      // uploading it twice would double the PCIe traffic of a call the link already bounds -- a v2 that starts `shift` (<= 16)
      // records into v1 shares v1's upload
      // (the addresses are compared as integers -- the two pointers need not belong to one array -- and the vectors must really overlap:
      // shift <= n; two separate short arrays that happen to sit within 16 records of each other are uploaded separately: ADVICE r4)
      const uintptr_t a1 = (uintptr_t)v1, a2 = (uintptr_t)v2;
      const size_t shift = (v2 && a2 >= a1 && (a2 - a1) % rec == 0 && (a2 - a1) / rec <= 16 && (a2 - a1) / rec <= n) ? (size_t)((a2 - a1) / rec) : (size_t)-1;
      const bool shared = shift != (size_t)-1;
      char* d_v2 = v2 ? (shared ? d_v1 + shift * rec : d_v1 + vb) : nullptr;
      char* d_rho = d_v1 + (v2 ? 2 : 1) * vb;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n_pieces) break;
        const size_t p0 = i * piece, m = n - p0 < piece ? n - p0 : piece;
        hipError_t e = hipMemcpyAsync(d_v1, v1 + p0 * rec, (m + (shared ? shift : 0)) * rec, hipMemcpyHostToDevice, S->compute);
        if (e == hipSuccess && v2 && !shared) e = hipMemcpyAsync(d_v2, v2 + p0 * rec, m * rec, hipMemcpyHostToDevice, S->compute);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rho, rho + p0 * 4, m * 32, hipMemcpyHostToDevice, S->compute);
        if (e != hipSuccess) {
          std::fprintf(stderr, "[mi355zk] dense multiexp (host buffers): HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
          P.rc = ZK_ERR_DEVICE;
          return;
        }
        J a = J::zero(), b = J::zero();
        const int rc = GROUP == 1 ? msm_g1_dense_device(d_v1, d_v2, d_rho, m, S->compute, reinterpret_cast<uint64_t*>(&a), v2 ? reinterpret_cast<uint64_t*>(&b) : nullptr)
                                  : msm_g2_dense_device(d_v1, d_v2, d_rho, m, S->compute, reinterpret_cast<uint64_t*>(&a), v2 ? reinterpret_cast<uint64_t*>(&b) : nullptr);
        if (rc != ZK_OK) { P.rc = rc; return; }
        jac_add(P.s, a);
        if (v2) jac_add(P.sx, b);
      }
    };
    {
      DeviceGuard guard;
      std::vector<std::thread> th;
      size_t started = 1;
      try {
        for (; started < workers; ++started) th.emplace_back([&, started] { run_guarded(parts[started].rc, [&] { work(started); }); });
      } catch (const std::exception&) {
      }
      run_guarded(parts[0].rc, [&] { work(0); });   // (a worker takes pieces until none is left: the ones that got no thread are covered by the others)
      for (auto& t : th) t.join();
    }
    for (auto& pt : parts) {
      if (pt.rc != ZK_OK) return pt.rc;
      jac_add(total, pt.s);
      if (v2) jac_add(total2, pt.sx);
    }
  }
  std::memcpy(out_s, &total, sizeof total);
  if (v2) std::memcpy(out_sx, &total2, sizeof total2);
  return ZK_OK;
}

// best_fft / the domain operations on a HOST array (what a bellman shim calls with `&mut [Scalar<E>]`): upload, transform in place
// on the device, copy back.  Device buffer and stream are leased from the pools of the host-buffer entry points -- round 2
// hipMalloc'ed and hipFree'd per call (both synchronise the whole device, i.e. every other thread's multiexp) and ran on the null
// stream.  `a` is written by the final copy only: on a device failure (rc < 0) the caller's array is untouched and it can fall
// back to its own serial_fft.
int ntt_host(uint64_t* a, uint32_t log_n, int op, const uint64_t* omega) {
  if (!a) return ZK_ERR_BAD_ARGS;
  if (log_n > 28) return ZK_ERR_BAD_ARGS;
  const size_t bytes = (size_t)32 << log_n;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  StageLease stage_lease;
  HostStage* S = host_stage(dev, &stage_lease);
  if (S == nullptr) return ZK_ERR_DEVICE;
  DensityPool::Lease buf;   // (a grow-only device buffer pool; the lease synchronises the stream before the buffer is handed on)
  int rc = buf.acquire(dev, bytes, S->compute);
  if (rc) return rc;
  void* d = buf.b->p;
  ZK_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, S->compute));
  if (omega) {
    Fr w;
    std::memcpy(&w, omega, 32);
    rc = ntt_run((Fr*)d, log_n, w, S->compute);
  } else {
    rc = domain_op_dev((Fr*)d, log_n, op, S->compute);
  }
  if (rc != ZK_OK) return rc;
  ZK_HIP(hipStreamSynchronize(S->compute));  // a failed kernel surfaces here, before the caller's array is touched
  ZK_HIP(hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, S->compute));
  ZK_HIP(hipStreamSynchronize(S->compute));
  return ZK_OK;
}

}  // namespace
}  // namespace zk

using namespace zk;

// out[r] = sum_{t in [row_ptr[r], row_ptr[r+1])} coeff[t] * bases[col[t]], affine (QAP evaluation, parameters.rs:225-294)
// CSR sanity on the device: flag |= 1 if some col[t] >= n_bases, |= 2 if row_ptr is not 0 = row_ptr[0] <= ... <= row_ptr[n_rows] = nnz
__global__ void __launch_bounds__(256) csr_check_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, uint64_t n_rows,
                                                       uint64_t nnz, uint64_t n_bases, uint32_t* __restrict__ flag) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += stride)
    if (col[t] >= n_bases) bad |= 1u;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
    const uint32_t v = row_ptr[r];
    if ((r == 0 && v != 0) || (r == n_rows && v != nnz) || (r < n_rows && v > row_ptr[r + 1])) bad |= 2u;
  }
  if (bad) atomicOr(flag, bad);
}

template <class F>
static int sparse_matvec(void* d_out, const void* d_bases, size_t n_bases, const uint32_t* d_row_ptr, const uint32_t* d_col, const void* d_coeffs,
                         size_t n_rows, size_t nnz, void* stream, int group, bool g2_trusted, void* d_scratch = nullptr, size_t scratch_bytes = 0) {
  // d_scratch: the caller's buffer for the term products (the host-buffer form leases it with its other buffers: hipMalloc / hipFree per
  // call synchronise the whole device, i.e. every other thread's multiexp): room for the terms, 256 B of flags and one byte per base
  if (!d_out || !d_row_ptr || (nnz && (!d_bases || !d_col || !d_coeffs)) || n_rows >= (1ull << 31) || nnz >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  if (n_rows == 0) return ZK_OK;
  // G2 without the caller's promise: when the bases are reused (nnz >= 2 n_bases: a circuit has ~3 terms per Lagrange coefficient), ONE
  // membership test per base (psi(P) == mu P: 127 doublings + 68 additions) buys the psi split (a third fewer operations) and the
  // r - 1 shortcut for every term of a member, and only the terms of the other bases take the plain windows -- the result is the
  // reference's either way.  (2^20 bases, 2.9 M terms: 156 -> ~155 ms general coefficients, 90 -> ~61 ms with 90 % unit coefficients.)
  const bool by_member = std::is_same<F, Fq2>::value && !g2_trusted && nnz >= 2 * n_bases && n_bases > 0;
  const size_t terms_bytes = ((nnz ? nnz : 1) * sizeof(Affine<F>) + 255) & ~(size_t)255;
  Affine<F>* d_terms = nullptr;
  const bool own = d_scratch == nullptr || scratch_bytes < terms_bytes + 256 + (by_member ? n_bases : 0);
  if (own) ZK_HIP(hipMalloc(&d_terms, terms_bytes + 256 + (by_member ? n_bases : 0)));
  else d_terms = (Affine<F>*)d_scratch;
  uint8_t* d_member = by_member ? reinterpret_cast<uint8_t*>(d_terms) + terms_bytes + 256 : nullptr;
  {
    // the ABI cannot trust the index arrays: an out-of-range column would be an out-of-bounds gather in batch_exp
    uint32_t* d_flag = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_terms) + terms_bytes);
    uint32_t h_flag = 0;
    hipError_t e = hipMemsetAsync(d_flag, 0, 4, (hipStream_t)stream);
    if (e == hipSuccess) {
      const uint64_t work = nnz > n_rows + 1 ? nnz : n_rows + 1;
      hipLaunchKernelGGL(csr_check_kernel, dim3((unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                         d_row_ptr, d_col, (uint64_t)n_rows, (uint64_t)nnz, (uint64_t)n_bases, d_flag);
      e = hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess || h_flag) {
      if (own) (void)hipFree(d_terms);
      if (e != hipSuccess) ZK_HIP(e);
      return ZK_ERR_BAD_ARGS;
    }
  }
  int rc = ZK_OK;
  if constexpr (std::is_same<F, Fq2>::value)
    if (by_member) rc = g2_subgroup_flags(d_bases, n_bases, stream, d_member);
  if (rc == ZK_OK) rc = batch_exp<F>(d_terms, d_bases, 0, d_coeffs, 0, nnz, stream, d_col, /*shortcut_unit_scalars=*/true, g2_trusted, d_member);
  if (rc == ZK_OK)
    rc = group == 1 ? segsum_g1_device(d_terms, nnz, d_row_ptr, (uint32_t)n_rows, (hipStream_t)stream, d_out)
                    : segsum_g2_device(d_terms, nnz, d_row_ptr, (uint32_t)n_rows, (hipStream_t)stream, d_out);
  if (own) (void)hipFree(d_terms);
  return rc;
}

// The QAP evaluation on HOST buffers, over the device set (SURVEY 8f row 3 for a single-process caller: MPCParameters::new over a
// 2^20+-constraint circuit): the rows of the CSR matrix are independent, so device d takes the d-th contiguous ROW range -- its slice of
// (col, coeff), a row_ptr rebased to zero, and the WHOLE base vector (any row may name any Lagrange coefficient) -- and writes its rows
// of the output; no exchange.  The index arrays are validated on the device as in the _dev form; row_ptr[0] == 0, row_ptr[n_rows] == nnz
// and monotonicity across the cuts are checked here.
template <class F>
static int sparse_matvec_host(uint8_t* out, const uint8_t* bases, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col, const uint64_t* coeffs,
                              size_t n_rows, size_t nnz, int group, bool g2_trusted) {
  constexpr size_t rec = sizeof(Affine<F>);
  if (!row_ptr || (n_rows && !out) || (nnz && (!bases || !col || !coeffs)) || n_rows >= (1ull << 31) || nnz >= (1ull << 31) || n_bases >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  if (n_rows == 0) return ZK_OK;
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return ZK_ERR_BAD_ARGS;
  std::vector<int> devs = devset_snapshot();
  if (devs.empty()) {
    int cur = 0;
    ZK_HIP(hipGetDevice(&cur));
    devs.push_back(cur);
  }
  size_t parts = devs.size();
  while (parts > 1 && n_rows / parts < 128) --parts;
  // cuts of equal WEIGHT (rows + terms: a row costs a normalisation, a term an addition chain), found by one walk over row_ptr -- the
  // variables of a circuit are far from equally used (the constant ONE sits in most constraints)
  std::vector<size_t> cut(parts + 1, n_rows);
  cut[0] = 0;
  {
    const size_t weight = n_rows + nnz;
    size_t d = 1;
    for (size_t r = 0; r < n_rows && d < parts; ++r)
      while (d < parts && r + (size_t)row_ptr[r] >= weight * d / parts) cut[d++] = r;
  }
  std::vector<int> rcs(parts, ZK_OK);
  auto run_range = [&](size_t d) {
    const size_t r0 = cut[d], r1 = cut[d + 1];
    if (r1 == r0) return;
    const uint32_t t0 = row_ptr[r0], t1 = row_ptr[r1];
    if (t1 < t0) { rcs[d] = ZK_ERR_BAD_ARGS; return; }
    const size_t rows = r1 - r0, terms = t1 - t0;
    if (hipSetDevice(devs[d]) != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    StageLease stage_lease;
    HostStage* S = host_stage(devs[d], &stage_lease);
    if (S == nullptr) { rcs[d] = ZK_ERR_DEVICE; return; }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_bases = 0, o_out = o_bases + al((n_bases ? n_bases : 1) * rec), o_rp = o_out + al(rows * rec), o_col = o_rp + al((rows + 1) * 4),
                 o_cf = o_col + al((terms ? terms : 1) * 4), o_scr = o_cf + al((terms ? terms : 1) * 32),
                 scr_bytes = al((terms ? terms : 1) * rec) + 256 + al(n_bases), total = o_scr + scr_bytes;
    DensityPool::Lease buf;
    if (int rc = buf.acquire(devs[d], total, S->compute)) { rcs[d] = rc; return; }
    char* base = (char*)buf.b->p;
    std::vector<uint32_t> rp(rows + 1);
    for (size_t r = 0; r <= rows; ++r) {
      const uint32_t v = row_ptr[r0 + r];
      if (v < t0 || v > t1) { rcs[d] = ZK_ERR_BAD_ARGS; return; }   // (monotone inside the range is checked on the device)
      rp[r] = v - t0;
    }
    hipError_t e = hipSuccess;
    if (n_bases) e = hipMemcpyAsync(base + o_bases, bases, n_bases * rec, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess) e = hipMemcpyAsync(base + o_rp, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess && terms) e = hipMemcpyAsync(base + o_col, col + t0, terms * 4, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess && terms) e = hipMemcpyAsync(base + o_cf, coeffs + (size_t)t0 * 4, terms * 32, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(S->compute);   // (rp is a local vector)
    if (e != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    int rc = sparse_matvec<F>(base + o_out, base + o_bases, n_bases, (const uint32_t*)(base + o_rp), (const uint32_t*)(base + o_col), base + o_cf, rows, terms,
                              (void*)S->compute, group, g2_trusted, base + o_scr, scr_bytes);
    if (rc != ZK_OK) { rcs[d] = rc; return; }
    e = hipMemcpyAsync(out + r0 * rec, base + o_out, rows * rec, hipMemcpyDeviceToHost, S->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(S->compute);
    if (e != hipSuccess) rcs[d] = ZK_ERR_DEVICE;
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < parts; ++started) th.emplace_back([&, started] { run_guarded(rcs[started], [&] { run_range(started); }); });
    } catch (const std::exception&) {
    }
    run_guarded(rcs[0], [&] { run_range(0); });
    for (size_t d = started; d < parts; ++d) run_guarded(rcs[d], [&] { run_range(d); });
    for (auto& t : th) t.join();
  }
  for (int rc : rcs)
    if (rc != ZK_OK) return rc;
  return ZK_OK;
}

extern "C" {

int mi355zk_init(const int* device_ids, int n_devices) {
  return abi_guard([&]() -> int {
    if (n_devices < 0 || (n_devices > 0 && device_ids == nullptr)) return ZK_ERR_BAD_ARGS;
    int count = 0;
    ZK_HIP(hipGetDeviceCount(&count));
    auto check = [](int dev) -> int {
      hipDeviceProp_t prop;
      ZK_HIP(hipGetDeviceProperties(&prop, dev));
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "[mi355zk] device %d is %s; this library contains gfx950 code only\n", dev, prop.gcnArchName);
        return ZK_ERR_DEVICE;
      }
      return ZK_OK;
    };
    for (int i = 0; i < n_devices; ++i)
      if (device_ids[i] < 0 || device_ids[i] >= count) return ZK_ERR_BAD_ARGS;
    if (n_devices > 0) ZK_HIP(hipSetDevice(device_ids[0]));
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    // the LAST call defines the device set: more than one id = the single-process multi-GPU mode of the host-buffer multiexps
    // (msm_host_multi), one id or none = one device, as before
    std::vector<int> set;
    if (n_devices > 1) set.assign(device_ids, device_ids + n_devices);
    int rc = ZK_OK;
    if (set.empty()) {
      rc = check(dev);
      if (rc == ZK_OK) rc = ntt_configure();
    } else {
      for (size_t i = 0; i < set.size() && rc == ZK_OK; ++i) {
        bool seen = false;
        for (size_t k = 0; k < i; ++k) seen = seen || set[k] == set[i];
        if (seen) continue;
        rc = check(set[i]);
        if (rc == ZK_OK && hipSetDevice(set[i]) != hipSuccess) rc = ZK_ERR_DEVICE;
        if (rc == ZK_OK) rc = ntt_configure();
      }
      (void)hipSetDevice(dev);
    }
    if (rc != ZK_OK) return rc;
    std::lock_guard<std::mutex> lk(g_devset_mu);
    g_devset = set;
    return ZK_OK;
  });
}
int mi355zk_visible_devices(void) {
  return abi_guard([&]() -> int {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
  });
}
int mi355zk_device_count(void) {
  return abi_guard([&]() -> int {
    std::lock_guard<std::mutex> lk(g_devset_mu);
    return g_devset.empty() ? 1 : (int)g_devset.size();
  });
}

void mi355zk_shutdown(void) {
  abi_guard_void([&] {
    DeviceGuard guard;
    prof_release_all();
    ntt_release_all();
    exp_scratch_release_all();
    mul_slots_release_all();
    host_entry_release_all();
    msm_release_g1();
    msm_release_g2();
  });
}

const char* mi355zk_version(void) { return "mi355zk 0.3 (gfx950)"; }
int mi355zk_abi_version(void) { return MI355ZK_ABI_VERSION; }

int mi355zk_bases_cache_pin(const void* host_bases, size_t n_bases, int group) { return abi_guard([&]() -> int { return bases_cache_pin(host_bases, n_bases, group); }); }
int mi355zk_bases_cache_pin_tables(const void* host_bases, size_t n_bases, int group) { return abi_guard([&]() -> int { return bases_cache_pin(host_bases, n_bases, group, true); }); }
int mi355zk_bases_cache_info(const void* host_bases, size_t* device_bytes, size_t* table_bytes) {
  return abi_guard([&]() -> int {
    size_t d = 0, t = 0;
    int found = 0;
    {
      std::lock_guard<std::mutex> lk(g_bc_mu);
      for (auto& e : g_bc)
        if ((e->host == host_bases || e->owner == host_bases) && e->ready) { d += e->bytes; t += e->table_bytes; found = 1; }
    }
    if (device_bytes) *device_bytes = d;
    if (table_bytes) *table_bytes = t;
    return found;
  });
}
void mi355zk_bases_cache_invalidate(const void* host_bases) {
  abi_guard_void([&] {
    int dev = 0;
    const bool have = hipGetDevice(&dev) == hipSuccess;
    bases_cache_invalidate(host_bases);
    if (have) (void)hipSetDevice(dev);
  });
}

int mi355zk_bn254_g1_msm(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                         const uint32_t* density, size_t density_bits, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return msm_host_entry<1>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  });
}
int mi355zk_bn254_g2_msm(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                         const uint32_t* density, size_t density_bits, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return msm_host_entry<2>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  });
}
int mi355zk_bn254_g1_msm_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                             const uint32_t* density, size_t density_bits, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz);
  });
}
int mi355zk_bn254_g2_msm_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                             const uint32_t* density, size_t density_bits, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz);
  });
}
// one window group of a multiexp (multi-GPU sharding by windows; shard.py)
int mi355zk_bn254_g1_msm_part_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                  const uint32_t* density, size_t density_bits, uint32_t window_groups, uint32_t window_group, void* stream,
                                  uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group);
  });
}
int mi355zk_bn254_g2_msm_part_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                  const uint32_t* density, size_t density_bits, uint32_t window_groups, uint32_t window_group, void* stream,
                                  uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group);
  });
}
// flags (MI355ZK_MSM_*) + window groups in one entry point
int mi355zk_bn254_g1_msm_ex_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                const uint32_t* density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups || (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY)) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group, flags);
  });
}
int mi355zk_bn254_g2_msm_ex_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                const uint32_t* density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups || (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY)) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group, flags);
  });
}
int mi355zk_msm_table_geometry(size_t n_bases, int group, uint32_t* window_bits, uint32_t* n_windows) {
  return abi_guard([&]() -> int {
    if ((group != 1 && group != 2) || n_bases >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    uint32_t c = 0, W = 0;
    msm_table_geometry(n_bases ? n_bases : 1, group, &c, &W, nullptr);
    if (window_bits) *window_bits = c;
    if (n_windows) *n_windows = W;
    return ZK_OK;
  });
}
int mi355zk_bn254_g1_msm_table_build_dev(const void* d_bases, size_t n_bases, void* d_table, size_t table_bytes, void* stream) {
  return abi_guard([&]() -> int {
    return msm_table_build<1>(d_bases, n_bases, d_table, table_bytes, stream);
  });
}
int mi355zk_bn254_g2_msm_table_build_dev(const void* d_bases, size_t n_bases, void* d_table, size_t table_bytes, void* stream) {
  return abi_guard([&]() -> int {
    return msm_table_build<2>(d_bases, n_bases, d_table, table_bytes, stream);
  });
}
int mi355zk_bn254_g1_msm_table_dev(const void* d_table, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                   const uint32_t* density, size_t density_bits, uint32_t flags, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_table, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, 1, 0, flags, nullptr, true);
  });
}
int mi355zk_bn254_g2_msm_table_dev(const void* d_table, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                   const uint32_t* density, size_t density_bits, uint32_t flags, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_table, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, 1, 0, flags, nullptr, true);
  });
}
int mi355zk_bn254_g1_dense_multiexp(const uint8_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return dense_host<1>(bases, nullptr, scalars, n, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g2_dense_multiexp(const uint8_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return dense_host<2>(bases, nullptr, scalars, n, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g1_merge_pairs(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t out_s[12], uint64_t out_sx[12]) {
  return abi_guard([&]() -> int {
    if (!v2) return ZK_ERR_BAD_ARGS;
    return dense_host<1>(v1, v2, rho, n, out_s, out_sx);
  });
}
int mi355zk_bn254_g2_merge_pairs(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t out_s[24], uint64_t out_sx[24]) {
  return abi_guard([&]() -> int {
    if (!v2) return ZK_ERR_BAD_ARGS;
    return dense_host<2>(v1, v2, rho, n, out_s, out_sx);
  });
}
int mi355zk_bn254_g1_dense_multiexp_dev(const void* d_bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (!out_xyz || (n && (!d_bases || !d_scalars)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g1_dense_device(d_bases, nullptr, d_scalars, n, (hipStream_t)stream, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g2_dense_multiexp_dev(const void* d_bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (!out_xyz || (n && (!d_bases || !d_scalars)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g2_dense_device(d_bases, nullptr, d_scalars, n, (hipStream_t)stream, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g1_merge_pairs_dev(const void* d_v1, const void* d_v2, const void* d_rho, size_t n, void* stream, uint64_t out_s[12],
                                     uint64_t out_sx[12]) {
  return abi_guard([&]() -> int {
    if (!out_s || !out_sx || (n && (!d_v1 || !d_v2 || !d_rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g1_dense_device(d_v1, d_v2, d_rho, n, (hipStream_t)stream, out_s, out_sx);
  });
}
int mi355zk_bn254_g2_merge_pairs_dev(const void* d_v1, const void* d_v2, const void* d_rho, size_t n, void* stream, uint64_t out_s[24],
                                     uint64_t out_sx[24]) {
  return abi_guard([&]() -> int {
    if (!out_s || !out_sx || (n && (!d_v1 || !d_v2 || !d_rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g2_dense_device(d_v1, d_v2, d_rho, n, (hipStream_t)stream, out_s, out_sx);
  });
}
long long mi355zk_last_error_index(void) { return t_last_err_index; }
int mi355zk_msm_window_bits(size_t n_scalars, int* n_windows) {
  return abi_guard([&]() -> int {
    uint32_t c = 0, W = 0;
    msm_geometry(n_scalars, 1, &c, &W);
    if (n_windows) *n_windows = (int)W;
    return (int)c;
  });
}
int mi355zk_msm_window_bits_groups(size_t n_scalars, uint32_t window_groups, int* n_windows) {
  return abi_guard([&]() -> int {
    uint32_t c = 0, W = 0;
    msm_geometry(n_scalars, window_groups, &c, &W);
    if (n_windows) *n_windows = (int)W;
    return (int)c;
  });
}
// (test hook) digit extraction of one scalar on the host: see msm_selftest_digits in msm_g1.hip
int mi355zk_selftest_msm_digits(size_t n_scalars, uint32_t window_groups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct,
                                int32_t* digits, uint32_t* geom) {
  return abi_guard([&]() -> int {
    return msm_selftest_digits(n_scalars, window_groups, scalar, w_start, w_stop, direct, digits, geom);
  });
}

int mi355zk_bn254_fr_ntt(uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
  return abi_guard([&]() -> int {
    if (!omega) return ZK_ERR_BAD_ARGS;
    return ntt_host(a, log_n, -1, omega);
  });
}
int mi355zk_bn254_fr_domain_op(uint64_t* a, uint32_t log_n, int op) {
  return abi_guard([&]() -> int {
    if (op < 0 || op > 3) return ZK_ERR_BAD_ARGS;
    return ntt_host(a, log_n, op, nullptr);
  });
}
int mi355zk_bn254_fr_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_FFT, nullptr); }); }
int mi355zk_bn254_fr_ifft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_IFFT, nullptr); }); }
int mi355zk_bn254_fr_coset_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_COSET_FFT, nullptr); }); }
int mi355zk_bn254_fr_icoset_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_ICOSET_FFT, nullptr); }); }
int mi355zk_bn254_fr_ntt_dev(void* d_a, uint32_t log_n, const uint64_t omega[4], void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || !omega || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr w;
    std::memcpy(&w, omega, 32);
    return ntt_run((Fr*)d_a, log_n, w, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_ntt_scaled_dev(void* d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t pre_g[4], const uint64_t post_c[4],
                                    const uint64_t post_g[4], void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || !omega || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr w, g, c, h;
    std::memcpy(&w, omega, 32);
    if (pre_g) std::memcpy(&g, pre_g, 32);
    if (post_c) std::memcpy(&c, post_c, 32);
    if (post_g) std::memcpy(&h, post_g, 32);
    return ntt_run_scaled((Fr*)d_a, log_n, w, pre_g ? &g : nullptr, post_c ? &c : nullptr, post_g ? &h : nullptr, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_op_dev(void* d_a, uint32_t log_n, int op, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || log_n > 28) return ZK_ERR_BAD_ARGS;
    return domain_op_dev((Fr*)d_a, log_n, op, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_op_batch_dev(void* const* d_arrays, uint32_t batch, uint32_t log_n, int op, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_arrays || batch == 0 || batch > 64 || log_n > 28) return ZK_ERR_BAD_ARGS;
    for (uint32_t t = 0; t < batch; ++t) {
      if (!d_arrays[t]) return ZK_ERR_BAD_ARGS;
      for (uint32_t u = 0; u < t; ++u)
        if (d_arrays[u] == d_arrays[t]) return ZK_ERR_BAD_ARGS;   // (the transforms work in place: one array twice would be transformed twice at once)
    }
    return domain_op_batch_dev(reinterpret_cast<Fr* const*>(d_arrays), batch, log_n, op, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_constants(uint32_t log_n, uint64_t omega[4], uint64_t omegainv[4], uint64_t geninv[4], uint64_t minv[4]) {
  return abi_guard([&]() -> int {
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    if (omega) std::memcpy(omega, &D.omega, 32);
    if (omegainv) std::memcpy(omegainv, &D.omegainv, 32);
    if (geninv) std::memcpy(geninv, &D.geninv, 32);
    if (minv) std::memcpy(minv, &D.minv, 32);
    return ZK_OK;
  });
}

// EvaluationDomain::z (domain.rs:207-212) and divide_by_z_on_coset (domain.rs:217-234)
int mi355zk_bn254_fr_domain_z(uint32_t log_n, const uint64_t tau[4], uint64_t out[4]) {
  return abi_guard([&]() -> int {
    if (!tau || !out || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr t;
    std::memcpy(&t, tau, 32);
    for (uint32_t i = 0; i < log_n; ++i) t = sqr(t);   // tau.pow(&[m]) with m = 2^log_n
    t = sub(t, Fr::one());
    std::memcpy(out, &t, 32);
    return ZK_OK;
  });
}
int mi355zk_bn254_fr_divide_by_z_on_coset_dev(void* d_a, uint32_t log_n, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr z = fr_from_u64(7);                             // E::Fr::multiplicative_generator() (fr.rs:5)
    for (uint32_t i = 0; i < log_n; ++i) z = sqr(z);
    z = sub(z, Fr::one());
    return ntt_scale((Fr*)d_a, log_n, inv(z), nullptr, (hipStream_t)stream);
  });
}

// EvaluationDomain<Point<G1>>::{fft, ifft} on affine records (group.rs:22-51, domain.rs:154-173; prepare_phase2.rs:68-131)
int mi355zk_bn254_g1_point_fft_dev(void* d_points_affine, uint32_t log_n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_points_affine || log_n > 28 || (mode & ~(MI355ZK_FFT_INVERSE | MI355ZK_G2_TRUSTED_SUBGROUP))) return ZK_ERR_BAD_ARGS;
    const int inverse = mode & MI355ZK_FFT_INVERSE;
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    return point_fft_g1(d_points_affine, log_n, inverse ? D.omegainv : D.omega, inverse != 0, to_canonical(D.minv), (hipStream_t)stream);
  });
}

// point codecs (ec.rs:763-946, 1136-1344): wire encodings <-> raw affine records
int mi355zk_bn254_g1_decode_dev(void* d_out_affine, const void* d_in_bytes, size_t n, int compressed, int checked, void* stream, long long* err_index) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_affine || !d_in_bytes)) || ((uintptr_t)d_in_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_decode(1, d_out_affine, d_in_bytes, n, compressed, checked, (hipStream_t)stream, err_index);
  });
}
int mi355zk_bn254_g2_decode_dev(void* d_out_affine, const void* d_in_bytes, size_t n, int compressed, int checked, void* stream, long long* err_index) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_affine || !d_in_bytes)) || ((uintptr_t)d_in_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_decode(2, d_out_affine, d_in_bytes, n, compressed, checked, (hipStream_t)stream, err_index);
  });
}
int mi355zk_bn254_g1_encode_dev(void* d_out_bytes, const void* d_in_affine, size_t n, int compressed, void* stream) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_bytes || !d_in_affine)) || ((uintptr_t)d_out_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_encode(1, d_out_bytes, d_in_affine, n, compressed, (hipStream_t)stream);
  });
}
int mi355zk_bn254_g2_encode_dev(void* d_out_bytes, const void* d_in_affine, size_t n, int compressed, void* stream) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_bytes || !d_in_affine)) || ((uintptr_t)d_out_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_encode(2, d_out_bytes, d_in_affine, n, compressed, (hipStream_t)stream);
  });
}

int mi355zk_bn254_g2_point_fft_dev(void* d_points_affine, uint32_t log_n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_points_affine || log_n > 28 || (mode & ~(MI355ZK_FFT_INVERSE | MI355ZK_G2_TRUSTED_SUBGROUP))) return ZK_ERR_BAD_ARGS;
    const int inverse = mode & MI355ZK_FFT_INVERSE;
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    // Without the caller's promise: a transform of points that ALL lie in the order-r subgroup stays in it, so one membership test per
    // input (127 doublings + 68 additions each, against log_n / 2 multiplications per point) earns the psi split for every stage; one
    // record outside and the whole transform runs the plain windows.  Either way the reference's result.
    bool split = (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0;
    if (!split && log_n >= 5) {
      long long bad = -1;
      rc = g2_subgroup_check(d_points_affine, (size_t)1 << log_n, stream, &bad);
      if (rc) return rc;
      split = bad < 0;
    }
    return point_fft_g2(d_points_affine, log_n, inverse ? D.omegainv : D.omega, inverse != 0, to_canonical(D.minv), (hipStream_t)stream, split);
  });
}

int mi355zk_bn254_g2_subgroup_check_dev(const void* d_points_affine, size_t n, void* stream, long long* bad_index) {
  return abi_guard([&]() -> int {
    return g2_subgroup_check(d_points_affine, n, stream, bad_index);
  });
}
int mi355zk_selftest_g2_in_subgroup(const uint64_t affine_pt[16]) {  // the same test on the HOST: 1 in the subgroup, 0 not, < 0 bad arguments
  if (!affine_pt) return -1;
  G2Affine p;
  std::memcpy(&p, affine_pt, sizeof p);
  return g2_in_subgroup(p) ? 1 : 0;
}
int mi355zk_bn254_g1_batch_mul_dev(void* d_out_affine, const uint64_t base_affine[8], const void* d_scalars, size_t n, void* stream) {
  return abi_guard([&]() -> int {
    return batch_mul<Fq>(d_out_affine, base_affine, d_scalars, n, stream);
  });
}
int mi355zk_bn254_g2_batch_mul_dev(void* d_out_affine, const uint64_t base_affine[16], const void* d_scalars, size_t n, void* stream) {
  return abi_guard([&]() -> int {
    return batch_mul<Fq2>(d_out_affine, base_affine, d_scalars, n, stream);
  });
}
int mi355zk_bn254_g1_sparse_matvec(uint8_t* out_affine, const uint8_t* bases_affine, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col,
                                   const uint64_t* coeffs, size_t n_rows, size_t nnz, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec_host<Fq>(out_affine, bases_affine, n_bases, row_ptr, col, coeffs, n_rows, nnz, 1, false);
  });
}
int mi355zk_bn254_g2_sparse_matvec(uint8_t* out_affine, const uint8_t* bases_affine, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col,
                                   const uint64_t* coeffs, size_t n_rows, size_t nnz, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec_host<Fq2>(out_affine, bases_affine, n_bases, row_ptr, col, coeffs, n_rows, nnz, 2, (flags & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}
int mi355zk_bn254_g1_sparse_matvec_dev(void* d_out_affine, const void* d_bases_affine, size_t n_bases, const uint32_t* d_row_ptr,
                                       const uint32_t* d_col, const void* d_coeffs, size_t n_rows, size_t nnz, void* stream, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec<Fq>(d_out_affine, d_bases_affine, n_bases, d_row_ptr, d_col, d_coeffs, n_rows, nnz, stream, 1, false);
  });
}
int mi355zk_bn254_g2_sparse_matvec_dev(void* d_out_affine, const void* d_bases_affine, size_t n_bases, const uint32_t* d_row_ptr,
                                       const uint32_t* d_col, const void* d_coeffs, size_t n_rows, size_t nnz, void* stream, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec<Fq2>(d_out_affine, d_bases_affine, n_bases, d_row_ptr, d_col, d_coeffs, n_rows, nnz, stream, 2, (flags & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}

int mi355zk_bn254_g1_batch_exp(uint8_t* out_affine, const uint8_t* bases_affine, const uint64_t* scalars, size_t n, int mode) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp_host<Fq>(out_affine, bases_affine, scalars, n, mode & MI355ZK_EXP_SAME_SCALAR, false);
  });
}
int mi355zk_bn254_g2_batch_exp(uint8_t* out_affine, const uint8_t* bases_affine, const uint64_t* scalars, size_t n, int mode) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp_host<Fq2>(out_affine, bases_affine, scalars, n, mode & MI355ZK_EXP_SAME_SCALAR, (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}
int mi355zk_bn254_g1_batch_exp_dev(void* d_out_affine, const void* d_bases_affine, const void* d_scalars, size_t n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp<Fq>(d_out_affine, d_bases_affine, 0, d_scalars, mode & MI355ZK_EXP_SAME_SCALAR, n, stream);
  });
}
int mi355zk_bn254_g2_batch_exp_dev(void* d_out_affine, const void* d_bases_affine, const void* d_scalars, size_t n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp<Fq2>(d_out_affine, d_bases_affine, 0, d_scalars, mode & MI355ZK_EXP_SAME_SCALAR, n, stream, nullptr, false,
                          (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}

// host-side group helpers (joining per-GPU partial sums, normalising results)
int mi355zk_bn254_g1_add(uint64_t acc_xyz[12], const uint64_t other_xyz[12]) {
  return abi_guard([&]() -> int {
    if (!acc_xyz || !other_xyz) return ZK_ERR_BAD_ARGS;
    G1Jacobian a, b;
    std::memcpy(&a, acc_xyz, sizeof a);
    std::memcpy(&b, other_xyz, sizeof b);
    jac_add(a, b);
    std::memcpy(acc_xyz, &a, sizeof a);
    return ZK_OK;
  });
}
int mi355zk_bn254_g2_add(uint64_t acc_xyz[24], const uint64_t other_xyz[24]) {
  return abi_guard([&]() -> int {
    if (!acc_xyz || !other_xyz) return ZK_ERR_BAD_ARGS;
    G2Jacobian a, b;
    std::memcpy(&a, acc_xyz, sizeof a);
    std::memcpy(&b, other_xyz, sizeof b);
    jac_add(a, b);
    std::memcpy(acc_xyz, &a, sizeof a);
    return ZK_OK;
  });
}
// acc = k * acc on the host (CurveProjective::mul_assign, ec.rs:538-560: most significant bit first, leading zeros skipped): the
// handful of single-point products a proof assembly makes (prover.rs:300-333: vk.delta_g1.mul(r) ...)
extern "C++" template <class J>
static int host_scalar_mul(uint64_t* acc_xyz, const uint64_t k[4]) {
  if (!acc_xyz || !k) return ZK_ERR_BAD_ARGS;
  J base, res = J::zero();
  std::memcpy(&base, acc_xyz, sizeof base);
  bool found_one = false;
  for (int i = 255; i >= 0; --i) {
    const bool bit = (k[i >> 6] >> (i & 63)) & 1;
    if (found_one) jac_double(res);
    else found_one = bit;
    if (bit) jac_add(res, base);
  }
  std::memcpy(acc_xyz, &res, sizeof res);
  return ZK_OK;
}
int mi355zk_bn254_g1_mul(uint64_t acc_xyz[12], const uint64_t scalar[4]) { return abi_guard([&]() -> int { return host_scalar_mul<G1Jacobian>(acc_xyz, scalar); }); }
int mi355zk_bn254_g2_mul(uint64_t acc_xyz[24], const uint64_t scalar[4]) { return abi_guard([&]() -> int { return host_scalar_mul<G2Jacobian>(acc_xyz, scalar); }); }
// into_affine (ec.rs:596-629); infinity -> all-zero record
int mi355zk_bn254_g1_to_affine(uint64_t out_xy[8], const uint64_t xyz[12]) {
  return abi_guard([&]() -> int {
    if (!out_xy || !xyz) return ZK_ERR_BAD_ARGS;
    G1Jacobian p;
    std::memcpy(&p, xyz, sizeof p);
    G1Affine r{Fq::zero(), Fq::zero()};
    if (!p.is_zero()) {
      Fq zi = inv(p.z), zi2 = sqr(zi);
      r.x = mul(p.x, zi2);
      r.y = mul(p.y, mul(zi2, zi));
    }
    std::memcpy(out_xy, &r, sizeof r);
    return ZK_OK;
  });
}
int mi355zk_bn254_g2_to_affine(uint64_t out_xy[16], const uint64_t xyz[24]) {
  return abi_guard([&]() -> int {
    if (!out_xy || !xyz) return ZK_ERR_BAD_ARGS;
    G2Jacobian p;
    std::memcpy(&p, xyz, sizeof p);
    G2Affine r{Fq2::zero(), Fq2::zero()};
    if (!p.is_zero()) {
      Fq2 zi = inv(p.z), zi2 = sqr(zi);
      r.x = mul(p.x, zi2);
      r.y = mul(p.y, mul(zi2, zi));
    }
    std::memcpy(out_xy, &r, sizeof r);
    return ZK_OK;
  });
}

int mi355zk_malloc(void** d_ptr, size_t bytes) {
  return abi_guard([&]() -> int {
    if (!d_ptr) return ZK_ERR_BAD_ARGS;
    ZK_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ZK_OK;
  });
}
int mi355zk_free(void* d_ptr) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipFree(d_ptr));
    return ZK_OK;
  });
}
int mi355zk_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ZK_OK;
  });
}
int mi355zk_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
  });
}
int mi355zk_sync(void* stream) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZK_OK;
  });
}

void mi355zk_prof_enable(int on) { abi_guard_void([&] { prof_enable(on < 0 ? 0 : on > 2 ? 1 : on); }); }
void mi355zk_prof_only(const char* name) { abi_guard_void([&] { prof_only(name); }); }
void mi355zk_prof_reset(void) { abi_guard_void([&] { prof_reset(); }); }
int mi355zk_prof_get(const char* kernel, double* total_ms, long* count) {
  return abi_guard([&]() -> int {
    double t = 0;
    long c = 0;
    bool ok = kernel && prof_get(kernel, &t, &c);
    if (total_ms) *total_ms = t;
    if (count) *count = c;
    return ok ? ZK_OK : ZK_ERR_BAD_ARGS;
  });
}

}  // extern "C"
