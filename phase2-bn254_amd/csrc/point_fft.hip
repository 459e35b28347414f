// Radix-2 FFT over CURVE POINTS (BN254 G1) for gfx950: EvaluationDomain<Point<G1>>::{fft, ifft}.
//
// Reference path: bellman/src/group.rs:22-51 (`Point<G>`: group_mul_assign = scalar
// multiplication of a projective point by an Fr twiddle, add, sub) under bellman/src/domain.rs:154-173,274-317,
// driven by powersoftau/src/bin/prepare_phase2.rs:68-131 (affine tau-powers -> ifft -> batch_normalization ->
// Lagrange-basis points) -- the dominant cost of `prepare_phase2`.
//
// Every butterfly is a 254-bit scalar multiplication (split in two 128-bit halves by the curve's endomorphism, glv.hpp), so the
// work is n/2 * log n * ~300k integer mads: pure ALU.
// Layout: a working array of U-form JACOBIAN points (curveu.hpp JacU, every coordinate in the 2^261 domain, 112 B per
// point) in HBM; one lane per butterfly per stage, DIT after a bit-reversed load.  The twiddle multiplication uses
// fixed signed 4-bit windows over a per-lane table {1..8} * t in a scratch array laid out [entry][lane] (JacTabU:
// Jacobian + Z^2 + Z^3), so all 64 lanes of a wave add at the same 64 places although their twiddles differ --
// plain double-and-add would make the wave pay an addition on nearly every bit.  Table build, the 256 doublings and
// the closing u + t / u - t run through ONE loop with a single inlined doubling and a single inlined addition.
// Input and output are affine raw records (64 B, all-zero = infinity); the output is normalised with one inversion
// per 16 points (api.hip batch_normalize), i.e. it is what `batch_normalization` + `into_affine` leave
// (ec.rs:251-299, 596-629), which makes parity bit-exact.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "glv.hpp"
#include "device_util.hpp"

namespace zk {

// api.hip: io[i] = (X, Y), z[i] = Z  ->  affine records, 16 points per inversion
int batch_normalize_g1(void* d_io_affine, const void* d_z, uint64_t n, hipStream_t st);

namespace {

using JU = JacU<FqParams>;
using TU = JacTabU<FqParams>;

struct alignas(16) PtJ {
  uint32_t w[28];  // x, y, z (9 limbs each) + 1 pad
};
__device__ __forceinline__ JU pt_load(const PtJ* p) {
  PtJ t;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&t);
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = q[i];
  JU r;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r.x.l[i] = t.w[i]; r.y.l[i] = t.w[9 + i]; r.z.l[i] = t.w[18 + i]; }
  return r;
}
__device__ __forceinline__ void pt_store(PtJ* p, const JU& v) {
  PtJ t;
#pragma unroll
  for (int i = 0; i < 9; ++i) { t.w[i] = v.x.l[i]; t.w[9 + i] = v.y.l[i]; t.w[18 + i] = v.z.l[i]; }
  t.w[27] = 0;
  const uint4* s = reinterpret_cast<const uint4*>(&t);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = s[i];
}

// affine raw records -> working points at the bit-reversed position (domain.rs:288-293)
__global__ void __launch_bounds__(256) pfft_load_kernel(const G1Affine* __restrict__ in, PtJ* __restrict__ work, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  G1Affine a = in[i];
  JU v = JU::zero();
  if (!a.is_zero()) {
    const FqU c266 = UPow2<FqParams, 266>::get();   // x*2^256 * 2^266 / 2^261 = x * 2^261
    v.x = u_mul(u_from_std(a.x), c266);
    v.y = u_mul(u_from_std(a.y), c266);
    v.z = UPow2<FqParams, 261>::get();              // one
  }
  uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  pt_store(work + r, v);
}

// signed 4-bit digits of a magnitude m < 2^128 (5 limbs): m = sum d_j 16^j, d_j in [-8, 8], j <= 32; magnitudes as nibbles, signs as bits
__device__ __forceinline__ void recode16(const uint32_t* k, uint32_t mag[5], uint32_t sgn[2]) {
  sgn[0] = sgn[1] = 0;
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < 5; ++w) {
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t d = ((k[w] >> (4 * q)) & 15u) + carry;
      carry = d > 8u ? 1u : 0u;
      if (carry) {
        d = 16u - d;
        sgn[w >> 2] |= 1u << (8 * (w & 3) + q);
      }
      m |= d << (4 * q);
    }
    mag[w] = m;
  }
}

// The shared program.  Entries 1..8 of the lane's table hold 1t..8t (entry 1 is rewritten with the product at the end).  The twiddle is
// split by the GLV endomorphism (glv.hpp): w t = k1 t + k2 phi(t), |k1|, |k2| < 2^128, phi(X, Y, Z) = (beta X, Y, Z) -- 132 doublings
// instead of 256.
//   steps 0..6      table:  2t = 2*1t, 3t = 2t + 1t, 4t = 2*2t, 5t = 4t + 1t, 6t = 2*3t, 7t = 6t + 1t, 8t = 2*4t
//   steps 7..171    33 windows of five steps: four doublings (the fourth adds the window's k1 digit entry), then the k2 digit entry
//                   with X multiplied by beta
//   step  172       (mode butterfly) entry 1 := product;  steps 173 / 174: a[i0] = u + product, a[i1] = u - product
// mode 0: butterfly of stage s (domain.rs:303-309);  mode 1: every point times the scalar `c` (ifft's 1/m, domain.rs:163-173)
__global__ void __launch_bounds__(256) pfft_stage_kernel(PtJ* __restrict__ work, const uint32_t* __restrict__ tw_canon, uint32_t log_n,
                                                        uint32_t s, uint64_t b0, uint64_t n_chunk, TU* __restrict__ tab, int mode, Fr c) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunk) return;
  const uint64_t b = b0 + t;
  uint64_t i0, i1;
  uint32_t kk[8];
  bool unit = false;  // twiddle w^0 = 1: no multiplication
  if (mode == 0) {
    const uint64_t m = 1ull << s, j = b & (m - 1);
    i0 = ((b >> s) << (s + 1)) + j;
    i1 = i0 + m;
    unit = j == 0;
    const uint32_t* kp = tw_canon + (j << (log_n - 1 - s)) * 8;
#pragma unroll
    for (int l = 0; l < 8; ++l) kk[l] = kp[l];
  } else {
    i0 = i1 = b;
#pragma unroll
    for (int l = 0; l < 8; ++l) kk[l] = c.l[l];
  }
  const JU u = mode == 0 ? pt_load(work + i0) : JU::zero();
  JU acc = pt_load(work + i1);
  const GlvSplit g = glv_split(kk);
  uint32_t mag1[5], sgn1[2], mag2[5], sgn2[2];
  recode16(g.k1, mag1, sgn1);
  recode16(g.k2, mag2, sgn2);
  const FqU betaU = u_mul(u_from_std(glv_beta()), UPow2<FqParams, 266>::get());   // beta, 2^261 domain
  const bool t_inf = acc.is_zero();
  if (t_inf && mode == 1) return;
  TU e1 = jacu_tab_entry(acc);  // 1t (an infinity t keeps z == 0: every sum below then returns the other operand)
  tab[t] = e1;
  constexpr uint32_t PROG[7] = {0x1102, 0x0013, 0x2104, 0x0015, 0x3106, 0x0017, 0x4108};  // nibbles: load, double, add, store
  constexpr int MAIN0 = 7, WINDOWS = 33, STEP_STORE = MAIN0 + 5 * WINDOWS, STEP_SUM = STEP_STORE + 1, STEP_DIF = STEP_STORE + 2;
  const int first = (unit || t_inf) ? STEP_STORE : 0;
  const int last = mode == 0 ? STEP_DIF : STEP_STORE - 1;
#pragma unroll 1
  for (int step = first; step <= last; ++step) {
    uint32_t load = 0, dbl_it = 0, add = 0, store = 0, negate = 0, phi = 0;
    if (step < MAIN0) {
      const uint32_t pr = PROG[step];
      load = pr >> 12;
      dbl_it = (pr >> 8) & 15u;
      add = (pr >> 4) & 15u;
      store = pr & 15u;
    } else if (step < STEP_STORE) {
      const int m = step - MAIN0;
      if (m == 0) acc = JU::zero();
      const int win = m / 5, sub = m - 5 * win, j = WINDOWS - 1 - win;
      if (sub < 4) {
        dbl_it = 1;
        if (sub == 3) {
          add = (mag1[j >> 3] >> (4 * (j & 7))) & 15u;
          negate = (((sgn1[j >> 5] >> (j & 31)) & 1u) != 0) != g.neg1;
        }
      } else {
        add = (mag2[j >> 3] >> (4 * (j & 7))) & 15u;
        negate = (((sgn2[j >> 5] >> (j & 31)) & 1u) != 0) != g.neg2;
        phi = 1;
      }
    } else if (step == STEP_STORE) {
      store = 1;                       // the product (or t itself when the twiddle is 1) becomes entry 1
    } else {
      acc = u;
      add = 1;
      negate = step == STEP_DIF;
    }
    if (load) {
      const TU e = tab[(uint64_t)(load - 1) * n_chunk + t];
      acc = JU{e.x, e.y, e.z};
    }
    if (dbl_it) acc = jacu_double(acc);
    if (add) {
      TU e = tab[(uint64_t)(add - 1) * n_chunk + t];
      if (phi) e.x = u_mul(e.x, betaU);                  // X < 6p: < 1.08p
      if (!e.z.limbs_all_zero()) jacu_add_tab(acc, e, negate != 0);
    }
    if (store) tab[(uint64_t)(store - 1) * n_chunk + t] = jacu_tab_entry(acc);
    if (step == STEP_SUM) pt_store(work + i0, acc);
    if (step == STEP_DIF) pt_store(work + i1, acc);
  }
  if (mode == 1) pt_store(work + i0, acc);
}

// working points -> (X, Y) in the output record and Z in zbuf, memory format; batch_normalize_g1 finishes
__global__ void __launch_bounds__(256) pfft_store_kernel(const PtJ* __restrict__ work, G1Affine* __restrict__ out, Fq* __restrict__ zbuf,
                                                        uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  const G1Jacobian r = jacu_to_std(pt_load(work + i));
  out[i] = G1Affine{r.x, r.y};
  zbuf[i] = r.z;
}

// tw[e] = canonical(omega^e), e < count
__global__ void pfft_twiddle_kernel(uint32_t* tw, Fr omega, uint64_t count) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fr c = to_canonical(pow_u64(omega, e));
#pragma unroll
  for (int l = 0; l < 8; ++l) tw[e * 8 + l] = c.l[l];
}

}  // namespace

// d_points: 2^log_n affine raw records, in place.  scale: every output is multiplied by scale_canon (ifft: m^-1).
int point_fft_g1(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st) {
  const uint64_t n = 1ull << log_n;
  const uint64_t lanes_max = scale ? n : (n >= 2 ? n / 2 : 1);
  const uint64_t chunk = lanes_max < (1ull << 20) ? lanes_max : (1ull << 20);  // table: 8 x 192 B per lane
  char* buf = nullptr;
  const size_t o_work = 0, o_tw = o_work + ((n * sizeof(PtJ) + 255) & ~(size_t)255), o_z = o_tw + (((n / 2 + 1) * 32 + 255) & ~(size_t)255),
               o_tab = o_z + ((n * sizeof(Fq) + 255) & ~(size_t)255), total = o_tab + 8 * chunk * sizeof(TU);
  ZK_HIP(hipMalloc(&buf, total));
  PtJ* work = (PtJ*)(buf + o_work);
  uint32_t* tw = (uint32_t*)(buf + o_tw);
  Fq* zbuf = (Fq*)(buf + o_z);
  TU* tab = (TU*)(buf + o_tab);
  if (n >= 2) hipLaunchKernelGGL(pfft_twiddle_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, tw, omega, n / 2);
  hipLaunchKernelGGL(pfft_load_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const G1Affine*)d_points, work, log_n);
  for (uint32_t s = 0; s < log_n; ++s)
    for (uint64_t b0 = 0; b0 < n / 2; b0 += chunk) {
      const uint64_t m = n / 2 - b0 < chunk ? n / 2 - b0 : chunk;
      hipLaunchKernelGGL(pfft_stage_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, s, b0, m, tab, 0, Fr::zero());
    }
  if (scale)
    for (uint64_t b0 = 0; b0 < n; b0 += chunk) {
      const uint64_t m = n - b0 < chunk ? n - b0 : chunk;
      hipLaunchKernelGGL(pfft_stage_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, 0u, b0, m, tab, 1, scale_canon);
    }
  hipLaunchKernelGGL(pfft_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, (G1Affine*)d_points, zbuf, log_n);
  hipError_t e = hipGetLastError();
  int rc = e == hipSuccess ? batch_normalize_g1(d_points, zbuf, n, st) : ZK_ERR_DEVICE;
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  ZK_HIP(e);
  return rc;
}

}  // namespace zk
