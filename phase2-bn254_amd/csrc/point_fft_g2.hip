// Radix-2 FFT over CURVE POINTS of BN254 G2: EvaluationDomain<Point<G2>>::{fft, ifft}.
//
// Reference path: bellman/src/group.rs:22-51 under bellman/src/domain.rs:154-173,274-317, driven by
// powersoftau/src/bin/prepare_phase2.rs:68-131 (the tau-powers in G2 -> Lagrange basis, `coeffs_g2`).
// Same network and the same program as point_fft.hip (G1): bit-reversed load into a working array of JACOBIAN
// points, one lane per butterfly per stage, the twiddle multiplication by fixed signed 4-bit windows over a per-lane
// table {1..8} * t in scratch ([entry][lane]) so that the lanes of a wave add at the same places, affine raw records
// (128 B, all-zero = infinity) in and out with one inversion per 8 points.  The group law runs on the U-form Fq2 Jacobian
// arithmetic of curveu.hpp (JacU2: 29-bit lazy limbs, 2^261 domain; round 1 ran the memory-format Fq2 formulas at half the
// rate): the working array holds U-form points between the stages; table build, doublings and the closing u + t / u - t share
// ONE inlined jacu2_double and ONE inlined jacu2_add_tab (the Fq2 group law is > 100 KB of gfx950 code per copy, and
// out-of-line calls with these operands go through scratch and crawl).
#include <hip/hip_runtime.h>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "glv.hpp"
#include "device_util.hpp"

namespace zk {

// api.hip: io[i] = (X, Y), z[i] = Z  ->  affine records, 8 points per inversion
int batch_normalize_g2(void* d_io_affine, const void* d_z, uint64_t n, hipStream_t st);

namespace {

struct alignas(16) J2 {   // a working-array point: U-form Jacobian, 216 bytes + padding to whole 16-byte words
  JacU2 p;
  uint32_t pad[2];
};
static_assert(sizeof(J2) == 224 && sizeof(JacTabU2) % 16 == 0, "16-byte copies");

template <class T>
__device__ __forceinline__ T v_load(const T* p) {
  T r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) d[i] = q[i];
  return r;
}
template <class T>
__device__ __forceinline__ void v_store(T* p, const T& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) d[i] = s[i];
}
__device__ __forceinline__ J2 j2_of(const JacU2& q) {
  J2 r;
  r.p = q;
  r.pad[0] = r.pad[1] = 0;
  return r;
}

__global__ void __launch_bounds__(256) pfft2_load_kernel(const G2Affine* __restrict__ in, J2* __restrict__ work, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  G2Affine a = in[i];
  JacU2 v = JacU2::zero();
  if (!a.is_zero()) {
    const JacTabU2 e = jacu2_tab_from_affine(a.x, a.y);    // (x, y, one) in the 2^261 domain
    v = JacU2{e.x, e.y, e.z};
  }
  uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  v_store(work + r, j2_of(v));
}

// The program of point_fft.hip's pfft_stage_kernel, on U-form Jacobian points:
//   steps 0..6 table (2t .. 8t), then the windows of the twiddle, then entry 1 := product, a[i0] = u + product, a[i1] = u - product
//   (mode 0, domain.rs:303-309);  mode 1: every point times the scalar `c` (ifft's 1/m, domain.rs:163-173).
// SPLIT = false (the default): 64 plain signed 4-bit windows of the canonical twiddle -- four doublings and a table addition each, the
//   group law only, so the transform is the reference's (group.rs:38-51 over wnaf.rs:4-71) for EVERY vector of points of the twist.
// SPLIT = true (MI355ZK_G2_TRUSTED_SUBGROUP): 33 windows of the twiddle split over psi (glv.hpp: w t = k1 t + k2 psi(t)) -- four doublings,
//   the k1 digit, the k2 digit through psi; psi(t) = mu t holds in the order-r subgroup only, and a transform of subgroup points stays in it.
template <bool SPLIT>
__global__ void __launch_bounds__(256) pfft2_stage_kernel(J2* __restrict__ work, const uint32_t* __restrict__ tw_canon, uint32_t log_n,
                                                         uint32_t s, uint64_t b0, uint64_t n_chunk, JacTabU2* __restrict__ tab, int mode, Fr c) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunk) return;
  const uint64_t b = b0 + t;
  uint64_t i0, i1;
  uint32_t kk[8];
  bool unit = false;
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
  const JacU2 u = mode == 0 ? v_load(work + i0).p : JacU2::zero();
  JacU2 acc = v_load(work + i1).p;
  // signed 4-bit digits: of both halves of the split twiddle (k1, k2 < 2^128), or of the twiddle itself (canonical, < r < 2^254: 64 nibbles, no carry out)
  constexpr int NW = SPLIT ? 5 : 8;
  uint32_t mag1[NW], mag2[SPLIT ? 5 : 1], sgn1[2], sgn2[2];
  Fq2U cxU, cyU;
  if constexpr (SPLIT) {
    const Glv2Split g = glv2_split(kk);
    signed_nibbles<5, 5>(g.k1, mag1, sgn1);
    signed_nibbles<5, 5>(g.k2, mag2, sgn2);
    const FqU C266 = UPow2<FqParams, 266>::get();
    const Fq2 cxs = glv2_cx(), cys = glv2_cy();
    cxU = Fq2U{u_mul(u_from_std(cxs.c0), C266), u_mul(u_from_std(cxs.c1), C266)};   // 2^261 domain, < 2p
    cyU = Fq2U{u_mul(u_from_std(cys.c0), C266), u_mul(u_from_std(cys.c1), C266)};
  } else {
    signed_nibbles<8, 8>(kk, mag1, sgn1);
  }
  const bool t_inf = acc.is_zero();
  if (t_inf && mode == 1) return;
  if (!t_inf) v_store(tab + t, jacu2_tab_entry(acc));
  constexpr uint32_t PROG[7] = {0x1102, 0x0013, 0x2104, 0x0015, 0x3106, 0x0017, 0x4108};  // nibbles: load, double, add, store
  constexpr int MAIN0 = 7, WINDOWS = SPLIT ? 33 : 64, PER = SPLIT ? 5 : 4, STEP_STORE = MAIN0 + PER * WINDOWS, STEP_SUM = STEP_STORE + 1, STEP_DIF = STEP_STORE + 2;
  const int first = (unit || t_inf) ? STEP_SUM : 0;   // twiddle one, or t = infinity: the product is t itself (entry 1 already holds it)
  const int last = mode == 0 ? STEP_DIF : STEP_STORE - 1;
#pragma unroll 1
  for (int step = first; step <= last; ++step) {
    uint32_t load = 0, dbl_it = 0, add = 0, store = 0, negate = 0, psi = 0;
    if (step < MAIN0) {
      const uint32_t pr = PROG[step];
      load = pr >> 12;
      dbl_it = (pr >> 8) & 15u;
      add = (pr >> 4) & 15u;
      store = pr & 15u;
    } else if (step < STEP_STORE) {
      const int m = step - MAIN0;   // per window: four doublings (the fourth adds the k1 digit), then (SPLIT) the k2 digit through psi
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
    } else if (step == STEP_STORE) {
      store = acc.is_zero() ? 0u : 1u;   // (an infinite product: nothing to add below)
      if (!store) { v_store(work + i0, j2_of(u)); v_store(work + i1, j2_of(u)); break; }
    } else {
      acc = u;
      add = t_inf ? 0u : 1u;             // u +- infinity = u
      negate = step == STEP_DIF;
    }
    if (load) {
      const JacTabU2 e = v_load(tab + (uint64_t)(load - 1) * n_chunk + t);
      acc = JacU2{e.x, e.y, e.z};
    }
    if (dbl_it) acc = jacu2_double(acc);
    if (add) {
      JacTabU2 e = v_load(tab + (uint64_t)(add - 1) * n_chunk + t);
      if constexpr (SPLIT) {
        if (psi) e = jacu2_tab_psi(e, cxU, cyU);
        jacu2_add_tab(acc, e, negate != 0);
      } else {
        if (!e.z.limbs_all_zero()) jacu2_add_tab(acc, e, negate != 0);   // (an infinite multiple d t, d <= 8: never on the twist, whose order is odd and has no small factor)
      }
    }
    if (store) v_store(tab + (uint64_t)(store - 1) * n_chunk + t, jacu2_tab_entry(acc));
    if (step == STEP_SUM) v_store(work + i0, j2_of(acc));
    if (step == STEP_DIF) v_store(work + i1, j2_of(acc));
  }
  if (mode == 1) v_store(work + i0, j2_of(acc));
}

__global__ void __launch_bounds__(256) pfft2_store_kernel(const J2* __restrict__ work, G2Affine* __restrict__ out, Fq2* __restrict__ zbuf,
                                                         uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  const Jacobian<Fq2> r = jacu2_to_std(v_load(work + i).p);
  out[i] = G2Affine{r.x, r.y};
  zbuf[i] = r.z;
}

__global__ void pfft2_twiddle_kernel(uint32_t* tw, Fr omega, uint64_t count) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fr c = to_canonical(pow_u64(omega, e));
#pragma unroll
  for (int l = 0; l < 8; ++l) tw[e * 8 + l] = c.l[l];
}

}  // namespace

// d_points: 2^log_n affine raw G2 records (128 B), in place.  scale: every output is multiplied by scale_canon (ifft: m^-1).
int point_fft_g2(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st, bool trusted_subgroup) {
  const uint64_t n = 1ull << log_n;
  const uint64_t lanes_max = scale ? n : (n >= 2 ? n / 2 : 1);
  const uint64_t chunk = lanes_max < (1ull << 19) ? lanes_max : (1ull << 19);  // table: 8 x 368 B per lane
  char* buf = nullptr;
  const size_t o_work = 0, o_tw = o_work + ((n * sizeof(J2) + 255) & ~(size_t)255), o_z = o_tw + (((n / 2 + 1) * 32 + 255) & ~(size_t)255),
               o_tab = o_z + ((n * sizeof(Fq2) + 255) & ~(size_t)255), total = o_tab + 8 * chunk * sizeof(JacTabU2);
  ZK_HIP(hipMalloc(&buf, total));
  J2* work = (J2*)(buf + o_work);
  uint32_t* tw = (uint32_t*)(buf + o_tw);
  Fq2* zbuf = (Fq2*)(buf + o_z);
  JacTabU2* tab = (JacTabU2*)(buf + o_tab);
  if (n >= 2) hipLaunchKernelGGL(pfft2_twiddle_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, tw, omega, n / 2);
  hipLaunchKernelGGL(pfft2_load_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const G2Affine*)d_points, work, log_n);
  for (uint32_t s = 0; s < log_n; ++s)
    for (uint64_t b0 = 0; b0 < n / 2; b0 += chunk) {
      const uint64_t m = n / 2 - b0 < chunk ? n / 2 - b0 : chunk;
      if (trusted_subgroup)
        hipLaunchKernelGGL(pfft2_stage_kernel<true>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, s, b0, m, tab, 0, Fr::zero());
      else
        hipLaunchKernelGGL(pfft2_stage_kernel<false>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, s, b0, m, tab, 0, Fr::zero());
    }
  if (scale)
    for (uint64_t b0 = 0; b0 < n; b0 += chunk) {
      const uint64_t m = n - b0 < chunk ? n - b0 : chunk;
      if (trusted_subgroup)
        hipLaunchKernelGGL(pfft2_stage_kernel<true>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, 0u, b0, m, tab, 1, scale_canon);
      else
        hipLaunchKernelGGL(pfft2_stage_kernel<false>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, work, tw, log_n, 0u, b0, m, tab, 1, scale_canon);
    }
  hipLaunchKernelGGL(pfft2_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, (G2Affine*)d_points, zbuf, log_n);
  hipError_t e = hipGetLastError();
  int rc = e == hipSuccess ? batch_normalize_g2(d_points, zbuf, n, st) : ZK_ERR_DEVICE;
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  ZK_HIP(e);
  return rc;
}

}  // namespace zk
