// Radix-2 FFT over CURVE POINTS of BN254 G2: EvaluationDomain<Point<G2>>::{fft, ifft}.
//
// Reference path: bellman/src/group.rs:22-51 under bellman/src/domain.rs:154-173,274-317, driven by
// powersoftau/src/bin/prepare_phase2.rs:68-131 (the tau-powers in G2 -> Lagrange basis, `coeffs_g2`).
// Same network and data flow as point_fft.hip (G1): bit-reversed load into a working array of XYZZ points,
// one lane per butterfly per stage with the twiddle scalar multiplication done by double-and-add, affine raw
// records (128 B, all-zero = infinity) in and out.  The group law runs on the memory-format Fq2 arithmetic
// (curve.hpp / field.hpp); every butterfly is ~380 G2 operations, pure integer-ALU work.
#include <hip/hip_runtime.h>

#include "../../include/mi355zk.h"
#include "curve.hpp"
#include "device_util.hpp"

namespace zk {
namespace {

using P2 = XYZZ<Fq2>;

__device__ __forceinline__ P2 p2_load(const P2* p) {
  P2 r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(P2) / 16); ++i) d[i] = q[i];
  return r;
}
__device__ __forceinline__ void p2_store(P2* p, const P2& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(P2) / 16); ++i) d[i] = s[i];
}

// Scalar multiplication is MSB-first double-and-add (the group element ec.rs:544-563 computes).  Each kernel below
// keeps exactly ONE inlined xyzz_add and ONE inlined xyzz_double: the Fq2 group law is ~150 KB of gfx950 code per
// copy, and out-of-line (noinline) device calls with these 256-byte operands go through scratch and crawl.
__global__ void __launch_bounds__(256) pfft2_load_kernel(const G2Affine* __restrict__ in, P2* __restrict__ work, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  G2Affine a = in[i];
  P2 v = P2::zero();
  if (!a.is_zero()) {
    v.x = a.x;
    v.y = a.y;
    v.zz = Fq2::one();
    v.zzz = Fq2::one();
  }
  uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  p2_store(work + r, v);
}

// stage s (m = 2^s):  t = w^(j * n/2m) * a[k+j+m];  a[k+j+m] = a[k+j] - t;  a[k+j] += t   (domain.rs:303-309).
// Iterations 0..255 are the bits of the twiddle (skipped for w^0 = 1), iterations 256 / 257 the sum and the difference.
__global__ void __launch_bounds__(256) pfft2_stage_kernel(P2* __restrict__ work, const uint32_t* __restrict__ tw_canon, uint32_t log_n,
                                                         uint32_t s) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (1u << (log_n - 1))) return;
  const uint32_t m = 1u << s, j = b & (m - 1);
  const uint32_t i0 = ((b >> s) << (s + 1)) + j, i1 = i0 + m;
  const P2 u = p2_load(work + i0);
  const P2 t = p2_load(work + i1);
  const uint32_t* k = tw_canon + ((uint64_t)j << (log_n - 1 - s)) * 8;
  P2 acc = j == 0 ? t : P2::zero();
  bool found = false;
  for (int it = j == 0 ? 256 : 0; it < 258; ++it) {
    P2 A, B;
    bool do_add = true;
    if (it < 256) {
      const int bit = 255 - it;
      const bool on = (k[bit >> 5] >> (bit & 31)) & 1;
      if (found) acc = xyzz_double(acc);
      else found = on;
      do_add = on;
      A = acc;
      B = t;
    } else {
      A = u;
      B = acc;
      if (it == 257) B.y = neg(B.y);
    }
    if (do_add) {
      xyzz_add(A, B);
      if (it < 256) acc = A;
      else p2_store(work + (it == 256 ? i0 : i1), A);
    }
  }
}

__global__ void __launch_bounds__(256) pfft2_store_kernel(const P2* __restrict__ work, G2Affine* __restrict__ out, uint32_t log_n, int scale,
                                                         Fr c_canon) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  P2 v = p2_load(work + i);
  if (scale) {
    const P2 base = v;
    bool found = false;
    v = P2::zero();
    for (int bit = 255; bit >= 0; --bit) {
      const bool on = (c_canon.l[bit >> 5] >> (bit & 31)) & 1;
      if (found) v = xyzz_double(v);
      else found = on;
      if (on) xyzz_add(v, base);
    }
  }
  out[i] = xyzz_to_affine(v);
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
int point_fft_g2(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st) {
  const uint64_t n = 1ull << log_n;
  P2* work = nullptr;
  uint32_t* tw = nullptr;
  ZK_HIP(hipMalloc(&work, n * sizeof(P2)));
  hipError_t e = hipMalloc(&tw, (n / 2 + 1) * 32);
  if (e != hipSuccess) { (void)hipFree(work); ZK_HIP(e); }
  auto fail = [&](hipError_t err) { (void)hipFree(work); (void)hipFree(tw); return err; };
  if (n >= 2) hipLaunchKernelGGL(pfft2_twiddle_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, tw, omega, n / 2);
  hipLaunchKernelGGL(pfft2_load_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const G2Affine*)d_points, work, log_n);
  for (uint32_t s = 0; s < log_n; ++s)
    hipLaunchKernelGGL(pfft2_stage_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, work, tw, log_n, s);
  hipLaunchKernelGGL(pfft2_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, (G2Affine*)d_points, log_n, scale ? 1 : 0,
                     scale_canon);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  ZK_HIP(fail(e));
  return ZK_OK;
}

}  // namespace zk
