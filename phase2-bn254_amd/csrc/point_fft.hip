// Radix-2 FFT over CURVE POINTS (BN254 G1) for gfx950: EvaluationDomain<Point<G1>>::{fft, ifft}.
//
// Reference path: bellman/src/group.rs:22-51 (`Point<G>`: group_mul_assign = scalar
// multiplication of a projective point by an Fr twiddle, add, sub) under bellman/src/domain.rs:154-173,274-317,
// driven by powersoftau/src/bin/prepare_phase2.rs:68-131 (affine tau-powers -> ifft -> batch_normalization ->
// Lagrange-basis points) -- the dominant cost of `prepare_phase2`.
//
// Every butterfly is a 254-bit scalar multiplication, so the work is n/2 * log n * ~3700 field products: pure
// integer-ALU work.  Layout: a working array of XYZZ points in U-form (fieldu.hpp, all four coordinates in the
// 2^261 domain, 144 B per point) in HBM; one lane per butterfly per stage, DIT after a bit-reversed load;
// twiddles come from a table of CANONICAL scalars w^e (e < n/2).  Input and output are affine raw records
// (64 B, all-zero = infinity), i.e. the output is what `batch_normalization` + `into_affine` leave
// (ec.rs:251-299, 596-629), which makes parity bit-exact.
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "device_util.hpp"

namespace zk {
namespace {

using XU = XYZZU<FqParams>;

// a + b for two accumulators whose coordinates all live in the 2^261 domain (add-2008-s), complete.
// Invariants in and out: X < 6p, Y < 2p, ZZ < 2p, ZZZ < 2p, N-form (same bookkeeping as curveu.hpp).
__device__ XU xu_add(const XU& a, const XU& b) {
  if (a.is_zero()) return b;
  if (b.is_zero()) return a;
  FqU u1 = u_mul(a.x, b.zz);                       // 6 * 2 c + 1 < 1.08p
  FqU u2 = u_mul(b.x, a.zz);
  FqU s1 = u_mul(a.y, b.zzz);                      // < 1.03p
  FqU s2 = u_mul(b.y, a.zzz);
  FqU p = u_sub<2, 1>(u2, u1);                     // < 3.1p, N
  FqU r = u_sub<2, 1>(s2, s1);                     // < 3.1p, N
  FqU pp = u_sqr(p);                               // < 1.06p
  FqU ppp = u_mul(p, pp);                          // < 1.02p
  FqU q = u_mul(u1, pp);                           // < 1.01p
  FqU rr = u_sqr(r);                               // < 1.06p
  XU o;
  o.x = u_sub<4, 3>(rr, u_add(ppp, u_dbl(q)));     // PPP + 2Q < 3.1p, limbs < 3 * 2^29;  X3 < 5.1p
  FqU d = u_sub<8, 1>(q, o.x);                     // < 9.1p
  FqU ns1 = u_sub<2, 1>(FqU::zero(), s1);          // 2p - S1
  o.y = u_mul2(r, d, ns1, ppp);                    // R*D - S1*PPP: (3.1 * 9.1 + 2 * 1.02) c + 1 < 1.2p
  o.zz = u_mul(u_mul(a.zz, b.zz), pp);             // < 2p
  o.zzz = u_mul(u_mul(a.zzz, b.zzz), ppp);
  if (u_is_zero_lt2p(o.zz)) {                      // P == 0: same x (ec.rs:398-408)
    if (u_is_zero_lt8p(r)) return xyzzu_double(a);
    return XU::zero();
  }
  return o;
}

__device__ __forceinline__ XU xu_neg(const XU& a) {
  if (a.is_zero()) return a;
  XU r = a;
  r.y = u_sub<2, 1>(FqU::zero(), a.y);             // 2p - Y
  return r;
}

// k * b, k a canonical 256-bit scalar (MSB-first double-and-add: the group element ec.rs:544-563 computes)
__device__ XU xu_mul(const XU& b, const uint32_t k[8]) {
  XU acc = XU::zero();
  bool found = false;
  for (int bit = 255; bit >= 0; --bit) {
    bool on = (k[bit >> 5] >> (bit & 31)) & 1;
    if (found) acc = xyzzu_double(acc);
    else found = on;
    if (on) acc = xu_add(acc, b);
  }
  return acc;
}

struct alignas(16) PtU {
  uint32_t w[36];
};
__device__ __forceinline__ XU pt_load(const PtU* p) {
  PtU t;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&t);
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = q[i];
  XU r;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r.x.l[i] = t.w[i]; r.y.l[i] = t.w[9 + i]; r.zz.l[i] = t.w[18 + i]; r.zzz.l[i] = t.w[27 + i]; }
  return r;
}
__device__ __forceinline__ void pt_store(PtU* p, const XU& v) {
  PtU t;
#pragma unroll
  for (int i = 0; i < 9; ++i) { t.w[i] = v.x.l[i]; t.w[9 + i] = v.y.l[i]; t.w[18 + i] = v.zz.l[i]; t.w[27 + i] = v.zzz.l[i]; }
  const uint4* s = reinterpret_cast<const uint4*>(&t);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = s[i];
}

// affine raw records -> working points at the bit-reversed position (domain.rs:288-293)
__global__ void __launch_bounds__(256) pfft_load_kernel(const G1Affine* __restrict__ in, PtU* __restrict__ work, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  G1Affine a = in[i];
  XU v = XU::zero();
  if (!a.is_zero()) {
    const FqU c266 = UPow2<FqParams, 266>::get();   // x*2^256 * 2^266 / 2^261 = x * 2^261
    v.x = u_mul(u_from_std(a.x), c266);
    v.y = u_mul(u_from_std(a.y), c266);
    v.zz = UPow2<FqParams, 261>::get();             // one
    v.zzz = v.zz;
  }
  uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  pt_store(work + r, v);
}

// stage s (m = 2^s):  t = w^(j * n/2m) * a[k+j+m];  a[k+j+m] = a[k+j] - t;  a[k+j] += t   (domain.rs:303-309)
__global__ void __launch_bounds__(256) pfft_stage_kernel(PtU* __restrict__ work, const uint32_t* __restrict__ tw_canon, uint32_t log_n,
                                                        uint32_t s) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (1u << (log_n - 1))) return;
  const uint32_t m = 1u << s, j = b & (m - 1);
  const uint32_t i0 = ((b >> s) << (s + 1)) + j, i1 = i0 + m;
  XU u = pt_load(work + i0);
  XU t = pt_load(work + i1);
  if (j != 0) {  // w^0 = 1
    uint32_t k[8];
    const uint32_t* kp = tw_canon + ((uint64_t)j << (log_n - 1 - s)) * 8;
#pragma unroll
    for (int l = 0; l < 8; ++l) k[l] = kp[l];
    t = xu_mul(t, k);
  }
  pt_store(work + i0, xu_add(u, t));
  pt_store(work + i1, xu_add(u, xu_neg(t)));
}

// (optional) scale by the canonical scalar `c` (ifft: m^-1, domain.rs:163-173), then normalise to affine
__global__ void __launch_bounds__(256) pfft_store_kernel(const PtU* __restrict__ work, G1Affine* __restrict__ out, uint32_t log_n, int scale,
                                                        Fr c_canon) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  XU v = pt_load(work + i);
  if (scale) v = xu_mul(v, c_canon.l);
  G1XYZZ sres = G1XYZZ::zero();
  if (!v.is_zero()) {
    const FqU c256 = UPow2<FqParams, 256>::get();   // v*2^261 * 2^256 / 2^261 = v * 2^256 (memory format)
    sres.x = u_to_std_lt2p(u_mul(v.x, c256));
    sres.y = u_to_std_lt2p(u_mul(v.y, c256));
    sres.zz = u_to_std_lt2p(u_mul(v.zz, c256));
    sres.zzz = u_to_std_lt2p(u_mul(v.zzz, c256));
  }
  out[i] = xyzz_to_affine(sres);
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

// d_points: 2^log_n affine raw records, in place.  inverse != 0: omega = omegainv and every output is scaled by minv.
int point_fft_g1(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st) {
  const uint64_t n = 1ull << log_n;
  PtU* work = nullptr;
  uint32_t* tw = nullptr;
  ZK_HIP(hipMalloc(&work, n * sizeof(PtU)));
  hipError_t e = hipMalloc(&tw, (n / 2 + 1) * 32);
  if (e != hipSuccess) { (void)hipFree(work); ZK_HIP(e); }
  auto fail = [&](hipError_t err) { (void)hipFree(work); (void)hipFree(tw); return err; };
  if (n >= 2) hipLaunchKernelGGL(pfft_twiddle_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, tw, omega, n / 2);
  hipLaunchKernelGGL(pfft_load_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const G1Affine*)d_points, work, log_n);
  for (uint32_t s = 0; s < log_n; ++s)
    hipLaunchKernelGGL(pfft_stage_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, work, tw, log_n, s);
  hipLaunchKernelGGL(pfft_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, (G1Affine*)d_points, log_n, scale ? 1 : 0,
                     scale_canon);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  ZK_HIP(fail(e));
  return ZK_OK;
}

}  // namespace zk
