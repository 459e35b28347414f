// Field-product throughput of the two Fq representations on gfx950: the memory format (8 x 32-bit limbs, field.hpp: column-wise
// v_mad_u64_u32 + v_addc_co_u32 carry word, inline asm) against the U-form (9 x 29-bit lazy limbs, fieldu.hpp: one v_mad_u64_u32
// per partial product, no carry flags).  Every lane runs CHAINS independent chains x = x * y of ITERS products; 1024 SIMDs are
// filled at 1, 2 and 4 waves per SIMD.  Output: products per second for the whole device and cycles per wave-product per SIMD.
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
  hipFree(d);
  return 0;
}
