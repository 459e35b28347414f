// Random 64-byte record gathers (four 16-byte loads per lane, the base gather of msm_accumulate_kernel) against the FOOTPRINT of the table:
// how much of the MSM's sensitivity to gather locality (profiles/r05_compact_pairs_ab.txt) is the Infinity Cache (256 MiB) / TLB reach, and what the
// device delivers for random records at the 4 GiB of a 2^26-point base vector.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather_footprint.hip -o tools/bin/ubench_gather_footprint
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int K>
__global__ void __launch_bounds__(256) gather64(const uint4* __restrict__ table, uint32_t mask, uint32_t salt, uint32_t* __restrict__ sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const uint32_t r = mix(gid * 0x9e3779b9u + k * 0x85ebca6bu + salt) & mask;
    const uint4* p = table + (size_t)r * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[gid & 1023] = acc;
}
int main(int argc, char** argv) {
  const size_t max_bytes = 16ull << 30;
  uint4* table = nullptr; uint32_t* sink = nullptr;
  const bool contiguous = argc > 1 && argv[1][0] == 'c';   // `ubench_gather_footprint c`: hipExtMallocWithFlags(hipDeviceMallocContiguous) -- physically contiguous VRAM, large PTE fragments
  if (contiguous) { CK(hipExtMallocWithFlags((void**)&table, max_bytes, hipDeviceMallocContiguous)); printf("(physically contiguous allocation)\n"); }
  else CK(hipMalloc(&table, max_bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(table, 1, max_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t lanes = 1u << 24; constexpr int K = 16;
  for (size_t mb : {64ull, 256ull, 1024ull, 4096ull, 16384ull}) {
    const uint32_t mask = (uint32_t)((mb << 20) / 64) - 1u;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((gather64<K>), dim3(lanes / 256), dim3(256), 0, 0, table, mask, 7u + rep, sink);
    CK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL((gather64<K>), dim3(lanes / 256), dim3(256), 0, 0, table, mask, 100u + rep, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = 4.0 * lanes * K;
    printf("random 64-B records over %6zu MiB: %.2f G records/s = %.0f GB/s\n", mb, recs / ms / 1e6, recs * 64 / ms / 1e6);
  }
  return 0;
}
