// FETCH_SIZE calibration for the access patterns of msm_accumulate_kernel on gfx950 (MI355X):
// what does rocprofv3's FETCH_SIZE report for a random gather of 64-byte records (four back-to-back 16-byte loads per
// lane, the base gather), of single 16-byte words (the index-list loads), of 128-byte records, and for a plain coalesced
// stream of the same table?  Each kernel prints the bytes it logically reads; run under
//   rocprofv3 --pmc FETCH_SIZE -d <dir> -- tools/bin/ubench_gather
// and compare per kernel (tools/rocpd_summary.py <db> --pmc).  The table (4 GiB) is far larger than L2 + Infinity Cache.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o tools/bin/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// REC bytes per record (16, 64 or 128), K gathers per lane
template <int REC, int K>
__global__ void __launch_bounds__(256) gather_kernel(const uint4* __restrict__ table, uint32_t n_rec_mask, uint32_t* __restrict__ sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const uint32_t r = mix(gid * 0x9e3779b9u + k * 0x85ebca6bu + 12345u) & n_rec_mask;
    const uint4* p = table + (size_t)r * (REC / 16);
    uint4 v[REC / 16];
#pragma unroll
    for (int q = 0; q < REC / 16; ++q) v[q] = p[q];
#pragma unroll
    for (int q = 0; q < REC / 16; ++q) acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
  }
  if (acc == 0x12345678u) sink[gid & 1023] = acc;
}

__global__ void __launch_bounds__(256) stream_kernel(const uint4* __restrict__ table, size_t n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = table[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int REC, int K>
int run(const char* name, const uint4* table, size_t table_bytes, uint32_t* sink) {
  const uint32_t lanes = 1u << 24;
  const uint32_t mask = (uint32_t)(table_bytes / REC) - 1u;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((gather_kernel<REC, K>), dim3(lanes / 256), dim3(256), 0, 0, table, mask, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)lanes * K * REC;
  printf("%-28s %d gathers/lane of %3d B: logical %.3f GB in %.3f ms = %.1f GB/s  (gather_kernel<%d, %d>)\n", name, K, REC, bytes / 1e9, ms, bytes / ms / 1e6, REC, K);
  return 0;
}

int main() {
  const size_t table_bytes = (size_t)4 << 30;
  uint4* table = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&table, table_bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(table, 1, table_bytes));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, table, table_bytes / 16, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s logical %.3f GB in %.3f ms = %.1f GB/s  (stream_kernel)\n", "coalesced 16 B/lane stream", table_bytes / 1e9, ms, table_bytes / ms / 1e6);
  if (run<64, 8>("random 64-B records", table, table_bytes, sink)) return 1;
  if (run<16, 8>("random 16-B words", table, table_bytes, sink)) return 1;
  if (run<128, 8>("random 128-B records", table, table_bytes, sink)) return 1;
  CK(hipFree(table));
  CK(hipFree(sink));
  return 0;
}
