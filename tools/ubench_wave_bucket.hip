// Bucket accumulation two ways, on the kernels' own arithmetic (curveu.hpp), for the design note in DESIGN.md section 4:
//   (A) ONE LANE PER BUCKET -- what msm_accumulate_kernel does: a lane walks its bucket's points with the mixed addition
//       (8 products + 2 squarings per point), accumulator in VGPRs;
//   (B) ONE WAVE PER BUCKET, "ballot / shuffle reduction of the bucket sum" -- what the north-star sketch prescribes: the
//       lanes of a wave take one point each (padding lanes hold infinity), and the 64 partial sums are folded with a
//       log2(64)-level shuffle tree of FULL additions (12 products + 2 squarings each, every level on all 64 lanes);
//   (B2) the same with the wave split into sub-groups of 32 / 16 lanes (one bucket per sub-group, 5 / 4 levels): the best case
//       for the shuffle scheme when buckets hold ~26 points, as they do at 2^26 points with 12 windows.
// Points are random field elements (the arithmetic does not care; the exceptional branches never trigger), buckets are
// contiguous runs of LEN records, so (A)'s lanes gather 64-byte records LEN * 64 B apart as in the real kernel.
// Output: bucket-points accumulated per second for the whole device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../phase2-bn254_amd/csrc/curveu.hpp"
using namespace zk;
using Acc = XYZZU<FqParams>;

__device__ __forceinline__ G1Affine ld(const G1Affine* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  G1Affine r;
  r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w; r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
  r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w; r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
  return r;
}

// (A)
__global__ void __launch_bounds__(256) k_lane(const G1Affine* __restrict__ pts, uint32_t n_buckets, uint32_t len, G1XYZZ* __restrict__ out) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_buckets) return;
  Acc acc = Acc::zero();
  const G1Affine* p = pts + (uint64_t)b * len;
  G1Affine cur = ld(p);
  for (uint32_t k = 0; k < len; ++k) {
    G1Affine nxt = cur;
    if (k + 1 < len) nxt = ld(p + k + 1);
    xyzzu_add_mixed(acc, cur.x, cur.y, false);
    cur = nxt;
  }
  out[b] = xyzzu_to_r(acc);
}

// (B), (B2): GROUP lanes per bucket (64, 32 or 16); len <= GROUP
__device__ __forceinline__ FpU<FqParams> shfl_down(const FpU<FqParams>& v, int d) {
  FpU<FqParams> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = __shfl_down(v.l[i], d, 64);
  return r;
}
template <int GROUP>
__global__ void __launch_bounds__(256) k_wave(const G1Affine* __restrict__ pts, uint32_t n_buckets, uint32_t len, G1XYZZ* __restrict__ out) {
  const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP;   // bucket
  const uint32_t lane = threadIdx.x % GROUP;
  if (gid >= n_buckets) return;  // (whole groups: n_buckets * GROUP is a multiple of the block size)
  Acc acc = Acc::zero();
  if (lane < len) {
    G1Affine p = ld(pts + (uint64_t)gid * len + lane);
    xyzzu_add_mixed(acc, p.x, p.y, false);   // the first point of an accumulator: two products by a constant
  }
  // accumulator domains (ZZ, ZZZ: 2^266) -> register form of the R domain, so that full additions close
  {
    const FpU<FqParams> c256 = UPow2<FqParams, 256>::get();
    if (!acc.is_zero()) { acc.zz = u_mul(acc.zz, c256); acc.zzz = u_mul(acc.zzz, c256); }
  }
#pragma unroll
  for (int d = GROUP / 2; d >= 1; d >>= 1) {
    Acc o{shfl_down(acc.x, d), shfl_down(acc.y, d), shfl_down(acc.zz, d), shfl_down(acc.zzz, d)};
    if (lane + d >= (uint32_t)GROUP) o = Acc::zero();
    xyzzr_add(acc, o);
  }
  if (lane == 0) out[gid] = xyzzr_store(acc);
}

int main() {
  const uint32_t len = 26;                 // mean bucket population at 2^26 points, 12 windows of 5 * 2^19 / 2 buckets
  const uint32_t n_buckets = 1u << 20;
  const uint64_t n_pts = (uint64_t)n_buckets * len;
  std::vector<uint32_t> h(n_pts * 16);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
  for (uint64_t i = 0; i < n_pts; ++i) { h[i * 16 + 7] &= 0x0fffffffu; h[i * 16 + 15] &= 0x0fffffffu; h[i * 16 + 8] |= 1u; }   // < p, y != 0
  G1Affine* d_pts = nullptr;
  G1XYZZ* d_out = nullptr;
  hipMalloc(&d_pts, n_pts * 64);
  hipMalloc(&d_out, (size_t)n_buckets * sizeof(G1XYZZ));
  hipMemcpy(d_pts, h.data(), n_pts * 64, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char* name, uint64_t pts_done, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::printf("%-58s %8.3f ms  %8.2f G bucket-points/s\n", name, ms, (double)pts_done / ms * 1e-6);
  };
  time("(A)  one lane per bucket (mixed additions in VGPRs)", n_pts, [&] { hipLaunchKernelGGL(k_lane, dim3(n_buckets / 256), dim3(256), 0, 0, d_pts, n_buckets, len, d_out); });
  time("(B)  one wave per bucket (64-lane shuffle tree, 6 levels)", n_pts, [&] { hipLaunchKernelGGL(k_wave<64>, dim3(n_buckets / 4), dim3(256), 0, 0, d_pts, n_buckets, len, d_out); });
  time("(B2) 32 lanes per bucket (5 levels)", n_pts, [&] { hipLaunchKernelGGL(k_wave<32>, dim3(n_buckets / 8), dim3(256), 0, 0, d_pts, n_buckets, len, d_out); });
  const uint32_t len16 = 16;
  time("(A)  one lane per bucket, 16 points per bucket", (uint64_t)n_buckets * len16, [&] { hipLaunchKernelGGL(k_lane, dim3(n_buckets / 256), dim3(256), 0, 0, d_pts, n_buckets, len16, d_out); });
  time("(B2) 16 lanes per bucket, 16 points per bucket (4 levels)", (uint64_t)n_buckets * len16, [&] { hipLaunchKernelGGL(k_wave<16>, dim3(n_buckets / 16), dim3(256), 0, 0, d_pts, n_buckets, len16, d_out); });
  hipFree(d_pts); hipFree(d_out);
  return 0;
}
