// Instruction-throughput microbenchmark for the integer / fp64 VALU ops that a
// 256-bit Montgomery multiplier can be built from on gfx950 (MI355X).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Prints cycles per wave-instruction per SIMD for each op (assuming 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int NCHAIN = 8;  // independent dependency chains per lane

struct OpMad64 { using T = uint64_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc"); } };
struct OpMulLo { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a)); } };
struct OpMulHi { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a)); } };
struct OpAdd { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a)); } };
struct OpAddCo { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(a) : "vcc"); } };
struct OpAddcCo { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc"); } };
struct OpLshlAdd64 { using T = uint64_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { uint64_t aa = ((uint64_t)a << 32) | b; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(aa)); } };
struct OpMad24 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); } };
struct OpMulHi24 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a)); } };
struct OpFma64 { using T = double; static __device__ void op(T& x, uint32_t a, uint32_t b) { double aa = 1.0 + 1e-9 * a, bb = 1e-12 * b; asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(aa), "v"(bb)); } };
struct OpFma32 { using T = float; static __device__ void op(T& x, uint32_t a, uint32_t b) { float aa = 1.0f + 1e-7f * a, bb = 1e-9f * b; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(aa), "v"(bb)); } };
struct OpDot4 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); } };
struct OpDot2 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); } };
struct OpMadU16 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); } };
struct OpPkMadU16 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); } };
struct OpLshr64 { using T = uint64_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x)); } };
struct OpAlignbit { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(x) : "v"(a)); } };
struct OpAnd { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a)); } };
struct OpLshr32 { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x)); } };
struct OpCndmask { using T = uint32_t; static __device__ void op(T& x, uint32_t a, uint32_t b) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc"); } };

template <class Op>
__global__ void __launch_bounds__(256) kern(uint32_t* out, uint32_t seed) {
  typename Op::T acc[NCHAIN];
  uint32_t a = (seed | 1u) + threadIdx.x, b = seed * 3u + 7u;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) acc[c] = (typename Op::T)(seed + c);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c) Op::op(acc[c], a, b);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) s += (uint32_t)acc[c];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <class Op>
static int run(const char* name, uint32_t* d_out) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int bpc : {1, 2, 4}) {
    int grid = cus * bpc;
    kern<Op><<<grid, 256>>>(d_out, 12345u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0));
      kern<Op><<<grid, 256>>>(d_out, 12345u + r);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    double wave_instr = (double)grid * 4 * ITERS * NCHAIN;
    double per_simd_per_s = wave_instr / (cus * 4.0) / (best * 1e-3);
    printf("%-18s waves/SIMD=%d  %8.3f ms  %7.2f cyc/wave-instr/SIMD(@2.4GHz)  %9.1f Glane-ops/s\n", name, bpc, best,
           2.4e9 / per_simd_per_s, wave_instr * 64 / (best * 1e-3) / 1e9);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint32_t* d_out; CK(hipMalloc(&d_out, 4096));
  run<OpMad64>("v_mad_u64_u32", d_out); run<OpMulLo>("v_mul_lo_u32", d_out); run<OpMulHi>("v_mul_hi_u32", d_out);
  run<OpAdd>("v_add_u32", d_out); run<OpAddCo>("v_add_co_u32", d_out); run<OpAddcCo>("v_addc_co_u32", d_out);
  run<OpLshlAdd64>("v_lshl_add_u64", d_out); run<OpLshr64>("v_lshrrev_b64", d_out); run<OpAlignbit>("v_alignbit_b32", d_out); run<OpAnd>("v_and_b32", d_out); run<OpLshr32>("v_lshrrev_b32", d_out); run<OpMad24>("v_mad_u32_u24", d_out); run<OpMulHi24>("v_mul_hi_u32_u24", d_out);
  run<OpFma64>("v_fma_f64", d_out); run<OpFma32>("v_fma_f32", d_out); run<OpDot4>("v_dot4_u32_u8", d_out);
  run<OpDot2>("v_dot2_u32_u16", d_out); run<OpMadU16>("v_mad_u32_u16", d_out); run<OpPkMadU16>("v_pk_mad_u16", d_out);
  run<OpCndmask>("v_cndmask_b32", d_out);
  return 0;
}
