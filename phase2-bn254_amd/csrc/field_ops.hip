// Elementwise Fr kernels of the path's callers and the field-multiplier microbenchmark.
//   EvaluationDomain::mul_assign  bellman/src/domain.rs:236-249  (pointwise product, prover.rs:221-236)
//   EvaluationDomain::sub_assign  bellman/src/domain.rs:251-260
// plus mi355zk_ubench_fp_mul: the measured Montgomery-product rate of this library on the device -- the
// integer-ALU roofline the MSM / NTT kernels are priced against.
#include <hip/hip_runtime.h>

#include <vector>

#include <cstring>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "glv.hpp"
#include "device_util.hpp"

namespace zk {
namespace {

__device__ __forceinline__ Fr ld(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void st(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// op 0: a *= b   op 1: a -= b
__global__ void __launch_bounds__(256) fr_pointwise_kernel(Fr* __restrict__ a, const Fr* __restrict__ b, uint64_t n, int op) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    Fr x = ld(a + i), y = ld(b + i);
    st(a + i, op == 0 ? mul(x, y) : sub(x, y));
  }
}

// out[i] = into_repr(in[i]): Montgomery form -> canonical integer (one Montgomery reduction; out may alias in)
__global__ void __launch_bounds__(256) fr_into_repr_kernel(Fr* out, const Fr* in, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    st(out + i, to_canonical(ld(in + i)));
}

// every lane runs `iters` dependent products x <- x * y on 4 independent chains (ILP like the group law)
template <class PR>
__global__ void __launch_bounds__(256) fp_mul_ubench_kernel(Fp<PR> a, Fp<PR> b, uint32_t iters, Fp<PR>* out) {
  Fp<PR> x0 = a, x1 = b, x2 = add(a, b), x3 = sub(a, b);
  x0.l[0] ^= 0;  // keep lanes identical: the result of chain 0 is checked against the oracle
  for (uint32_t i = 0; i < iters; ++i) {
    x0 = mul(x0, b);
    x1 = mul(x1, a);
    x2 = mul(x2, b);
    x3 = mul(x3, a);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = x0;
    out[1] = x1;
    out[2] = x2;
    out[3] = x3;
  }
  // defeat dead-code elimination for the other lanes without memory traffic
  if ((x1.l[0] ^ x2.l[1] ^ x3.l[2]) == 0x9e3779b9u && x0.l[7] == 0xffffffffu) out[4 + (blockIdx.x & 3)] = x1;
}

template <class PR>
int ubench(uint32_t blocks, uint32_t iters, const uint64_t* a_raw, const uint64_t* b_raw, uint64_t* out_raw, float* ms) {
  Fp<PR> a, b;
  std::memcpy(&a, a_raw, 32);
  std::memcpy(&b, b_raw, 32);
  Fp<PR>* d_out = nullptr;
  ZK_HIP(hipMalloc(&d_out, 8 * sizeof(Fp<PR>)));
  hipEvent_t e0, e1;
  ZK_HIP(hipEventCreate(&e0));
  ZK_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(fp_mul_ubench_kernel<PR>, dim3(blocks), dim3(256), 0, 0, a, b, iters, d_out);  // warm-up
  ZK_HIP(hipDeviceSynchronize());
  ZK_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(fp_mul_ubench_kernel<PR>, dim3(blocks), dim3(256), 0, 0, a, b, iters, d_out);
  ZK_HIP(hipEventRecord(e1, 0));
  ZK_HIP(hipEventSynchronize(e1));
  ZK_HIP(hipEventElapsedTime(ms, e0, e1));
  ZK_HIP(hipMemcpy(out_raw, d_out, 4 * sizeof(Fp<PR>), hipMemcpyDeviceToHost));
  ZK_HIP(hipFree(d_out));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ZK_OK;
}

int pointwise(void* d_a, const void* d_b, size_t n, void* stream, int op) {
  if ((!d_a || !d_b) && n) return ZK_ERR_BAD_ARGS;
  if (n == 0) return ZK_OK;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(fr_pointwise_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (Fr*)d_a, (const Fr*)d_b, (uint64_t)n, op);
  ZK_HIP(hipGetLastError());
  return ZK_OK;
}

}  // namespace
}  // namespace zk

// ---- host-side self-test hooks for the U-form arithmetic (same source as the kernels, compiled for the host)
template <class PR>
static void selftest_u_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  zk::FpU<PR> x, y;
  std::memcpy(&x, a, 36);
  std::memcpy(&y, b, 36);
  zk::FpU<PR> r = zk::u_mul(x, y);
  std::memcpy(out, &r, 36);
}
template <class PR>
static int selftest_u_sub(int k, int s, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  zk::FpU<PR> x, y, r;
  std::memcpy(&x, a, 36);
  std::memcpy(&y, b, 36);
  if (k == 1 && s == 1) r = zk::u_sub<1, 1>(x, y);
  else if (k == 2 && s == 1) r = zk::u_sub<2, 1>(x, y);
  else if (k == 4 && s == 1) r = zk::u_sub<4, 1>(x, y);
  else if (k == 4 && s == 2) r = zk::u_sub<4, 2>(x, y);
  else if (k == 4 && s == 3) r = zk::u_sub<4, 3>(x, y);
  else if (k == 8 && s == 1) r = zk::u_sub<8, 1>(x, y);
  else return ZK_ERR_BAD_ARGS;
  std::memcpy(out, &r, 36);
  return ZK_OK;
}

template <class PR>
static void selftest_u_mul_shoup(const uint32_t* a, const uint32_t* w_plain, uint32_t* out, uint32_t* out_wq) {
  zk::FpU<PR> x;
  std::memcpy(&x, a, 36);
  zk::Fp<PR> c;
  std::memcpy(&c, w_plain, 32);
  const zk::FpU<PR> w = zk::u_from_std(c), wq = zk::u_shoup_quotient<PR>(c.l);
  const zk::FpU<PR> r = zk::u_mul_shoup(x, w, wq);
  std::memcpy(out, &r, 36);
  std::memcpy(out_wq, &wq, 36);
}

extern "C" {

// the product by a table constant (fieldu.hpp u_mul_shoup): a on 9 u32 limbs, w_plain the canonical integer w < p on 8 x 32-bit words;
// out = a * w - q * p (N-form, < 2p for a < 160p), out_wq = floor(w * 2^261 / p) on 9 limbs
int mi355zk_selftest_u_mul_shoup(int which, const uint32_t a[9], const uint32_t w_plain[8], uint32_t out[9], uint32_t out_wq[9]) {
  return zk::abi_guard([&]() -> int {
    if (!a || !w_plain || !out || !out_wq) return ZK_ERR_BAD_ARGS;
    if (which == 0) selftest_u_mul_shoup<zk::FqParams>(a, w_plain, out, out_wq);
    else selftest_u_mul_shoup<zk::FrParams>(a, w_plain, out, out_wq);
    return ZK_OK;
  });
}

// a, b, out: 9 u32 limbs (radix 2^29).  which: 0 Fq, 1 Fr.  out = a*b*2^-261 mod p (N-form, lazily reduced)
int mi355zk_selftest_u_mul(int which, const uint32_t a[9], const uint32_t b[9], uint32_t out[9]) {
  return zk::abi_guard([&]() -> int {
    if (!a || !b || !out) return ZK_ERR_BAD_ARGS;
    if (which == 0) selftest_u_mul<zk::FqParams>(a, b, out);
    else selftest_u_mul<zk::FrParams>(a, b, out);
    return ZK_OK;
  });
}
// out = carry(a + k*p - b) for the (k, s) pairs the kernels use
int mi355zk_selftest_u_sub(int which, int k, int s, const uint32_t a[9], const uint32_t b[9], uint32_t out[9]) {
  return zk::abi_guard([&]() -> int {
    if (!a || !b || !out) return ZK_ERR_BAD_ARGS;
    return which == 0 ? selftest_u_sub<zk::FqParams>(k, s, a, b, out) : selftest_u_sub<zk::FrParams>(k, s, a, b, out);
  });
}
// memory format <-> U limbs round trip pieces: out_u = u_from_std(a);  out_std = u_to_std_lt2p(in_u) (needs in_u < 2p, N-form)
int mi355zk_selftest_u_pack(int which, const uint64_t a_std[4], uint32_t out_u[9], const uint32_t in_u[9], uint64_t out_std[4]) {
  return zk::abi_guard([&]() -> int {
    if (which == 0) {
      if (a_std && out_u) { zk::Fq x; std::memcpy(&x, a_std, 32); zk::FqU u = zk::u_from_std(x); std::memcpy(out_u, &u, 36); }
      if (in_u && out_std) { zk::FqU u; std::memcpy(&u, in_u, 36); zk::Fq x = zk::u_to_std_lt2p(u); std::memcpy(out_std, &x, 32); }
    } else {
      if (a_std && out_u) { zk::Fr x; std::memcpy(&x, a_std, 32); zk::FrU u = zk::u_from_std(x); std::memcpy(out_u, &u, 36); }
      if (in_u && out_std) { zk::FrU u; std::memcpy(&u, in_u, 36); zk::Fr x = zk::u_to_std_lt2p(u); std::memcpy(out_std, &x, 32); }
    }
    return ZK_OK;
  });
}
// out_std = u_to_std_lt32p(in_u): canonical reduction of an N-form value < 32p without a product (the NTT's closing step)
int mi355zk_selftest_u_reduce32(int which, const uint32_t in_u[9], uint64_t out_std[4]) {
  return zk::abi_guard([&]() -> int {
    if (!in_u || !out_std) return ZK_ERR_BAD_ARGS;
    if (which == 0) { zk::FqU u; std::memcpy(&u, in_u, 36); zk::Fq x = zk::u_to_std_lt32p(u); std::memcpy(out_std, &x, 32); }
    else { zk::FrU u; std::memcpy(&u, in_u, 36); zk::Fr x = zk::u_to_std_lt32p(u); std::memcpy(out_std, &x, 32); }
    return ZK_OK;
  });
}
// bucket accumulation of n signed affine G1 points on the HOST: mode 0 = saturated-limb XYZZ (curve.hpp),
// mode 1 = U-form XYZZ (curveu.hpp).  out = memory-format XYZZ (X, Y, ZZ, ZZZ; 16 u64).
int mi355zk_selftest_g1_accumulate(int mode, const uint64_t* affine_pts, const uint8_t* negate, size_t n, uint64_t out_xyzz[16]) {
  return zk::abi_guard([&]() -> int {
    if ((!affine_pts || !negate) && n) return ZK_ERR_BAD_ARGS;
    if (!out_xyzz) return ZK_ERR_BAD_ARGS;
    zk::G1XYZZ r;
    if (mode == 3) {
      // the PAIR-per-bucket addition (curveu.hpp: pair_add_mixed(PairAcc1, ..), msm_accumulate_pair_g1_kernel) replayed on the host as
      // its two lanes E = (X, ZZ), O = (Y, ZZZ): the same rounds, subtraction constants, exchanges and rare branches, on the same
      // host + device primitives (the device version differs by the DPP moves and the selects that pick a lane's operands)
      using namespace zk;
      const FqU zero = FqU::zero(), C = UPow2<FqParams, 266>::get();
      FqU Ea = zero, Ez = zero, Oa = zero, Oz = zero;
      for (size_t i = 0; i < n; ++i) {
        G1Affine p;
        std::memcpy(&p, affine_pts + 8 * i, 64);
        const FqU x2 = u_from_std(p.x);
        FqU y2 = u_from_std(p.y);
        if (negate[i]) y2 = u_sub<1, 1>(zero, y2);
        if (Ez.limbs_all_zero()) {
          if (!Oz.limbs_all_zero()) return ZK_ERR_BAD_ARGS;   // (the lanes agree on infinity)
          Ea = u_mul(x2, C); Oa = u_mul(y2, C); Ez = C; Oz = C;
          continue;
        }
        const FqU P = u_sub<8, 1>(u_mul(x2, Ez), Ea), R = u_sub<2, 1>(u_mul(y2, Oz), Oa);   // round 1
        const FqU PP = u_sqr(P), RR = u_sqr(R);                                              // round 2
        const FqU Q = u_mul(Ea, PP), PPP = u_mul(P, PP);                                     // round 3
        const FqU ZZ3 = u_mul(Ez, PP), ZZZ3 = u_mul(Oz, PPP);                                // round 4
        const FqU X3 = u_sub<4, 3>(RR, u_add(PPP, u_dbl(Q)));
        const FqU D = u_sub<8, 1>(Q, X3);
        const FqU Y3 = u_mul2(R, D, u_sub<2, 1>(zero, Oa), PPP);                             // round 5
        if (u_is_zero_lt2p(ZZ3) != u_is_zero_lt2p(ZZZ3)) return ZK_ERR_BAD_ARGS;            // (the lanes agree on P == 0)
        if (u_is_zero_lt2p(ZZ3)) {
          if (u_is_zero_lt8p(R)) {
            const XYZZU<FqParams> dbl = xyzzu_double_affine(x2, y2);
            Ea = dbl.x; Oa = dbl.y; Ez = dbl.zz; Oz = dbl.zzz;
          } else {
            Ea = Oa = Ez = Oz = zero;
          }
          continue;
        }
        Ea = X3; Oa = Y3; Ez = ZZ3; Oz = ZZZ3;
      }
      r = xyzzu_to_std(XYZZU<FqParams>{Ea, Oa, Ez, Oz});
    } else if (mode == 0) {
      zk::G1XYZZ acc = zk::G1XYZZ::zero();
      for (size_t i = 0; i < n; ++i) {
        zk::G1Affine p;
        std::memcpy(&p, affine_pts + 8 * i, 64);
        zk::xyzz_add_mixed(acc, p.x, p.y, negate[i] != 0);
      }
      r = acc;
    } else {
      // mode 2: the accumulator goes through an R-domain record after every third point (xyzzu_to_r, then xyzzu_from_r): what a
      // bucket carried across the chunks of a streamed multiexp does
      zk::XYZZU<zk::FqParams> acc = zk::XYZZU<zk::FqParams>::zero();
      for (size_t i = 0; i < n; ++i) {
        zk::G1Affine p;
        std::memcpy(&p, affine_pts + 8 * i, 64);
        zk::xyzzu_add_mixed(acc, p.x, p.y, negate[i] != 0);
        if (mode == 2 && i % 3 == 2) acc = zk::xyzzu_from_r(zk::xyzzu_to_r(acc));
      }
      r = mode == 2 ? zk::xyzzr_to_std(zk::xyzzu_to_r(acc)) : zk::xyzzu_to_std(acc);
    }
    std::memcpy(out_xyzz, &r, sizeof r);
    return ZK_OK;
  });
}

// The R-domain records of the G1 bucket reduction (curveu.hpp) on the HOST: n signed affine points are accumulated into n_groups
// buckets (group[i] < n_groups) by the U-form mixed addition, every bucket becomes a record (xyzzu_to_r), and the records are summed
// in group order -- mode 0: one running sum kept in registers (xyzzr_add), mode 1: through a record after every addition
// (xyzzr_load / xyzzr_store, what the LDS trees do).  out = memory-format XYZZ of the total (xyzzr_to_std).
int mi355zk_selftest_g1_record_sum(int mode, const uint64_t* affine_pts, const uint8_t* negate, const uint32_t* group, size_t n, size_t n_groups,
                                   uint64_t out_xyzz[16]) {
  return zk::abi_guard([&]() -> int {
    if ((!affine_pts || !negate || !group) && n) return ZK_ERR_BAD_ARGS;
    if (!out_xyzz || n_groups == 0) return ZK_ERR_BAD_ARGS;
    std::vector<zk::XYZZU<zk::FqParams>> acc(n_groups, zk::XYZZU<zk::FqParams>::zero());
    for (size_t i = 0; i < n; ++i) {
      if (group[i] >= n_groups) return ZK_ERR_BAD_ARGS;
      zk::G1Affine p;
      std::memcpy(&p, affine_pts + 8 * i, 64);
      zk::xyzzu_add_mixed(acc[group[i]], p.x, p.y, negate[i] != 0);
    }
    zk::G1XYZZ total = zk::G1XYZZ::zero();
    zk::XYZZU<zk::FqParams> run = zk::XYZZU<zk::FqParams>::zero();
    for (size_t g = 0; g < n_groups; ++g) {
      const zk::G1XYZZ rec = zk::xyzzu_to_r(acc[g]);
      if (mode == 0) {
        zk::xyzzr_add(run, zk::xyzzr_load(rec));
      } else {
        zk::XYZZU<zk::FqParams> t = zk::xyzzr_load(total);
        zk::xyzzr_add(t, zk::xyzzr_load(rec));
        total = zk::xyzzr_store(t);
      }
    }
    if (mode == 0) total = zk::xyzzr_store(run);
    const zk::G1XYZZ r = zk::xyzzr_to_std(total);
    std::memcpy(out_xyzz, &r, sizeof r);
    return ZK_OK;
  });
}

// the same for G2 (16 u64 per affine point; out = memory-format XYZZ over Fq2: 32 u64)
int mi355zk_selftest_g2_record_sum(int mode, const uint64_t* affine_pts, const uint8_t* negate, const uint32_t* group, size_t n, size_t n_groups,
                                   uint64_t out_xyzz[32]) {
  return zk::abi_guard([&]() -> int {
    if ((!affine_pts || !negate || !group) && n) return ZK_ERR_BAD_ARGS;
    if (!out_xyzz || n_groups == 0) return ZK_ERR_BAD_ARGS;
    std::vector<zk::XYZZU2> acc(n_groups, zk::XYZZU2::zero());
    for (size_t i = 0; i < n; ++i) {
      if (group[i] >= n_groups) return ZK_ERR_BAD_ARGS;
      zk::G2Affine p;
      std::memcpy(&p, affine_pts + 16 * i, 128);
      zk::xyzzu2_add_mixed(acc[group[i]], p.x, p.y, negate[i] != 0);
    }
    zk::G2XYZZ total = zk::G2XYZZ::zero();
    zk::XYZZU2 run = zk::XYZZU2::zero();
    for (size_t g = 0; g < n_groups; ++g) {
      const zk::G2XYZZ rec = zk::xyzzu_to_r(acc[g]);
      if (mode == 0) {
        zk::xyzzr_add(run, zk::xyzzr_load(rec));
      } else {
        zk::XYZZU2 t = zk::xyzzr_load(total);
        zk::xyzzr_add(t, zk::xyzzr_load(rec));
        total = zk::xyzzr_store(t);
      }
    }
    if (mode == 0) total = zk::xyzzr_store(run);
    const zk::G2XYZZ r = zk::xyzzr_to_std(total);
    std::memcpy(out_xyzz, &r, sizeof r);
    return ZK_OK;
  });
}

// GLV split of a canonical scalar (glv.hpp) on the HOST: out = k1 magnitude (5 u32), k2 magnitude (5 u32), sign of k1, sign of k2
int mi355zk_selftest_glv_split(const uint32_t k[8], uint32_t out[12]) {
  return zk::abi_guard([&]() -> int {
    if (!k || !out) return ZK_ERR_BAD_ARGS;
    const zk::GlvSplit g = zk::glv_split(k);
    for (int i = 0; i < 5; ++i) { out[i] = g.k1[i]; out[5 + i] = g.k2[i]; }
    out[10] = g.neg1 ? 1u : 0u;
    out[11] = g.neg2 ? 1u : 0u;
    return ZK_OK;
  });
}

// width-5 non-adjacent form (glv.hpp glv_wnaf5) of a magnitude on 5 u32 limbs, on the HOST: digits[164]; returns the top digit's index (-1: zero)
int mi355zk_selftest_glv_wnaf5(const uint32_t m[5], int8_t digits[164]) {
  return zk::abi_guard([&]() -> int {
    if (!m || !digits) return -2;
    static_assert(zk::GLV_WNAF_LEN == 164, "header comment");
    return zk::glv_wnaf5(m, digits);
  });
}

// the G2 split k = k1 + k2 mu (glv.hpp) on the HOST: out = k1 (5 u32), k2 (5 u32)
int mi355zk_selftest_glv2_split(const uint32_t k[8], uint32_t out[10]) {
  return zk::abi_guard([&]() -> int {
    if (!k || !out) return ZK_ERR_BAD_ARGS;
    const zk::Glv2Split g = zk::glv2_split(k);
    for (int i = 0; i < 5; ++i) { out[i] = g.k1[i]; out[5 + i] = g.k2[i]; }
    return ZK_OK;
  });
}
// psi of an affine G2 point through the table-entry path the kernels use (jacu2_tab_from_affine -> jacu2_tab_psi), Jacobian out
int mi355zk_selftest_g2_psi(const uint64_t affine_pt[16], uint64_t out_xyz[24]) {
  return zk::abi_guard([&]() -> int {
    if (!affine_pt || !out_xyz) return ZK_ERR_BAD_ARGS;
    zk::G2Affine p;
    std::memcpy(&p, affine_pt, sizeof p);
    const zk::FqU C266 = zk::UPow2<zk::FqParams, 266>::get();
    const zk::Fq2 cxs = zk::glv2_cx(), cys = zk::glv2_cy();
    const zk::Fq2U cxU{zk::u_mul(zk::u_from_std(cxs.c0), C266), zk::u_mul(zk::u_from_std(cxs.c1), C266)};
    const zk::Fq2U cyU{zk::u_mul(zk::u_from_std(cys.c0), C266), zk::u_mul(zk::u_from_std(cys.c1), C266)};
    const zk::JacTabU2 e = zk::jacu2_tab_psi(zk::jacu2_tab_from_affine(p.x, p.y), cxU, cyU);
    zk::JacU2 acc = zk::JacU2::zero();
    zk::jacu2_add_tab(acc, e, false);
    const zk::G2Jacobian r = zk::jacu2_to_std(acc);
    std::memcpy(out_xyz, &r, sizeof r);
    return ZK_OK;
  });
}

// same for G2 (16 u64 per affine point; out = X, Y, ZZ, ZZZ over Fq2: 32 u64)
int mi355zk_selftest_g2_accumulate(int mode, const uint64_t* affine_pts, const uint8_t* negate, size_t n, uint64_t out_xyzz[32]) {
  return zk::abi_guard([&]() -> int {
    if ((!affine_pts || !negate) && n) return ZK_ERR_BAD_ARGS;
    if (!out_xyzz) return ZK_ERR_BAD_ARGS;
    zk::G2XYZZ r;
    if (mode == 3) {
      // the PAIR-per-bucket addition over Fq2 (curveu.hpp: pair_add_mixed(PairAcc2, ..), msm_accumulate_pair_kernel) replayed on the host
      // as its two lanes E = (X, ZZ), O = (Y, ZZZ) -- see the G1 hook; the rare doubling runs xyzzu2_double_affine as on the device
      using namespace zk;
      const FqU zero = FqU::zero(), C = UPow2<FqParams, 266>::get();
      Fq2U Ea = Fq2U::zero(), Ez = Fq2U::zero(), Oa = Fq2U::zero(), Oz = Fq2U::zero();
      auto sqr = [&](const Fq2U& v, const FqU& diff) { return Fq2U{u_mul(u_carry(u_add(v.c0, v.c1)), diff), u_mul(u_dbl(v.c0), v.c1)}; };
      for (size_t i = 0; i < n; ++i) {
        G2Affine p;
        std::memcpy(&p, affine_pts + 16 * i, 128);
        const Fq2U x2 = f2u_from_std(p.x);
        Fq2U y2 = f2u_from_std(p.y);
        if (negate[i]) y2 = Fq2U{u_sub<1, 1>(zero, y2.c0), u_sub<1, 1>(zero, y2.c1)};
        if (Ez.limbs_all_zero()) {
          if (!Oz.limbs_all_zero()) return ZK_ERR_BAD_ARGS;
          Ea = Fq2U{u_mul(x2.c0, C), u_mul(x2.c1, C)};
          Oa = Fq2U{u_mul(y2.c0, C), u_mul(y2.c1, C)};
          Ez = Oz = Fq2U{C, zero};
          continue;
        }
        const Fq2U U2 = f2u_mul<2>(x2, Ez), S2 = f2u_mul<2>(y2, Oz);                          // round 1
        const Fq2U P{u_sub<8, 1>(U2.c0, Ea.c0), u_sub<8, 1>(U2.c1, Ea.c1)}, R{u_sub<2, 1>(S2.c0, Oa.c0), u_sub<2, 1>(S2.c1, Oa.c1)};
        const Fq2U PP = sqr(P, u_sub<10, 1>(P.c0, P.c1)), RR = sqr(R, u_sub<4, 1>(R.c0, R.c1));   // round 2
        const Fq2U Q = f2u_mul<4>(Ea, PP), PPP = f2u_mul<4>(P, PP);                           // round 3
        const Fq2U ZZ3 = f2u_mul<4>(Ez, PP), ZZZ3 = f2u_mul<4>(Oz, PPP);                      // round 4
        const Fq2U X3{u_sub<4, 3>(RR.c0, u_add(PPP.c0, u_dbl(Q.c0))), u_sub<4, 3>(RR.c1, u_add(PPP.c1, u_dbl(Q.c1)))};
        const Fq2U D = f2u_sub<8>(Q, X3);
        const FqU ny0 = u_sub<2, 1>(zero, Oa.c0), ny1 = u_sub<2, 1>(zero, Oa.c1), nd1 = u_sub<16, 1>(zero, D.c1);
        const Fq2U Y3{u_mul4(R.c0, D.c0, R.c1, nd1, ny0, PPP.c0, Oa.c1, PPP.c1), u_mul4(R.c0, D.c1, R.c1, D.c0, ny0, PPP.c1, ny1, PPP.c0)};   // round 5
        const bool ze = u_is_zero_lt2p(ZZ3.c0) && u_is_zero_lt2p(ZZ3.c1), zo = u_is_zero_lt2p(ZZZ3.c0) && u_is_zero_lt2p(ZZZ3.c1);
        if (ze != zo) return ZK_ERR_BAD_ARGS;
        if (ze) {
          if (u_is_zero_lt8p(R.c0) && u_is_zero_lt8p(R.c1)) {
            const XYZZU2 dbl = xyzzu2_double_affine(x2, y2);
            Ea = dbl.x; Oa = dbl.y; Ez = dbl.zz; Oz = dbl.zzz;
          } else {
            Ea = Oa = Ez = Oz = Fq2U::zero();
          }
          continue;
        }
        Ea = X3; Oa = Y3; Ez = ZZ3; Oz = ZZZ3;
      }
      r = xyzzu2_to_std(XYZZU2{Ea, Oa, Ez, Oz});
    } else if (mode == 0) {
      zk::G2XYZZ acc = zk::G2XYZZ::zero();
      for (size_t i = 0; i < n; ++i) {
        zk::G2Affine p;
        std::memcpy(&p, affine_pts + 16 * i, 128);
        zk::xyzz_add_mixed(acc, p.x, p.y, negate[i] != 0);
      }
      r = acc;
    } else {
      zk::XYZZU2 acc = zk::XYZZU2::zero();  // mode 2: through a record after every third point (see the G1 hook)
      for (size_t i = 0; i < n; ++i) {
        zk::G2Affine p;
        std::memcpy(&p, affine_pts + 16 * i, 128);
        zk::xyzzu2_add_mixed(acc, p.x, p.y, negate[i] != 0);
        if (mode == 2 && i % 3 == 2) acc = zk::xyzzu_from_r(zk::xyzzu_to_r(acc));
      }
      r = mode == 2 ? zk::xyzzr_to_std(zk::xyzzu_to_r(acc)) : zk::xyzzu2_to_std(acc);
    }
    std::memcpy(out_xyzz, &r, sizeof r);
    return ZK_OK;
  });
}

// k * P for ONE G2 point on the HOST with the program batch_exp_win_u2_kernel runs (table 1P..8P, signed 4-bit windows, 256
// doublings) on the U-form Fq2 Jacobian arithmetic of curveu.hpp.  out = memory-format Jacobian X, Y, Z (24 u64).
int mi355zk_selftest_g2_scalar_mul_u(const uint64_t affine_pt[16], const uint64_t scalar[4], uint64_t out_xyz[24]) {
  return zk::abi_guard([&]() -> int {
    if (!affine_pt || !scalar || !out_xyz) return ZK_ERR_BAD_ARGS;
    zk::G2Affine base;
    std::memcpy(&base, affine_pt, 128);
    uint32_t s[8];
    std::memcpy(s, scalar, 32);
    zk::JacU2 acc = zk::JacU2::zero();
    if (!base.is_zero()) {
      zk::JacTabU2 tab[8];
      tab[0] = zk::jacu2_tab_from_affine(base.x, base.y);
      for (int e = 2; e <= 8; ++e) {
        const zk::JacTabU2& src = tab[(e & 1) ? e - 2 : e / 2 - 1];
        zk::JacU2 q{src.x, src.y, src.z};
        if (e & 1) zk::jacu2_add_tab(q, tab[0], false);
        else q = zk::jacu2_double(q);
        tab[e - 1] = zk::jacu2_tab_entry(q);
      }
      int dig[65];
      uint32_t carry = 0;
      for (int j = 0; j < 64; ++j) {
        uint32_t d = ((s[j >> 3] >> (4 * (j & 7))) & 15u) + carry;
        carry = d > 8u ? 1u : 0u;
        dig[j] = carry ? (int)d - 16 : (int)d;
      }
      for (int j = 63; j >= 0; --j) {
        for (int rep = 0; rep < 4; ++rep) acc = zk::jacu2_double(acc);
        if (dig[j]) zk::jacu2_add_tab(acc, tab[(dig[j] < 0 ? -dig[j] : dig[j]) - 1], dig[j] < 0);
      }
    }
    const zk::Jacobian<zk::Fq2> r = zk::jacu2_to_std(acc);
    std::memcpy(out_xyz, &r, sizeof r);
    return ZK_OK;
  });
}

int mi355zk_bn254_fr_mul_assign_dev(void* d_a, const void* d_b, size_t n, void* stream) { return zk::abi_guard([&]() -> int { return zk::pointwise(d_a, d_b, n, stream, 0); }); }
int mi355zk_bn254_fr_sub_assign_dev(void* d_a, const void* d_b, size_t n, void* stream) { return zk::abi_guard([&]() -> int { return zk::pointwise(d_a, d_b, n, stream, 1); }); }
int mi355zk_bn254_fr_into_repr_dev(void* d_out, const void* d_in, size_t n, void* stream) {
  return zk::abi_guard([&]() -> int {
    if ((!d_out || !d_in) && n) return ZK_ERR_BAD_ARGS;
    if (n == 0) return ZK_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(zk::fr_into_repr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (zk::Fr*)d_out, (const zk::Fr*)d_in, (uint64_t)n);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
  });
}

int mi355zk_ubench_fp_mul(int which, uint32_t blocks, uint32_t iters, const uint64_t a[4], const uint64_t b[4], uint64_t out[16], float* ms) {
  return zk::abi_guard([&]() -> int {
    if (!a || !b || !out || !ms) return ZK_ERR_BAD_ARGS;
    return which == 0 ? zk::ubench<zk::FqParams>(blocks, iters, a, b, out, ms) : zk::ubench<zk::FrParams>(blocks, iters, a, b, out, ms);
  });
}

}  // extern "C"
