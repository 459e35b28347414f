// Point codecs for BN254 G1 / G2 on gfx950: the reference's wire encodings <-> the raw affine records the kernels use.
//
// Reference ("point codecs"): pairing/src/bn256/ec.rs
//   G1Uncompressed :763-845   G1Compressed :867-946   G2Uncompressed :1136-1229   G2Compressed :1255-1344
//   get_point_from_x :110-131 (y = sqrt(x^3 + b), the root picked by the "greatest" flag)   is_on_curve :133-148
//   Fq::sqrt (ff_derive, q = 3 mod 4)   Fq2::sqrt pairing/src/bn256/fq2.rs:211-261
// powersoftau reads and writes its accumulators through these (batched_accumulator.rs read_points_chunk /
// write_point); a compressed response costs one 254-bit exponentiation per point, which is why this is GPU work.
//
// Wire format: big-endian canonical coordinates (x then y; Fq2 as c1 then c0); top bits of byte 0: bit 7 = y is the
// lexicographically larger root (compressed), bit 6 = infinity.  Decoded points are raw records (Montgomery limbs,
// all-zero = infinity).  One lane per record.  The first failing record (lowest index) is reported with the code
// of its GroupDecodingError: 4 NotOnCurve, 6 CoordinateDecodingError, 7 UnexpectedCompressionMode, 8 UnexpectedInformation.
//
// Reference quirk reproduced for bit-exactness: Fq2::sqrt compares against fq.rs:434-439 NEGATIVE_ONE,
// which is -(2^256 mod r) for the scalar modulus r; so it never answers None (a non-residue x yields a "point" off
// the curve, exactly as the reference's G2Compressed::into_affine does) and never takes its alpha == -1 branch.
#include <hip/hip_runtime.h>

#include "../../include/mi355zk.h"
#include "curve.hpp"
#include "device_util.hpp"

namespace zk {
namespace {

constexpr int DEC_NOT_ON_CURVE = 4, DEC_COORD = 6, DEC_COMPRESSION_MODE = 7, DEC_UNEXPECTED_INFO = 8;

struct CodecConsts {
  uint32_t e_q3_4[8];   // (q - 3) / 4
  uint32_t e_q1_2[8];   // (q - 1) / 2
  Fq b1;                // 3
  Fq2 b2;               // 3 / (9 + u)
  Fq neg_one;           // -1
  Fq quirk_neg_one;     // the reference's NEGATIVE_ONE: Fr's -1 limbs
};

// 32 big-endian bytes -> canonical limbs; `mask_flags`: clear the two flag bits of the first byte
__device__ __forceinline__ Fq be_load(const uint32_t* w, bool mask_flags) {
  Fq r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.l[7 - k] = __builtin_bswap32(w[k]);
  if (mask_flags) r.l[7] &= 0x3fffffffu;
  return r;
}
__device__ __forceinline__ void be_store(uint32_t* w, const Fq& canon) {
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = __builtin_bswap32(canon.l[7 - k]);
}
__device__ __forceinline__ bool lt_modulus(const Fq& c) {
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (c.l[i] != FqParams::P[i]) return c.l[i] < FqParams::P[i];
  }
  return false;
}
// canonical order (Ord for Fq = order of into_repr()): -1, 0, 1
__device__ __forceinline__ int cmp_fq(const Fq& a, const Fq& b) {
  Fq x = to_canonical(a), y = to_canonical(b);
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (x.l[i] != y.l[i]) return x.l[i] < y.l[i] ? -1 : 1;
  }
  return 0;
}
__device__ __forceinline__ int cmp_fq2(const Fq2& a, const Fq2& b) {  // fq2.rs:20-31: c1 first
  int c = cmp_fq(a.c1, b.c1);
  return c ? c : cmp_fq(a.c0, b.c0);
}

__device__ Fq2 f2_pow(const Fq2& a, const uint32_t* e) {
  Fq2 res = Fq2::one();
  bool found = false;
  for (int i = 255; i >= 0; --i) {
    bool bit = (e[i >> 5] >> (i & 31)) & 1;
    if (found) res = sqr(res);
    else found = bit;
    if (bit) res = mul(res, a);
  }
  return res;
}

// ff_derive's sqrt for q = 3 mod 4
__device__ bool fq_sqrt(Fq& r, const Fq& a, const CodecConsts& K) {
  Fq a1 = pow_limbs(a, K.e_q3_4, 8);
  Fq a0 = mul(sqr(a1), a);
  if (a0 == K.neg_one) return false;
  r = mul(a1, a);
  return true;
}
// fq2.rs:211-261
__device__ bool fq2_sqrt_ref(Fq2& r, const Fq2& a, const CodecConsts& K) {
  if (a.is_zero()) { r = Fq2::zero(); return true; }
  Fq2 a1 = f2_pow(a, K.e_q3_4);
  Fq2 alpha = mul(sqr(a1), a);
  Fq2 a0 = mul(Fq2{alpha.c0, neg(alpha.c1)}, alpha);  // frobenius_map(1) = conjugation
  const Fq2 neg1{K.quirk_neg_one, Fq::zero()};
  if (a0 == neg1) return false;
  a1 = mul(a1, a);
  if (alpha == neg1) {
    a1 = mul(a1, Fq2{Fq::zero(), Fq::one()});
  } else {
    alpha = f2_pow(add(alpha, Fq2::one()), K.e_q1_2);
    a1 = mul(a1, alpha);
  }
  r = a1;
  return true;
}

__device__ __forceinline__ void report(unsigned long long* err, size_t i, int code) {
  atomicMin(err, ((unsigned long long)i << 8) | (unsigned)code);
}

// WORDS = 8 (G1 compressed), 16 (G1 uncompressed / G2 compressed), 32 (G2 uncompressed)
template <int WORDS>
__device__ __forceinline__ bool rest_is_zero(const uint32_t* w) {
  uint32_t acc = w[0] & ~0xc0u;  // byte 0 is the low byte of the first little-endian word; its two top bits are flags
#pragma unroll
  for (int k = 1; k < WORDS; ++k) acc |= w[k];
  return acc == 0;
}

template <bool COMPRESSED>
__global__ void __launch_bounds__(256) g1_decode_kernel(const uint32_t* __restrict__ in, G1Affine* __restrict__ out, size_t n, int checked,
                                                       CodecConsts K, unsigned long long* __restrict__ err) {
  constexpr int WORDS = COMPRESSED ? 8 : 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) w[k] = in[i * WORDS + k];
  G1Affine p{Fq::zero(), Fq::zero()};
  const uint32_t b0 = w[0] & 0xffu;
  int code = 0;
  if (b0 & 0x40u) {
    if (!rest_is_zero<WORDS>(w)) code = DEC_UNEXPECTED_INFO;
  } else if (!COMPRESSED && (b0 & 0x80u)) {
    code = DEC_UNEXPECTED_INFO;  // ec.rs:797-801
  } else {
    const bool greatest = (b0 & 0x80u) != 0;
    Fq xc = be_load(w, true);
    if (!lt_modulus(xc)) code = DEC_COORD;
    else {
      Fq x = from_canonical(xc);
      if (COMPRESSED) {
        Fq y;
        if (!fq_sqrt(y, add(mul(sqr(x), x), K.b1), K)) code = DEC_NOT_ON_CURVE;
        else {
          Fq negy = neg(y);
          p.x = x;
          p.y = ((cmp_fq(y, negy) < 0) != greatest) ? y : negy;
        }
      } else {
        Fq yc = be_load(w + 8, false);
        if (!lt_modulus(yc)) code = DEC_COORD;
        else {
          Fq y = from_canonical(yc);
          if (checked && !(sqr(y) == add(mul(sqr(x), x), K.b1))) code = DEC_NOT_ON_CURVE;
          else { p.x = x; p.y = y; }
        }
      }
    }
  }
  if (code) report(err, i, code);
  out[i] = p;
}

template <bool COMPRESSED>
__global__ void __launch_bounds__(256) g1_encode_kernel(const G1Affine* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  constexpr int WORDS = COMPRESSED ? 8 : 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = in[i];
  uint32_t w[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) w[k] = 0;
  if (p.is_zero()) {
    w[0] = 0x40u;
  } else {
    be_store(w, to_canonical(p.x));
    if (COMPRESSED) {
      if (cmp_fq(p.y, neg(p.y)) > 0) w[0] |= 0x80u;
    } else {
      be_store(w + (COMPRESSED ? 0 : 8), to_canonical(p.y));
    }
  }
#pragma unroll
  for (int k = 0; k < WORDS; ++k) out[i * WORDS + k] = w[k];
}

template <bool COMPRESSED>
__global__ void __launch_bounds__(256) g2_decode_kernel(const uint32_t* __restrict__ in, G2Affine* __restrict__ out, size_t n, int checked,
                                                       CodecConsts K, unsigned long long* __restrict__ err) {
  constexpr int WORDS = COMPRESSED ? 16 : 32;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) w[k] = in[i * WORDS + k];
  G2Affine p{Fq2::zero(), Fq2::zero()};
  const uint32_t b0 = w[0] & 0xffu;
  int code = 0;
  if (!COMPRESSED && (b0 & 0x80u)) {
    code = DEC_COMPRESSION_MODE;  // ec.rs:1158-1161, tested before the infinity flag
  } else if (b0 & 0x40u) {
    if (!rest_is_zero<WORDS>(w)) code = DEC_UNEXPECTED_INFO;
  } else {
    const bool greatest = (b0 & 0x80u) != 0;
    Fq x1c = be_load(w, true), x0c = be_load(w + 8, false);  // c1 first on the wire
    if (!lt_modulus(x0c) || !lt_modulus(x1c)) code = DEC_COORD;
    else {
      Fq2 x{from_canonical(x0c), from_canonical(x1c)};
      if (COMPRESSED) {
        Fq2 y;
        if (!fq2_sqrt_ref(y, add(mul(sqr(x), x), K.b2), K)) code = DEC_NOT_ON_CURVE;
        else {
          Fq2 negy = neg(y);
          p.x = x;
          p.y = ((cmp_fq2(y, negy) < 0) != greatest) ? y : negy;
        }
      } else {
        Fq y1c = be_load(w + (COMPRESSED ? 0 : 16), false), y0c = be_load(w + (COMPRESSED ? 0 : 24), false);
        if (!lt_modulus(y0c) || !lt_modulus(y1c)) code = DEC_COORD;
        else {
          Fq2 y{from_canonical(y0c), from_canonical(y1c)};
          if (checked && !(sqr(y) == add(mul(sqr(x), x), K.b2))) code = DEC_NOT_ON_CURVE;
          else { p.x = x; p.y = y; }
        }
      }
    }
  }
  if (code) report(err, i, code);
  out[i] = p;
}

template <bool COMPRESSED>
__global__ void __launch_bounds__(256) g2_encode_kernel(const G2Affine* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  constexpr int WORDS = COMPRESSED ? 16 : 32;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G2Affine p = in[i];
  uint32_t w[WORDS];
#pragma unroll
  for (int k = 0; k < WORDS; ++k) w[k] = 0;
  if (p.is_zero()) {
    w[0] = 0x40u;
  } else {
    be_store(w, to_canonical(p.x.c1));
    be_store(w + 8, to_canonical(p.x.c0));
    if (COMPRESSED) {
      if (cmp_fq2(p.y, neg(p.y)) > 0) w[0] |= 0x80u;
    } else {
      be_store(w + (COMPRESSED ? 0 : 16), to_canonical(p.y.c1));
      be_store(w + (COMPRESSED ? 0 : 24), to_canonical(p.y.c0));
    }
  }
#pragma unroll
  for (int k = 0; k < WORDS; ++k) out[i * WORDS + k] = w[k];
}

CodecConsts make_consts() {
  CodecConsts K{};
  // (q - 3) / 4 and (q - 1) / 2 from the modulus
  uint32_t t[8];
  for (int i = 0; i < 8; ++i) t[i] = FqParams::P[i];
  t[0] -= 3;  // P[0] ends in ...47: no borrow
  for (int i = 0; i < 8; ++i) K.e_q3_4[i] = (t[i] >> 2) | (i < 7 ? t[i + 1] << 30 : 0);
  for (int i = 0; i < 8; ++i) t[i] = FqParams::P[i];
  t[0] -= 1;
  for (int i = 0; i < 8; ++i) K.e_q1_2[i] = (t[i] >> 1) | (i < 7 ? t[i + 1] << 31 : 0);
  const Fq one = Fq::one();
  const Fq three = add(add(one, one), one);
  const Fq nine = add(add(three, three), three);
  K.b1 = three;                                                        // fq.rs:11-16
  K.b2 = mul(Fq2{three, Fq::zero()}, inv(Fq2{nine, one}));            // 3 / (9 + u): the value of fq.rs:18-31
  K.neg_one = neg(one);
  const Fr neg_one_r = neg(Fr::one());                                // the reference's NEGATIVE_ONE (fq.rs:434-439)
  for (int i = 0; i < 8; ++i) K.quirk_neg_one.l[i] = neg_one_r.l[i];
  return K;
}

template <class DecodeLaunch>
int run_decode(size_t n, hipStream_t st, long long* err_index, DecodeLaunch&& launch) {
  if (err_index) *err_index = -1;
  if (n == 0) return ZK_OK;
  unsigned long long* d_err = nullptr;
  ZK_HIP(hipMalloc(&d_err, 8));
  hipError_t e = hipMemsetAsync(d_err, 0xff, 8, st);
  unsigned long long h_err = ~0ull;
  if (e == hipSuccess) {
    launch(d_err);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_err);
  ZK_HIP(e);
  if (h_err == ~0ull) return ZK_OK;
  if (err_index) *err_index = (long long)(h_err >> 8);
  return (int)(h_err & 0xff);
}

}  // namespace

int codec_decode(int group, void* d_out, const void* d_in, size_t n, int compressed, int checked, hipStream_t st, long long* err_index) {
  static const CodecConsts K = make_consts();
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  return run_decode(n, st, err_index, [&](unsigned long long* d_err) {
    const uint32_t* in = (const uint32_t*)d_in;
    if (group == 1) {
      if (compressed) hipLaunchKernelGGL(g1_decode_kernel<true>, grid, block, 0, st, in, (G1Affine*)d_out, n, checked, K, d_err);
      else hipLaunchKernelGGL(g1_decode_kernel<false>, grid, block, 0, st, in, (G1Affine*)d_out, n, checked, K, d_err);
    } else {
      if (compressed) hipLaunchKernelGGL(g2_decode_kernel<true>, grid, block, 0, st, in, (G2Affine*)d_out, n, checked, K, d_err);
      else hipLaunchKernelGGL(g2_decode_kernel<false>, grid, block, 0, st, in, (G2Affine*)d_out, n, checked, K, d_err);
    }
  });
}

int codec_encode(int group, void* d_out, const void* d_in, size_t n, int compressed, hipStream_t st) {
  if (n == 0) return ZK_OK;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  uint32_t* out = (uint32_t*)d_out;
  if (group == 1) {
    if (compressed) hipLaunchKernelGGL(g1_encode_kernel<true>, grid, block, 0, st, (const G1Affine*)d_in, out, n);
    else hipLaunchKernelGGL(g1_encode_kernel<false>, grid, block, 0, st, (const G1Affine*)d_in, out, n);
  } else {
    if (compressed) hipLaunchKernelGGL(g2_encode_kernel<true>, grid, block, 0, st, (const G2Affine*)d_in, out, n);
    else hipLaunchKernelGGL(g2_encode_kernel<false>, grid, block, 0, st, (const G2Affine*)d_in, out, n);
  }
  ZK_HIP(hipGetLastError());
  return ZK_OK;
}

}  // namespace zk
