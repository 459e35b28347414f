// ceremony.hpp -- C++17 host-side mirror of the ceremony-side callers next to the hot path, over
// the C ABI of include/mi355zk.h (header only; link with -lmi355zk).  Same names and argument meaning as the reference
// functions they replace; host vectors in, host vectors out (the device round trip is inside), errors as exceptions:
//   batch_exp(bases, exps)               powersoftau/src/batched_accumulator.rs:1130-1181   (exps[i] * coeff folded by the caller)
//   batch_exp(bases, coeff)              phase2/src/parameters.rs:423-470                    (every point by delta^-1)
//   dense_multiexp(bases, exponents)     powersoftau/src/utils.rs:189-292
//   merge_pairs(v1, v2, rho)             powersoftau/src/utils.rs:112-131, phase2/src/utils.rs:59-105 (rho drawn by the caller)
//   power_pairs(v, rho)                  powersoftau/src/utils.rs:133-135
//   eval_qap(bases, row_ptr, col, coeff) the per-variable sums of MPCParameters::new, phase2/src/parameters.rs:225-294
//   point_fft / point_ifft(points)       EvaluationDomain<Point<G>>::{fft, ifft}, powersoftau/src/bin/prepare_phase2.rs:68-131
//   decode_points / encode_points        EncodedPoint::{into_affine[_unchecked], from_affine}, pairing/src/bn256/ec.rs:763-946, 1136-1344
// The Python twin is phase2-bn254_amd/ceremony.py (device-resident tensors, no copies).
#pragma once

#include <cstring>

#include "bellman.hpp"

namespace ceremony {

using bellman::FrRepr;
using bellman::G1Affine;
using bellman::G1Projective;
using bellman::G2Affine;
using bellman::G2Projective;
using bellman::SynthesisError;

// pairing/src/lib.rs GroupDecodingError, as raised by EncodedPoint::into_affine
struct GroupDecodingError : std::runtime_error {
  enum Kind { NotOnCurve = 4, NotInSubgroup = 5, CoordinateDecodingError = 6, UnexpectedCompressionMode = 7, UnexpectedInformation = 8 } kind;
  long long index;
  GroupDecodingError(int k, long long idx) : std::runtime_error("GroupDecodingError " + std::to_string(k) + " at point " + std::to_string(idx)), kind((Kind)k), index(idx) {}
};

namespace detail {
// device buffer holding a copy of a host array
class DeviceArray {
 public:
  DeviceArray(const void* host, size_t bytes) : bytes_(bytes) {
    if (mi355zk_malloc(&p_, bytes ? bytes : 1) != 0) throw SynthesisError(SynthesisError::Device);
    if (host && bytes && mi355zk_memcpy_h2d(p_, host, bytes) != 0) { mi355zk_free(p_); throw SynthesisError(SynthesisError::Device); }
  }
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  ~DeviceArray() { mi355zk_free(p_); }
  void* get() const { return p_; }
  void download(void* host) const {
    if (mi355zk_sync(nullptr) != 0 || (bytes_ && mi355zk_memcpy_d2h(host, p_, bytes_) != 0)) throw SynthesisError(SynthesisError::Device);
  }
 private:
  void* p_ = nullptr;
  size_t bytes_;
};
inline void check(int rc) { if (rc != 0) throw SynthesisError(SynthesisError::Device); }
template <class G> struct Abi;
template <> struct Abi<G1Affine> {
  static int batch_exp(void* o, const void* b, const void* s, size_t n, int same) { return mi355zk_bn254_g1_batch_exp_dev(o, b, s, n, same, nullptr); }
  static int dense(const void* b, const void* s, size_t n, uint64_t* out) { return mi355zk_bn254_g1_dense_multiexp_dev(b, s, n, nullptr, out); }
  static int merge(const void* a, const void* b, const void* r, size_t n, uint64_t* s, uint64_t* sx) { return mi355zk_bn254_g1_merge_pairs_dev(a, b, r, n, nullptr, s, sx); }
  static int matvec(void* o, const void* b, size_t nb, const uint32_t* rp, const uint32_t* c, const void* k, size_t rows, size_t nnz, int flags) { return mi355zk_bn254_g1_sparse_matvec_dev(o, b, nb, rp, c, k, rows, nnz, nullptr, flags); }
  static int fft(void* p, uint32_t log_n, int inv) { return mi355zk_bn254_g1_point_fft_dev(p, log_n, inv, nullptr); }
  static int decode(void* o, const void* in, size_t n, int c, int chk, long long* idx) { return mi355zk_bn254_g1_decode_dev(o, in, n, c, chk, nullptr, idx); }
  static int encode(void* o, const void* in, size_t n, int c) { return mi355zk_bn254_g1_encode_dev(o, in, n, c, nullptr); }
  static constexpr size_t enc_size(bool compressed) { return compressed ? 32 : 64; }
};
template <> struct Abi<G2Affine> {
  static int batch_exp(void* o, const void* b, const void* s, size_t n, int same) { return mi355zk_bn254_g2_batch_exp_dev(o, b, s, n, same, nullptr); }
  static int dense(const void* b, const void* s, size_t n, uint64_t* out) { return mi355zk_bn254_g2_dense_multiexp_dev(b, s, n, nullptr, out); }
  static int merge(const void* a, const void* b, const void* r, size_t n, uint64_t* s, uint64_t* sx) { return mi355zk_bn254_g2_merge_pairs_dev(a, b, r, n, nullptr, s, sx); }
  static int matvec(void* o, const void* b, size_t nb, const uint32_t* rp, const uint32_t* c, const void* k, size_t rows, size_t nnz, int flags) { return mi355zk_bn254_g2_sparse_matvec_dev(o, b, nb, rp, c, k, rows, nnz, nullptr, flags); }
  static int fft(void* p, uint32_t log_n, int inv) { return mi355zk_bn254_g2_point_fft_dev(p, log_n, inv, nullptr); }
  static int decode(void* o, const void* in, size_t n, int c, int chk, long long* idx) { return mi355zk_bn254_g2_decode_dev(o, in, n, c, chk, nullptr, idx); }
  static int encode(void* o, const void* in, size_t n, int c) { return mi355zk_bn254_g2_encode_dev(o, in, n, c, nullptr); }
  static constexpr size_t enc_size(bool compressed) { return compressed ? 64 : 128; }
};
}  // namespace detail

// trusted_subgroup (batch_exp, eval_qap, point_fft): the caller's promise MI355ZK_G2_TRUSTED_SUBGROUP -- every G2 record is in the order-r
// subgroup, the kernels may split scalars over the twist's endomorphism.  Default: plain windows, the reference's wNAF result for every
// record its decoders admit.  Ignored for G1.
// out[i] = exps[i] * bases[i], normalised to affine (batched_accumulator.rs:1130-1181)
template <class G>
std::vector<G> batch_exp(const std::vector<G>& bases, const std::vector<FrRepr>& exps, bool trusted_subgroup = false) {
  if (exps.size() != bases.size()) throw std::invalid_argument("batch_exp: one exponent per base");
  detail::DeviceArray b(bases.data(), bases.size() * sizeof(G)), e(exps.data(), exps.size() * 32), o(nullptr, bases.size() * sizeof(G));
  detail::check(detail::Abi<G>::batch_exp(o.get(), b.get(), e.get(), bases.size(), trusted_subgroup ? MI355ZK_G2_TRUSTED_SUBGROUP : 0));
  std::vector<G> out(bases.size());
  o.download(out.data());
  return out;
}
// out[i] = coeff * bases[i] (parameters.rs:423-470)
template <class G>
std::vector<G> batch_exp(const std::vector<G>& bases, const FrRepr& coeff, bool trusted_subgroup = false) {
  detail::DeviceArray b(bases.data(), bases.size() * sizeof(G)), e(coeff.data(), 32), o(nullptr, bases.size() * sizeof(G));
  detail::check(detail::Abi<G>::batch_exp(o.get(), b.get(), e.get(), bases.size(), MI355ZK_EXP_SAME_SCALAR | (trusted_subgroup ? MI355ZK_G2_TRUSTED_SUBGROUP : 0)));
  std::vector<G> out(bases.size());
  o.download(out.data());
  return out;
}

template <class G>
typename G::Projective dense_multiexp(const std::vector<G>& bases, const std::vector<FrRepr>& exponents) {
  if (exponents.size() != bases.size()) throw std::invalid_argument("dense_multiexp: one exponent per base");  // utils.rs:193-195 panics
  detail::DeviceArray b(bases.data(), bases.size() * sizeof(G)), e(exponents.data(), exponents.size() * 32);
  typename G::Projective out{};
  detail::check(detail::Abi<G>::dense(b.get(), e.get(), bases.size(), reinterpret_cast<uint64_t*>(&out)));
  return out;
}

// (sum rho_i v1[i], sum rho_i v2[i]): one digit extraction and one sort for both sums
template <class G>
std::pair<typename G::Projective, typename G::Projective> merge_pairs(const std::vector<G>& v1, const std::vector<G>& v2, const std::vector<FrRepr>& rho) {
  if (v1.size() != v2.size() || rho.size() != v1.size()) throw std::invalid_argument("merge_pairs: equal lengths");  // utils.rs:118 asserts
  detail::DeviceArray a(v1.data(), v1.size() * sizeof(G)), b(v2.data(), v2.size() * sizeof(G)), r(rho.data(), rho.size() * 32);
  std::pair<typename G::Projective, typename G::Projective> out{};
  detail::check(detail::Abi<G>::merge(a.get(), b.get(), r.get(), v1.size(), reinterpret_cast<uint64_t*>(&out.first), reinterpret_cast<uint64_t*>(&out.second)));
  return out;
}
template <class G>
std::pair<typename G::Projective, typename G::Projective> power_pairs(const std::vector<G>& v, const std::vector<FrRepr>& rho) {
  return merge_pairs(std::vector<G>(v.begin(), v.end() - 1), std::vector<G>(v.begin() + 1, v.end()), rho);  // utils.rs:133-135
}

// out[v] = sum over the terms t in [row_ptr[v], row_ptr[v+1]) of coeff[t] * bases[col[t]], affine (parameters.rs:281-294 + batch_normalization)
template <class G>
std::vector<G> eval_qap(const std::vector<G>& bases, const std::vector<uint32_t>& row_ptr, const std::vector<uint32_t>& col, const std::vector<FrRepr>& coeff,
                        bool trusted_subgroup = false) {
  if (row_ptr.empty() || col.size() != coeff.size() || row_ptr.back() != col.size()) throw std::invalid_argument("eval_qap: malformed CSR");
  const size_t rows = row_ptr.size() - 1;
  detail::DeviceArray b(bases.data(), bases.size() * sizeof(G)), rp(row_ptr.data(), row_ptr.size() * 4), c(col.data(), col.size() * 4),
      k(coeff.data(), coeff.size() * 32), o(nullptr, rows * sizeof(G));
  detail::check(detail::Abi<G>::matvec(o.get(), b.get(), bases.size(), (const uint32_t*)rp.get(), (const uint32_t*)c.get(), k.get(), rows, col.size(),
                                      trusted_subgroup ? MI355ZK_G2_TRUSTED_SUBGROUP : 0));
  std::vector<G> out(rows);
  o.download(out.data());
  return out;
}

// in place on `points` (a power-of-two number of them); the ifft includes the 1/m scaling and the normalisation (prepare_phase2.rs:102-131)
template <class G>
void point_fft(std::vector<G>& points, bool inverse, bool trusted_subgroup = false) {
  const size_t n = points.size();
  if (n == 0 || (n & (n - 1))) throw std::invalid_argument("point_fft: power-of-two length");
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) ++log_n;
  detail::DeviceArray p(points.data(), n * sizeof(G));
  detail::check(detail::Abi<G>::fft(p.get(), log_n, (inverse ? MI355ZK_FFT_INVERSE : 0) | (trusted_subgroup ? MI355ZK_G2_TRUSTED_SUBGROUP : 0)));
  p.download(points.data());
}
template <class G> void point_ifft(std::vector<G>& points, bool trusted_subgroup = false) { point_fft(points, true, trusted_subgroup); }

template <class G>
std::vector<uint8_t> encode_points(const std::vector<G>& points, bool compressed) {
  const size_t sz = detail::Abi<G>::enc_size(compressed);
  detail::DeviceArray p(points.data(), points.size() * sizeof(G)), o(nullptr, points.size() * sz);
  detail::check(detail::Abi<G>::encode(o.get(), p.get(), points.size(), compressed ? 1 : 0));
  std::vector<uint8_t> out(points.size() * sz);
  o.download(out.data());
  return out;
}
// throws GroupDecodingError for the first bad record (into_affine when checked, into_affine_unchecked otherwise)
template <class G>
std::vector<G> decode_points(const std::vector<uint8_t>& data, bool compressed, bool checked = true) {
  const size_t sz = detail::Abi<G>::enc_size(compressed);
  if (data.size() % sz) throw std::invalid_argument("decode_points: whole records");
  const size_t n = data.size() / sz;
  detail::DeviceArray in(data.data(), data.size()), o(nullptr, n * sizeof(G));
  long long idx = -1;
  int rc = detail::Abi<G>::decode(o.get(), in.get(), n, compressed ? 1 : 0, checked ? 1 : 0, &idx);
  if (rc >= 4 && rc <= 8) throw GroupDecodingError(rc, idx);
  detail::check(rc);
  std::vector<G> out(n);
  o.download(out.data());
  return out;
}

}  // namespace ceremony
