// bellman.hpp -- C++17 host-side mirror of the reference's interface for the accelerated path, over the
// C ABI of include/mi355zk.h (header only; link with -lmi355zk).  The reference is compiled (Rust) code and
// this image has no Rust toolchain, so this header is the compiled-language twin of the Rust shim specified
// in INTEGRATION.md: same names, argument meaning and error behaviour as
//   bellman/src/multiexp.rs:330-340   multiexp(pool, bases, density_map, exponents) -> Future<Projective>
//   bellman/src/source.rs:36-140      (Arc<Vec<G>>, usize) source builder, FullDensity, DensityTracker
//   bellman/src/domain.rs:30-260      EvaluationDomain<Scalar>
//   bellman/src/multicore.rs:17-72    Worker
//   bellman/src/cs.rs:156-173         SynthesisError
#pragma once

#include <array>
#include <cstdint>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355zk.h"

namespace bellman {

// ---- cs.rs:156-173 (the variants this path raises)
struct SynthesisError : std::runtime_error {
  enum Kind { UnexpectedIdentity, IoErrorUnexpectedEof, PolynomialDegreeTooLarge, Device } kind;
  long long index;
  SynthesisError(Kind k, long long idx = -1)
      : std::runtime_error(k == UnexpectedIdentity ? "UnexpectedIdentity" : k == IoErrorUnexpectedEof ? "IoError(UnexpectedEof)"
                           : k == PolynomialDegreeTooLarge ? "PolynomialDegreeTooLarge" : "mi355zk device failure"),
        kind(k), index(idx) {}
};

// ---- in-memory element types (the reference's own layouts)
using FrRepr = std::array<uint64_t, 4>;   // canonical scalar (into_repr)
using Fr = std::array<uint64_t, 4>;       // Montgomery limbs (Scalar<Bn256> is a transparent wrapper, group.rs:53)
struct G1Projective { std::array<uint64_t, 4> x, y, z; bool is_zero() const { return !(z[0] | z[1] | z[2] | z[3]); } };
struct G2Projective { std::array<uint64_t, 8> x, y, z; bool is_zero() const { uint64_t o = 0; for (auto v : z) o |= v; return !o; } };
struct G1Affine {  // raw record x || y, all-zero == infinity (ec.rs:653-706)
  std::array<uint64_t, 4> x, y;
  using Projective = G1Projective;
  bool is_zero() const { uint64_t o = 0; for (auto v : x) o |= v; for (auto v : y) o |= v; return !o; }
};
struct G2Affine {  // x.c0 || x.c1 || y.c0 || y.c1
  std::array<uint64_t, 8> x, y;
  using Projective = G2Projective;
  bool is_zero() const { uint64_t o = 0; for (auto v : x) o |= v; for (auto v : y) o |= v; return !o; }
};
static_assert(sizeof(G1Affine) == 64 && sizeof(G2Affine) == 128 && sizeof(G1Projective) == 96 && sizeof(G2Projective) == 192, "ABI layouts");

// ---- multicore.rs:17-72.  The reference's Worker is a CPU pool; here it names the GPU(s) this process drives: Worker(3) one
// device; Worker({0, 1, .., 7}) the single-process multi-GPU mode of mi355zk_init -- host-buffer multiexps of >= 2^20 exponents are
// cut into one point range per device and joined on the host, shorter calls take the devices in turn (include/mi355zk.h).  The
// last Worker constructed defines the library's device set.
class Worker {
 public:
  explicit Worker(int device = -1) {
    int rc = device >= 0 ? mi355zk_init(&device, 1) : mi355zk_init(nullptr, 0);
    if (rc != 0) throw SynthesisError(SynthesisError::Device);
  }
  explicit Worker(const std::vector<int>& devices) {
    if (mi355zk_init(devices.data(), (int)devices.size()) != 0) throw SynthesisError(SynthesisError::Device);
  }
  uint32_t log_num_cpus() const { return 0; }  // multicore.rs:37-39; the serial/parallel FFT split is moot on the GPU
};

// ---- source.rs:72-140
struct FullDensity {
  bool has_query_size() const { return false; }
  size_t get_query_size() const { return 0; }
  const uint32_t* words() const { return nullptr; }
  size_t bits() const { return 0; }
};
class DensityTracker {
 public:
  void add_element() { if ((n_ & 31) == 0) w_.push_back(0); ++n_; }
  void inc(size_t idx) {
    if (idx >= n_) throw std::out_of_range("DensityTracker::inc");  // bv.get(idx).unwrap() (source.rs:129)
    uint32_t m = 1u << (idx & 31);
    if (!(w_[idx >> 5] & m)) { w_[idx >> 5] |= m; ++total_; }
  }
  size_t get_total_density() const { return total_; }
  bool has_query_size() const { return true; }
  size_t get_query_size() const { return n_; }
  const uint32_t* words() const { static const uint32_t zero = 0; return w_.empty() ? &zero : w_.data(); }
  size_t bits() const { return n_; }
 private:
  std::vector<uint32_t> w_;
  size_t n_ = 0, total_ = 0;
};

// ---- source.rs:36-70: the `(Arc<Vec<G>>, usize)` source builder
template <class G>
using SourceBuilder = std::pair<std::shared_ptr<const std::vector<G>>, size_t>;

namespace detail {
inline int msm(const G1Affine* b, size_t nb, size_t off, const FrRepr* e, size_t ne, const uint32_t* d, size_t db, G1Projective* out) {
  return mi355zk_bn254_g1_msm(reinterpret_cast<const uint8_t*>(b), nb, off, reinterpret_cast<const uint64_t*>(e), ne, d, db, reinterpret_cast<uint64_t*>(out));
}
inline int msm(const G2Affine* b, size_t nb, size_t off, const FrRepr* e, size_t ne, const uint32_t* d, size_t db, G2Projective* out) {
  return mi355zk_bn254_g2_msm(reinterpret_cast<const uint8_t*>(b), nb, off, reinterpret_cast<const uint64_t*>(e), ne, d, db, reinterpret_cast<uint64_t*>(out));
}
}  // namespace detail

// ---- multiexp.rs:330.  Returns a READY future (futures-0.1 `future::result`, what singlecore::Worker::compute
// already does, singlecore.rs:33-47); get() rethrows the SynthesisError of the failed call.
template <class G, class D>
std::future<typename G::Projective> multiexp(const Worker&, const SourceBuilder<G>& bases, const D& density_map,
                                             const std::shared_ptr<const std::vector<FrRepr>>& exponents) {
  if (density_map.has_query_size() && density_map.get_query_size() != exponents->size())
    throw std::logic_error("assertion failed: query_size == exponents.len()");  // multiexp.rs:347-352
  std::promise<typename G::Projective> pr;
  typename G::Projective out{};
  int rc = detail::msm(bases.first->data(), bases.first->size(), bases.second, exponents->data(), exponents->size(), density_map.words(),
                       density_map.bits(), &out);
  if (rc == MI355ZK_OK) pr.set_value(out);
  else if (rc == MI355ZK_ERR_UNEXPECTED_IDENTITY) pr.set_exception(std::make_exception_ptr(SynthesisError(SynthesisError::UnexpectedIdentity, mi355zk_last_error_index())));
  else if (rc == MI355ZK_ERR_UNEXPECTED_EOF) pr.set_exception(std::make_exception_ptr(SynthesisError(SynthesisError::IoErrorUnexpectedEof, mi355zk_last_error_index())));
  else pr.set_exception(std::make_exception_ptr(SynthesisError(SynthesisError::Device)));
  return pr.get_future();
}

// ---- domain.rs:30-260 for G = Scalar<Bn256>
class EvaluationDomain {
 public:
  static constexpr uint32_t FR_S = 28;  // fr.rs:31-34
  static EvaluationDomain from_coeffs(std::vector<Fr> coeffs) {  // domain.rs:52-99
    size_t n = coeffs.size();
    if (n > ((size_t)1 << FR_S) - 1) throw SynthesisError(SynthesisError::PolynomialDegreeTooLarge);
    size_t m = 1;
    uint32_t exp = 0;
    while (m < n) {
      m *= 2;
      exp += 1;
      if (exp > FR_S) throw SynthesisError(SynthesisError::PolynomialDegreeTooLarge);
    }
    coeffs.resize(m, Fr{0, 0, 0, 0});  // group_zero (domain.rs:89)
    EvaluationDomain d;
    d.coeffs_ = std::move(coeffs);
    d.exp_ = exp;
    return d;
  }
  const std::vector<Fr>& as_ref() const { return coeffs_; }
  std::vector<Fr>& as_mut() { return coeffs_; }
  std::vector<Fr> into_coeffs() && { return std::move(coeffs_); }
  uint32_t exp() const { return exp_; }
  void fft(const Worker&) { op(MI355ZK_OP_FFT); }                // domain.rs:154
  void ifft(const Worker&) { op(MI355ZK_OP_IFFT); }              // domain.rs:159
  void coset_fft(const Worker&) { op(MI355ZK_OP_COSET_FFT); }    // domain.rs:191
  void icoset_fft(const Worker&) { op(MI355ZK_OP_ICOSET_FFT); }  // domain.rs:197
  // the elementwise steps of the prover's H pipeline (prover.rs:217-241).  This host-vector mirror uploads, runs the device
  // op and downloads; a caller that keeps the polynomial in HBM calls the `_dev` entry points directly (prover.py does).
  Fr z(const Fr& tau) const {                                    // domain.rs:207-212
    Fr out{};
    if (mi355zk_bn254_fr_domain_z(exp_, tau.data(), out.data()) != 0) throw SynthesisError(SynthesisError::Device);
    return out;
  }
  void divide_by_z_on_coset(const Worker&) {                     // domain.rs:217-234
    with_device([&](void* d) { return mi355zk_bn254_fr_divide_by_z_on_coset_dev(d, exp_, nullptr); });
  }
  void mul_assign(const Worker&, const EvaluationDomain& other) {  // domain.rs:236-249
    if (other.coeffs_.size() != coeffs_.size()) throw std::logic_error("assertion failed: self.coeffs.len() == other.coeffs.len()");
    pointwise(other, mi355zk_bn254_fr_mul_assign_dev);
  }
  void sub_assign(const Worker&, const EvaluationDomain& other) {  // domain.rs:251-260
    if (other.coeffs_.size() != coeffs_.size()) throw std::logic_error("assertion failed: self.coeffs.len() == other.coeffs.len()");
    pointwise(other, mi355zk_bn254_fr_sub_assign_dev);
  }
  // scalars_into_representations (prover.rs:110-129): Montgomery -> canonical, on the device
  std::vector<FrRepr> into_representations() const {
    std::vector<FrRepr> out(coeffs_.size());
    void* d = nullptr;
    if (mi355zk_malloc(&d, coeffs_.size() * 32) != 0) throw SynthesisError(SynthesisError::Device);
    int rc = mi355zk_memcpy_h2d(d, coeffs_.data(), coeffs_.size() * 32);
    if (rc == 0) rc = mi355zk_bn254_fr_into_repr_dev(d, d, coeffs_.size(), nullptr);
    if (rc == 0) rc = mi355zk_sync(nullptr);
    if (rc == 0) rc = mi355zk_memcpy_d2h(out.data(), d, coeffs_.size() * 32);
    (void)mi355zk_free(d);
    if (rc != 0) throw SynthesisError(SynthesisError::Device);
    return out;
  }
 private:
  template <class Fn>
  void with_device(Fn&& fn) {
    void* d = nullptr;
    if (mi355zk_malloc(&d, coeffs_.size() * 32) != 0) throw SynthesisError(SynthesisError::Device);
    int rc = mi355zk_memcpy_h2d(d, coeffs_.data(), coeffs_.size() * 32);
    if (rc == 0) rc = fn(d);
    if (rc == 0) rc = mi355zk_sync(nullptr);
    if (rc == 0) rc = mi355zk_memcpy_d2h(coeffs_.data(), d, coeffs_.size() * 32);
    (void)mi355zk_free(d);
    if (rc != 0) throw SynthesisError(SynthesisError::Device);
  }
  void pointwise(const EvaluationDomain& other, int (*fn)(void*, const void*, size_t, void*)) {
    void* db = nullptr;
    if (mi355zk_malloc(&db, coeffs_.size() * 32) != 0) throw SynthesisError(SynthesisError::Device);
    int rc0 = mi355zk_memcpy_h2d(db, other.coeffs_.data(), coeffs_.size() * 32);
    try {
      if (rc0 != 0) throw SynthesisError(SynthesisError::Device);
      with_device([&](void* d) { return fn(d, db, coeffs_.size(), nullptr); });
    } catch (...) {
      (void)mi355zk_free(db);
      throw;
    }
    (void)mi355zk_free(db);
  }
  void op(int which) {
    int rc = mi355zk_bn254_fr_domain_op(reinterpret_cast<uint64_t*>(coeffs_.data()), exp_, which);
    if (rc != 0) throw SynthesisError(SynthesisError::Device);
  }
  std::vector<Fr> coeffs_;
  uint32_t exp_ = 0;
};

}  // namespace bellman
