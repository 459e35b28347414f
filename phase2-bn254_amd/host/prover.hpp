// C++ mirror of the CALLER of the hot path: bellman_ce's Groth16 prover, restated over host/bellman.hpp (and through it the C ABI
// of include/mi355zk.h).  Same names, argument meaning and order of operations as
//   bellman/src/groth16/mod.rs:104-198     VerifyingKey, Parameters
//   bellman/src/groth16/mod.rs:429-483     ParameterSource for &Parameters (get_vk / get_h / get_l / get_a / get_b_g1 / get_b_g2)
//   bellman/src/groth16/prover.rs:131-151  ProvingAssignment (what synthesis leaves behind)
//   bellman/src/groth16/prover.rs:202-343  create_proof
// The eight multiexps are QUEUED before the first wait, as the reference queues them on its CpuPool (prover.rs:250-298): here from
// eight std::async threads over the host-buffer entry points, whose bases cache keeps the parameter vectors of pinned Parameters (Parameters::pin) on the device from the
// second proof on.  (prover.py is the twin that keeps the polynomials in HBM between the domain steps.)
#pragma once

#include <future>
#include <memory>
#include <vector>

#include "bellman.hpp"

namespace bellman {
namespace groth16 {

struct VerifyingKey {  // mod.rs:104-131 (the fields create_proof reads)
  G1Affine alpha_g1, beta_g1, delta_g1;
  G2Affine beta_g2, delta_g2;
};

struct Parameters {  // mod.rs:216-238
  VerifyingKey vk;
  std::shared_ptr<const std::vector<G1Affine>> h, l, a, b_g1;
  std::shared_ptr<const std::vector<G2Affine>> b_g2;
  // ParameterSource (mod.rs:429-483)
  const VerifyingKey& get_vk(size_t) const { return vk; }
  SourceBuilder<G1Affine> get_h(size_t) const { return {h, 0}; }
  SourceBuilder<G1Affine> get_l(size_t) const { return {l, 0}; }
  std::pair<SourceBuilder<G1Affine>, SourceBuilder<G1Affine>> get_a(size_t num_inputs, size_t) const { return {{a, 0}, {a, num_inputs}}; }
  std::pair<SourceBuilder<G1Affine>, SourceBuilder<G1Affine>> get_b_g1(size_t num_inputs, size_t) const { return {{b_g1, 0}, {b_g1, num_inputs}}; }
  std::pair<SourceBuilder<G2Affine>, SourceBuilder<G2Affine>> get_b_g2(size_t num_inputs, size_t) const { return {{b_g2, 0}, {b_g2, num_inputs}}; }
  // The five vectors are immutable by type (shared_ptr<const vector>, the reference's Arc<Vec<G>>): pin() promises exactly that
  // to the library (mi355zk_bases_cache_pin), so that the host-buffer multiexps of every later proof find the vectors on the device
  // and only the exponents cross PCIe; unpin() ends the promise -- call it before the Parameters object (the last owner of the
  // vectors) goes away.  Without pin() every multiexp uploads its bases again: correct, slower.
  // pin(true): the library may also keep each vector's WINDOW TABLE on the device (mi355zk_bases_cache_pin_tables): the multiexps of
  // the third proof on run in table mode -- one bucket set for all windows, include/mi355zk.h -- at 13 - 15 times the device memory.
  void pin(bool tables = false) const {
    auto one = [tables](const void* p, size_t n, int group) {
      if (n) (void)(tables ? mi355zk_bases_cache_pin_tables(p, n, group) : mi355zk_bases_cache_pin(p, n, group));
    };
    one(h->data(), h->size(), 1); one(l->data(), l->size(), 1); one(a->data(), a->size(), 1); one(b_g1->data(), b_g1->size(), 1);
    one(b_g2->data(), b_g2->size(), 2);
  }
  void unpin() const {
    for (const void* p : {(const void*)h->data(), (const void*)l->data(), (const void*)a->data(), (const void*)b_g1->data(), (const void*)b_g2->data()})
      if (p) mi355zk_bases_cache_invalidate(p);
  }
};

struct ProvingAssignment {  // prover.rs:131-151; a, b, c, input_assignment, aux_assignment hold Montgomery Fr like Vec<Scalar<E>> / Vec<E::Fr>
  DensityTracker a_aux_density, b_input_density, b_aux_density;
  std::vector<Fr> a, b, c;
  std::vector<Fr> input_assignment, aux_assignment;
};

struct Proof {  // mod.rs:21-26
  G1Affine a;
  G2Affine b;
  G1Affine c;
};

namespace detail {
// field_elements_into_representations (prover.rs:89-108): Montgomery -> canonical, on the device
inline std::shared_ptr<const std::vector<FrRepr>> into_representations(const std::vector<Fr>& v) {
  auto out = std::make_shared<std::vector<FrRepr>>(v.size());
  if (v.empty()) return out;
  void* d = nullptr;
  if (mi355zk_malloc(&d, v.size() * 32) != 0) throw SynthesisError(SynthesisError::Device);
  int rc = mi355zk_memcpy_h2d(d, v.data(), v.size() * 32);
  if (rc == 0) rc = mi355zk_bn254_fr_into_repr_dev(d, d, v.size(), nullptr);
  if (rc == 0) rc = mi355zk_sync(nullptr);
  if (rc == 0) rc = mi355zk_memcpy_d2h(out->data(), d, v.size() * 32);
  (void)mi355zk_free(d);
  if (rc != 0) throw SynthesisError(SynthesisError::Device);
  return out;
}
// single-point group operations of the proof assembly (prover.rs:300-333), on the host like the reference's
inline G1Projective into_projective(const G1Affine& p) {
  G1Projective r{};
  if (p.is_zero()) return r;
  r.x = p.x;
  r.y = p.y;
  r.z = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};  // Fq::one() (fq.rs:39-44)
  return r;
}
inline G2Projective into_projective(const G2Affine& p) {
  G2Projective r{};
  if (p.is_zero()) return r;
  r.x = p.x;
  r.y = p.y;
  r.z = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL, 0, 0, 0, 0};  // Fq2::one()
  return r;
}
inline void add_assign(G1Projective& a, const G1Projective& b) {
  if (mi355zk_bn254_g1_add(reinterpret_cast<uint64_t*>(&a), reinterpret_cast<const uint64_t*>(&b)) != 0) throw SynthesisError(SynthesisError::Device);
}
inline void add_assign(G2Projective& a, const G2Projective& b) {
  if (mi355zk_bn254_g2_add(reinterpret_cast<uint64_t*>(&a), reinterpret_cast<const uint64_t*>(&b)) != 0) throw SynthesisError(SynthesisError::Device);
}
inline void mul_assign(G1Projective& a, const FrRepr& k) {
  if (mi355zk_bn254_g1_mul(reinterpret_cast<uint64_t*>(&a), k.data()) != 0) throw SynthesisError(SynthesisError::Device);
}
inline void mul_assign(G2Projective& a, const FrRepr& k) {
  if (mi355zk_bn254_g2_mul(reinterpret_cast<uint64_t*>(&a), k.data()) != 0) throw SynthesisError(SynthesisError::Device);
}
inline G1Affine into_affine(const G1Projective& p) {
  G1Affine r{};
  if (mi355zk_bn254_g1_to_affine(reinterpret_cast<uint64_t*>(&r), reinterpret_cast<const uint64_t*>(&p)) != 0) throw SynthesisError(SynthesisError::Device);
  return r;
}
inline G2Affine into_affine(const G2Projective& p) {
  G2Affine r{};
  if (mi355zk_bn254_g2_to_affine(reinterpret_cast<uint64_t*>(&r), reinterpret_cast<const uint64_t*>(&p)) != 0) throw SynthesisError(SynthesisError::Device);
  return r;
}
}  // namespace detail

// prover.rs:202-343.  r, s: canonical representations (FrRepr) of the blinding scalars.
inline Proof create_proof(const Worker& worker, const Parameters& params, ProvingAssignment prover, const FrRepr& r, const FrRepr& s) {
  using namespace detail;
  const VerifyingKey& vk = params.get_vk(prover.input_assignment.size());

  // ---- h (prover.rs:217-247)
  std::shared_ptr<const std::vector<FrRepr>> h_repr;
  {
    EvaluationDomain a = EvaluationDomain::from_coeffs(std::move(prover.a));
    EvaluationDomain b = EvaluationDomain::from_coeffs(std::move(prover.b));
    EvaluationDomain c = EvaluationDomain::from_coeffs(std::move(prover.c));
    a.ifft(worker);
    a.coset_fft(worker);
    b.ifft(worker);
    b.coset_fft(worker);
    c.ifft(worker);
    c.coset_fft(worker);
    a.mul_assign(worker, b);
    a.sub_assign(worker, c);
    a.divide_by_z_on_coset(worker);
    a.icoset_fft(worker);
    std::vector<FrRepr> repr = a.into_representations();   // scalars_into_representations (prover.rs:110-129)
    repr.resize(repr.size() - 1);                           // a.truncate(a_len)
    h_repr = std::make_shared<const std::vector<FrRepr>>(std::move(repr));
  }
  // every multiexp is queued before the first wait (prover.rs:250-298)
  auto queue1 = [&](SourceBuilder<G1Affine> src, auto density, std::shared_ptr<const std::vector<FrRepr>> e) {
    return std::async(std::launch::async, [&worker, src, density, e] { return multiexp<G1Affine>(worker, src, density, e).get(); });
  };
  auto queue2 = [&](SourceBuilder<G2Affine> src, auto density, std::shared_ptr<const std::vector<FrRepr>> e) {
    return std::async(std::launch::async, [&worker, src, density, e] { return multiexp<G2Affine>(worker, src, density, e).get(); });
  };
  auto h = queue1(params.get_h(h_repr->size()), FullDensity{}, h_repr);

  // ---- the assignments (prover.rs:256-298)
  auto input_assignment = into_representations(prover.input_assignment);
  auto aux_assignment = into_representations(prover.aux_assignment);
  auto l = queue1(params.get_l(aux_assignment->size()), FullDensity{}, aux_assignment);
  const size_t a_aux_density_total = prover.a_aux_density.get_total_density();
  auto a_src = params.get_a(input_assignment->size(), a_aux_density_total);
  auto a_inputs = queue1(a_src.first, FullDensity{}, input_assignment);
  auto a_aux = queue1(a_src.second, prover.a_aux_density, aux_assignment);
  const size_t b_input_density_total = prover.b_input_density.get_total_density();
  const size_t b_aux_density_total = prover.b_aux_density.get_total_density();
  auto b1_src = params.get_b_g1(b_input_density_total, b_aux_density_total);
  auto b_g1_inputs = queue1(b1_src.first, prover.b_input_density, input_assignment);
  auto b_g1_aux = queue1(b1_src.second, prover.b_aux_density, aux_assignment);
  auto b2_src = params.get_b_g2(b_input_density_total, b_aux_density_total);
  auto b_g2_inputs = queue2(b2_src.first, prover.b_input_density, input_assignment);
  auto b_g2_aux = queue2(b2_src.second, prover.b_aux_density, aux_assignment);

  // ---- assembly (prover.rs:300-343)
  if (vk.delta_g1.is_zero() || vk.delta_g2.is_zero()) throw SynthesisError(SynthesisError::UnexpectedIdentity);  // subversion-CRS check
  G1Projective g_a = into_projective(vk.delta_g1);
  mul_assign(g_a, r);
  add_assign(g_a, into_projective(vk.alpha_g1));
  G2Projective g_b = into_projective(vk.delta_g2);
  mul_assign(g_b, s);
  add_assign(g_b, into_projective(vk.beta_g2));
  G1Projective g_c = into_projective(vk.delta_g1);       // delta_g1 * (r * s): the same point as (delta_g1 * r) * s
  mul_assign(g_c, r);
  mul_assign(g_c, s);
  {
    G1Projective t = into_projective(vk.alpha_g1);
    mul_assign(t, s);
    add_assign(g_c, t);
    t = into_projective(vk.beta_g1);
    mul_assign(t, r);
    add_assign(g_c, t);
  }
  G1Projective a_answer = a_inputs.get();
  add_assign(a_answer, a_aux.get());
  add_assign(g_a, a_answer);
  mul_assign(a_answer, s);
  add_assign(g_c, a_answer);
  G1Projective b1_answer = b_g1_inputs.get();
  add_assign(b1_answer, b_g1_aux.get());
  G2Projective b2_answer = b_g2_inputs.get();
  add_assign(b2_answer, b_g2_aux.get());
  add_assign(g_b, b2_answer);
  mul_assign(b1_answer, r);
  add_assign(g_c, b1_answer);
  add_assign(g_c, h.get());
  add_assign(g_c, l.get());
  return Proof{into_affine(g_a), into_affine(g_b), into_affine(g_c)};
}

}  // namespace groth16
}  // namespace bellman
