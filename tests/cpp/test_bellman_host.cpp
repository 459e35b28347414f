// C++ twin of the reference's own tests for this path, written against host/bellman.hpp (the product) with
// the CPU oracle as the checker (TEST INFRASTRUCTURE: links oracle/_build/liboracle.so).
//   bellman/src/multiexp.rs:479-518   naive sum == multiexp           (test_with_bls12, restated on BN254)
//   bellman/src/domain.rs:427-463     ifft(fft(a)) == a, coset variants
//   bellman/src/source.rs:44-70       Source errors through the future
// Run by tests/test_gpu_cpp_host.py on the GPU box; prints "ok <name>" lines and exits 0 on success.
#include <cstdio>
#include <cstring>
#include <random>

#include "../../phase2-bn254_amd/host/bellman.hpp"

extern "C" {
void oracle_g1_mul_many_affine(uint64_t* out_affine, const uint64_t base_affine[8], const uint64_t* ks, size_t n);
void oracle_g1_naive_multiexp(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]);
void oracle_g1_to_affine(uint64_t r[8], const uint64_t p[12]);
int oracle_fr_domain_op(uint64_t* a, uint32_t log_n, int op, uint32_t log_cpus);
void oracle_fe_from_canonical(int which, uint64_t r[4], const uint64_t a[4]);
}

using namespace bellman;
static const uint64_t R_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static FrRepr rand_scalar(std::mt19937_64& g) {
  for (;;) {
    FrRepr r{g(), g(), g(), g() & ((1ULL << 62) - 1)};
    for (int i = 3; i >= 0; --i) {
      if (r[i] < R_LIMBS[i]) return r;
      if (r[i] > R_LIMBS[i]) break;
    }
  }
}
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  Worker worker(0);
  std::mt19937_64 gen(0x3dbe6259);
  uint64_t g1[8];
  { uint64_t one[4] = {1, 0, 0, 0}, two[4] = {2, 0, 0, 0}; oracle_fe_from_canonical(0, g1, one); oracle_fe_from_canonical(0, g1 + 4, two); }

  {  // naive == multiexp
    const size_t n = 1500;
    std::vector<FrRepr> ks(n), es(n);
    for (auto& k : ks) k = rand_scalar(gen);
    for (auto& e : es) e = rand_scalar(gen);
    auto bases = std::make_shared<std::vector<G1Affine>>(n);
    oracle_g1_mul_many_affine(reinterpret_cast<uint64_t*>(bases->data()), g1, reinterpret_cast<const uint64_t*>(ks.data()), n);
    auto exps = std::make_shared<const std::vector<FrRepr>>(es);
    G1Projective fast = multiexp<G1Affine>(worker, {bases, 0}, FullDensity{}, exps).get();
    uint64_t naive[12], a[8], b[8];
    oracle_g1_naive_multiexp(reinterpret_cast<const uint64_t*>(bases->data()), reinterpret_cast<const uint64_t*>(es.data()), n, naive);
    oracle_g1_to_affine(a, reinterpret_cast<const uint64_t*>(&fast));
    oracle_g1_to_affine(b, naive);
    CHECK(std::memcmp(a, b, 64) == 0);
    std::puts("ok multiexp_equals_naive");

    // density map + errors through the future
    DensityTracker d;
    for (size_t i = 0; i < n; ++i) d.add_element();
    for (size_t i = 0; i < n; i += 3) d.inc(i);
    CHECK(d.get_total_density() == (n + 2) / 3);
    G1Projective sparse = multiexp<G1Affine>(worker, {bases, 0}, d, exps).get();
    CHECK(!sparse.is_zero());
    auto short_bases = std::make_shared<std::vector<G1Affine>>(bases->begin(), bases->begin() + 10);
    try { multiexp<G1Affine>(worker, {short_bases, 0}, FullDensity{}, exps).get(); CHECK(false); }
    catch (const SynthesisError& e) { CHECK(e.kind == SynthesisError::IoErrorUnexpectedEof && e.index == 10); }
    (*short_bases)[4] = G1Affine{};
    try { multiexp<G1Affine>(worker, {short_bases, 0}, FullDensity{}, exps).get(); CHECK(false); }
    catch (const SynthesisError& e) { CHECK(e.kind == SynthesisError::UnexpectedIdentity && e.index == 4); }
    std::puts("ok density_and_source_errors");
  }

  for (uint32_t log_n : {1u, 7u, 13u}) {  // fft_consistency + oracle parity of all four ops
    std::vector<Fr> a((size_t)1 << log_n);
    for (auto& x : a) x = rand_scalar(gen);  // any value < r is a valid Montgomery representation
    for (int op = 0; op < 4; ++op) {
      std::vector<Fr> want = a;
      CHECK(oracle_fr_domain_op(reinterpret_cast<uint64_t*>(want.data()), log_n, op, 31) == 0);
      EvaluationDomain d = EvaluationDomain::from_coeffs(a);
      if (op == 0) d.fft(worker); else if (op == 1) d.ifft(worker); else if (op == 2) d.coset_fft(worker); else d.icoset_fft(worker);
      CHECK(std::memcmp(d.as_ref().data(), want.data(), want.size() * 32) == 0);
    }
    EvaluationDomain d = EvaluationDomain::from_coeffs(a);
    d.fft(worker); d.ifft(worker);
    CHECK(std::memcmp(d.as_ref().data(), a.data(), a.size() * 32) == 0);
    d.coset_fft(worker); d.icoset_fft(worker);
    CHECK(std::memcmp(d.as_ref().data(), a.data(), a.size() * 32) == 0);
  }
  {  // from_coeffs pads ragged input with zeros (domain.rs:89)
    std::vector<Fr> a(5);
    for (auto& x : a) x = rand_scalar(gen);
    EvaluationDomain d = EvaluationDomain::from_coeffs(a);
    CHECK(d.exp() == 3 && d.as_ref().size() == 8 && d.as_ref()[7] == (Fr{0, 0, 0, 0}));
  }
  std::puts("ok evaluation_domain");
  return 0;
}
