// C++ twin of the reference's own tests for this path, written against host/bellman.hpp (the product) with
// the CPU oracle as the checker (TEST INFRASTRUCTURE: links oracle/_build/liboracle.so).
//   bellman/src/multiexp.rs:479-518   naive sum == multiexp           (test_with_bls12, restated on BN254)
//   bellman/src/domain.rs:427-463     ifft(fft(a)) == a, coset variants
//   bellman/src/source.rs:44-70       Source errors through the future
// Run by tests/test_gpu_cpp_host.py on the GPU box; prints "ok <name>" lines and exits 0 on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../phase2-bn254_amd/host/bellman.hpp"
#include "../../phase2-bn254_amd/host/ceremony.hpp"
#include "../../phase2-bn254_amd/host/prover.hpp"

extern "C" {
void oracle_g1_mul_many_affine(uint64_t* out_affine, const uint64_t base_affine[8], const uint64_t* ks, size_t n);
void oracle_g1_naive_multiexp(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]);
void oracle_g1_to_affine(uint64_t r[8], const uint64_t p[12]);
int oracle_fr_domain_op(uint64_t* a, uint32_t log_n, int op, uint32_t log_cpus);
void oracle_fe_from_canonical(int which, uint64_t r[4], const uint64_t a[4]);
void oracle_fe_mul(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
void oracle_fe_sub(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
void oracle_g1_mul(uint64_t p[12], const uint64_t k[4]);
void oracle_g1_from_affine(uint64_t r[12], const uint64_t p[8]);
void oracle_g1_add(uint64_t p[12], const uint64_t o[12]);
void oracle_g1_encode(uint8_t* out, const uint64_t* affine, size_t n, int compressed);
void oracle_g2_mul_many_affine(uint64_t* out_affine, const uint64_t base_affine[16], const uint64_t* ks, size_t n);
void oracle_g2_naive_multiexp(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[24]);
void oracle_g2_to_affine(uint64_t r[16], const uint64_t p[24]);
void oracle_g2_from_affine(uint64_t r[24], const uint64_t p[16]);
void oracle_g2_mul(uint64_t p[24], const uint64_t k[4]);
void oracle_g2_add(uint64_t p[24], const uint64_t o[24]);
void oracle_fe_to_canonical(int which, uint64_t r[4], const uint64_t a[4]);
void oracle_fe_inv(int which, uint64_t r[4], const uint64_t a[4]);
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

    // the single-process multi-GPU mode: the same calls cut into cells over a device set (here four logical devices on GPU 0);
    // same affine point, same SynthesisError and exponent index as the one-device call
    setenv("MI355ZK_MULTI_MIN_LOG", "6", 1);
    {
      Worker four(std::vector<int>{0, 0, 0, 0});
      CHECK(mi355zk_device_count() == 4);
      G1Projective cut = multiexp<G1Affine>(four, {bases, 0}, FullDensity{}, exps).get();
      oracle_g1_to_affine(a, reinterpret_cast<const uint64_t*>(&cut));
      CHECK(std::memcmp(a, b, 64) == 0);
      G1Projective sparse4 = multiexp<G1Affine>(four, {bases, 0}, d, exps).get();
      uint64_t s1[8], s4[8];
      oracle_g1_to_affine(s1, reinterpret_cast<const uint64_t*>(&sparse));
      oracle_g1_to_affine(s4, reinterpret_cast<const uint64_t*>(&sparse4));
      CHECK(std::memcmp(s1, s4, 64) == 0);
      auto holed = std::make_shared<std::vector<G1Affine>>(*bases);
      (*holed)[1300] = G1Affine{};
      (*holed)[700] = G1Affine{};
      try { multiexp<G1Affine>(four, {holed, 0}, FullDensity{}, exps).get(); CHECK(false); }
      catch (const SynthesisError& e) { CHECK(e.kind == SynthesisError::UnexpectedIdentity && e.index == 700); }
      auto fewer = std::make_shared<std::vector<G1Affine>>(bases->begin(), bases->begin() + 1000);
      try { multiexp<G1Affine>(four, {fewer, 0}, FullDensity{}, exps).get(); CHECK(false); }
      catch (const SynthesisError& e) { CHECK(e.kind == SynthesisError::IoErrorUnexpectedEof && e.index == 1000); }
    }
    unsetenv("MI355ZK_MULTI_MIN_LOG");
    Worker back(0);
    CHECK(mi355zk_device_count() == 1);
    std::puts("ok multi_device_cells");
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
  {  // the H pipeline's elementwise steps (prover.rs:217-241) and scalars_into_representations against oracle field arithmetic
    const uint32_t log_n = 9;
    std::vector<Fr> a((size_t)1 << log_n), b(a.size()), c(a.size());
    for (auto& x : a) x = rand_scalar(gen);
    for (auto& x : b) x = rand_scalar(gen);
    for (auto& x : c) x = rand_scalar(gen);
    EvaluationDomain da = EvaluationDomain::from_coeffs(a), db = EvaluationDomain::from_coeffs(b), dc = EvaluationDomain::from_coeffs(c);
    da.mul_assign(worker, db);
    da.sub_assign(worker, dc);
    da.divide_by_z_on_coset(worker);
    // want[i] = (a[i] b[i] - c[i]) * z(g)^-1 with g = 7:  z(g) * want[i] == a[i] b[i] - c[i]
    uint64_t seven_c[4] = {7, 0, 0, 0}, seven[4];
    oracle_fe_from_canonical(1, seven, seven_c);
    Fr g{seven[0], seven[1], seven[2], seven[3]};
    const Fr zg = da.z(g);
    for (size_t i = 0; i < a.size(); i += 37) {
      uint64_t ab[4], lhs[4], rhs[4];
      oracle_fe_mul(1, ab, a[i].data(), b[i].data());
      oracle_fe_sub(1, rhs, ab, c[i].data());
      oracle_fe_mul(1, lhs, da.as_ref()[i].data(), zg.data());
      CHECK(std::memcmp(lhs, rhs, 32) == 0);
    }
    const std::vector<FrRepr> reps = EvaluationDomain::from_coeffs(a).into_representations();
    for (size_t i = 0; i < a.size(); i += 41) {
      uint64_t back[4];
      oracle_fe_from_canonical(1, back, reps[i].data());
      CHECK(std::memcmp(back, a[i].data(), 32) == 0);
    }
  }
  std::puts("ok evaluation_domain");

  {  // ceremony.hpp: the reference's test_power_pairs (powersoftau/src/utils.rs:90-109) with the known x instead of the pairing
    const size_t n = 100;
    FrRepr x = rand_scalar(gen);
    std::vector<FrRepr> ones(n, FrRepr{1, 0, 0, 0});
    std::vector<G1Affine> gens(n);
    for (auto& p : gens) std::memcpy(&p, g1, 64);
    // v[i] = x^i * G by repeated batch_exp with the shared scalar x:  v[0] = G, v[i+1] = x * v[i]
    std::vector<G1Affine> v(1), cur(1);
    std::memcpy(&v[0], g1, 64);
    cur = v;
    for (size_t i = 1; i < n; ++i) { cur = ceremony::batch_exp(cur, x); v.push_back(cur[0]); }
    std::vector<FrRepr> rho(n - 1);
    for (auto& r : rho) r = rand_scalar(gen);
    auto [s, sx] = ceremony::power_pairs(v, rho);
    uint64_t xs[12], a[8], b[8];
    std::memcpy(xs, &s, 96);
    oracle_g1_mul(xs, x.data());
    oracle_g1_to_affine(a, xs);
    oracle_g1_to_affine(b, reinterpret_cast<const uint64_t*>(&sx));
    CHECK(std::memcmp(a, b, 64) == 0);                       // same_ratio(power_pairs(v), (g2, g2^x))
    // batch_exp with per-point exponents against the oracle
    std::vector<FrRepr> es(8);
    for (auto& e : es) e = rand_scalar(gen);
    std::vector<G1Affine> first8(v.begin(), v.begin() + 8);
    auto scaled = ceremony::batch_exp(first8, es);
    for (size_t i = 0; i < 8; ++i) {
      uint64_t j[12], want[8];
      oracle_g1_from_affine(j, reinterpret_cast<const uint64_t*>(&first8[i]));
      oracle_g1_mul(j, es[i].data());
      oracle_g1_to_affine(want, j);
      CHECK(std::memcmp(&scaled[i], want, 64) == 0);
    }
    // eval_qap on a 2-row CSR: row 0 = es[0]*v[1] + es[1]*v[3], row 1 empty
    auto rows = ceremony::eval_qap<G1Affine>(v, {0, 2, 2}, {1, 3}, {es[0], es[1]});
    uint64_t t0[12], t1[12], want[8];
    oracle_g1_from_affine(t0, reinterpret_cast<const uint64_t*>(&v[1])); oracle_g1_mul(t0, es[0].data());
    oracle_g1_from_affine(t1, reinterpret_cast<const uint64_t*>(&v[3])); oracle_g1_mul(t1, es[1].data());
    oracle_g1_add(t0, t1);
    oracle_g1_to_affine(want, t0);
    CHECK(std::memcmp(&rows[0], want, 64) == 0 && rows[1].is_zero());
    // codecs: bytes equal the oracle's, round trip, and the reference's error for a corrupted record
    auto enc = ceremony::encode_points(first8, true);
    std::vector<uint8_t> want_enc(8 * 32);
    oracle_g1_encode(want_enc.data(), reinterpret_cast<const uint64_t*>(first8.data()), 8, 1);
    CHECK(enc == want_enc);
    auto back = ceremony::decode_points<G1Affine>(enc, true);
    CHECK(std::memcmp(back.data(), first8.data(), 8 * 64) == 0);
    enc[5 * 32] = 0x7f;
    try { ceremony::decode_points<G1Affine>(enc, true); CHECK(false); }
    catch (const ceremony::GroupDecodingError& e) { CHECK(e.kind == ceremony::GroupDecodingError::UnexpectedInformation && e.index == 5); }
    // point fft round trip
    std::vector<G1Affine> pts(v.begin(), v.begin() + 16), orig = pts;
    ceremony::point_ifft(pts);
    CHECK(std::memcmp(pts.data(), orig.data(), 16 * 64) != 0);
    ceremony::point_fft(pts, false);
    CHECK(std::memcmp(pts.data(), orig.data(), 16 * 64) == 0);
  }
  std::puts("ok ceremony_mirror");

  {  // host/prover.hpp: groth16 create_proof (prover.rs:202-343) on a synthetic 2^8-constraint instance, against a proof assembled
     // from the oracle's domain ops, field arithmetic, naive multiexps and group law
    using namespace bellman::groth16;
    static const uint64_t G2_GEN[16] = {0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL,
                                        0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL,
                                        0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
                                        0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
    const uint32_t log_m = 8;
    const size_t m = (size_t)1 << log_m, num_inputs = 5, num_aux = m - 11;
    auto g1_pts = [&](size_t n) {
      std::vector<FrRepr> ks(n);
      for (auto& k : ks) k = rand_scalar(gen);
      auto v = std::make_shared<std::vector<G1Affine>>(n);
      oracle_g1_mul_many_affine(reinterpret_cast<uint64_t*>(v->data()), g1, reinterpret_cast<const uint64_t*>(ks.data()), n);
      return v;
    };
    auto g2_pts = [&](size_t n) {
      std::vector<FrRepr> ks(n);
      for (auto& k : ks) k = rand_scalar(gen);
      auto v = std::make_shared<std::vector<G2Affine>>(n);
      oracle_g2_mul_many_affine(reinterpret_cast<uint64_t*>(v->data()), G2_GEN, reinterpret_cast<const uint64_t*>(ks.data()), n);
      return v;
    };
    // witness-like canonical assignments (zeros, ones, random), kept in Montgomery form like the prover's Vec<Fr>
    auto witness = [&](size_t n, std::vector<FrRepr>& canon) {
      std::vector<Fr> mont(n);
      canon.resize(n);
      for (size_t i = 0; i < n; ++i) {
        const uint64_t kind = gen() % 10;
        canon[i] = kind < 2 ? FrRepr{0, 0, 0, 0} : kind < 5 ? FrRepr{1, 0, 0, 0} : rand_scalar(gen);
        oracle_fe_from_canonical(1, mont[i].data(), canon[i].data());
      }
      return mont;
    };
    ProvingAssignment pa;
    std::vector<FrRepr> inp_c, aux_c;
    pa.input_assignment = witness(num_inputs, inp_c);
    pa.aux_assignment = witness(num_aux, aux_c);
    std::vector<char> a_bits(num_aux), bi_bits(num_inputs), ba_bits(num_aux);
    for (size_t i = 0; i < num_aux; ++i) { pa.a_aux_density.add_element(); if ((a_bits[i] = gen() & 1)) pa.a_aux_density.inc(i); }
    for (size_t i = 0; i < num_inputs; ++i) { pa.b_input_density.add_element(); if ((bi_bits[i] = gen() & 1)) pa.b_input_density.inc(i); }
    for (size_t i = 0; i < num_aux; ++i) { pa.b_aux_density.add_element(); if ((ba_bits[i] = (gen() % 5) < 2)) pa.b_aux_density.inc(i); }
    const size_t na = pa.a_aux_density.get_total_density(), nbi = pa.b_input_density.get_total_density(), nba = pa.b_aux_density.get_total_density();
    std::vector<Fr> ev[3];
    for (auto& v : ev) { v.resize(m); for (auto& x : v) x = rand_scalar(gen); }   // any value < r is a valid Montgomery element
    pa.a = ev[0]; pa.b = ev[1]; pa.c = ev[2];
    Parameters params;
    params.h = g1_pts(m - 1); params.l = g1_pts(num_aux); params.a = g1_pts(num_inputs + na); params.b_g1 = g1_pts(nbi + nba); params.b_g2 = g2_pts(nbi + nba);
    {
      auto v1 = g1_pts(3); auto v2 = g2_pts(2);
      params.vk = VerifyingKey{(*v1)[0], (*v1)[1], (*v1)[2], (*v2)[0], (*v2)[1]};
    }
    const FrRepr r = rand_scalar(gen), s = rand_scalar(gen);
    params.pin();  // the shim's promise: these Arc<Vec<G>> are immutable -- the second proof below finds the bases on the device
    const Proof proof = create_proof(worker, params, pa, r, s);

    // ---- the same proof from the oracle
    for (auto& v : ev) { CHECK(oracle_fr_domain_op(reinterpret_cast<uint64_t*>(v.data()), log_m, 1, 0) == 0); CHECK(oracle_fr_domain_op(reinterpret_cast<uint64_t*>(v.data()), log_m, 2, 0) == 0); }
    uint64_t zg[4], zinv[4];
    {
      uint64_t seven_c[4] = {7, 0, 0, 0}, seven[4], one_c[4] = {1, 0, 0, 0}, one[4], acc[4];
      oracle_fe_from_canonical(1, seven, seven_c);
      oracle_fe_from_canonical(1, one, one_c);
      std::memcpy(acc, seven, 32);
      for (uint32_t i = 0; i < log_m; ++i) oracle_fe_mul(1, acc, acc, acc);   // 7^m
      oracle_fe_sub(1, zg, acc, one);
      oracle_fe_inv(1, zinv, zg);
    }
    std::vector<Fr> hq(m);
    for (size_t i = 0; i < m; ++i) {
      uint64_t ab[4], d[4];
      oracle_fe_mul(1, ab, ev[0][i].data(), ev[1][i].data());
      oracle_fe_sub(1, d, ab, ev[2][i].data());
      oracle_fe_mul(1, hq[i].data(), d, zinv);
    }
    CHECK(oracle_fr_domain_op(reinterpret_cast<uint64_t*>(hq.data()), log_m, 3, 0) == 0);
    std::vector<FrRepr> h_c(m - 1);
    for (size_t i = 0; i + 1 < m; ++i) oracle_fe_to_canonical(1, h_c[i].data(), hq[i].data());
    auto sum1 = [&](const std::vector<G1Affine>& bases, size_t off, const std::vector<FrRepr>& e, const std::vector<char>* bits, uint64_t out[12]) {
      std::vector<G1Affine> b; std::vector<FrRepr> k;
      size_t next = off;
      for (size_t i = 0; i < e.size(); ++i) if (!bits || (*bits)[i]) { b.push_back(bases[next++]); k.push_back(e[i]); }
      oracle_g1_naive_multiexp(reinterpret_cast<const uint64_t*>(b.data()), reinterpret_cast<const uint64_t*>(k.data()), b.size(), out);
    };
    auto sum2 = [&](const std::vector<G2Affine>& bases, size_t off, const std::vector<FrRepr>& e, const std::vector<char>* bits, uint64_t out[24]) {
      std::vector<G2Affine> b; std::vector<FrRepr> k;
      size_t next = off;
      for (size_t i = 0; i < e.size(); ++i) if (!bits || (*bits)[i]) { b.push_back(bases[next++]); k.push_back(e[i]); }
      oracle_g2_naive_multiexp(reinterpret_cast<const uint64_t*>(b.data()), reinterpret_cast<const uint64_t*>(k.data()), b.size(), out);
    };
    uint64_t h_o[12], l_o[12], a_in[12], a_ax[12], b1_in[12], b1_ax[12], b2_in[24], b2_ax[24];
    sum1(*params.h, 0, h_c, nullptr, h_o);
    sum1(*params.l, 0, aux_c, nullptr, l_o);
    sum1(*params.a, 0, inp_c, nullptr, a_in);
    sum1(*params.a, num_inputs, aux_c, &a_bits, a_ax);
    sum1(*params.b_g1, 0, inp_c, &bi_bits, b1_in);
    sum1(*params.b_g1, nbi, aux_c, &ba_bits, b1_ax);
    sum2(*params.b_g2, 0, inp_c, &bi_bits, b2_in);
    sum2(*params.b_g2, nbi, aux_c, &ba_bits, b2_ax);
    auto mul1 = [&](const G1Affine& p, const FrRepr& k, uint64_t out[12]) { oracle_g1_from_affine(out, reinterpret_cast<const uint64_t*>(&p)); oracle_g1_mul(out, k.data()); };
    uint64_t g_a[12], g_b[24], g_c[12], t[12], t2[24];
    mul1(params.vk.delta_g1, r, g_a);
    oracle_g1_from_affine(t, reinterpret_cast<const uint64_t*>(&params.vk.alpha_g1)); oracle_g1_add(g_a, t);
    oracle_g2_from_affine(g_b, reinterpret_cast<const uint64_t*>(&params.vk.delta_g2)); oracle_g2_mul(g_b, s.data());
    oracle_g2_from_affine(t2, reinterpret_cast<const uint64_t*>(&params.vk.beta_g2)); oracle_g2_add(g_b, t2);
    mul1(params.vk.delta_g1, r, g_c); oracle_g1_mul(g_c, s.data());
    mul1(params.vk.alpha_g1, s, t); oracle_g1_add(g_c, t);
    mul1(params.vk.beta_g1, r, t); oracle_g1_add(g_c, t);
    oracle_g1_add(a_in, a_ax); oracle_g1_add(g_a, a_in); oracle_g1_mul(a_in, s.data()); oracle_g1_add(g_c, a_in);
    oracle_g1_add(b1_in, b1_ax); oracle_g2_add(b2_in, b2_ax); oracle_g2_add(g_b, b2_in);
    oracle_g1_mul(b1_in, r.data()); oracle_g1_add(g_c, b1_in);
    oracle_g1_add(g_c, h_o); oracle_g1_add(g_c, l_o);
    uint64_t wa[8], wb[16], wc[8];
    oracle_g1_to_affine(wa, g_a); oracle_g2_to_affine(wb, g_b); oracle_g1_to_affine(wc, g_c);
    CHECK(std::memcmp(wa, &proof.a, 64) == 0);
    CHECK(std::memcmp(wb, &proof.b, 128) == 0);
    CHECK(std::memcmp(wc, &proof.c, 64) == 0);
    // a second proof on the same parameters (bases served from the device cache) is the same proof
    const Proof again = create_proof(worker, params, pa, r, s);
    CHECK(std::memcmp(&again, &proof, sizeof proof) == 0);
    // pinned WITH TABLES: the next proof builds the window tables of the cached vectors and runs its multiexps in table mode, the one
    // after that finds them -- the same proof every time
    params.pin(true);
    const Proof third = create_proof(worker, params, pa, r, s);
    const Proof fourth = create_proof(worker, params, pa, r, s);
    size_t dev_bytes = 0, tab_bytes = 0;
    CHECK(mi355zk_bases_cache_info(params.h->data(), &dev_bytes, &tab_bytes) == 1 && tab_bytes > dev_bytes);
    params.unpin();
    CHECK(mi355zk_bases_cache_info(params.h->data(), &dev_bytes, &tab_bytes) == 0);
    CHECK(std::memcmp(&third, &proof, sizeof proof) == 0);
    CHECK(std::memcmp(&fourth, &proof, sizeof proof) == 0);
  }
  std::puts("ok groth16_create_proof");
  return 0;
}
