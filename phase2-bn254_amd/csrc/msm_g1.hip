// G1 instantiation of the Pippenger pipeline (msm_impl.hpp); see there for the design.
#define ZK_CHAIN_MAD 1  // fieldu.hpp u_mad: one dependent mad chain per column (measured faster in this TU)
#include "msm_impl.hpp"

namespace zk {

std::atomic<int> g_msm_inflight[16];

int msm_g1_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[12], long long* err_index, uint32_t wgroups, uint32_t wgroup,
                  bool scalars_mont, MsmChunks* chunks, uint64_t table_stride, uint32_t table_c) {
  G1Jacobian r;
  int rc = msm_device<Fq>((const G1Affine*)d_bases, n_bases, base_offset, (const uint32_t*)d_scalars, n, d_density, d_dprefix, st, &r, err_index,
                          false, nullptr, nullptr, wgroups, wgroup, scalars_mont, chunks, table_stride, table_c);
  if (rc == ZK_OK) std::memcpy(out_xyz, &r, sizeof r);
  return rc;
}

void msm_geometry(uint64_t n, uint32_t wgroups, uint32_t* c, uint32_t* W) {
  MsmGeom G = choose_geom(n, 1, wgroups ? wgroups : 1);
  *c = G.c;
  *W = G.W;
}

// table mode (msm_device): the window layout a table of n_bases points is built for -- width[w] bits per window, window w scaled by
// 2^(width[0] + .. + width[w-1])
void msm_table_geometry(uint64_t n_bases, int group, uint32_t* c, uint32_t* W, uint8_t width[64]) {
  const MsmGeom G = make_geom(table_window_bits(n_bases, group));
  *c = G.c;
  *W = G.W;
  if (width) for (uint32_t w = 0; w < 64; ++w) width[w] = w < G.W ? G.width[w] : 0;
}

// powersoftau::utils::dense_multiexp (powersoftau/src/utils.rs:189-292): bases.len() == exponents.len(), infinity
// bases add nothing; with d_bases2 the merge_pairs form (utils.rs:112-128, phase2/src/utils.rs:59-105): both
// base vectors against ONE exponent vector, sharing digit extraction and sorts.
int msm_g1_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz,
                         uint64_t* out2_xyz) {
  G1Jacobian r, r2;
  long long err = -1;
  int rc = msm_device<Fq>((const G1Affine*)d_bases, n, 0, (const uint32_t*)d_scalars, n, nullptr, nullptr, st, &r, &err, true,
                          (const G1Affine*)d_bases2, d_bases2 ? &r2 : nullptr);
  if (rc == ZK_OK) {
    std::memcpy(out_xyz, &r, sizeof r);
    if (d_bases2 && out2_xyz) std::memcpy(out2_xyz, &r2, sizeof r2);
  }
  return rc;
}

int segsum_g1_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out) {
  return segsum_device<Fq>((const G1Affine*)d_points, nnz, d_row_ptr, n_rows, st, (G1Affine*)d_out);
}

// host self-test hook for the digit extraction (tests/test_msm_digits_host.py): the signed digits of one scalar for the geometry
// chosen for (n, wgroups), windows [w_start, w_stop) -- direct != 0: the way a window-group rank computes them (chain started one
// window below w_start, fallback on the sign boundary), direct == 0: the plain chain from window 0.  digits[w] = signed digit
// (0 = no bucket) for the windows produced, INT32_MIN elsewhere; geom = {c, W, nb, rmul, rshift, width[0..W), shift[0..W)}.
int msm_selftest_digits(uint64_t n, uint32_t wgroups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct, int32_t* digits,
                        uint32_t* geom) {
  const MsmGeom G = choose_geom(n, 1, wgroups ? wgroups : 1);
  if (G.W == 0) return ZK_ERR_BAD_ARGS;
  if (geom) {
    geom[0] = G.c; geom[1] = G.W; geom[2] = G.nb; geom[3] = G.rmul; geom[4] = G.rshift;
    for (uint32_t w = 0; w < G.W; ++w) { geom[5 + w] = G.width[w]; geom[5 + G.W + w] = G.shift[w]; }
  }
  if (!digits || !scalar) return ZK_OK;
  if (w_stop > G.W) w_stop = G.W;
  uint32_t s[9];
  for (int l = 0; l < 8; ++l) s[l] = scalar[l];
  s[8] = 0;
  for (uint32_t w = 0; w < G.W; ++w) digits[w] = INT32_MIN;
  auto emit = [&](uint32_t w, uint32_t d, uint32_t neg) { digits[w] = neg ? -(int32_t)d : (int32_t)d; };
  auto from0 = [&](uint32_t w, uint32_t d, uint32_t neg) { if (w >= w_start) emit(w, d, neg); };
  if (direct) { ZK_DISPATCH_RMUL(G.rmul, msm_scalar_digits<RM>(s, G, w_start, w_stop, emit)); }
  else { ZK_DISPATCH_RMUL(G.rmul, (void)msm_scalar_digits_from<RM>(s, G, 0, w_stop, from0)); }
  return ZK_OK;
}

void msm_release_g1() { ws_release_all(); }

}  // namespace zk
