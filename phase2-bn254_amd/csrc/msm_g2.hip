// G2 instantiation of the Pippenger pipeline (msm_impl.hpp); see there for the design.
#include "msm_impl.hpp"

namespace zk {

int msm_g2_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[24], long long* err_index, uint32_t wgroups, uint32_t wgroup,
                  bool scalars_mont, MsmChunks* chunks, uint64_t table_stride, uint32_t table_c) {
  G2Jacobian r;
  int rc = msm_device<Fq2>((const G2Affine*)d_bases, n_bases, base_offset, (const uint32_t*)d_scalars, n, d_density, d_dprefix, st, &r, err_index,
                          false, nullptr, nullptr, wgroups, wgroup, scalars_mont, chunks, table_stride, table_c);
  if (rc == ZK_OK) std::memcpy(out_xyz, &r, sizeof r);
  return rc;
}

// powersoftau::utils::dense_multiexp (powersoftau/src/utils.rs:189-292): bases.len() == exponents.len(), infinity
// bases add nothing; with d_bases2 the merge_pairs form (utils.rs:112-128, phase2/src/utils.rs:59-105): both
// base vectors against ONE exponent vector, sharing digit extraction and sorts.
int msm_g2_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz,
                         uint64_t* out2_xyz) {
  G2Jacobian r, r2;
  long long err = -1;
  int rc = msm_device<Fq2>((const G2Affine*)d_bases, n, 0, (const uint32_t*)d_scalars, n, nullptr, nullptr, st, &r, &err, true,
                          (const G2Affine*)d_bases2, d_bases2 ? &r2 : nullptr);
  if (rc == ZK_OK) {
    std::memcpy(out_xyz, &r, sizeof r);
    if (d_bases2 && out2_xyz) std::memcpy(out2_xyz, &r2, sizeof r2);
  }
  return rc;
}

int segsum_g2_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out) {
  return segsum_device<Fq2>((const G2Affine*)d_points, nnz, d_row_ptr, n_rows, st, (G2Affine*)d_out);
}

void msm_release_g2() { ws_release_all(); }

}  // namespace zk
