// G2 instantiation of the Pippenger pipeline (msm_impl.hpp); see there for the design.
#include "msm_impl.hpp"

namespace zk {

int msm_g2_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[24], long long* err_index) {
  G2Jacobian r;
  int rc = msm_device<Fq2>((const G2Affine*)d_bases, n_bases, base_offset, (const uint32_t*)d_scalars, n, d_density, d_dprefix, st, &r, err_index);
  if (rc == ZK_OK) std::memcpy(out_xyz, &r, sizeof r);
  return rc;
}

void msm_release_g2() { ws_release_all(); }

}  // namespace zk
