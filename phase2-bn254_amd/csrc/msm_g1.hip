// G1 instantiation of the Pippenger pipeline (msm_impl.hpp); see there for the design.
#include "msm_impl.hpp"

namespace zk {

int msm_g1_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[12], long long* err_index) {
  G1Jacobian r;
  int rc = msm_device<Fq>((const G1Affine*)d_bases, n_bases, base_offset, (const uint32_t*)d_scalars, n, d_density, d_dprefix, st, &r, err_index);
  if (rc == ZK_OK) std::memcpy(out_xyz, &r, sizeof r);
  return rc;
}

void msm_geometry(uint64_t n, uint32_t* c, uint32_t* W) {
  MsmGeom G = choose_geom(n, 1);
  *c = G.c;
  *W = G.W;
}

void msm_release_g1() { ws_release_all(); }

}  // namespace zk
