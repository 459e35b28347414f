// Internal interfaces between the translation units of libmi355zk.so (not part of the C ABI: include/mi355zk.h is).
//   api.hip         the C ABI: argument checking, domain constants, device-resident entry points, profiling hooks, lifecycle
//   scalar_mul.hip  the scalar-multiplication kernels and their launchers: batch_exp / batch_mul / window-table build / G2 membership
//   (msm_g1.hip, msm_g2.hip, ntt.hip, point_fft*.hip, codec.hip, field_ops.hip: the kernels behind the functions declared below)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "curveu.hpp"
#include "device_util.hpp"

namespace zk {
// ntt.hip
int ntt_run(Fr* d_a, uint32_t log_n, const Fr& omega, hipStream_t st);
int ntt_run_scaled(Fr* d_a, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st);
int ntt_run_batch(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, const Fr& omega, const Fr* pre_g, const Fr* post_c, const Fr* post_g, hipStream_t st);
int ntt_scale(Fr* d_a, uint32_t log_n, const Fr& c, const Fr* g, hipStream_t st);
void ntt_release_all();
int ntt_configure();
// msm.hip
int msm_g1_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[12], long long* err_index, uint32_t wgroups, uint32_t wgroup, bool scalars_mont,
                  MsmChunks* chunks, uint64_t table_stride = 0, uint32_t table_c = 0);
void msm_table_geometry(uint64_t n_bases, int group, uint32_t* c, uint32_t* W, uint8_t width[64]);
int msm_g2_device(const void* d_bases, uint64_t n_bases, uint64_t base_offset, const void* d_scalars, uint64_t n, const uint32_t* d_density,
                  const uint32_t* d_dprefix, hipStream_t st, uint64_t out_xyz[24], long long* err_index, uint32_t wgroups, uint32_t wgroup, bool scalars_mont,
                  MsmChunks* chunks, uint64_t table_stride = 0, uint32_t table_c = 0);
int msm_g1_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz, uint64_t* out2_xyz);
int msm_g2_dense_device(const void* d_bases, const void* d_bases2, const void* d_scalars, uint64_t n, hipStream_t st, uint64_t* out_xyz, uint64_t* out2_xyz);
// point_fft.hip
int point_fft_g1(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st);
// codec.hip
int codec_decode(int group, void* d_out, const void* d_in, size_t n, int compressed, int checked, hipStream_t st, long long* err_index);
int codec_encode(int group, void* d_out, const void* d_in, size_t n, int compressed, hipStream_t st);
// point_fft_g2.hip
int point_fft_g2(void* d_points, uint32_t log_n, const Fr& omega, bool scale, const Fr& scale_canon, hipStream_t st, bool trusted_subgroup);
int segsum_g1_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out);
int segsum_g2_device(const void* d_points, uint64_t nnz, const uint32_t* d_row_ptr, uint32_t n_rows, hipStream_t st, void* d_out);
void msm_release_g1();
void msm_release_g2();
void msm_geometry(uint64_t n, uint32_t wgroups, uint32_t* c, uint32_t* W);
int msm_selftest_digits(uint64_t n, uint32_t wgroups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct, int32_t* digits,
                        uint32_t* geom);

// scalar_mul.hip
// out[i] = k[i or 0] * P[i or base_index[i] or 0], affine (batched_accumulator.rs:1130-1181, parameters.rs:423-470); see the definition for the modes
template <class F>
int batch_exp(void* d_out, const void* d_bases, int same_base, const void* d_scalars, int same_scalar, size_t n, void* stream,
              const uint32_t* d_base_index = nullptr, bool shortcut_unit_scalars = false, bool g2_trusted = false,
              const uint8_t* d_g2_member = nullptr);
extern template int batch_exp<Fq>(void*, const void*, int, const void*, int, size_t, void*, const uint32_t*, bool, bool, const uint8_t*);
extern template int batch_exp<Fq2>(void*, const void*, int, const void*, int, size_t, void*, const uint32_t*, bool, bool, const uint8_t*);
template <class F>
int batch_mul(void* d_out, const uint64_t* base_raw, const void* d_scalars, size_t n, void* stream);
extern template int batch_mul<Fq>(void*, const uint64_t*, const void*, size_t, void*);
extern template int batch_mul<Fq2>(void*, const uint64_t*, const void*, size_t, void*);
template <int GROUP>
int msm_table_build(const void* d_bases, size_t n, void* d_table, size_t table_bytes, void* stream);
extern template int msm_table_build<1>(const void*, size_t, void*, size_t, void*);
extern template int msm_table_build<2>(const void*, size_t, void*, size_t, void*);
int g2_subgroup_flags(const void* d_points, size_t n, void* stream, uint8_t* d_member);
int g2_subgroup_check(const void* d_points, size_t n, void* stream, long long* bad_index);
bool g2_in_subgroup_host(const Affine<Fq2>& p);   // the membership test of the kernels, run on the host (self-test hook)
void exp_scratch_release_all();
void mul_slots_release_all();

// api.hip
int domain_op_dev(Fr* d_a, uint32_t log_n, int op, hipStream_t st);   // EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} on a device array

// host_entry.hip
struct DeviceGuard {  // the calling thread's current device is its own business: restore it
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
extern thread_local long long t_last_err_index;   // exponent index of the last Source error of this thread's calls (mi355zk_last_error_index)
// the device-resident multiexp behind every _dev entry point (table: d_bases is a window table; chunks: exponents handed over while the call runs)
template <int GROUP>
int msm_dev_entry(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars, const uint32_t* density,
                  size_t density_bits, void* stream, uint64_t* out_xyz, uint32_t wgroups = 1, uint32_t wgroup = 0, uint32_t flags = 0,
                  MsmChunks* chunks = nullptr, bool table = false);
// host buffers: whole on one device (streamed upload, pinned-bases cache) or cut into cells over the device set of mi355zk_init
template <int GROUP>
int msm_host_entry(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars, const uint32_t* density,
                   size_t density_bits, uint64_t* out_xyz);
template <class F>
int batch_exp_host(uint8_t* out, const uint8_t* bases, const uint64_t* scalars, size_t n, int same_scalar, bool g2_trusted);
template <int GROUP>
int dense_host(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t* out_s, uint64_t* out_sx);
int ntt_host(uint64_t* a, uint32_t log_n, int op, const uint64_t* omega);
template <class F>
int sparse_matvec(void* d_out, const void* d_bases, size_t n_bases, const uint32_t* d_row_ptr, const uint32_t* d_col, const void* d_coeffs,
                  size_t n_rows, size_t nnz, void* stream, int group, bool g2_trusted, void* d_scratch = nullptr, size_t scratch_bytes = 0);
template <class F>
int sparse_matvec_host(uint8_t* out, const uint8_t* bases, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col, const uint64_t* coeffs,
                       size_t n_rows, size_t nnz, int group, bool g2_trusted);
extern template int msm_dev_entry<1>(const void*, size_t, size_t, const void*, size_t, const uint32_t*, size_t, void*, uint64_t*, uint32_t, uint32_t, uint32_t, MsmChunks*, bool);
extern template int msm_dev_entry<2>(const void*, size_t, size_t, const void*, size_t, const uint32_t*, size_t, void*, uint64_t*, uint32_t, uint32_t, uint32_t, MsmChunks*, bool);
extern template int msm_host_entry<1>(const uint8_t*, size_t, size_t, const uint64_t*, size_t, const uint32_t*, size_t, uint64_t*);
extern template int msm_host_entry<2>(const uint8_t*, size_t, size_t, const uint64_t*, size_t, const uint32_t*, size_t, uint64_t*);
extern template int batch_exp_host<Fq>(uint8_t*, const uint8_t*, const uint64_t*, size_t, int, bool);
extern template int batch_exp_host<Fq2>(uint8_t*, const uint8_t*, const uint64_t*, size_t, int, bool);
extern template int dense_host<1>(const uint8_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*, uint64_t*);
extern template int dense_host<2>(const uint8_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*, uint64_t*);
extern template int sparse_matvec<Fq>(void*, const void*, size_t, const uint32_t*, const uint32_t*, const void*, size_t, size_t, void*, int, bool, void*, size_t);
extern template int sparse_matvec<Fq2>(void*, const void*, size_t, const uint32_t*, const uint32_t*, const void*, size_t, size_t, void*, int, bool, void*, size_t);
extern template int sparse_matvec_host<Fq>(uint8_t*, const uint8_t*, size_t, const uint32_t*, const uint32_t*, const uint64_t*, size_t, size_t, int, bool);
extern template int sparse_matvec_host<Fq2>(uint8_t*, const uint8_t*, size_t, const uint32_t*, const uint32_t*, const uint64_t*, size_t, size_t, int, bool);
int bases_cache_pin(const void* host, size_t n, int group, bool tables = false);   // the caller's promise of immutability (include/mi355zk.h)
void bases_cache_invalidate(const void* host);
int bases_cache_info(const void* host_bases, size_t* device_bytes, size_t* table_bytes);
void devset_set(const std::vector<int>& set);   // the device set of mi355zk_init (empty: the current device)
int devset_count();
void host_entry_release_all();
}  // namespace zk
