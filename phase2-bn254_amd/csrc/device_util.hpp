// Shared host-side helpers for the HIP translation units of libmi355zk.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>

// Return codes of the C ABI (include/mi355zk.h).
#define ZK_OK 0
#define ZK_ERR_UNEXPECTED_IDENTITY 1  // bellman/src/source.rs:50-52
#define ZK_ERR_UNEXPECTED_EOF 2       // bellman/src/source.rs:46-48,62-64
#define ZK_ERR_BAD_ARGS 3
#define ZK_ERR_DEVICE (-1)            // any HIP failure; message on stderr

#define ZK_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t zk_e_ = (expr);                                                                    \
    if (zk_e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "[mi355zk] HIP error %d (%s) at %s:%d: %s\n", (int)zk_e_,              \
                   hipGetErrorString(zk_e_), __FILE__, __LINE__, #expr);                          \
      return ZK_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)

namespace zk {

// A multiexp whose exponents arrive in CHUNKS: the host-buffer entry point (api.hip: msm_host_entry) uploads them over PCIe while
// the kernels of the earlier chunks run.  msm_device (msm_impl.hpp) evaluates chunk c = exponents [cuts[c], cuts[c+1]) when told
// where they are, with ONE geometry and ONE bucket array for the whole call.
struct MsmChunks {
  uint32_t n_chunks = 0;
  const uint64_t* cuts = nullptr;  // n_chunks + 1 exponent indices: 0 = cuts[0] < ... < cuts[n_chunks] = n, inner cuts multiples of 32
  // the device pointer of chunk c's exponents; makes `st` wait (device-side) until they have arrived
  virtual int acquire(uint32_t c, hipStream_t st, const void** d_scalars) = 0;
  // the digit kernel of chunk c -- the only reader of its exponents -- has been enqueued on `st`
  virtual int digits_enqueued(uint32_t c, hipStream_t st) = 0;
  virtual ~MsmChunks() {}
};

// Lightweight per-kernel timing used by bench.py's roofline leg: when enabled, the library brackets
// the named kernels with hipEvents on the launch stream and accumulates their durations.
struct KernelTimer {
  const char* name;
  double total_ms = 0.0;
  long count = 0;
};

void prof_enable(bool on);
bool prof_enabled();
// record start/stop events around a launch on `st`; resolved lazily by prof_collect()
void prof_begin(int slot, hipStream_t st);
void prof_end(int slot, hipStream_t st);
void prof_collect();
int prof_slot(const char* name);  // find-or-create
bool prof_get(const char* name, double* total_ms, long* count);
void prof_reset();

}  // namespace zk
