// C ABI of libmi355zk.so (include/mi355zk.h): the extern "C" entry points (argument checking, abi_guard), the domain constants of
// bellman/src/domain.rs:52-99, lifecycle, and the kernel-timing hooks used by bench.py.  The machinery behind the host-buffer entry points
// (Source / Density plan, bases cache, streamed upload, multi-GPU cells) is host_entry.hip, the scalar-multiplication kernels scalar_mul.hip;
// api_internal.hpp is the interface between the three.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <string>
#include <type_traits>
#include <algorithm>
#include <vector>

#include "../../include/mi355zk.h"
#include "curveu.hpp"
#include "glv.hpp"
#include "device_util.hpp"
#include "api_internal.hpp"

namespace zk {

// ------------------------------------------------------------------------------------------------
// profiling hooks
namespace {
struct ProfSlot {
  std::string name;
  double total_ms = 0;
  long count = 0;
  struct Pair {
    int dev;
    hipEvent_t a, b;
  };
  std::vector<Pair> pending;
  // the begin events waiting for their end, keyed by the STREAM they were recorded on: several host threads run msm_device at once (the
  // cells of the multi-GPU mode, the prover's eight multiexps), each on its own stream -- one `open` per slot paired thread A's end with
  // thread B's begin
  struct Open {
    hipStream_t st;
    hipEvent_t e;
    int dev;
  };
  std::vector<Open> open;
};
std::mutex g_prof_mu;
std::vector<ProfSlot> g_prof;
int g_prof_on = 0;                     // 0 off, 1 every slot, 2 only the slot named by prof_only (the dominant kernel: two events per call)
int g_prof_only = -1;
std::map<int, std::vector<hipEvent_t>> g_prof_free;   // per device; events are reused: creating one costs several microseconds on the launching thread
hipEvent_t prof_event(int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  *dev_out = dev;
  auto& pool = g_prof_free[dev];
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void prof_enable(int on) { g_prof_on = on; }
bool prof_enabled() { return g_prof_on != 0; }
int prof_slot(const char* name) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (size_t i = 0; i < g_prof.size(); ++i)
    if (g_prof[i].name == name) return (int)i;
  g_prof.emplace_back();
  g_prof.back().name = name;
  return (int)g_prof.size() - 1;
}
void prof_only(const char* name) { g_prof_only = name ? prof_slot(name) : -1; }
void prof_begin(int slot, hipStream_t st) {
  if (g_prof_on == 0 || (g_prof_on == 2 && slot != g_prof_only)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int dev = 0;
  hipEvent_t e = prof_event(&dev);
  if (e == nullptr) return;
  (void)hipEventRecord(e, st);
  for (auto& o : g_prof[slot].open)
    if (o.st == st && o.dev == dev) {     // (a begin without its end on this (device, stream): an error path.  Keyed by BOTH: two host threads on
                                          //  different devices may each use their device's NULL stream -- ADVICE r5)
      g_prof_free[o.dev].push_back(o.e);
      o.e = e;
      o.dev = dev;
      return;
    }
  g_prof[slot].open.push_back(ProfSlot::Open{st, e, dev});
}
void prof_end(int slot, hipStream_t st) {
  if (g_prof_on == 0 || (g_prof_on == 2 && slot != g_prof_only)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  auto& open = g_prof[slot].open;
  int dev = 0;
  hipEvent_t e = prof_event(&dev);
  if (e == nullptr) return;
  size_t k = 0;
  while (k < open.size() && !(open[k].st == st && open[k].dev == dev)) ++k;
  if (k == open.size()) {               // no begin on this (device, stream)
    g_prof_free[dev].push_back(e);
    return;
  }
  (void)hipEventRecord(e, st);
  g_prof[slot].pending.push_back(ProfSlot::Pair{dev, open[k].e, e});
  open.erase(open.begin() + (long)k);
}
void prof_collect() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof) {
    for (auto& pr : s.pending) {
      if (hipEventSynchronize(pr.b) == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
          s.total_ms += ms;
          s.count += 1;
        }
      }
      g_prof_free[pr.dev].push_back(pr.a);
      g_prof_free[pr.dev].push_back(pr.b);
    }
    s.pending.clear();
  }
}
bool prof_get(const char* name, double* total_ms, long* count) {
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof)
    if (s.name == name) {
      *total_ms = s.total_ms;
      *count = s.count;
      return true;
    }
  return false;
}
void prof_release_all() {   // mi355zk_shutdown: the pooled events go back to their devices
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& kv : g_prof_free) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    for (hipEvent_t e : kv.second) (void)hipEventDestroy(e);
  }
  g_prof_free.clear();
}
void prof_reset() {
  prof_collect();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& s : g_prof) {
    s.total_ms = 0;
    s.count = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// domain constants (host arithmetic, same field code as the kernels)
namespace {

Fr fr_from_u64(uint64_t v) {
  Fr c = Fr::zero();
  c.l[0] = (uint32_t)v;
  c.l[1] = (uint32_t)(v >> 32);
  return from_canonical(c);
}

// ff_derive: ROOT_OF_UNITY = GENERATOR^((r-1) >> S) with GENERATOR = 7, S = 28 (fr.rs:3-6,31-34)
Fr fr_root_of_unity() {
  uint32_t e[8];
  for (int i = 0; i < 8; ++i) e[i] = FrParams::P[i];
  e[0] -= 1;
  for (int i = 0; i < 8; ++i) e[i] = (e[i] >> 28) | (i < 7 ? e[i + 1] << 4 : 0);
  return pow_limbs(fr_from_u64(7), e, 8);
}

struct DomainConsts {
  Fr omega, omegainv, geninv, minv, gen;
};

// EvaluationDomain::from_coeffs for m = 2^exp (domain.rs:61-98)
int domain_consts(uint32_t exp, DomainConsts* d) {
  if (exp > 28) return ZK_ERR_BAD_ARGS;  // PolynomialDegreeTooLarge (domain.rs:75-77)
  static const Fr rou = fr_root_of_unity();
  Fr w = rou;
  for (uint32_t i = exp; i < 28; ++i) w = sqr(w);
  d->omega = w;
  d->omegainv = inv(w);
  d->gen = fr_from_u64(7);
  d->geninv = inv(d->gen);
  d->minv = inv(fr_from_u64(1ull << exp));
  return ZK_OK;
}
}  // namespace

int domain_op_dev(Fr* d_a, uint32_t log_n, int op, hipStream_t st) {
  DomainConsts D;
  int rc = domain_consts(log_n, &D);
  if (rc) return rc;
  switch (op) {
    case MI355ZK_OP_FFT:  // domain.rs:154-157
      return ntt_run(d_a, log_n, D.omega, st);
    case MI355ZK_OP_IFFT:  // domain.rs:159-174: best_fft(omegainv), then *= minv (fused into the last pass)
      return ntt_run_scaled(d_a, log_n, D.omegainv, nullptr, &D.minv, nullptr, st);
    case MI355ZK_OP_COSET_FFT:  // domain.rs:191-195: distribute_powers(g) (fused into the first pass), then fft
      return ntt_run_scaled(d_a, log_n, D.omega, &D.gen, nullptr, nullptr, st);
    case MI355ZK_OP_ICOSET_FFT:  // domain.rs:197-203: ifft, then distribute_powers(geninv) (both fused into the last pass)
      return ntt_run_scaled(d_a, log_n, D.omegainv, nullptr, &D.minv, &D.geninv, st);
    default:
      return ZK_ERR_BAD_ARGS;
  }
}

// the same operation on `batch` arrays of one size: one launch per pass over all of them (ntt.hip: ntt_run_batch), eight arrays at a time
int domain_op_batch_dev(Fr* const* d_arrays, uint32_t batch, uint32_t log_n, int op, hipStream_t st) {
  DomainConsts D;
  int rc = domain_consts(log_n, &D);
  if (rc) return rc;
  // from 2^21 on a pass has more tiles than the device has workgroup slots and pipelines by itself; a joint launch only makes the
  // transforms compete for the Infinity Cache (2^22: 0.459 -> 0.465 ms per transform): one at a time there
  const uint32_t per_launch = log_n <= 20 ? 8u : 1u;
  for (uint32_t done = 0; done < batch; done += per_launch) {
    const uint32_t k = batch - done < per_launch ? batch - done : per_launch;
    switch (op) {
      case MI355ZK_OP_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omega, nullptr, nullptr, nullptr, st); break;
      case MI355ZK_OP_IFFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omegainv, nullptr, &D.minv, nullptr, st); break;
      case MI355ZK_OP_COSET_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omega, &D.gen, nullptr, nullptr, st); break;
      case MI355ZK_OP_ICOSET_FFT: rc = ntt_run_batch(d_arrays + done, k, log_n, D.omegainv, nullptr, &D.minv, &D.geninv, st); break;
      default: return ZK_ERR_BAD_ARGS;
    }
    if (rc) return rc;
  }
  return ZK_OK;
}

}  // namespace zk

using namespace zk;

extern "C" {

int mi355zk_init(const int* device_ids, int n_devices) {
  return abi_guard([&]() -> int {
    if (n_devices < 0 || (n_devices > 0 && device_ids == nullptr)) return ZK_ERR_BAD_ARGS;
    int count = 0;
    ZK_HIP(hipGetDeviceCount(&count));
    auto check = [](int dev) -> int {
      hipDeviceProp_t prop;
      ZK_HIP(hipGetDeviceProperties(&prop, dev));
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "[mi355zk] device %d is %s; this library contains gfx950 code only\n", dev, prop.gcnArchName);
        return ZK_ERR_DEVICE;
      }
      return ZK_OK;
    };
    for (int i = 0; i < n_devices; ++i)
      if (device_ids[i] < 0 || device_ids[i] >= count) return ZK_ERR_BAD_ARGS;
    if (n_devices > 0) ZK_HIP(hipSetDevice(device_ids[0]));
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    // the LAST call defines the device set: more than one id = the single-process multi-GPU mode of the host-buffer multiexps
    // (msm_host_multi), one id or none = one device, as before
    std::vector<int> set;
    if (n_devices > 1) set.assign(device_ids, device_ids + n_devices);
    int rc = ZK_OK;
    if (set.empty()) {
      rc = check(dev);
      if (rc == ZK_OK) rc = ntt_configure();
    } else {
      for (size_t i = 0; i < set.size() && rc == ZK_OK; ++i) {
        bool seen = false;
        for (size_t k = 0; k < i; ++k) seen = seen || set[k] == set[i];
        if (seen) continue;
        rc = check(set[i]);
        if (rc == ZK_OK && hipSetDevice(set[i]) != hipSuccess) rc = ZK_ERR_DEVICE;
        if (rc == ZK_OK) rc = ntt_configure();
      }
      (void)hipSetDevice(dev);
    }
    if (rc != ZK_OK) return rc;
    devset_set(set);
    return ZK_OK;
  });
}
int mi355zk_visible_devices(void) {
  return abi_guard([&]() -> int {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
  });
}
int mi355zk_device_count(void) {
  return abi_guard([&]() -> int {
    return devset_count();
  });
}

void mi355zk_shutdown(void) {
  abi_guard_void([&] {
    DeviceGuard guard;
    prof_release_all();
    ntt_release_all();
    exp_scratch_release_all();
    mul_slots_release_all();
    host_entry_release_all();
    msm_release_g1();
    msm_release_g2();
  });
}

const char* mi355zk_version(void) { return "mi355zk 0.3 (gfx950)"; }
int mi355zk_abi_version(void) { return MI355ZK_ABI_VERSION; }

int mi355zk_bases_cache_pin(const void* host_bases, size_t n_bases, int group) { return abi_guard([&]() -> int { return bases_cache_pin(host_bases, n_bases, group); }); }
int mi355zk_bases_cache_pin_tables(const void* host_bases, size_t n_bases, int group) { return abi_guard([&]() -> int { return bases_cache_pin(host_bases, n_bases, group, true); }); }
int mi355zk_bases_cache_info(const void* host_bases, size_t* device_bytes, size_t* table_bytes) {
  return abi_guard([&]() -> int { return bases_cache_info(host_bases, device_bytes, table_bytes); });
}
void mi355zk_bases_cache_invalidate(const void* host_bases) {
  abi_guard_void([&] {
    int dev = 0;
    const bool have = hipGetDevice(&dev) == hipSuccess;
    bases_cache_invalidate(host_bases);
    if (have) (void)hipSetDevice(dev);
  });
}

int mi355zk_bn254_g1_msm(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                         const uint32_t* density, size_t density_bits, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return msm_host_entry<1>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  });
}
int mi355zk_bn254_g2_msm(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                         const uint32_t* density, size_t density_bits, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return msm_host_entry<2>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  });
}
int mi355zk_bn254_g1_msm_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                             const uint32_t* density, size_t density_bits, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz);
  });
}
int mi355zk_bn254_g2_msm_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                             const uint32_t* density, size_t density_bits, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz);
  });
}
// one window group of a multiexp (multi-GPU sharding by windows; shard.py)
int mi355zk_bn254_g1_msm_part_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                  const uint32_t* density, size_t density_bits, uint32_t window_groups, uint32_t window_group, void* stream,
                                  uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group);
  });
}
int mi355zk_bn254_g2_msm_part_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                  const uint32_t* density, size_t density_bits, uint32_t window_groups, uint32_t window_group, void* stream,
                                  uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group);
  });
}
// flags (MI355ZK_MSM_*) + window groups in one entry point
int mi355zk_bn254_g1_msm_ex_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                const uint32_t* density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups || (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY)) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group, flags);
  });
}
int mi355zk_bn254_g2_msm_ex_dev(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                const uint32_t* density, size_t density_bits, uint32_t flags, uint32_t window_groups, uint32_t window_group,
                                void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (window_groups == 0 || window_group >= window_groups || (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY)) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_bases, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, window_groups, window_group, flags);
  });
}
int mi355zk_msm_table_geometry(size_t n_bases, int group, uint32_t* window_bits, uint32_t* n_windows) {
  return abi_guard([&]() -> int {
    if ((group != 1 && group != 2) || n_bases >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    uint32_t c = 0, W = 0;
    msm_table_geometry(n_bases ? n_bases : 1, group, &c, &W, nullptr);
    if (window_bits) *window_bits = c;
    if (n_windows) *n_windows = W;
    return ZK_OK;
  });
}
int mi355zk_bn254_g1_msm_table_build_dev(const void* d_bases, size_t n_bases, void* d_table, size_t table_bytes, void* stream) {
  return abi_guard([&]() -> int {
    return msm_table_build<1>(d_bases, n_bases, d_table, table_bytes, stream);
  });
}
int mi355zk_bn254_g2_msm_table_build_dev(const void* d_bases, size_t n_bases, void* d_table, size_t table_bytes, void* stream) {
  return abi_guard([&]() -> int {
    return msm_table_build<2>(d_bases, n_bases, d_table, table_bytes, stream);
  });
}
int mi355zk_bn254_g1_msm_table_dev(const void* d_table, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                   const uint32_t* density, size_t density_bits, uint32_t flags, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<1>(d_table, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, 1, 0, flags, nullptr, true);
  });
}
int mi355zk_bn254_g2_msm_table_dev(const void* d_table, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                                   const uint32_t* density, size_t density_bits, uint32_t flags, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_MSM_SCALARS_MONTGOMERY) return ZK_ERR_BAD_ARGS;
    return msm_dev_entry<2>(d_table, n_bases, base_offset, d_scalars, n_scalars, density, density_bits, stream, out_xyz, 1, 0, flags, nullptr, true);
  });
}
int mi355zk_bn254_g1_dense_multiexp(const uint8_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    return dense_host<1>(bases, nullptr, scalars, n, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g2_dense_multiexp(const uint8_t* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    return dense_host<2>(bases, nullptr, scalars, n, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g1_merge_pairs(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t out_s[12], uint64_t out_sx[12]) {
  return abi_guard([&]() -> int {
    if (!v2) return ZK_ERR_BAD_ARGS;
    return dense_host<1>(v1, v2, rho, n, out_s, out_sx);
  });
}
int mi355zk_bn254_g2_merge_pairs(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t out_s[24], uint64_t out_sx[24]) {
  return abi_guard([&]() -> int {
    if (!v2) return ZK_ERR_BAD_ARGS;
    return dense_host<2>(v1, v2, rho, n, out_s, out_sx);
  });
}
int mi355zk_bn254_g1_dense_multiexp_dev(const void* d_bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xyz[12]) {
  return abi_guard([&]() -> int {
    if (!out_xyz || (n && (!d_bases || !d_scalars)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g1_dense_device(d_bases, nullptr, d_scalars, n, (hipStream_t)stream, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g2_dense_multiexp_dev(const void* d_bases, const void* d_scalars, size_t n, void* stream, uint64_t out_xyz[24]) {
  return abi_guard([&]() -> int {
    if (!out_xyz || (n && (!d_bases || !d_scalars)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g2_dense_device(d_bases, nullptr, d_scalars, n, (hipStream_t)stream, out_xyz, nullptr);
  });
}
int mi355zk_bn254_g1_merge_pairs_dev(const void* d_v1, const void* d_v2, const void* d_rho, size_t n, void* stream, uint64_t out_s[12],
                                     uint64_t out_sx[12]) {
  return abi_guard([&]() -> int {
    if (!out_s || !out_sx || (n && (!d_v1 || !d_v2 || !d_rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g1_dense_device(d_v1, d_v2, d_rho, n, (hipStream_t)stream, out_s, out_sx);
  });
}
int mi355zk_bn254_g2_merge_pairs_dev(const void* d_v1, const void* d_v2, const void* d_rho, size_t n, void* stream, uint64_t out_s[24],
                                     uint64_t out_sx[24]) {
  return abi_guard([&]() -> int {
    if (!out_s || !out_sx || (n && (!d_v1 || !d_v2 || !d_rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
    return msm_g2_dense_device(d_v1, d_v2, d_rho, n, (hipStream_t)stream, out_s, out_sx);
  });
}
long long mi355zk_last_error_index(void) { return t_last_err_index; }
int mi355zk_msm_window_bits(size_t n_scalars, int* n_windows) {
  return abi_guard([&]() -> int {
    uint32_t c = 0, W = 0;
    msm_geometry(n_scalars, 1, &c, &W);
    if (n_windows) *n_windows = (int)W;
    return (int)c;
  });
}
int mi355zk_msm_window_bits_groups(size_t n_scalars, uint32_t window_groups, int* n_windows) {
  return abi_guard([&]() -> int {
    uint32_t c = 0, W = 0;
    msm_geometry(n_scalars, window_groups, &c, &W);
    if (n_windows) *n_windows = (int)W;
    return (int)c;
  });
}
// (test hook) digit extraction of one scalar on the host: see msm_selftest_digits in msm_g1.hip
int mi355zk_selftest_msm_digits(size_t n_scalars, uint32_t window_groups, const uint32_t scalar[8], uint32_t w_start, uint32_t w_stop, int direct,
                                int32_t* digits, uint32_t* geom) {
  return abi_guard([&]() -> int {
    return msm_selftest_digits(n_scalars, window_groups, scalar, w_start, w_stop, direct, digits, geom);
  });
}

int mi355zk_bn254_fr_ntt(uint64_t* a, uint32_t log_n, const uint64_t omega[4]) {
  return abi_guard([&]() -> int {
    if (!omega) return ZK_ERR_BAD_ARGS;
    return ntt_host(a, log_n, -1, omega);
  });
}
int mi355zk_bn254_fr_domain_op(uint64_t* a, uint32_t log_n, int op) {
  return abi_guard([&]() -> int {
    if (op < 0 || op > 3) return ZK_ERR_BAD_ARGS;
    return ntt_host(a, log_n, op, nullptr);
  });
}
int mi355zk_bn254_fr_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_FFT, nullptr); }); }
int mi355zk_bn254_fr_ifft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_IFFT, nullptr); }); }
int mi355zk_bn254_fr_coset_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_COSET_FFT, nullptr); }); }
int mi355zk_bn254_fr_icoset_fft(uint64_t* a, uint32_t log_n) { return abi_guard([&]() -> int { return ntt_host(a, log_n, MI355ZK_OP_ICOSET_FFT, nullptr); }); }
int mi355zk_bn254_fr_ntt_dev(void* d_a, uint32_t log_n, const uint64_t omega[4], void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || !omega || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr w;
    std::memcpy(&w, omega, 32);
    return ntt_run((Fr*)d_a, log_n, w, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_ntt_scaled_dev(void* d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t pre_g[4], const uint64_t post_c[4],
                                    const uint64_t post_g[4], void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || !omega || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr w, g, c, h;
    std::memcpy(&w, omega, 32);
    if (pre_g) std::memcpy(&g, pre_g, 32);
    if (post_c) std::memcpy(&c, post_c, 32);
    if (post_g) std::memcpy(&h, post_g, 32);
    return ntt_run_scaled((Fr*)d_a, log_n, w, pre_g ? &g : nullptr, post_c ? &c : nullptr, post_g ? &h : nullptr, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_op_dev(void* d_a, uint32_t log_n, int op, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || log_n > 28) return ZK_ERR_BAD_ARGS;
    return domain_op_dev((Fr*)d_a, log_n, op, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_op_batch_dev(void* const* d_arrays, uint32_t batch, uint32_t log_n, int op, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_arrays || batch == 0 || batch > 64 || log_n > 28) return ZK_ERR_BAD_ARGS;
    for (uint32_t t = 0; t < batch; ++t) {
      if (!d_arrays[t]) return ZK_ERR_BAD_ARGS;
      for (uint32_t u = 0; u < t; ++u)
        if (d_arrays[u] == d_arrays[t]) return ZK_ERR_BAD_ARGS;   // (the transforms work in place: one array twice would be transformed twice at once)
    }
    return domain_op_batch_dev(reinterpret_cast<Fr* const*>(d_arrays), batch, log_n, op, (hipStream_t)stream);
  });
}
int mi355zk_bn254_fr_domain_constants(uint32_t log_n, uint64_t omega[4], uint64_t omegainv[4], uint64_t geninv[4], uint64_t minv[4]) {
  return abi_guard([&]() -> int {
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    if (omega) std::memcpy(omega, &D.omega, 32);
    if (omegainv) std::memcpy(omegainv, &D.omegainv, 32);
    if (geninv) std::memcpy(geninv, &D.geninv, 32);
    if (minv) std::memcpy(minv, &D.minv, 32);
    return ZK_OK;
  });
}

// EvaluationDomain::z (domain.rs:207-212) and divide_by_z_on_coset (domain.rs:217-234)
int mi355zk_bn254_fr_domain_z(uint32_t log_n, const uint64_t tau[4], uint64_t out[4]) {
  return abi_guard([&]() -> int {
    if (!tau || !out || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr t;
    std::memcpy(&t, tau, 32);
    for (uint32_t i = 0; i < log_n; ++i) t = sqr(t);   // tau.pow(&[m]) with m = 2^log_n
    t = sub(t, Fr::one());
    std::memcpy(out, &t, 32);
    return ZK_OK;
  });
}
int mi355zk_bn254_fr_divide_by_z_on_coset_dev(void* d_a, uint32_t log_n, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_a || log_n > 28) return ZK_ERR_BAD_ARGS;
    Fr z = fr_from_u64(7);                             // E::Fr::multiplicative_generator() (fr.rs:5)
    for (uint32_t i = 0; i < log_n; ++i) z = sqr(z);
    z = sub(z, Fr::one());
    return ntt_scale((Fr*)d_a, log_n, inv(z), nullptr, (hipStream_t)stream);
  });
}

// EvaluationDomain<Point<G1>>::{fft, ifft} on affine records (group.rs:22-51, domain.rs:154-173; prepare_phase2.rs:68-131)
int mi355zk_bn254_g1_point_fft_dev(void* d_points_affine, uint32_t log_n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_points_affine || log_n > 28 || (mode & ~(MI355ZK_FFT_INVERSE | MI355ZK_G2_TRUSTED_SUBGROUP))) return ZK_ERR_BAD_ARGS;
    const int inverse = mode & MI355ZK_FFT_INVERSE;
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    return point_fft_g1(d_points_affine, log_n, inverse ? D.omegainv : D.omega, inverse != 0, to_canonical(D.minv), (hipStream_t)stream);
  });
}

// point codecs (ec.rs:763-946, 1136-1344): wire encodings <-> raw affine records
int mi355zk_bn254_g1_decode_dev(void* d_out_affine, const void* d_in_bytes, size_t n, int compressed, int checked, void* stream, long long* err_index) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_affine || !d_in_bytes)) || ((uintptr_t)d_in_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_decode(1, d_out_affine, d_in_bytes, n, compressed, checked, (hipStream_t)stream, err_index);
  });
}
int mi355zk_bn254_g2_decode_dev(void* d_out_affine, const void* d_in_bytes, size_t n, int compressed, int checked, void* stream, long long* err_index) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_affine || !d_in_bytes)) || ((uintptr_t)d_in_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_decode(2, d_out_affine, d_in_bytes, n, compressed, checked, (hipStream_t)stream, err_index);
  });
}
int mi355zk_bn254_g1_encode_dev(void* d_out_bytes, const void* d_in_affine, size_t n, int compressed, void* stream) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_bytes || !d_in_affine)) || ((uintptr_t)d_out_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_encode(1, d_out_bytes, d_in_affine, n, compressed, (hipStream_t)stream);
  });
}
int mi355zk_bn254_g2_encode_dev(void* d_out_bytes, const void* d_in_affine, size_t n, int compressed, void* stream) {
  return abi_guard([&]() -> int {
    if ((n && (!d_out_bytes || !d_in_affine)) || ((uintptr_t)d_out_bytes & 3)) return ZK_ERR_BAD_ARGS;
    return codec_encode(2, d_out_bytes, d_in_affine, n, compressed, (hipStream_t)stream);
  });
}

int mi355zk_bn254_g2_point_fft_dev(void* d_points_affine, uint32_t log_n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (!d_points_affine || log_n > 28 || (mode & ~(MI355ZK_FFT_INVERSE | MI355ZK_G2_TRUSTED_SUBGROUP))) return ZK_ERR_BAD_ARGS;
    const int inverse = mode & MI355ZK_FFT_INVERSE;
    DomainConsts D;
    int rc = domain_consts(log_n, &D);
    if (rc) return rc;
    // Without the caller's promise: a transform of points that ALL lie in the order-r subgroup stays in it, so one membership test per
    // input (127 doublings + 68 additions each, against log_n / 2 multiplications per point) earns the psi split for every stage; one
    // record outside and the whole transform runs the plain windows.  Either way the reference's result.
    bool split = (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0;
    if (!split && log_n >= 5) {
      long long bad = -1;
      rc = g2_subgroup_check(d_points_affine, (size_t)1 << log_n, stream, &bad);
      if (rc) return rc;
      split = bad < 0;
    }
    return point_fft_g2(d_points_affine, log_n, inverse ? D.omegainv : D.omega, inverse != 0, to_canonical(D.minv), (hipStream_t)stream, split);
  });
}

int mi355zk_bn254_g2_subgroup_check_dev(const void* d_points_affine, size_t n, void* stream, long long* bad_index) {
  return abi_guard([&]() -> int {
    return g2_subgroup_check(d_points_affine, n, stream, bad_index);
  });
}
int mi355zk_selftest_g2_in_subgroup(const uint64_t affine_pt[16]) {  // the same test on the HOST: 1 in the subgroup, 0 not, < 0 bad arguments
  if (!affine_pt) return -1;
  G2Affine p;
  std::memcpy(&p, affine_pt, sizeof p);
  return g2_in_subgroup_host(p) ? 1 : 0;
}
int mi355zk_bn254_g1_batch_mul_dev(void* d_out_affine, const uint64_t base_affine[8], const void* d_scalars, size_t n, void* stream) {
  return abi_guard([&]() -> int {
    return batch_mul<Fq>(d_out_affine, base_affine, d_scalars, n, stream);
  });
}
int mi355zk_bn254_g2_batch_mul_dev(void* d_out_affine, const uint64_t base_affine[16], const void* d_scalars, size_t n, void* stream) {
  return abi_guard([&]() -> int {
    return batch_mul<Fq2>(d_out_affine, base_affine, d_scalars, n, stream);
  });
}
int mi355zk_bn254_g1_sparse_matvec(uint8_t* out_affine, const uint8_t* bases_affine, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col,
                                   const uint64_t* coeffs, size_t n_rows, size_t nnz, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec_host<Fq>(out_affine, bases_affine, n_bases, row_ptr, col, coeffs, n_rows, nnz, 1, false);
  });
}
int mi355zk_bn254_g2_sparse_matvec(uint8_t* out_affine, const uint8_t* bases_affine, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col,
                                   const uint64_t* coeffs, size_t n_rows, size_t nnz, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec_host<Fq2>(out_affine, bases_affine, n_bases, row_ptr, col, coeffs, n_rows, nnz, 2, (flags & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}
int mi355zk_bn254_g1_sparse_matvec_dev(void* d_out_affine, const void* d_bases_affine, size_t n_bases, const uint32_t* d_row_ptr,
                                       const uint32_t* d_col, const void* d_coeffs, size_t n_rows, size_t nnz, void* stream, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec<Fq>(d_out_affine, d_bases_affine, n_bases, d_row_ptr, d_col, d_coeffs, n_rows, nnz, stream, 1, false);
  });
}
int mi355zk_bn254_g2_sparse_matvec_dev(void* d_out_affine, const void* d_bases_affine, size_t n_bases, const uint32_t* d_row_ptr,
                                       const uint32_t* d_col, const void* d_coeffs, size_t n_rows, size_t nnz, void* stream, int flags) {
  return abi_guard([&]() -> int {
    if (flags & ~MI355ZK_G2_TRUSTED_SUBGROUP) return ZK_ERR_BAD_ARGS;
    return sparse_matvec<Fq2>(d_out_affine, d_bases_affine, n_bases, d_row_ptr, d_col, d_coeffs, n_rows, nnz, stream, 2, (flags & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}

int mi355zk_bn254_g1_batch_exp(uint8_t* out_affine, const uint8_t* bases_affine, const uint64_t* scalars, size_t n, int mode) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp_host<Fq>(out_affine, bases_affine, scalars, n, mode & MI355ZK_EXP_SAME_SCALAR, false);
  });
}
int mi355zk_bn254_g2_batch_exp(uint8_t* out_affine, const uint8_t* bases_affine, const uint64_t* scalars, size_t n, int mode) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp_host<Fq2>(out_affine, bases_affine, scalars, n, mode & MI355ZK_EXP_SAME_SCALAR, (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}
int mi355zk_bn254_g1_batch_exp_dev(void* d_out_affine, const void* d_bases_affine, const void* d_scalars, size_t n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp<Fq>(d_out_affine, d_bases_affine, 0, d_scalars, mode & MI355ZK_EXP_SAME_SCALAR, n, stream);
  });
}
int mi355zk_bn254_g2_batch_exp_dev(void* d_out_affine, const void* d_bases_affine, const void* d_scalars, size_t n, int mode, void* stream) {
  return abi_guard([&]() -> int {
    if (mode & ~(MI355ZK_EXP_SAME_SCALAR | MI355ZK_G2_TRUSTED_SUBGROUP)) return ZK_ERR_BAD_ARGS;
    return batch_exp<Fq2>(d_out_affine, d_bases_affine, 0, d_scalars, mode & MI355ZK_EXP_SAME_SCALAR, n, stream, nullptr, false,
                          (mode & MI355ZK_G2_TRUSTED_SUBGROUP) != 0);
  });
}

// host-side group helpers (joining per-GPU partial sums, normalising results)
int mi355zk_bn254_g1_add(uint64_t acc_xyz[12], const uint64_t other_xyz[12]) {
  return abi_guard([&]() -> int {
    if (!acc_xyz || !other_xyz) return ZK_ERR_BAD_ARGS;
    G1Jacobian a, b;
    std::memcpy(&a, acc_xyz, sizeof a);
    std::memcpy(&b, other_xyz, sizeof b);
    jac_add(a, b);
    std::memcpy(acc_xyz, &a, sizeof a);
    return ZK_OK;
  });
}
int mi355zk_bn254_g2_add(uint64_t acc_xyz[24], const uint64_t other_xyz[24]) {
  return abi_guard([&]() -> int {
    if (!acc_xyz || !other_xyz) return ZK_ERR_BAD_ARGS;
    G2Jacobian a, b;
    std::memcpy(&a, acc_xyz, sizeof a);
    std::memcpy(&b, other_xyz, sizeof b);
    jac_add(a, b);
    std::memcpy(acc_xyz, &a, sizeof a);
    return ZK_OK;
  });
}
// acc = k * acc on the host (CurveProjective::mul_assign, ec.rs:538-560: most significant bit first, leading zeros skipped): the
// handful of single-point products a proof assembly makes (prover.rs:300-333: vk.delta_g1.mul(r) ...)
extern "C++" template <class J>
static int host_scalar_mul(uint64_t* acc_xyz, const uint64_t k[4]) {
  if (!acc_xyz || !k) return ZK_ERR_BAD_ARGS;
  J base, res = J::zero();
  std::memcpy(&base, acc_xyz, sizeof base);
  bool found_one = false;
  for (int i = 255; i >= 0; --i) {
    const bool bit = (k[i >> 6] >> (i & 63)) & 1;
    if (found_one) jac_double(res);
    else found_one = bit;
    if (bit) jac_add(res, base);
  }
  std::memcpy(acc_xyz, &res, sizeof res);
  return ZK_OK;
}
int mi355zk_bn254_g1_mul(uint64_t acc_xyz[12], const uint64_t scalar[4]) { return abi_guard([&]() -> int { return host_scalar_mul<G1Jacobian>(acc_xyz, scalar); }); }
int mi355zk_bn254_g2_mul(uint64_t acc_xyz[24], const uint64_t scalar[4]) { return abi_guard([&]() -> int { return host_scalar_mul<G2Jacobian>(acc_xyz, scalar); }); }
// into_affine (ec.rs:596-629); infinity -> all-zero record
int mi355zk_bn254_g1_to_affine(uint64_t out_xy[8], const uint64_t xyz[12]) {
  return abi_guard([&]() -> int {
    if (!out_xy || !xyz) return ZK_ERR_BAD_ARGS;
    G1Jacobian p;
    std::memcpy(&p, xyz, sizeof p);
    G1Affine r{Fq::zero(), Fq::zero()};
    if (!p.is_zero()) {
      Fq zi = inv(p.z), zi2 = sqr(zi);
      r.x = mul(p.x, zi2);
      r.y = mul(p.y, mul(zi2, zi));
    }
    std::memcpy(out_xy, &r, sizeof r);
    return ZK_OK;
  });
}
int mi355zk_bn254_g2_to_affine(uint64_t out_xy[16], const uint64_t xyz[24]) {
  return abi_guard([&]() -> int {
    if (!out_xy || !xyz) return ZK_ERR_BAD_ARGS;
    G2Jacobian p;
    std::memcpy(&p, xyz, sizeof p);
    G2Affine r{Fq2::zero(), Fq2::zero()};
    if (!p.is_zero()) {
      Fq2 zi = inv(p.z), zi2 = sqr(zi);
      r.x = mul(p.x, zi2);
      r.y = mul(p.y, mul(zi2, zi));
    }
    std::memcpy(out_xy, &r, sizeof r);
    return ZK_OK;
  });
}

int mi355zk_malloc(void** d_ptr, size_t bytes) {
  return abi_guard([&]() -> int {
    if (!d_ptr) return ZK_ERR_BAD_ARGS;
    ZK_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ZK_OK;
  });
}
int mi355zk_free(void* d_ptr) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipFree(d_ptr));
    return ZK_OK;
  });
}
int mi355zk_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ZK_OK;
  });
}
int mi355zk_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
  });
}
int mi355zk_sync(void* stream) {
  return abi_guard([&]() -> int {
    ZK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ZK_OK;
  });
}

void mi355zk_prof_enable(int on) { abi_guard_void([&] { prof_enable(on < 0 ? 0 : on > 2 ? 1 : on); }); }
void mi355zk_prof_only(const char* name) { abi_guard_void([&] { prof_only(name); }); }
void mi355zk_prof_reset(void) { abi_guard_void([&] { prof_reset(); }); }
int mi355zk_prof_get(const char* kernel, double* total_ms, long* count) {
  return abi_guard([&]() -> int {
    double t = 0;
    long c = 0;
    bool ok = kernel && prof_get(kernel, &t, &c);
    if (total_ms) *total_ms = t;
    if (count) *count = c;
    return ok ? ZK_OK : ZK_ERR_BAD_ARGS;
  });
}

}  // extern "C"
