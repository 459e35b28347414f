// Shared host-side helpers for the HIP translation units of libmi355zk.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

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

// A few persistent HOST threads for the window join that ends a multiexp (msm_impl.hpp): the ~200 window sums of a call are
// independent until the final chain of doublings, and on G2 their join is 0.5 ms of host arithmetic -- 9 % of a 2^20 call.
// run(n, fn) executes fn(0) .. fn(n - 1) on the calling thread and up to HELPERS helpers and returns when all are done; it
// returns false WITHOUT running anything when another caller is using the helpers (the prover joins eight multiexps at once:
// those callers take the single-threaded path rather than queue here).
class JoinPool {
 public:
  static constexpr unsigned HELPERS = 7;
  template <class Fn>
  bool run(uint32_t count, Fn&& f) {
    if (!owner_.try_lock()) return false;
    struct Release { std::mutex& m; ~Release() { m.unlock(); } } release{owner_};
    const std::function<void(uint32_t)> fn(std::forward<Fn>(f));
    {
      std::unique_lock<std::mutex> lk(mu_);
      if (threads_.empty()) for (unsigned t = 0; t < HELPERS; ++t) threads_.emplace_back([this] { worker(); });
      idle_cv_.wait(lk, [&] { return active_ == 0; });  // nobody is still inside work() of an earlier job
      fn_ = &fn;
      n_ = count;
      done_.store(0);
      next_.store(0);  // (last: a helper that sees the new counter sees the new job)
      ++gen_;
    }
    cv_.notify_all();
    work();
    while (done_.load(std::memory_order_acquire) < count) std::this_thread::yield();  // helpers finishing their last item
    return true;
  }
  ~JoinPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }

 private:
  void work() {
    for (;;) {
      const uint32_t i = next_.fetch_add(1);
      if (i >= n_) break;
      (*fn_)(i);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      ++active_;
      lk.unlock();
      work();
      lk.lock();
      if (--active_ == 0) idle_cv_.notify_all();
    }
  }
  std::mutex owner_, mu_;
  std::condition_variable cv_, idle_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0;
  std::atomic<uint32_t> next_{0}, done_{0};
  unsigned active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// Lightweight per-kernel timing used by bench.py's roofline leg: when enabled, the library brackets
// the named kernels with hipEvents on the launch stream and accumulates their durations.
struct KernelTimer {
  const char* name;
  double total_ms = 0.0;
  long count = 0;
};

void prof_enable(int on);   // 0 off, 1 every slot, 2 only the slot of prof_only
void prof_only(const char* name);
bool prof_enabled();
// record start/stop events around a launch on `st`; resolved lazily by prof_collect()
void prof_begin(int slot, hipStream_t st);
void prof_end(int slot, hipStream_t st);
void prof_collect();
int prof_slot(const char* name);  // find-or-create
bool prof_get(const char* name, double* total_ms, long* count);
void prof_reset();

// Every extern "C" entry point runs its body through one of these: the library's hosts are C / Rust / ctypes callers, and a C++
// exception (std::bad_alloc from a staging vector, std::system_error from a lock) must not unwind across that boundary.  It becomes
// a device error (rc < 0: the caller falls back to its own CPU path).
template <class Fn>
inline int abi_guard(Fn&& fn) noexcept {
  try {
    return fn();
  } catch (...) {
    return ZK_ERR_DEVICE;
  }
}
template <class Fn>
inline void abi_guard_void(Fn&& fn) noexcept {
  try {
    fn();
  } catch (...) {
  }
}

}  // namespace zk
