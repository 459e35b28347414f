// The host-buffer side of the C ABI (include/mi355zk.h): the Source / QueryDensity plan of bellman/src/source.rs, the device-resident call
// wrapper (msm_dev_entry), the pinned-bases cache, the streamed upload of a host-buffer multiexp (msm_host_run), the single-process multi-GPU
// mode (msm_host_multi), and the host-buffer forms of batch_exp, dense_multiexp / merge_pairs, the NTT and the QAP sparse matvec.  Split out of
// api.hip in round 6 (the extern "C" wrappers stay there); the interface to the other translation units is api_internal.hpp.
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
thread_local long long t_last_err_index = -1;

// Source / QueryDensity contract (source.rs:36-118, multiexp.rs:92): returns the number of exponents
// to process, the exponent index of the first UnexpectedEof (or -1) and, for a density map, the
// per-word exclusive prefix popcounts.
struct DensityPlan {
  uint64_t n = 0;
  long long eof_index = -1;
  std::vector<uint32_t> prefix;
};

int plan_density(size_t n_bases, size_t base_offset, size_t n_scalars, const uint32_t* density, size_t density_bits, DensityPlan* P) {
  uint64_t n = n_scalars;
  if (density != nullptr && density_bits < n) n = density_bits;  // zip() stops at the shorter (multiexp.rs:92)
  P->n = n;
  uint64_t avail = base_offset < n_bases ? n_bases - base_offset : 0;
  if (density == nullptr) {
    if (n > avail) P->eof_index = (long long)avail;
    return ZK_OK;
  }
  uint64_t words = (n + 31) / 32;
  P->prefix.resize(words ? words : 1);
  uint64_t used = 0;
  for (uint64_t w = 0; w < words; ++w) {
    P->prefix[w] = (uint32_t)used;
    uint32_t v = density[w];
    if (w == words - 1 && (n & 31)) v &= (1u << (n & 31)) - 1u;
    uint32_t pc = (uint32_t)__builtin_popcount(v);
    if (P->eof_index < 0 && used + pc > avail) {
      // the (avail - used + 1)-th set bit of this word is the first exponent without a base
      uint64_t need = avail - used;
      for (uint32_t b = 0; b < 32; ++b)
        if ((v >> b) & 1) {
          if (need == 0) { P->eof_index = (long long)(w * 32 + b); break; }
          --need;
        }
    }
    used += pc;
  }
  return ZK_OK;
}

// device copies of density maps (words + prefix popcounts): grow-only buffers, leased per call
struct DensityPool {
  struct Buf {
    int dev = -1;
    void* p = nullptr;
    size_t bytes = 0;
    bool busy = false;
  };
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::vector<Buf*>& all() { static std::vector<Buf*> v; return v; }
  struct Lease {
    Buf* b = nullptr;
    hipStream_t st = nullptr;
    int acquire(int dev, size_t bytes, hipStream_t stream) {
      st = stream;
      {
        std::lock_guard<std::mutex> lk(mu());
        for (Buf* x : all())  // the smallest idle buffer that fits, else the largest idle one (regrown below)
          if (!x->busy && x->dev == dev) {
            if (b == nullptr) { b = x; continue; }
            const bool fits = x->bytes >= bytes, bfits = b->bytes >= bytes;
            if (fits ? (!bfits || x->bytes < b->bytes) : (!bfits && x->bytes > b->bytes)) b = x;
          }
        if (b == nullptr) {
          b = new Buf();
          b->dev = dev;
          all().push_back(b);
        }
        b->busy = true;
      }
      if (b->bytes < bytes) {
        if (b->p) ZK_HIP(hipFree(b->p));  // idle: its last user's stream was synchronised before the release
        b->p = nullptr;
        b->bytes = 0;
        ZK_HIP(hipMalloc(&b->p, bytes));
        b->bytes = bytes;
      }
      return ZK_OK;
    }
    ~Lease() {
      if (b == nullptr) return;
      (void)hipStreamSynchronize(st);  // (idle already after a completed call: the result came back over this stream)
      std::lock_guard<std::mutex> lk(mu());
      b->busy = false;
    }
  };
};

template <int GROUP>
int msm_dev_entry(const void* d_bases, size_t n_bases, size_t base_offset, const void* d_scalars, size_t n_scalars,
                  const uint32_t* density, size_t density_bits, void* stream, uint64_t* out_xyz, uint32_t wgroups, uint32_t wgroup,
                  uint32_t flags, MsmChunks* chunks, bool table) {
  // table: d_bases is the window table msm_table_build made of a vector of n_bases points (table mode, msm_impl.hpp)
  // chunks != nullptr: the exponents are handed over chunk by chunk while the call runs (msm_host_entry); d_scalars is unused
  t_last_err_index = -1;
  if (!out_xyz || (n_scalars && !d_scalars && !chunks) || (n_bases && !d_bases)) return ZK_ERR_BAD_ARGS;
  if (n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  hipStream_t st = (hipStream_t)stream;
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;  // exponents before the first Eof
  uint32_t* d_density = nullptr;
  uint32_t* d_prefix = nullptr;
  DensityPool::Lease density_lease;
  if (density != nullptr && n > 0) {
    // leased from a small pool for the duration of the call: hipMalloc / hipFree per call would synchronise the whole device and
    // with it every other thread's multiexp, and a buffer per host thread would outlive short-lived caller threads
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    size_t words = (n + 31) / 32;
    rc = density_lease.acquire(dev, words * 8, st);
    if (rc) return rc;
    DensityPool::Buf& buf = *density_lease.b;
    d_density = (uint32_t*)buf.p;
    d_prefix = d_density + words;
    ZK_HIP(hipMemcpyAsync(d_density, density, words * 4, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(d_prefix, P.prefix.data(), words * 4, hipMemcpyHostToDevice, st));
  }
  long long err_index = -1;
  const bool mont = (flags & MI355ZK_MSM_SCALARS_MONTGOMERY) != 0;
  if (chunks && (chunks->n_chunks == 0 || chunks->cuts[chunks->n_chunks] != n)) return ZK_ERR_BAD_ARGS;
  uint32_t tc = 0, tW = 0;
  if (table) msm_table_geometry(n_bases, GROUP, &tc, &tW, nullptr);
  const uint64_t tstride = table ? (uint64_t)n_bases : 0;
  if (table && n_bases == 0 && n > 0) return ZK_ERR_BAD_ARGS;
  if (GROUP == 1) rc = msm_g1_device(d_bases, n_bases, base_offset, d_scalars, n, d_density, d_prefix, st, out_xyz, &err_index, wgroups, wgroup, mont, chunks, tstride, tc);
  else rc = msm_g2_device(d_bases, n_bases, base_offset, d_scalars, n, d_density, d_prefix, st, out_xyz, &err_index, wgroups, wgroup, mont, chunks, tstride, tc);
  if (rc == ZK_ERR_UNEXPECTED_IDENTITY) {
    // the kernels report the lowest BASE index that was the identity under a non-zero exponent; the exponent that owns it
    // is the (index - base_offset)-th selected one (source.rs:101-118): itself under FullDensity
    long long rank = err_index - (long long)base_offset;
    if (density != nullptr) {
      size_t w = 0;
      const size_t words = (n + 31) / 32;
      while (w + 1 < words && (long long)P.prefix[w + 1] <= rank) ++w;
      uint32_t word = density[w];
      if ((w + 1) * 32 > n) word &= (n & 31) ? ((1u << (n & 31)) - 1u) : 0xffffffffu;
      long long need = rank - (long long)P.prefix[w];
      uint32_t b = 0;
      for (; b < 32; ++b)
        if ((word >> b) & 1u) { if (need == 0) break; --need; }
      rank = (long long)(w * 32 + b);
    }
    t_last_err_index = rank;
    return rc;
  }
  if (rc == ZK_ERR_BAD_ARGS) t_last_err_index = err_index;  // a non-canonical exponent (>= 2^254): its index
  if (rc != ZK_OK) return rc;
  if (P.eof_index >= 0) { t_last_err_index = P.eof_index; return ZK_ERR_UNEXPECTED_EOF; }
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-buffer entry points (SURVEY 8b "Ownership"): the caller's bases and scalars live in (pageable) host memory.
//   * BASES CACHE: the CRS / tau-table is reused across calls (`Arc<Vec<G>>` inside groth16::Parameters, groth16/mod.rs:216-238),
//     so the device copy of a base vector the caller has PINNED (mi355zk_bases_cache_pin: "this host vector is immutable until
//     I invalidate it" -- the shim holds a clone of the Arc, so the allocation can neither be rewritten nor freed and reused)
//     stays on the device, keyed by (host pointer, length, group) with a fingerprint of sampled records as a safety net;
//     LRU-bounded (env MI355ZK_BASES_CACHE_GB, default 64; 0 disables).  Vectors that were not pinned are uploaded on every call
//     (env MI355ZK_BASES_CACHE_IMPLICIT=1 restores round 2's behaviour: every vector is treated as pinned).
//   * STREAMED UPLOAD: a large call is cut into chunks of ~2^24 exponents; a copy thread uploads chunk i + 1 (its scalars into
//     one of two staging buffers, its bases -- when they are not cached yet -- straight into the cache entry) on a copy stream
//     while the calling thread runs the multiexp of chunk i on a compute stream; the Jacobian partials are added on the host.
//     PCIe and the kernels overlap; the first call is bound by the link (96 B per exponent), later calls by the kernels.

struct BasesEntry {
  const void* host = nullptr;   // first record of what is cached: the pinned vector itself, or the SLICE of it a multi-GPU cell consumes
  const void* owner = nullptr;  // the pinned vector the records belong to (== host unless a slice): what invalidate / info are asked about
  size_t n = 0;
  int group = 0, dev = 0;
  uint64_t fp = 0;
  void* d = nullptr;
  size_t bytes = 0;
  uint64_t tick = 0;
  bool ready = false;      // fully uploaded
  std::mutex fill_mu;      // held by the call that uploads it
  // the vector's WINDOW TABLE (table mode, msm_impl.hpp), for vectors pinned with mi355zk_bases_cache_pin_tables: built by the first
  // call that finds the entry ready, counted against the cache's capacity, freed with the entry
  bool want_table = false, table_failed = false;
  void* table = nullptr;
  size_t table_bytes = 0;
  size_t table_reserved = 0;  // bytes set aside under g_bc_mu while the table is being built (concurrent builds cannot overbook the cache)
  std::mutex table_mu;
};
std::mutex g_bc_mu;
std::vector<std::shared_ptr<BasesEntry>> g_bc;
uint64_t g_bc_tick = 0;

uint64_t fnv1a(uint64_t h, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
// first / last 4 KiB and 4096 records spread over the array: cheap (~0.3 MB hashed), and a different CRS at the same address
// is caught; a few records rewritten IN PLACE are not -- which is why caching is OPT-IN: only vectors the caller pinned (declared
// immutable) are served from the device copy, the fingerprint is a second line of defence, not the contract
uint64_t bases_fingerprint(const uint8_t* p, size_t bytes, size_t rec) {
  uint64_t h = 0xcbf29ce484222325ull;
  const size_t edge = bytes < 4096 ? bytes : 4096;
  h = fnv1a(h, p, edge);
  h = fnv1a(h, p + bytes - edge, edge);
  const size_t nrec = bytes / rec;
  for (size_t k = 1; k <= 4096 && nrec > 0; ++k) h = fnv1a(h, p + (nrec * k / 4097) * rec, rec);
  return h;
}
size_t bases_cache_cap() {
  static const char* env = std::getenv("MI355ZK_BASES_CACHE_GB");
  const double gb = env ? std::atof(env) : 64.0;
  return gb <= 0 ? 0 : (size_t)(gb * 1073741824.0);
}
// the vectors the caller declared immutable (mi355zk_bases_cache_pin)
struct BasesPin {
  const void* host;
  size_t n;
  int group;
  bool tables;
};
std::vector<BasesPin> g_bc_pins;  // under g_bc_mu
bool bases_cache_implicit() {
  static const char* env = std::getenv("MI355ZK_BASES_CACHE_IMPLICIT");
  return env && env[0] == '1';
}
int bases_cache_pin(const void* host, size_t n, int group, bool tables) {
  if (!host || n == 0 || (group != 1 && group != 2)) return ZK_ERR_BAD_ARGS;
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (auto& p : g_bc_pins)
    if (p.host == host && p.n == n && p.group == group) {
      p.tables = p.tables || tables;
      return ZK_OK;
    }
  g_bc_pins.push_back(BasesPin{host, n, group, tables});
  return ZK_OK;
}
// returns the entry (locked for filling when *fill == true: the caller uploads and then sets ready) or nullptr (cache off / not
// pinned / no room)
// the vector a multi-GPU cell's slice was cut from (set by the cell's thread around its msm_host_run): the owner of an IMPLICITLY cached
// slice, so that mi355zk_bases_cache_invalidate(vector) reaches the slices on every device (a pinned vector's slices find
// their owner in the pin list)
thread_local const void* t_bases_parent = nullptr;
std::shared_ptr<BasesEntry> bases_lookup(const void* host, size_t n, int group, size_t bytes, int dev, bool* fill) {
  *fill = false;
  const size_t cap = bases_cache_cap();
  if (cap == 0 || bytes > cap) return nullptr;
  bool want_table = false;
  const void* owner = t_bases_parent ? t_bases_parent : host;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    // pinned: the vector itself, or a record range INSIDE a pinned vector (the single-process multi-GPU mode caches on each device only
    // the slice its cell consumes: SURVEY 8e "the tau-table slice stays resident on its GPU")
    bool pinned = false;
    const size_t rec = group == 1 ? 64 : 128;
    for (auto& p : g_bc_pins) {
      if (p.group != group) continue;
      const char* lo = (const char*)p.host;
      if ((const char*)host >= lo && (const char*)host + n * rec <= lo + p.n * rec) {
        pinned = true;
        owner = p.host;
        want_table = want_table || (p.tables && p.host == host && p.n == n);   // (tables for whole vectors only: a slice's calls are cells)
      }
    }
    if (!pinned && !bases_cache_implicit()) return nullptr;
  }
  const uint64_t fp = bases_fingerprint((const uint8_t*)host, bytes, group == 1 ? 64 : 128);
  std::shared_ptr<BasesEntry> hit;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc)
      if (e->host == host && e->n == n && e->group == group && e->dev == dev && e->fp == fp) { hit = e; break; }
    if (hit) { hit->tick = ++g_bc_tick; hit->want_table = hit->want_table || want_table; }
  }
  if (hit) {
    std::lock_guard<std::mutex> wait_fill(hit->fill_mu);  // another thread may still be uploading it
    if (hit->ready) return hit;
    return nullptr;                                       // its upload failed: go uncached
  }
  auto e = std::make_shared<BasesEntry>();
  e->host = host; e->owner = owner; e->n = n; e->group = group; e->dev = dev; e->fp = fp; e->bytes = bytes; e->want_table = want_table;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    // the capacity is PER DEVICE (a process may drive several: mi355zk_init with n_devices > 1 keeps a copy of a pinned vector on
    // every device that evaluates cells over it)
    size_t used = 0;
    for (auto& x : g_bc)
      if (x->dev == dev) used += x->bytes + x->table_bytes + x->table_reserved;
    while (used + bytes > cap && !g_bc.empty()) {           // evict least recently used entries nobody is filling
      size_t victim = g_bc.size();
      for (size_t i = 0; i < g_bc.size(); ++i)
        if (g_bc[i]->dev == dev && g_bc[i]->ready && g_bc[i].use_count() == 1 && g_bc[i]->table_reserved == 0 &&
            (victim == g_bc.size() || g_bc[i]->tick < g_bc[victim]->tick))
          victim = i;
      if (victim == g_bc.size()) break;
      (void)hipFree(g_bc[victim]->d);
      (void)hipFree(g_bc[victim]->table);
      used -= g_bc[victim]->bytes + g_bc[victim]->table_bytes;
      g_bc.erase(g_bc.begin() + (long)victim);
    }
    if (used + bytes > cap) return nullptr;
    if (hipMalloc(&e->d, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e->tick = ++g_bc_tick;
    e->fill_mu.lock();
    g_bc.push_back(e);
  }
  *fill = true;
  return e;
}
void bases_drop(const std::shared_ptr<BasesEntry>& e) {  // a failed upload
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (size_t i = 0; i < g_bc.size(); ++i)
    if (g_bc[i] == e) { g_bc.erase(g_bc.begin() + (long)i); break; }
  (void)hipFree(e->d);
  (void)hipFree(e->table);
  e->d = e->table = nullptr;
  e->table_bytes = 0;
}

// two staging buffers for scalar chunks, a bases buffer for uncached calls, the two streams.  Leased from a pool for the duration
// of a call (callers come and go -- the prover queues its multiexps from short-lived threads -- and their buffers must not pile up)
struct HostStage {
  int dev = -1;
  bool busy = false;
  void* sc[2] = {nullptr, nullptr};
  size_t sc_bytes = 0;
  void* bases = nullptr;
  size_t bases_bytes = 0;
  hipStream_t copy = nullptr, compute = nullptr;
};
std::mutex g_stage_mu;
std::vector<HostStage*> g_stages;  // the pool: as many stages as there have been concurrent host-buffer calls
struct StageLease {
  HostStage* s = nullptr;
  ~StageLease() {
    if (s == nullptr) return;
    // every exit of msm_host_entry has joined its copy thread; the compute stream is idle after the last chunk's result came back,
    // except on an error path
    (void)hipStreamSynchronize(s->compute);
    (void)hipStreamSynchronize(s->copy);
    std::lock_guard<std::mutex> lk(g_stage_mu);
    s->busy = false;
  }
};
HostStage* host_stage(int dev, StageLease* lease) {
  HostStage* mine = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (HostStage* s : g_stages)  // the idle stage of this device with the largest staging buffers
      if (!s->busy && s->dev == dev && (mine == nullptr || s->sc_bytes > mine->sc_bytes)) mine = s;
    if (mine) mine->busy = true;
  }
  if (mine == nullptr) {
    mine = new HostStage();
    mine->dev = dev;
    mine->busy = true;
    if (hipStreamCreateWithFlags(&mine->copy, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&mine->compute, hipStreamNonBlocking) != hipSuccess) {
      delete mine;
      return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_stage_mu);
    g_stages.push_back(mine);
  }
  lease->s = mine;
  return mine;
}
int stage_reserve(void** p, size_t* have, size_t want) {
  if (*have >= want) return ZK_OK;
  if (*p) ZK_HIP(hipFree(*p));
  *p = nullptr;
  *have = 0;
  ZK_HIP(hipMalloc(p, want));
  *have = want;
  return ZK_OK;
}

// forget the device copies of the base vector at `host` (nullptr: of every vector); entries in use stay until their call ends
void bases_cache_invalidate(const void* host) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::lock_guard<std::mutex> lk(g_bc_mu);
  for (size_t i = 0; i < g_bc_pins.size();) {  // the promise of immutability ends here
    if (host == nullptr || g_bc_pins[i].host == host) g_bc_pins.erase(g_bc_pins.begin() + (long)i);
    else ++i;
  }
  for (size_t i = 0; i < g_bc.size();) {
    if ((host == nullptr || g_bc[i]->host == host || g_bc[i]->owner == host) && g_bc[i]->ready && g_bc[i].use_count() == 1) {
      (void)hipSetDevice(g_bc[i]->dev);
      (void)hipFree(g_bc[i]->d);
      (void)hipFree(g_bc[i]->table);
      g_bc.erase(g_bc.begin() + (long)i);
    } else {
      if (host == nullptr || g_bc[i]->host == host || g_bc[i]->owner == host) g_bc[i]->fp ^= 0x9e3779b97f4a7c15ull;  // in use: never matched again
      ++i;
    }
  }
  (void)hipSetDevice(cur);
}

void host_entry_release_all() {
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc) { (void)hipSetDevice(e->dev); (void)hipFree(e->d); (void)hipFree(e->table); }
    g_bc.clear();
  }
  std::lock_guard<std::mutex> lk(g_stage_mu);
  for (HostStage* s : g_stages) {
    (void)hipSetDevice(s->dev);
    (void)hipFree(s->sc[0]); (void)hipFree(s->sc[1]); (void)hipFree(s->bases);
    s->sc[0] = s->sc[1] = s->bases = nullptr;
    s->sc_bytes = s->bases_bytes = 0;
  }
  std::lock_guard<std::mutex> dl(DensityPool::mu());
  for (DensityPool::Buf* b : DensityPool::all()) {
    (void)hipSetDevice(b->dev);
    (void)hipFree(b->p);
    b->p = nullptr;
    b->bytes = 0;
  }
}

// the window table of a ready cache entry whose vector was pinned with tables: built by the first call that asks (the others wait on
// table_mu), inside the cache's capacity (no eviction for it: a table that does not fit is not built and the calls stay plain)
template <int GROUP>
const void* bases_table(const std::shared_ptr<BasesEntry>& e, hipStream_t st) {
  if (!e || !e->want_table || !e->ready) return nullptr;
  std::lock_guard<std::mutex> lk(e->table_mu);
  if (e->table) return e->table;
  if (e->table_failed) return nullptr;
  uint32_t c = 0, W = 0;
  msm_table_geometry(e->n, GROUP, &c, &W, nullptr);
  const size_t bytes = (size_t)W * e->bytes;
  if ((uint64_t)W * e->n > 0x7fffffffull) { e->table_failed = true; return nullptr; }
  {
    // reserve the room before the build: the prover's eight threads build the tables of different vectors at the same time.  A
    // cache that is full NOW is not a failure of this vector -- the next call asks again, after evictions may have made room.
    std::lock_guard<std::mutex> g(g_bc_mu);
    size_t used = 0;
    for (auto& x : g_bc)
      if (x->dev == e->dev) used += x->bytes + x->table_bytes + x->table_reserved;
    if (used + bytes > bases_cache_cap()) return nullptr;
    e->table_reserved = bytes;
  }
  void* t = nullptr;
  bool ok = hipMalloc(&t, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  if (ok && msm_table_build<GROUP>(e->d, e->n, t, bytes, (void*)st) != ZK_OK) { (void)hipFree(t); ok = false; }
  std::lock_guard<std::mutex> g(g_bc_mu);
  e->table_reserved = 0;
  if (!ok) { e->table_failed = true; return nullptr; }  // allocation or build failed: not tried again for this entry
  e->table = t;
  e->table_bytes = bytes;
  return t;
}

constexpr uint64_t HOST_CHUNK_UPLOAD = 1ull << 23;  // exponents per chunk of a streamed call whose bases travel too (link-bound)
constexpr uint64_t HOST_CHUNK_MIN = 1ull << 21;     // smallest first chunk of a call whose bases are on the device; below 4 of these the call is not cut

// One host-buffer multiexp on the calling thread's CURRENT device.  (wgroups, wgroup): only that group of scalar windows (a cell of
// the single-process multi-GPU mode below; (1, 0) is the whole multiexp).
template <int GROUP>
int msm_host_run(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                 const uint32_t* density, size_t density_bits, uint64_t* out_xyz, uint32_t wgroups = 1, uint32_t wgroup = 0) {
  t_last_err_index = -1;
  if (!out_xyz || (n_scalars && !scalars) || (n_bases && !bases)) return ZK_ERR_BAD_ARGS;
  if (n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  constexpr size_t bsz = GROUP == 1 ? 64 : 128;
  constexpr size_t jac_words = GROUP == 1 ? 12 : 24;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  StageLease stage_lease;
  HostStage* S = host_stage(dev, &stage_lease);
  if (S == nullptr) return ZK_ERR_DEVICE;

  // the exponents this call evaluates and the bases they consume (source.rs:36-118)
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  const uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;   // exponents before the first Eof
  auto rank_of = [&](uint64_t i) -> uint64_t {                          // bases consumed by exponents [0, i)
    if (density == nullptr) return i;
    if (i == 0) return 0;
    const uint64_t w = i >> 5;
    uint64_t r = w < P.prefix.size() ? P.prefix[w] : 0;
    if (w >= P.prefix.size()) {  // i == n on a word boundary past the last planned word
      const uint64_t lw = P.prefix.size() - 1;
      uint32_t v = density[lw];
      if ((lw + 1) * 32 > P.n) v &= (P.n & 31) ? ((1u << (P.n & 31)) - 1u) : 0xffffffffu;
      return P.prefix[lw] + (uint32_t)__builtin_popcount(v);
    }
    if (i & 31) r += (uint32_t)__builtin_popcount(density[w] & ((1u << (i & 31)) - 1u));
    return r;
  };

  // ---- bases: cached, being cached by this call, or (not pinned / cache off / full) the leased stage's buffer
  bool fill = false;
  std::shared_ptr<BasesEntry> entry = n_bases ? bases_lookup(bases, n_bases, GROUP, n_bases * bsz, dev, &fill) : nullptr;
  void* d_bases = entry ? entry->d : nullptr;
  bool upload_bases = fill;
  if (!entry && n_bases) {
    rc = stage_reserve(&S->bases, &S->bases_bytes, n_bases * bsz);
    if (rc) return rc;
    d_bases = S->bases;
    upload_bases = true;
  }
  struct FillGuard {  // whatever happens, the entry is either ready or gone when this call returns
    std::shared_ptr<BasesEntry> e;
    bool fill, ok = false;
    ~FillGuard() {
      if (!fill) return;
      e->ready = ok;
      e->fill_mu.unlock();
      if (!ok) bases_drop(e);
    }
  } guard{entry, fill};

  // ---- chunks (cut at multiples of 32 exponents, so that density words are not shared between chunks).  Every chunk runs digits ->
  // partition -> accumulate into the ONE bucket array of the call (msm_device, MsmChunks); what a chunk costs on top of its share
  // of the work is the re-partition of the bucket bounds and one read + write of every bucket record it touches (~1.3 ms at 2^26).
  std::vector<uint64_t> cuts{0};
  const char* env_grow = std::getenv("MI355ZK_HOST_CHUNK_GROWTH");  // percent (read per call: tools/exp_host_chunks.py sweeps it in one process)
  const char* env_first = std::getenv("MI355ZK_HOST_CHUNK_FIRST");  // log2 of the first chunk (bases on the device)
  // (test hook, read on every call: MI355ZK_HOST_CHUNK_TEST = exponents per chunk, a multiple of 32 -- cuts calls of ANY size, so
  // that the chunked path can be held against the CPU oracle at sizes the oracle finishes in seconds)
  const char* env_test = std::getenv("MI355ZK_HOST_CHUNK_TEST");
  const uint64_t test_chunk = env_test ? (uint64_t)std::strtoull(env_test, nullptr, 10) & ~31ull : 0;
  if (test_chunk >= 32) {
    for (uint64_t lo = test_chunk; lo < n; lo += test_chunk) cuts.push_back(lo);
  } else if (n >= 4 * HOST_CHUNK_MIN) {
    if (upload_bases) {
      // Bases travelling too (96 B per exponent): the link is the bottleneck and the kernels of a chunk finish long before the
      // next one has arrived; even chunks, small enough that the last one's kernels are a short tail behind the last byte.
      uint64_t k = (n + HOST_CHUNK_UPLOAD - 1) / HOST_CHUNK_UPLOAD;
      if (k < 2) k = 2;
      const uint64_t per = ((n + k - 1) / k + 31) & ~31ull;
      for (uint64_t lo = per; lo < n; lo += per) cuts.push_back(lo);
    } else {
      // Bases on the device: the kernels are the bottleneck (~1 G exponents/s against ~1.7 G/s of link).  Only the FIRST chunk's
      // upload is exposed, so it is small; each following chunk may be ~1.8 x the previous one and still arrive before the
      // kernels of its predecessor are done.
      const double grow = env_grow && std::atoi(env_grow) >= 100 ? std::atoi(env_grow) / 100.0 : 1.8;
      uint64_t sz = n / 20 > HOST_CHUNK_MIN ? n / 20 : HOST_CHUNK_MIN;
      if (env_first && std::atoi(env_first) >= 16 && std::atoi(env_first) <= 30) sz = 1ull << std::atoi(env_first);
      sz = (sz + 31) & ~31ull;
      uint64_t lo = 0;
      while (n - lo > sz + sz / 2) {  // the last chunk takes what is left, up to 1.5 x the next size
        lo += sz;
        cuts.push_back(lo);
        sz = ((uint64_t)((double)sz * grow) + 31) & ~31ull;
      }
    }
  }
  if (n) cuts.push_back(n);
  const uint64_t n_chunks = cuts.size() - 1;
  uint64_t max_chunk = 0;
  for (uint64_t c = 0; c < n_chunks; ++c) max_chunk = std::max(max_chunk, cuts[c + 1] - cuts[c]);
  const size_t sc_bytes = (size_t)max_chunk * 32;
  if (n) {
    for (int k = 0; k < 2; ++k) {
      size_t have = S->sc_bytes;
      rc = stage_reserve(&S->sc[k], &have, sc_bytes);
      if (rc) { S->sc_bytes = 0; return rc; }
    }
    if (S->sc_bytes < sc_bytes) S->sc_bytes = sc_bytes;
  }

  // The copy thread: for chunk c, scalars -> staging[c & 1] and (when uploading) the bases the chunk consumes; afterwards the
  // bases outside the consumed range, so that a cache entry is complete.  staging[c & 1] is free again once the DIGIT kernel of
  // chunk c - 2 -- the only reader of a chunk's exponents -- has run: the compute side records an event behind it.
  struct Feed : MsmChunks {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t copied = 0, digits = 0;          // chunks uploaded / chunks whose digit kernel has been enqueued
    bool copy_failed = false, abort_copy = false;
    void* sc[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;               // ev[c]: recorded behind chunk c's digit kernel
    int acquire(uint32_t c, hipStream_t, const void** d) override {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return copy_failed || copied > c; });  // (a host-side wait: the earlier chunks' kernels are already queued)
      if (copy_failed) return ZK_ERR_DEVICE;
      *d = sc[c & 1];
      return ZK_OK;
    }
    int digits_enqueued(uint32_t c, hipStream_t st) override {
      ZK_HIP(hipEventRecord(ev[c], st));
      std::lock_guard<std::mutex> lk(mu);
      digits = c + 1;
      cv.notify_all();
      return ZK_OK;
    }
    ~Feed() override {
      for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
  } feed;
  feed.n_chunks = (uint32_t)n_chunks;
  feed.cuts = cuts.data();
  feed.sc[0] = S->sc[0];
  feed.sc[1] = S->sc[1];
  feed.ev.reserve(n_chunks);
  for (uint64_t c = 0; c < n_chunks; ++c) {
    hipEvent_t e;
    ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    feed.ev.push_back(e);
  }
  const uint64_t b_lo = base_offset < n_bases ? base_offset : n_bases;
  static const bool trace = std::getenv("MI355ZK_TRACE_HOST") != nullptr;  // timeline of the streamed call on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  auto copy_fn = [&]() {
    auto fail = [&] { std::lock_guard<std::mutex> lk(feed.mu); feed.copy_failed = true; feed.cv.notify_all(); };
    if (hipSetDevice(dev) != hipSuccess) { fail(); return; }
    uint64_t b_done = b_lo;  // bases [b_lo, b_done) are on the device
    for (uint64_t c = 0; c < n_chunks; ++c) {
      if (c >= 2) {
        {
          std::unique_lock<std::mutex> lk(feed.mu);
          feed.cv.wait(lk, [&] { return feed.abort_copy || feed.digits >= c - 1; });
          if (feed.abort_copy) return;
        }
        if (hipEventSynchronize(feed.ev[c - 2]) != hipSuccess) { fail(); return; }
      } else {
        std::lock_guard<std::mutex> lk(feed.mu);
        if (feed.abort_copy) return;
      }
      const uint64_t lo = cuts[c], hi = cuts[c + 1];
      hipError_t e = hipMemcpyAsync(S->sc[c & 1], scalars + lo * 4, (hi - lo) * 32, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess && upload_bases) {
        uint64_t b_hi = base_offset + rank_of(hi);
        if (b_hi > n_bases) b_hi = n_bases;
        if (b_hi > b_done) {
          e = hipMemcpyAsync((char*)d_bases + b_done * bsz, bases + b_done * bsz, (b_hi - b_done) * bsz, hipMemcpyHostToDevice, S->copy);
          b_done = b_hi;
        }
      }
      if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
      if (e != hipSuccess) { fail(); return; }
      if (trace) std::fprintf(stderr, "[mi355zk] host entry: chunk %llu (%llu exponents) uploaded at %.2f ms\n", (unsigned long long)c, (unsigned long long)(hi - lo), ms_now());
      std::lock_guard<std::mutex> lk(feed.mu);
      feed.copied = c + 1;
      feed.cv.notify_all();
    }
    if (upload_bases && entry) {  // the rest of the vector (not needed by this call) completes the cache entry
      hipError_t e = hipSuccess;
      if (b_lo > 0) e = hipMemcpyAsync(d_bases, bases, b_lo * bsz, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess && b_done < n_bases)
        e = hipMemcpyAsync((char*)d_bases + b_done * bsz, bases + b_done * bsz, (n_bases - b_done) * bsz, hipMemcpyHostToDevice, S->copy);
      if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
      if (e != hipSuccess) fail();
    }
  };

  uint64_t result_xyz[jac_words];
  {
    // a call that evaluates no exponent returns the reference's Projective::zero() = (0, 1, 0) (ec.rs:229-235), as msm_device does
    using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
    static_assert(sizeof(J) == sizeof result_xyz, "Jacobian layout");
    const J zero = J::zero();
    std::memcpy(result_xyz, &zero, sizeof result_xyz);
  }
  int result = ZK_OK;
  long long err_idx = -1;
  bool aborted = false;
  if (n_chunks > 0) {
    std::thread copier(copy_fn);
    // a vector pinned WITH TABLES, already on the device, in a call that is not cut: table mode
    // (not for a handful of exponents over a long vector -- the prover's input multiexps over its 2^20-point a / b queries: the
    // table's window width comes from the VECTOR's length, and zeroing + reducing 2^19 buckets for a few points costs more than the
    // plain call, which picks its window from n)
    const bool table_pays = n * 8 >= n_bases;
    const void* d_table = (n_chunks == 1 && entry && !fill && wgroups == 1 && table_pays) ? bases_table<GROUP>(entry, S->compute) : nullptr;
    result = msm_dev_entry<GROUP>(d_table ? d_table : d_bases, n_bases, base_offset, nullptr, n_scalars, density, density_bits, (void*)S->compute, result_xyz,
                                  wgroups, wgroup, 0, &feed, d_table != nullptr);
    err_idx = t_last_err_index;
    if (trace) std::fprintf(stderr, "[mi355zk] host entry: result at %.2f ms (%llu chunks)\n", ms_now(), (unsigned long long)n_chunks);
    {
      std::lock_guard<std::mutex> lk(feed.mu);
      // a call that failed before it had taken every chunk leaves the copy thread waiting: release it
      aborted = result != ZK_OK && result != ZK_ERR_UNEXPECTED_EOF && feed.digits < n_chunks;
      feed.abort_copy = aborted;
      feed.cv.notify_all();
    }
    copier.join();
    if (feed.copy_failed) result = ZK_ERR_DEVICE;
  } else if (upload_bases && entry && n_bases) {
    // nothing to evaluate, but the entry was created: fill it
    hipError_t e = hipMemcpyAsync(d_bases, bases, n_bases * bsz, hipMemcpyHostToDevice, S->copy);
    if (e == hipSuccess) e = hipStreamSynchronize(S->copy);
    if (e != hipSuccess) result = ZK_ERR_DEVICE;
    if (result == ZK_OK && P.eof_index >= 0) { result = ZK_ERR_UNEXPECTED_EOF; err_idx = P.eof_index; }
  } else if (P.eof_index >= 0) {
    result = ZK_ERR_UNEXPECTED_EOF;
    err_idx = P.eof_index;
  }
  guard.ok = result != ZK_ERR_DEVICE && !feed.copy_failed && !(fill && aborted);  // an aborted streamed upload is incomplete
  t_last_err_index = err_idx;
  if (result != ZK_OK && result != ZK_ERR_UNEXPECTED_EOF) return result;
  std::memcpy(out_xyz, result_xyz, sizeof result_xyz);
  return result;
}

// ------------------------------------------------------------------------------------------------
// SINGLE-PROCESS MULTI-GPU MODE.  The consumer this library is a drop-in for is ONE Rust process (phase2/src/bin/prove.rs ->
// bellman/src/groth16/prover.rs:250-298 -> multiexp.rs:330-355), so the 8 GPUs of a node must be reachable through the C ABI, not
// only through one rank per GPU (shard.py).  mi355zk_init(ids, n > 1) records a DEVICE SET; a host-buffer multiexp of at least
// 2^MI355ZK_MULTI_MIN_LOG exponents (default 20) is then cut into cells -- contiguous POINT RANGES (cut at multiples of 32
// exponents so that density words are not shared), optionally x groups of scalar windows -- and every cell is one msm_host_run on
// its own device from its own host thread: its exponents cross ITS PCIe link while its kernels run (the streamed upload above), its
// base vector is cached on that device when the caller pinned it.  The N Jacobian partials (96 / 192 B) come back to the host --
// SURVEY 8e's "or D2H of 8 records": inside one process there is nothing for RCCL to do -- and are joined there with the rule
// shard.exchange defines: a failing cell's error carries its GLOBAL exponent index, the lowest index wins, Eof (planned for the whole
// call) before identity at one index.  Smaller calls run whole, on the devices of the set in turn (the prover's eight concurrent
// multiexps spread over the node).
// Why point ranges and not shard.py's window groups: a rank of shard.py holds its exponents in HBM; here every cell uploads its own,
// and a window-group cell would upload ALL exponents of its range over its link (2^26 on 8 devices: 1 GiB per device against 256 MiB).
// MI355ZK_MULTI_PLAN="PxW" forces P point ranges x W window groups (P * W <= devices) for experiments and for the tests.
std::mutex g_devset_mu;
std::vector<int> g_devset;                 // HIP device ids of the set (a test may repeat one id: logical devices sharing a GPU)
std::atomic<unsigned> g_devset_turn{0};

std::vector<int> devset_snapshot() {
  std::lock_guard<std::mutex> lk(g_devset_mu);
  return g_devset;
}

// a cell / range / worker body run so that nothing is thrown out of a host thread or across the C ABI (std::bad_alloc from a
// staging vector, a std::system_error from a lock): the unit fails as a device error
template <class Fn>
void run_guarded(int& rc, Fn&& fn) noexcept {
  try {
    fn();
  } catch (...) {
    rc = ZK_ERR_DEVICE;
  }
}


// bases consumed by exponents [0, i) of a planned call (prefix popcount of the density map; i itself under FullDensity)
uint64_t density_rank(const DensityPlan& P, const uint32_t* density, uint64_t i) {
  if (density == nullptr || i == 0) return density == nullptr ? i : 0;
  const uint64_t w = i >> 5;
  if (w >= P.prefix.size()) {  // i == n on a word boundary past the last planned word
    const uint64_t lw = P.prefix.size() - 1;
    uint32_t v = density[lw];
    if ((lw + 1) * 32 > P.n) v &= (P.n & 31) ? ((1u << (P.n & 31)) - 1u) : 0xffffffffu;
    return P.prefix[lw] + (uint32_t)__builtin_popcount(v);
  }
  uint64_t r = P.prefix[w];
  if (i & 31) r += (uint32_t)__builtin_popcount(density[w] & ((1u << (i & 31)) - 1u));
  return r;
}

template <int GROUP>
int msm_host_multi(const std::vector<int>& devs, const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars,
                   size_t n_scalars, const uint32_t* density, size_t density_bits, uint64_t* out_xyz) {
  using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
  t_last_err_index = -1;
  DensityPlan P;
  int rc = plan_density(n_bases, base_offset, n_scalars, density, density_bits, &P);
  if (rc) return rc;
  const uint64_t n = P.eof_index >= 0 ? (uint64_t)P.eof_index : P.n;  // exponents before the first Eof
  // ---- the plan: point ranges x window groups
  uint32_t pg = (uint32_t)devs.size(), wg = 1;
  if (const char* env = std::getenv("MI355ZK_MULTI_PLAN")) {
    unsigned a = 0, b = 0;
    if (std::sscanf(env, "%ux%u", &a, &b) == 2 && a >= 1 && b >= 1 && (size_t)a * b <= devs.size()) { pg = a; wg = b; }
  }
  if (wg > 1) {  // the window count of the range's geometry must divide (choose_geom takes care of that; W == 0: no such layout)
    uint32_t c = 0, W = 0;
    msm_geometry((n + pg - 1) / pg, wg, &c, &W);
    if (W == 0 || W % wg) wg = 1;
  }
  while (pg > 1 && n / pg < 32) --pg;
  std::vector<uint64_t> cut(pg + 1, 0);
  for (uint32_t r = 1; r < pg; ++r) cut[r] = ((n * r / pg) + 31) & ~31ull;
  cut[pg] = n;
  struct Cell {
    int dev = 0;
    uint64_t lo = 0, hi = 0;
    uint32_t wgi = 0;
    int rc = ZK_OK;
    long long err = -1;
    J part;
  };
  std::vector<Cell> cells;
  for (uint32_t r = 0; r < pg; ++r)
    for (uint32_t g = 0; g < wg; ++g) {
      if (cut[r + 1] <= cut[r]) continue;
      Cell c;
      c.dev = devs[cells.size() % devs.size()];
      c.lo = cut[r];
      c.hi = cut[r + 1];
      c.wgi = g;
      c.part = J::zero();
      cells.push_back(c);
    }
  static const bool trace = std::getenv("MI355ZK_TRACE_HOST") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto run_cell = [&](Cell& c) {
    if (hipSetDevice(c.dev) != hipSuccess) { c.rc = ZK_ERR_DEVICE; return; }
    // The cell sees only the SLICE of the base vector its exponents consume -- [boff, boff + used) -- as a vector of its own: that is what
    // its device allocates, uploads and (inside a pinned vector) keeps: 2^26 G1 points on 8 devices are 512 MiB per device, not 4 GiB
    // (SURVEY 8e).  The ranges end before the first exponent without a base (the Eof is planned above for the whole call), so the slice
    // holds every base the cell asks for.
    constexpr size_t bsz = GROUP == 1 ? 64 : 128;
    const uint64_t boff = base_offset + density_rank(P, density, c.lo);
    const uint64_t used = density_rank(P, density, c.hi) - density_rank(P, density, c.lo);
    struct ParentScope {
      explicit ParentScope(const void* p) { t_bases_parent = p; }
      ~ParentScope() { t_bases_parent = nullptr; }
    } parent_scope(bases);
    c.rc = msm_host_run<GROUP>(bases + boff * bsz, used, 0, scalars + c.lo * 4, c.hi - c.lo, density ? density + (c.lo >> 5) : nullptr,
                               density ? c.hi - c.lo : 0, reinterpret_cast<uint64_t*>(&c.part), wg, c.wgi);
    c.err = t_last_err_index;
    if (trace)
      std::fprintf(stderr, "[mi355zk] multi: cell [%llu, %llu) window group %u/%u on device %d: rc %d at %.2f ms\n", (unsigned long long)c.lo,
                   (unsigned long long)c.hi, c.wgi, wg, c.dev, c.rc,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < cells.size(); ++started) th.emplace_back([&, started] { run_guarded(cells[started].rc, [&] { run_cell(cells[started]); }); });
    } catch (const std::exception&) {
      // (no more host threads to be had: the cells that did not get one run here, one after the other -- nothing is thrown across the C ABI)
    }
    if (!cells.empty()) run_guarded(cells[0].rc, [&] { run_cell(cells[0]); });
    for (size_t i = started; i < cells.size(); ++i) run_guarded(cells[i].rc, [&] { run_cell(cells[i]); });
    for (auto& t : th) t.join();
  }
  // ---- the join.  Device failures first, then a non-canonical exponent (bad arguments: the single-device call reports it before
  // anything else too), then the Source errors by global exponent index.
  J total = J::zero();
  long long bad_idx = -1, ident_idx = -1;
  for (Cell& c : cells) {
    if (c.rc < 0) return c.rc;
    if (c.rc == ZK_ERR_BAD_ARGS) {
      const long long g = c.err >= 0 ? c.err + (long long)c.lo : -1;
      if (bad_idx < 0 || (g >= 0 && g < bad_idx)) bad_idx = g >= 0 ? g : bad_idx;
      if (g < 0) { t_last_err_index = -1; return ZK_ERR_BAD_ARGS; }
    } else if (c.rc == ZK_ERR_UNEXPECTED_IDENTITY) {
      const long long g = c.err + (long long)c.lo;
      if (ident_idx < 0 || g < ident_idx) ident_idx = g;
    } else if (c.rc != ZK_OK) {
      return ZK_ERR_DEVICE;  // (a cell never reports Eof: the ranges end before the first exponent without a base)
    }
  }
  if (bad_idx >= 0) { t_last_err_index = bad_idx; return ZK_ERR_BAD_ARGS; }
  if (ident_idx >= 0) { t_last_err_index = ident_idx; return ZK_ERR_UNEXPECTED_IDENTITY; }  // (every exponent before the Eof has a lower index)
  for (Cell& c : cells) jac_add(total, c.part);
  std::memcpy(out_xyz, &total, sizeof total);
  if (P.eof_index >= 0) { t_last_err_index = P.eof_index; return ZK_ERR_UNEXPECTED_EOF; }
  return ZK_OK;
}

// the host-buffer entry points: whole on one device, or cut into cells over the device set
template <int GROUP>
int msm_host_entry(const uint8_t* bases, size_t n_bases, size_t base_offset, const uint64_t* scalars, size_t n_scalars,
                   const uint32_t* density, size_t density_bits, uint64_t* out_xyz) {
  const std::vector<int> devs = devset_snapshot();
  if (devs.size() <= 1) return msm_host_run<GROUP>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  if (!out_xyz || (n_scalars && !scalars) || (n_bases && !bases) || n_bases >= (1ull << 31) || n_scalars >= (1ull << 31)) {
    t_last_err_index = -1;
    return ZK_ERR_BAD_ARGS;
  }
  const char* env = std::getenv("MI355ZK_MULTI_MIN_LOG");  // (read per call: the tests lower it)
  const int min_log = env ? std::atoi(env) : 20;
  if (n_scalars >= (1ull << (min_log < 0 ? 0 : min_log > 30 ? 30 : min_log)) && n_scalars >= 64)
    return msm_host_multi<GROUP>(devs, bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
  // A short call runs whole on ONE device: the one that already holds the (pinned) vector if there is one -- the device copy of a
  // parameter vector is then made once per process, not once per device of the set (8 x 4 GiB for a 2^26-point CRS) -- else the next in
  // turn, so that the prover's concurrent multiexps over its different vectors spread over the node on first touch.
  int pick = -1;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc)
      if (e->host == bases && e->n == n_bases && e->group == GROUP && std::find(devs.begin(), devs.end(), e->dev) != devs.end()) { pick = e->dev; break; }
  }
  if (pick < 0) pick = devs[g_devset_turn.fetch_add(1) % devs.size()];
  DeviceGuard guard;
  ZK_HIP(hipSetDevice(pick));
  return msm_host_run<GROUP>(bases, n_bases, base_offset, scalars, n_scalars, density, density_bits, out_xyz);
}

// ------------------------------------------------------------------------------------------------
// batch_exp on HOST buffers, over the device set: what `MPCParameters::contribute` (phase2/src/parameters.rs:423-470: every point of L
// and H times delta^-1) and powersoftau's `batch_exp` (batched_accumulator.rs:1130-1181) are to a single-process caller.  The points are
// independent, so they shard by CONTIGUOUS POINT RANGE with no exchange at all (shard.batch_exp_sharded is the
// one-process-per-GPU form): device d of mi355zk_init's set takes range d -- upload, the batch_exp kernels, download -- from its own
// host thread; with one device the whole vector is one range.  Ranges are worked off in pieces of 2^18 points (a piece's buffers
// come from the grow-only pool, so a 2^26-point vector does not allocate 10 GiB).  g2_trusted: the promise flag of batch_exp_dev.
template <class F>
int batch_exp_host(uint8_t* out, const uint8_t* bases, const uint64_t* scalars, size_t n, int same_scalar, bool g2_trusted) {
  if ((n && (!out || !bases)) || !scalars) return ZK_ERR_BAD_ARGS;
  if (n == 0) return ZK_OK;
  if (n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  constexpr size_t rec = sizeof(Affine<F>);
  std::vector<int> devs = devset_snapshot();
  if (devs.empty()) {
    int cur = 0;
    ZK_HIP(hipGetDevice(&cur));
    devs.push_back(cur);
  }
  size_t parts = devs.size();
  while (parts > 1 && n / parts < 1024) --parts;
  std::vector<int> rcs(parts, ZK_OK);
  auto run_range = [&](size_t d) {
    const size_t lo = n * d / parts, hi = n * (d + 1) / parts;
    if (hipSetDevice(devs[d]) != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    StageLease stage_lease;
    HostStage* S = host_stage(devs[d], &stage_lease);
    if (S == nullptr) { rcs[d] = ZK_ERR_DEVICE; return; }
    // pieces of 2^18 points, double-buffered
    const size_t piece = (size_t)1 << 18;
    const size_t m_max = hi - lo < piece ? hi - lo : piece;
    const size_t in_bytes = (m_max * rec + 255) & ~(size_t)255;
    const size_t sc_bytes = same_scalar ? 256 : ((m_max * 32 + 255) & ~(size_t)255);
    DensityPool::Lease buf;   // (the grow-only device buffer pool of the host-buffer entry points)
    int rc = buf.acquire(devs[d], 4 * in_bytes + 2 * sc_bytes, S->compute);
    if (rc) { rcs[d] = rc; return; }
    char* base = (char*)buf.b->p;
    char* d_in[2] = {base, base + in_bytes};
    char* d_out[2] = {base + 2 * in_bytes, base + 3 * in_bytes};
    char* d_sc[2] = {base + 4 * in_bytes, base + 4 * in_bytes + sc_bytes};
    hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    auto cleanup = [&] {
      (void)hipStreamSynchronize(S->copy);
      (void)hipStreamSynchronize(S->compute);
      for (int k = 0; k < 2; ++k) { if (up[k]) (void)hipEventDestroy(up[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    };
    auto fail = [&](hipError_t e) {
      std::fprintf(stderr, "[mi355zk] batch_exp (host buffers): HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
      rcs[d] = ZK_ERR_DEVICE;
      cleanup();
    };
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2; ++k)
      if ((e = hipEventCreateWithFlags(&up[k], hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming)) != hipSuccess) return fail(e);
    if (same_scalar && (e = hipMemcpyAsync(d_sc[0], scalars, 32, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return fail(e);
    // ONE host thread keeps the device busy although copies from / to PAGEABLE host memory block it: the kernels of piece i + 1 are
    // always queued before the thread waits for piece i's download, and piece i + 2 is uploaded (into the buffer piece i's kernels
    // have finished with: its download has just returned) while piece i + 1 computes.
    const size_t n_pieces = (hi - lo + piece - 1) / piece;
    auto upload_and_launch = [&](size_t i) -> bool {
      const size_t p0 = lo + i * piece, m = hi - p0 < piece ? hi - p0 : piece;
      const int k = (int)(i & 1);
      if ((e = hipMemcpyAsync(d_in[k], bases + p0 * rec, m * rec, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return false;
      if (!same_scalar && (e = hipMemcpyAsync(d_sc[k], scalars + p0 * 4, m * 32, hipMemcpyHostToDevice, S->copy)) != hipSuccess) return false;
      if ((e = hipEventRecord(up[k], S->copy)) != hipSuccess || (e = hipStreamWaitEvent(S->compute, up[k], 0)) != hipSuccess) return false;
      rc = batch_exp<F>(d_out[k], d_in[k], 0, same_scalar ? d_sc[0] : d_sc[k], same_scalar, m, (void*)S->compute, nullptr, false, g2_trusted);
      if (rc) return false;
      return (e = hipEventRecord(done[k], S->compute)) == hipSuccess;
    };
    auto bail = [&] {
      if (rc) { rcs[d] = rc; cleanup(); }
      else fail(e);
    };
    for (size_t i = 0; i < 2 && i < n_pieces; ++i)
      if (!upload_and_launch(i)) return bail();
    for (size_t i = 0; i < n_pieces; ++i) {
      const size_t p0 = lo + i * piece, m = hi - p0 < piece ? hi - p0 : piece;
      const int k = (int)(i & 1);
      if ((e = hipStreamWaitEvent(S->copy, done[k], 0)) != hipSuccess) return fail(e);
      if ((e = hipMemcpyAsync(out + p0 * rec, d_out[k], m * rec, hipMemcpyDeviceToHost, S->copy)) != hipSuccess) return fail(e);
      if ((e = hipStreamSynchronize(S->copy)) != hipSuccess) return fail(e);   // piece i is on the host; its buffers are free
      if (i + 2 < n_pieces && !upload_and_launch(i + 2)) return bail();
    }
    if ((e = hipStreamSynchronize(S->compute)) != hipSuccess) return fail(e);
    cleanup();
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < parts; ++started) th.emplace_back([&, started] { run_guarded(rcs[started], [&] { run_range(started); }); });
    } catch (const std::exception&) {
    }
    run_guarded(rcs[0], [&] { run_range(0); });
    for (size_t d = started; d < parts; ++d) run_guarded(rcs[d], [&] { run_range(d); });
    for (auto& t : th) t.join();
  }
  for (int rc : rcs)
    if (rc != ZK_OK) return rc;
  return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// dense_multiexp / merge_pairs on HOST buffers, over the device set: the verification multiexps of the ceremony code (SURVEY 8f row 2:
// powersoftau/src/utils.rs:112-135, 189-292; phase2/src/utils.rs:59-105) for a single-process caller.  sum_i rho_i * v_i is linear in the
// points, so the vectors are cut into pieces of 2^22 points, every piece is one device call (msm_g*_dense_device: digits and partition
// shared by the two sums of merge_pairs) and the Jacobian partials are added on the host.  The pieces are dealt to TWO host threads per
// device of mi355zk_init's set (one piece uploads -- pageable copies block their thread -- while the other computes); v2 == nullptr:
// dense_multiexp.  No Source errors (infinity bases add nothing: the reference's dense contract).
template <int GROUP>
int dense_host(const uint8_t* v1, const uint8_t* v2, const uint64_t* rho, size_t n, uint64_t* out_s, uint64_t* out_sx) {
  using J = typename std::conditional<GROUP == 1, G1Jacobian, G2Jacobian>::type;
  constexpr size_t rec = GROUP == 1 ? 64 : 128;
  if (!out_s || (v2 && !out_sx) || (n && (!v1 || !rho)) || n >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  J total = J::zero(), total2 = J::zero();
  if (n > 0) {
    std::vector<int> devs = devset_snapshot();
    if (devs.empty()) {
      int cur = 0;
      ZK_HIP(hipGetDevice(&cur));
      devs.push_back(cur);
    }
    size_t piece = (size_t)1 << 22;
    if (const char* env = std::getenv("MI355ZK_DENSE_PIECE_TEST")) {   // (test hook, read per call: points per piece, so that the cut can be held against the oracle)
      const size_t v = (size_t)std::strtoull(env, nullptr, 10);
      if (v >= 16) piece = v;
    }
    const size_t n_pieces = (n + piece - 1) / piece;
    size_t workers = 2 * devs.size();
    if (workers > n_pieces) workers = n_pieces;
    struct Part { int rc = ZK_OK; J s, sx; };
    std::vector<Part> parts(workers);
    for (auto& pt : parts) { pt.s = J::zero(); pt.sx = J::zero(); }
    std::atomic<size_t> next{0};
    auto work = [&](size_t wk) {
      Part& P = parts[wk];
      const int dev = devs[wk % devs.size()];
      if (hipSetDevice(dev) != hipSuccess) { P.rc = ZK_ERR_DEVICE; return; }
      StageLease stage_lease;
      HostStage* S = host_stage(dev, &stage_lease);
      if (S == nullptr) { P.rc = ZK_ERR_DEVICE; return; }
      const size_t m_max = n < piece ? n : piece;
      const size_t vb = ((m_max + 16) * rec + 255) & ~(size_t)255;
      DensityPool::Lease buf;
      if (int rc = buf.acquire(dev, (v2 ? 2 : 1) * vb + m_max * 32, S->compute)) { P.rc = rc; return; }
      char* d_v1 = (char*)buf.b->p;
      // power_pairs (utils.rs:133-135) is merge_pairs(v[0 .. n-1], v[1 .. n]): the two vectors are ONE array seen at two offsets, and
      // uploading it twice would double the PCIe traffic of a call the link already bounds -- a v2 that starts `shift` (<= 16)
      // records into v1 shares v1's upload
      // (the addresses are compared as integers -- the two pointers need not belong to one array -- and the vectors must really overlap:
      // shift <= n; two separate short arrays that happen to sit within 16 records of each other are uploaded separately: ADVICE r4)
      const uintptr_t a1 = (uintptr_t)v1, a2 = (uintptr_t)v2;
      const size_t shift = (v2 && a2 >= a1 && (a2 - a1) % rec == 0 && (a2 - a1) / rec <= 16 && (a2 - a1) / rec <= n) ? (size_t)((a2 - a1) / rec) : (size_t)-1;
      const bool shared = shift != (size_t)-1;
      char* d_v2 = v2 ? (shared ? d_v1 + shift * rec : d_v1 + vb) : nullptr;
      char* d_rho = d_v1 + (v2 ? 2 : 1) * vb;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n_pieces) break;
        const size_t p0 = i * piece, m = n - p0 < piece ? n - p0 : piece;
        hipError_t e = hipMemcpyAsync(d_v1, v1 + p0 * rec, (m + (shared ? shift : 0)) * rec, hipMemcpyHostToDevice, S->compute);
        if (e == hipSuccess && v2 && !shared) e = hipMemcpyAsync(d_v2, v2 + p0 * rec, m * rec, hipMemcpyHostToDevice, S->compute);
        if (e == hipSuccess) e = hipMemcpyAsync(d_rho, rho + p0 * 4, m * 32, hipMemcpyHostToDevice, S->compute);
        if (e != hipSuccess) {
          std::fprintf(stderr, "[mi355zk] dense multiexp (host buffers): HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
          P.rc = ZK_ERR_DEVICE;
          return;
        }
        J a = J::zero(), b = J::zero();
        const int rc = GROUP == 1 ? msm_g1_dense_device(d_v1, d_v2, d_rho, m, S->compute, reinterpret_cast<uint64_t*>(&a), v2 ? reinterpret_cast<uint64_t*>(&b) : nullptr)
                                  : msm_g2_dense_device(d_v1, d_v2, d_rho, m, S->compute, reinterpret_cast<uint64_t*>(&a), v2 ? reinterpret_cast<uint64_t*>(&b) : nullptr);
        if (rc != ZK_OK) { P.rc = rc; return; }
        jac_add(P.s, a);
        if (v2) jac_add(P.sx, b);
      }
    };
    {
      DeviceGuard guard;
      std::vector<std::thread> th;
      size_t started = 1;
      try {
        for (; started < workers; ++started) th.emplace_back([&, started] { run_guarded(parts[started].rc, [&] { work(started); }); });
      } catch (const std::exception&) {
      }
      run_guarded(parts[0].rc, [&] { work(0); });   // (a worker takes pieces until none is left: the ones that got no thread are covered by the others)
      for (auto& t : th) t.join();
    }
    for (auto& pt : parts) {
      if (pt.rc != ZK_OK) return pt.rc;
      jac_add(total, pt.s);
      if (v2) jac_add(total2, pt.sx);
    }
  }
  std::memcpy(out_s, &total, sizeof total);
  if (v2) std::memcpy(out_sx, &total2, sizeof total2);
  return ZK_OK;
}

// best_fft / the domain operations on a HOST array (what a bellman shim calls with `&mut [Scalar<E>]`): upload, transform in place
// on the device, copy back.  Device buffer and stream are leased from the pools of the host-buffer entry points -- round 2
// hipMalloc'ed and hipFree'd per call (both synchronise the whole device, i.e. every other thread's multiexp) and ran on the null
// stream.  `a` is written by the final copy only: on a device failure (rc < 0) the caller's array is untouched and it can fall
// back to its own serial_fft.
int ntt_host(uint64_t* a, uint32_t log_n, int op, const uint64_t* omega) {
  if (!a) return ZK_ERR_BAD_ARGS;
  if (log_n > 28) return ZK_ERR_BAD_ARGS;
  const size_t bytes = (size_t)32 << log_n;
  int dev = 0;
  ZK_HIP(hipGetDevice(&dev));
  StageLease stage_lease;
  HostStage* S = host_stage(dev, &stage_lease);
  if (S == nullptr) return ZK_ERR_DEVICE;
  DensityPool::Lease buf;   // (a grow-only device buffer pool; the lease synchronises the stream before the buffer is handed on)
  int rc = buf.acquire(dev, bytes, S->compute);
  if (rc) return rc;
  void* d = buf.b->p;
  ZK_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, S->compute));
  if (omega) {
    Fr w;
    std::memcpy(&w, omega, 32);
    rc = ntt_run((Fr*)d, log_n, w, S->compute);
  } else {
    rc = domain_op_dev((Fr*)d, log_n, op, S->compute);
  }
  if (rc != ZK_OK) return rc;
  ZK_HIP(hipStreamSynchronize(S->compute));  // a failed kernel surfaces here, before the caller's array is touched
  ZK_HIP(hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, S->compute));
  ZK_HIP(hipStreamSynchronize(S->compute));
  return ZK_OK;
}



// out[r] = sum_{t in [row_ptr[r], row_ptr[r+1])} coeff[t] * bases[col[t]], affine (QAP evaluation, parameters.rs:225-294)
// CSR sanity on the device: flag |= 1 if some col[t] >= n_bases, |= 2 if row_ptr is not 0 = row_ptr[0] <= ... <= row_ptr[n_rows] = nnz
__global__ void __launch_bounds__(256) csr_check_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, uint64_t n_rows,
                                                       uint64_t nnz, uint64_t n_bases, uint32_t* __restrict__ flag) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += stride)
    if (col[t] >= n_bases) bad |= 1u;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
    const uint32_t v = row_ptr[r];
    if ((r == 0 && v != 0) || (r == n_rows && v != nnz) || (r < n_rows && v > row_ptr[r + 1])) bad |= 2u;
  }
  if (bad) atomicOr(flag, bad);
}

template <class F>
int sparse_matvec(void* d_out, const void* d_bases, size_t n_bases, const uint32_t* d_row_ptr, const uint32_t* d_col, const void* d_coeffs,
                         size_t n_rows, size_t nnz, void* stream, int group, bool g2_trusted, void* d_scratch, size_t scratch_bytes) {
  // d_scratch: the caller's buffer for the term products (the host-buffer form leases it with its other buffers: hipMalloc / hipFree per
  // call synchronise the whole device, i.e. every other thread's multiexp): room for the terms, 256 B of flags and one byte per base
  if (!d_out || !d_row_ptr || (nnz && (!d_bases || !d_col || !d_coeffs)) || n_rows >= (1ull << 31) || nnz >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  if (n_rows == 0) return ZK_OK;
  // G2 without the caller's promise: when the bases are reused (nnz >= 2 n_bases: a circuit has ~3 terms per Lagrange coefficient), ONE
  // membership test per base (psi(P) == mu P: 127 doublings + 68 additions) buys the psi split (a third fewer operations) and the
  // r - 1 shortcut for every term of a member, and only the terms of the other bases take the plain windows -- the result is the
  // reference's either way.  (2^20 bases, 2.9 M terms: 156 -> ~155 ms general coefficients, 90 -> ~61 ms with 90 % unit coefficients.)
  const bool by_member = std::is_same<F, Fq2>::value && !g2_trusted && nnz >= 2 * n_bases && n_bases > 0;
  const size_t terms_bytes = ((nnz ? nnz : 1) * sizeof(Affine<F>) + 255) & ~(size_t)255;
  Affine<F>* d_terms = nullptr;
  const bool own = d_scratch == nullptr || scratch_bytes < terms_bytes + 256 + (by_member ? n_bases : 0);
  if (own) ZK_HIP(hipMalloc(&d_terms, terms_bytes + 256 + (by_member ? n_bases : 0)));
  else d_terms = (Affine<F>*)d_scratch;
  uint8_t* d_member = by_member ? reinterpret_cast<uint8_t*>(d_terms) + terms_bytes + 256 : nullptr;
  {
    // the ABI cannot trust the index arrays: an out-of-range column would be an out-of-bounds gather in batch_exp
    uint32_t* d_flag = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_terms) + terms_bytes);
    uint32_t h_flag = 0;
    hipError_t e = hipMemsetAsync(d_flag, 0, 4, (hipStream_t)stream);
    if (e == hipSuccess) {
      const uint64_t work = nnz > n_rows + 1 ? nnz : n_rows + 1;
      hipLaunchKernelGGL(csr_check_kernel, dim3((unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                         d_row_ptr, d_col, (uint64_t)n_rows, (uint64_t)nnz, (uint64_t)n_bases, d_flag);
      e = hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess || h_flag) {
      if (own) (void)hipFree(d_terms);
      if (e != hipSuccess) ZK_HIP(e);
      return ZK_ERR_BAD_ARGS;
    }
  }
  int rc = ZK_OK;
  if constexpr (std::is_same<F, Fq2>::value)
    if (by_member) rc = g2_subgroup_flags(d_bases, n_bases, stream, d_member);
  if (rc == ZK_OK) rc = batch_exp<F>(d_terms, d_bases, 0, d_coeffs, 0, nnz, stream, d_col, /*shortcut_unit_scalars=*/true, g2_trusted, d_member);
  if (rc == ZK_OK)
    rc = group == 1 ? segsum_g1_device(d_terms, nnz, d_row_ptr, (uint32_t)n_rows, (hipStream_t)stream, d_out)
                    : segsum_g2_device(d_terms, nnz, d_row_ptr, (uint32_t)n_rows, (hipStream_t)stream, d_out);
  if (own) (void)hipFree(d_terms);
  return rc;
}

// The QAP evaluation on HOST buffers, over the device set (SURVEY 8f row 3 for a single-process caller: MPCParameters::new over a
// 2^20+-constraint circuit): the rows of the CSR matrix are independent, so device d takes the d-th contiguous ROW range -- its slice of
// (col, coeff), a row_ptr rebased to zero, and the WHOLE base vector (any row may name any Lagrange coefficient) -- and writes its rows
// of the output; no exchange.  The index arrays are validated on the device as in the _dev form; row_ptr[0] == 0, row_ptr[n_rows] == nnz
// and monotonicity across the cuts are checked here.
template <class F>
int sparse_matvec_host(uint8_t* out, const uint8_t* bases, size_t n_bases, const uint32_t* row_ptr, const uint32_t* col, const uint64_t* coeffs,
                              size_t n_rows, size_t nnz, int group, bool g2_trusted) {
  constexpr size_t rec = sizeof(Affine<F>);
  if (!row_ptr || (n_rows && !out) || (nnz && (!bases || !col || !coeffs)) || n_rows >= (1ull << 31) || nnz >= (1ull << 31) || n_bases >= (1ull << 31)) return ZK_ERR_BAD_ARGS;
  if (n_rows == 0) return ZK_OK;
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return ZK_ERR_BAD_ARGS;
  std::vector<int> devs = devset_snapshot();
  if (devs.empty()) {
    int cur = 0;
    ZK_HIP(hipGetDevice(&cur));
    devs.push_back(cur);
  }
  size_t parts = devs.size();
  while (parts > 1 && n_rows / parts < 128) --parts;
  // cuts of equal WEIGHT (rows + terms: a row costs a normalisation, a term an addition chain), found by one walk over row_ptr -- the
  // variables of a circuit are far from equally used (the constant ONE sits in most constraints)
  std::vector<size_t> cut(parts + 1, n_rows);
  cut[0] = 0;
  {
    const size_t weight = n_rows + nnz;
    size_t d = 1;
    for (size_t r = 0; r < n_rows && d < parts; ++r)
      while (d < parts && r + (size_t)row_ptr[r] >= weight * d / parts) cut[d++] = r;
  }
  std::vector<int> rcs(parts, ZK_OK);
  auto run_range = [&](size_t d) {
    const size_t r0 = cut[d], r1 = cut[d + 1];
    if (r1 == r0) return;
    const uint32_t t0 = row_ptr[r0], t1 = row_ptr[r1];
    if (t1 < t0) { rcs[d] = ZK_ERR_BAD_ARGS; return; }
    const size_t rows = r1 - r0, terms = t1 - t0;
    if (hipSetDevice(devs[d]) != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    StageLease stage_lease;
    HostStage* S = host_stage(devs[d], &stage_lease);
    if (S == nullptr) { rcs[d] = ZK_ERR_DEVICE; return; }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_bases = 0, o_out = o_bases + al((n_bases ? n_bases : 1) * rec), o_rp = o_out + al(rows * rec), o_col = o_rp + al((rows + 1) * 4),
                 o_cf = o_col + al((terms ? terms : 1) * 4), o_scr = o_cf + al((terms ? terms : 1) * 32),
                 scr_bytes = al((terms ? terms : 1) * rec) + 256 + al(n_bases), total = o_scr + scr_bytes;
    DensityPool::Lease buf;
    if (int rc = buf.acquire(devs[d], total, S->compute)) { rcs[d] = rc; return; }
    char* base = (char*)buf.b->p;
    std::vector<uint32_t> rp(rows + 1);
    for (size_t r = 0; r <= rows; ++r) {
      const uint32_t v = row_ptr[r0 + r];
      if (v < t0 || v > t1) { rcs[d] = ZK_ERR_BAD_ARGS; return; }   // (monotone inside the range is checked on the device)
      rp[r] = v - t0;
    }
    hipError_t e = hipSuccess;
    if (n_bases) e = hipMemcpyAsync(base + o_bases, bases, n_bases * rec, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess) e = hipMemcpyAsync(base + o_rp, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess && terms) e = hipMemcpyAsync(base + o_col, col + t0, terms * 4, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess && terms) e = hipMemcpyAsync(base + o_cf, coeffs + (size_t)t0 * 4, terms * 32, hipMemcpyHostToDevice, S->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(S->compute);   // (rp is a local vector)
    if (e != hipSuccess) { rcs[d] = ZK_ERR_DEVICE; return; }
    int rc = sparse_matvec<F>(base + o_out, base + o_bases, n_bases, (const uint32_t*)(base + o_rp), (const uint32_t*)(base + o_col), base + o_cf, rows, terms,
                              (void*)S->compute, group, g2_trusted, base + o_scr, scr_bytes);
    if (rc != ZK_OK) { rcs[d] = rc; return; }
    e = hipMemcpyAsync(out + r0 * rec, base + o_out, rows * rec, hipMemcpyDeviceToHost, S->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(S->compute);
    if (e != hipSuccess) rcs[d] = ZK_ERR_DEVICE;
  };
  {
    DeviceGuard guard;
    std::vector<std::thread> th;
    size_t started = 1;
    try {
      for (; started < parts; ++started) th.emplace_back([&, started] { run_guarded(rcs[started], [&] { run_range(started); }); });
    } catch (const std::exception&) {
    }
    run_guarded(rcs[0], [&] { run_range(0); });
    for (size_t d = started; d < parts; ++d) run_guarded(rcs[d], [&] { run_range(d); });
    for (auto& t : th) t.join();
  }
  for (int rc : rcs)
    if (rc != ZK_OK) return rc;
  return ZK_OK;
}

// ---- what api.hip's lifecycle / cache wrappers need of the state above
void devset_set(const std::vector<int>& set) {
  std::lock_guard<std::mutex> lk(g_devset_mu);
  g_devset = set;
}
int devset_count() {
  std::lock_guard<std::mutex> lk(g_devset_mu);
  return g_devset.empty() ? 1 : (int)g_devset.size();
}
int bases_cache_info(const void* host_bases, size_t* device_bytes, size_t* table_bytes) {
  size_t d = 0, t = 0;
  int found = 0;
  {
    std::lock_guard<std::mutex> lk(g_bc_mu);
    for (auto& e : g_bc)
      if ((e->host == host_bases || e->owner == host_bases) && e->ready) { d += e->bytes; t += e->table_bytes; found = 1; }
  }
  if (device_bytes) *device_bytes = d;
  if (table_bytes) *table_bytes = t;
  return found;
}

template int msm_dev_entry<1>(const void*, size_t, size_t, const void*, size_t, const uint32_t*, size_t, void*, uint64_t*, uint32_t, uint32_t, uint32_t, MsmChunks*, bool);
template int msm_dev_entry<2>(const void*, size_t, size_t, const void*, size_t, const uint32_t*, size_t, void*, uint64_t*, uint32_t, uint32_t, uint32_t, MsmChunks*, bool);
template int msm_host_entry<1>(const uint8_t*, size_t, size_t, const uint64_t*, size_t, const uint32_t*, size_t, uint64_t*);
template int msm_host_entry<2>(const uint8_t*, size_t, size_t, const uint64_t*, size_t, const uint32_t*, size_t, uint64_t*);
template int batch_exp_host<Fq>(uint8_t*, const uint8_t*, const uint64_t*, size_t, int, bool);
template int batch_exp_host<Fq2>(uint8_t*, const uint8_t*, const uint64_t*, size_t, int, bool);
template int dense_host<1>(const uint8_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*, uint64_t*);
template int dense_host<2>(const uint8_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*, uint64_t*);
template int sparse_matvec<Fq>(void*, const void*, size_t, const uint32_t*, const uint32_t*, const void*, size_t, size_t, void*, int, bool, void*, size_t);
template int sparse_matvec<Fq2>(void*, const void*, size_t, const uint32_t*, const uint32_t*, const void*, size_t, size_t, void*, int, bool, void*, size_t);
template int sparse_matvec_host<Fq>(uint8_t*, const uint8_t*, size_t, const uint32_t*, const uint32_t*, const uint64_t*, size_t, size_t, int, bool);
template int sparse_matvec_host<Fq2>(uint8_t*, const uint8_t*, size_t, const uint32_t*, const uint32_t*, const uint64_t*, size_t, size_t, int, bool);
}  // namespace zk
