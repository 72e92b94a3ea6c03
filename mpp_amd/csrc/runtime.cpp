// runtime.cpp -- HIP-event profiler, run-time RCCL binding, host-side helpers of the set-up passes.
#include "runtime.hpp"

#include <atomic>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>

#include <dlfcn.h>

#include <cstring>
#include <type_traits>

// RCCL is bound at run time (dlopen: single-GPU use never needs the library), through hand-written prototypes.  When the
// header is there at build time, the prototypes and the three enumerators this file passes as plain ints are checked against it
// -- a 1-rank all-reduce is the identity for any datatype / operation, so no single-GPU test could catch a wrong constant.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define MPPGPU_RCCL_HEADER_VERSION NCCL_VERSION_CODE
#else
#define MPPGPU_RCCL_HEADER_VERSION 0
#endif

namespace mpp {

const char *kclass_name(int k) {
  static const char *names[KC_COUNT] = {"spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce", "spmv_pc"};
  return (k >= 0 && k < KC_COUNT) ? names[k] : "?";
}

// ---------------- profiler: HIP events recorded on the solver's own stream around every launch group ----------------
void Profiler::begin(int cls) {
  open = enabled && ((mask >> cls) & 1u);
  if (!open) return;
  open = (calls[cls]++ % stride) == 0;
  if (!open) return;
  if (used == pool.size()) {
    if (pool.size() >= 8192) flush();
    else {
      Pair pr;
      HIP_CHECK(hipEventCreate(&pr.a));
      HIP_CHECK(hipEventCreate(&pr.b));
      pr.cls = cls;
      pool.push_back(pr);
    }
  }
  pool[used].cls = cls;
  HIP_CHECK(hipEventRecord(pool[used].a, stream));
}
void Profiler::end() {
  if (!open) return;
  open = false;
  HIP_CHECK(hipEventRecord(pool[used].b, stream));
  ++used;
}
void Profiler::flush() {
  if (!used) return;
  HIP_CHECK(hipStreamSynchronize(stream));
  for (size_t i = 0; i < used; ++i) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, pool[i].a, pool[i].b));
    launches[pool[i].cls] += 1;
    total_ms[pool[i].cls] += ms;
  }
  used = 0;
}
void Profiler::reset() {
  flush();
  for (int k = 0; k < KC_COUNT; ++k) { launches[k] = 0; total_ms[k] = 0.0; calls[k] = 0; }
}
Profiler::~Profiler() {
  for (auto &pr : pool) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
}

// ---------------- RCCL through dlopen: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy ----------------
namespace {
struct NcclUniqueId { char internal[128]; };
using fn_get_id = int (*)(NcclUniqueId *);
using fn_init_rank = int (*)(void **, int, NcclUniqueId, int);
using fn_allreduce = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
using fn_destroy = int (*)(void *);
using fn_sendrecv = int (*)(void *, size_t, int, int, void *, hipStream_t);   // ncclSend (const void*) / ncclRecv
using fn_group = int (*)();
using fn_errstr = const char *(*)(int);
constexpr int kNcclFloat64 = 8;  // ncclDataType_t ncclDouble (rccl.h)
constexpr int kNcclSum = 0;      // ncclRedOp_t ncclSum
constexpr int kNcclMax = 2;      // ncclRedOp_t ncclMax
using fn_version = int (*)(int *);
#if MPPGPU_RCCL_HEADER_VERSION
static_assert((int)ncclDouble == kNcclFloat64 && (int)ncclFloat64 == kNcclFloat64, "ncclDouble");
static_assert((int)ncclSum == kNcclSum && (int)ncclMax == kNcclMax, "ncclSum / ncclMax");
static_assert(sizeof(ncclUniqueId) == sizeof(NcclUniqueId) && alignof(ncclUniqueId) == alignof(NcclUniqueId), "ncclUniqueId");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) &&
                  sizeof(ncclRedOp_t) == sizeof(int), "handle and enum widths");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclAllReduce),
                           ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value,
              "ncclAllReduce");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclCommAbort), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommAbort");
static_assert(std::is_same<decltype(&ncclGetVersion), ncclResult_t (*)(int *)>::value, "ncclGetVersion");
#endif

void *open_rccl() {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    void *l = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (l) return l;
  }
  throw Error(std::string("cannot load RCCL (librccl.so.1): ") + dlerror());
}
template <typename F>
F sym(void *lib, const char *name) {
  void *p = dlsym(lib, name);
  if (!p) throw Error(std::string("RCCL symbol missing: ") + name);
  return reinterpret_cast<F>(p);
}
// the library that was loaded must be the generation the prototypes above were checked against: same major version as the
// header (NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch from 2.9 on), and 2.x at all
int checked_version(void *lib) {
  int v = 0;
  const int rc = sym<fn_version>(lib, "ncclGetVersion")(&v);
  if (rc != 0) throw Error("ncclGetVersion failed");
  const int hdr = MPPGPU_RCCL_HEADER_VERSION;
  if (v < 20000 || (hdr && v / 10000 != hdr / 10000))
    throw Error("RCCL " + std::to_string(v) + " was loaded at run time, but the bindings were checked against rccl.h " + std::to_string(hdr) +
                ": datatype / reduction enumerators and prototypes may differ");
  return v;
}
void nccl_check(void *lib, int rc, const char *what) {
  if (rc == 0) return;
  const char *msg = "?";
  if (void *p = dlsym(lib, "ncclGetErrorString")) msg = reinterpret_cast<fn_errstr>(p)(rc);
  throw Error(std::string(what) + " failed: " + msg);
}
}  // namespace

int Comm::runtime_version() {
  void *l = open_rccl();
  return checked_version(l);
}
void Comm::get_unique_id(void *id128) {
  void *l = open_rccl();
  (void)checked_version(l);
  NcclUniqueId id;
  nccl_check(l, sym<fn_get_id>(l, "ncclGetUniqueId")(&id), "ncclGetUniqueId");
  std::memcpy(id128, id.internal, 128);
}
void Comm::init(int nr, int r, const void *id128) {
  if (comm) throw Error("communicator already initialised");
  lib = open_rccl();
  version = checked_version(lib);
  NcclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  nccl_check(lib, sym<fn_init_rank>(lib, "ncclCommInitRank")(&comm, nr, id, r), "ncclCommInitRank");
  nranks = nr;
  rank = r;
  std::fprintf(stderr, "mppgpu: rank %d of %d: RCCL %d.%d.%d communicator ready\n", r, nr, version / 10000, (version / 100) % 100, version % 100);
}
void LocalGroup::abort() {
  std::lock_guard<std::mutex> lk(m);
  aborted = true;
  cv.notify_all();
}
void LocalGroup::allreduce(int rank, double *v, int count, bool is_max) {
  std::unique_lock<std::mutex> lk(m);
  if (aborted) throw Error("in-process group: another rank failed");
  for (int i = 0; i < count; ++i) slots[(size_t)rank * 64 + i] = v[i];
  const long long gen = generation;
  if (++arrived == nranks) {
    // the last arriver reduces in rank order and publishes the result in every rank's slots
    for (int i = 0; i < count; ++i) {
      double acc = slots[i];
      for (int r = 1; r < nranks; ++r) acc = is_max ? std::max(acc, slots[(size_t)r * 64 + i]) : acc + slots[(size_t)r * 64 + i];
      for (int r = 0; r < nranks; ++r) slots[(size_t)r * 64 + i] = acc;
    }
    arrived = 0;
    ++generation;
    cv.notify_all();
  } else {
    cv.wait(lk, [&] { return generation != gen || aborted; });
    if (generation == gen) throw Error("in-process group: another rank failed");
  }
  for (int i = 0; i < count; ++i) v[i] = slots[(size_t)rank * 64 + i];
}
void Comm::init_local(std::shared_ptr<LocalGroup> g, int r) {
  if (active()) throw Error("communicator already initialised");
  local = std::move(g);
  nranks = local->nranks;
  rank = r;
  HIP_CHECK(hipHostMalloc(&h_buf, 64 * sizeof(double), hipHostMallocDefault));
}
static void local_allreduce(Comm &c, double *dev, int count, hipStream_t s, bool is_max) {
  if (count > 64) throw Error("in-process all-reduce: more than 64 values");
  HIP_CHECK(hipMemcpyAsync(c.h_buf, dev, count * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  c.local->allreduce(c.rank, c.h_buf, count, is_max);
  HIP_CHECK(hipMemcpyAsync(dev, c.h_buf, count * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));
}
void Comm::abort_peers() {
  if (local) local->abort();
}
void Comm::allreduce_sum(double *dev, int count, hipStream_t s) {
  ++n_allreduce;
  if (local) { local_allreduce(*this, dev, count, s, false); return; }
  if (!comm) return;
  static fn_allreduce f = nullptr;
  if (!f) f = sym<fn_allreduce>(lib, "ncclAllReduce");
  nccl_check(lib, f(dev, dev, (size_t)count, kNcclFloat64, kNcclSum, comm, s), "ncclAllReduce");
}
void Comm::register_sendbuf(const double *sendbuf, const std::vector<int> &peer, const std::vector<int> &send_ptr) {
  if (!local) return;   // the RCCL path needs no registration
  std::lock_guard<std::mutex> lk(local->m);
  local->sendbuf[rank] = sendbuf;
  local->send_off[rank].clear();
  for (size_t k = 0; k < peer.size(); ++k) local->send_off[rank].emplace_back(peer[k], send_ptr[k]);
}
void Comm::exchange(const double *sendbuf, const std::vector<int> &peer, const std::vector<int> &send_ptr, double *recv,
                    const std::vector<int> &recv_ptr, hipStream_t s) {
  const int nn = (int)peer.size();
  ++n_exchange;
  if (local) {
    register_sendbuf(sendbuf, peer, send_ptr);
    HIP_CHECK(hipStreamSynchronize(s));   // my packed values are in place
    local->barrier(rank);                 // ... and everybody else's
    for (int k = 0; k < nn; ++k) {
      const int q = peer[k], cnt = recv_ptr[k + 1] - recv_ptr[k];
      int off = -1;
      for (auto &de : local->send_off[q]) if (de.first == rank) off = de.second;
      if (off < 0 || !local->sendbuf[q]) throw Error("halo exchange: rank " + std::to_string(q) + " has no send list for rank " + std::to_string(rank));
      if (cnt > 0)
        HIP_CHECK(hipMemcpyAsync(recv + recv_ptr[k], local->sendbuf[q] + off, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    local->barrier(rank);                 // nobody repacks before every reader is done
    return;
  }
  if (!comm) throw Error("halo exchange without a communicator");
  static fn_sendrecv fsend = nullptr, frecv = nullptr;
  static fn_group gstart = nullptr, gend = nullptr;
  if (!fsend) {
    fsend = sym<fn_sendrecv>(lib, "ncclSend"); frecv = sym<fn_sendrecv>(lib, "ncclRecv");
    gstart = sym<fn_group>(lib, "ncclGroupStart"); gend = sym<fn_group>(lib, "ncclGroupEnd");
  }
  nccl_check(lib, gstart(), "ncclGroupStart");
  for (int k = 0; k < nn; ++k) {
    const size_t ns = (size_t)(send_ptr[k + 1] - send_ptr[k]), nr = (size_t)(recv_ptr[k + 1] - recv_ptr[k]);
    if (ns) nccl_check(lib, fsend(const_cast<double *>(sendbuf) + send_ptr[k], ns, kNcclFloat64, peer[k], comm, s), "ncclSend");
    if (nr) nccl_check(lib, frecv(recv + recv_ptr[k], nr, kNcclFloat64, peer[k], comm, s), "ncclRecv");
  }
  nccl_check(lib, gend(), "ncclGroupEnd");
}
void Comm::allreduce_max(double *dev, int count, hipStream_t s) {
  ++n_allreduce;
  if (local) { local_allreduce(*this, dev, count, s, true); return; }
  if (!comm) return;
  static fn_allreduce f = nullptr;
  if (!f) f = sym<fn_allreduce>(lib, "ncclAllReduce");
  nccl_check(lib, f(dev, dev, (size_t)count, kNcclFloat64, kNcclMax, comm, s), "ncclAllReduce");
}
Comm::~Comm() {
  if (h_buf) (void)hipHostFree(h_buf);
  if (comm && lib) {
    // a communicator with a collective that never completed (a peer is gone) is aborted, not destroyed: ncclCommDestroy
    // would wait for the queue to drain
    if (void *p = dlsym(lib, poisoned ? "ncclCommAbort" : "ncclCommDestroy")) (void)reinterpret_cast<fn_destroy>(p)(comm);
  }
}


// ---- host-side helpers of the set-up passes ----
int host_threads() {
  static const int n = [] {
    if (const char *e = std::getenv("MPPGPU_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) return std::min(v, 256); }
    int cores = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    // a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota): more runnable threads than the quota get throttled in whole
    // scheduler periods, which costs more than the extra threads bring
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      long long period = 0;
      if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
        cores = std::min(cores, (int)std::max(1LL, std::atoll(q) / period));
      std::fclose(f);
    } else {
      long long quota = -1, period = 0;
      if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
      if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
      if (quota > 0 && period > 0) cores = std::min(cores, (int)std::max(1LL, quota / period));
    }
    return std::max(1, std::min(cores, 32));
  }();
  return n;
}
void host_parallel_ranges(long long n, const std::function<void(long long, long long)> &fn) {
  if (n <= 0) return;
  const int nt = (int)std::min<long long>(host_threads(), (n + 4095) / 4096);   // not worth a thread below a few thousand items
  if (nt <= 1) { fn(0, n); return; }
  std::vector<std::thread> pool;
  std::exception_ptr err;
  std::mutex m;
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&, t] {
      try { fn(n * t / nt, n * (t + 1) / nt); }
      catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); }
    });
  for (auto &th : pool) th.join();
  if (err) std::rethrow_exception(err);
}
void host_parallel_tasks(int ntasks, const std::function<void(int)> &fn) {
  if (ntasks <= 0) return;
  const int nt = std::min(host_threads(), ntasks);
  if (nt <= 1) { for (int q = 0; q < ntasks; ++q) fn(q); return; }
  std::vector<std::thread> pool;
  std::exception_ptr err;
  std::mutex m;
  std::atomic<int> next{0};
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&] {
      try { for (int q = next++; q < ntasks; q = next++) fn(q); }
      catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); }
    });
  for (auto &th : pool) th.join();
  if (err) std::rethrow_exception(err);
}
int host_prefix_sums(int *a, long long n) {
  if (n <= 0) return 0;
  const int nchunk = (int)std::min<long long>(4 * (long long)host_threads(), (n + 65535) / 65536);
  std::vector<long long> sum((size_t)nchunk + 1, 0);
  std::vector<int> mx((size_t)nchunk, 0);
  host_parallel_tasks(nchunk, [&](int q) {
    const long long b = n * q / nchunk, e = n * (q + 1) / nchunk;
    long long t = 0;
    int m = 0;
    for (long long i = b; i < e; ++i) { t += a[i + 1]; m = std::max(m, a[i + 1]); }
    sum[(size_t)q + 1] = t; mx[(size_t)q] = m;
  });
  sum[0] = a[0];
  for (int q = 0; q < nchunk; ++q) sum[(size_t)q + 1] += sum[(size_t)q];
  if (sum[(size_t)nchunk] > 0x7fffffffLL) throw Error("more than 2^31 - 1 entries: the 32-bit pointer arrays cannot hold the pattern");
  host_parallel_tasks(nchunk, [&](int q) {
    const long long b = n * q / nchunk, e = n * (q + 1) / nchunk;
    int run = (int)sum[(size_t)q];
    for (long long i = b; i < e; ++i) { run += a[i + 1]; a[i + 1] = run; }
  });
  return *std::max_element(mx.begin(), mx.end());
}
static double wall_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool poison_allocations() {
  static const bool on = [] { const char *e = std::getenv("MPPGPU_POISON"); return e && *e && std::strcmp(e, "0") != 0; }();
  return on;
}
hipStream_t &poison_stream() {
  static thread_local hipStream_t s = nullptr;
  return s;
}
PhaseTimer::PhaseTimer(const char *n) : name(n), t0(std::getenv("MPPGPU_TIMING") ? wall_seconds() : -1.0) {}
PhaseTimer::~PhaseTimer() {
  if (t0 >= 0.0) std::fprintf(stderr, "[mppgpu] %-28s %8.3f s (%d host threads)\n", name, wall_seconds() - t0, host_threads());
}

}  // namespace mpp
