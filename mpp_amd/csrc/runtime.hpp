// runtime.hpp -- host-side runtime of the VSFM solve path: device buffers, HIP-event profiling, RCCL binding,
// problem layout (permutation, CSR pattern, slot tables) and the Newton / Krylov drivers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace mpp {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      throw ::mpp::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                         std::to_string(__LINE__) + ")");                                                 \
  } while (0)

// ---- host-side helpers of the set-up passes (one-off O(N) / O(nnz) index work on the caller's mesh) ----
// threads the set-up may use: the cores this process may run on, at most 32; MPPGPU_HOST_THREADS overrides
int host_threads();
// fn(begin, end) over [0, n) in contiguous chunks, one per thread; the first exception thrown is rethrown to the caller
void host_parallel_ranges(long long n, const std::function<void(long long, long long)> &fn);
template <class F>
inline void host_parallel_for(long long n, F &&body) {   // body(i) for every i in [0, n); iterations must be independent
  host_parallel_ranges(n, [&](long long b, long long e) { for (long long i = b; i < e; ++i) body(i); });
}
// fn(q) for q in [0, ntasks): a few large tasks, one thread each (host_parallel_for would run so few items on one thread)
void host_parallel_tasks(int ntasks, const std::function<void(int)> &fn);
// a[i + 1] += a[i] for i = 0 .. n-1 (a holds n + 1 counts, a[0] stays): the running sums of a pointer array, in two parallel passes
// over fixed chunks; returns the largest a[i + 1] before the sums (the longest list)
int host_prefix_sums(int *a, long long n);
// wall-clock of a set-up phase on stderr when MPPGPU_TIMING is set
struct PhaseTimer {
  const char *name;
  double t0;
  explicit PhaseTimer(const char *n);
  ~PhaseTimer();
};

bool poison_allocations();   // environment variable MPPGPU_POISON set and not "0"
// the stream of the handle the calling thread is working for (set at every entry point of the C ABI): MPPGPU_POISON fills fresh
// allocations on THAT stream and waits for it alone, so that several handles -- each driven by its own host thread, one of them
// possibly capturing a graph -- can run poisoned side by side (a hipMemset on the legacy stream + hipDeviceSynchronize could not)
hipStream_t &poison_stream();
// std::vector whose resize() leaves trivially constructible elements untouched: the set-up arrays of a 10^8-cell mesh are sized
// first and then written by all host threads (first touch in parallel); value-initialising them on one thread was most of the set-up
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
  template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
  using std::allocator<T>::allocator;
  template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using BigVec = std::vector<T, DefaultInitAllocator<T>>;

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    if (count) HIP_CHECK(hipMalloc(&p, count * sizeof(T)));
    // MPPGPU_POISON=1 (debugging): fresh device memory is filled with 0xFF bytes -- NaN as a double, -1 as an index -- so that a
    // kernel reading something nobody wrote shows in the results instead of depending on what the allocation happened to hold
    if (count && poison_allocations()) {   // (before any copy into it: uploads follow on the same stream)
      HIP_CHECK(hipMemsetAsync(p, 0xFF, count * sizeof(T), poison_stream()));
      HIP_CHECK(hipStreamSynchronize(poison_stream()));
    }
    n = count;
  }
  // The source is pageable host memory, often a temporary of the caller that dies (or is refilled) right after the call.  On this
  // runtime hipMemcpyAsync from pageable memory returns only after the source has been consumed -- measured, 8 B to 1 GiB, source
  // overwritten or freed and reused at once, no corrupted element (tools/upload_probe.hip, profiles/r04/upload_probe.json); the
  // call takes the time of the copy and a stream synchronisation behind it ~10 us.
  void upload(const T *src, size_t count, hipStream_t s) {
    alloc(count);
    if (count) HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T> &v, hipStream_t s) { upload(v.data(), v.size(), s); }
  void upload(const BigVec<T> &v, hipStream_t s) { upload(v.data(), v.size(), s); }
  void zero(hipStream_t s) {
    if (n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

// ---- kernel classes for the HIP-event profiler ----
enum KClass : int { KC_SPMV = 0, KC_PC_APPLY, KC_PC_SETUP, KC_RESIDUAL, KC_JACOBIAN, KC_VEC, KC_REDUCE, KC_SPMV_PC, KC_COUNT };
const char *kclass_name(int k);

struct Profiler {
  bool enabled = false;
  unsigned mask = ~0u;     // kernel classes that are timed (bit k = class k); the others cost nothing
  bool open = false;       // begin() recorded an event that end() has to close
  int stride = 1;          // every stride-th launch of a class is timed (an event pair costs ~10 us of queue time: it matters
                           // where kernels last tens of microseconds); calls[] counts them all
  long long calls[KC_COUNT] = {0};
  hipStream_t stream = nullptr;
  struct Pair { hipEvent_t a, b; int cls; };
  std::vector<Pair> pool;
  size_t used = 0;
  long long launches[KC_COUNT] = {0};
  double total_ms[KC_COUNT] = {0};
  void begin(int cls);
  void end();
  void flush();   // synchronises the stream and folds recorded pairs into the totals
  void reset();
  ~Profiler();
};
struct ProfScope {
  Profiler &p;
  ProfScope(Profiler &pr, int cls) : p(pr) { p.begin(cls); }
  ~ProfScope() { p.end(); }
};

// In-process group: several handles of ONE process (one host thread each, any devices) reduce through host memory.
// Same call sites as the RCCL path; exists so that the multi-rank control flow can be exercised on a single GPU
// (RCCL accepts one rank per device) and for hosts that drive several GPUs from one process.
struct LocalGroup {
  int nranks;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::vector<double> slots;   // nranks x 64
  // halo exchange: every rank publishes its packed send buffer and where each destination's part starts
  std::vector<const double *> sendbuf;
  std::vector<std::vector<std::pair<int, int>>> send_off;   // per rank: (destination, offset)
  bool aborted = false;        // a rank failed: every waiter (now or later) leaves with an error instead of blocking
  void abort();
  explicit LocalGroup(int n) : nranks(n), slots((size_t)n * 64, 0.0), sendbuf(n, nullptr), send_off(n) {}
  void barrier(int rank) { double d = 0.0; allreduce(rank, &d, 0, false); }
  // every rank contributes `count` (<= 64) values; returns, in rank order (deterministic), the sum or max in `v`
  void allreduce(int rank, double *v, int count, bool is_max);
};

// ---- RCCL, bound at run time (dlopen) so that single-GPU use never needs the library ----
struct Comm {
  void *lib = nullptr;
  void *comm = nullptr;
  std::shared_ptr<LocalGroup> local;
  double *h_buf = nullptr;     // pinned staging of the in-process path
  int nranks = 1, rank = 0;
  long long n_allreduce = 0, n_exchange = 0;   // collectives issued since the last mppgpu_reset_stats
  int version = 0;             // RCCL's ncclGetVersion (checked against the header the bindings were verified with: runtime.cpp)
  bool poisoned = false;       // a wait for this communicator's work timed out: abort it at destruction instead of draining it
  static int runtime_version();   // loads RCCL, runs the version guard, returns ncclGetVersion
  bool active() const { return comm != nullptr || local != nullptr; }
  void init_local(std::shared_ptr<LocalGroup> g, int rank);
  static void get_unique_id(void *id128);
  void init(int nranks, int rank, const void *id128);
  void allreduce_sum(double *dev, int count, hipStream_t s);
  // halo exchange: `sendbuf` holds, for neighbour k, the values for peer[k] at [send_ptr[k], send_ptr[k+1]); the values
  // received from peer[k] land at recv[recv_ptr[k] .. recv_ptr[k+1]).  RCCL: grouped ncclSend / ncclRecv on the stream.
  void register_sendbuf(const double *sendbuf, const std::vector<int> &peer, const std::vector<int> &send_ptr);
  void exchange(const double *sendbuf, const std::vector<int> &peer, const std::vector<int> &send_ptr, double *recv,
                const std::vector<int> &recv_ptr, hipStream_t s);
  void allreduce_max(double *dev, int count, hipStream_t s);
  // a failure on this rank: wake the in-process peers with an error (RCCL peers cannot be reached from here: the
  // launcher ends the other rank processes when one exits with an error, see bench.py)
  void abort_peers();
  ~Comm();
};

struct Condition {
  int kind = 0, cond_type = 0, swap_order = 0, n = 0;
  int eqn = 0;                         // coupled TH: 0 mass equation, 1 heat equation
  std::vector<int> cell;               // external 0-based
  std::vector<double> dist_up, dist_dn, area, uvec, value;
  std::vector<double> sink_p, sink_n;  // down-regulated sinks: pot_mass_sink_pressure / _exponent
  std::vector<double> other;           // coupled TH: temperature (mass eqn) / pressure (heat eqn) of the boundary auxvar
  int offset = 0;                      // first entry in the concatenated BC (or SS) numbering
};

struct Options {
  int ksp_type = 1;        // 0 preonly, 1 gmres, 2 bcgs       (PETSc default: gmres)
  int pc_type = 2;         // 0 none, 1 ilu (level / box solves), 2 bjacobi (per soil column), 3 jacobi
  bool pc_is_ilu = true;   // "ilu" -- PETSc's default PC, and this library's since round 5 -- served by the column solver on vertical-only
                           // grids (ILU(0) of a block-tridiagonal matrix in natural ordering IS the per-column LU), by ILU(0) elsewhere
  double ksp_rtol = 1e-5, ksp_atol = 1e-50, ksp_dtol = 1e4;
  int ksp_max_it = 10000, gmres_restart = 30;
  double snes_rtol = 1e-8, snes_atol = 1e-50, snes_stol = 1e-10, snes_divtol = 1e4;  // stol: MultiPhysicsProbBaseType.F90:1112
  int snes_max_it = 50, snes_max_funcs = 10000;
  int linesearch_type = 2;             // -snes_linesearch_type: 1 basic, 2 bt (PETSc's default for newtonls), 3 l2, 4 cp
  bool use_dynamic_linesearch = false; // solver%use_dynamic_linesearch (src/mpp/dtypes/SolverType.F90:43): other line searches
                                       // are tried before a time-step cut (SystemOfEquationsBaseType.F90:432-506)
  int spmv_variant = -1;   // -1 auto: pipelined wave CSR-stream for large matrices, block CSR-stream for small ones
  int assembly = -1;       // -1 auto, 0 general kernels, 1 fused column kernels
  int nseg = 0;            // column segments per thread block row (0 = auto)
  bool fused_assembly = true;  // F and J of a line-search trial point in one kernel ...
  bool track_mass_exchange = false;  // accumulate VAR_LATERAL_MASS_EXCHANGED / VAR_BC_MASS_EXCHANGED after every sub-step
  bool ilu_graph = true;       // replay the level launches of the ILU(0) triangular solves from hipGraphs ...
  int ilu_graph_nodes = 128;   // ... of at most this many kernel nodes each (Solver::run_graph)
  bool fused_ls_dots = true;   // line search: ||Y||, F.(J Y), ||X|| reduced inside the SpMV (compact-index kernels)
  bool fused_rows = true;      // column march with the boundary fluxes folded in: complete rows -> ||F||, ||X|| and the
                               // column LU come out of the assembly pass (see Solver::rows_complete)
  bool defer_test0 = true;     // bcgs: iteration 0's convergence test is made when the first iteration's scalars come back
  bool ell_assembly = true;    // general assembly on the ELL copy of the incidence lists (k_flux_rows_ell)
  int ilu_chain = 0;           // ILU(0) solves: 0 one launch per level (graph replay; the default), 1 / 2 one launch per triangle with
                               // tiles chained by flags (1: where it pays, 2: always).  Opt-in: without ilu_chain_tickets the chain's
                               // progress relies on workgroups being dispatched in index order, which HIP does not promise
  bool setup_device = true;    // set-up passes that have a device route (setup_kernels.hip) take it
  bool setup_verify = false;   // ... and the host route as well, compared entry by entry (tests)
  bool spmv_cols_dots = true;  // scalar system on soil columns: the line search's product + sums from the column tridiagonals at their computed places
  bool lazy_tri_l = true;      // column LU of the scalar system: the march does not store l where the fused SpMV + solve kernels will run
  bool th_spmv_tri = true;     // coupled TH on soil columns, block-Jacobi: SpMV + block-Thomas solve + dots as one kernel (k_th_spmv_tri)
  bool th_lazy_factor = true;  // coupled TH, BiCGStab + block-Jacobi: the block factorisation rides on the first solve's sweep (k_th_tri_factor<true, .>)
  bool th_inv_fly = true;      // ... and the fused products (k_th_spmv_tri<.., true>) redo the pivots' inverses from the rows they hold: plane inv neither written nor read
  bool th_stage = true;        // TH factorisation / index-free SpMV: matrix rows staged through LDS (off: the lane-by-lane reads, for the A/B)
  bool th_spmv_cols = true;    // coupled TH on soil columns: y = A x from the block-tridiagonal values at their computed places (no index)
  bool th_fused_dots = true;   // coupled TH, bjacobi: the Krylov dot products of M^{-1} r come out of the block column solve
  int ilu_box = 1;             // ILU(0) solves box by box when the pattern is a structured grid in natural ordering (BoxGrid); 0: never, 2: or fail
  bool ilu_box_factor = true;  // ... and the numeric factorisation too (extents other than 2); false: by the elimination program of the general path
  bool ilu_box_img_out = true;   // box solves: the way back to the row order through an LDS image of the box (k_box_gather_out: 0.53 ms at 512^3)
                                 // instead of the per-row slot map (k_gather: 0.79 ms)
  bool ilu_box_stream = true;  // ... as streams: a wave owns the NB1 boxes of one (J, K), pencils by ticket in the order of J + K, faces handed over 8 steps
                               // at a time through sentinel-initialised face buffers (k_box_lower_stream); needs 8 x 8 box cross-sections
  bool ilu_box_fused_io = false; // streamed box solves: the lower solve gathers its right-hand side from the Krylov vector, the upper one scatters the
                                 // solution into it (no k_box_gather_in / k_gather passes around the solves).  Measured no faster at 512^3 (5.09 against
                                 // 4.99 ms per application: a lane-per-line gather costs the solve what the two coalescing passes cost): off
  bool ilu_box_queue = false;  // ... with ONE launch per triangle: boxes handed out by ticket in level order to ilu_box_workers resident waves, a box
                               // waits for its three predecessors' flags (k_box_lower_queue; liveness independent of the dispatch order)
  int ilu_box_workers = 1024;
  int ilu_box_len = 0;         // box extent along the fastest direction (0: 64 from 2^25 rows, else 32)
  bool ilu_chain_tickets = false;  // chained ILU(0): tiles handed out by an atomic ticket instead of in dispatch order (kernels.hpp):
                                   // liveness then does not depend on the dispatch order, at +50-75 % per application
  unsigned ilu_chain_spins = 1u << 16;   // polls (~0.4 us apart: ~26 ms) before a chained wait gives up and the apply is redone per level
  bool ls_ahead = true;        // the line search's first reductions ride behind the Krylov solver's last update
  bool fused_spmv_pc = true;   // vertical-only grids, column preconditioner: M^{-1} A v as one kernel (k_tri_solve_lds<..., SPMV>) instead of
                               // the CSR SpMV followed by the column solve -- A v is never written or read back
  bool fused_first = true;     // ... and the start of a BiCGStab solve -- rp = U^{-1}(L^{-1} F) and v = M^{-1} A rp -- as one kernel too
  bool thermal_reference_defects = false;   // reproduce GoveqnThermalKSPTemperatureSoilType.F90:906-908 (stale area) and :1215 (dhsdT**area)
  bool fold_finalize = false;  // reductions of <= 4096 partial sums finished by the fused SpMV + column-solve kernels themselves (measured: see DESIGN 12)
  bool bc_inline = true;       // complete-row march: boundary fluxes of a column's end cells evaluated in the march's prologue
  bool fused_waxpy = true;     // ... and s = r - alpha v inside the kernel of the second product
  bool lazy_r = true;          // BiCGStab: r = s - omega t is not written by an iteration that is expected to be the last (only its norm)
  bool ls_spec = true;         // ... and the search's first trial point (the full step) is evaluated there too (LsAhead::spec)
  bool cols3d = true;          // 3-D meshes of complete soil columns: column march (accumulation + chain links) + lateral pass instead of the
                               // general auxvar pass + row gather over every connection (Solver::use_cols3d)
  bool cols3d_rows = true;     // ... with the march's tridiagonal handed over in three coalesced arrays and the lateral pass writing whole CSR rows
  bool ls_fold = true;         // the search's F . (J Y) and ||Y|| out of the kernels of a one-iteration BiCGStab solve (F . (J p), F . (J s) reduced by the
                               // two fused product kernels, Y . Y by the update): no product pass of the search's own (Solver::ksp_bcgs, fold_ls)
  bool wide_xr = true;         // BiCGStab update kernel on a one-pair-per-lane grid (large vectors)
  double collective_timeout = 900.0;   // seconds a rank waits for device scalars that depend on the other ranks (an all-reduce) before it
                               // gives up with an error -- a peer that failed or never arrived must not leave this rank waiting forever
  bool host_poll = true;       // scalar read-backs: the reducing kernel publishes into pinned host memory and the host polls a
                               // sequence word (Solver::fetch_scalars); false: hipMemcpyAsync + hipStreamSynchronize
  bool soil_classes = true;    // column march: one byte per cell + a class table when the mesh has <= 256 distinct soil tuples
  bool level_geometry = true;  // column march: per-level upwind weight / gravity distance when they are the same in every column
  int tri_lds = -1;            // column solve with L^{-1} r parked in LDS between the sweeps: 0 off, 1 on, -1 auto (Solver::tri_lds)
  int tri_group = 24;          // ... and levels kept in flight per stream and wave (8, 16, 24)
  int spmv_tpw = 0;            // > 0: 64-row tiles per wave of the compact-index SpMV (default 2 / 4 by variant)
  bool ls_trace = false;       // one line per bt line search on stderr: Newton iteration, trial points, accepted lambda (development aid)
  bool fused_predict = true;   // skip the fused Jacobian at the first trial when the previous line search backtracked
  int fused_trials = 2;        // ... for the first `fused_trials` trial points of a line search (later ones: F only)
};

struct Solver;  // all state behind a handle

}  // namespace mpp
