// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
// All of these are HBM-bound fp64/int32 streaming kernels (arithmetic intensity 0.1-0.5 flop/B, far below the
// fp64 ridge), so the design rules are: coalesced 8-byte-per-lane streams in struct-of-arrays layout, grids of
// >= 8 blocks per CU, no atomics, deterministic two-stage reductions with wavefront shuffles.  MFMA is not used:
// no dense block contraction appears anywhere on this path.
//
// Internal cell ordering is "tiled level-major": soil columns are grouped in tiles of 64 (one wavefront); inside
// tile t, cell (level k, column c) sits at t*64*nz + k*tw + (c - 64 t), tw = tile width (64, last tile may be
// narrower).  The 64 lanes of a wave work on 64 neighbouring columns at the same depth, so every load/store of a
// column march is one contiguous 512-byte transaction, consecutive levels are consecutive in memory (a wave
// streams a contiguous nz*512-byte region per array: DRAM-page and TLB friendly, measured +25 % over a global
// level-major layout), and the vertical neighbours of a row are exactly +-tw rows away (x re-use stays in L1/L2).
// Streaming kernels use one element per lane and huge grids instead of grid-stride loops: on MI355X a 1.7 GB copy
// runs at 6.3 TB/s that way versus 4.8 TB/s with a 4096-block grid-stride loop (tools/membench.hip).
#include "kernels.hpp"

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <type_traits>

namespace mpp {

namespace {

constexpr int kBlock = 256;

// once-streamed data is loaded / stored non-temporal: on MI355X a CSR-like stream mix runs at 6.1 TB/s that way
// versus 5.4 TB/s with default cache policy (tools/membench.hip csr_mix vs csr_mix_nt)
template <class T>
__device__ __forceinline__ T ldnt(const T *p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void stnt(T v, T *p) { __builtin_nontemporal_store(v, p); }
typedef double dpair __attribute__((ext_vector_type(2)));   // two doubles, one 16-byte access

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block-wide sum in a fixed order (lanes -> wave leaders -> thread 0); valid result in thread 0
__device__ __forceinline__ double block_sum(double v, double *lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kBlock / 64; ++k) t += lds[k];
  }
  return t;
}

// ---- second stage of a reduction inside the producing kernel (Fold, kernels.hpp) ----
__device__ void bcgs_scalar_step(int step, double *sc);
struct Publish { double *host; unsigned long long *seq; unsigned long long value; };
__device__ __forceinline__ void publish_scalars(const Publish &p, const double *sc);
// a block's partial sum: an agent-scope store when another block is going to read it in this very launch
__device__ __forceinline__ void store_partial(double *p, double v, bool folded) {
  if (folded) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
// Called by every thread of every block, after `writer` (one thread per block) has stored the block's partial sums
// partials[j * nb + block], j < f.nvec.  The sum of a vector's nb partial sums is formed exactly as k_finalize's single-pass block
// forms it (256 strided accumulators, wave sums, the four wave sums in order); BLOCK = 64: one wave stands in for the four.
template <int BLOCK>
__device__ __forceinline__ void fold_tail(const Fold &f, const double *partials, int nb, bool writer) {
  if (!f.counter) return;
  __shared__ int fold_last;
  __shared__ double fold_lds[kBlock / 64];
  if (writer) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the partial sums are acknowledged
    const unsigned ticket = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fold_last = (ticket == (unsigned)nb - 1u);
  }
  __syncthreads();
  if (!fold_last) return;
  for (int j = 0; j < f.nvec; ++j) {
    const double *pj = partials + (size_t)j * nb;
    double t;
    if (BLOCK == kBlock) {
      double acc = 0.0;
      for (int b = threadIdx.x; b < nb; b += kBlock) acc += __hip_atomic_load(pj + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = block_sum(acc, fold_lds);
    } else {
      t = 0.0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) {
        double acc = 0.0;
        for (int b = w * 64 + (int)threadIdx.x; b < nb; b += kBlock) acc += __hip_atomic_load(pj + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t += wave_sum(acc);               // (lane 0 holds the sum)
      }
    }
    if (threadIdx.x == 0) __hip_atomic_store(f.sums + S_SUM0 + j, 0.0 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.step >= 0) bcgs_scalar_step(f.step, f.sc);
  }
  publish_scalars(Publish{f.pub_host, f.pub_seq, f.pub_value}, f.sc);
}

template <int SAT_T = -1, bool NT = false>
__device__ __forceinline__ Soil load_soil(const CellArrays &ca, size_t i) {
  Soil s;
  s.sat_res = NT ? ldnt(ca.sat_res + i) : ca.sat_res[i];
  s.alpha = NT ? ldnt(ca.alpha + i) : ca.alpha[i];
  s.m = NT ? ldnt(ca.m + i) : ca.m[i];
  s.por = NT ? ldnt(ca.por + i) : ca.por[i];
  s.sf = SAT_T >= 0 ? SAT_T : (ca.satf ? (int)ca.satf[i] : ca.sat_func);   // per-cell dispatch (SaturationFunction.F90:564-650)
  if (s.sf == kSatSBC) {
    s.pu = ca.sbc_pu[i]; s.ps = ca.sbc_ps[i]; s.b2 = ca.sbc_b2[i]; s.b3 = ca.sbc_b3[i];
  } else {
    s.pu = s.ps = s.b2 = s.b3 = 0.0;
  }
  if (ca.por_slope) { s.por_pref = ca.por_pref[i]; s.por_slope = ca.por_slope[i]; } else { s.por_pref = s.por_slope = 0.0; }
  return s;
}

// cell states: 0 inactive (mesh%is_active false), 1 active, 2 ghost -- a cell owned by another rank: its auxvars are
// evaluated and its neighbours exchange flux with it, but its own row is an identity row (see Solver::halo_exchange)
__device__ __forceinline__ bool is_active(const CellArrays &ca, size_t i) {
  return ca.active ? ca.active[i] != 0 : true;
}
__device__ __forceinline__ double frac_of(const CellArrays &ca, size_t i) { return ca.frac_cell ? ca.frac_cell[i] : ca.frac_liq; }
__device__ __forceinline__ bool owns_row(const CellArrays &ca, size_t i) {
  return ca.active ? ca.active[i] == 1 : true;
}

// ------------------------------------------------------------------------------------------------
// accumulation term only (RichardsODEPressureAccum, GoveqnRichardsODEPressureType.F90:1603-1634) -- PreSolve
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accum(CellArrays ca, int n, double dtInv, double *out) {
  MPP_STAGE_POW(ca.powtab);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  double acc = 0.0;
  if (owns_row(ca, i)) {
    Soil s = load_soil(ca, i);
    Aux a = auxvar_compute<false>(s.sf, ca.eos, s, ca.P[i], frac_of(ca, i), pt);
    double por, dpor;
    porosity(s, ca.por_slope != nullptr, ca.P[i], por, dpor);
    acc = por * a.den * a.sat * ca.vol[i] * dtInv;
  }
  out[i] = acc;
}

// Column LU (Thomas) as every kernel stores and applies it (round 5: the back substitution in product form):
//   l_k = lo_k inv_{k-1},  inv_k = 1 / fma(-l_k, up_{k-1}, di_k),  g_k = up_k inv_k   (plane c holds g, not the upper entries)
//   forward  d_k = r_k - l_k d_{k-1};   backward  x_k = fma(-g_k, x_{k+1}, d_k inv_k)
// A kernel that has inv_k in a register when it parks d_k (the fused products: they redo the pivots) parks d_k inv_k, and its back
// sweep reads g alone -- 8 instead of 16 bytes a cell; the assembly march leaves its forward-eliminated residual scaled the same way.
// The two products are rounded by themselves wherever they are formed (tri_scale), so that every kernel gives the same bits.
__device__ __forceinline__ double tri_scale(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double tri_back(double t, double g, double next) { return fma(-g, next, t); }   // t = d inv

// ------------------------------------------------------------------------------------------------
// fused column march: auxvars + accumulation + vertical fluxes (+ derivatives) in one pass.
// Restates RichardsODEComputeResidual / ComputeJacobian (GoveqnRichardsODEPressureType.F90:388-453) for meshes
// whose internal connections are vertical only.  One thread owns levels [k0,k1) of one column and keeps the
// previous level's auxvars in registers, so every auxvar and every flux is evaluated once (plus one halo level
// per segment end when nseg > 1).  Sum order per row equals the reference's serial connection loop.
// ------------------------------------------------------------------------------------------------
// FUSE (nseg == 1, boundary conditions only on the end cells of a column, their fluxes evaluated beforehand by k_bc
// into side arrays): every row is complete when the march leaves it, so the kernel also leaves the partial sums of
// F.F and P.P (the line search's ||F||, ||X||) and, when a.tl is set, the column's LU factors (k_tri_factor's arithmetic).
// Software pipeline (round 2; profiles/r02_pmc_sq_march.md): a wave spends most of its life waiting on memory at 3-4 waves
// per SIMD, so the march keeps the operands of TWO levels in flight, in two named register sets that alternate -- no copies,
// so using one set waits for that set only (vector-memory operations retire in order) while the other stays in flight.
// Everything a level needs from memory is part of its fetch (pressure or the two vectors of a trial point, soil, geometry,
// cell state, source, the residual to eliminate); fetches are unconditional loads with the level index clamped, so every
// path issues the same number of operations and the compiler can count them instead of draining the queue.  The first CSR
// slot of a row is computed, not loaded: rows of a column tile are level-major with 2 entries in the end rows and 3 between.
#ifndef MPP_MARCH_DEPTH_RESIDUAL
#define MPP_MARCH_DEPTH_RESIDUAL 2
#endif
constexpr int kMarchDepthResidual = MPP_MARCH_DEPTH_RESIDUAL;
// the residual-only march of the benchmark's soil (no real power at all) needs 102 registers, six more than five waves per SIMD allow.
// Asked for five (amdgpu_waves_per_eu: 96 registers + 32 bytes of scratch per lane) it ran 3-4 % SLOWER (1.044 / 1.070 -> 1.084 / 1.098 ms,
// profiles/r05/occupancy_hints_ab.json): occupancy is not what holds it; the hint stays off (1 = none)
#ifndef MPP_RESIDUAL_MARCH_WAVES
#define MPP_RESIDUAL_MARCH_WAVES 1
#endif
// C3D (never with FUSE): the march of a 3-D mesh of complete columns -- tridiagonal at the CSR slots of the 7-point rows, auxvars left for
// the lateral pass (ColsArgs::slot_di, aux); a template parameter so that the marches of vertical-only grids keep their registers
template <int MODE, int SAT_T, int DEN_T, bool FUSE, bool C3D = false>  // MODE 0 residual, 1 Jacobian, 2 both; SAT_T / DEN_T: see physics.hpp
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((MODE == 0 && SAT_T == kSatVGHalf && DEN_T == 0) ? MPP_RESIDUAL_MARCH_WAVES : 1)))
void k_assemble_cols(ColsArgs a) {
  __shared__ double red[kBlock / 64];
  extern __shared__ __attribute__((aligned(16))) double cls_lds[];   // [7][ncls] when soil classes are in use
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  constexpr bool GEN = SAT_T < 0;   // general soil: per-cell saturation functions and / or the linear porosity model
  constexpr bool NEEDS_POW = !(SAT_T == kSatVGHalf && DEN_T == 0);   // (van Genuchten m = 1/2 with TGDPB01: no real power at all)
  __shared__ double pow_lds_[NEEDS_POW ? kPowTableDoubles : 1];
  PowTab pt{nullptr, nullptr};
  if (NEEDS_POW) pt = stage_pow_tables(a.cell.powtab, pow_lds_);
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = c < a.ncols;
  const bool classes = !GEN && a.soil_cls != nullptr;
  if (classes) {
    for (int q = threadIdx.x; q < 7 * a.ncls; q += kBlock) cls_lds[q] = a.cls_table[q];
    __syncthreads();
  }
  if (!FUSE && !valid) return;
  double accF = 0.0, accP = 0.0;
  if (valid) {
    const int nz = a.nz;
    const int tile = c >> 6, lane = c & 63;
    const int tw = min(64, a.ncols - (tile << 6));          // width of this column tile
    const size_t cbase = (size_t)tile * 64 * nz + lane;      // cell (k,c) = cbase + k*tw
    const int seg = FUSE ? 0 : blockIdx.y;
    const int nseg = FUSE ? 1 : a.nseg;
    const int k0 = (int)((long long)nz * seg / nseg);
    const int k1 = (int)((long long)nz * (seg + 1) / nseg);
    if (!FUSE && k0 >= k1) return;
    const int kstart = k0 > 0 ? k0 - 1 : 0;
    const int kend = k1 < nz ? k1 + 1 : nz;
    const CellArrays &ca = a.cell;
    const bool factor = FUSE && JAC && a.tl != nullptr;
    const bool trial = FUSE && a.trial_x != nullptr;
    const bool lin_por = GEN && ca.por_slope != nullptr;
    const bool need_Fin = factor && !RES && a.fwd != nullptr;   // mode 1: the residual to forward-eliminate is in a.F
    const int jb = FUSE ? a.col_bc[c] : -1, jt = FUSE ? a.col_bc[a.ncols + c] : -1;  // side entries of rows 0 / nz-1
    // first CSR slot of row (k, c) -- the pattern of a vertical-only grid in the tiled level-major order (checked on the host)
    const int slot_tile = tile * 64 * (3 * nz - 2);
    auto row_slot = [&](int k) -> int {
      if (nz == 1) return slot_tile + lane;          // one-cell columns: a diagonal matrix
      if (k == 0) return slot_tile + 2 * lane;
      if (k == nz - 1) return slot_tile + 2 * tw + 3 * (nz - 2) * tw + 2 * lane;
      return slot_tile + 2 * tw + 3 * ((k - 1) * tw + lane);
    };
    double inv_pp = 0.0, c_pp = 0.0;   // 1/pivot and super-diagonal of the row before the one being completed
    double d_pp = 0.0;                 // ... and its forward-eliminated residual
    // Boundary fluxes of this column's two end cells, evaluated here instead of by a separate k_bc launch (a.bc_ptr != nullptr): the
    // same loop over the cell's boundary entries, the same functions (GE:1815-1877, 2076-2139).  Kept in four registers until the
    // rows are finished.
    double sF0 = 0.0, sF1 = 0.0, sD0 = 0.0, sD1 = 0.0;   // bottom / top
    const bool bc_inline = FUSE && a.bc_ptr != nullptr;
    if (bc_inline) {
#pragma unroll
      for (int end = 0; end < 2; ++end) {   // (unrolled: the four sums stay in registers)
        const int j = end ? jt : jb;
        if (j < 0) continue;
        const size_t cell = cbase + (size_t)(end ? nz - 1 : 0) * tw;
        if (!owns_row(ca, cell)) continue;   // inactive (or ghost) cell: no boundary flux
        Soil s = load_soil(ca, cell);
        const double Pc = trial ? fma(a.trial_a, a.trial_y[cell], a.trial_x[cell]) : ca.P[cell];
        const Aux in = auxvar_compute<JAC, SAT_T, DEN_T>(s.sf, ca.eos, s, Pc, frac_of(ca, cell), pt);
        double f = 0.0, d = 0.0;
        for (int e = a.bc_ptr[j]; e < a.bc_ptr[j + 1]; ++e) {
          const int t = a.bc_type[e];
          const int ctype = t & 0xffff, swap = t >> 16;
          const double pb = (ctype == kCondMassFlux) ? in.P : a.bc_value[e];
          const Aux bc = auxvar_compute<JAC, SAT_T, DEN_T>(s.sf, ca.eos, s, pb, ca.frac_liq, pt);
          ConnGeom g;
          g.Dq = a.bc_Dq[e]; g.upweight = 0.0; g.dist_gravity = a.bc_dgrav[e]; g.area = a.bc_area[e];
          double Jup = 0.0, Jdn = 0.0, flux, jdn_outer;
          if (!swap) {
            flux = richards_flux<JAC>(bc, in, g, ctype, Jup, Jdn);
            jdn_outer = Jdn;
          } else {  // RichardsFlux swap_order (RichardsMod.F90:96-112)
            flux = -richards_flux<JAC>(in, bc, g, ctype, Jup, Jdn);
            jdn_outer = -Jup;
          }
          if (RES) f += flux;
          if (JAC) d += -jdn_outer;
        }
        if (end) { sF1 = f; sD1 = d; } else { sF0 = f; sD0 = d; }
      }
    }

    Aux prev = {};
    bool prev_act = false;
    double f_prev = 0.0, dconn_prev = 0.0, dacc_prev = 0.0, lo_prev = 0.0, up_prev = 0.0, src_prev = 0.0, Fin_prev = 0.0;

    // row ip = (kp, c) is complete: boundary contribution, store, reductions, elimination step
    auto finish_row = [&](int kp, size_t ip, bool last) {
      const int js = (kp == 0) ? jb : (last ? jt : -1);
      double f_row = 0.0;
      if (RES) {
        double f = f_prev;
        if (a.src) f -= src_prev;
        if (FUSE && js >= 0) f += bc_inline ? (kp == 0 ? sF0 : sF1) : a.side_f[js];
        stnt(f, a.F + ip);
        if (FUSE) accF += f * f;
        f_row = f;
      }
      if (JAC) {
        double di = dconn_prev + dacc_prev;
        if (FUSE && js >= 0) di += bc_inline ? (kp == 0 ? sD0 : sD1) : a.side_d[js];
        if (C3D) {   // 7-point rows: the column's tridiagonal at its CSR slots (the lateral pass adds to the diagonal afterwards;
                                    // never with the complete-row march: compiled out of it -- its registers are spoken for)
          if (a.tri3) {   // coalesced, for the lateral pass to place (it writes the whole row)
            const size_t nn = (size_t)a.ncols * nz;
            stnt(kp > 0 ? lo_prev : 0.0, a.tri3 + ip); stnt(di, a.tri3 + nn + ip); stnt(last ? 0.0 : up_prev, a.tri3 + 2 * nn + ip);
          } else {
            if (kp > 0) a.val[a.slot_lo[ip]] = lo_prev;
            a.val[a.slot_di[ip]] = di;
            if (!last) a.val[a.slot_up[ip]] = up_prev;
          }
        } else {
          int slot = row_slot(kp);
          if (kp > 0) a.val[slot++] = lo_prev;
          a.val[slot++] = di;
          if (!last) a.val[slot] = up_prev;
        }
        if (factor) {
          const double l = lo_prev * inv_pp;
          const double piv = fma(-l, c_pp, di);   // (written as the fma it is: k_tri_solve_lds recomputes l and must round alike)
          const double inv = 1.0 / piv;
          if (a.store_l) { stnt(l, a.tl + ip); stnt(inv, a.tinvb + ip); }   // (read by the plain solves only)
          stnt(tri_scale(up_prev, inv), a.tc + ip);
          inv_pp = inv; c_pp = up_prev;
          if (a.fwd) {   // forward elimination of the residual as well (left scaled: d inv): the Krylov solver's first M^{-1} F starts from here
            const double d = (RES ? f_row : Fin_prev) - l * d_pp;
            stnt(tri_scale(d, inv), a.fwd + ip);
            d_pp = d;
          }
        }
      }
    };

    // the streamed operands of one level and of the connection below it
    struct Level { double P, y, sat_res, alpha, m, por, vol, accp, Dq, upw, dgrav, pref, slope, frac, src, Fin, pu, ps, b2, b3;
                   int sf, cls; signed char act; };
    const bool lev_geo = a.lev_upw != nullptr;
    auto fetch = [&](int kreq) {
      Level L;
      const int k = min(kreq, kend - 1);   // past the end: a harmless repeat of the last level (never used)
      const size_t i = cbase + (size_t)k * tw;
      if (trial) { L.P = ldnt(a.trial_x + i); L.y = ldnt(a.trial_y + i); }   // W = X + a Y is formed when the level is used
      else { L.P = ldnt(ca.P + i); L.y = 0.0; }
      L.cls = 0;
      if (classes) { L.cls = ldnt(a.soil_cls + i); L.sat_res = L.alpha = L.m = 0.0; }   // the parameters come from LDS at use
      else {
        L.sat_res = ldnt(ca.sat_res + i); L.alpha = ldnt(ca.alpha + i);
        L.m = (SAT_T == kSatVGHalf) ? 0.5 : ldnt(ca.m + i);
      }
      if (GEN) {
        L.por = ldnt(ca.por + i); L.vol = ldnt(ca.vol + i);
        L.sf = ca.satf ? (int)ca.satf[i] : ca.sat_func;
        if (lin_por) { L.pref = ldnt(ca.por_pref + i); L.slope = ldnt(ca.por_slope + i); } else { L.pref = L.slope = 0.0; }
      } else {
        L.por = a.lev_porvol ? a.lev_porvol[k] : ldnt(a.porvol + i);   // (constant) porosity only ever multiplies the volume here
        L.vol = 1.0;
        L.pref = L.slope = 0.0; L.sf = SAT_T;
      }
      if (SAT_T == kSatSBC && !classes && a.sbc_b) {   // one smoothing kind in the mesh: p_u and the one non-zero coefficient
        L.pu = ldnt(ca.sbc_pu + i); L.b2 = ldnt(a.sbc_b + i); L.ps = L.b3 = 0.0;   // (p_s and the other one: formed at use)
      } else if (L.sf == kSatSBC && !classes) { L.pu = ca.sbc_pu[i]; L.ps = ca.sbc_ps[i]; L.b2 = ca.sbc_b2[i]; L.b3 = ca.sbc_b3[i]; }
      else L.pu = L.ps = L.b2 = L.b3 = 0.0;
      L.accp = RES ? ldnt(a.accum_prev + i) : 0.0;
      L.act = ca.active ? ca.active[i] : (signed char)1;
      L.frac = ca.frac_cell ? ca.frac_cell[i] : ca.frac_liq;
      L.src = (RES && a.src) ? ldnt(a.src + i) : 0.0;
      L.Fin = need_Fin ? ldnt(a.F + i) : 0.0;
      if (k > kstart) {
        const size_t ic = i - tw;
        L.Dq = ldnt(a.DqA + ic);                 // conductance x face area: the flux is linear in it
        if (lev_geo) { L.upw = a.lev_upw[k]; L.dgrav = a.lev_dgrav[k]; }   // wave-uniform: scalar loads
        else { L.upw = ldnt(a.conn.upw + ic); L.dgrav = ldnt(a.conn.dgrav + ic); }
      } else { L.Dq = L.upw = L.dgrav = 0.0; }
      return L;
    };

    auto process = [&](int k, const Level &L) {
      const size_t i = cbase + (size_t)k * tw;
      const bool act = L.act != 0;
      const bool owned = (k >= k0 && k < k1);
      Aux cur = {};
      double f_cur = 0.0, dacc_cur = 1.0;  // inactive cell: diagonal 1 (GE:1682-1684)
      double Pi = L.P;
      if (trial) {   // the line search's trial point W = X + a Y, formed here and stored for the next pass
        Pi = fma(a.trial_a, L.y, L.P);
        stnt(Pi, a.trial_w + i);
      }
      cur.P = Pi;
      if (FUSE && RES) accP += Pi * Pi;
      if (act) {
        Soil s;
        s.sat_res = L.sat_res; s.alpha = L.alpha; s.m = L.m; s.por = L.por;
        s.por_pref = L.pref; s.por_slope = L.slope; s.sf = L.sf;
        s.pu = L.pu; s.ps = L.ps; s.b2 = L.b2; s.b3 = L.b3;
        if (SAT_T == kSatSBC && !classes && a.sbc_b) {
          s.ps = -0.9 / s.alpha;                       // SatFunc_Set_SBC_*: ps = -0.9/alpha, the same IEEE quotient as on the host
          if (a.sbc_b_is_b3) { s.b3 = L.b2; s.b2 = 0.0; }
        }
        if (classes) {
          const double *t = cls_lds + L.cls;
          const int nc = a.ncls;
          s.sat_res = t[0]; s.alpha = t[nc];
          s.m = (SAT_T == kSatVGHalf) ? 0.5 : t[2 * nc];
          if (SAT_T == kSatSBC) { s.pu = t[3 * nc]; s.ps = t[4 * nc]; s.b2 = t[5 * nc]; s.b3 = t[6 * nc]; }
        }
        cur = auxvar_compute<JAC, SAT_T, DEN_T>(L.sf, ca.eos, s, Pi, L.frac, pt);
        if (C3D && owned) {   // for the lateral pass of a 3-D mesh (k_flux_rows_ell<.., LAT>)
          a.aux.kr[i] = cur.kr; a.aux.den[i] = cur.den;
          if (JAC) { a.aux.dkr[i] = cur.dkr; a.aux.dden[i] = cur.dden; }
        }
        if (owned) {
          const double vol = L.vol;
          if (lin_por) {
            double por, dpor;
            porosity(s, true, Pi, por, dpor);
            if (RES) f_cur = por * cur.den * cur.sat * vol * a.dtInv;
            if (JAC) dacc_cur = (dpor * cur.den * cur.sat + por * cur.dden * cur.sat + por * cur.den * cur.dsat) * vol * a.dtInv;
          } else {
            if (RES) f_cur = s.por * cur.den * cur.sat * vol * a.dtInv;
            if (JAC) dacc_cur = (s.por * cur.dden * cur.sat + s.por * cur.den * cur.dsat) * vol * a.dtInv;
          }
        }
      }
      if (C3D && owned && !act) {   // (k_auxvars leaves zeros for an inactive cell)
        a.aux.kr[i] = 0.0; a.aux.den[i] = 0.0;
        if (JAC) { a.aux.dkr[i] = 0.0; a.aux.dden[i] = 0.0; }
      }
      if (RES && owned) f_cur = f_cur - L.accp;
      double dconn_cur = 0.0, lo_cur = 0.0;
      if (k > kstart) {
        if (prev_act && act) {
          ConnGeom g;
          g.Dq = L.Dq; g.upweight = L.upw; g.dist_gravity = L.dgrav; g.area = 1.0;
          double Jup = 0.0, Jdn = 0.0;
          const double flux = richards_flux<JAC>(prev, cur, g, 0, Jup, Jdn);
          if (RES) { f_prev -= flux; f_cur += flux; }
          if (JAC) { dconn_prev += Jup; up_prev = Jdn; lo_cur = -Jup; dconn_cur += -Jdn; }
        }
        const int kp = k - 1;
        if (kp >= k0) finish_row(kp, i - tw, false);  // row kp is owned and now complete (kp < nz-1 holds here)
      }
      prev = cur; prev_act = act;
      f_prev = f_cur; dconn_prev = dconn_cur; dacc_prev = dacc_cur; lo_prev = lo_cur; up_prev = 0.0;
      src_prev = L.src; Fin_prev = L.Fin;
    };

    // DEPTH register sets rotate (named by a compile-time index: no copies).  Two levels in flight per wave; the residual-only pass (which
    // streams least per level) was also built 3 and 4 deep in round 4: 0.716 / 0.709 / 0.713 ms at depth 3 / 2 / 4, 1024 x 1024 x 100
    // (profiles/r04/kbench_march_depth.json): depth is not what holds it at 0.67 of the peak on the bytes it moves
    constexpr int DEPTH = (MODE == 0 && !GEN) ? kMarchDepthResidual : 2;
    Level LV[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { LV[d] = fetch(kstart + d); __builtin_amdgcn_sched_barrier(0); }
    for (int k = kstart; k < kend; k += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (d == 0 || k + d < kend) {
          process(k + d, LV[d]);
          __builtin_amdgcn_sched_barrier(0);   // program order = issue order: a register set is refilled right after its use
          LV[d] = fetch(k + d + DEPTH);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (kend == nz && k1 == nz) finish_row(nz - 1, cbase + (size_t)(nz - 1) * tw, true);  // top row of the column
  }
  if (FUSE && RES) {   // (not folded: the march is at its SGPR limit, the descriptor's kernel arguments would spill)
    double t = block_sum(accF, red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    t = block_sum(accP, red);
    if (threadIdx.x == 0) a.partials[(size_t)gridDim.x + blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// general meshes, pass 1: auxvars of every cell (UpdateAuxVarsIntrn, GE:1417-1442) + accumulation term
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_auxvars(GenArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  MPP_STAGE_POW(a.cell.powtab);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const CellArrays &ca = a.cell;
  double f = 0.0, dacc = 1.0;
  Aux x = {};
  if (is_active(ca, i)) {
    Soil s = load_soil(ca, i);
    x = auxvar_compute<JAC>(s.sf, ca.eos, s, ca.P[i], frac_of(ca, i), pt);
    if (owns_row(ca, i)) {
      const double vol = ca.vol[i];
      if (ca.por_slope) {   // linear porosity model: all three terms of RichardsODEPressureAccumDeriv (GE:1675-1677)
        double por, dpor;
        porosity(s, true, ca.P[i], por, dpor);
        if (RES) f = por * x.den * x.sat * vol * a.dtInv;
        if (JAC) dacc = (dpor * x.den * x.sat + por * x.dden * x.sat + por * x.den * x.dsat) * vol * a.dtInv;
      } else {
        if (RES) f = s.por * x.den * x.sat * vol * a.dtInv;
        if (JAC) dacc = (s.por * x.dden * x.sat + s.por * x.den * x.dsat) * vol * a.dtInv;
      }
    }
  }
  a.aux.kr[i] = x.kr;
  a.aux.den[i] = x.den;
  if (JAC) { a.aux.dkr[i] = x.dkr; a.aux.dden[i] = x.dden; }
  if (RES) a.F[i] = f - a.accum_prev[i];
  if (JAC) a.val[a.diag_slot[i]] = dacc;
}

// general meshes, pass 2: one thread per row gathers its incident connections in the caller's connection order
// (same per-row sum order as the serial loops at GE:1756-1810 and 2002-2071); no atomics, no write conflicts.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_flux_rows(GenArgs a) {
  const int *__restrict__ rowptr = a.rowptr;
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const CellArrays &ca = a.cell;
  const int dslot = JAC ? a.diag_slot[i] : 0;
  if (JAC) {
    for (int s = rowptr[i]; s < rowptr[i + 1]; ++s)
      if (s != dslot) a.val[s] = 0.0;
  }
  double f = RES ? a.F[i] : 0.0;
  if (owns_row(ca, i)) {
    Aux me;
    me.P = ca.P[i]; me.kr = a.aux.kr[i]; me.den = a.aux.den[i];
    me.dkr = JAC ? a.aux.dkr[i] : 0.0; me.dden = JAC ? a.aux.dden[i] : 0.0;
    me.sat = me.dsat = 0.0;
    double dconn = 0.0;
    for (int e = a.rc_ptr[i]; e < a.rc_ptr[i + 1]; ++e) {
      const int enc = a.rc_conn[e];
      const int ic = enc >> 1, side = enc & 1;
      const int o = side ? a.cup[ic] : a.cdn[ic];
      if (!is_active(ca, o)) continue;
      Aux ot;
      ot.P = ca.P[o]; ot.kr = a.aux.kr[o]; ot.den = a.aux.den[o];
      ot.dkr = JAC ? a.aux.dkr[o] : 0.0; ot.dden = JAC ? a.aux.dden[o] : 0.0;
      ot.sat = ot.dsat = 0.0;
      ConnGeom g;
      g.Dq = a.conn.Dq[ic]; g.upweight = a.conn.upw[ic]; g.dist_gravity = a.conn.dgrav[ic]; g.area = a.conn.area[ic];
      double Jup = 0.0, Jdn = 0.0;
      if (side == 0) {  // this row is the upwind cell of the connection
        const double flux = richards_flux<JAC>(me, ot, g, 0, Jup, Jdn);
        if (RES) f -= flux;
        if (JAC) { dconn += Jup; a.val[a.rc_slot[e]] += Jdn; }
      } else {
        const double flux = richards_flux<JAC>(ot, me, g, 0, Jup, Jdn);
        if (RES) f += flux;
        if (JAC) { dconn += -Jdn; a.val[a.rc_slot[e]] += -Jup; }
      }
    }
    if (JAC) a.val[dslot] = dconn + a.val[dslot];
  }
  if (RES) {
    if (a.src) f -= a.src[i];
    a.F[i] = f;
  }
}

// The same pass on an ELL copy of the incidence lists (ell_*[k*n + row], k-th incident connection of the row in the caller's
// connection order, -1 when the row has fewer): every index a row needs arrives with one coalesced load, so the loads of
// all its neighbours are in flight together instead of a chain of dependent look-ups per connection; off-diagonal entries
// are stored once (every pattern entry of a row belongs to exactly one incident connection: checked at set-up).
// LAT (round 5): the lateral half of a 3-D mesh of complete soil columns -- the column march has done the accumulation term and the
// chain links (connection index < n in the internal numbering) and left the auxvars; those entries of the lists are skipped here.
template <int MODE, int K, bool LAT = false>
__global__ __launch_bounds__(kBlock) void k_flux_rows_ell(GenArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const CellArrays &ca = a.cell;
  const size_t n = (size_t)a.n;
  int enc[K], oth[K], slot[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    enc[k] = k < a.ell_k ? a.ell_enc[k * n + i] : -1;
    if (LAT && enc[k] >= 0 && (enc[k] >> 1) < a.n) enc[k] = -1;   // a chain link: the march's
    oth[k] = enc[k] >= 0 ? a.ell_o[k * n + i] : (int)i;
    slot[k] = (JAC && enc[k] >= 0) ? a.ell_slot[k * n + i] : -1;
  }
  const bool mine = owns_row(ca, i);
  Aux me;
  me.P = ca.P[i]; me.kr = a.aux.kr[i]; me.den = a.aux.den[i];
  me.dkr = JAC ? a.aux.dkr[i] : 0.0; me.dden = JAC ? a.aux.dden[i] : 0.0;
  me.sat = me.dsat = 0.0;
  double f = RES ? a.F[i] : 0.0;
  double dconn = 0.0;
  // every operand of every neighbour is requested before the first flux is evaluated: one memory round trip per row
  Aux ot[K];
  ConnGeom g[K];
  bool live[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int o = oth[k];
    const int ic = enc[k] >= 0 ? (enc[k] >> 1) : 0;
    live[k] = mine && enc[k] >= 0 && is_active(ca, o);
    ot[k].P = ca.P[o]; ot[k].kr = a.aux.kr[o]; ot[k].den = a.aux.den[o];
    ot[k].dkr = JAC ? a.aux.dkr[o] : 0.0; ot[k].dden = JAC ? a.aux.dden[o] : 0.0;
    ot[k].sat = ot[k].dsat = 0.0;
    // (LAT: conn.Dq holds conductance x face area -- the flux is linear in it, as in the march -- and conn.area is null: three gathers)
    g[k].Dq = a.conn.Dq[ic]; g[k].upweight = a.conn.upw[ic]; g[k].dist_gravity = a.conn.dgrav[ic];
    g[k].area = (LAT && !a.conn.area) ? 1.0 : a.conn.area[ic];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int side = enc[k] & 1;
    double Jup = 0.0, Jdn = 0.0;
    const double flux = side == 0 ? richards_flux<JAC>(me, ot[k], g[k], 0, Jup, Jdn) : richards_flux<JAC>(ot[k], me, g[k], 0, Jup, Jdn);
    if (RES && live[k]) f = side == 0 ? f - flux : f + flux;
    if (JAC) {
      if (live[k]) dconn += side == 0 ? Jup : -Jdn;
      if (slot[k] >= 0) a.val[slot[k]] = live[k] ? (side == 0 ? Jdn : -Jup) : 0.0;
    }
  }
  if (LAT && JAC && a.tri3) {   // the march left the column's tridiagonal in three arrays: this pass writes the whole row
    const double lo = ldnt(a.tri3 + i), di = ldnt(a.tri3 + n + i), up = ldnt(a.tri3 + 2 * n + i);
    const int sl = a.slot_lo[i], su = a.slot_up[i];
    if (sl >= 0) a.val[sl] = lo;
    a.val[a.diag_slot[i]] = mine ? dconn + di : di;
    if (su >= 0) a.val[su] = up;
  } else if (JAC && mine) a.val[a.diag_slot[i]] = dconn + a.val[a.diag_slot[i]];
  if (RES) {
    if (a.src) f -= a.src[i];
    a.F[i] = f;
  }
}

// ------------------------------------------------------------------------------------------------
// boundary conditions (GE:1815-1877 residual, 2076-2139 Jacobian): the BC auxvar has the soil of its id_dn
// cell (MultiPhysicsProbVSFM.F90:422-446) and pressure = condition value for Dirichlet/seepage, = P_dn for
// mass-flux conditions (GE:1517-1531).
// MODE 0 residual (F += flux), 1 Jacobian (diag += -Jdn), 2 both, 4 flux output [kg/s]
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_bc(BcArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2), OUT = (MODE == 4);
  MPP_STAGE_POW(a.cell.powtab);
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= a.ncond_cells) return;
  const CellArrays &ca = a.cell;
  const size_t cell = a.cc_cell[j];
  if (!owns_row(ca, cell)) {   // inactive (or ghost) cell: no boundary flux
    if (a.side_f) { if (RES) a.side_f[j] = 0.0; if (JAC) a.side_d[j] = 0.0; }
    return;
  }
  Soil s = load_soil(ca, cell);
  const double Pc = a.trial_x ? fma(a.trial_a, a.trial_y[cell], a.trial_x[cell]) : ca.P[cell];
  const Aux in = auxvar_compute<JAC>(s.sf, ca.eos, s, Pc, frac_of(ca, cell), pt);
  const bool side = a.side_f != nullptr;   // fluxes into per-condition-cell side arrays (complete-row column march)
  double f = (RES && !side) ? a.F[cell] : 0.0;
  double d = 0.0;
  for (int e = a.cc_ptr[j]; e < a.cc_ptr[j + 1]; ++e) {
    const int t = a.type[e];
    const int ctype = t & 0xffff, swap = t >> 16;
    const double pb = (ctype == kCondMassFlux) ? in.P : a.value[e];
    const Aux bc = auxvar_compute<JAC>(s.sf, ca.eos, s, pb, ca.frac_liq, pt);
    ConnGeom g;
    g.Dq = a.Dq[e]; g.upweight = 0.0; g.dist_gravity = a.dgrav[e]; g.area = a.area[e];
    double Jup = 0.0, Jdn = 0.0, flux, jdn_outer;
    if (!swap) {
      flux = richards_flux<JAC>(bc, in, g, ctype, Jup, Jdn);
      jdn_outer = Jdn;
    } else {  // RichardsFlux swap_order (RichardsMod.F90:96-112)
      flux = -richards_flux<JAC>(in, bc, g, ctype, Jup, Jdn);
      jdn_outer = -Jup;
    }
    if (RES) f += flux;
    if (JAC) d += -jdn_outer;
    if (OUT) a.flux_out[e] = flux * kFMWH2O;
  }
  if (side) {
    if (RES) a.side_f[j] = f;
    if (JAC) a.side_d[j] = d;
    return;
  }
  if (RES) a.F[cell] = f;
  if (JAC) a.val[a.diag_slot[cell]] += d;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_sinks(SinkArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2), OUT = (MODE == 4);
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= a.ncells) return;
  const size_t cell = a.cell[j];
  if (a.active && !a.active[cell]) return;
  const double dP = a.P[cell] - kPressureRef;
  double f = RES ? a.F[cell] : 0.0, d = 0.0;
  for (int e = a.ptr[j]; e < a.ptr[j + 1]; ++e) {
    const double v = a.value[e], Pc = a.pc[e], n = a.expo[e];
    const bool campbell = (a.type[e] == 512);
    const double pw = (dP <= 0.0) ? pow(dP / Pc, n) : 0.0;
    double rate;   // actual rate [kg/s]
    if (campbell) {
      const double factor = (dP <= 0.0) ? 1.0 + pw : 1.0;
      rate = v / factor;
      if (RES) f = f - v / factor / kFMWH2O;
      if (JAC && dP <= 0.0) d += (v / kFMWH2O) * (n * pw) / (dP * (factor * factor));
    } else {
      const double factor = (dP <= 0.0) ? exp(-pw) : 1.0;
      rate = v * factor;
      if (RES) f = f - v * factor / kFMWH2O;
      if (JAC && dP <= 0.0) d += (v / kFMWH2O) * (n * pw) * factor / dP;
    }
    if (OUT) a.flux_out[e] = rate;
  }
  if (RES) a.F[cell] = f;
  if (JAC) a.val[a.diag_slot[cell]] += d;
}

__global__ __launch_bounds__(kBlock) void k_conn_flux(CellArrays ca, ConnArrays cn, const int *cup, const int *cdn,
                                                      int nc, double *out) {
  MPP_STAGE_POW(ca.powtab);
  const int ic = blockIdx.x * kBlock + threadIdx.x;
  if (ic >= nc) return;
  const size_t u = cup[ic], d = cdn[ic];
  double flux = 0.0;
  if (is_active(ca, u) && is_active(ca, d)) {
    Soil su = load_soil(ca, u), sd = load_soil(ca, d);
    Aux au = auxvar_compute<false>(su.sf, ca.eos, su, ca.P[u], frac_of(ca, u), pt);
    Aux ad = auxvar_compute<false>(sd.sf, ca.eos, sd, ca.P[d], frac_of(ca, d), pt);
    ConnGeom g;
    g.Dq = cn.Dq[ic]; g.upweight = cn.upw[ic]; g.dist_gravity = cn.dgrav[ic]; g.area = cn.area[ic];
    double j1, j2;
    flux = richards_flux<false>(au, ad, g, 0, j1, j2) * kFMWH2O;  // internal_flux = flux*FMWH2O (GE:1809)
  }
  out[ic] = flux;
}

// soe auxvar fields filled by RichardsODEPressureSetDataInSOEAuxVar (GE:1170-1195)
__global__ __launch_bounds__(kBlock) void k_cell_var(CellArrays ca, int n, int var_type, double *out) {
  MPP_STAGE_POW(ca.powtab);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  Soil s = load_soil(ca, i);
  const double P = ca.P[i];
  Aux a = auxvar_compute<false>(s.sf, ca.eos, s, P, frac_of(ca, i), pt);
  double v, por, dpor;
  porosity(s, ca.por_slope != nullptr, P, por, dpor);
  if (var_type == 608) v = a.sat;
  else if (var_type == 610) v = por * a.den * kFMWH2O * a.sat * ca.vol[i];
  else v = (P - kPressureRef) / (a.den * kFMWH2O * kGravity);
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// CSR SpMV y = A x  (fp64 values, int32 columns, int32 rowptr) -- PETSc MatMult_SeqAIJ's role
// ------------------------------------------------------------------------------------------------
// variant 0: one lane per row.  In level-major ordering the 64 rows of a wave are 64 neighbouring columns at one
// depth: the x gathers (i-ncols, i, i+ncols) are three contiguous 512-byte segments, rowptr/y are coalesced and
// the val/col streams are contiguous per wave (1536 B / 768 B for 3-point rows).
__global__ __launch_bounds__(kBlock) void k_spmv_row(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, const double *__restrict__ x,
                                                     double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  const int s = rowptr[i], e = rowptr[i + 1];
  double sum = 0.0;
  for (int k = s; k < e; ++k) sum += val[k] * x[col[k]];
  y[i] = sum;
}

// variant 1: CSR-stream.  The block's val/col segment is read with fully coalesced loads, the products are
// staged in LDS, then each lane sums its row from LDS in row order (same sum order as variant 0).
constexpr int kStreamCap = 256 * 8;  // products per block in LDS (16 KiB); larger blocks fall back to variant 0
__global__ __launch_bounds__(kBlock) void k_spmv_stream(int n, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double prod[kStreamCap];
  const int r0 = blockIdx.x * kBlock;
  const int nrows = min(kBlock, n - r0);
  const int tid = threadIdx.x;
  const int s0 = rowptr[r0];
  const int cnt = rowptr[r0 + nrows] - s0;
  const int r = r0 + tid;
  if (cnt <= kStreamCap) {
    for (int k = tid; k < cnt; k += kBlock) prod[k] = val[s0 + k] * x[col[s0 + k]];
    __syncthreads();
    if (tid < nrows) {
      const int s = rowptr[r] - s0, e = rowptr[r + 1] - s0;
      double sum = 0.0;
      for (int k = s; k < e; ++k) sum += prod[k];
      y[r] = sum;
    }
  } else if (tid < nrows) {
    double sum = 0.0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) sum += val[k] * x[col[k]];
    y[r] = sum;
  }
}

// variant 17: software-pipelined wave-level CSR-stream.  One wave walks TPW consecutive 64-row tiles; while the x
// gathers, the LDS row sums and the y store of tile t run, the val/col/rowptr loads of tile t+1 are already in
// flight (issued AFTER the gathers of tile t: vmcnt retires in order, so the gathers can be waited for while the
// younger prefetch stays outstanding).  Steady state costs one memory round trip per tile instead of three.
template <int J>
__global__ __launch_bounds__(kBlock) void k_spmv_pipe(int n, int tpw, const int *__restrict__ rowptr,
                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double prod[kBlock / 64][64 * J];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *lp = prod[w];
  const int ntiles = (n + 63) >> 6;
  const int tile0 = (blockIdx.x * (kBlock / 64) + w) * tpw;
  if (tile0 >= ntiles) return;
  const int tend = min(tile0 + tpw, ntiles);
  int sA = rowptr[tile0 << 6], sB = rowptr[min((tile0 + 1) << 6, n)];
  double vn[J];
  int cn[J], rsn = 0, ren = 0;
  auto prefetch = [&](int tile, int s0, int cnt) {
    const int r = (tile << 6) + lane;
    if (r < n) {
      rsn = __builtin_nontemporal_load(rowptr + r);
      ren = __builtin_nontemporal_load(rowptr + r + 1);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k = lane + 64 * j;
      const bool ok = k < cnt;
      vn[j] = ok ? __builtin_nontemporal_load(val + s0 + k) : 0.0;
      cn[j] = ok ? __builtin_nontemporal_load(col + s0 + k) : 0;
    }
  };
  prefetch(tile0, sA, sB - sA);
  for (int tile = tile0; tile < tend; ++tile) {
    const int s0 = sA, cnt = sB - sA;
    double vc[J];
    int cc[J];
    const int rs = rsn, re = ren;
#pragma unroll
    for (int j = 0; j < J; ++j) { vc[j] = vn[j]; cc[j] = cn[j]; }
    // x gathers of this tile first ...
    double xs[J];
#pragma unroll
    for (int j = 0; j < J; ++j) xs[j] = (lane + 64 * j < cnt) ? x[cc[j]] : 0.0;
    __builtin_amdgcn_sched_barrier(0);
    // ... then the next tile's streams, which stay in flight behind them
    if (tile + 1 < tend) {
      sA = sB;
      sB = rowptr[min((tile + 2) << 6, n)];
      prefetch(tile + 1, sA, sB - sA);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (lane + 64 * j < cnt) lp[lane + 64 * j] = vc[j] * xs[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = (tile << 6) + lane;
    if (r < n) {
      double sum = 0.0;
      for (int k = rs - s0; k < re - s0; ++k) sum += lp[k];
      __builtin_nontemporal_store(sum, y + r);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// variants 20/21: CSR SpMV on a COMPACT INDEX built once from the (fixed) CSR pattern: per row a 16-bit start offset
// inside its 64-row tile instead of the 32-bit row pointer, per entry a 16-bit column offset from the tile's first row instead
// of the 32-bit column -- 8 index bytes per 3-point row instead of 16, i.e. 47.8 instead of 55.8 bytes per row of HBM traffic.
// Available when every |col - 64*tile| < 2^15 (true for the tiled level-major layout, where a row's neighbours are within
// one tile width); values, x, y and the summation order are unchanged.
// SPLIT (field-blocked systems, rows [P(1..N); T(1..N)]): a row's columns lie near its own cell in either block, so the
// 16-bit entry is (offset << 1) | block with the offset taken from the tile's first cell in that block (split = N).
// DOTS: the same product with the line search's reductions fused in and y never written: leaves the partial sums of x.x,
// f.(A x) and (xx != nullptr) xx.xx, layout partials[j*gridDim.x + b] -- SNESLineSearchApply_BT needs only ||Y||, F.(J Y), ||X||.
// Schedule (round 2; profiles/r02_kbench_spmv_depth.json): a wave owns `tpw` consecutive tiles and keeps the index, values and
// row data of up to THREE of them in flight in three named register sets that rotate -- no copies, so using a set waits for
// that set only; the set two tiles ahead is requested right behind a tile's x gathers.  With 2 tiles per wave (the default)
// every load of a wave is in flight at once: 0.880 -> 0.812 ms at 105 M rows against the one-tile look-ahead of round 1
// (3 tiles 0.833, 4: 0.842, 6: 0.875, 12: 0.90; 1 tile per wave 0.996: too many waves for too little work each).
template <int J, bool SPLIT, bool DOTS>
__global__ __launch_bounds__(kBlock) void k_spmv_c(int n, int tpw, int split, const int *__restrict__ tileptr,
                                                   const unsigned short *__restrict__ roff, const short *__restrict__ dcol,
                                                   const double *__restrict__ val, const double *__restrict__ x,
                                                   double *__restrict__ y, const double *__restrict__ f,
                                                   const double *__restrict__ xx, double *__restrict__ partials) {
  __shared__ double prod[kBlock / 64][64 * J];
  __shared__ double red[kBlock / 64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *lp = prod[w];
  const int ntiles = (n + 63) >> 6;
  const int tile0 = (blockIdx.x * (kBlock / 64) + w) * tpw;
  if (!DOTS && tile0 >= ntiles) return;           // (with the dots, idle waves still take part in the block sums)
  const int tend = min(tile0 + tpw, ntiles);      // tend <= tile0 for an idle wave
  struct Set { double v[J], fr, xr, qr; int c[J]; int rs, cnt; };
  auto fetch = [&](int tile) {
    Set S;
    const bool live = tile < tend;                 // wave-uniform; a dead set loads nothing (cnt = 0)
    const int tc = live ? tile : max(tend - 1, 0);
    const int s0 = live ? tileptr[tc] : 0;
    S.cnt = live ? tileptr[tc + 1] - s0 : 0;
    const int r = (tc << 6) + lane;
    const bool row = live && r < n;
    S.rs = row ? (int)__builtin_nontemporal_load(roff + r) : S.cnt;
    if (DOTS) {
      S.fr = row ? __builtin_nontemporal_load(f + r) : 0.0;
      S.xr = row ? x[r] : 0.0;
      S.qr = (row && xx) ? __builtin_nontemporal_load(xx + r) : 0.0;
    } else S.fr = S.xr = S.qr = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k = lane + 64 * j;
      const bool ok = k < S.cnt;
      S.v[j] = ok ? __builtin_nontemporal_load(val + s0 + k) : 0.0;
      S.c[j] = ok ? (int)__builtin_nontemporal_load(dcol + s0 + k) : 0;
    }
    return S;
  };
  double acc_xx = 0.0, acc_fw = 0.0, acc_qq = 0.0;
  auto body = [&](int tile, const Set &S, auto refill) {
    const int base = tile << 6;
    const int base_lo = SPLIT ? (base < split ? base : base - split) : base;
    double xs[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int ci = SPLIT ? base_lo + (S.c[j] >> 1) + ((S.c[j] & 1) ? split : 0) : base + S.c[j];
      xs[j] = (lane + 64 * j < S.cnt) ? x[ci] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    refill();                                      // the set two tiles ahead goes out behind this tile's gathers
    __builtin_amdgcn_sched_barrier(0);
    // end of this lane's row = start of the next lane's (rows beyond n carry `cnt`)
    const int nxt = __shfl_down(S.rs, 1, 64);
    const int re = (lane == 63) ? S.cnt : nxt;
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (lane + 64 * j < S.cnt) lp[lane + 64 * j] = S.v[j] * xs[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = base + lane;
    if (r < n) {
      double sum = 0.0;
      for (int k = S.rs; k < re; ++k) sum += lp[k];
      if (DOTS) { acc_fw += S.fr * sum; acc_xx += S.xr * S.xr; acc_qq += S.qr * S.qr; }
      else __builtin_nontemporal_store(sum, y + r);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  Set A = fetch(tile0), B = fetch(tile0 + 1), C = fetch(tend);   // C starts dead (loads nothing)
  for (int tile = tile0; tile < tend; tile += 3) {
    body(tile, A, [&] { C = fetch(tile + 2); });
    if (tile + 1 < tend) body(tile + 1, B, [&] { A = fetch(tile + 3); });
    if (tile + 2 < tend) body(tile + 2, C, [&] { B = fetch(tile + 4); });
  }
  if (DOTS) {
    double t = block_sum(acc_xx, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    t = block_sum(acc_fw, red);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t;
    if (xx) {
      t = block_sum(acc_qq, red);
      if (threadIdx.x == 0) partials[(size_t)2 * gridDim.x + blockIdx.x] = t;
    }
  }
}

// The line search's product on a vertical-only column grid without an index (round 4): x.x, f.(A x), q.q (q may be null) with A the
// block diagonal of column tridiagonals whose entries lie at computed places -- rows of a column tile are level-major, 2 entries in
// the end rows and 3 between (the layout k_tri_solve_lds reads).  One lane per row; a wave's 64 rows of a middle level of a full
// tile are 192 consecutive values: three coalesced loads and a transposition through LDS, as there; any other row reads its two or
// three values directly.  Row sums as k_spmv_c forms them (products rounded one by one, lower + diagonal + upper in that order).
// 41 instead of 49 bytes per row; a block walks kColsDotsChunks pieces of 256 rows so that the second stage sums a few thousand
// partial sums.  Partial sums: partials[j * gridDim.x + block], j = 0 (x.x), 1 (f.(A x)), 2 (q.q).
constexpr int kColsDotsChunks = 8;
__global__ __launch_bounds__(kBlock) void k_spmv_cols_dots(int n, int ncols, int nz, const double *__restrict__ val,
                                                           const double *__restrict__ x, const double *__restrict__ f,
                                                           const double *__restrict__ q, double *__restrict__ partials) {
#pragma clang fp contract(off)
  __shared__ double tr[kBlock / 64][3 * 64];
  __shared__ double red[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_tile = (size_t)64 * nz;
  double acc_xx = 0.0, acc_fw = 0.0, acc_qq = 0.0;
  for (int ch = 0; ch < kColsDotsChunks; ++ch) {
    const size_t i = ((size_t)blockIdx.x * kColsDotsChunks + ch) * kBlock + threadIdx.x;
    const size_t i0 = i - lane;
    if (i0 >= (size_t)n) continue;   // (the whole wave)
    const int tile = (int)(i0 / per_tile), tw = min(64, ncols - (tile << 6));
    const double *vt = val + (size_t)tile * 64 * (3 * (size_t)nz - 2);   // this tile's values (tiles before it are full)
    const bool row = i < (size_t)n;
    // a full tile: the wave's rows are one level
    const int k0 = (int)((i0 - (size_t)tile * per_tile) >> 6);
    const bool staged = tw == 64 && k0 > 0 && k0 < nz - 1;
    double lo = 0.0, di = 0.0, up = 0.0;
    int k = 0, c = 0;
    if (row) {
      const size_t rel = i - (size_t)tile * per_tile;
      k = (int)(rel / tw); c = (int)(rel - (size_t)k * tw);
    }
    if (staged) {
      const size_t b = (size_t)2 * 64 + (size_t)3 * (k0 - 1) * 64 + lane;
      const double e0 = ldnt(vt + b), e1 = ldnt(vt + b + 64), e2 = ldnt(vt + b + 128);
      tr[wave][lane] = e0; tr[wave][64 + lane] = e1; tr[wave][128 + lane] = e2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      lo = tr[wave][3 * lane]; di = tr[wave][3 * lane + 1]; up = tr[wave][3 * lane + 2];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (these reads precede the next piece's writes)
      __builtin_amdgcn_wave_barrier();
    } else if (row) {
      if (k == 0) { di = ldnt(vt + 2 * c); up = ldnt(vt + 2 * c + 1); }
      else if (k == nz - 1) { const size_t s0 = (size_t)2 * tw + (size_t)3 * (nz - 2) * tw + 2 * c; lo = ldnt(vt + s0); di = ldnt(vt + s0 + 1); }
      else { const size_t s0 = (size_t)2 * tw + (size_t)3 * (k - 1) * tw + 3 * c; lo = ldnt(vt + s0); di = ldnt(vt + s0 + 1); up = ldnt(vt + s0 + 2); }
    }
    if (row) {
      const double xm = k > 0 ? x[i - tw] : 0.0, x0 = x[i], xp = k < nz - 1 ? x[i + tw] : 0.0;
      // (an entry the row does not have is 0 here and meets x = 0: the sum of the products that exist, in entry order)
      double sum = 0.0;
      if (k > 0) sum += lo * xm;
      sum += di * x0;
      if (k < nz - 1) sum += up * xp;
      acc_fw += ldnt(f + i) * sum;
      acc_xx += x0 * x0;
      if (q) { const double qv = ldnt(q + i); acc_qq += qv * qv; }
    }
  }
  double t = block_sum(acc_xx, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  t = block_sum(acc_fw, red);
  if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t;
  if (q) {
    t = block_sum(acc_qq, red);
    if (threadIdx.x == 0) partials[(size_t)2 * gridDim.x + blockIdx.x] = t;
  }
}

// variants 22/23: the compact index for patterns whose columns lie further than 2^15 rows from their tile (3-D grids: the
// y-neighbour of a cell sits nx/64 column tiles away).  A tile's columns cluster around a few places -- its own rows, the
// tiles next to it, the two y-neighbour tiles -- so each tile carries up to 8 window bases (int32) and an entry is
// (window << 13) | offset, 13-bit offset from its window's base.  Same 8 index bytes per 3-point row (+0.5 byte per row
// for the bases), same values, x, y and summation order as every other variant.
// (103 registers at J = 7; asked for five waves per SIMD -- 96 registers, 36 bytes of scratch -- the 7-point product went from 0.445 to
// 0.538 ms: the spills sit in its inner loop; off, profiles/r05/occupancy_hints_ab.json)
#ifndef MPP_SPMV_MB_WAVES
#define MPP_SPMV_MB_WAVES 1
#endif
template <int J>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MPP_SPMV_MB_WAVES))) void k_spmv_pipe_mb(int n, int tpw, const int *__restrict__ tileptr,
                                                         const unsigned short *__restrict__ roff,
                                                         const unsigned short *__restrict__ mcol, const int *__restrict__ tbase,
                                                         const double *__restrict__ val, const double *__restrict__ x,
                                                         double *__restrict__ y) {
  __shared__ double prod[kBlock / 64][64 * J];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *lp = prod[w];
  const int ntiles = (n + 63) >> 6;
  const int tile0 = (blockIdx.x * (kBlock / 64) + w) * tpw;
  if (tile0 >= ntiles) return;
  const int tend = min(tile0 + tpw, ntiles);
  struct Set { double v[J]; int c[J]; int rs, cnt, tb; };   // same three-set schedule as k_spmv_c
  auto fetch = [&](int tile) {
    Set S;
    const bool live = tile < tend;
    const int tc = live ? tile : tend - 1;
    const int s0 = live ? tileptr[tc] : 0;
    S.cnt = live ? tileptr[tc + 1] - s0 : 0;
    const int r = (tc << 6) + lane;
    S.rs = (live && r < n) ? (int)__builtin_nontemporal_load(roff + r) : S.cnt;
    S.tb = live ? tbase[(size_t)tc * 8 + (lane & 7)] : 0;   // lanes 0..7 hold the tile's window bases
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k = lane + 64 * j;
      const bool ok = k < S.cnt;
      S.v[j] = ok ? __builtin_nontemporal_load(val + s0 + k) : 0.0;
      S.c[j] = ok ? (int)__builtin_nontemporal_load(mcol + s0 + k) : 0;
    }
    return S;
  };
  auto body = [&](int tile, const Set &S, auto refill) {
    const int base = tile << 6;
    double xs[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int wb = __shfl(S.tb, S.c[j] >> 13, 64);         // base of the entry's window
      xs[j] = (lane + 64 * j < S.cnt) ? x[wb + (S.c[j] & 8191)] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    refill();
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = __shfl_down(S.rs, 1, 64);
    const int re = (lane == 63) ? S.cnt : nxt;
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (lane + 64 * j < S.cnt) lp[lane + 64 * j] = S.v[j] * xs[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = base + lane;
    if (r < n) {
      double sum = 0.0;
      for (int k = S.rs; k < re; ++k) sum += lp[k];
      __builtin_nontemporal_store(sum, y + r);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  Set A = fetch(tile0), B = fetch(tile0 + 1), C = fetch(tend);
  for (int tile = tile0; tile < tend; tile += 3) {
    body(tile, A, [&] { C = fetch(tile + 2); });
    if (tile + 1 < tend) body(tile + 1, B, [&] { A = fetch(tile + 3); });
    if (tile + 2 < tend) body(tile + 2, C, [&] { B = fetch(tile + 4); });
  }
}

// ------------------------------------------------------------------------------------------------
// block-Jacobi preconditioner, one block per soil column = exact LU of the column's tridiagonal (Thomas).
// factor: l_k = a_k / b'_{k-1}, b'_k = b_k - l_k c_{k-1};  stored: l, 1/b', c   (level-major, coalesced)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tri_factor(TriArgs a) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
  const size_t cbase = (size_t)tile * 64 * a.nz + (c & 63);
  double inv_prev = 0.0, c_prev = 0.0;
#pragma unroll 4
  for (int k = 0; k < a.nz; ++k) {
    const size_t i = cbase + (size_t)k * tw;
    const int sl = a.tri_lo[i], su = a.tri_up[i];
    const double lo = sl >= 0 ? a.val[sl] : 0.0;
    const double di = a.val[a.tri_di[i]];
    const double up = su >= 0 ? a.val[su] : 0.0;
    const double l = lo * inv_prev;
    const double piv = fma(-l, c_prev, di);
    const double inv = 1.0 / piv;
    a.l[i] = l; a.invb[i] = inv; a.c[i] = tri_scale(up, inv);
    inv_prev = inv; c_prev = up;
  }
}

// the column LU and its application along linked chains of cells (ColChainArgs: a rank's piece of a split 3-D mesh); k_tri_factor's and
// k_tri_solve's arithmetic, level by level
__global__ __launch_bounds__(kBlock) void k_chain_tri_factor(ColChainArgs a) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.nchains) return;
  double inv_prev = 0.0, c_prev = 0.0;
  for (int i = a.start[c]; i >= 0; i = a.nxt[i]) {
    const int sl = a.t.tri_lo[i], su = a.t.tri_up[i];
    const double lo = sl >= 0 ? a.t.val[sl] : 0.0;
    const double di = a.t.val[a.t.tri_di[i]];
    const double up = su >= 0 ? a.t.val[su] : 0.0;
    const double l = lo * inv_prev;
    const double piv = fma(-l, c_prev, di);
    const double inv = 1.0 / piv;
    a.t.l[i] = l; a.t.invb[i] = inv; a.t.c[i] = tri_scale(up, inv);
    inv_prev = inv; c_prev = up;
  }
}
__global__ __launch_bounds__(kBlock) void k_chain_tri_solve(ColChainArgs a, const double *__restrict__ r, double *__restrict__ z) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.nchains) return;
  double prev = 0.0;
  int last = -1;
  for (int i = a.start[c]; i >= 0; i = a.nxt[i]) {
    const double d = r[i] - a.t.l[i] * prev;
    z[i] = d;
    prev = d;
    last = i;
  }
  double next = 0.0;
  for (int i = last; i >= 0; i = a.prv[i]) {
    const double xk = tri_back(tri_scale(z[i], a.t.invb[i]), a.t.c[i], next);
    z[i] = xk;
    next = xk;
  }
}

__global__ __launch_bounds__(kBlock) void k_tri_solve(TriArgs a, const double *__restrict__ r, double *__restrict__ z) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
  const size_t cbase = (size_t)tile * 64 * a.nz + (c & 63);
  double prev = 0.0;
#pragma unroll 4
  for (int k = 0; k < a.nz; ++k) {
    const size_t i = cbase + (size_t)k * tw;
    const double d = ldnt(r + i) - ldnt(a.l + i) * prev;
    z[i] = d;
    prev = d;
  }
  double next = 0.0;
#pragma unroll 4
  for (int k = a.nz - 1; k >= 0; --k) {
    const size_t i = cbase + (size_t)k * tw;
    const double xk = tri_back(tri_scale(z[i], ldnt(a.invb + i)), ldnt(a.c + i), next);
    stnt(xk, z + i);
    next = xk;
  }
}

__global__ __launch_bounds__(kBlock) void k_jacobi_setup(int n, const int *diag_slot, const double *val, double *invd) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) invd[i] = 1.0 / val[diag_slot[i]];
}

// ------------------------------------------------------------------------------------------------
// ILU(0), level scheduled, in the elimination order `ord` (PETSc MatLUFactorNumeric_SeqAIJ with levels = 0): rows of one
// level only read rows of earlier levels.
// ------------------------------------------------------------------------------------------------
// ILU(0) numeric factorisation of the rows at positions [lo, hi) of the level order (PETSc MatLUFactorNumeric_SeqAIJ,
// levels = 0): the symbolic part -- which entries meet -- was resolved on the host into a per-row program
__global__ __launch_bounds__(kBlock) void k_ilu_factor_program(const long long *__restrict__ ptr, const int *__restrict__ prog,
                                                               const int *__restrict__ rows, const int *__restrict__ diag_slot,
                                                               int lo, int hi, double *__restrict__ lu,
                                                               double *__restrict__ idiag) {
  const int p = lo + blockIdx.x * kBlock + threadIdx.x;
  if (p >= hi) return;
  long long w = ptr[p];
  const long long we = ptr[p + 1];
  while (w < we) {
    const int mslot = prog[w], k = prog[w + 1], m = prog[w + 2];
    w += 3;
    const double mult = lu[mslot] * idiag[k];
    lu[mslot] = mult;
    for (int t = 0; t < m; ++t, w += 2) lu[prog[w + 1]] -= mult * lu[prog[w]];
  }
  const int i = rows[p];
  idiag[i] = 1.0 / lu[diag_slot[i]];
}

// level-ordered ELL triangular solves (see kernels.hpp): one lane per position of the level, K entries each
__global__ __launch_bounds__(kBlock) void k_ilu_repack(long long total, const int *__restrict__ slot,
                                                       const double *__restrict__ lu, double *__restrict__ out) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= total) return;
  const int s = slot[e];
  out[e] = s >= 0 ? lu[s] : 0.0;
}
template <int K>
__global__ __launch_bounds__(kBlock) void k_ell_lower(int n, int Kdyn, const int *__restrict__ col, const double *__restrict__ val,
                                                      int lo, int hi, double *__restrict__ y) {
  const int p = lo + blockIdx.x * kBlock + threadIdx.x;
  if (p >= hi) return;
  const int kk = K > 0 ? K : Kdyn;
  double s = y[p];
  if (K > 0) {
    int c[K > 0 ? K : 1];
    double v[K > 0 ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) { c[k] = col[(size_t)k * n + p]; v[k] = val[(size_t)k * n + p]; }
#pragma unroll
    for (int k = 0; k < K; ++k) s -= v[k] * y[c[k]];
  } else {
    for (int k = 0; k < kk; ++k) s -= val[(size_t)k * n + p] * y[col[(size_t)k * n + p]];
  }
  y[p] = s;
}
template <int K>
__global__ __launch_bounds__(kBlock) void k_ell_upper(int n, int Kdyn, const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ idiag, int lo, int hi, double *__restrict__ x) {
  const int p = lo + blockIdx.x * kBlock + threadIdx.x;
  if (p >= hi) return;
  const int kk = K > 0 ? K : Kdyn;
  double s = x[p];
  const double id = idiag[p];
  if (K > 0) {
    int c[K > 0 ? K : 1];
    double v[K > 0 ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) { c[k] = col[(size_t)k * n + p]; v[k] = val[(size_t)k * n + p]; }
#pragma unroll
    for (int k = 0; k < K; ++k) s -= v[k] * x[c[k]];
  } else {
    for (int k = 0; k < kk; ++k) s -= val[(size_t)k * n + p] * x[col[(size_t)k * n + p]];
  }
  x[p] = s * id;
}

// chained triangular solve (see ChainArgs): flags instead of kernel boundaries.  Values other workgroups wrote during this
// launch are read and written with agent-scope atomics (they bypass the non-coherent caches).
// Two ways of handing out tiles.  a.ticket != nullptr: a workgroup takes a TICKET when it starts running (one returning
// atomic add) and owns the tiles [ticket * tpb, (ticket + 1) * tpb), in ascending order; a tile depends on lower-numbered
// tiles only and tickets go in ascending order to workgroups that are already running, so the workgroup that owns the lowest
// unfinished tile is always resident: liveness does not rest on the dispatch order.  a.ticket == nullptr: tile = blockIdx.x;
// the lowest unfinished tile is resident only as long as the hardware dispatches workgroups in index order, which it does on
// this chip but HIP does not promise -- this mode is faster (no atomic in front of a tile's loads) and is what option
// ilu_chain selects unless ilu_chain_tickets is set; the spin budget and the host's per-level redo are its safety net.
// The counter is never reset: launch number L of a chain hands out [L * nblocks, (L + 1) * nblocks), ticket_base = L * nblocks
// (unsigned arithmetic, wrap-around included).  A wait still has a spin budget; exhausting it flags the application,
// which the host then redoes with per-level launches (Solver::op_pc_apply).
// waitcnt immediate 0x0F70 = vmcnt(0) with expcnt / lgkmcnt left at their maxima in the gfx9 encoding this file targets
// (gfx950: vmcnt = bits [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]).
template <int K, bool UPPER>
__global__ __launch_bounds__(kBlock) void k_ell_chain(ChainArgs a) {
  __shared__ unsigned s_first;
  int tfirst = blockIdx.x * a.tiles_per_block;      // a.ticket == nullptr: tiles in dispatch order (see ChainArgs::ticket)
  if (a.ticket) {
    if (threadIdx.x == 0) s_first = (atomicAdd(a.ticket, 1u) - a.ticket_base) * (unsigned)a.tiles_per_block;
    __syncthreads();
    tfirst = (int)s_first;
  }
  const int tlast = min(tfirst + a.tiles_per_block, a.ntiles);
  const int kk = K > 0 ? K : a.K;
  constexpr int KR = K > 0 ? K : 1;
  for (int t = tfirst; t < tlast; ++t) {
    const int lo = a.tile_lo[t], hi = a.tile_hi[t];
    const int p = lo + threadIdx.x;
    const bool row = p < hi;
    int c[KR];
    double w[KR];
    if (K > 0 && row) {   // the tile's own matrix entries do not depend on anybody: fetch them before waiting
#pragma unroll
      for (int k = 0; k < K; ++k) { c[k] = a.col[(size_t)k * a.n + p]; w[k] = a.val[(size_t)k * a.n + p]; }
    }
    double s = row ? a.v[p] : 0.0;
    const double id = (UPPER && row) ? a.idiag[p] : 1.0;
    if ((int)threadIdx.x < a.dmax) {
      const int d = a.dep[(size_t)threadIdx.x * a.ntiles + t];
      if (d >= 0) {
        unsigned spins = 0;
        // Flags and solution entries travel as agent-scope (sc1) accesses, which are served at the device's coherence
        // point; no cache-wide invalidate / write-back (what an acquire / release fence costs, ~30 us per level here).
        while (__hip_atomic_load(a.flags + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
          __builtin_amdgcn_s_sleep(16);   // ~0.4 us between polls: thousands of waiting tiles must not flood the fabric
          if (++spins > a.max_spins) { a.sc[S_ERR] = 1.0; break; }
        }
      }
    }
    __syncthreads();
    if (row) {
      if (K > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k)
          s -= w[k] * (c[k] == p ? 0.0 : __hip_atomic_load(a.v + c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      } else {
        for (int k = 0; k < kk; ++k) {
          const int ck = a.col[(size_t)k * a.n + p];
          if (ck != p) s -= a.val[(size_t)k * a.n + p] * __hip_atomic_load(a.v + ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __hip_atomic_store(a.v + p, UPPER ? s * id : s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's sc1 stores have been acknowledged at the coherence point ...
    __syncthreads();                      // ... and so have every other wave's, before the tile is declared done
    if (threadIdx.x == 0) __hip_atomic_store(a.flags + t, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------
// soil heat diffusion.  ThermKSPTempSoilAuxVarCompute
// (src/mpp/auxvar/ThermalKSPTemperatureSoilAuxType.F90:66-168): itype 1 soil/crop, 2 wetland, 3 land ice
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void thermal_auxvar(int itype, bool shallow, double T, double liq, double ice, double snow,
                                               bool nsnow_zero, double por, double tkmg, double tkdry, double csol,
                                               double dz, double &therm_cond, double &heat_cap) {
  const double cpliq = 4.188e3, cpice = 2.11727e3, denh2o = 1.0e3, denice = 0.917e3;
  const double tkice = 2.290, tkwat = 0.57, thk_bedrock = 3.0, tfrz = 273.15;
  if (itype == 1) {
    if (shallow) {
      double satw = (liq / denh2o + ice / denice) / (dz * por);
      satw = fmin(1.0, satw);
      if (satw > .1e-6) {
        const double dke = (T >= tfrz) ? fmax(0.0, log10(satw) + 1.0) : satw;
        const double fl = (liq / (denh2o * dz)) / (liq / (denh2o * dz) + ice / (denice * dz));
        const double dksat = tkmg * pow(tkwat, fl * por) * pow(tkice, (1.0 - fl) * por);
        therm_cond = dke * dksat + (1.0 - dke) * tkdry;
      } else {
        therm_cond = tkdry;
      }
      heat_cap = csol * (1.0 - por) * dz + ice * cpice + liq * cpliq;
      if (nsnow_zero) heat_cap = heat_cap + snow * cpice;
    } else {
      therm_cond = thk_bedrock;
      heat_cap = csol * (1.0 - por) * dz + ice * cpice + liq * cpliq;
    }
    heat_cap = heat_cap / dz;
  } else if (itype == 2 && !shallow) {
    therm_cond = thk_bedrock;
    heat_cap = csol;
  } else {
    therm_cond = (T < tfrz) ? tkice : tkwat;
    heat_cap = ice * cpice + liq * cpliq;
    if (nsnow_zero) heat_cap = heat_cap + snow * cpice;
    heat_cap = heat_cap / dz;
  }
}

__global__ __launch_bounds__(kBlock) void k_thermal_aux(ThermalArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  double k, c;
  thermal_auxvar(a.itype[i], a.shallow[i] != 0, a.T[i], a.liq[i], a.ice[i], a.snow_water[i], a.nsnow_zero[i] != 0,
                 a.por[i], a.tkmg[i], a.tkdry[i], a.csol[i], a.dz[i], k, c);
  a.therm_cond[i] = k;
  a.heat_cap[i] = c;
}

// ThermalKSPTempSoilAccum + Divergence (GE :664-860) and ComputeOperatorsDiag (:1064-1147), one lane per row,
// incident connections in the caller's connection order
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_thermal_rows(ThermalArgs a) {
  constexpr bool RHS = (MODE == 0 || MODE == 2), OPR = (MODE == 1 || MODE == 2);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const bool act = a.active ? a.active[i] == 1 : true;   // a ghost cell (2) keeps an identity row
  const int dslot = a.diag_slot[i];
  if (OPR)
    for (int s = a.rowptr[i]; s < a.rowptr[i + 1]; ++s) a.val[s] = 0.0;
  double b = 0.0, diag = 1.0;
  if (act) {
    const double acc = a.heat_cap[i] * a.vol[i] / (a.dt * a.tuning[i]);
    diag = acc;
    b = acc * a.T[i];
    const double Ti = a.T[i], ki = a.therm_cond[i];
    for (int e = a.rc_ptr[i]; e < a.rc_ptr[i + 1]; ++e) {
      const int enc = a.rc_conn[e], ic = enc >> 1, side = enc & 1;
      const int o = side ? a.cup[ic] : a.cdn[ic];
      if (a.active && !a.active[o]) continue;
      const double du = a.cdu[ic], dd = a.cdd[ic], dist = du + dd, area = a.carea[ic];
      const double ku = side ? a.therm_cond[o] : ki, kd = side ? ki : a.therm_cond[o];
      const double Tu = side ? a.T[o] : Ti, Td = side ? Ti : a.T[o];
      const double kav = ku * kd * dist / (ku * dd + kd * du);
      if (RHS) {
        const double flux = -kav * (Tu - Td) / dist;
        if (side == 0) b = b + a.cnfac * flux * area * 1.0; else b = b - a.cnfac * flux * area * 1.0;
      }
      if (OPR) {
        const double value = (1.0 - a.cnfac) * kav / dist * area;
        diag += value * 1.0;
        a.val[a.rc_slot[e]] += -value * 1.0;
      }
    }
  }
  if (RHS) {
    if (a.src && act) b = b + a.src[i];
    a.rhs[i] = b;
  }
  if (OPR) a.val[dslot] = diag;
}

// boundary conditions of the thermal operator: COND_DIRICHLET (505), COND_HEAT_FLUX (507) (GE :862-944, 1149-1224)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_thermal_bc(ThermalBcArgs a) {
  constexpr bool RHS = (MODE == 0 || MODE == 2), OPR = (MODE == 1 || MODE == 2);
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= a.ncond_cells) return;
  const ThermalArgs &t = a.t;
  const size_t cell = a.cc_cell[j];
  if (t.active && t.active[cell] != 1) return;
  double b = RHS ? t.rhs[cell] : 0.0, d = 0.0;
  for (int e = a.cc_ptr[j]; e < a.cc_ptr[j + 1]; ++e) {
    const int ctype = a.type[e];
    if (ctype == 505) {
      if (!a.bc_active[e]) continue;
      // the BC auxvar: land unit and soil constants of its id_dn cell, no water, boundary temperature -- and is_soil_shallow
      // FALSE, which MPPThermalSetSoils never sets on boundary auxvars (MultiPhysicsProbThermal.F90:186-204; auxvar :65)
      double kbc, cbc;
      thermal_auxvar(t.itype[cell], false, a.value[e], 0.0, 0.0, 0.0, true, t.por[cell], t.tkmg[cell],
                     t.tkdry[cell], t.csol[cell], t.dz[cell], kbc, cbc);
      const double du = a.du[e], dd = a.dd[e], dist = du + dd, kc = t.therm_cond[cell];
      const double kav = kbc * kc * dist / (kbc * dd + kc * du);
      if (RHS) b = b + kav / dist * a.value[e] * (a.reference_defects ? a.area_dirichlet_rhs[e] : a.area[e]) * 1.0;
      if (OPR) d += a.frac[e] * (1.0 - t.cnfac) * kav / dist * a.area[e] * 1.0;
    } else if (ctype == 507) {
      // cur_cond%value = H - dH/dT * T of the cell (ThermKSPTempSoilGetFromSOEAuxVarsBC, GE :339-354; T = soln_prev)
      if (RHS) b = b + 1.0 * (a.value[e] - a.dhsdT[e] * t.T[cell]) * a.frac[e] * a.area[e];
      if (OPR) d += a.reference_defects ? -a.frac[e] * pow(a.dhsdT[e], a.area[e]) * 1.0   // `dhsdT**area` as written (GE :1215)
                                        : -a.frac[e] * a.dhsdT[e] * a.area[e] * 1.0;
    }
  }
  if (RHS) t.rhs[cell] = b;
  if (OPR) t.val[t.diag_slot[cell]] += d;
}

// ------------------------------------------------------------------------------------------------
// vector kernels (grid-stride, 8 B per lane coalesced)
// ------------------------------------------------------------------------------------------------
inline int vec_grid(int n) {  // one 16-byte pair per lane
  long long b = ((long long)n / 2 + kBlock - 1) / kBlock;
  return (int)(b < 1 ? 1 : b);
}

// grid-stride loop over 16-byte pairs (arrays come from hipMalloc: 256-byte aligned), scalar tail by one thread
template <class F2, class F1>
__device__ __forceinline__ void vec2_loop(int n, F2 f2, F1 f1) {
  const size_t n2 = (size_t)n >> 1;
  // the launch covers n/2 pairs exactly (vec_grid) or kRedPairs-per-block chunks (red_grid); the loop handles both
  const size_t per_block = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t span = (per_block + kBlock - 1) / kBlock * kBlock;
  const size_t lo = (size_t)blockIdx.x * span, hi = lo + span < n2 ? lo + span : n2;
  for (size_t q = lo + threadIdx.x; q < hi; q += kBlock) f2(q);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) f1((size_t)n - 1);
}
// read-only reductions: grid-stride over 16-byte pairs with a 4096-block grid (reads tolerate the stride:
// 6.5 TB/s measured, tools/membench.hip read_gs) so that only 4096 partial sums per vector are produced
template <class F2, class F1>
__device__ __forceinline__ void red2_loop(int n, F2 f2, F1 f1) {
  const size_t n2 = (size_t)n >> 1;
  for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < n2; q += (size_t)gridDim.x * kBlock) f2(q);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) f1((size_t)n - 1);
}
#define D2(p) reinterpret_cast<double2 *>(p)
#define CD2(p) reinterpret_cast<const double2 *>(p)

__global__ __launch_bounds__(kBlock) void k_copy(int n, const double *__restrict__ x, double *__restrict__ y) {
  vec2_loop(n, [&](size_t q) { D2(y)[q] = CD2(x)[q]; }, [&](size_t i) { y[i] = x[i]; });
}
__global__ __launch_bounds__(kBlock) void k_set(int n, double v, double *__restrict__ y) {
  vec2_loop(n, [&](size_t q) { D2(y)[q] = make_double2(v, v); }, [&](size_t i) { y[i] = v; });
}
// ---- ILU(0) solves of a structured 7-point grid by box wavefronts (BoxGrid in kernels.hpp) ----
// Factor values and the diagonal's reciprocals into the box layout: slot t holds internal cell cellmap[t] (-1: padding, all zero).
// An entry is told apart by the distance of its column from the row in the caller's numbering.
__global__ __launch_bounds__(kBlock) void k_box_fill(long long nslots, BoxGrid g, const int *__restrict__ cellmap, const int *__restrict__ int2ext,
                                                     const int *__restrict__ rowptr, const int *__restrict__ col, int nown, const double *__restrict__ lu,
                                                     const double *__restrict__ idiag, double *Ld, double *Ls, double *Lw, double *Ue, double *Un,
                                                     double *Uu, double *Di) {
  const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (t >= nslots) return;
  double ld = 0.0, ls = 0.0, lw = 0.0, ue = 0.0, un = 0.0, uu = 0.0, di = 0.0;
  const int r = cellmap[t];
  if (r >= 0) {
    const int e = int2ext[r], s3 = g.n1 * g.n2;
    di = idiag[r];
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
      if (col[q] >= nown) continue;   // a ghost column: multiplies a zero in the solves
      const int d = int2ext[col[q]] - e;
      const double v = lu[q];
      if (d == -1) lw = v;
      else if (d == 1) ue = v;
      else if (g.n2 > 1 && d == -g.n1) ls = v;
      else if (g.n2 > 1 && d == g.n1) un = v;
      else if (g.n3 > 1 && d == -s3) ld = v;
      else if (g.n3 > 1 && d == s3) uu = v;
    }
  }
  Ld[t] = ld; Ls[t] = ls; Lw[t] = lw; Ue[t] = ue; Un[t] = un; Uu[t] = uu; Di[t] = di;
}
// The matrix itself into the box layout, for the numeric factorisation by boxes (k_box_factor): a row's lower entries, diagonal and
// upper entries, and -- transposed -- the upper entries of its three lower neighbours that point back at it (T: a(c-1, c) etc.).
// Padding slots: zeros and a unit diagonal (the factorisation divides by it).
__global__ __launch_bounds__(kBlock) void k_box_load(long long nslots, BoxGrid g, const int *__restrict__ cellmap, const int *__restrict__ int2ext,
                                                     const int *__restrict__ rowptr, const int *__restrict__ col, int nown, const double *__restrict__ val,
                                                     double *Ld, double *Ls, double *Lw, double *Ue, double *Un, double *Uu, double *Di,
                                                     double *Td, double *Ts, double *Tw) {
  const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (t >= nslots) return;
  double ld = 0.0, ls = 0.0, lw = 0.0, ue = 0.0, un = 0.0, uu = 0.0, di = 1.0, td = 0.0, ts = 0.0, tw = 0.0;
  const int r = cellmap[t];
  if (r >= 0) {
    const int e = int2ext[r], s3 = g.n1 * g.n2;
    auto back = [&](int rn) {   // a(rn, r)
      for (int q = rowptr[rn]; q < rowptr[rn + 1]; ++q) if (col[q] == r) return val[q];
      return 0.0;
    };
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
      const int c = col[q];
      if (c >= nown) continue;        // a ghost column
      const int d = int2ext[c] - e;
      const double v = val[q];
      if (d == 0) di = v;
      else if (d == -1) { lw = v; tw = back(c); }
      else if (d == 1) ue = v;
      else if (g.n2 > 1 && d == -g.n1) { ls = v; ts = back(c); }
      else if (g.n2 > 1 && d == g.n1) un = v;
      else if (g.n3 > 1 && d == -s3) { ld = v; td = back(c); }
      else if (g.n3 > 1 && d == s3) uu = v;
    }
  }
  Ld[t] = ld; Ls[t] = ls; Lw[t] = lw; Ue[t] = ue; Un[t] = un; Uu[t] = uu; Di[t] = di; Td[t] = td; Ts[t] = ts; Tw[t] = tw;
}
// The right-hand side into the box layout (and the solution out of it), one workgroup per box: a box's cells are read (written)
// line by line along the fastest direction -- consecutive cells of the caller's numbering: coalesced accesses of ext2int and, in the
// layouts the library uses, of the vector -- through an LDS image of the box that is written out (read in) row by row.  Reading r
// through the slot order instead (k_gather_or_zero) moves two cells per 64-byte sector (PMC: 55 bytes per cell at 256^3).
// The image is written line by line: a wave's 64 cells of a grid line land 128 doubles apart (two step rows), i.e. in ONE pair of
// banks.  Swizzle: the row-pair number (bits 7..10 of the slot) is XORed into bits 1..4, a permutation inside every aligned group of
// 32 doubles: the line-order accesses spread over all banks, the slot-order copies stay conflict-free, the image keeps its size.
// (Padding the image by two doubles per 128 instead cost the fourth workgroup per CU: 1.03 against 0.82 ms, round 4, not kept.  The
// swizzle by itself changed nothing measurable -- 0.84 ms -- what the passes lacked was loads in flight: see kBoxIoUnroll.)
__device__ __forceinline__ int box_img(int t) { return t ^ (((t >> 7) & 15) << 1); }
// (Eight cells per thread and pass: their index loads go out together, then their value loads -- one dependent pair at a time left
// the 16 waves a CU holds of this kernel with too little in flight.  Padding slots are recognised when the image is written out, so
// the image is not cleared first: one barrier less.)
constexpr int kBoxIoUnroll = 8;
__device__ __forceinline__ bool box_slot_is_cell(const BoxGrid &g, int I, int J, int K, int t) {   // slot t of box (I, J, K) holds a grid cell
  const int row = ((t >> 7) << 1) | (t & 1), line = (t & 127) >> 1, jp = line & (g.B2 - 1), kp = line >> g.lg2, ip = row - jp - kp;
  return ip >= 0 && ip < g.B1 && I * g.B1 + ip < g.n1 && J * g.B2 + jp < g.n2 && K * g.B3 + kp < g.n3;
}
__global__ __launch_bounds__(kBlock) void k_box_gather_in(BoxGrid g, const int *__restrict__ ext2int, const double *__restrict__ r,
                                                          double *__restrict__ v) {
  extern __shared__ double img[];   // S * 64 doubles, swizzled (box_img)
  const int b = blockIdx.x, nslot = g.S * 64, ncell = 64 * g.B1;
  const int I = b % g.NB1, J = (b / g.NB1) % g.NB2, K = b / (g.NB1 * g.NB2);
  for (int q0 = threadIdx.x; q0 < ncell; q0 += kBlock * kBoxIoUnroll) {
    int src[kBoxIoUnroll], dst[kBoxIoUnroll];
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) {
      const int q = q0 + u * kBlock;
      const int line = q / g.B1, ip = q - line * g.B1, jp = line & (g.B2 - 1), kp = line >> g.lg2;
      const int i = I * g.B1 + ip, j = J * g.B2 + jp, k = K * g.B3 + kp, row = ip + jp + kp;
      const bool ok = q < ncell && i < g.n1 && j < g.n2 && k < g.n3;
      src[u] = ok ? ext2int[i + g.n1 * (j + g.n2 * k)] : -1;
      dst[u] = box_img((row & ~1) * 64 + 2 * line + (row & 1));
    }
    double val[kBoxIoUnroll];
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) val[u] = src[u] >= 0 ? r[src[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) if (src[u] >= 0) img[dst[u]] = val[u];
  }
  __syncthreads();
  double *out = v + (size_t)b * nslot;
  for (int t = threadIdx.x; t < nslot; t += kBlock) out[t] = box_slot_is_cell(g, I, J, K, t) ? img[box_img(t)] : 0.0;
}
// ... and the way back: a box's solution rows into the image as they lie, then out line by line (coalesced writes of z and reads of
// ext2int; k_gather reads the solution through the per-row slot map, 16 useful bytes of every 32-byte sector)
__global__ __launch_bounds__(kBlock) void k_box_gather_out(BoxGrid g, const int *__restrict__ ext2int, const double *__restrict__ v,
                                                           double *__restrict__ z) {
  extern __shared__ double img[];
  const int b = blockIdx.x, nslot = g.S * 64, ncell = 64 * g.B1;
  const int I = b % g.NB1, J = (b / g.NB1) % g.NB2, K = b / (g.NB1 * g.NB2);
  const double *in = v + (size_t)b * nslot;
  for (int t0 = threadIdx.x; t0 < nslot; t0 += kBlock * kBoxIoUnroll) {
    double val[kBoxIoUnroll];
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) { const int t = t0 + u * kBlock; val[u] = t < nslot ? __builtin_nontemporal_load(in + t) : 0.0; }
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) { const int t = t0 + u * kBlock; if (t < nslot) img[box_img(t)] = val[u]; }
  }
  __syncthreads();
  for (int q0 = threadIdx.x; q0 < ncell; q0 += kBlock * kBoxIoUnroll) {
    int dst[kBoxIoUnroll], slot[kBoxIoUnroll];
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) {
      const int q = q0 + u * kBlock;
      const int line = q / g.B1, ip = q - line * g.B1, jp = line & (g.B2 - 1), kp = line >> g.lg2;
      const int i = I * g.B1 + ip, j = J * g.B2 + jp, k = K * g.B3 + kp, row = ip + jp + kp;
      const bool ok = q < ncell && i < g.n1 && j < g.n2 && k < g.n3;
      dst[u] = ok ? ext2int[i + g.n1 * (j + g.n2 * k)] : -1;
      slot[u] = box_img((row & ~1) * 64 + 2 * line + (row & 1));
    }
#pragma unroll
    for (int u = 0; u < kBoxIoUnroll; ++u) if (dst[u] >= 0) z[dst[u]] = img[slot[u]];
  }
}
__global__ __launch_bounds__(kBlock) void k_gather_or_zero(long long n, const int *idx, const double *in, double *out) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) { const int j = idx[i]; out[i] = j >= 0 ? in[j] : 0.0; }
}

// One wave solves one box; lane (j', k') walks its line of the box in i', one step row per step: at step h it holds cell
// i' = h - j' - k'.  The west neighbour's value is the lane's own previous result, south and down come from lanes -1 and -B2 one
// step earlier (a DPP shift and a bpermute); across a box face they come from the finished neighbour box, whose face the wave
// copies into LDS before it starts.  Padding slots carry zero coefficients and a zero right-hand side, so every lane computes
// every step (no branches).  Operands arrive as 16-byte loads (two step rows per lane and instruction, see box_slot) through a
// ring of DP row pairs: five memory instructions per pair, DP = 12 keeps 60 of the 63 the counter can hold in flight.
// The step is issue-bound (SQ counters, profiles/r02_pmc_sq_box_ilu_v2.md), so it is kept to ~15 vector instructions: row
// addresses are scalar, the face values are read through pointers that advance by a constant (margins instead of clamps).
// Per row the entries are eliminated in the order of the level-scheduled kernels (ascending column): the same bits.
__device__ __forceinline__ double lane_shift(double v, int bperm_addr) {   // v of the lane bperm_addr / 4
  const int lo = __builtin_amdgcn_ds_bpermute(bperm_addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(bperm_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL>   // 0x138: from lane - 1, 0x130: from lane + 1 (whole-wave DPP shifts; the end lane keeps 0)
__device__ __forceinline__ double lane_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false),
            hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// LDS of a box solve: 8 zeros (what a lane reads for a face nobody is behind), then the two face arrays.  Element (i', c) of a
// face with C lines is placed so that the lane of line c reads it at the step it works on the cell next to it, through a pointer
// that advances by C per step; zero rows of margin either side for the steps at which the lane is outside the box.
__host__ __device__ inline int box_face_doubles(const BoxGrid &g) {
  return 8 + (g.NB2 > 1 ? (g.S + g.B3 + 1) * g.B3 : 0) + (g.NB3 > 1 ? (g.S + g.B2 + 1) * g.B2 : 0);
}
// uniform row pointer + the lane's byte offset (32 bits): the address form that keeps the row arithmetic scalar
template <class T>
__device__ __forceinline__ T *at_lane(T *row, unsigned lane_bytes) {
  using B = std::conditional_t<std::is_const<T>::value, const char, char>;
  return reinterpret_cast<T *>(reinterpret_cast<B *>(row) + lane_bytes);   // (pointer arithmetic: the address space stays known)
}
constexpr int kFaceBatch = 8;   // face elements per lane requested up front (B1 * 8 / 64 of a 3-D box: all of them up to B1 = 64)
// One launch per triangle instead of one per box level (round 4): the boxes are handed out by TICKET in level order to a fixed set of
// resident waves (BoxQueue); a box waits for the flags of its three predecessor boxes (all of a lower level, hence lower tickets,
// hence taken by waves that are already running: liveness does not depend on the order workgroups are dispatched in), solves, and
// raises its own flag.  What another wave wrote during the launch -- the faces -- is read with agent-scope loads, and the solution is
// stored with agent-scope stores (served at the device's coherence point: the L2 of one XCD is not coherent with the others' inside a
// launch), acknowledged (vmcnt(0)) before the flag goes up.  A box's own operands (factor entries, right-hand side: written by
// earlier launches) are requested BEFORE it waits.  A wait has a spin budget; exhausting it sets sc[S_ERR] and the host redoes the
// application with one launch per level (Solver::op_pc_apply).
struct BoxQueue {
  const int *order = nullptr;      // boxes in the order they are handed out (ascending level; the upper solve: descending)
  int nboxes = 0;
  unsigned *flags = nullptr;       // flags[b] == epoch: box b is solved in this application
  unsigned epoch = 0;
  unsigned *ticket = nullptr;      // monotonic ticket counter (never reset) ...
  unsigned ticket_base = 0;        // ... and what earlier launches took from it
  unsigned max_spins = 0;
  double *sc = nullptr;
};
template <bool COH>
__device__ __forceinline__ double ld_shared_result(const double *p) {   // a value another wave of this launch may have written
  return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH>
__device__ __forceinline__ void st_result_pair(dpair *dst, unsigned lane_bytes, dpair v) {
  if (COH) {
    // one 16-byte store per lane with the agent-scope cache policy (sc1: written through to the device's coherence point).  HIP has no
    // 16-byte atomic store, and two 8-byte ones leave every cache line half-written per instruction: byte-masked write-throughs, measured
    // 4x slower than the per-level launches (gpurun_out/r04_run3: 28.8 against 6.65 ms per application at 512^3)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(at_lane(dst, lane_bytes)), "v"(v) : "memory");
  } else *at_lane(dst, lane_bytes) = v;
}
// lanes 0, 1, 2 wait for the flags of boxes p0, p1, p2 (-1: nobody)
__device__ __forceinline__ void box_wait(const BoxQueue &q, int lane, int p0, int p1, int p2) {
  const int pred = lane == 0 ? p0 : (lane == 1 ? p1 : (lane == 2 ? p2 : -1));
  if (pred >= 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(q.flags + pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != q.epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > q.max_spins) { q.sc[S_ERR] = 1.0; break; }
    }
  }
  __builtin_amdgcn_wave_barrier();
}
template <int DP, bool COH>
__device__ __forceinline__ void box_lower_body(const BoxGrid &g, const int b, const double *__restrict__ Ld, const double *__restrict__ Ls,
                                               const double *__restrict__ Lw, double *y, double *scratch, double *face, const BoxQueue &q) {
  const int lane = threadIdx.x, S = g.S, NP = S >> 1;
  const int jp = lane & (g.B2 - 1), kp = lane >> g.lg2;
  const int I = b % g.NB1, J = (b / g.NB1) % g.NB2, K = b / (g.NB1 * g.NB2);
  const bool useS = g.NB2 > 1, useD = g.NB3 > 1;
  double *fS = face + 8, *fD = fS + (useS ? (S + g.B3 + 1) * g.B3 : 0);
  for (int e = lane; e < box_face_doubles(g); e += 64) face[e] = 0.0;
  // element (i', k') of the south face is cell (i', B2-1, k') of box J-1, (i', j') of the lower face cell (i', j', B3-1) of box K-1;
  // the first kFaceBatch * 64 elements of either face are requested here and parked behind the ring's first fill (one round trip
  // to memory for both), longer faces follow in a loop
  const int nfS = J > 0 ? g.B1 * g.B3 : 0, nfD = K > 0 ? g.B1 * g.B2 : 0;
  auto srcS = [&](int e) { const int i = e / g.B3, k = e - i * g.B3; return y + box_slot(g, b - g.NB1, i + g.B2 - 1 + k, g.B2 - 1 + g.B2 * k); };
  auto srcD = [&](int e) { const int i = e >> g.lg2, j = e & (g.B2 - 1); return y + box_slot(g, b - g.NB1 * g.NB2, i + j + g.B3 - 1, j + g.B2 * (g.B3 - 1)); };
  auto dstS = [&](int e) { const int i = e / g.B3, k = e - i * g.B3; return fS + (i + g.B3 - 1) * g.B3 + k; };
  auto dstD = [&](int e) { const int i = e >> g.lg2, j = e & (g.B2 - 1); return fD + (i + g.B2 - 1) * g.B2 + j; };
  double vS[kFaceBatch], vD[kFaceBatch];
  const int hW = jp + kp;                                       // the step at which the lane is on the west face
  double yW = 0.0;
  auto request_faces = [&]() {
#pragma unroll
    for (int qq = 0; qq < kFaceBatch; ++qq) {
      const int e = qq * 64 + lane;
      vS[qq] = ld_shared_result<COH>(e < nfS ? srcS(e) : y + (size_t)b * S * 64);      // (nobody behind, or past the face: any valid address)
      vD[qq] = ld_shared_result<COH>(e < nfD ? srcD(e) : y + (size_t)b * S * 64);
    }
    yW = ld_shared_result<COH>(I > 0 ? y + box_slot(g, b - 1, g.B1 - 1 + hW, lane) : y + (size_t)b * S * 64);
    if (I == 0) yW = 0.0;
  };
  if (!COH) request_faces();                                    // one launch per level: behind nobody, first in the queue
  const size_t row0 = (size_t)b * S * 64;                       // the box's first slot (uniform); row pair p is 64 dpairs further on
  const dpair *Ld2 = reinterpret_cast<const dpair *>(Ld + row0), *Ls2 = reinterpret_cast<const dpair *>(Ls + row0),
              *Lw2 = reinterpret_cast<const dpair *>(Lw + row0);
  dpair *y2 = reinterpret_cast<dpair *>(y + row0), *sink = reinterpret_cast<dpair *>(scratch);
  const unsigned lane16 = threadIdx.x * 16u;
  dpair ld[DP], ls[DP], lw[DP], rr[DP];
  auto fetch = [&](int u, int p) {
    const size_t o = (size_t)min(p, NP - 1) * 64;               // uniform
    ld[u] = ldnt(at_lane(Ld2 + o, lane16)); ls[u] = ldnt(at_lane(Ls2 + o, lane16)); lw[u] = ldnt(at_lane(Lw2 + o, lane16));
    rr[u] = *at_lane(y2 + o, lane16);
  };
#pragma unroll
  for (int u = 0; u < DP; ++u) { fetch(u, u); __builtin_amdgcn_sched_barrier(0); }
  if (COH) {   // the box's own operands are on their way: now wait for the three boxes behind it, then ask for their faces
    box_wait(q, lane, I > 0 ? b - 1 : -1, J > 0 ? b - g.NB1 : -1, K > 0 ? b - g.NB1 * g.NB2 : -1);
    request_faces();
  }
  __syncthreads();                                              // (one wave: the zeros are in place before the face values)
#pragma unroll
  for (int qq = 0; qq < kFaceBatch; ++qq) {
    const int e = qq * 64 + lane;
    if (e < nfS) *dstS(e) = vS[qq];
    if (e < nfD) *dstD(e) = vD[qq];
  }
  for (int e = kFaceBatch * 64 + lane; e < max(nfS, nfD); e += 64) {
    if (e < nfS) *dstS(e) = ld_shared_result<COH>(srcS(e));
    if (e < nfD) *dstD(e) = ld_shared_result<COH>(srcD(e));
  }
  __syncthreads();
  const double *pS = useS ? fS + (g.B3 - 1 - kp) * g.B3 + kp : face, *pD = useD ? fD + (g.B2 - 1 - jp) * g.B2 + jp : face;   // (face[0] = 0)
  const int incS = useS ? g.B3 : 0, incD = useD ? g.B2 : 0;
  const bool inS = jp != 0, inD = kp != 0;
  const int addrB = ((lane - g.B2) & 63) << 2;
  double yprev = 0.0, nS = *pS, nD = *pD;
  auto step = [&](int h, double r, double cd, double cs, double cw) {
    const double a1 = lane_dpp<0x138>(yprev), aB = lane_shift(yprev, addrB);
    const double yd = inD ? aB : nD, ys = inS ? a1 : nS, yw = h == hW ? yW : yprev;
    pS += h + 1 < S ? incS : 0; pD += h + 1 < S ? incD : 0;     // (not past the last step: the loop's tail pairs compute nothing real)
    nS = *pS; nD = *pD;                                         // the next step's face values, off the dependency chain
    double s = r;
    s -= cd * yd;
    s -= cs * ys;
    s -= cw * yw;
    yprev = s;
    return s;
  };
  for (int p0 = 0; p0 < NP; p0 += DP) {
#pragma unroll
    for (int u = 0; u < DP; ++u) {
      const int p = p0 + u;
      dpair out;
      out.x = step(2 * p, rr[u].x, ld[u].x, ls[u].x, lw[u].x);
      out.y = step(2 * p + 1, rr[u].y, ld[u].y, ls[u].y, lw[u].y);
      dpair *dst = p < NP ? y2 + (size_t)p * 64 : sink;         // past the box: a scratch row (keeps the loop body straight)
      st_result_pair<COH>(dst, lane16, out);
      fetch(u, p + DP);
      __builtin_amdgcn_sched_barrier(0);   // keep the issue order pair by pair: the wait of a pair then counts the younger loads
    }
  }
}
template <int DP>
__global__ __launch_bounds__(64) void k_box_lower(BoxGrid g, const int *__restrict__ boxes, const double *__restrict__ Ld,
                                                  const double *__restrict__ Ls, const double *__restrict__ Lw, double *y,
                                                  double *scratch) {
  extern __shared__ double face[];
  box_lower_body<DP, false>(g, boxes[blockIdx.x], Ld, Ls, Lw, y, scratch, face, BoxQueue());
}
// the ticket of the wave's next box (uniform), or -1 when the queue is empty
__device__ __forceinline__ int box_next(const BoxQueue &q) {
  unsigned t = 0;
  if (threadIdx.x == 0) t = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.ticket_base;
  t = __builtin_amdgcn_readfirstlane(t);
  return t < (unsigned)q.nboxes ? (int)t : -1;
}
__device__ __forceinline__ void box_done(const BoxQueue &q, int b) {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every store of this wave has been acknowledged at the coherence point
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(q.flags + b, q.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();                      // (the faces in LDS are free again)
}
template <int DP>
__global__ __launch_bounds__(64) void k_box_lower_queue(BoxGrid g, BoxQueue q, const double *__restrict__ Ld, const double *__restrict__ Ls,
                                                        const double *__restrict__ Lw, double *y, double *scratch) {
  extern __shared__ double face[];
  for (int t = box_next(q); t >= 0; t = box_next(q)) {
    const int b = q.order[t];
    box_lower_body<DP, true>(g, b, Ld, Ls, Lw, y, scratch, face, q);
    box_done(q, b);
  }
}
template <int DP, bool COH>
__device__ __forceinline__ void box_upper_body(const BoxGrid &g, const int b, const double *__restrict__ Ue, const double *__restrict__ Un,
                                               const double *__restrict__ Uu, const double *__restrict__ Di, double *x, double *scratch,
                                               double *face, const BoxQueue &q) {
  const int lane = threadIdx.x, S = g.S, NP = S >> 1;
  const int jp = lane & (g.B2 - 1), kp = lane >> g.lg2;
  const int I = b % g.NB1, J = (b / g.NB1) % g.NB2, K = b / (g.NB1 * g.NB2);
  const bool useN = g.NB2 > 1, useU = g.NB3 > 1;
  const int M = g.B2 + g.B3 - 2;
  double *fN = face + 8, *fU = fN + (useN ? (S + g.B3 + 1) * g.B3 : 0);
  for (int e = lane; e < box_face_doubles(g); e += 64) face[e] = 0.0;
  // (i', k') of the north face: cell (i', 0, k') of box J+1, wanted by lane (B2-1, k') at step h = i' + B2-1 + k';
  // (i', j') of the upper face: cell (i', j', 0) of box K+1, wanted by lane (j', B3-1) at step h = i' + j' + B3-1.
  // Stored behind M = B2 + B3 - 2 rows of (zero) margin: the lanes run up to M steps outside the box on that side.
  const int nfN = J < g.NB2 - 1 ? g.B1 * g.B3 : 0, nfU = K < g.NB3 - 1 ? g.B1 * g.B2 : 0;
  auto srcN = [&](int e) { const int i = e / g.B3, k = e - i * g.B3; return x + box_slot(g, b + g.NB1, i + k, g.B2 * k); };
  auto srcU = [&](int e) { const int i = e >> g.lg2, j = e & (g.B2 - 1); return x + box_slot(g, b + g.NB1 * g.NB2, i + j, j); };
  auto dstN = [&](int e) { const int i = e / g.B3, k = e - i * g.B3; return fN + (i + M) * g.B3 + k; };
  auto dstU = [&](int e) { const int i = e >> g.lg2, j = e & (g.B2 - 1); return fU + (i + M) * g.B2 + j; };
  double vN[kFaceBatch], vU[kFaceBatch];
  const int hE = g.B1 - 1 + jp + kp;                            // the step at which the lane is on the east face
  double xE = 0.0;
  auto request_faces = [&]() {
#pragma unroll
    for (int qq = 0; qq < kFaceBatch; ++qq) {
      const int e = qq * 64 + lane;
      vN[qq] = ld_shared_result<COH>(e < nfN ? srcN(e) : x + (size_t)b * S * 64);
      vU[qq] = ld_shared_result<COH>(e < nfU ? srcU(e) : x + (size_t)b * S * 64);
    }
    xE = ld_shared_result<COH>(I < g.NB1 - 1 ? x + box_slot(g, b + 1, jp + kp, lane) : x + (size_t)b * S * 64);
    if (I == g.NB1 - 1) xE = 0.0;
  };
  if (!COH) request_faces();
  const size_t row0 = (size_t)b * S * 64;
  const dpair *Ue2 = reinterpret_cast<const dpair *>(Ue + row0), *Un2 = reinterpret_cast<const dpair *>(Un + row0),
              *Uu2 = reinterpret_cast<const dpair *>(Uu + row0), *Di2 = reinterpret_cast<const dpair *>(Di + row0);
  dpair *x2 = reinterpret_cast<dpair *>(x + row0), *sink = reinterpret_cast<dpair *>(scratch);
  const unsigned lane16 = threadIdx.x * 16u;
  dpair ue[DP], un[DP], uu[DP], di[DP], rr[DP];
  auto fetch = [&](int u, int p) {
    const size_t o = (size_t)max(p, 0) * 64;
    ue[u] = ldnt(at_lane(Ue2 + o, lane16)); un[u] = ldnt(at_lane(Un2 + o, lane16)); uu[u] = ldnt(at_lane(Uu2 + o, lane16));
    di[u] = ldnt(at_lane(Di2 + o, lane16));
    rr[u] = *at_lane(x2 + o, lane16);
  };
#pragma unroll
  for (int u = 0; u < DP; ++u) { fetch(u, NP - 1 - u); __builtin_amdgcn_sched_barrier(0); }
  if (COH) {   // the boxes ahead: east, north, up
    box_wait(q, lane, I < g.NB1 - 1 ? b + 1 : -1, J < g.NB2 - 1 ? b + g.NB1 : -1, K < g.NB3 - 1 ? b + g.NB1 * g.NB2 : -1);
    request_faces();
  }
  __syncthreads();
#pragma unroll
  for (int qq = 0; qq < kFaceBatch; ++qq) {
    const int e = qq * 64 + lane;
    if (e < nfN) *dstN(e) = vN[qq];
    if (e < nfU) *dstU(e) = vU[qq];
  }
  for (int e = kFaceBatch * 64 + lane; e < max(nfN, nfU); e += 64) {
    if (e < nfN) *dstN(e) = ld_shared_result<COH>(srcN(e));
    if (e < nfU) *dstU(e) = ld_shared_result<COH>(srcU(e));
  }
  __syncthreads();
  // first step h = S - 1: i' = S - 1 - (B2 - 1) - kp on the north face, S - 1 - jp - (B3 - 1) on the upper one
  const double *pN = useN ? fN + (S - 1 - (g.B2 - 1) - kp + M) * g.B3 + kp : face,
               *pU = useU ? fU + (S - 1 - jp - (g.B3 - 1) + M) * g.B2 + jp : face;
  const int incN = useN ? g.B3 : 0, incU = useU ? g.B2 : 0;
  const bool inN = jp != g.B2 - 1, inU = kp != g.B3 - 1;
  const int addrB = ((lane + g.B2) & 63) << 2;
  double xprev = 0.0, nN = *pN, nU = *pU;
  auto step = [&](int h, double r, double ce, double cn, double cu, double d) {
    const double a1 = lane_dpp<0x130>(xprev), aB = lane_shift(xprev, addrB);
    const double xe = h == hE ? xE : xprev, xn = inN ? a1 : nN, xu = inU ? aB : nU;
    pN -= h > 0 ? incN : 0; pU -= h > 0 ? incU : 0;
    nN = *pN; nU = *pU;
    double s = r;
    s -= ce * xe;
    s -= cn * xn;
    s -= cu * xu;
    s *= d;
    xprev = s;
    return s;
  };
  for (int p0 = NP - 1; p0 >= 0; p0 -= DP) {
#pragma unroll
    for (int u = 0; u < DP; ++u) {
      const int p = p0 - u;
      dpair out;
      out.y = step(2 * p + 1, rr[u].y, ue[u].y, un[u].y, uu[u].y, di[u].y);
      out.x = step(2 * p, rr[u].x, ue[u].x, un[u].x, uu[u].x, di[u].x);
      dpair *dst = p >= 0 ? x2 + (size_t)p * 64 : sink;
      st_result_pair<COH>(dst, lane16, out);
      fetch(u, p - DP);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int DP>
__global__ __launch_bounds__(64) void k_box_upper(BoxGrid g, const int *__restrict__ boxes, const double *__restrict__ Ue,
                                                  const double *__restrict__ Un, const double *__restrict__ Uu,
                                                  const double *__restrict__ Di, double *x, double *scratch) {
  extern __shared__ double face[];
  box_upper_body<DP, false>(g, boxes[blockIdx.x], Ue, Un, Uu, Di, x, scratch, face, BoxQueue());
}
template <int DP>
__global__ __launch_bounds__(64) void k_box_upper_queue(BoxGrid g, BoxQueue q, const double *__restrict__ Ue, const double *__restrict__ Un,
                                                        const double *__restrict__ Uu, const double *__restrict__ Di, double *x, double *scratch) {
  extern __shared__ double face[];
  for (int t = box_next(q); t >= 0; t = box_next(q)) {
    const int b = q.order[t];
    box_upper_body<DP, true>(g, b, Ue, Un, Uu, Di, x, scratch, face, q);
    box_done(q, b);
  }
}

// ---- the same two solves as STREAMS (round 4): one launch per triangle, fine-grained hand-over of the box faces ----
// What the per-level launches cost at 512^3 (profiles/r04/box_trace_512_len64.json): 134 box levels per triangle, 13 us each even
// when the level holds ONE box (dispatch + the 78 dependent steps of a box), 27 us for the largest levels: 2.5 ms per triangle for
// 6.5 GB, a third of what the memory system does.  Here a wave owns a PENCIL -- the NB1 boxes (I = 0 .. NB1-1) of one (J, K) -- and
// streams through it; the pencil behind it in J and the one behind it in K are at most a few steps ahead of it:
//   * a lane on the south (down) side of its box at step h needs the value the producer pencil's lane on the opposite side computed
//     at ITS step h + B2 - 1 (h + B3 - 1) of the same box: all B3 (B2) values of one consumer step come from one producer step;
//   * so every step a producer stores one line of 8 values per outgoing face -- agent-scope stores into a face buffer of its own,
//     line (h - 7) of box I -- and a consumer, every 8 steps, loads the next 8 lines of its two incoming faces with one coalesced
//     agent-scope load each (512 B), waits until none of the values it needs is the SENTINEL the buffer was initialised with (the
//     payload is the flag: measured 0.6-0.7 us from store to sight on an idle chip, tools/handoff_bench.hip), parks them in LDS
//     for its next 8 steps, and puts the sentinel back (the buffer is ready for the next application).
// Pencils are handed out by ticket in the order of J + K (the upper solve: descending), so a pencil's producers hold lower tickets
// and are running: liveness does not depend on the dispatch order.  Nothing else crosses between waves: the solution itself is
// stored with plain stores (the other triangle is another launch), the west face of a box is the wave's own previous box.
// Arithmetic and order per row are those of k_box_lower / k_box_upper: the same bits.  Needs B2 = B3 = 8 and S a multiple of 8.
constexpr int kStreamLead = 24, kStreamTail = 40;   // lines of margin in front of / behind the S lines of a box in a face buffer
                                                    // (steps outside 0 .. S-1 of the loops' rounded-up trip counts land there)
__host__ __device__ inline int stream_lines(const BoxGrid &g) { return kStreamLead + g.S + kStreamTail; }   // lines per box and face
__device__ __forceinline__ bool is_sentinel(double v) { return __double_as_longlong(v) == -1LL; }
struct StreamArgs {
  const int *order;        // pencils (J + NB2 * K) in ticket order
  int npencils;
  unsigned *ticket; unsigned ticket_base, max_spins;
  double *sc;
  double *faceJ, *faceK;   // this triangle's face buffers: [pencil][box I][line][8]
};
// GATHER: the right-hand side is read from `rvec` (the Krylov vector, internal row order) through the slot map `cellmap` (internal row
// of every slot, -1 for padding) instead of from y -- no k_box_gather_in pass in front of the solve.  A lane walks consecutive cells
// of one grid line, i.e. consecutive rows of a column tile: its 8-byte reads fill whole cache lines over 8 steps.  A padding slot
// reads row 0 instead of a zero: its value only ever meets zero coefficients (and stays finite).  The indices of a row pair travel
// with its factor entries (DP pairs ahead), the values DP / 2 pairs ahead.
typedef int ipair __attribute__((ext_vector_type(2)));
template <int DP, bool GATHER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_box_lower_stream(BoxGrid g, StreamArgs q, const double *__restrict__ Ld, const double *__restrict__ Ls,
                                                         const double *__restrict__ Lw, double *y, double *scratch,
                                                         const double *__restrict__ rvec, const int *__restrict__ cellmap) {
  static_assert(DP % 4 == 0, "a chunk of 8 steps is 4 row pairs");
  extern __shared__ double lds[];
  const int lane = threadIdx.x, S = g.S, NP = S >> 1, NCH = S >> 3;
  const int jp = lane & 7, kp = lane >> 3;
  const int LB = stream_lines(g);
  double *lineJ = lds, *lineK = lds + (size_t)(S + 2 * DP + 8) * 8;
  const size_t pen_stride = (size_t)g.NB1 * LB * 8;
  const bool inS = jp != 0, inD = kp != 0, mJ = jp == 7, mK = kp == 7;
  const int addrB = ((lane - 8) & 63) << 2;
  const int hW = jp + kp;
  const unsigned lane16 = threadIdx.x * 16u;
  dpair *sink = reinterpret_cast<dpair *>(scratch);
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.ticket_base;
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)q.npencils) return;
    const int pen = q.order[t];
    const int J = pen % g.NB2, K = pen / g.NB2;
    const bool hasJ = J > 0, hasK = K > 0, outJ = J < g.NB2 - 1, outK = K < g.NB3 - 1;
    const double *inJ = q.faceJ + (size_t)(hasJ ? pen - 1 : pen) * pen_stride, *inK = q.faceK + (size_t)(hasK ? pen - g.NB2 : pen) * pen_stride;
    double *myJ = q.faceJ + (size_t)pen * pen_stride, *myK = q.faceK + (size_t)pen * pen_stride;
    for (int I = 0; I < g.NB1; ++I) {
      const int b = I + g.NB1 * pen;
      const size_t row0 = (size_t)b * S * 64;
      const dpair *Ld2 = reinterpret_cast<const dpair *>(Ld + row0), *Ls2 = reinterpret_cast<const dpair *>(Ls + row0),
                  *Lw2 = reinterpret_cast<const dpair *>(Lw + row0);
      dpair *y2 = reinterpret_cast<dpair *>(y + row0);
      const ipair *cm2 = reinterpret_cast<const ipair *>(cellmap + row0);
      dpair ld[DP], ls[DP], lw[DP], rr[DP];
      ipair ix[GATHER ? DP : 1];
      auto fetch = [&](int u, int p) {
        const size_t o = (size_t)min(p, NP - 1) * 64;
        ld[u] = ldnt(at_lane(Ld2 + o, lane16)); ls[u] = ldnt(at_lane(Ls2 + o, lane16)); lw[u] = ldnt(at_lane(Lw2 + o, lane16));
        if (GATHER) ix[u] = ldnt(at_lane(cm2 + o, threadIdx.x * 8u));
        else rr[u] = *at_lane(y2 + o, lane16);
      };
      auto gather = [&](int u) {   // the two right-hand side values of the row pair whose indices sit in ix[u]
        if (GATHER) { rr[u].x = rvec[max(ix[u].x, 0)]; rr[u].y = rvec[max(ix[u].y, 0)]; }
      };
#pragma unroll
      for (int u = 0; u < DP; ++u) { fetch(u, u); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int u = 0; u < DP / 2; ++u) { gather(u); __builtin_amdgcn_sched_barrier(0); }
      double yW = *(I > 0 ? y + box_slot(g, b - 1, g.B1 - 1 + hW, lane) : y + row0);   // (the wave's own store of the box before)
      if (I == 0) yW = 0.0;
      // the incoming faces of this box: chunk c = lines 8c .. 8c+7, lane l holds element (line 8c + l/8, l%8)
      const double *bJ = inJ + ((size_t)I * LB + kStreamLead) * 8, *bK = inK + ((size_t)I * LB + kStreamLead) * 8;
      double cJ = 0.0, cK = 0.0;
      auto request = [&](int c) {
        if (hasJ) cJ = __hip_atomic_load(bJ + 64 * c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hasK) cK = __hip_atomic_load(bK + 64 * c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto accept = [&](int c) {   // chunk c is in cJ / cK or on its way: wait for its values, park them in LDS, put the sentinel back
        const bool need = 8 * c + (lane >> 3) < S - 7;   // (lines from S-7 on have no producer step: no real cell reads them)
        unsigned spins = 0;
        while (__any((need && hasJ && is_sentinel(cJ)) || (need && hasK && is_sentinel(cK)))) {
          __builtin_amdgcn_s_sleep(2);
          request(c);
          if (++spins > q.max_spins) { if (lane == 0) q.sc[S_ERR] = 1.0; break; }
        }
        lineJ[64 * c + lane] = (need && hasJ) ? cJ : 0.0;
        lineK[64 * c + lane] = (need && hasK) ? cK : 0.0;
        if (need && hasJ) const_cast<double *>(bJ)[64 * c + lane] = __longlong_as_double(-1LL);
        if (need && hasK) const_cast<double *>(bK)[64 * c + lane] = __longlong_as_double(-1LL);
        __builtin_amdgcn_wave_barrier();
      };
      request(0);
      accept(0);
      if (NCH > 1) request(1);
      const double *pS = lineJ + kp, *pD = lineK + jp;
      double *oJ = myJ + ((size_t)I * LB + kStreamLead - 7) * 8 + kp, *oK = myK + ((size_t)I * LB + kStreamLead - 7) * 8 + jp;   // line h - 7
      double yprev = 0.0;
      auto step = [&](int h, double r, double cd, double cs, double cw) {
        const double a1 = lane_dpp<0x138>(yprev), aB = lane_shift(yprev, addrB);
        const double nS = *pS, nD = *pD;
        pS += 8; pD += 8;
        const double yd = inD ? aB : nD, ys = inS ? a1 : nS, yw = h == hW ? yW : yprev;
        double s = r;
        s -= cd * yd;
        s -= cs * ys;
        s -= cw * yw;
        yprev = s;
        if (outJ && mJ) __hip_atomic_store(oJ, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (outK && mK) __hip_atomic_store(oK, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oJ += 8; oK += 8;
        return s;
      };
      for (int p0 = 0; p0 < NP; p0 += DP) {
#pragma unroll
        for (int u = 0; u < DP; ++u) {
          const int p = p0 + u;
          if (u % 4 == 0) {   // a chunk boundary: steps 2p .. 2p+7 read lines 8c .. 8c+7, c = p / 4
            const int c = p >> 2;
            if (p > 0 && c < NCH) { accept(c); if (c + 1 < NCH) request(c + 1); }
          }
          dpair out;
          out.x = step(2 * p, rr[u].x, ld[u].x, ls[u].x, lw[u].x);
          out.y = step(2 * p + 1, rr[u].y, ld[u].y, ls[u].y, lw[u].y);
          dpair *dst = p < NP ? y2 + (size_t)p * 64 : sink;
          *at_lane(dst, lane16) = out;
          fetch(u, p + DP);
          gather((u + DP / 2) % DP);   // (the indices of pair p + DP / 2 were requested DP / 2 pairs ago)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}
// SCATTER: the solution is also written to `zvec` (internal row order) through the slot map: no k_gather pass behind the solve
template <int DP, bool SCATTER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_box_upper_stream(BoxGrid g, StreamArgs q, const double *__restrict__ Ue, const double *__restrict__ Un,
                                                         const double *__restrict__ Uu, const double *__restrict__ Di, double *x, double *scratch,
                                                         double *__restrict__ zvec, const int *__restrict__ cellmap) {
  static_assert(DP % 4 == 0, "a chunk of 8 steps is 4 row pairs");
  extern __shared__ double lds[];
  const int lane = threadIdx.x, S = g.S, NP = S >> 1, NCH = S >> 3;
  const int jp = lane & 7, kp = lane >> 3;
  const int LB = stream_lines(g);
  const int MRG = 2 * DP + 8;                         // lines of LDS in front of line 0 (the loop's last round runs below step 0)
  double *lineJ = lds + (size_t)MRG * 8, *lineK = lineJ + (size_t)(S + MRG) * 8;
  const size_t pen_stride = (size_t)g.NB1 * LB * 8;
  const bool inN = jp != 7, inU = kp != 7, mJ = jp == 0, mK = kp == 0;
  const int addrB = ((lane + 8) & 63) << 2;
  const int hE = g.B1 - 1 + jp + kp;
  const unsigned lane16 = threadIdx.x * 16u;
  dpair *sink = reinterpret_cast<dpair *>(scratch);
  for (;;) {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.ticket_base;
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= (unsigned)q.npencils) return;
    const int pen = q.order[t];
    const int J = pen % g.NB2, K = pen / g.NB2;
    const bool hasJ = J < g.NB2 - 1, hasK = K < g.NB3 - 1, outJ = J > 0, outK = K > 0;
    const double *inJ = q.faceJ + (size_t)(hasJ ? pen + 1 : pen) * pen_stride, *inK = q.faceK + (size_t)(hasK ? pen + g.NB2 : pen) * pen_stride;
    double *myJ = q.faceJ + (size_t)pen * pen_stride, *myK = q.faceK + (size_t)pen * pen_stride;
    for (int I = g.NB1 - 1; I >= 0; --I) {
      const int b = I + g.NB1 * pen;
      const size_t row0 = (size_t)b * S * 64;
      const dpair *Ue2 = reinterpret_cast<const dpair *>(Ue + row0), *Un2 = reinterpret_cast<const dpair *>(Un + row0),
                  *Uu2 = reinterpret_cast<const dpair *>(Uu + row0), *Di2 = reinterpret_cast<const dpair *>(Di + row0);
      dpair *x2 = reinterpret_cast<dpair *>(x + row0);
      const ipair *cm2 = reinterpret_cast<const ipair *>(cellmap + row0);
      dpair ue[DP], un[DP], uu[DP], di[DP], rr[DP];
      ipair ix[SCATTER ? DP : 1];
      auto fetch = [&](int u, int p) {
        const size_t o = (size_t)max(p, 0) * 64;
        ue[u] = ldnt(at_lane(Ue2 + o, lane16)); un[u] = ldnt(at_lane(Un2 + o, lane16)); uu[u] = ldnt(at_lane(Uu2 + o, lane16));
        di[u] = ldnt(at_lane(Di2 + o, lane16));
        rr[u] = *at_lane(x2 + o, lane16);
        if (SCATTER) ix[u] = ldnt(at_lane(cm2 + o, threadIdx.x * 8u));
      };
#pragma unroll
      for (int u = 0; u < DP; ++u) { fetch(u, NP - 1 - u); __builtin_amdgcn_sched_barrier(0); }
      double xE = *(I < g.NB1 - 1 ? x + box_slot(g, b + 1, jp + kp, lane) : x + row0);   // (the wave's own store of the box before)
      if (I == g.NB1 - 1) xE = 0.0;
      const double *bJ = inJ + ((size_t)I * LB + kStreamLead) * 8, *bK = inK + ((size_t)I * LB + kStreamLead) * 8;
      double cJ = 0.0, cK = 0.0;
      auto request = [&](int c) {
        if (hasJ) cJ = __hip_atomic_load(bJ + 64 * c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hasK) cK = __hip_atomic_load(bK + 64 * c + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto accept = [&](int c) {
        const bool need = 8 * c + (lane >> 3) >= 7;   // (lines below 7 have no producer step)
        unsigned spins = 0;
        while (__any((need && hasJ && is_sentinel(cJ)) || (need && hasK && is_sentinel(cK)))) {
          __builtin_amdgcn_s_sleep(2);
          request(c);
          if (++spins > q.max_spins) { if (lane == 0) q.sc[S_ERR] = 1.0; break; }
        }
        lineJ[64 * c + lane] = (need && hasJ) ? cJ : 0.0;
        lineK[64 * c + lane] = (need && hasK) ? cK : 0.0;
        if (need && hasJ) const_cast<double *>(bJ)[64 * c + lane] = __longlong_as_double(-1LL);
        if (need && hasK) const_cast<double *>(bK)[64 * c + lane] = __longlong_as_double(-1LL);
        __builtin_amdgcn_wave_barrier();
      };
      request(NCH - 1);
      accept(NCH - 1);
      if (NCH > 1) request(NCH - 2);
      const double *pN = lineJ + (size_t)(S - 1) * 8 + kp, *pU = lineK + (size_t)(S - 1) * 8 + jp;
      double *oJ = myJ + ((size_t)I * LB + kStreamLead + S - 1 + 7) * 8 + kp, *oK = myK + ((size_t)I * LB + kStreamLead + S - 1 + 7) * 8 + jp;   // line h + 7
      double xprev = 0.0;
      auto step = [&](int h, double r, double ce, double cn, double cu, double d) {
        const double a1 = lane_dpp<0x130>(xprev), aB = lane_shift(xprev, addrB);
        const double nN = *pN, nU = *pU;
        pN -= 8; pU -= 8;
        const double xe = h == hE ? xE : xprev, xn = inN ? a1 : nN, xu = inU ? aB : nU;
        double s = r;
        s -= ce * xe;
        s -= cn * xn;
        s -= cu * xu;
        s *= d;
        xprev = s;
        if (outJ && mJ) __hip_atomic_store(oJ, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (outK && mK) __hip_atomic_store(oK, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oJ -= 8; oK -= 8;
        return s;
      };
      for (int p0 = NP - 1; p0 >= 0; p0 -= DP) {
#pragma unroll
        for (int u = 0; u < DP; ++u) {
          const int p = p0 - u;
          if (u % 4 == 0) {   // steps 2p+1 .. 2p-6 read lines 8c+7 .. 8c, c = p / 4 (NP is a multiple of 4: p = 3 mod 4 here)
            const int c = p >> 2;
            if (p < NP - 1 && c >= 0 && p >= 0) { accept(c); if (c >= 1) request(c - 1); }
          }
          dpair out;
          out.y = step(2 * p + 1, rr[u].y, ue[u].y, un[u].y, uu[u].y, di[u].y);
          out.x = step(2 * p, rr[u].x, ue[u].x, un[u].x, uu[u].x, di[u].x);
          dpair *dst = p >= 0 ? x2 + (size_t)p * 64 : sink;
          *at_lane(dst, lane16) = out;
          if (SCATTER && p >= 0) {
            if (ix[u].x >= 0) zvec[ix[u].x] = out.x;
            if (ix[u].y >= 0) zvec[ix[u].y] = out.y;
          }
          fetch(u, p - DP);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// Numeric ILU(0) of the same grids, box by box.  On a 7-point pattern (extents other than 2) no upper entry of a pivot row meets
// an off-diagonal entry of the row being eliminated, so only the diagonal changes: with the pivots in ascending order (down,
// south, west) m = a(c,k) / u(k,k) replaces a(c,k) and u(c,c) -= m * a(k,c) -- the arithmetic of the elimination program of
// the general kernel (k_ilu_factor_program), in its order: the same bits.  The recurrence runs on the reciprocal pivots,
// which travel between cells exactly as the solution does in k_box_lower; in place: L <- multipliers, Di <- 1 / u(c,c).
template <int DP>
__global__ __launch_bounds__(64) void k_box_factor(BoxGrid g, const int *__restrict__ boxes, double *Ld, double *Ls, double *Lw,
                                                   const double *__restrict__ Td, const double *__restrict__ Ts,
                                                   const double *__restrict__ Tw, double *Di, double *scratch) {
  extern __shared__ double face[];
  const int b = boxes[blockIdx.x], lane = threadIdx.x, S = g.S, NP = S >> 1;
  const int jp = lane & (g.B2 - 1), kp = lane >> g.lg2;
  const int I = b % g.NB1, J = (b / g.NB1) % g.NB2, K = b / (g.NB1 * g.NB2);
  const bool useS = g.NB2 > 1, useD = g.NB3 > 1;
  double *fS = face + 8, *fD = fS + (useS ? (S + g.B3 + 1) * g.B3 : 0);
  for (int e = lane; e < box_face_doubles(g); e += 64) face[e] = 0.0;
  __syncthreads();
  if (J > 0)     // reciprocal pivots on the faces of the boxes behind (as the faces of y in k_box_lower)
    for (int e = lane; e < g.B1 * g.B3; e += 64) {
      const int i = e / g.B3, k = e - i * g.B3;
      fS[(i + g.B3 - 1) * g.B3 + k] = Di[box_slot(g, b - g.NB1, i + g.B2 - 1 + k, g.B2 - 1 + g.B2 * k)];
    }
  if (K > 0)
    for (int e = lane; e < g.B1 * g.B2; e += 64) {
      const int i = e >> g.lg2, j = e & (g.B2 - 1);
      fD[(i + g.B2 - 1) * g.B2 + j] = Di[box_slot(g, b - g.NB1 * g.NB2, i + j + g.B3 - 1, j + g.B2 * (g.B3 - 1))];
    }
  const int hW = jp + kp;
  double idW = 0.0;
  if (I > 0) idW = Di[box_slot(g, b - 1, g.B1 - 1 + hW, lane)];
  const size_t row0 = (size_t)b * S * 64;
  dpair *Ld2 = reinterpret_cast<dpair *>(Ld + row0), *Ls2 = reinterpret_cast<dpair *>(Ls + row0), *Lw2 = reinterpret_cast<dpair *>(Lw + row0),
        *Di2 = reinterpret_cast<dpair *>(Di + row0), *sink = reinterpret_cast<dpair *>(scratch);
  const dpair *Td2 = reinterpret_cast<const dpair *>(Td + row0), *Ts2 = reinterpret_cast<const dpair *>(Ts + row0),
              *Tw2 = reinterpret_cast<const dpair *>(Tw + row0);
  const unsigned lane16 = threadIdx.x * 16u;
  dpair ld[DP], ls[DP], lw[DP], td[DP], ts[DP], tw[DP], cc[DP];
  auto fetch = [&](int u, int p) {
    const size_t o = (size_t)min(p, NP - 1) * 64;
    ld[u] = *at_lane(Ld2 + o, lane16); ls[u] = *at_lane(Ls2 + o, lane16); lw[u] = *at_lane(Lw2 + o, lane16);
    td[u] = ldnt(at_lane(Td2 + o, lane16)); ts[u] = ldnt(at_lane(Ts2 + o, lane16)); tw[u] = ldnt(at_lane(Tw2 + o, lane16));
    cc[u] = *at_lane(Di2 + o, lane16);
  };
#pragma unroll
  for (int u = 0; u < DP; ++u) { fetch(u, u); __builtin_amdgcn_sched_barrier(0); }
  __syncthreads();
  const double *pS = useS ? fS + (g.B3 - 1 - kp) * g.B3 + kp : face, *pD = useD ? fD + (g.B2 - 1 - jp) * g.B2 + jp : face;   // (face[0] = 0)
  const int incS = useS ? g.B3 : 0, incD = useD ? g.B2 : 0;
  const bool inS = jp != 0, inD = kp != 0;
  const int addrB = ((lane - g.B2) & 63) << 2;
  double idprev = 0.0, nS = *pS, nD = *pD;
  struct Out { double md, ms, mw, id; };
  auto step = [&](int h, double acc, double cd, double cs, double cw, double xd, double xs, double xw) {
    const double a1 = lane_dpp<0x138>(idprev), aB = lane_shift(idprev, addrB);
    const double idd = inD ? aB : nD, ids = inS ? a1 : nS, idw = h == hW ? idW : idprev;
    pS += h + 1 < S ? incS : 0; pD += h + 1 < S ? incD : 0;
    nS = *pS; nD = *pD;
    Out o;
    double d = acc;
    o.md = cd * idd;
    d -= o.md * xd;
    o.ms = cs * ids;
    d -= o.ms * xs;
    o.mw = cw * idw;
    d -= o.mw * xw;
    o.id = 1.0 / d;
    idprev = o.id;
    return o;
  };
  for (int p0 = 0; p0 < NP; p0 += DP) {
#pragma unroll
    for (int u = 0; u < DP; ++u) {
      const int p = p0 + u;
      const Out x = step(2 * p, cc[u].x, ld[u].x, ls[u].x, lw[u].x, td[u].x, ts[u].x, tw[u].x);
      const Out y = step(2 * p + 1, cc[u].y, ld[u].y, ls[u].y, lw[u].y, td[u].y, ts[u].y, tw[u].y);
      const bool real = p < NP;
      const size_t o = (size_t)p * 64;
      dpair v;
      v.x = x.md; v.y = y.md; *at_lane(real ? Ld2 + o : sink, lane16) = v;
      v.x = x.ms; v.y = y.ms; *at_lane(real ? Ls2 + o : sink, lane16) = v;
      v.x = x.mw; v.y = y.mw; *at_lane(real ? Lw2 + o : sink, lane16) = v;
      v.x = x.id; v.y = y.id; *at_lane(real ? Di2 + o : sink, lane16) = v;
      fetch(u, p + DP);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_gather(int n, const int *idx, const double *in, double *out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(kBlock) void k_gather_flags(int n, const int *idx, const int *in, signed char *out, int zero_is_one) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) { const int v = in[idx[i]]; out[i] = (signed char)(zero_is_one ? (v == 0) : v); }
}
__global__ __launch_bounds__(kBlock) void k_scatter_mapped(int n, const int *idx, const int *map, const int *in, int *out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) out[idx[i]] = map[in[i]];
}
__global__ __launch_bounds__(kBlock) void k_scatter(int n, const int *idx, const double *in, double *out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) out[idx[i]] = in[i];
}
__global__ __launch_bounds__(kBlock) void k_pmult(int n, const double *a, const double *x, double *y) {
  vec2_loop(n, [&](size_t q) { const double2 u = CD2(a)[q], v = CD2(x)[q]; D2(y)[q] = make_double2(u.x * v.x, u.y * v.y); }, [&](size_t i) { y[i] = a[i] * x[i]; });
}
__global__ __launch_bounds__(kBlock) void k_waxpy(int n, double *w, const double *sc, int slot, double sign, double aval,
                                                  const double *x, const double *y) {
  const double a = slot >= 0 ? sign * sc[slot] : aval;
  vec2_loop(n, [&](size_t q) { const double2 u = CD2(x)[q], v = CD2(y)[q]; D2(w)[q] = make_double2(fma(a, u.x, v.x), fma(a, u.y, v.y)); }, [&](size_t i) { w[i] = fma(a, x[i], y[i]); });
}
// p <- r - omegaold*beta*v + beta*p  (VecAXPBYPCZ(P,1.0,-omegaold*beta,beta,R,V), bcgs.c)
__global__ __launch_bounds__(kBlock) void k_bcgs_p(int n, double *P, const double *Pold, const double *R, const double *V,
                                                   const double *sc) {
  const double beta = sc[S_BETA], ob = -sc[S_OMEGAOLD] * beta;
  vec2_loop(n, [&](size_t q) {
              const double2 r = CD2(R)[q], v = CD2(V)[q], p = CD2(Pold)[q];
              D2(P)[q] = make_double2(1.0 * r.x + ob * v.x + beta * p.x, 1.0 * r.y + ob * v.y + beta * p.y);
            }, [&](size_t i) { P[i] = 1.0 * R[i] + ob * V[i] + beta * Pold[i]; });
}

inline int red_grid(int n) {
  long long b = ((long long)n / 2 + kRedPairs - 1) / kRedPairs;
  return (int)(b < 1 ? 1 : (b > kMaxPartialBlocks ? kMaxPartialBlocks : b));
}

// x <- alpha p + omega s + x ; r <- s - omega t ; partial sums of r.r   (bcgs.c)
// FIRST: x is known to be zero (first iteration, zero initial guess) and is not read.  Later iterations also reduce
// (r,rp), the next iteration's rho, while r is in registers (partials row 1).
// XX (with FIRST): the partial sums of x.x as well, in row 1 -- the line search's ||Y||^2 when the solve ends with this iteration
template <bool FIRST, bool XX = false>
__global__ __launch_bounds__(kBlock) void k_bcgs_xr(int n, double *X, double *R, const double *P, const double *S,
                                                    const double *T, const double *RP, const double *sc, double *partials) {
  __shared__ double lds[kBlock / 64];
  const double alpha = sc[S_ALPHA], omega = sc[S_OMEGA];
  double acc = 0.0, acc2 = 0.0;
  vec2_loop(n, [&](size_t q) {
              const double2 s = CD2(S)[q], p = CD2(P)[q], t = CD2(T)[q], x = FIRST ? make_double2(0.0, 0.0) : CD2(X)[q];
              const double2 xn = make_double2(alpha * p.x + omega * s.x + x.x, alpha * p.y + omega * s.y + x.y);
              D2(X)[q] = xn;
              if (XX) { acc2 += xn.x * xn.x; acc2 += xn.y * xn.y; }
              const double r0 = s.x - omega * t.x, r1 = s.y - omega * t.y;
              if (R) D2(R)[q] = make_double2(r0, r1);   // (R == nullptr: only ||r|| is wanted -- the solve is expected to end here)
              acc += r0 * r0;
              acc += r1 * r1;
              if (!FIRST) { const double2 rp = CD2(RP)[q]; acc2 += r0 * rp.x; acc2 += r1 * rp.y; }
            }, [&](size_t i) {
              const double s = S[i];
              const double xn = alpha * P[i] + omega * s + (FIRST ? 0.0 : X[i]);
              X[i] = xn;
              if (XX) acc2 += xn * xn;
              const double r = s - omega * T[i];
              if (R) R[i] = r;
              acc += r * r;
              if (!FIRST) acc2 += r * RP[i];
            });
  double t = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  if (!FIRST || XX) {
    t = block_sum(acc2, lds);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t;
  }
}

// block-Jacobi apply with the Krylov dot products of its result fused in (the solve holds z_k in a register when it
// writes it): ND = 1 leaves the partial sums of z.z, ND = 2 those of z.w and z.z (in that order), layout partials[j*nb + b]
template <int ND>
__global__ __launch_bounds__(kBlock) void k_tri_solve_dots(TriArgs a, const double *__restrict__ r, double *__restrict__ z,
                                                           const double *__restrict__ w, double *__restrict__ partials) {
  __shared__ double lds[kBlock / 64];
  const int c = blockIdx.x * kBlock + threadIdx.x;
  double acc0 = 0.0, acc1 = 0.0;
  if (c < a.ncols) {
    const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
    const size_t cbase = (size_t)tile * 64 * a.nz + (c & 63);
    double prev = 0.0;
#pragma unroll 4
    for (int k = 0; k < a.nz; ++k) {
      const size_t i = cbase + (size_t)k * tw;
      const double d = ldnt(r + i) - ldnt(a.l + i) * prev;
      z[i] = d;
      prev = d;
    }
    double next = 0.0;
#pragma unroll 4
    for (int k = a.nz - 1; k >= 0; --k) {
      const size_t i = cbase + (size_t)k * tw;
      const double xk = tri_back(tri_scale(z[i], ldnt(a.invb + i)), ldnt(a.c + i), next);
      z[i] = xk;
      next = xk;
      if (ND == 2) { acc0 += xk * w[i]; acc1 += xk * xk; } else acc0 += xk * xk;
    }
  }
  const double t0 = block_sum(acc0, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  if (ND == 2) {
    __syncthreads();
    const double t1 = block_sum(acc1, lds);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t1;
  }
}

// The same solve with the forward-swept right-hand side kept on chip.  k_tri_solve_dots writes L^{-1} r to z and reads it
// back in the second sweep (16 of its 64 bytes per cell, PMC: profiles/r01_pmc_solver.md); here ONE WAVE owns a tile of 64
// columns and parks L^{-1} r in its own LDS slab (nz x 512 bytes: a 100-level column tile takes 51 KB, three waves per CU).
// At three waves per CU the memory system is kept busy by depth instead of by occupancy: the operands travel through a
// ring of R register groups of G levels each (2-3 streams, one 512-byte wave load per stream and level); a group is
// refilled as soon as it has been consumed, and consuming a group waits for that group only (vector-memory operations
// retire in order), so (R-1) G .. R G levels of every stream are in flight per wave at all times.  The loop bodies are
// straight-line code -- every path issues the same loads and stores (addresses clamped, idle lanes of a ragged last tile
// store to a scratch line) -- so that the compiler can count the outstanding operations instead of draining them.
// Arithmetic per column is the one of k_tri_solve_dots (same order, same roundings); the dot products are summed per lane
// over the column, then over the 64 lanes: one partial sum per tile, layout partials[j*nb + tile].
// SPMV (round 3): the right-hand side is not read but formed on the way, r = A x with A the column's tridiagonal taken from the
// CSR values (`r` then points at x): z = M^{-1} (A x) in one pass -- what BiCGStab / GMRES ask for twice per iteration on a
// vertical-only grid, where the CSR matrix IS the block diagonal of column tridiagonals.  A x is never written or read back
// (98 -> 72 bytes per cell for the SpMV + column solve pair).  Row k is summed as the CSR SpMV kernels sum it -- products rounded
// one by one, added in entry order (lower, diagonal, upper), no fused multiply-add -- so z equals the two-kernel result bit for
// bit.  The slots of a row are computed (rows of a column tile are level-major, 2 entries in the end rows and 3 between: the
// pattern of a vertical-only grid in the tiled layout); an entry a row does not have is read from a harmless slot and meets x = 0.
// SPMV == 2: x itself is produced first -- x = U^{-1} d, the back substitution of a right-hand side the assembly pass already
// forward-eliminated (k_tri_back_dots' arithmetic), written to `xout` and parked in the slab -- then z = M^{-1} (A x) as above with
// x taken from the slab: the start of a BiCGStab solve (rp = M^{-1} F, v = M^{-1} A rp) in one kernel, with (z,x), (z,z), (x,x).
// SPMV == 3: x = r + a v with a = -sc[S_ALPHA] (BiCGStab's s = r - alpha v, VecWAXPY) is formed on the way, written to `xout`, and is
// also the w of the dot products: the second half of a BiCGStab iteration's products -- s, t = M^{-1} A s, (t,s), (t,t) -- in one kernel.
// FDOT (round 5, with SPMV): the partial sums of fdot . (A x) as well -- row k's sum is in a register when the sweep eliminates it,
// so the line search's F . (J Y) of a one-iteration BiCGStab solve (Y = alpha p + omega s) costs one more 8-byte stream in the two
// kernels that form J p and J s instead of a product pass of its own; partial sums in row ND (+ 1 with SPMV == 2) of `partials`.
template <int ND, int G, int RF, int RB, int SPMV, bool FDOT = false>
__global__ __launch_bounds__(64) void k_tri_solve_lds(TriArgs a, const double *__restrict__ r, double *__restrict__ z,
                                                      const double *__restrict__ w, double *__restrict__ partials,
                                                      double *__restrict__ scratch64, double *__restrict__ xout,
                                                      const double *__restrict__ vsub, const double *__restrict__ sc, Fold fold,
                                                      const double *__restrict__ fdot = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double dsl[];   // [nz][64]
  const int lane = threadIdx.x, tile = blockIdx.x;
  const int nz = a.nz;
  const int tw = min(64, a.ncols - (tile << 6));
  const bool valid = lane < tw;
  const size_t cbase = (size_t)tile * 64 * nz + (valid ? lane : 0);   // idle lanes of the last tile shadow lane 0 ...
  double *zs = valid ? z + cbase : scratch64 + lane;                  // ... and store into a scratch line
  const size_t tws = valid ? (size_t)tw : 0;
  double accx = 0.0, accf = 0.0;
  if (SPMV == 2) {   // x = U^{-1} d, top down
    double *xs = valid ? xout + cbase : scratch64 + lane;
    double cb[RB][G], db[RB][G];
    auto fetch = [&](int q, int ktop) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const size_t i = cbase + (size_t)max(ktop - j, 0) * tw;
        // (g is read again by this kernel's last sweep, a tile's worth later: a plain load, so that the caches may keep it)
        cb[q][j] = a.c[i]; db[q][j] = ldnt(r + i);
      }
    };
#pragma unroll
    for (int q = 0; q < RB; ++q) { fetch(q, nz - 1 - q * G); __builtin_amdgcn_sched_barrier(0); }
    double next = 0.0;
    int k0 = nz - 1;
    for (; k0 - RB * G + 1 >= 0; k0 -= RB * G) {
#pragma unroll
      for (int q = 0; q < RB; ++q) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int k = k0 - q * G - j;
          const double xk = tri_back(db[q][j], cb[q][j], next);
          next = xk;
          xs[(size_t)k * tws] = xk;   // (plain store: the last sweep reads it back as the dot products' w)
          dsl[k * 64 + lane] = xk;
          accx += xk * xk;
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(q, k0 - (RB + q) * G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = k0 - q * G - j;
        if (k >= 0) {
          const double xk = tri_back(db[q][j], cb[q][j], next);
          next = xk;
          xs[(size_t)k * tws] = xk;   // (plain store: the last sweep reads it back as the dot products' w)
          dsl[k * 64 + lane] = xk;
          accx += xk * xk;
        }
      }
  }
  if (SPMV) {
#pragma clang fp contract(off)   // the row sums below must round like the CSR SpMV's; d = r - l prev is written as the fma it is elsewhere
    const int vl = valid ? lane : 0;
    const double *__restrict__ vt = a.val + (size_t)tile * 64 * (3 * nz - 2);   // this tile's CSR values
    // The 3 tw entries of a middle level are contiguous, row after row: the wave loads them as three fully coalesced segments
    // (lane l: entries l, tw + l, 2 tw + l) and turns them into (lower, diagonal, upper) of its own row -- entries 3l, 3l + 1,
    // 3l + 2 -- through a 1.5 KB LDS scratch when the level is used; three 8-byte loads at a 24-byte lane stride would touch every
    // line of the segment three times.  The two end levels (2 entries per row) are read directly.
    __shared__ double tr[3 * 64];
    double xb[RF][G], v0[RF][G], v1[RF][G], v2[RF][G], sb[SPMV == 3 ? RF : 1][SPMV == 3 ? G : 1], fb[FDOT ? RF : 1][FDOT ? G : 1];
    const double na = (SPMV == 3) ? -1.0 * sc[S_ALPHA] : 0.0;
    double *ss = (SPMV == 3) ? (valid ? xout + cbase : scratch64 + lane) : nullptr;
    auto fetch = [&](int q, int kfirst) {   // levels kfirst .. kfirst+G-1 (clamped) into group q
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = min(kfirst + j, nz - 1);
        const size_t i = cbase + (size_t)k * tw;
        xb[q][j] = (SPMV == 1 || SPMV == 3) ? ldnt(r + i) : 0.0;   // (SPMV == 2: x is in the slab)
        if (SPMV == 3) sb[q][j] = ldnt(vsub + i);
        if (FDOT) fb[q][j] = ldnt(fdot + i);
        if (k == 0) {                 // [di up]: lower slot harmless (meets x = 0)
          v0[q][j] = ldnt(vt + 2 * vl); v1[q][j] = v0[q][j]; v2[q][j] = ldnt(vt + 2 * vl + 1);
        } else if (k == nz - 1) {     // [lo di]: upper slot harmless
          const int s = 2 * tw + 3 * (nz - 2) * tw + 2 * vl;
          v0[q][j] = ldnt(vt + s); v1[q][j] = ldnt(vt + s + 1); v2[q][j] = v1[q][j];
        } else {                      // three coalesced segments of the level's 3 tw entries
          const int b = 2 * tw + 3 * (k - 1) * tw + vl;
          v0[q][j] = ldnt(vt + b); v1[q][j] = ldnt(vt + b + tw); v2[q][j] = ldnt(vt + b + 2 * tw);
        }
      }
    };
#pragma unroll
    for (int q = 0; q < RF; ++q) { fetch(q, q * G); __builtin_amdgcn_sched_barrier(0); }
    // level k arrives: row k-1 is complete (its upper neighbour is x_k).  Carried: x_{k-2}, x_{k-1}, row k-1's entries and factor
    // The elimination factor l_k = lo_k / piv_{k-1} is not read: the row's entries are here anyway, so the factorisation is redone on
    // the way (k_tri_factor's arithmetic: l = lo inv_prev, piv = fma(-l, up_prev, di), inv = 1 / piv -- one division per level in a
    // chain the memory stream leaves room for), 8 of the kernel's ~90 bytes per cell.  The stored factors stay for the back sweep.
    double xm2 = 0.0, xm1 = 0.0, plo = 0.0, pdi = 0.0, pup = 0.0, pl = 0.0, prev = 0.0, finv = 0.0, pf = 0.0;
    auto finish = [&](int kr, double xnext) {   // row kr from the carried values and x_{kr+1}
      const double p0 = plo * xm2, p1 = pdi * xm1, p2 = pup * xnext;
      const double rk = (p0 + p1) + p2;
      if (FDOT) accf += pf * rk;                // (row kr of A x meets fdot_kr: the line search's F . (J Y), see FDOT above)
      const double d = fma(-pl, prev, rk);
      prev = d;
      dsl[kr * 64 + lane] = tri_scale(d, finv);   // (finv: still row kr's -- parked as the back sweep wants it)
    };
    auto step = [&](int k, double xk, double lo, double di, double up, double vk, double fk) {
      if (SPMV == 2) xk = dsl[k * 64 + lane];   // (rows below k have been replaced by their eliminated sums already: not needed any more)
      if (SPMV == 3) { xk = fma(na, vk, xk); ss[(size_t)k * tws] = xk; }   // s = r - alpha v (k_waxpy's arithmetic), kept for the update kernel
      if (k > 0 && k < nz - 1) {   // a middle level: (lo, di, up) hold segment entries l, tw + l, 2 tw + l -> this lane's row
        if (valid) { tr[lane] = lo; tr[tw + lane] = di; tr[2 * tw + lane] = up; }
        __builtin_amdgcn_wave_barrier();
        lo = tr[3 * vl]; di = tr[3 * vl + 1]; up = tr[3 * vl + 2];
        __builtin_amdgcn_wave_barrier();
      }
      if (k > 0) finish(k - 1, xk);
      const double lk = lo * finv;              // (row 0: finv = 0, whatever the harmless lower slot holds)
      finv = 1.0 / fma(-lk, pup, di);           // pup: still row k-1's upper entry
      xm2 = (k > 0) ? xm1 : 0.0;   // x_{-1} = 0 meets the lower entry row 0 does not have
      xm1 = xk;
      plo = lo; pdi = di; pup = up; pl = lk;
      if (FDOT) pf = fk;
    };
    int k0 = 0;
    for (; k0 + RF * G <= nz; k0 += RF * G) {
#pragma unroll
      for (int q = 0; q < RF; ++q) {
#pragma unroll
        for (int j = 0; j < G; ++j) step(k0 + q * G + j, xb[q][j], v0[q][j], v1[q][j], v2[q][j], SPMV == 3 ? sb[q][j] : 0.0, FDOT ? fb[q][j] : 0.0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(q, k0 + (RF + q) * G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < RF; ++q)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = k0 + q * G + j;
        if (k < nz) step(k, xb[q][j], v0[q][j], v1[q][j], v2[q][j], SPMV == 3 ? sb[q][j] : 0.0, FDOT ? fb[q][j] : 0.0);
      }
    finish(nz - 1, 0.0);   // the top row: no upper neighbour
  } else {
    double rb[RF][G], lb[RF][G];
    auto fetch = [&](int q, int kfirst) {   // levels kfirst .. kfirst+G-1 (clamped) into group q
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const size_t i = cbase + (size_t)min(kfirst + j, nz - 1) * tw;
        rb[q][j] = ldnt(r + i); lb[q][j] = ldnt(a.l + i);
      }
    };
#pragma unroll
    for (int q = 0; q < RF; ++q) { fetch(q, q * G); __builtin_amdgcn_sched_barrier(0); }
    double prev = 0.0;
    int k0 = 0;
    for (; k0 + RF * G <= nz; k0 += RF * G) {
#pragma unroll
      for (int q = 0; q < RF; ++q) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const double d = rb[q][j] - lb[q][j] * prev;
          prev = d;
          dsl[(k0 + q * G + j) * 64 + lane] = d;
        }
        __builtin_amdgcn_sched_barrier(0);   // program order = issue order: a group is refilled right after its use
        fetch(q, k0 + (RF + q) * G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the last, partial round: its operands are in the ring already (the refills above were clamped to the last level)
#pragma unroll
    for (int q = 0; q < RF; ++q)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = k0 + q * G + j;
        if (k < nz) {
          const double d = rb[q][j] - lb[q][j] * prev;
          prev = d;
          dsl[k * 64 + lane] = d;
        }
      }
  }
  double acc0 = 0.0, acc1 = 0.0;
  {
    double cb[RB][G], ib[SPMV ? 1 : RB][SPMV ? 1 : G], wb[RB][G];
    auto fetch = [&](int q, int ktop) {   // levels ktop, ktop-1, ..., ktop-G+1 (clamped at 0) into group q
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const size_t i = cbase + (size_t)max(ktop - j, 0) * tw;
        cb[q][j] = ldnt(a.c + i);
        if (!SPMV) ib[q][j] = ldnt(a.invb + i);   // (SPMV: the slab holds d inv already)
        if (ND == 2) wb[q][j] = (SPMV >= 2) ? xout[i] : ldnt(w + i);   // (SPMV >= 2: w is the x this kernel wrote a tile's worth
                                                                         // ago -- same pointer, plain store and load: the caches may still hold it)
      }
    };
#pragma unroll
    for (int q = 0; q < RB; ++q) { fetch(q, nz - 1 - q * G); __builtin_amdgcn_sched_barrier(0); }
    double next = 0.0;
    int k0 = nz - 1;
    for (; k0 - RB * G + 1 >= 0; k0 -= RB * G) {
#pragma unroll
      for (int q = 0; q < RB; ++q) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int k = k0 - q * G - j;
          const double xk = tri_back(SPMV ? dsl[k * 64 + lane] : tri_scale(dsl[k * 64 + lane], ib[q][j]), cb[q][j], next);
          next = xk;
          stnt(xk, zs + (size_t)k * tws);
          if (ND == 2) { acc0 += xk * wb[q][j]; acc1 += xk * xk; } else acc0 += xk * xk;
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(q, k0 - (RB + q) * G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q)
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = k0 - q * G - j;
        if (k >= 0) {
          const double xk = tri_back(SPMV ? dsl[k * 64 + lane] : tri_scale(dsl[k * 64 + lane], ib[q][j]), cb[q][j], next);
          next = xk;
          stnt(xk, zs + (size_t)k * tws);
          if (ND == 2) { acc0 += xk * wb[q][j]; acc1 += xk * xk; } else acc0 += xk * xk;
        }
      }
  }
  if (!valid) acc0 = acc1 = accx = accf = 0.0;
  const bool folded = SPMV && fold.counter != nullptr;
  const double t0 = wave_sum(acc0);
  if (lane == 0) store_partial(partials + tile, t0, folded);
  if (ND == 2) {
    const double t1 = wave_sum(acc1);
    if (lane == 0) store_partial(partials + (size_t)gridDim.x + tile, t1, folded);
  }
  if (SPMV == 2) {
    const double t2 = wave_sum(accx);
    if (lane == 0) store_partial(partials + 2 * (size_t)gridDim.x + tile, t2, folded);
  }
  if (FDOT) {
    const double t3 = wave_sum(accf);
    if (lane == 0) store_partial(partials + (size_t)(ND + (SPMV == 2 ? 1 : 0)) * gridDim.x + tile, t3, folded);
  }
  if (SPMV) fold_tail<64>(fold, partials, (int)gridDim.x, lane == 0);
}

// backward substitution only, from a forward-eliminated right-hand side d (left by the assembly pass): z = U^{-1} d with
// the partial sums of z.z
__global__ __launch_bounds__(kBlock) void k_tri_back_dots(TriArgs a, const double *__restrict__ d, double *__restrict__ z,
                                                          double *__restrict__ partials) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  double acc = 0.0;
  if (c < a.ncols) {
    const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
    const size_t cbase = (size_t)tile * 64 * a.nz + (c & 63);
    double next = 0.0;
#pragma unroll 4
    for (int k = a.nz - 1; k >= 0; --k) {
      const size_t i = cbase + (size_t)k * tw;
      const double xk = tri_back(ldnt(d + i), ldnt(a.c + i), next);   // (d: the march's forward-eliminated residual, scaled by inv already)
      stnt(xk, z + i);
      next = xk;
      acc += xk * xk;
    }
  }
  // one partial sum per wave = per 64-column tile, as the LDS column kernels leave them: the same sum whichever kernel made z
  const double t = wave_sum(acc);
  const int tile = c >> 6;
  if ((threadIdx.x & 63) == 0 && tile < (a.ncols + 63) / 64) partials[tile] = t;
}

// dot-product patterns the solvers use; every distinct vector is streamed once
//  PAT 0: a.a            PAT 1: a.b            PAT 2: a.b, b.b          PAT 3: a.a, b.b, c.d         PAT 4: a.a, b.b
template <int PAT>
__global__ __launch_bounds__(kBlock) void k_dots_pat(int n, const double *__restrict__ a, const double *__restrict__ b,
                                                     const double *__restrict__ c, const double *__restrict__ d,
                                                     double *partials) {
  __shared__ double lds[kBlock / 64];
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  red2_loop(n, [&](size_t q) {
              const double2 va = CD2(a)[q];
              if (PAT == 0) { acc0 += va.x * va.x; acc0 += va.y * va.y; }
              if (PAT == 1) { const double2 vb = CD2(b)[q]; acc0 += va.x * vb.x; acc0 += va.y * vb.y; }
              if (PAT == 2) {
                const double2 vb = CD2(b)[q];
                acc0 += va.x * vb.x; acc0 += va.y * vb.y; acc1 += vb.x * vb.x; acc1 += vb.y * vb.y;
              }
              if (PAT == 3) {
                const double2 vb = CD2(b)[q], vc = CD2(c)[q], vd = CD2(d)[q];
                acc0 += va.x * va.x; acc0 += va.y * va.y; acc1 += vb.x * vb.x; acc1 += vb.y * vb.y;
                acc2 += vc.x * vd.x; acc2 += vc.y * vd.y;
              }
              if (PAT == 4) {
                const double2 vb = CD2(b)[q];
                acc0 += va.x * va.x; acc0 += va.y * va.y; acc1 += vb.x * vb.x; acc1 += vb.y * vb.y;
              }
            }, [&](size_t i) {
              const double va = a[i];
              if (PAT == 0) acc0 += va * va;
              if (PAT == 1) acc0 += va * b[i];
              if (PAT == 2) { acc0 += va * b[i]; acc1 += b[i] * b[i]; }
              if (PAT == 3) { acc0 += va * va; acc1 += b[i] * b[i]; acc2 += c[i] * d[i]; }
              if (PAT == 4) { acc0 += va * va; acc1 += b[i] * b[i]; }
            });
  constexpr int NV = (PAT == 0 || PAT == 1) ? 1 : (PAT == 3 ? 3 : 2);
  double t = block_sum(acc0, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  if (NV > 1) {
    t = block_sum(acc1, lds);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t;
  }
  if (NV > 2) {
    t = block_sum(acc2, lds);
    if (threadIdx.x == 0) partials[(size_t)2 * gridDim.x + blockIdx.x] = t;
  }
}

// multi-dot x . y_j, j0 <= j < j0+nv (nv <= kMdotGroup): x streamed once per group (GMRES Gram-Schmidt)
constexpr int kMdotGroup = 8;
__global__ __launch_bounds__(kBlock) void k_mdot(DotArgs a, int j0, int nv, double *partials) {
  __shared__ double lds[kBlock / 64];
  double acc[kMdotGroup];
#pragma unroll
  for (int j = 0; j < kMdotGroup; ++j) acc[j] = 0.0;
  const double *x = a.x[j0];
  // (pairs of rows with 16-byte accesses were measured here as well: 1.19 -> 1.38 ms per launch at 512^3, not kept; the update
  // kernel k_maxpy_neg gained 6 % from them)
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * kBlock) {
    const double xv = x[i];
#pragma unroll
    for (int j = 0; j < kMdotGroup; ++j)
      if (j < nv) acc[j] += xv * a.y[j0 + j][i];
  }
#pragma unroll
  for (int j = 0; j < kMdotGroup; ++j) {
    if (j < nv) {
      const double t = block_sum(acc[j], lds);
      if (threadIdx.x == 0) partials[(size_t)(j0 + j) * gridDim.x + blockIdx.x] = t;
    }
  }
}

// the scalar recurrences of BiCGStab between its vector kernels (one thread): the step codes of launch_bcgs_scalars
__device__ void bcgs_scalar_step(int step, double *sc) {
  if (step == 0 || step == 10) {  // rhoold = alpha = omegaold = 1; dp0 = ||M^{-1} b|| from the (rp,rp) just reduced
    sc[S_RHOOLD] = 1.0; sc[S_ALPHA] = 1.0; sc[S_OMEGAOLD] = 1.0; sc[S_FLAG] = 0.0;
    sc[S_DP0] = sqrt(sc[S_SUM0]);
  }
  if (step == 1 || step == 10) {  // rho <- (r,rp); beta = (rho/rhoold)*(alpha/omegaold)   [10 = 0 then 1, the first iteration]
    const double rho = sc[S_SUM0];
    sc[S_RHO] = rho;
    sc[S_BETA] = (rho / sc[S_RHOOLD]) * (sc[S_ALPHA] / sc[S_OMEGAOLD]);
  } else if (step == 2) {  // d1 = (v,rp); alpha = rho/d1
    const double d1 = sc[S_SUM0];
    sc[S_D1] = d1;
    sc[S_ALPHA] = sc[S_RHO] / d1;
    sc[S_FJP] = sc[S_SUM0 + 2];   // F . (J p) when the product kernel reduced it as its third sum (coupled system, ls_fold); else not read
  } else if (step == 3) {  // d1 = (s,t), d2 = (t,t); omega = d1/d2
    const double d1 = sc[S_SUM0], d2 = sc[S_SUM0 + 1];
    sc[S_ST] = d1; sc[S_D2] = d2;
    sc[S_FJS] = sc[S_SUM0 + 2];   // F . (J s), likewise
    if (d2 == 0.0) { sc[S_OMEGA] = 0.0; sc[S_FLAG] = 1.0; }  // t == 0: x += alpha p is the candidate solution
    else sc[S_OMEGA] = d1 / d2;
  } else if (step == 6) {  // steps 0 and 1 when the initial (rp,rp) was all-reduced together with the first (v,rp): it sits in S_SUM0+2
    sc[S_RHOOLD] = 1.0; sc[S_ALPHA] = 1.0; sc[S_OMEGAOLD] = 1.0; sc[S_FLAG] = 0.0;
    const double rho = sc[S_SUM0 + 2];
    sc[S_DP0] = sqrt(rho);
    sc[S_RHO] = rho;
    sc[S_BETA] = (rho / sc[S_RHOOLD]) * (sc[S_ALPHA] / sc[S_OMEGAOLD]);
  }
  if (step == 62) {  // 6 then 2: the solve's first kernel left (v,rp), (v,v), (rp,rp) in S_SUM0 .. S_SUM0+2
    sc[S_RHOOLD] = 1.0; sc[S_ALPHA] = 1.0; sc[S_OMEGAOLD] = 1.0; sc[S_FLAG] = 0.0;
    const double rho = sc[S_SUM0 + 2];
    sc[S_DP0] = sqrt(rho);
    sc[S_RHO] = rho;
    sc[S_BETA] = (rho / sc[S_RHOOLD]) * (sc[S_ALPHA] / sc[S_OMEGAOLD]);
    const double d1 = sc[S_SUM0];
    sc[S_D1] = d1;
    sc[S_ALPHA] = sc[S_RHO] / d1;
    sc[S_FJP] = sc[S_SUM0 + 3];   // F . (J p) when the kernel reduced it (k_tri_solve_lds FDOT; otherwise whatever the slot holds: not read)
  }
  if (step == 4 || step == 45) {  // end of iteration: dp^2 = (r,r); rhoold = rho; omegaold = omega; (r,rp) if it was reduced
    sc[S_DP] = sqrt(sc[S_SUM0]);
    sc[S_RRP] = sc[S_SUM0 + 1];
    sc[S_RHOOLD] = sc[S_RHO]; sc[S_OMEGAOLD] = sc[S_OMEGA];
  }
  if (step == 5 || step == 45) {  // step 1 with rho = the (r,rp) the iteration's update reduced   [45 = 4 then 5]
    const double rho = sc[S_RRP];
    sc[S_RHO] = rho;
    sc[S_BETA] = (rho / sc[S_RHOOLD]) * (sc[S_ALPHA] / sc[S_OMEGAOLD]);
  }
}
// The scalar block straight into pinned host memory, then a sequence word: the host polls that word instead of queueing a
// device-to-host copy behind the kernel and waiting for the stream (Solver::fetch_scalars).  Called by every thread of the
// block that holds the final values, after the thread that wrote them has passed a workgroup barrier (`sync_first`: this call
// places it).  System-scope stores; the sequence word follows the data (release) once every lane's stores are acknowledged.
__device__ __forceinline__ void publish_scalars(const Publish &p, const double *sc) {
  if (!p.host) return;
  __syncthreads();
  for (int q = threadIdx.x; q < S_COUNT; q += blockDim.x)
    __hip_atomic_store(p.host + q, __hip_atomic_load(sc + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this lane's stores have left
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(p.seq, p.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_bcgs_scalars(int step, double *sc, Publish pub) {
  if (blockIdx.x != 0) return;
  if (threadIdx.x == 0 && step >= 0) bcgs_scalar_step(step, sc);
  publish_scalars(pub, sc);
}

// Second stage of every reduction, in one launch: block (c, j) sums chunk c of vector j's partials in a fixed order into
// stage[j*nchunk + c]; the block whose ticket comes last sums the chunk sums (again in a fixed order, so the result does not
// depend on which block that is) into sums[S_SUM0 + j] and, step >= 0, runs the BiCGStab scalar recurrence that consumes them;
// pub.host != nullptr: it then publishes the scalar block to the host.
// Hand-off: chunk sums are agent-scope stores, acknowledged (vmcnt(0)) before the storing wave takes its ticket; the last
// block reads them with agent-scope loads after a workgroup barrier behind the returned ticket.  It leaves the counter at 0.
__global__ __launch_bounds__(kBlock) void k_finalize(int nb, int chunk, const double *partials, double *stage, double *sums,
                                                     unsigned *counter, double *sc, int step, Publish pub) {
  __shared__ double lds[kBlock / 64];
  __shared__ int last;
  const int c = blockIdx.x, j = blockIdx.y, nchunk = gridDim.x, nvec = gridDim.y;
  const int b0 = c * chunk, b1 = min(nb, b0 + chunk);
  double acc = 0.0;
  for (int b = b0 + threadIdx.x; b < b1; b += kBlock) acc += partials[(size_t)j * nb + b];
  const double t = block_sum(acc, lds);
  if (nchunk * nvec == 1) {   // nothing to wait for
    if (threadIdx.x == 0) {
      __hip_atomic_store(sums + S_SUM0, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (step >= 0) bcgs_scalar_step(step, sc);
    }
    publish_scalars(pub, sc);
    return;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(stage + (size_t)j * nchunk + c, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (ticket == (unsigned)(nchunk * nvec) - 1u);
  }
  __syncthreads();
  if (!last) return;
  for (int jj = 0; jj < nvec; ++jj) {
    double a2 = 0.0;
    for (int b = threadIdx.x; b < nchunk; b += kBlock)
      a2 += __hip_atomic_load(stage + (size_t)jj * nchunk + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double tt = nchunk == 1 ? a2 : block_sum(a2, lds);   // one chunk: thread 0 holds it (the sum of the first stage, as is)
    if (threadIdx.x == 0) __hip_atomic_store(sums + S_SUM0 + jj, tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (step >= 0) bcgs_scalar_step(step, sc);
  }
  publish_scalars(pub, sc);
}

// partials != nullptr: also the partial sums of y.y of the result (the norm GMRES takes next), one per block
// (two rows per thread as one 16-byte access per vector, four vectors' loads issued together: with one 8-byte load per vector and
// thread the update of a 134 M-row vector against 13 basis vectors ran at 0.67 of peak)
// h: the nvec Gram-Schmidt coefficients, h[j] for a.v[j]
__global__ __launch_bounds__(kBlock) void k_maxpy_neg(MaxpyArgs a, const double *__restrict__ h, double *y, double *partials) {
  __shared__ double lds[kBlock / 64];
  const size_t i = 2 * ((size_t)blockIdx.x * kBlock + threadIdx.x), n = (size_t)a.n;
  double sq = 0.0;
  if (i + 1 < n) {
    dpair v = *reinterpret_cast<const dpair *>(y + i);
    int j = 0;
    for (; j + 4 <= a.nvec; j += 4) {
      const dpair b0 = __builtin_nontemporal_load(reinterpret_cast<const dpair *>(a.v[j] + i)),
                  b1 = __builtin_nontemporal_load(reinterpret_cast<const dpair *>(a.v[j + 1] + i)),
                  b2 = __builtin_nontemporal_load(reinterpret_cast<const dpair *>(a.v[j + 2] + i)),
                  b3 = __builtin_nontemporal_load(reinterpret_cast<const dpair *>(a.v[j + 3] + i));
      const double s0 = h[j], s1 = h[j + 1], s2 = h[j + 2], s3 = h[j + 3];
      v.x -= s0 * b0.x; v.y -= s0 * b0.y;
      v.x -= s1 * b1.x; v.y -= s1 * b1.y;
      v.x -= s2 * b2.x; v.y -= s2 * b2.y;
      v.x -= s3 * b3.x; v.y -= s3 * b3.y;
    }
    for (; j < a.nvec; ++j) {
      const dpair b0 = __builtin_nontemporal_load(reinterpret_cast<const dpair *>(a.v[j] + i));
      const double s0 = h[j];
      v.x -= s0 * b0.x; v.y -= s0 * b0.y;
    }
    *reinterpret_cast<dpair *>(y + i) = v;
    sq = v.x * v.x + v.y * v.y;
  } else if (i < n) {   // the last row of an odd count
    double v = y[i];
    for (int j = 0; j < a.nvec; ++j) v -= h[j] * a.v[j][i];
    y[i] = v;
    sq = v * v;
  }
  if (partials) {
    const double t = block_sum(sq, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kBlock) void k_maxpy_val(MaxpyValArgs a, double *x) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)a.n) {
    double v = a.zero_x ? 0.0 : x[i];
    for (int j = 0; j < a.nvec; ++j) v += a.c[j] * a.v[j][i];
    x[i] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_scale_inv(int n, double *x, const double *sc, int slot) {
  const double d = sc[slot];
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) x[i] = x[i] / d;
}

}  // namespace

// ================================================================================================
// launchers
// ================================================================================================
static inline int nblk(long long n) { return (int)((n + kBlock - 1) / kBlock); }
#define LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, stream, __VA_ARGS__)

void launch_accum(const CellArrays &cell, int n, double dtInv, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_accum, dim3(nblk(n)), cell, n, dtInv, out);
}

template <int SAT_T, int DEN_T>
static void launch_cols_mode(int mode, const ColsArgs &a, dim3 grid, hipStream_t stream) {
  const size_t lds = (SAT_T >= 0 && a.soil_cls) ? (size_t)7 * a.ncls * sizeof(double) : 0;   // the class table, staged per block (<= 256 classes)
#define COLS(M, F) hipLaunchKernelGGL((k_assemble_cols<M, SAT_T, DEN_T, F>), grid, dim3(kBlock), lds, stream, a)
  if (a.col_bc) {   // complete-row variant (grid.y == 1)
    if (mode == 0) COLS(0, true); else if (mode == 1) COLS(1, true); else COLS(2, true);
  } else if (a.slot_di) {   // 3-D mesh of complete columns: a lateral pass follows
#define COLS3(M) hipLaunchKernelGGL((k_assemble_cols<M, SAT_T, DEN_T, false, true>), grid, dim3(kBlock), lds, stream, a)
    if (mode == 0) COLS3(0); else if (mode == 1) COLS3(1); else COLS3(2);
#undef COLS3
  } else {
    if (mode == 0) COLS(0, false); else if (mode == 1) COLS(1, false); else COLS(2, false);
  }
#undef COLS
}
void launch_assemble_cols(int mode, const ColsArgs &a, hipStream_t stream) {
  if (a.ncols <= 0) return;
  dim3 grid(nblk(a.ncols), a.col_bc ? 1 : a.nseg);
  const bool ifc = a.cell.eos.density_type == kDenIFC67;
  if (a.cell.satf || a.cell.por_slope) {   // per-cell saturation functions / linear porosity: the general-soil march
    ifc ? launch_cols_mode<-1, 1>(mode, a, grid, stream) : launch_cols_mode<-1, 0>(mode, a, grid, stream);
    return;
  }
  switch (a.cell.sat_func) {
    case kSatVG:
      if (a.cell.all_vg_half && !ifc) launch_cols_mode<kSatVGHalf, 0>(mode, a, grid, stream);
      else ifc ? launch_cols_mode<kSatVG, 1>(mode, a, grid, stream) : launch_cols_mode<kSatVG, 0>(mode, a, grid, stream);
      break;
    case kSatBC: ifc ? launch_cols_mode<kSatBC, 1>(mode, a, grid, stream) : launch_cols_mode<kSatBC, 0>(mode, a, grid, stream); break;
    default: ifc ? launch_cols_mode<kSatSBC, 1>(mode, a, grid, stream) : launch_cols_mode<kSatSBC, 0>(mode, a, grid, stream); break;
  }
}

void launch_assemble_general(int mode, const GenArgs &a, hipStream_t stream) {
  if (a.n <= 0) return;
  dim3 grid(nblk(a.n));
  if (a.ell_enc && a.ell_k <= 8) {
    if (mode == 0) LAUNCH(k_auxvars<0>, grid, a); else if (mode == 1) LAUNCH(k_auxvars<1>, grid, a); else LAUNCH(k_auxvars<2>, grid, a);
    if (a.ell_k <= 6) {
      if (mode == 0) LAUNCH((k_flux_rows_ell<0, 6>), grid, a); else if (mode == 1) LAUNCH((k_flux_rows_ell<1, 6>), grid, a);
      else LAUNCH((k_flux_rows_ell<2, 6>), grid, a);
    } else {
      if (mode == 0) LAUNCH((k_flux_rows_ell<0, 8>), grid, a); else if (mode == 1) LAUNCH((k_flux_rows_ell<1, 8>), grid, a);
      else LAUNCH((k_flux_rows_ell<2, 8>), grid, a);
    }
    return;
  }
  if (mode == 0) { LAUNCH(k_auxvars<0>, grid, a); LAUNCH(k_flux_rows<0>, grid, a); }
  else if (mode == 1) { LAUNCH(k_auxvars<1>, grid, a); LAUNCH(k_flux_rows<1>, grid, a); }
  else { LAUNCH(k_auxvars<2>, grid, a); LAUNCH(k_flux_rows<2>, grid, a); }
}

// the lateral entries of the K-wide ELL incidence lists, compacted into KL-wide ones (order kept); *overflow: a row has more than KL
__global__ __launch_bounds__(kBlock) void k_ell_lateral(int n, int K, const int *__restrict__ enc, const int *__restrict__ oth,
                                                        const int *__restrict__ slot, int KL, int *__restrict__ lenc, int *__restrict__ loth,
                                                        int *__restrict__ lslot, int *overflow) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x, N = (size_t)n;
  if (i >= N) return;
  int j = 0;
  for (int k = 0; k < K; ++k) {
    const int e = enc[k * N + i];
    if (e < 0 || (e >> 1) < n) continue;   // absent, or a chain link (connection index below n: the column march's)
    if (j < KL) { lenc[j * N + i] = e; loth[j * N + i] = oth[k * N + i]; lslot[j * N + i] = slot[k * N + i]; }
    ++j;
  }
  if (j > KL) *overflow = 1;
  for (; j < KL; ++j) { lenc[j * N + i] = -1; loth[j * N + i] = 0; lslot[j * N + i] = -1; }
}
void launch_ell_lateral(int n, int K, const int *enc, const int *oth, const int *slot, int KL, int *lenc, int *loth, int *lslot, int *overflow,
                        hipStream_t stream) {
  if (n > 0) LAUNCH(k_ell_lateral, dim3(nblk(n)), n, K, enc, oth, slot, KL, lenc, loth, lslot, overflow);
}

void launch_assemble_cols_lateral(int mode, const GenArgs &a, hipStream_t stream) {
  if (a.n <= 0) return;
  if (!a.ell_enc || a.ell_k > 8) throw std::runtime_error("internal: the lateral pass needs the ELL incidence lists");
  dim3 grid(nblk(a.n));
  if (a.ell_k <= 4) {   // (the compacted lateral lists: four gathers per row instead of six, fewer registers)
    if (mode == 0) LAUNCH((k_flux_rows_ell<0, 4, true>), grid, a); else if (mode == 1) LAUNCH((k_flux_rows_ell<1, 4, true>), grid, a);
    else LAUNCH((k_flux_rows_ell<2, 4, true>), grid, a);
  } else if (a.ell_k <= 6) {
    if (mode == 0) LAUNCH((k_flux_rows_ell<0, 6, true>), grid, a); else if (mode == 1) LAUNCH((k_flux_rows_ell<1, 6, true>), grid, a);
    else LAUNCH((k_flux_rows_ell<2, 6, true>), grid, a);
  } else {
    if (mode == 0) LAUNCH((k_flux_rows_ell<0, 8, true>), grid, a); else if (mode == 1) LAUNCH((k_flux_rows_ell<1, 8, true>), grid, a);
    else LAUNCH((k_flux_rows_ell<2, 8, true>), grid, a);
  }
}

void launch_bc(int mode, const BcArgs &a, hipStream_t stream) {
  if (a.ncond_cells <= 0) return;
  dim3 grid(nblk(a.ncond_cells));
  if (mode == 0) LAUNCH(k_bc<0>, grid, a);
  else if (mode == 1) LAUNCH(k_bc<1>, grid, a);
  else if (mode == 2) LAUNCH(k_bc<2>, grid, a);
  else LAUNCH(k_bc<4>, grid, a);
}

void launch_sinks(int mode, const SinkArgs &a, hipStream_t stream) {
  if (a.ncells <= 0) return;
  dim3 grid(nblk(a.ncells));
  if (mode == 0) LAUNCH(k_sinks<0>, grid, a);
  else if (mode == 1) LAUNCH(k_sinks<1>, grid, a);
  else if (mode == 2) LAUNCH(k_sinks<2>, grid, a);
  else LAUNCH(k_sinks<4>, grid, a);
}

void launch_conn_flux(const CellArrays &cell, const ConnArrays &conn, const int *cup, const int *cdn, int nc,
                      double *flux_out, hipStream_t stream) {
  if (nc > 0) LAUNCH(k_conn_flux, dim3(nblk(nc)), cell, conn, cup, cdn, nc, flux_out);
}
void launch_cell_var(const CellArrays &cell, int n, int var_type, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_cell_var, dim3(nblk(n)), cell, n, var_type, out);
}

void launch_spmv_compact(int tpw, int n, int max_row_nnz, int split, const int *tileptr, const unsigned short *roff,
                         const short *dcol, const double *val, const double *x, double *y, hipStream_t stream) {
  if (n <= 0) return;
  const long long tiles = ((long long)n + 63) / 64, waves = (tiles + tpw - 1) / tpw;
  const dim3 grid((int)((waves + 3) / 4));
#define SPMV_C(JJ, SP) LAUNCH((k_spmv_c<JJ, SP, false>), grid, n, tpw, split, tileptr, roff, dcol, val, x, y, nullptr, nullptr, nullptr)
  if (split > 0) { if (max_row_nnz <= 7) SPMV_C(7, true); else SPMV_C(8, true); }
  else if (max_row_nnz <= 3) SPMV_C(3, false);
  else if (max_row_nnz <= 7) SPMV_C(7, false);
  else SPMV_C(8, false);
#undef SPMV_C
}

void launch_spmv_multibase(int tpw, int n, int max_row_nnz, const int *tileptr, const unsigned short *roff,
                           const unsigned short *mcol, const int *tbase, const double *val, const double *x, double *y,
                           hipStream_t stream) {
  if (n <= 0) return;
  const long long tiles = ((long long)n + 63) / 64, waves = (tiles + tpw - 1) / tpw;
  const dim3 grid((int)((waves + 3) / 4));
  if (max_row_nnz <= 3) LAUNCH(k_spmv_pipe_mb<3>, grid, n, tpw, tileptr, roff, mcol, tbase, val, x, y);
  else if (max_row_nnz <= 7) LAUNCH(k_spmv_pipe_mb<7>, grid, n, tpw, tileptr, roff, mcol, tbase, val, x, y);
  else LAUNCH(k_spmv_pipe_mb<8>, grid, n, tpw, tileptr, roff, mcol, tbase, val, x, y);
}

int spmv_compact_dots_blocks(int tpw, int n) {
  const long long tiles = ((long long)n + 63) / 64, waves = (tiles + tpw - 1) / tpw;
  return (int)((waves + 3) / 4);
}
int spmv_cols_dots_blocks(int n) { return (int)(((long long)n + (long long)kBlock * kColsDotsChunks - 1) / ((long long)kBlock * kColsDotsChunks)); }
int launch_spmv_cols_dots(int n, int ncols, int nz, const double *val, const double *x, const double *f, const double *q, double *partials,
                          hipStream_t stream) {
  if (n <= 0) return 0;
  const dim3 grid((unsigned)spmv_cols_dots_blocks(n));
  LAUNCH(k_spmv_cols_dots, grid, n, ncols, nz, val, x, f, q, partials);
  return (int)grid.x;
}
int launch_spmv_compact_dots(int tpw, int n, int max_row_nnz, int split, const int *tileptr, const unsigned short *roff,
                             const short *dcol, const double *val, const double *x, const double *f, const double *xx,
                             double *partials, hipStream_t stream) {
  const int nb = spmv_compact_dots_blocks(tpw, n);
  if (n <= 0) return 0;
  const dim3 grid(nb);
#define SPMV_CD(JJ, SP) LAUNCH((k_spmv_c<JJ, SP, true>), grid, n, tpw, split, tileptr, roff, dcol, val, x, nullptr, f, xx, partials)
  if (split > 0) { if (max_row_nnz <= 7) SPMV_CD(7, true); else SPMV_CD(8, true); }
  else if (max_row_nnz <= 3) SPMV_CD(3, false);
  else if (max_row_nnz <= 7) SPMV_CD(7, false);
  else SPMV_CD(8, false);
#undef SPMV_CD
  return nb;
}

void launch_spmv(int variant, int n, int max_row_nnz, const int *rowptr, const int *col, const double *val,
                 const double *x, double *y, hipStream_t stream) {
  if (n <= 0) return;
  if (variant == 1) LAUNCH(k_spmv_stream, dim3(nblk(n)), n, rowptr, col, val, x, y);
  else if (variant == 17 && max_row_nnz <= 8) {
    const int tpw = 4;
    const long long tiles = ((long long)n + 63) / 64, waves = (tiles + tpw - 1) / tpw;
    const dim3 grid((int)((waves + 3) / 4));
    if (max_row_nnz <= 3) LAUNCH(k_spmv_pipe<3>, grid, n, tpw, rowptr, col, val, x, y);
    else if (max_row_nnz <= 7) LAUNCH(k_spmv_pipe<7>, grid, n, tpw, rowptr, col, val, x, y);
    else LAUNCH(k_spmv_pipe<8>, grid, n, tpw, rowptr, col, val, x, y);
  }
  else LAUNCH(k_spmv_row, dim3(nblk(n)), n, rowptr, col, val, x, y);   // variant 0, and rows of more than 8 entries
}

void launch_tri_factor(const TriArgs &a, hipStream_t stream) {
  if (a.ncols > 0) LAUNCH(k_tri_factor, dim3(nblk(a.ncols)), a);
}
void launch_chain_tri_factor(const ColChainArgs &a, hipStream_t stream) {
  if (a.nchains > 0) LAUNCH(k_chain_tri_factor, dim3(nblk(a.nchains)), a);
}
void launch_chain_tri_solve(const ColChainArgs &a, const double *r, double *z, hipStream_t stream) {
  if (a.nchains > 0) LAUNCH(k_chain_tri_solve, dim3(nblk(a.nchains)), a, r, z);
}
void launch_tri_solve(const TriArgs &a, const double *r, double *z, hipStream_t stream) {
  if (a.ncols > 0) LAUNCH(k_tri_solve, dim3(nblk(a.ncols)), a, r, z);
}
int launch_tri_solve_dots(const TriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream) {
  const dim3 grid(nblk(a.ncols));
  if (w) LAUNCH(k_tri_solve_dots<2>, grid, a, r, z, w, partials);
  else LAUNCH(k_tri_solve_dots<1>, grid, a, r, z, w, partials);
  return (int)grid.x;
}
int tri_solve_lds_blocks(int ncols) { return (ncols + 63) / 64; }
int launch_tri_solve_lds(const TriArgs &a, const double *r, double *z, const double *w, double *partials, double *scratch64,
                         int group, hipStream_t stream) {
  const dim3 grid(tri_solve_lds_blocks(a.ncols));
  const size_t shmem = (size_t)a.nz * 64 * sizeof(double);
#define TRI_LDS(ND, G, RF, RB) hipLaunchKernelGGL((k_tri_solve_lds<ND, G, RF, RB, 0>), grid, dim3(64), shmem, stream, a, r, z, w, partials, scratch64, (double *)nullptr, (const double *)nullptr, (const double *)nullptr, Fold(), (const double *)nullptr)
  // `group` = levels in flight per stream: rings of 4-level groups
  if (w) { if (group <= 8) TRI_LDS(2, 4, 2, 2); else if (group <= 16) TRI_LDS(2, 4, 4, 3); else TRI_LDS(2, 4, 6, 4); }
  else { if (group <= 8) TRI_LDS(1, 4, 2, 2); else if (group <= 16) TRI_LDS(1, 4, 4, 4); else TRI_LDS(1, 4, 6, 5); }
#undef TRI_LDS
  return (int)grid.x;
}
int launch_spmv_tri_lds(const TriArgs &a, const double *x, double *z, const double *w, double *partials, double *scratch64, hipStream_t stream,
                        const Fold &fold) {
  const dim3 grid(tri_solve_lds_blocks(a.ncols));
  const size_t shmem = (size_t)a.nz * 64 * sizeof(double);
  // forward: five streams (x, three matrix entries, l) in a ring of 3 groups of 4 levels = 60 loads in flight (the counter holds 63)
  if (w) hipLaunchKernelGGL((k_tri_solve_lds<2, 4, 3, 4, 1>), grid, dim3(64), shmem, stream, a, x, z, w, partials, scratch64, (double *)nullptr, (const double *)nullptr, (const double *)nullptr, fold, (const double *)nullptr);
  else hipLaunchKernelGGL((k_tri_solve_lds<1, 4, 3, 5, 1>), grid, dim3(64), shmem, stream, a, x, z, w, partials, scratch64, (double *)nullptr, (const double *)nullptr, (const double *)nullptr, fold, (const double *)nullptr);
  return (int)grid.x;
}
int launch_back_spmv_tri_lds(const TriArgs &a, const double *d, double *x, double *z, double *partials, double *scratch64, hipStream_t stream,
                             const Fold &fold, const double *fdot) {
  const dim3 grid(tri_solve_lds_blocks(a.ncols));
  const size_t shmem = (size_t)a.nz * 64 * sizeof(double);
  // (round 5: back sweeps of 6 and 7 groups -- they carry one stream less since plane c holds up / pivot -- measured equal to 4:
  // profiles/r05/lu_product_form_ab.json)
  if (fdot) hipLaunchKernelGGL((k_tri_solve_lds<2, 4, 3, 4, 2, true>), grid, dim3(64), shmem, stream, a, d, z, (const double *)x, partials, scratch64, x, (const double *)nullptr, (const double *)nullptr, fold, fdot);
  else hipLaunchKernelGGL((k_tri_solve_lds<2, 4, 3, 4, 2>), grid, dim3(64), shmem, stream, a, d, z, (const double *)x, partials, scratch64, x, (const double *)nullptr, (const double *)nullptr, fold, (const double *)nullptr);
  return (int)grid.x;
}
int launch_waxpy_spmv_tri_lds(const TriArgs &a, const double *r, const double *v, const double *sc, double *s, double *z, double *partials,
                              double *scratch64, hipStream_t stream, const Fold &fold, const double *fdot) {
  const dim3 grid(tri_solve_lds_blocks(a.ncols));
  const size_t shmem = (size_t)a.nz * 64 * sizeof(double);
  // forward: six streams (r, v, three matrix entries, l) in a ring of 2 groups of 5 levels = 60 loads in flight (fdot: one more stream, 60 loads)
  if (fdot) hipLaunchKernelGGL((k_tri_solve_lds<2, 5, 2, 4, 3, true>), grid, dim3(64), shmem, stream, a, r, z, (const double *)s, partials, scratch64, s, v, sc, fold, fdot);
  else hipLaunchKernelGGL((k_tri_solve_lds<2, 5, 2, 4, 3>), grid, dim3(64), shmem, stream, a, r, z, (const double *)s, partials, scratch64, s, v, sc, fold, (const double *)nullptr);
  return (int)grid.x;
}
int launch_tri_back_dots(const TriArgs &a, const double *d, double *z, double *partials, hipStream_t stream) {
  const dim3 grid(nblk(a.ncols));
  LAUNCH(k_tri_back_dots, grid, a, d, z, partials);
  return (a.ncols + 63) / 64;
}
void launch_jacobi_setup(int n, const int *diag_slot, const double *val, double *invd, hipStream_t stream) {
  if (n > 0) LAUNCH(k_jacobi_setup, dim3(nblk(n)), n, diag_slot, val, invd);
}

void launch_ilu_factor_program(const long long *ptr, const int *prog, const int *rows, const int *diag_slot, int lo, int hi,
                               double *lu, double *idiag, hipStream_t stream) {
  if (hi > lo) LAUNCH(k_ilu_factor_program, dim3(nblk(hi - lo)), ptr, prog, rows, diag_slot, lo, hi, lu, idiag);
}
void launch_ilu_repack(long long total, const int *slot, const double *lu, double *out, hipStream_t stream) {
  if (total > 0) LAUNCH(k_ilu_repack, dim3(nblk(total)), total, slot, lu, out);
}
void launch_ell_chain(const ChainArgs &a, hipStream_t stream) {
  if (a.ntiles <= 0) return;
  const dim3 grid((a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block);
  const bool up = a.idiag != nullptr;
  switch (a.K) {
    case 1: if (up) LAUNCH((k_ell_chain<1, true>), grid, a); else LAUNCH((k_ell_chain<1, false>), grid, a); break;
    case 2: if (up) LAUNCH((k_ell_chain<2, true>), grid, a); else LAUNCH((k_ell_chain<2, false>), grid, a); break;
    case 3: if (up) LAUNCH((k_ell_chain<3, true>), grid, a); else LAUNCH((k_ell_chain<3, false>), grid, a); break;
    default: if (up) LAUNCH((k_ell_chain<0, true>), grid, a); else LAUNCH((k_ell_chain<0, false>), grid, a);
  }
}
void launch_ell_lower_level(int n, int K, const int *col, const double *val, int lo, int hi, double *y, hipStream_t stream) {
  if (hi <= lo) return;
  const dim3 grid(nblk(hi - lo));
  switch (K) {
    case 1: LAUNCH(k_ell_lower<1>, grid, n, K, col, val, lo, hi, y); break;
    case 2: LAUNCH(k_ell_lower<2>, grid, n, K, col, val, lo, hi, y); break;
    case 3: LAUNCH(k_ell_lower<3>, grid, n, K, col, val, lo, hi, y); break;
    default: LAUNCH(k_ell_lower<0>, grid, n, K, col, val, lo, hi, y);
  }
}
void launch_ell_upper_level(int n, int K, const int *col, const double *val, const double *idiag, int lo, int hi, double *x,
                            hipStream_t stream) {
  if (hi <= lo) return;
  const dim3 grid(nblk(hi - lo));
  switch (K) {
    case 1: LAUNCH(k_ell_upper<1>, grid, n, K, col, val, idiag, lo, hi, x); break;
    case 2: LAUNCH(k_ell_upper<2>, grid, n, K, col, val, idiag, lo, hi, x); break;
    case 3: LAUNCH(k_ell_upper<3>, grid, n, K, col, val, idiag, lo, hi, x); break;
    default: LAUNCH(k_ell_upper<0>, grid, n, K, col, val, idiag, lo, hi, x);
  }
}
void launch_thermal_aux(const ThermalArgs &a, hipStream_t stream) {
  if (a.n > 0) LAUNCH(k_thermal_aux, dim3(nblk(a.n)), a);
}
void launch_thermal_rows(int mode, const ThermalArgs &a, hipStream_t stream) {
  if (a.n <= 0) return;
  if (mode == 0) LAUNCH(k_thermal_rows<0>, dim3(nblk(a.n)), a);
  else if (mode == 1) LAUNCH(k_thermal_rows<1>, dim3(nblk(a.n)), a);
  else LAUNCH(k_thermal_rows<2>, dim3(nblk(a.n)), a);
}
void launch_thermal_bc(int mode, const ThermalBcArgs &a, hipStream_t stream) {
  if (a.ncond_cells <= 0) return;
  if (mode == 0) LAUNCH(k_thermal_bc<0>, dim3(nblk(a.ncond_cells)), a);
  else if (mode == 1) LAUNCH(k_thermal_bc<1>, dim3(nblk(a.ncond_cells)), a);
  else LAUNCH(k_thermal_bc<2>, dim3(nblk(a.ncond_cells)), a);
}

void launch_copy(int n, const double *x, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_copy, dim3(vec_grid(n)), n, x, y);
}
void launch_set(int n, double v, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_set, dim3(vec_grid(n)), n, v, y);
}
void launch_box_fill(long long nslots, const BoxGrid &g, const int *cellmap, const int *int2ext, const int *rowptr, const int *col, int nown,
                     const double *lu, const double *idiag, double *const L[3], double *const U[3], double *Di, hipStream_t stream) {
  if (nslots > 0) LAUNCH(k_box_fill, dim3(nblk(nslots)), nslots, g, cellmap, int2ext, rowptr, col, nown, lu, idiag, L[0], L[1], L[2], U[0], U[1], U[2], Di);
}
void launch_box_load(long long nslots, const BoxGrid &g, const int *cellmap, const int *int2ext, const int *rowptr, const int *col, int nown,
                     const double *val, double *const L[3], double *const U[3], double *Di, double *const T[3], hipStream_t stream) {
  if (nslots > 0) LAUNCH(k_box_load, dim3(nblk(nslots)), nslots, g, cellmap, int2ext, rowptr, col, nown, val, L[0], L[1], L[2], U[0], U[1], U[2], Di,
                         T[0], T[1], T[2]);
}
void launch_box_factor(const BoxGrid &g, const int *boxes, int nboxes, double *const L[3], const double *const T[3], double *Di, double *scratch,
                       hipStream_t stream) {
  const size_t lds = (size_t)box_face_doubles(g) * sizeof(double);
  if (nboxes > 0) hipLaunchKernelGGL((k_box_factor<kBoxRingFactor>), dim3(nboxes), dim3(64), lds, stream, g, boxes, L[0], L[1], L[2], T[0], T[1], T[2],
                                     Di, scratch);
}
static size_t box_img_bytes(const BoxGrid &g) { return (size_t)g.S * 64 * sizeof(double); }
bool box_gather_in_fits(const BoxGrid &g) { return box_img_bytes(g) <= 48 * 1024; }
void launch_box_gather_out(const BoxGrid &g, const int *ext2int, const double *v, double *z, hipStream_t stream) {
  const long long nboxes = (long long)g.NB1 * g.NB2 * g.NB3;
  if (nboxes > 0) hipLaunchKernelGGL(k_box_gather_out, dim3((unsigned)nboxes), dim3(kBlock), box_img_bytes(g), stream, g, ext2int, v, z);
}
void launch_box_gather_in(const BoxGrid &g, const int *ext2int, const double *r, double *v, hipStream_t stream) {
  const long long nboxes = (long long)g.NB1 * g.NB2 * g.NB3;
  if (nboxes > 0) hipLaunchKernelGGL(k_box_gather_in, dim3((unsigned)nboxes), dim3(kBlock), box_img_bytes(g), stream, g, ext2int, r, v);
}
void launch_gather_or_zero(long long n, const int *idx, const double *in, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_gather_or_zero, dim3(nblk(n)), n, idx, in, out);
}
void launch_box_lower(const BoxGrid &g, const int *boxes, int nboxes, const double *const L[3], double *y, double *scratch, hipStream_t stream) {
  const size_t lds = (size_t)box_face_doubles(g) * sizeof(double);
  if (nboxes > 0) hipLaunchKernelGGL((k_box_lower<kBoxRing>), dim3(nboxes), dim3(64), lds, stream, g, boxes, L[0], L[1], L[2], y, scratch);
}
void launch_box_upper(const BoxGrid &g, const int *boxes, int nboxes, const double *const U[3], const double *Di, double *x, double *scratch,
                      hipStream_t stream) {
  const size_t lds = (size_t)box_face_doubles(g) * sizeof(double);
  if (nboxes > 0) hipLaunchKernelGGL((k_box_upper<kBoxRingUpper>), dim3(nboxes), dim3(64), lds, stream, g, boxes, U[0], U[1], U[2], Di, x, scratch);
}
void launch_box_solve_queue(const BoxGrid &g, const BoxQueueArgs &lo, const BoxQueueArgs &up, const double *const L[3], const double *const U[3],
                            const double *Di, double *v, double *scratch, hipStream_t stream) {
  const size_t lds = (size_t)box_face_doubles(g) * sizeof(double);
  if (lo.nboxes <= 0) return;
  BoxQueue ql, qu;
  ql.order = lo.order; ql.nboxes = lo.nboxes; ql.flags = lo.flags; ql.epoch = lo.epoch; ql.ticket = lo.ticket; ql.ticket_base = lo.ticket_base;
  ql.max_spins = lo.max_spins; ql.sc = lo.sc;
  qu.order = up.order; qu.nboxes = up.nboxes; qu.flags = up.flags; qu.epoch = up.epoch; qu.ticket = up.ticket; qu.ticket_base = up.ticket_base;
  qu.max_spins = up.max_spins; qu.sc = up.sc;
  hipLaunchKernelGGL((k_box_lower_queue<kBoxRing>), dim3(lo.nworkers), dim3(64), lds, stream, g, ql, L[0], L[1], L[2], v, scratch);
  hipLaunchKernelGGL((k_box_upper_queue<kBoxRingUpper>), dim3(up.nworkers), dim3(64), lds, stream, g, qu, U[0], U[1], U[2], Di, v, scratch);
}
bool box_stream_applies(const BoxGrid &g) { return g.B2 == 8 && g.B3 == 8 && g.S % 8 == 0 && g.S >= 16; }
size_t box_stream_face_doubles(const BoxGrid &g) { return (size_t)g.NB1 * g.NB2 * g.NB3 * stream_lines(g) * 8; }
void launch_box_solve_stream(const BoxGrid &g, const BoxStreamArgs &a, const double *const L[3], const double *const U[3], const double *Di,
                             double *v, double *scratch, hipStream_t stream, const double *r, double *z, const int *cellmap) {
#ifndef MPP_BOX_STREAM_DP
#define MPP_BOX_STREAM_DP 8
#endif
  constexpr int DPS = MPP_BOX_STREAM_DP;   // row pairs of operands in flight per wave (a multiple of 4)
  if (a.npencils <= 0) return;
  const size_t one = box_stream_face_doubles(g);
  StreamArgs lo{a.order_lower, a.npencils, a.ticket, a.ticket_base[0], a.max_spins, a.sc, a.faces, a.faces + one};
  StreamArgs up{a.order_upper, a.npencils, a.ticket + 1, a.ticket_base[1], a.max_spins, a.sc, a.faces + 2 * one, a.faces + 3 * one};
  const size_t lds_lo = (size_t)2 * (g.S + 2 * DPS + 8) * 8 * sizeof(double), lds_up = (size_t)2 * (g.S + 2 * DPS + 8) * 8 * sizeof(double);
  if (r) hipLaunchKernelGGL((k_box_lower_stream<DPS, true>), dim3(a.nworkers), dim3(64), lds_lo, stream, g, lo, L[0], L[1], L[2], v, scratch, r, cellmap);
  else hipLaunchKernelGGL((k_box_lower_stream<DPS, false>), dim3(a.nworkers), dim3(64), lds_lo, stream, g, lo, L[0], L[1], L[2], v, scratch, r, cellmap);
  if (z) hipLaunchKernelGGL((k_box_upper_stream<DPS, true>), dim3(a.nworkers), dim3(64), lds_up, stream, g, up, U[0], U[1], U[2], Di, v, scratch, z, cellmap);
  else hipLaunchKernelGGL((k_box_upper_stream<DPS, false>), dim3(a.nworkers), dim3(64), lds_up, stream, g, up, U[0], U[1], U[2], Di, v, scratch, z, cellmap);
}
void launch_gather(int n, const int *idx, const double *in, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_gather, dim3(nblk(n)), n, idx, in, out);
}
void launch_gather_flags(int n, const int *idx, const int *in, signed char *out, int zero_is_one, hipStream_t stream) {
  if (n > 0) LAUNCH(k_gather_flags, dim3(nblk(n)), n, idx, in, out, zero_is_one);
}
void launch_scatter_mapped(int n, const int *idx, const int *map, const int *in, int *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_scatter_mapped, dim3(nblk(n)), n, idx, map, in, out);
}
void launch_scatter(int n, const int *idx, const double *in, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_scatter, dim3(nblk(n)), n, idx, in, out);
}
void launch_pointwise_mult(int n, const double *a, const double *x, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_pmult, dim3(vec_grid(n)), n, a, x, y);
}
void launch_waxpy(int n, double *w, const double *sc, int slot, double sign, double aval, const double *x,
                  const double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_waxpy, dim3(vec_grid(n)), n, w, sc, slot, sign, aval, x, y);
}
void launch_bcgs_p(int n, double *P, const double *Pold, const double *R, const double *V, const double *sc,
                   hipStream_t stream) {
  if (n > 0) LAUNCH(k_bcgs_p, dim3(vec_grid(n)), n, P, Pold, R, V, sc);
}
int bcgs_xr_blocks(int n, bool wide) { return wide ? vec_grid(n) : red_grid(n); }
int launch_bcgs_xr(int n, bool first, bool wide, double *X, double *R, const double *P, const double *S, const double *T,
                   const double *RP, const double *sc, double *partials, hipStream_t stream, bool xx) {
  // a kernel that writes two vectors wants one pair per lane (a capped grid streams ~20 % slower, tools/membench.hip);
  // its many partial sums go through the two-stage finalisation
  const int g = wide ? vec_grid(n) : red_grid(n);
  if (first && xx) LAUNCH((k_bcgs_xr<true, true>), dim3(g), n, X, R, P, S, T, RP, sc, partials);
  else if (first) LAUNCH(k_bcgs_xr<true>, dim3(g), n, X, R, P, S, T, RP, sc, partials);
  else LAUNCH(k_bcgs_xr<false>, dim3(g), n, X, R, P, S, T, RP, sc, partials);
  return g;
}
int launch_dots(const DotArgs &a, double *partials, hipStream_t stream) {
  const int g = red_grid(a.n);
  const double *const *x = a.x, *const *y = a.y;
  // pointer-pattern dispatch: each distinct vector is read once
  if (a.nvec == 1 && x[0] == y[0]) LAUNCH(k_dots_pat<0>, dim3(g), a.n, x[0], x[0], x[0], x[0], partials);
  else if (a.nvec == 1) LAUNCH(k_dots_pat<1>, dim3(g), a.n, x[0], y[0], x[0], x[0], partials);
  else if (a.nvec == 2 && x[1] == y[1] && y[0] == x[1] && x[0] != y[0]) LAUNCH(k_dots_pat<2>, dim3(g), a.n, x[0], y[0], x[0], x[0], partials);
  else if (a.nvec == 2 && x[0] == y[0] && x[1] == y[1]) LAUNCH(k_dots_pat<4>, dim3(g), a.n, x[0], x[1], x[0], x[0], partials);
  else if (a.nvec == 3 && x[0] == y[0] && x[1] == y[1]) LAUNCH(k_dots_pat<3>, dim3(g), a.n, x[0], x[1], x[2], y[2], partials);
  else {
    bool same_x = true;
    for (int j = 1; j < a.nvec; ++j) same_x &= (x[j] == x[0]);
    (void)same_x;  // k_mdot reads a.x[j0] once per group; correct for any x, optimal when all x are equal
    for (int j0 = 0; j0 < a.nvec; j0 += kMdotGroup) {
      const int nv = a.nvec - j0 < kMdotGroup ? a.nvec - j0 : kMdotGroup;
      bool grp_same = true;
      for (int j = 1; j < nv; ++j) grp_same &= (x[j0 + j] == x[j0]);
      if (grp_same) LAUNCH(k_mdot, dim3(g), a, j0, nv, partials);
      else for (int j = 0; j < nv; ++j) LAUNCH(k_mdot, dim3(g), a, j0 + j, 1, partials);
    }
  }
  return g;
}
constexpr int kFinalizeChunk = 2048;
// up to two chunks' worth goes through one block per vector (the sums' order is then that of a single pass)
int finalize_chunks(int nblocks) { return nblocks <= 2 * kFinalizeChunk ? 1 : (nblocks + kFinalizeChunk - 1) / kFinalizeChunk; }
void launch_finalize(int nblocks, int nvec, const double *partials, double *sums, const Finalize &f, hipStream_t stream) {
  if (nvec <= 0) return;
  const int nchunk = finalize_chunks(nblocks);
  LAUNCH(k_finalize, dim3(nchunk, nvec), nblocks, nchunk == 1 ? nblocks : kFinalizeChunk, partials, f.stage, sums, f.counter, f.sc, f.step,
         Publish{f.pub_host, f.pub_seq, f.pub_value});
}
int maxpy_blocks(int n) { return nblk(((long long)n + 1) / 2); }   // two rows per thread
int launch_maxpy_neg(const MaxpyArgs &a, const double *h, double *y, double *partials, hipStream_t stream) {
  if (a.n <= 0 || a.nvec <= 0) return 0;
  for (int j = 0; j < a.nvec; ++j)   // (16-byte accesses)
    if (((uintptr_t)a.v[j] | (uintptr_t)y) & 15) throw std::runtime_error("internal: multi-axpy operands must be 16-byte aligned");
  const dim3 grid(maxpy_blocks(a.n));
  LAUNCH(k_maxpy_neg, grid, a, h, y, partials);
  return (int)grid.x;
}
void launch_maxpy_val(const MaxpyValArgs &a, double *x, hipStream_t stream) {
  if (a.n > 0 && a.nvec > 0) LAUNCH(k_maxpy_val, dim3(nblk(a.n)), a, x);
}
void launch_scale(int n, double *x, const double *sc, int slot_inv_of, hipStream_t stream) {
  if (n > 0) LAUNCH(k_scale_inv, dim3(nblk(n)), n, x, sc, slot_inv_of);
}
void launch_bcgs_scalars(int step, double *sc, hipStream_t stream) {
  hipLaunchKernelGGL(k_bcgs_scalars, dim3(1), dim3(64), 0, stream, step, sc, Publish{nullptr, nullptr, 0ull});
}
void launch_publish(int step, double *sc, double *host, unsigned long long *seq, unsigned long long value, hipStream_t stream) {
  hipLaunchKernelGGL(k_bcgs_scalars, dim3(1), dim3(128), 0, stream, step, sc, Publish{host, seq, value});
}

}  // namespace mpp
