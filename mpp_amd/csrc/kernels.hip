// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
// All of these are HBM-bound fp64/int32 streaming kernels (arithmetic intensity 0.1-0.5 flop/B, far below the
// fp64 ridge), so the design rules are: coalesced 8-byte-per-lane streams in struct-of-arrays layout, grids of
// >= 8 blocks per CU, no atomics, deterministic two-stage reductions with wavefront shuffles.  MFMA is not used:
// no dense block contraction appears anywhere on this path.
//
// Internal cell ordering is level-major (cell (k,c) of column c, level k sits at k*ncols + c), so that the 64
// lanes of a wave work on 64 neighbouring soil columns at the same depth: every load/store of a column march is
// a contiguous 512-byte transaction and the vertical neighbours of a row are exactly +-ncols away.
#include "kernels.hpp"

namespace mpp {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ Soil load_soil(const CellArrays &ca, size_t i) {
  Soil s;
  s.sat_res = ca.sat_res[i];
  s.alpha = ca.alpha[i];
  s.m = ca.m[i];
  s.por = ca.por[i];
  if (ca.sat_func == kSatSBC) {
    s.pu = ca.sbc_pu[i]; s.ps = ca.sbc_ps[i]; s.b2 = ca.sbc_b2[i]; s.b3 = ca.sbc_b3[i];
  } else {
    s.pu = s.ps = s.b2 = s.b3 = 0.0;
  }
  return s;
}

__device__ __forceinline__ bool is_active(const CellArrays &ca, size_t i) {
  return ca.active ? ca.active[i] != 0 : true;
}

// ------------------------------------------------------------------------------------------------
// accumulation term only (RichardsODEPressureAccum, GoveqnRichardsODEPressureType.F90:1603-1634) -- PreSolve
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accum(CellArrays ca, int n, double dtInv, double *out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  double acc = 0.0;
  if (is_active(ca, i)) {
    Soil s = load_soil(ca, i);
    Aux a = auxvar_compute<false>(ca.sat_func, ca.eos, s, ca.P[i], ca.frac_liq);
    acc = s.por * a.den * a.sat * ca.vol[i] * dtInv;
  }
  out[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// fused column march: auxvars + accumulation + vertical fluxes (+ derivatives) in one pass.
// Restates RichardsODEComputeResidual / ComputeJacobian (GoveqnRichardsODEPressureType.F90:388-453) for meshes
// whose internal connections are vertical only.  One thread owns levels [k0,k1) of one column and keeps the
// previous level's auxvars in registers, so every auxvar and every flux is evaluated once (plus one halo level
// per segment end when nseg > 1).  Sum order per row equals the reference's serial connection loop.
// ------------------------------------------------------------------------------------------------
template <int MODE>  // 0 residual, 1 Jacobian, 2 both
__global__ __launch_bounds__(kBlock) void k_assemble_cols(ColsArgs a) {
  const int *__restrict__ rowptr = a.rowptr;
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  const int nz = a.nz, ncols = a.ncols;
  const int seg = blockIdx.y;
  const int k0 = (int)((long long)nz * seg / a.nseg);
  const int k1 = (int)((long long)nz * (seg + 1) / a.nseg);
  if (k0 >= k1) return;
  const int kstart = k0 > 0 ? k0 - 1 : 0;
  const int kend = k1 < nz ? k1 + 1 : nz;
  const CellArrays &ca = a.cell;

  Aux prev = {};
  bool prev_act = false;
  double f_prev = 0.0, dconn_prev = 0.0, dacc_prev = 0.0, lo_prev = 0.0, up_prev = 0.0;

  for (int k = kstart; k < kend; ++k) {
    const size_t i = (size_t)k * ncols + c;
    const bool act = is_active(ca, i);
    const bool owned = (k >= k0 && k < k1);
    Aux cur = {};
    double f_cur = 0.0, dacc_cur = 1.0;  // inactive cell: diagonal 1 (GE:1682-1684)
    const double Pi = ca.P[i];
    cur.P = Pi;
    if (act) {
      Soil s = load_soil(ca, i);
      cur = auxvar_compute<JAC>(ca.sat_func, ca.eos, s, Pi, ca.frac_liq);
      if (owned) {
        const double vol = ca.vol[i];
        if (RES) f_cur = s.por * cur.den * cur.sat * vol * a.dtInv;
        if (JAC) dacc_cur = (s.por * cur.dden * cur.sat + s.por * cur.den * cur.dsat) * vol * a.dtInv;
      }
    }
    if (RES && owned) f_cur = f_cur - a.accum_prev[i];
    double dconn_cur = 0.0, lo_cur = 0.0;
    if (k > kstart) {
      if (prev_act && act) {
        const size_t ic = i - ncols;
        ConnGeom g;
        g.Dq = a.conn.Dq[ic]; g.upweight = a.conn.upw[ic]; g.dist_gravity = a.conn.dgrav[ic]; g.area = a.conn.area[ic];
        double Jup = 0.0, Jdn = 0.0;
        const double flux = richards_flux<JAC>(prev, cur, g, 0, Jup, Jdn);
        if (RES) { f_prev -= flux; f_cur += flux; }
        if (JAC) { dconn_prev += Jup; up_prev = Jdn; lo_cur = -Jup; dconn_cur += -Jdn; }
      }
      const int kp = k - 1;
      if (kp >= k0) {  // row kp is owned and now complete
        const size_t ip = i - ncols;
        if (RES) {
          double f = f_prev;
          if (a.src) f -= a.src[ip];
          a.F[ip] = f;
        }
        if (JAC) {
          int slot = rowptr[ip];
          if (kp > 0) a.val[slot++] = lo_prev;
          a.val[slot++] = dconn_prev + dacc_prev;
          a.val[slot] = up_prev;  // kp < nz-1 always holds here
        }
      }
    }
    prev = cur; prev_act = act;
    f_prev = f_cur; dconn_prev = dconn_cur; dacc_prev = dacc_cur; lo_prev = lo_cur; up_prev = 0.0;
  }
  if (kend == nz && k1 == nz) {  // top row of the column
    const size_t ip = (size_t)(nz - 1) * ncols + c;
    if (RES) {
      double f = f_prev;
      if (a.src) f -= a.src[ip];
      a.F[ip] = f;
    }
    if (JAC) {
      int slot = rowptr[ip];
      if (nz > 1) a.val[slot++] = lo_prev;
      a.val[slot] = dconn_prev + dacc_prev;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// general meshes, pass 1: auxvars of every cell (UpdateAuxVarsIntrn, GE:1417-1442) + accumulation term
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_auxvars(GenArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const CellArrays &ca = a.cell;
  double f = 0.0, dacc = 1.0;
  Aux x = {};
  if (is_active(ca, i)) {
    Soil s = load_soil(ca, i);
    x = auxvar_compute<JAC>(ca.sat_func, ca.eos, s, ca.P[i], ca.frac_liq);
    const double vol = ca.vol[i];
    if (RES) f = s.por * x.den * x.sat * vol * a.dtInv;
    if (JAC) dacc = (s.por * x.dden * x.sat + s.por * x.den * x.dsat) * vol * a.dtInv;
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
  if (is_active(ca, i)) {
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

// ------------------------------------------------------------------------------------------------
// boundary conditions (GE:1815-1877 residual, 2076-2139 Jacobian): the BC auxvar has the soil of its id_dn
// cell (MultiPhysicsProbVSFM.F90:422-446) and pressure = condition value for Dirichlet/seepage, = P_dn for
// mass-flux conditions (GE:1517-1531).
// MODE 0 residual (F += flux), 1 Jacobian (diag += -Jdn), 2 both, 4 flux output [kg/s]
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_bc(BcArgs a) {
  constexpr bool RES = (MODE == 0 || MODE == 2), JAC = (MODE == 1 || MODE == 2), OUT = (MODE == 4);
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= a.ncond_cells) return;
  const CellArrays &ca = a.cell;
  const size_t cell = a.cc_cell[j];
  if (!is_active(ca, cell)) return;
  Soil s = load_soil(ca, cell);
  const Aux in = auxvar_compute<JAC>(ca.sat_func, ca.eos, s, ca.P[cell], ca.frac_liq);
  double f = RES ? a.F[cell] : 0.0;
  double d = 0.0;
  for (int e = a.cc_ptr[j]; e < a.cc_ptr[j + 1]; ++e) {
    const int t = a.type[e];
    const int ctype = t & 0xffff, swap = t >> 16;
    const double pb = (ctype == kCondMassFlux) ? in.P : a.value[e];
    const Aux bc = auxvar_compute<JAC>(ca.sat_func, ca.eos, s, pb, ca.frac_liq);
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
  if (RES) a.F[cell] = f;
  if (JAC) a.val[a.diag_slot[cell]] += d;
}

__global__ __launch_bounds__(kBlock) void k_conn_flux(CellArrays ca, ConnArrays cn, const int *cup, const int *cdn,
                                                      int nc, double *out) {
  const int ic = blockIdx.x * kBlock + threadIdx.x;
  if (ic >= nc) return;
  const size_t u = cup[ic], d = cdn[ic];
  double flux = 0.0;
  if (is_active(ca, u) && is_active(ca, d)) {
    Soil su = load_soil(ca, u), sd = load_soil(ca, d);
    Aux au = auxvar_compute<false>(ca.sat_func, ca.eos, su, ca.P[u], ca.frac_liq);
    Aux ad = auxvar_compute<false>(ca.sat_func, ca.eos, sd, ca.P[d], ca.frac_liq);
    ConnGeom g;
    g.Dq = cn.Dq[ic]; g.upweight = cn.upw[ic]; g.dist_gravity = cn.dgrav[ic]; g.area = cn.area[ic];
    double j1, j2;
    flux = richards_flux<false>(au, ad, g, 0, j1, j2) * kFMWH2O;  // internal_flux = flux*FMWH2O (GE:1809)
  }
  out[ic] = flux;
}

// soe auxvar fields filled by RichardsODEPressureSetDataInSOEAuxVar (GE:1170-1195)
__global__ __launch_bounds__(kBlock) void k_cell_var(CellArrays ca, int n, int var_type, double *out) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  Soil s = load_soil(ca, i);
  const double P = ca.P[i];
  Aux a = auxvar_compute<false>(ca.sat_func, ca.eos, s, P, ca.frac_liq);
  double v;
  if (var_type == 608) v = a.sat;
  else if (var_type == 610) v = s.por * a.den * kFMWH2O * a.sat * ca.vol[i];
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

// ------------------------------------------------------------------------------------------------
// block-Jacobi preconditioner, one block per soil column = exact LU of the column's tridiagonal (Thomas).
// factor: l_k = a_k / b'_{k-1}, b'_k = b_k - l_k c_{k-1};  stored: l, 1/b', c   (level-major, coalesced)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tri_factor(TriArgs a) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  double inv_prev = 0.0, c_prev = 0.0;
#pragma unroll 4
  for (int k = 0; k < a.nz; ++k) {
    const size_t i = (size_t)k * a.ncols + c;
    const int sl = a.tri_lo[i], su = a.tri_up[i];
    const double lo = sl >= 0 ? a.val[sl] : 0.0;
    const double di = a.val[a.tri_di[i]];
    const double up = su >= 0 ? a.val[su] : 0.0;
    const double l = lo * inv_prev;
    const double piv = di - l * c_prev;
    const double inv = 1.0 / piv;
    a.l[i] = l; a.invb[i] = inv; a.c[i] = up;
    inv_prev = inv; c_prev = up;
  }
}

__global__ __launch_bounds__(kBlock) void k_tri_solve(TriArgs a, const double *__restrict__ r, double *__restrict__ z) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  double prev = 0.0;
#pragma unroll 4
  for (int k = 0; k < a.nz; ++k) {
    const size_t i = (size_t)k * a.ncols + c;
    const double d = r[i] - a.l[i] * prev;
    z[i] = d;
    prev = d;
  }
  double next = 0.0;
#pragma unroll 4
  for (int k = a.nz - 1; k >= 0; --k) {
    const size_t i = (size_t)k * a.ncols + c;
    const double xk = (z[i] - a.c[i] * next) * a.invb[i];
    z[i] = xk;
    next = xk;
  }
}

__global__ __launch_bounds__(kBlock) void k_jacobi_setup(int n, const int *diag_slot, const double *val, double *invd) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) invd[i] = 1.0 / val[diag_slot[i]];
}

// ------------------------------------------------------------------------------------------------
// vector kernels (grid-stride, 8 B per lane coalesced)
// ------------------------------------------------------------------------------------------------
inline int vec_grid(int n) {
  long long b = ((long long)n + kBlock - 1) / kBlock;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

__global__ __launch_bounds__(kBlock) void k_copy(int n, const double *x, double *y) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) y[i] = x[i];
}
__global__ __launch_bounds__(kBlock) void k_set(int n, double v, double *y) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) y[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_gather(int n, const int *idx, const double *in, double *out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(kBlock) void k_scatter(int n, const int *idx, const double *in, double *out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) out[idx[i]] = in[i];
}
__global__ __launch_bounds__(kBlock) void k_pmult(int n, const double *a, const double *x, double *y) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) y[i] = a[i] * x[i];
}
__global__ __launch_bounds__(kBlock) void k_waxpy(int n, double *w, const double *sc, int slot, double sign, double aval,
                                                  const double *x, const double *y) {
  const double a = slot >= 0 ? sign * sc[slot] : aval;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) w[i] = y[i] + a * x[i];
}
// p <- r - omegaold*beta*v + beta*p  (VecAXPBYPCZ(P,1.0,-omegaold*beta,beta,R,V), bcgs.c)
__global__ __launch_bounds__(kBlock) void k_bcgs_p(int n, double *P, const double *R, const double *V, const double *sc) {
  const double beta = sc[S_BETA], ob = -sc[S_OMEGAOLD] * beta;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock)
    P[i] = 1.0 * R[i] + ob * V[i] + beta * P[i];
}

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

inline int red_grid(int n) {
  long long b = ((long long)n + (long long)kBlock * 4 - 1) / ((long long)kBlock * 4);
  return (int)(b < 1 ? 1 : (b > kMaxPartialBlocks ? kMaxPartialBlocks : b));
}

// x <- alpha p + omega s + x ; r <- s - omega t ; partial sums of r.r   (bcgs.c)
__global__ __launch_bounds__(kBlock) void k_bcgs_xr(int n, double *X, double *R, const double *P, const double *S,
                                                    const double *T, const double *sc, double *partials) {
  __shared__ double lds[kBlock / 64];
  const double alpha = sc[S_ALPHA], omega = sc[S_OMEGA];
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) {
    const double s = S[i];
    X[i] = alpha * P[i] + omega * s + X[i];
    const double r = s - omega * T[i];
    R[i] = r;
    acc += r * r;
  }
  const double t = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <int NV>
__global__ __launch_bounds__(kBlock) void k_dots(DotArgs a, double *partials) {
  __shared__ double lds[kBlock / 64];
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * kBlock) {
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += a.x[j][i] * a.y[j][i];
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const double t = block_sum(acc[j], lds);
    if (threadIdx.x == 0) partials[(size_t)j * gridDim.x + blockIdx.x] = t;
  }
}
// any number of vectors (GMRES Gram-Schmidt): one pass per group of 4
__global__ __launch_bounds__(kBlock) void k_dots_dyn(DotArgs a, int j0, int nv, double *partials) {
  __shared__ double lds[kBlock / 64];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * kBlock) {
    for (int j = 0; j < nv; ++j) acc[j] += a.x[j0 + j][i] * a.y[j0 + j][i];
  }
  for (int j = 0; j < nv; ++j) {
    const double t = block_sum(acc[j], lds);
    if (threadIdx.x == 0) partials[(size_t)(j0 + j) * gridDim.x + blockIdx.x] = t;
  }
}

// second stage: block j sums partials[j*nb .. j*nb+nb) in a fixed order into sc[S_SUM0+j]
__global__ __launch_bounds__(kBlock) void k_finalize(int nb, const double *partials, double *sc) {
  __shared__ double lds[kBlock / 64];
  const int j = blockIdx.x;
  double acc = 0.0;
  for (int b = threadIdx.x; b < nb; b += kBlock) acc += partials[(size_t)j * nb + b];
  const double t = block_sum(acc, lds);
  if (threadIdx.x == 0) sc[S_SUM0 + j] = t;
}

__global__ void k_bcgs_scalars(int step, double *sc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (step == 0) {  // rhoold = alpha = omegaold = 1
    sc[S_RHOOLD] = 1.0; sc[S_ALPHA] = 1.0; sc[S_OMEGAOLD] = 1.0; sc[S_FLAG] = 0.0;
  } else if (step == 1) {  // rho <- (r,rp); beta = (rho/rhoold)*(alpha/omegaold)
    const double rho = sc[S_SUM0];
    sc[S_RHO] = rho;
    sc[S_BETA] = (rho / sc[S_RHOOLD]) * (sc[S_ALPHA] / sc[S_OMEGAOLD]);
  } else if (step == 2) {  // d1 = (v,rp); alpha = rho/d1
    const double d1 = sc[S_SUM0];
    sc[S_D1] = d1;
    sc[S_ALPHA] = sc[S_RHO] / d1;
  } else if (step == 3) {  // d1 = (s,t), d2 = (t,t); omega = d1/d2
    const double d1 = sc[S_SUM0], d2 = sc[S_SUM0 + 1];
    sc[S_D1] = d1; sc[S_D2] = d2;
    if (d2 == 0.0) { sc[S_OMEGA] = 0.0; sc[S_FLAG] = 1.0; }  // t == 0: x += alpha p is the candidate solution
    else sc[S_OMEGA] = d1 / d2;
  } else if (step == 4) {  // end of iteration: dp^2 = (r,r); rhoold = rho; omegaold = omega
    sc[S_DP] = sqrt(sc[S_SUM0]);
    sc[S_RHOOLD] = sc[S_RHO]; sc[S_OMEGAOLD] = sc[S_OMEGA];
  }
}

__global__ __launch_bounds__(kBlock) void k_maxpy_neg(MaxpyArgs a, const double *sc, double *y) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * kBlock) {
    double v = y[i];
    for (int j = 0; j < a.nvec; ++j) v -= sc[S_SUM0 + j] * a.v[j][i];
    y[i] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_maxpy_val(MaxpyValArgs a, double *x) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * kBlock) {
    double v = x[i];
    for (int j = 0; j < a.nvec; ++j) v += a.c[j] * a.v[j][i];
    x[i] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_scale_inv(int n, double *x, const double *sc, int slot) {
  const double d = sc[slot];
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) x[i] = x[i] / d;
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

void launch_assemble_cols(int mode, const ColsArgs &a, hipStream_t stream) {
  if (a.ncols <= 0) return;
  dim3 grid(nblk(a.ncols), a.nseg);
  if (mode == 0) LAUNCH(k_assemble_cols<0>, grid, a);
  else if (mode == 1) LAUNCH(k_assemble_cols<1>, grid, a);
  else LAUNCH(k_assemble_cols<2>, grid, a);
}

void launch_assemble_general(int mode, const GenArgs &a, hipStream_t stream) {
  if (a.n <= 0) return;
  dim3 grid(nblk(a.n));
  if (mode == 0) { LAUNCH(k_auxvars<0>, grid, a); LAUNCH(k_flux_rows<0>, grid, a); }
  else if (mode == 1) { LAUNCH(k_auxvars<1>, grid, a); LAUNCH(k_flux_rows<1>, grid, a); }
  else { LAUNCH(k_auxvars<2>, grid, a); LAUNCH(k_flux_rows<2>, grid, a); }
}

void launch_bc(int mode, const BcArgs &a, hipStream_t stream) {
  if (a.ncond_cells <= 0) return;
  dim3 grid(nblk(a.ncond_cells));
  if (mode == 0) LAUNCH(k_bc<0>, grid, a);
  else if (mode == 1) LAUNCH(k_bc<1>, grid, a);
  else if (mode == 2) LAUNCH(k_bc<2>, grid, a);
  else LAUNCH(k_bc<4>, grid, a);
}

void launch_conn_flux(const CellArrays &cell, const ConnArrays &conn, const int *cup, const int *cdn, int nc,
                      double *flux_out, hipStream_t stream) {
  if (nc > 0) LAUNCH(k_conn_flux, dim3(nblk(nc)), cell, conn, cup, cdn, nc, flux_out);
}
void launch_cell_var(const CellArrays &cell, int n, int var_type, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_cell_var, dim3(nblk(n)), cell, n, var_type, out);
}

void launch_spmv(int variant, int n, const int *rowptr, const int *col, const double *val, const double *x, double *y,
                 hipStream_t stream) {
  if (n <= 0) return;
  if (variant == 1) LAUNCH(k_spmv_stream, dim3(nblk(n)), n, rowptr, col, val, x, y);
  else LAUNCH(k_spmv_row, dim3(nblk(n)), n, rowptr, col, val, x, y);
}

void launch_tri_factor(const TriArgs &a, hipStream_t stream) {
  if (a.ncols > 0) LAUNCH(k_tri_factor, dim3(nblk(a.ncols)), a);
}
void launch_tri_solve(const TriArgs &a, const double *r, double *z, hipStream_t stream) {
  if (a.ncols > 0) LAUNCH(k_tri_solve, dim3(nblk(a.ncols)), a, r, z);
}
void launch_jacobi_setup(int n, const int *diag_slot, const double *val, double *invd, hipStream_t stream) {
  if (n > 0) LAUNCH(k_jacobi_setup, dim3(nblk(n)), n, diag_slot, val, invd);
}

void launch_copy(int n, const double *x, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_copy, dim3(vec_grid(n)), n, x, y);
}
void launch_set(int n, double v, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_set, dim3(vec_grid(n)), n, v, y);
}
void launch_gather(int n, const int *idx, const double *in, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_gather, dim3(vec_grid(n)), n, idx, in, out);
}
void launch_scatter(int n, const int *idx, const double *in, double *out, hipStream_t stream) {
  if (n > 0) LAUNCH(k_scatter, dim3(vec_grid(n)), n, idx, in, out);
}
void launch_pointwise_mult(int n, const double *a, const double *x, double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_pmult, dim3(vec_grid(n)), n, a, x, y);
}
void launch_waxpy(int n, double *w, const double *sc, int slot, double sign, double aval, const double *x,
                  const double *y, hipStream_t stream) {
  if (n > 0) LAUNCH(k_waxpy, dim3(vec_grid(n)), n, w, sc, slot, sign, aval, x, y);
}
void launch_bcgs_p(int n, double *P, const double *R, const double *V, const double *sc, hipStream_t stream) {
  if (n > 0) LAUNCH(k_bcgs_p, dim3(vec_grid(n)), n, P, R, V, sc);
}
int launch_bcgs_xr(int n, double *X, double *R, const double *P, const double *S, const double *T, const double *sc,
                   double *partials, hipStream_t stream) {
  const int g = red_grid(n);
  LAUNCH(k_bcgs_xr, dim3(g), n, X, R, P, S, T, sc, partials);
  return g;
}
int launch_dots(const DotArgs &a, double *partials, hipStream_t stream) {
  const int g = red_grid(a.n);
  switch (a.nvec) {
    case 1: LAUNCH(k_dots<1>, dim3(g), a, partials); break;
    case 2: LAUNCH(k_dots<2>, dim3(g), a, partials); break;
    case 3: LAUNCH(k_dots<3>, dim3(g), a, partials); break;
    case 4: LAUNCH(k_dots<4>, dim3(g), a, partials); break;
    default:
      for (int j0 = 0; j0 < a.nvec; j0 += 4) {
        const int nv = a.nvec - j0 < 4 ? a.nvec - j0 : 4;
        LAUNCH(k_dots_dyn, dim3(g), a, j0, nv, partials);
      }
  }
  return g;
}
void launch_finalize_partials(int nblocks, int nvec, const double *partials, double *sc, hipStream_t stream) {
  LAUNCH(k_finalize, dim3(nvec), nblocks, partials, sc);
}
void launch_maxpy_neg(const MaxpyArgs &a, const double *sc, double *y, hipStream_t stream) {
  if (a.n > 0 && a.nvec > 0) LAUNCH(k_maxpy_neg, dim3(vec_grid(a.n)), a, sc, y);
}
void launch_maxpy_val(const MaxpyValArgs &a, double *x, hipStream_t stream) {
  if (a.n > 0 && a.nvec > 0) LAUNCH(k_maxpy_val, dim3(vec_grid(a.n)), a, x);
}
void launch_scale(int n, double *x, const double *sc, int slot_inv_of, hipStream_t stream) {
  if (n > 0) LAUNCH(k_scale_inv, dim3(vec_grid(n)), n, x, sc, slot_inv_of);
}
void launch_bcgs_scalars(int step, double *sc, hipStream_t stream) {
  hipLaunchKernelGGL(k_bcgs_scalars, dim3(1), dim3(64), 0, stream, step, sc);
}

}  // namespace mpp
