// th_kernels.hip -- HIP kernels (gfx950) of the coupled thermal-hydrology system of equations:
// Richards mass equation + enthalpy-based heat equation on one mesh, unknowns field-blocked [P(1..N); T(1..N)],
// one 2N x 2N CSR Jacobian whose four N x N blocks all have the mesh's pattern (diagonal + internal connections).
//
// Restates SOETHResidual / SOETHJacobian (src/mpp/soe/SystemOfEquationsTHType.F90:736-1003) with
//   heat equation residual + d/dT   src/mpp/ge/GoveqnThermalEnthalpySoilType.F90:1174-1716
//   d(heat)/dP                      src/mpp/ge/GoveqnThermalEnthalpySoilType.F90:2083-2376
//   d(mass)/dT                      src/mpp/ge/GoveqnRichardsODEPressureType.F90:2473-2612
//   mass equation + d/dP            src/mpp/ge/GoveqnRichardsODEPressureType.F90:388-453, 1603-2200
// Three passes, no atomics: (1) one thread per cell evaluates both equations' auxvars once, stores what the flux
// pass needs (13 doubles/cell) and writes the accumulation terms; (2) one thread per cell gathers its incident
// connections in the caller's order and writes its two residual rows and its 2 x 2 x (row length) Jacobian
// entries; (3) one thread per cell with boundary connections.  Slot arithmetic instead of slot tables: with
// crp = the mesh pattern's row pointer, len = crp[i+1]-crp[i], nnzc = crp[N], entry s of cell row i lives at
//   (P,P): crp[i] + s          (P,T): crp[i] + s + len          (T,P): 2 nnzc + crp[i] + s       (T,T): ... + len
#include "kernels.hpp"
#include "th_physics.hpp"

namespace mpp {

namespace {

constexpr int kBlock = 256;
// The fused product + block-Thomas kernel (one lane per column) needs 130-140 registers, a few more than four waves per SIMD allow.
// Asked for four (128 registers, 8 bytes of scratch in the variant that also reduces the line search's sum) it gains where the grid is
// a whole number of rounds at four waves and not at three: 512 x 512 columns are 4096 waves = 16 per CU, one round instead of 1.33 --
// 147.2 / 147.0 -> 149.2 / 150.0 Newton it/s at 512 x 512 x 80 (profiles/r05/th_waves_ab.json; round 5's first try on the kernel of
// that time, which read two factor planes more, had shown nothing: profiles/r05/occupancy_hints_ab.json).  The stand-alone solves
// (24 bytes of scratch at 128 registers) stay as the compiler makes them.  Not kept (same file): BiCGStab's s = r - alpha v formed
// where this kernel loads its operand, as the scalar system's SPMV == 3 -- at 128 registers the two extra streams spill (40 bytes of
// scratch) and the launch and 16 bytes a cell they save do not show: 149.8 / 150.0 / 149.8 against 150.0 / 149.7 / 151.0 Newton it/s.
#ifndef MPP_TH_SOLVE_WAVES
#define MPP_TH_SOLVE_WAVES 4
#endif

// active[i]: 0 inactive cell, 1 active cell of this rank, 2 ghost cell (a copy of a cell another rank owns: a cell like any other
// for its neighbours' fluxes, an identity row with a zero right-hand side itself)
__device__ __forceinline__ bool th_active(const ThArgs &a, size_t i) { return a.active ? a.active[i] != 0 : true; }
__device__ __forceinline__ bool th_ghost(const ThArgs &a, size_t i) { return a.active ? a.active[i] == 2 : false; }

__device__ __forceinline__ Soil th_load_soil(const ThArgs &a, size_t i) {
  Soil s;
  s.sat_res = a.sat_res[i]; s.alpha = a.alpha[i]; s.m = a.m[i]; s.por = a.por[i];
  s.sf = a.satf ? (int)a.satf[i] : a.sat_func;   // satParams are per auxvar (ThermalEnthalpySoilAuxType.F90:156-157): per-cell dispatch
  if (s.sf == kSatSBC) { s.pu = a.sbc_pu[i]; s.ps = a.sbc_ps[i]; s.b2 = a.sbc_b2[i]; s.b3 = a.sbc_b3[i]; }
  else { s.pu = s.ps = s.b2 = s.b3 = 0.0; }
  if (a.por_slope) { s.por_pref = a.por_pref[i]; s.por_slope = a.por_slope[i]; } else { s.por_pref = s.por_slope = 0.0; }
  return s;
}

// the stored part of a cell's auxvars (what pass 2 and 3 read back)
__device__ __forceinline__ ThAux th_load_aux(const ThArgs &a, size_t i) {
  ThAux x;
  x.P = a.x[i]; x.T = a.x[(size_t)a.n + i];
  x.kr = a.kr[i]; x.dkr = a.dkr[i]; x.sat = 0.0; x.dsat = 0.0;
  x.den_m = a.den_m[i]; x.ddenP_m = a.ddenP_m[i]; x.ddenT_m = a.ddenT_m[i];
  x.den_e = a.den_e[i]; x.ddenP_e = a.ddenP_e[i]; x.ddenT_e = a.ddenT_e[i];
  x.hl = a.hl[i]; x.dhl_dP = a.dhl_dP[i]; x.dhl_dT = a.dhl_dT[i];
  x.ul = x.dul_dP = x.dul_dT = 0.0;
  x.tc = a.tc[i]; x.dtc_dP = a.dtc_dP[i];
  return x;
}
__device__ __forceinline__ Aux mass_view(const ThAux &x) { return Aux{x.P, x.kr, x.dkr, x.den_m, x.ddenP_m, 0.0, 0.0}; }
__device__ __forceinline__ Aux heat_view(const ThAux &x) { return Aux{x.P, x.kr, x.dkr, x.den_e, x.ddenP_e, 0.0, 0.0}; }

// ------------------------------------------------------------------------------------------------
// pass 1.  MODE 0: residual accumulation; 2: + the four diagonal accumulation derivatives; 3: PreSolve
// (accumulation terms only, written to `F` without subtracting accum_prev)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_th_aux(ThArgs a) {
  constexpr bool JAC = (MODE == 2), PRE = (MODE == 3);
  MPP_STAGE_POW(a.powtab);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t n = (size_t)a.n;
  if (i >= n) return;
  const bool ghost = th_ghost(a, i), act = th_active(a, i) && !ghost;
  const Soil s = th_load_soil(a, i);
  const ThSoil ts{a.tc_wet[i], a.tc_dry[i], a.talpha[i]};
  const double P = a.x[i], T = a.x[n + i];
  const ThAux x = th_auxvar_compute<true>(s.sf, a.density_type, a.ie_type, s, ts, P, T, true, pt);  // every cell (GE:942-961)
  if (!PRE) {
    a.kr[i] = x.kr; a.dkr[i] = x.dkr;
    a.den_m[i] = x.den_m; a.ddenP_m[i] = x.ddenP_m; a.ddenT_m[i] = x.ddenT_m;
    a.den_e[i] = x.den_e; a.ddenP_e[i] = x.ddenP_e; a.ddenT_e[i] = x.ddenT_e;
    a.hl[i] = x.hl; a.dhl_dP[i] = x.dhl_dP; a.dhl_dT[i] = x.dhl_dT;
    a.tc[i] = x.tc; a.dtc_dP[i] = x.dtc_dP;
  }
  const double vol = a.vol[i];
  const double dsoil = a.den_soil[i], csoil = a.heat_cap[i];
  double fm = 0.0, fe = 0.0;
  const bool lin_por = a.por_slope != nullptr;   // POROSITY_LINEAR: por(P), dpor/dP enter both accumulation terms
  double por = s.por, dpor = 0.0;
  if (lin_por) porosity(s, true, P, por, dpor);
  if (act) {
    fm = por * x.den_m * x.sat * vol * a.dtInv;                                                         // GE(Richards):1603-1634
    fe = (por * x.den_e * x.sat * x.ul + (1.0 - por) * dsoil * csoil * (T - 273.15)) * vol * a.dtInv;  // GE(heat):1174-1219
  }
  if (PRE) { a.F[i] = fm; a.F[n + i] = fe; return; }
  a.F[i] = ghost ? 0.0 : fm - a.accum_prev[i];
  a.F[n + i] = ghost ? 0.0 : fe - a.accum_prev[n + i];
  if (JAC) {
    const int r0 = a.crowptr[i], len = a.crowptr[i + 1] - r0;
    const int sPP = r0 + a.cdiag[i], sPT = sPP + len, sTP = 2 * a.nnzc + sPP, sTT = sTP + len;
    double dPP = 1.0, dPT = 1.0, dTP = 1.0, dTT = 1.0;  // inactive cell: 1 on all four diagonals (GE:1682; 2501; 2136; 1283)
    if (ghost) dPT = dTP = 0.0;                         // ghost cell: identity rows
    if (act) {
      dPT = (por * x.ddenT_m * x.sat) * vol * a.dtInv;
      dTT = ((por * x.ddenT_e * x.sat * x.ul + por * x.den_e * x.sat * x.dul_dT) + (1.0 - por) * dsoil * csoil) * vol * a.dtInv;
      if (lin_por) {   // GE(Richards):1675-1677; GE(heat):2178-2182
        dPP = (dpor * x.den_m * x.sat + por * x.ddenP_m * x.sat + por * x.den_m * x.dsat) * vol * a.dtInv;
        dTP = (dpor * x.den_e * x.sat * x.ul + por * x.ddenP_e * x.sat * x.ul + por * x.den_e * x.dsat * x.ul + por * x.den_e * x.sat * x.dul_dP +
               (-dpor * dsoil * csoil * (T - 273.15))) * vol * a.dtInv;
      } else {
        dPP = (por * x.ddenP_m * x.sat + por * x.den_m * x.dsat) * vol * a.dtInv;
        dTP = (por * x.ddenP_e * x.sat * x.ul + por * x.den_e * x.dsat * x.ul + por * x.den_e * x.sat * x.dul_dP) * vol * a.dtInv;
      }
    }
    a.val[sPP] = dPP; a.val[sPT] = dPT; a.val[sTP] = dTP; a.val[sTT] = dTT;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2: internal connections, row-gather
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_th_rows(ThArgs a) {
  constexpr bool JAC = (MODE == 2);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t n = (size_t)a.n;
  if (i >= n) return;
  const int r0 = a.crowptr[i], len = a.crowptr[i + 1] - r0;
  const int bPP = r0, bPT = r0 + len, bTP = 2 * a.nnzc + r0, bTT = bTP + len;   // add the cell-pattern slot s
  const int sd = a.cdiag[i];
  if (JAC) {
    for (int s = r0; s < r0 + len; ++s)
      if (s != sd) { a.val[bPP + s] = 0.0; a.val[bPT + s] = 0.0; a.val[bTP + s] = 0.0; a.val[bTT + s] = 0.0; }
  }
  double fm = a.F[i], fe = a.F[n + i];
  if (th_ghost(a, i)) { a.F[i] = 0.0; a.F[n + i] = 0.0; return; }   // (the off-diagonal entries were zeroed above)
  if (th_active(a, i)) {
    const ThAux me = th_load_aux(a, i);
    double dPP = 0.0, dPT = 0.0, dTP = 0.0, dTT = 0.0;
    for (int e = a.rc_ptr[i]; e < a.rc_ptr[i + 1]; ++e) {
      const int enc = a.rc_conn[e];
      const int ic = enc >> 1, side = enc & 1;
      const size_t o = side ? a.cup[ic] : a.cdn[ic];
      if (!th_active(a, o)) continue;
      const ThAux ot = th_load_aux(a, o);
      const ThAux &up = side ? ot : me;
      const ThAux &dn = side ? me : ot;
      ConnGeom gm, ge;
      gm.Dq = a.Dq_m[ic]; gm.upweight = a.upw[ic]; gm.dist_gravity = a.dgrav[ic]; gm.area = a.area[ic];
      ge = gm; ge.Dq = a.Dq_e[ic];
      const double du = a.cdu[ic], dd = a.cdd[ic];
      // mass equation
      double Jup = 0.0, Jdn = 0.0;     // negatives of d(flux)/dP (RichardsMod.F90:326-334)
      const double flux_m = richards_flux<JAC>(mass_view(up), mass_view(dn), gm, 0, Jup, Jdn);
      // heat equation: its own Darcy flux (density at max(P, Pref), its own permeability) carries the enthalpy
      double Eup = 0.0, Edn = 0.0;
      const double mflux = richards_flux<JAC>(heat_view(up), heat_view(dn), ge, 0, Eup, Edn);
      double t1, t2;
      const double eflux = thermal_enthalpy_flux<false>(up, dn, mflux, 0.0, 0.0, gm.area, du, dd, true, t1, t2);
      const double sgn = side ? 1.0 : -1.0;    // up row: f -= flux; dn row: f += flux
      fm += sgn * flux_m;
      fe += sgn * eflux;
      if (JAC) {
        const int so = a.rc_slot[e];           // cell-pattern slot of (this row, other cell)
        // (P,P): J(up,up)+=Jup, J(up,dn)+=Jdn, J(dn,up)-=Jup, J(dn,dn)-=Jdn (GE:2051-2069)
        if (side == 0) { dPP += Jup; a.val[bPP + so] += Jdn; } else { dPP += -Jdn; a.val[bPP + so] += -Jup; }
        // (P,T): true derivatives; (up,up) -= Tu, (up,dn) -= Td, (dn,up) += Tu, (dn,dn) += Td (GE:2560-2597)
        double Tu, Td;
        richards_flux_dT(up.P, up.kr, up.den_m, up.ddenT_m, dn.P, dn.kr, dn.den_m, dn.ddenT_m, gm, 0, Tu, Td);
        if (side == 0) { dPT += -Tu; a.val[bPT + so] += -Td; } else { dPT += Td; a.val[bPT + so] += Tu; }
        // (T,P)
        double Pu, Pd;
        thermal_enthalpy_flux_dP(up, dn, mflux, -Eup, -Edn, gm.area, du, dd, true, Pu, Pd);
        if (side == 0) { dTP += -Pu; a.val[bTP + so] += -Pd; } else { dTP += Pd; a.val[bTP + so] += Pu; }
        // (T,T)
        double mTu, mTd, Hu, Hd;
        richards_flux_dT(up.P, up.kr, up.den_e, up.ddenT_e, dn.P, dn.kr, dn.den_e, dn.ddenT_e, ge, 0, mTu, mTd);
        thermal_enthalpy_flux<true>(up, dn, mflux, mTu, mTd, gm.area, du, dd, true, Hu, Hd);
        if (side == 0) { dTT += -Hu; a.val[bTT + so] += -Hd; } else { dTT += Hd; a.val[bTT + so] += Hu; }
      }
    }
    if (JAC) {
      a.val[bPP + sd] = dPP + a.val[bPP + sd];   // divergence first, accumulation last (GE:425-453; heat :1092-1120)
      a.val[bTT + sd] = dTT + a.val[bTT + sd];
      a.val[bPT + sd] = a.val[bPT + sd] + dPT;   // accumulation first, then connections (GE:2501-2597; heat :2136-2250)
      a.val[bTP + sd] = a.val[bTP + sd] + dTP;
    }
  }
  if (a.src) { fm -= a.src[i]; fe += a.src[n + i]; }   // COND_MASS_RATE: f -= value/FMWH2O (GE:1896); COND_HEAT_RATE: f += value (heat :1481)
  a.F[i] = fm;
  a.F[n + i] = fe;
}

// ------------------------------------------------------------------------------------------------
// vertical-only grids: passes 1 and 2 fused into one column march (one lane per soil column, tiled level-major layout).
// Every cell's auxvars and every connection's fluxes are evaluated once and live in registers; nothing but x, the soil
// parameters and the geometry is read, nothing but F and the 12 Jacobian entries per cell is written.  Row k-1 is
// complete when level k has been visited; its entries sit at crp, crp+1, crp+2 of each block (sorted columns: the cell
// below, the cell, the cell above), so no slot tables are needed.  Same per-row summation order as the 3-pass kernels.
// ------------------------------------------------------------------------------------------------
// 2 x 2 blocks of the per-column block-tridiagonal systems (block Thomas: k_th_tri_factor, and the march when it factorises)
struct B2 { double pp, pt, tp, tt; };
__device__ __forceinline__ B2 b2_load(const double *p, size_t n, size_t i) { return B2{p[i], p[n + i], p[2 * n + i], p[3 * n + i]}; }
__device__ __forceinline__ void b2_store(double *p, size_t n, size_t i, const B2 &b) {
  p[i] = b.pp; p[n + i] = b.pt; p[2 * n + i] = b.tp; p[3 * n + i] = b.tt;
}

struct ThRowAcc {      // what a row accumulates before it is written
  double fm, fe;
  double dPP, dPT, dTP, dTT;        // diagonal entries: connection part
  double aPP, aPT, aTP, aTT;        // diagonal entries: accumulation part
  double lPP, lPT, lTP, lTT;        // coupling to the cell below
};
// Registers (round 4): the march carried the level below (15 auxvar values) and its unfinished row (14 sums) through the evaluation
// of the current cell's equation of state -- the part that needs the most registers -- and ran at 256 + 60 registers, one wave per
// SIMD, on code that is bound by instruction issue.  The 29 values are parked in LDS across that evaluation (one column of a
// [29][block] array per thread: no bank conflicts, no barrier) and the kernel is compiled for two waves per SIMD.
constexpr int kColsBlock = 128, kParked = 29;
// (Round 4 also built a variant that took the boundary sums from side arrays and eliminated every finished block row itself -- no
// k_th_bc pass behind the march, no k_th_tri_factor: the extra live values spilled (212 bytes of scratch per thread) and the pass took
// 4.15 ms against 2.21 + 0.78 ms for march + factorisation at 512 x 512 x 80, gpurun_out/r04_run10; not kept.)
template <int MODE, int EOS_T>   // EOS_T: th_auxvar_compute's (1: IFC-67 density + enthalpy at compile time, -1: run-time choice)
__global__ __launch_bounds__(kColsBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_th_cols(ThArgs a) {
  constexpr bool JAC = (MODE == 2);
  __shared__ double pow_lds_[kPowTableDoubles];
  __shared__ double park[JAC ? kParked : 1][kColsBlock];
  const PowTab pt = stage_pow_tables(a.powtab, pow_lds_);
  const int c = blockIdx.x * kColsBlock + threadIdx.x;
  if (c >= a.ncols) return;
  const int nz = a.nz;
  const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
  const size_t cbase = (size_t)tile * 64 * nz + (c & 63), n = (size_t)a.n;
  ThAux prev = {};
  bool prev_act = false;
  ThRowAcc rp = {};
  for (int k = 0; k <= nz; ++k) {
    const size_t i = cbase + (size_t)k * tw;
    ThAux cur;
    bool act = false;
    ThRowAcc rc = {};
    double uPP = 0.0, uPT = 0.0, uTP = 0.0, uTT = 0.0;   // row k-1's coupling to cell k
    if (k < nz) {
      act = th_active(a, i);
      const Soil s = th_load_soil(a, i);
      const ThSoil ts{a.tc_wet[i], a.tc_dry[i], a.talpha[i]};
      const double P = a.x[i], T = a.x[n + i];
      if (JAC) {   // park the carried state across the equation of state (the residual-only march fits two waves without: measured
                   // 1.03 -> 2.17 ms when it parked too, gpurun_out/r04_run8)
        const int t = threadIdx.x;
        const double v[kParked] = {prev.P, prev.T, prev.kr, prev.dkr, prev.den_m, prev.ddenP_m, prev.ddenT_m, prev.den_e, prev.ddenP_e, prev.ddenT_e,
                                   prev.hl, prev.dhl_dP, prev.dhl_dT, prev.tc, prev.dtc_dP, rp.fm, rp.fe, rp.dPP, rp.dPT, rp.dTP, rp.dTT, rp.aPP, rp.aPT,
                                   rp.aTP, rp.aTT, rp.lPP, rp.lPT, rp.lTP, rp.lTT};
#pragma unroll
        for (int q = 0; q < kParked; ++q) park[q][t] = v[q];
        asm volatile("" ::: "memory");
      }
      cur = th_auxvar_compute<true, EOS_T>(s.sf, a.density_type, a.ie_type, s, ts, P, T, true, pt);
      if (JAC) {
        asm volatile("" ::: "memory");
        const int t = threadIdx.x;
        double v[kParked];
#pragma unroll
        for (int q = 0; q < kParked; ++q) v[q] = park[q][t];
        prev.P = v[0]; prev.T = v[1]; prev.kr = v[2]; prev.dkr = v[3]; prev.den_m = v[4]; prev.ddenP_m = v[5]; prev.ddenT_m = v[6];
        prev.den_e = v[7]; prev.ddenP_e = v[8]; prev.ddenT_e = v[9]; prev.hl = v[10]; prev.dhl_dP = v[11]; prev.dhl_dT = v[12];
        prev.tc = v[13]; prev.dtc_dP = v[14];
        rp.fm = v[15]; rp.fe = v[16]; rp.dPP = v[17]; rp.dPT = v[18]; rp.dTP = v[19]; rp.dTT = v[20]; rp.aPP = v[21]; rp.aPT = v[22];
        rp.aTP = v[23]; rp.aTT = v[24]; rp.lPP = v[25]; rp.lPT = v[26]; rp.lTP = v[27]; rp.lTT = v[28];
      }
      const double vol = a.vol[i], dsoil = a.den_soil[i], csoil = a.heat_cap[i];
      double fm = 0.0, fe = 0.0;
      rc.aPP = rc.aPT = rc.aTP = rc.aTT = 1.0;
      if (act) {
        const bool lin_por = a.por_slope != nullptr;
        double por = s.por, dpor = 0.0;
        if (lin_por) porosity(s, true, P, por, dpor);
        fm = por * cur.den_m * cur.sat * vol * a.dtInv;
        fe = (por * cur.den_e * cur.sat * cur.ul + (1.0 - por) * dsoil * csoil * (T - 273.15)) * vol * a.dtInv;
        if (JAC) {
          rc.aPT = (por * cur.ddenT_m * cur.sat) * vol * a.dtInv;
          rc.aTT = ((por * cur.ddenT_e * cur.sat * cur.ul + por * cur.den_e * cur.sat * cur.dul_dT) + (1.0 - por) * dsoil * csoil) * vol * a.dtInv;
          if (lin_por) {
            rc.aPP = (dpor * cur.den_m * cur.sat + por * cur.ddenP_m * cur.sat + por * cur.den_m * cur.dsat) * vol * a.dtInv;
            rc.aTP = (dpor * cur.den_e * cur.sat * cur.ul + por * cur.ddenP_e * cur.sat * cur.ul + por * cur.den_e * cur.dsat * cur.ul +
                      por * cur.den_e * cur.sat * cur.dul_dP + (-dpor * dsoil * csoil * (T - 273.15))) * vol * a.dtInv;
          } else {
            rc.aPP = (por * cur.ddenP_m * cur.sat + por * cur.den_m * cur.dsat) * vol * a.dtInv;
            rc.aTP = (por * cur.ddenP_e * cur.sat * cur.ul + por * cur.den_e * cur.dsat * cur.ul + por * cur.den_e * cur.sat * cur.dul_dP) * vol * a.dtInv;
          }
        }
      }
      rc.fm = fm - a.accum_prev[i];
      rc.fe = fe - a.accum_prev[n + i];
      if (k > 0 && prev_act && act) {   // the vertical connection (k-1) -> (k), stored at the index of cell k-1
        const size_t ic = i - tw;
        ConnGeom gm, ge;
        gm.Dq = a.Dq_m[ic]; gm.upweight = a.upw[ic]; gm.dist_gravity = a.dgrav[ic]; gm.area = a.area[ic];
        ge = gm; ge.Dq = a.Dq_e[ic];
        const double du = a.cdu[ic], dd = a.cdd[ic];
        double Jup = 0.0, Jdn = 0.0, Eup = 0.0, Edn = 0.0, t1, t2;
        const double flux_m = richards_flux<JAC>(mass_view(prev), mass_view(cur), gm, 0, Jup, Jdn);
        const double mflux = richards_flux<JAC>(heat_view(prev), heat_view(cur), ge, 0, Eup, Edn);
        const double eflux = thermal_enthalpy_flux<false>(prev, cur, mflux, 0.0, 0.0, gm.area, du, dd, true, t1, t2);
        rp.fm -= flux_m; rc.fm += flux_m;
        rp.fe -= eflux; rc.fe += eflux;
        if (JAC) {
          rp.dPP += Jup; uPP = Jdn; rc.lPP = -Jup; rc.dPP += -Jdn;
          double Tu, Td, Pu, Pd, mTu, mTd, Hu, Hd;
          richards_flux_dT(prev.P, prev.kr, prev.den_m, prev.ddenT_m, cur.P, cur.kr, cur.den_m, cur.ddenT_m, gm, 0, Tu, Td);
          rp.dPT += -Tu; uPT = -Td; rc.lPT = Tu; rc.dPT += Td;
          thermal_enthalpy_flux_dP(prev, cur, mflux, -Eup, -Edn, gm.area, du, dd, true, Pu, Pd);
          rp.dTP += -Pu; uTP = -Pd; rc.lTP = Pu; rc.dTP += Pd;
          richards_flux_dT(prev.P, prev.kr, prev.den_e, prev.ddenT_e, cur.P, cur.kr, cur.den_e, cur.ddenT_e, ge, 0, mTu, mTd);
          thermal_enthalpy_flux<true>(prev, cur, mflux, mTu, mTd, gm.area, du, dd, true, Hu, Hd);
          rp.dTT += -Hu; uTT = -Hd; rc.lTT = Hu; rc.dTT += Hd;
        }
      }
    }
    if (k > 0) {   // row k-1 is complete
      const size_t ip = i - tw;
      const int kp = k - 1;
      double fm = rp.fm, fe = rp.fe;
      if (a.src) { fm -= a.src[ip]; fe += a.src[n + ip]; }
      a.F[ip] = fm; a.F[n + ip] = fe;
      if (JAC) {
        const int r0 = a.crowptr[ip], len = a.crowptr[ip + 1] - r0;
        size_t g0 = (size_t)2 * r0, g1 = g0 + len, h0 = (size_t)2 * a.nnzc + g0, h1 = h0 + len;   // first entry of each block
        if (kp > 0) { a.val[g0++] = rp.lPP; a.val[g1++] = rp.lPT; a.val[h0++] = rp.lTP; a.val[h1++] = rp.lTT; }
        a.val[g0++] = rp.dPP + rp.aPP;   // divergence first, accumulation last (GE:425-453; heat :1092-1120)
        a.val[h1++] = rp.dTT + rp.aTT;
        a.val[g1++] = rp.aPT + rp.dPT;   // accumulation first, then connections (GE:2501-2597; heat :2136-2250)
        a.val[h0++] = rp.aTP + rp.dTP;
        if (kp < nz - 1) { a.val[g0] = uPP; a.val[g1] = uPT; a.val[h0] = uTP; a.val[h1] = uTT; }
      }
    }
    prev = cur; prev_act = act; rp = rc;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 3: boundary connections of both equations, grouped by cell.  Mass equation: Dirichlet / seepage / mass-flux
// (GE:1815-1877, 2076-2139); heat equation: COND_DIRICHLET (heat :1391-1452, 1627-1680, 2283-2370).  A boundary
// auxvar has the soil of its id_dn cell; the mass equation's carries a temperature, the heat equation's a pressure
// (assigned by the caller, as the reference drivers do).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_th_bc(ThBcArgs b) {
  constexpr bool JAC = (MODE == 2);
  const ThArgs &a = b.t;
  MPP_STAGE_POW(a.powtab);
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= b.ncond_cells) return;
  const size_t cell = b.cc_cell[j];
  const size_t n = (size_t)a.n;
  if (!th_active(a, cell) || th_ghost(a, cell)) return;
  const Soil s = th_load_soil(a, cell);
  const ThSoil ts{a.tc_wet[cell], a.tc_dry[cell], a.talpha[cell]};
  const ThAux in = th_auxvar_compute<true>(s.sf, a.density_type, a.ie_type, s, ts, a.x[cell], a.x[n + cell], true, pt);
  double fm = a.F[cell], fe = a.F[n + cell];
  double dPP = 0.0, dTP = 0.0, dTT = 0.0;
  for (int e = b.cc_ptr[j]; e < b.cc_ptr[j + 1]; ++e) {
    const int t = b.type[e];
    const int ctype = t & 0xffff, swap = (t >> 16) & 1, heat = (t >> 20) & 1;
    ConnGeom g;
    g.Dq = b.Dq[e]; g.upweight = 0.0; g.dist_gravity = b.dgrav[e]; g.area = b.area[e];
    if (!heat) {
      const double pb = (ctype == kCondMassFlux) ? in.P : b.value[e];
      const ThAux bc = th_auxvar_compute<true>(s.sf, a.density_type, a.ie_type, s, ts, pb, b.other[e], true, pt);
      double Jup = 0.0, Jdn = 0.0, flux, jdn_outer;
      if (!swap) { flux = richards_flux<JAC>(mass_view(bc), mass_view(in), g, ctype, Jup, Jdn); jdn_outer = Jdn; }
      else { flux = -richards_flux<JAC>(mass_view(in), mass_view(bc), g, ctype, Jup, Jdn); jdn_outer = -Jup; }
      fm += flux;
      if (JAC) dPP += -jdn_outer;
    } else {
      const ThAux bc = th_auxvar_compute<true>(s.sf, a.density_type, a.ie_type, s, ts, b.other[e], b.value[e], false, pt);
      double Eup = 0.0, Edn = 0.0, t1, t2;
      const double mflux = richards_flux<JAC>(heat_view(bc), heat_view(in), g, ctype, Eup, Edn);
      const double du = b.du[e], dd = b.dd[e];
      fe += thermal_enthalpy_flux<false>(bc, in, mflux, 0.0, 0.0, g.area, du, dd, false, t1, t2);
      if (JAC) {
        double Pu, Pd, mTu, mTd, Hu, Hd;
        thermal_enthalpy_flux_dP(bc, in, mflux, -Eup, -Edn, g.area, du, dd, false, Pu, Pd);
        dTP += Pd;
        richards_flux_dT(bc.P, bc.kr, bc.den_e, bc.ddenT_e, in.P, in.kr, in.den_e, in.ddenT_e, g, ctype, mTu, mTd);
        thermal_enthalpy_flux<true>(bc, in, mflux, mTu, mTd, g.area, du, dd, false, Hu, Hd);
        dTT += Hd;
      }
    }
  }
  a.F[cell] = fm;
  a.F[n + cell] = fe;
  if (JAC) {
    const int r0 = a.crowptr[cell], len = a.crowptr[cell + 1] - r0;
    const int sPP = r0 + a.cdiag[cell];
    a.val[sPP] += dPP;
    a.val[2 * a.nnzc + sPP] += dTP;
    a.val[2 * a.nnzc + sPP + len] += dTT;
  }
}

// ------------------------------------------------------------------------------------------------
// block-Jacobi for the coupled system on vertical-only grids: per soil column the unknowns (P_0,T_0,...,P_nz-1,T_nz-1)
// form a block-tridiagonal system with 2 x 2 blocks; block Thomas is its exact LU (the role PETSc's bjacobi + sub-PC lu
// plays).  One lane per column, tiled level-major layout as everywhere else.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ B2 b2_load_val(const ThTriArgs &a, size_t i, int s) {   // the 2 x 2 block of cell row i at cell slot s
  const int r0 = a.crowptr[i], len = a.crowptr[i + 1] - r0;
  const size_t g = (size_t)r0 + s, h = (size_t)2 * a.nnzc + g;
  return B2{a.val[g], a.val[g + len], a.val[h], a.val[h + len]};
}
__device__ __forceinline__ double th_block_sum(double v, double *lds) {   // lanes -> wave leaders -> thread 0 (fixed order)
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int q = 0; q < kBlock / 64; ++q) t += lds[q];
  return t;
}
// Block Thomas as the kernels below store and apply it (round 5; before: L = lo inv_prev, the pivots' inverses and a copy of the upper
// blocks, 96 bytes a cell read by every solve):
//   G_k = inv_k up_k,  inv_k = (di_k - lo_k G_{k-1})^{-1};   forward  e_k = inv_k (r_k - lo_k e_{k-1});   backward  z_k = e_k - G_k z_{k+1}
// A solve reads inv on the way up and G on the way down (64 bytes a cell), and lo is the matrix's own lower block: the fused
// k_th_spmv_tri holds it anyway (the row it multiplies), the stand-alone solves read the copy in plane L.  Every product and sum of
// the elimination is rounded by itself (no contraction) in every kernel that performs it: they must agree bit for bit.
__device__ __forceinline__ B2 b2_mul_nc(const B2 &x, const B2 &y) {
#pragma clang fp contract(off)
  return B2{x.pp * y.pp + x.pt * y.tp, x.pp * y.pt + x.pt * y.tt, x.tp * y.pp + x.tt * y.tp, x.tp * y.pt + x.tt * y.tt};
}
__device__ __forceinline__ void th_factor_level(const B2 &lo, const B2 &di, const B2 &up, const B2 &g_prev, B2 &inv, B2 &g) {
#pragma clang fp contract(off)
  const B2 lg = b2_mul_nc(lo, g_prev);
  const B2 piv{di.pp - lg.pp, di.pt - lg.pt, di.tp - lg.tp, di.tt - lg.tt};
  const double id = 1.0 / (piv.pp * piv.tt - piv.pt * piv.tp);
  inv = B2{piv.tt * id, -piv.pt * id, -piv.tp * id, piv.pp * id};
  g = b2_mul_nc(inv, up);
}
// e = inv (r - lo e_prev), the forward step of every solve
__device__ __forceinline__ void th_forward_level(const B2 &lo, const B2 &inv, double rp, double rt, double &ep, double &et) {
#pragma clang fp contract(off)
  const double tp = rp - (lo.pp * ep + lo.pt * et);
  const double tt = rt - (lo.tp * ep + lo.tt * et);
  ep = inv.pp * tp + inv.pt * tt;
  et = inv.tp * tp + inv.tt * tt;
}

template <int ND> __device__ __forceinline__ void th_tri_backward_column(const ThTriArgs &a, double *__restrict__ z, const double *__restrict__ w,
                                                                         size_t cbase, int tw, double &acc0, double &acc1);

// The rows of a level of a full column tile are one contiguous piece of `val` per block row; lane by lane a row is a 48-byte stride
// (see k_th_spmv_cols), so a wave stages the two pieces of the level in LDS with 16-byte loads in address order and every lane
// takes its three blocks from there.  The slots (tri_lo, cdiag, tri_up, crowptr) are still read and checked against the column
// layout, level by level: a wave that finds anything else reads through them as before.
// SOLVE (round 5): the factorisation and the first solve of a Krylov solve, z = M^{-1} r with the sums of z.z (ND = 1) or z.w and z.z
// (ND = 2), as one kernel -- the elimination of r rides on the factorisation's sweep (the operands it needs are in registers), plane L
// is not written (the fused products that follow take lo from the matrix), and the factorisation's pass of its own (0.76 ms and 208
// bytes a cell at 512 x 512 x 80) is gone from a Newton iteration.
template <bool SOLVE, int ND>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MPP_TH_SOLVE_WAVES))) void k_th_tri_factor(ThTriArgs a, const double *__restrict__ r, double *__restrict__ z,
                                                          const double *__restrict__ w, double *__restrict__ partials) {
  typedef double dpair __attribute__((ext_vector_type(2)));   // two doubles, one 16-byte access
  __shared__ dpair stage[kBlock / 64][2][64 * 3];
  __shared__ double lds[kBlock / 64];
  const int c = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc0 = 0.0, acc1 = 0.0;
  if (c - lane < a.ncols) {   // (the whole wave)
    const bool live = c < a.ncols;
    const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
    const size_t cbase = (size_t)tile * 64 * a.nz + lane, n = (size_t)a.n;
    B2 g_prev{0, 0, 0, 0};
    double ep = 0.0, et = 0.0;
    dpair *sP = stage[wave][0], *sT = stage[wave][1];
    // the pieces of a level (and the right-hand side with them) are requested one level ahead, into registers, as in k_th_spmv_tri
    dpair nP[3], nT[3];
    double nrp = 0.0, nrt = 0.0;
    bool nstaged = false;
    auto prefetch = [&](int k) {
      nstaged = false;
      const size_t i = cbase + (size_t)k * tw;
      if (SOLVE && live) { nrp = r[i]; nrt = r[n + i]; }
      if (tw != 64 || !a.stage) return;   // (staged: every lane is a column)
      const int len = 3 - (k == 0) - (k == a.nz - 1);
      const int base = a.crowptr[i - lane], r0 = a.crowptr[i];
      bool mine = a.crowptr[i - lane + 64] - base == 64 * len && r0 - base == lane * len;
      if (!SOLVE) {   // the stand-alone factorisation also checks the slot tables against the layout; the combined kernel relies, like
                      // k_th_spmv_tri, on the rows' sorted columns (cell below, cell, cell above)
        const int sl = a.tri_lo[i], sd = a.cdiag[i], su = a.tri_up[i];
        mine = mine && sl == (k > 0 ? r0 : -1) && sd == r0 + (k > 0) && su == (k < a.nz - 1 ? sd + 1 : -1);
      }
      if (__builtin_amdgcn_ballot_w64(mine) != ~0ull) return;
      const dpair *gP = reinterpret_cast<const dpair *>(a.val + (size_t)2 * base),
                  *gT = reinterpret_cast<const dpair *>(a.val + (size_t)2 * a.nnzc + (size_t)2 * base);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < len) { nP[j] = __builtin_nontemporal_load(gP + j * 64 + lane); nT[j] = __builtin_nontemporal_load(gT + j * 64 + lane); }
      nstaged = true;
    };
    prefetch(0);
    for (int k = 0; k < a.nz; ++k) {
      const size_t i = cbase + (size_t)k * tw;
      const int len = 3 - (k == 0) - (k == a.nz - 1);
      const B2 zero{0, 0, 0, 0};
      B2 lo = zero, di = zero, up = zero;
      const bool staged = nstaged;
      const double rp = nrp, rt = nrt;
      if (staged) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < len) { sP[j * 64 + lane] = nP[j]; sT[j * 64 + lane] = nT[j]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (k + 1 < a.nz) prefetch(k + 1);
      if (staged) {
        const double *vP = reinterpret_cast<const double *>(sP) + lane * 2 * len, *vT = reinterpret_cast<const double *>(sT) + lane * 2 * len;
        const int d = k > 0;   // the diagonal block's place in the row
        if (k > 0) lo = B2{vP[0], vP[len], vT[0], vT[len]};
        di = B2{vP[d], vP[d + len], vT[d], vT[d + len]};
        if (k < a.nz - 1) up = B2{vP[d + 1], vP[d + 1 + len], vT[d + 1], vT[d + 1 + len]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the reads above precede the next level's writes)
        __builtin_amdgcn_wave_barrier();
      } else {
        if (!live) continue;
        const int sl = a.tri_lo[i], su = a.tri_up[i];
        if (sl >= 0) lo = b2_load_val(a, i, sl);
        di = b2_load_val(a, i, a.cdiag[i]);
        if (su >= 0) up = b2_load_val(a, i, su);
      }
      if (!live) continue;
      B2 inv, g;
      th_factor_level(lo, di, up, g_prev, inv, g);
      if (!SOLVE) b2_store(a.L, n, i, lo);
      if (!(SOLVE && a.inv_fly)) b2_store(a.inv, n, i, inv);   // (inv_fly: the products that follow redo the pivots from the rows they read)
      b2_store(a.C, n, i, g);
      g_prev = g;
      if (SOLVE) {
        th_forward_level(lo, inv, rp, rt, ep, et);
        z[i] = ep; z[n + i] = et;
      }
    }
    if (SOLVE && live) th_tri_backward_column<ND>(a, z, w, cbase, tw, acc0, acc1);
  }
  if (SOLVE) {
    const double t0 = th_block_sum(acc0, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t0;
    if (ND == 2) {
      const double t1 = th_block_sum(acc1, lds);
      if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t1;
    }
  }
}

// Both sweeps of a column's block-Thomas solve.  The operands of a level do not depend on the recurrence, only the arithmetic does:
// they are fetched kThDepth levels ahead into a ring of registers (the compiler's own loop waits for every level's loads before it
// starts the next ones).  Measured (round 4's storage): 0.68-0.71 ms at 512 x 512 x 80 either way -- the solve already ran at 5.5 TB/s
// of its 176 bytes per cell -- so the ring buys robustness at low occupancy, not time.  ND = 0: no dot products; 1: z.z; 2: z.w and z.z.
constexpr int kThDepth = 4;
__device__ __forceinline__ void th_tri_forward_column(const ThTriArgs &a, const double *__restrict__ r, double *__restrict__ z,
                                                      size_t cbase, int tw) {
  const size_t n = (size_t)a.n;
  B2 l[kThDepth], inv[kThDepth];
  double rp[kThDepth], rt[kThDepth];
  auto fetch = [&](int slot, int k) {
    const size_t i = cbase + (size_t)min(k, a.nz - 1) * tw;   // (past the last level: a harmless repeat)
    l[slot] = b2_load(a.L, n, i); inv[slot] = b2_load(a.inv, n, i); rp[slot] = r[i]; rt[slot] = r[n + i];
  };
#pragma unroll
  for (int j = 0; j < kThDepth; ++j) fetch(j, j);
  double ep = 0.0, et = 0.0;
  for (int k0 = 0; k0 < a.nz; k0 += kThDepth) {
#pragma unroll
    for (int j = 0; j < kThDepth; ++j) {
      const int k = k0 + j;
      if (k < a.nz) {
        const size_t i = cbase + (size_t)k * tw;
        th_forward_level(l[j], inv[j], rp[j], rt[j], ep, et);
        z[i] = ep; z[n + i] = et;
        fetch(j, k + kThDepth);
      }
    }
  }
}
template <int ND>
__device__ __forceinline__ void th_tri_backward_column(const ThTriArgs &a, double *__restrict__ z, const double *__restrict__ w,
                                                       size_t cbase, int tw, double &acc0, double &acc1) {
#pragma clang fp contract(off)
  const size_t n = (size_t)a.n;
  B2 g[kThDepth];
  double zp[kThDepth], zt[kThDepth], wp[kThDepth], wt[kThDepth];
  auto fetch = [&](int slot, int k) {
    const size_t i = cbase + (size_t)max(k, 0) * tw;
    g[slot] = b2_load(a.C, n, i); zp[slot] = z[i]; zt[slot] = z[n + i];
    if (ND == 2) { wp[slot] = w[i]; wt[slot] = w[n + i]; }
  };
#pragma unroll
  for (int j = 0; j < kThDepth; ++j) fetch(j, a.nz - 1 - j);
  double xp = 0.0, xt = 0.0;
  for (int k0 = a.nz - 1; k0 >= 0; k0 -= kThDepth) {
#pragma unroll
    for (int j = 0; j < kThDepth; ++j) {
      const int k = k0 - j;
      if (k >= 0) {
        const size_t i = cbase + (size_t)k * tw;
        const double yp = zp[j] - (g[j].pp * xp + g[j].pt * xt);
        const double yt = zt[j] - (g[j].tp * xp + g[j].tt * xt);
        const double wpj = wp[j], wtj = wt[j];
        fetch(j, k - kThDepth);
        z[i] = yp; z[n + i] = yt;
        xp = yp; xt = yt;
        if (ND == 2) { acc0 += yp * wpj; acc0 += yt * wtj; acc1 += yp * yp; acc1 += yt * yt; }
        else if (ND == 1) { acc0 += yp * yp; acc0 += yt * yt; }
      }
    }
  }
}
template <int ND>
__device__ __forceinline__ void th_tri_solve_column(const ThTriArgs &a, const double *__restrict__ r, double *__restrict__ z,
                                                    const double *__restrict__ w, int c, double &acc0, double &acc1) {
  const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
  const size_t cbase = (size_t)tile * 64 * a.nz + (c & 63);
  th_tri_forward_column(a, r, z, cbase, tw);
  th_tri_backward_column<ND>(a, z, w, cbase, tw, acc0, acc1);
}

__global__ __launch_bounds__(kBlock) void k_th_tri_solve(ThTriArgs a, const double *__restrict__ r, double *__restrict__ z) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= a.ncols) return;
  double a0 = 0.0, a1 = 0.0;
  th_tri_solve_column<0>(a, r, z, nullptr, c, a0, a1);
}

// the same solve with the Krylov dot products of its result fused in (the back substitution holds z in registers when it
// writes it): ND = 1 leaves the partial sums of z.z, ND = 2 those of z.w and z.z (in that order) over both fields, one per
// block, layout partials[j * gridDim.x + block]
template <int ND>
__global__ __launch_bounds__(kBlock) void k_th_tri_solve_dots(ThTriArgs a, const double *__restrict__ r, double *__restrict__ z,
                                                              const double *__restrict__ w, double *__restrict__ partials) {
  __shared__ double lds[kBlock / 64];
  const int c = blockIdx.x * kBlock + threadIdx.x;
  double acc0 = 0.0, acc1 = 0.0;
  if (c < a.ncols) th_tri_solve_column<ND>(a, r, z, w, c, acc0, acc1);
  // block-wide sums in a fixed order (lanes -> wave leaders -> thread 0)
  auto block_total = [&](double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < kBlock / 64; ++q) t += lds[q];
    return t;
  };
  const double t0 = block_total(acc0);
  if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  if (ND == 2) {
    const double t1 = block_total(acc1);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t1;
  }
}

// z = M^{-1} (A x) with the dot products of z, in ONE kernel (the role k_tri_solve_lds plays for the scalar system): the forward
// sweep of the block-Thomas solve forms its right-hand side -- the two rows of A x of its cell -- from the level's matrix rows
// (staged through LDS as in k_th_spmv_cols, the next level's pieces already in flight) and three levels of x kept in registers;
// A x is never written or read back (32 of the pair's 304 bytes per cell).  Row sums in the entry order of k_th_spmv_cols, the
// elimination in the order of th_tri_forward_column: the same bits as the two kernels.
// FDOT (round 5): the partial sums of fdot . (A x) over both fields as well, in row ND of `partials` -- the line search's F . (J p) /
// F . (J s) of a one-iteration BiCGStab solve out of the kernels that hold those rows anyway (Solver::ksp_bcgs, ls_fold)
// INVF (round 5, option th_inv_fly): the pivots' inverses are not read either -- the forward sweep holds the level's three blocks (the
// row it multiplies) and redoes the factorisation's step (th_factor_level: a division and 30 multiply-adds a level, the same bits
// as the stored inv and G) where it would read 32 bytes a cell.
template <int ND, bool FDOT = false, bool INVF = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MPP_TH_SOLVE_WAVES))) void k_th_spmv_tri(ThTriArgs a, const double *__restrict__ x, double *__restrict__ z,
                                                        const double *__restrict__ w, double *__restrict__ partials,
                                                        const double *__restrict__ fdot) {
#pragma clang fp contract(off)
  typedef double dpair __attribute__((ext_vector_type(2)));
  __shared__ dpair stage[kBlock / 64][2][64 * 3];
  __shared__ double lds[kBlock / 64];
  const int c = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc0 = 0.0, acc1 = 0.0, accf = 0.0;
  if (c - lane < a.ncols) {   // (the whole wave)
    const bool live = c < a.ncols;
    const int tile = c >> 6, tw = min(64, a.ncols - (tile << 6));
    const size_t cbase = (size_t)tile * 64 * a.nz + lane, n = (size_t)a.n;
    dpair *sP = stage[wave][0], *sT = stage[wave][1];
    dpair nP[3], nT[3];
    bool nstaged = false;
    auto prefetch = [&](int k) {   // the pieces of level k into registers, if the level lies where the column layout puts it
      nstaged = false;
      if (tw != 64 || !a.stage) return;
      const size_t i = cbase + (size_t)k * tw;
      const int len = 3 - (k == 0) - (k == a.nz - 1);
      const int base = a.crowptr[i - lane];
      const bool mine = a.crowptr[i - lane + 64] - base == 64 * len && a.crowptr[i] - base == lane * len;
      if (__builtin_amdgcn_ballot_w64(mine) != ~0ull) return;
      const dpair *gP = reinterpret_cast<const dpair *>(a.val + (size_t)2 * base),
                  *gT = reinterpret_cast<const dpair *>(a.val + (size_t)2 * a.nnzc + (size_t)2 * base);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < len) { nP[j] = __builtin_nontemporal_load(gP + j * 64 + lane); nT[j] = __builtin_nontemporal_load(gT + j * 64 + lane); }
      nstaged = true;
    };
    prefetch(0);
    // x of the levels k - 1, k, k + 1 (zero outside the column), the pivot's inverse of level k one level ahead
    double xPm = 0.0, xTm = 0.0, xP0 = 0.0, xT0 = 0.0, xPp = 0.0, xTp = 0.0;
    B2 inext{0, 0, 0, 0};
    if (live) {
      xP0 = x[cbase]; xT0 = x[n + cbase];
      if (a.nz > 1) { xPp = x[cbase + tw]; xTp = x[n + cbase + tw]; }
      if (!INVF) inext = b2_load(a.inv, n, cbase);
    }
    B2 g_prev{0, 0, 0, 0};
    double dp = 0.0, dt = 0.0;
    for (int k = 0; k < a.nz; ++k) {
      const size_t i = cbase + (size_t)k * tw;
      const int len = 3 - (k == 0) - (k == a.nz - 1);
      const bool staged = nstaged;
      if (staged) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < len) { sP[j * 64 + lane] = nP[j]; sT[j * 64 + lane] = nT[j]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (k + 1 < a.nz) prefetch(k + 1);
      B2 inv = inext;
      double xPn = 0.0, xTn = 0.0;
      if (live) {
        if (k + 2 < a.nz) { xPn = x[i + 2 * (size_t)tw]; xTn = x[n + i + 2 * (size_t)tw]; }
        if (!INVF && k + 1 < a.nz) inext = b2_load(a.inv, n, i + tw);
      }
      double sp = 0.0, st = 0.0;
      B2 lb{0, 0, 0, 0}, db{0, 0, 0, 0}, ub{0, 0, 0, 0};   // the row's blocks: lb is the elimination's multiplier; db, ub for INVF
      const bool lo = k > 0, up = k < a.nz - 1;
#define TH_ROW_PRODUCT(vP, vT, LEN)                                                         \
      {                                                                                     \
        int e = 0;                                                                          \
        if (lo) { const double p = vP[e] * xPm, q = vT[e] * xPm; sp += p; st += q; ++e; }   \
        { const double p = vP[e] * xP0, q = vT[e] * xP0; sp += p; st += q; ++e; }           \
        if (up) { const double p = vP[e] * xPp, q = vT[e] * xPp; sp += p; st += q; }        \
        e = LEN;                                                                            \
        if (lo) { const double p = vP[e] * xTm, q = vT[e] * xTm; sp += p; st += q; ++e; }   \
        { const double p = vP[e] * xT0, q = vT[e] * xT0; sp += p; st += q; ++e; }           \
        if (up) { const double p = vP[e] * xTp, q = vT[e] * xTp; sp += p; st += q; }        \
      }
      if (staged) {
        const double *vP = reinterpret_cast<const double *>(sP) + lane * 2 * len, *vT = reinterpret_cast<const double *>(sT) + lane * 2 * len;
        TH_ROW_PRODUCT(vP, vT, len)
        if (lo) lb = B2{vP[0], vP[len], vT[0], vT[len]};
        if (INVF) {
          const int d = lo;
          db = B2{vP[d], vP[d + len], vT[d], vT[d + len]};
          if (up) ub = B2{vP[d + 1], vP[d + 1 + len], vT[d + 1], vT[d + 1 + len]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the reads above precede the next level's writes)
        __builtin_amdgcn_wave_barrier();
      } else if (live) {
        const int r0 = a.crowptr[i], rl = a.crowptr[i + 1] - r0;
        const double *vP = a.val + (size_t)2 * r0, *vT = a.val + (size_t)2 * a.nnzc + (size_t)2 * r0;
        TH_ROW_PRODUCT(vP, vT, rl)
        if (lo) lb = B2{vP[0], vP[rl], vT[0], vT[rl]};
        if (INVF) {
          const int d = lo;
          db = B2{vP[d], vP[d + rl], vT[d], vT[d + rl]};
          if (up) ub = B2{vP[d + 1], vP[d + 1 + rl], vT[d + 1], vT[d + 1 + rl]};
        }
      }
#undef TH_ROW_PRODUCT
      if (live) {
        if (FDOT) { accf += fdot[i] * sp; accf += fdot[n + i] * st; }   // (the two rows of A x of this cell meet fdot)
        if (INVF) { B2 g; th_factor_level(lb, db, ub, g_prev, inv, g); g_prev = g; }
        th_forward_level(lb, inv, sp, st, dp, dt);
        z[i] = dp; z[n + i] = dt;
      }
      xPm = xP0; xTm = xT0; xP0 = xPp; xT0 = xTp; xPp = xPn; xTp = xTn;
    }
    if (live) th_tri_backward_column<ND>(a, z, w, cbase, tw, acc0, acc1);
  }
  auto block_total = [&](double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < kBlock / 64; ++q) t += lds[q];
    return t;
  };
  const double t0 = block_total(acc0);
  if (threadIdx.x == 0) partials[blockIdx.x] = t0;
  if (ND == 2) {
    const double t1 = block_total(acc1);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t1;
  }
  if (FDOT) {
    const double t2 = block_total(accf);
    if (threadIdx.x == 0) partials[(size_t)ND * gridDim.x + blockIdx.x] = t2;
  }
}

// y = A x of the coupled system on a vertical-only grid without an index: the 2N x 2N CSR matrix is, per soil column, block
// tridiagonal with 2 x 2 blocks whose entries sit at computed places (see the header: (P,P) at crp + s, (P,T) behind them, the heat
// rows 2 nnzc further on), and a row's columns are the cell below, the cell and the cell above in either field.  One lane per cell
// forms both of its rows -- 12 values, 6 operands -- in the entry order of the CSR kernels (products rounded one by one, no fused
// multiply-add): the same bits as k_spmv_c on the compact index, 128 instead of 152 bytes per cell.
// DOTS: the line search's pass -- x.x, f.(A x) and q.q (q may be null) reduced on the way, A x not written (the role k_spmv_c's DOTS
// variant plays on the compact index); a block then walks kThDotsChunks pieces of 256 rows so that the second stage sums a few
// thousand partial sums, not 10^5.  Partial sums: partials[j * gridDim.x + block], j = 0 (x.x), 1 (f.(A x)), 2 (q.q).
constexpr int kThDotsChunks = 8;
template <bool DOTS>
__global__ __launch_bounds__(kBlock) void k_th_spmv_cols(ThTriArgs a, const double *__restrict__ x, double *__restrict__ y,
                                                         const double *__restrict__ f, const double *__restrict__ q,
                                                         double *__restrict__ partials) {
#pragma clang fp contract(off)
  // A wave's 64 rows are the cells of one level of one column tile: their entries are one contiguous piece of `val` per block row
  // ([P rows; T rows]), 2 len doubles a row.  Read lane by lane a row is a 48-byte stride (a load instruction touches 24 cache
  // lines for 512 useful bytes: 0.73 ms = 3.7 TB/s at 512 x 512 x 80), so the piece goes through LDS: 16-byte loads in address
  // order, then every lane takes its row from the wave's own staging area (no workgroup barrier: nothing is shared across waves).
  typedef double dpair __attribute__((ext_vector_type(2)));   // two doubles, one 16-byte access
  __shared__ dpair stage[kBlock / 64][2][64 * 3];
  __shared__ double red[kBlock / 64];
  const size_t n = (size_t)a.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_tile = (size_t)64 * a.nz;
  double acc_xx = 0.0, acc_fw = 0.0, acc_qq = 0.0;
  constexpr int kChunks = DOTS ? kThDotsChunks : 1;
  for (int ch = 0; ch < kChunks; ++ch) {
    const size_t i = ((size_t)blockIdx.x * kChunks + ch) * kBlock + threadIdx.x;
    const size_t i0 = i - lane;
    if (i0 >= n) continue;   // (the whole wave)
    // cell i = (level k, column c) of tile t: i = t 64 nz + k tw + lane
    const int tile = (int)(i0 / per_tile), tw = min(64, a.ncols - (tile << 6));
    bool staged = false;
    int len = 0;
    if (a.stage && tw == 64 && i0 + 64 <= n) {   // (a full tile: the wave's rows are one level, all of the same length)
      const int k0 = (int)((i0 - (size_t)tile * per_tile) >> 6);
      len = 3 - (k0 == 0) - (k0 == a.nz - 1);
      const int base = a.crowptr[i0];
      const bool mine = a.crowptr[i0 + 64] - base == 64 * len && a.crowptr[i] - base == lane * len;
      if (__builtin_amdgcn_ballot_w64(mine) == ~0ull) {   // the pattern of the column layout, row for row: otherwise no staging
        const dpair *gP = reinterpret_cast<const dpair *>(a.val + (size_t)2 * base),
                    *gT = reinterpret_cast<const dpair *>(a.val + (size_t)2 * a.nnzc + (size_t)2 * base);
        dpair *sP = stage[wave][0], *sT = stage[wave][1];
        for (int j = 0; j < len; ++j) {
          sP[j * 64 + lane] = __builtin_nontemporal_load(gP + j * 64 + lane);
          sT[j * 64 + lane] = __builtin_nontemporal_load(gT + j * 64 + lane);
        }
        staged = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (i < n) {
      const int k = (int)((i - (size_t)tile * per_tile) / tw);
      const bool lo = k > 0, up = k < a.nz - 1;
      const double xPl = lo ? x[i - tw] : 0.0, xPd = x[i], xPu = up ? x[i + tw] : 0.0;
      const double xTl = lo ? x[n + i - tw] : 0.0, xTd = x[n + i], xTu = up ? x[n + i + tw] : 0.0;
      double sp = 0.0, st = 0.0;
#define TH_ROW_PRODUCT(vP, vT)                                                              \
      {                                                                                     \
        int e = 0;                                                                          \
        if (lo) { const double p = vP[e] * xPl, q_ = vT[e] * xPl; sp += p; st += q_; ++e; } \
        { const double p = vP[e] * xPd, q_ = vT[e] * xPd; sp += p; st += q_; ++e; }         \
        if (up) { const double p = vP[e] * xPu, q_ = vT[e] * xPu; sp += p; st += q_; }      \
        e = len;                                                                            \
        if (lo) { const double p = vP[e] * xTl, q_ = vT[e] * xTl; sp += p; st += q_; ++e; } \
        { const double p = vP[e] * xTd, q_ = vT[e] * xTd; sp += p; st += q_; ++e; }         \
        if (up) { const double p = vP[e] * xTu, q_ = vT[e] * xTu; sp += p; st += q_; }      \
      }
      if (staged) {   // (two copies of the arithmetic so that the LDS reads are LDS instructions, not flat ones)
        const double *sP = reinterpret_cast<const double *>(stage[wave][0]) + lane * 2 * len,
                     *sT = reinterpret_cast<const double *>(stage[wave][1]) + lane * 2 * len;
        TH_ROW_PRODUCT(sP, sT)
      } else {
        const int r0 = a.crowptr[i];
        len = a.crowptr[i + 1] - r0;
        const double *vP = a.val + (size_t)2 * r0, *vT = a.val + (size_t)2 * a.nnzc + (size_t)2 * r0;
        TH_ROW_PRODUCT(vP, vT)
      }
#undef TH_ROW_PRODUCT
      if (DOTS) {
        acc_xx += xPd * xPd; acc_xx += xTd * xTd;
        acc_fw += f[i] * sp; acc_fw += f[n + i] * st;
        if (q) { const double qp = q[i], qt = q[n + i]; acc_qq += qp * qp; acc_qq += qt * qt; }
      } else {
        __builtin_nontemporal_store(sp, y + i);
        __builtin_nontemporal_store(st, y + n + i);
      }
    }
    if (DOTS && staged) {   // (this piece's reads of the staging area precede the next piece's writes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (DOTS) {
    double t = th_block_sum(acc_xx, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    t = th_block_sum(acc_fw, red);
    if (threadIdx.x == 0) partials[(size_t)gridDim.x + blockIdx.x] = t;
    if (q) {
      t = th_block_sum(acc_qq, red);
      if (threadIdx.x == 0) partials[(size_t)2 * gridDim.x + blockIdx.x] = t;
    }
  }
}

inline dim3 nblk(long long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

}  // namespace

#define TH_LAUNCH(kernel, grid, ...) hipLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, stream, __VA_ARGS__)

void launch_th_tri_factor(const ThTriArgs &a, hipStream_t stream) {
  if (a.ncols > 0) TH_LAUNCH((k_th_tri_factor<false, 0>), nblk(a.ncols), a, (const double *)nullptr, (double *)nullptr, (const double *)nullptr, (double *)nullptr);
}
void launch_th_spmv_cols(const ThTriArgs &a, const double *x, double *y, hipStream_t stream) {
  if (a.n > 0) TH_LAUNCH(k_th_spmv_cols<false>, nblk(a.n), a, x, y, nullptr, nullptr, nullptr);
}
int th_spmv_dots_blocks(int n) { return (int)(((long long)n + (long long)kBlock * kThDotsChunks - 1) / ((long long)kBlock * kThDotsChunks)); }
int launch_th_spmv_cols_dots(const ThTriArgs &a, const double *x, const double *f, const double *q, double *partials, hipStream_t stream) {
  if (a.n <= 0) return 0;
  const dim3 grid((unsigned)th_spmv_dots_blocks(a.n));
  TH_LAUNCH(k_th_spmv_cols<true>, grid, a, x, nullptr, f, q, partials);
  return (int)grid.x;
}
void launch_th_tri_solve(const ThTriArgs &a, const double *r, double *z, hipStream_t stream) {
  if (a.ncols > 0) TH_LAUNCH(k_th_tri_solve, nblk(a.ncols), a, r, z);
}
int launch_th_spmv_tri(const ThTriArgs &a, const double *x, double *z, const double *w, double *partials, hipStream_t stream, const double *fdot) {
  if (a.ncols <= 0) return 0;
  const dim3 grid = nblk(a.ncols);
  if (a.inv_fly) {
    if (w && fdot) TH_LAUNCH((k_th_spmv_tri<2, true, true>), grid, a, x, z, w, partials, fdot);
    else if (w) TH_LAUNCH((k_th_spmv_tri<2, false, true>), grid, a, x, z, w, partials, (const double *)nullptr);
    else TH_LAUNCH((k_th_spmv_tri<1, false, true>), grid, a, x, z, w, partials, (const double *)nullptr);
  }
  else if (w && fdot) TH_LAUNCH((k_th_spmv_tri<2, true>), grid, a, x, z, w, partials, fdot);
  else if (w) TH_LAUNCH(k_th_spmv_tri<2>, grid, a, x, z, w, partials, (const double *)nullptr);
  else TH_LAUNCH(k_th_spmv_tri<1>, grid, a, x, z, w, partials, (const double *)nullptr);
  return (int)grid.x;
}
int launch_th_tri_factor_solve_dots(const ThTriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream) {
  if (a.ncols <= 0) return 0;
  const dim3 grid = nblk(a.ncols);
  if (w) TH_LAUNCH((k_th_tri_factor<true, 2>), grid, a, r, z, w, partials);
  else TH_LAUNCH((k_th_tri_factor<true, 1>), grid, a, r, z, w, partials);
  return (int)grid.x;
}
int launch_th_tri_solve_dots(const ThTriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream) {
  if (a.ncols <= 0) return 0;
  const dim3 grid = nblk(a.ncols);
  if (w) TH_LAUNCH(k_th_tri_solve_dots<2>, grid, a, r, z, w, partials);
  else TH_LAUNCH(k_th_tri_solve_dots<1>, grid, a, r, z, w, partials);
  return (int)grid.x;
}

// mode 0 residual, 2 residual + Jacobian, 3 PreSolve (accumulation terms of both equations into F)
void launch_th_assemble(int mode, const ThBcArgs &b, hipStream_t stream) {
  const ThArgs &a = b.t;
  if (a.n <= 0) return;
  const dim3 grid = nblk(a.n);
  if (mode == 3) { TH_LAUNCH(k_th_aux<3>, grid, a); return; }
  if (a.use_cols) {
    const bool ifc = a.density_type == kDenIFC67 && a.ie_type == kIEIFC67;
    const dim3 gc((unsigned)((a.ncols + kColsBlock - 1) / kColsBlock));
#define TH_COLS(M, E) hipLaunchKernelGGL((k_th_cols<M, E>), gc, dim3(kColsBlock), 0, stream, a)
    if (mode == 0) { if (ifc) TH_COLS(0, 1); else TH_COLS(0, -1); }
    else { if (ifc) TH_COLS(2, 1); else TH_COLS(2, -1); }
#undef TH_COLS
  } else if (mode == 0) { TH_LAUNCH(k_th_aux<0>, grid, a); TH_LAUNCH(k_th_rows<0>, grid, a); }
  else { TH_LAUNCH(k_th_aux<2>, grid, a); TH_LAUNCH(k_th_rows<2>, grid, a); }
  if (b.ncond_cells > 0) {
    if (mode == 0) TH_LAUNCH(k_th_bc<0>, nblk(b.ncond_cells), b);
    else TH_LAUNCH(k_th_bc<2>, nblk(b.ncond_cells), b);
  }
}

}  // namespace mpp
