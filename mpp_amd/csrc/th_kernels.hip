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

__device__ __forceinline__ bool th_active(const ThArgs &a, size_t i) { return a.active ? a.active[i] != 0 : true; }

__device__ __forceinline__ Soil th_load_soil(const ThArgs &a, size_t i) {
  Soil s;
  s.sat_res = a.sat_res[i]; s.alpha = a.alpha[i]; s.m = a.m[i]; s.por = a.por[i];
  if (a.sat_func == kSatSBC) { s.pu = a.sbc_pu[i]; s.ps = a.sbc_ps[i]; s.b2 = a.sbc_b2[i]; s.b3 = a.sbc_b3[i]; }
  else { s.pu = s.ps = s.b2 = s.b3 = 0.0; }
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
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t n = (size_t)a.n;
  if (i >= n) return;
  const bool act = th_active(a, i);
  const Soil s = th_load_soil(a, i);
  const ThSoil ts{a.tc_wet[i], a.tc_dry[i], a.talpha[i]};
  const double P = a.x[i], T = a.x[n + i];
  const ThAux x = th_auxvar_compute<true>(a.sat_func, a.density_type, a.ie_type, s, ts, P, T, true);  // every cell (GE:942-961)
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
  if (act) {
    fm = s.por * x.den_m * x.sat * vol * a.dtInv;                                                         // GE(Richards):1603-1634
    fe = (s.por * x.den_e * x.sat * x.ul + (1.0 - s.por) * dsoil * csoil * (T - 273.15)) * vol * a.dtInv;  // GE(heat):1174-1219
  }
  if (PRE) { a.F[i] = fm; a.F[n + i] = fe; return; }
  a.F[i] = fm - a.accum_prev[i];
  a.F[n + i] = fe - a.accum_prev[n + i];
  if (JAC) {
    const int r0 = a.crowptr[i], len = a.crowptr[i + 1] - r0;
    const int sPP = r0 + a.cdiag[i], sPT = sPP + len, sTP = 2 * a.nnzc + sPP, sTT = sTP + len;
    double dPP = 1.0, dPT = 1.0, dTP = 1.0, dTT = 1.0;  // inactive cell: 1 on all four diagonals (GE:1682; 2501; 2136; 1283)
    if (act) {
      dPP = (s.por * x.ddenP_m * x.sat + s.por * x.den_m * x.dsat) * vol * a.dtInv;
      dPT = (s.por * x.ddenT_m * x.sat) * vol * a.dtInv;
      dTP = (s.por * x.ddenP_e * x.sat * x.ul + s.por * x.den_e * x.dsat * x.ul + s.por * x.den_e * x.sat * x.dul_dP) * vol * a.dtInv;
      dTT = ((s.por * x.ddenT_e * x.sat * x.ul + s.por * x.den_e * x.sat * x.dul_dT) + (1.0 - s.por) * dsoil * csoil) * vol * a.dtInv;
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
// pass 3: boundary connections of both equations, grouped by cell.  Mass equation: Dirichlet / seepage / mass-flux
// (GE:1815-1877, 2076-2139); heat equation: COND_DIRICHLET (heat :1391-1452, 1627-1680, 2283-2370).  A boundary
// auxvar has the soil of its id_dn cell; the mass equation's carries a temperature, the heat equation's a pressure
// (assigned by the caller, as the reference drivers do).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_th_bc(ThBcArgs b) {
  constexpr bool JAC = (MODE == 2);
  const ThArgs &a = b.t;
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= b.ncond_cells) return;
  const size_t cell = b.cc_cell[j];
  const size_t n = (size_t)a.n;
  if (!th_active(a, cell)) return;
  const Soil s = th_load_soil(a, cell);
  const ThSoil ts{a.tc_wet[cell], a.tc_dry[cell], a.talpha[cell]};
  const ThAux in = th_load_aux(a, cell);
  double fm = a.F[cell], fe = a.F[n + cell];
  double dPP = 0.0, dTP = 0.0, dTT = 0.0;
  for (int e = b.cc_ptr[j]; e < b.cc_ptr[j + 1]; ++e) {
    const int t = b.type[e];
    const int ctype = t & 0xffff, swap = (t >> 16) & 1, heat = (t >> 20) & 1;
    ConnGeom g;
    g.Dq = b.Dq[e]; g.upweight = 0.0; g.dist_gravity = b.dgrav[e]; g.area = b.area[e];
    if (!heat) {
      const double pb = (ctype == kCondMassFlux) ? in.P : b.value[e];
      const ThAux bc = th_auxvar_compute<true>(a.sat_func, a.density_type, a.ie_type, s, ts, pb, b.other[e], true);
      double Jup = 0.0, Jdn = 0.0, flux, jdn_outer;
      if (!swap) { flux = richards_flux<JAC>(mass_view(bc), mass_view(in), g, ctype, Jup, Jdn); jdn_outer = Jdn; }
      else { flux = -richards_flux<JAC>(mass_view(in), mass_view(bc), g, ctype, Jup, Jdn); jdn_outer = -Jup; }
      fm += flux;
      if (JAC) dPP += -jdn_outer;
    } else {
      const ThAux bc = th_auxvar_compute<true>(a.sat_func, a.density_type, a.ie_type, s, ts, b.other[e], b.value[e], false);
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

inline dim3 nblk(long long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

}  // namespace

#define TH_LAUNCH(kernel, grid, ...) hipLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, stream, __VA_ARGS__)

// mode 0 residual, 2 residual + Jacobian, 3 PreSolve (accumulation terms of both equations into F)
void launch_th_assemble(int mode, const ThBcArgs &b, hipStream_t stream) {
  const ThArgs &a = b.t;
  if (a.n <= 0) return;
  const dim3 grid = nblk(a.n);
  if (mode == 3) { TH_LAUNCH(k_th_aux<3>, grid, a); return; }
  if (mode == 0) { TH_LAUNCH(k_th_aux<0>, grid, a); TH_LAUNCH(k_th_rows<0>, grid, a); }
  else { TH_LAUNCH(k_th_aux<2>, grid, a); TH_LAUNCH(k_th_rows<2>, grid, a); }
  if (b.ncond_cells > 0) {
    if (mode == 0) TH_LAUNCH(k_th_bc<0>, nblk(b.ncond_cells), b);
    else TH_LAUNCH(k_th_bc<2>, nblk(b.ncond_cells), b);
  }
}

}  // namespace mpp
