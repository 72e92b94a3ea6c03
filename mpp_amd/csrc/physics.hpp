// physics.hpp -- constitutive laws and the two-point Darcy flux of the VSFM, as inlined device functions.
//
// Behavioural contract (what each function must reproduce) -- reference file:line
//   saturation / relative permeability  src/mpp/util/SaturationFunction.F90:747-857 (VG), 900-990 (BC), 1027-1140 (SBC)
//   water density                       src/mpp/util/EOSWaterMod.F90:80-99, 102-178, 181-344
//   viscosity                           src/mpp/util/EOSWaterMod.F90:568-586
//   porosity                            src/mpp/util/PorosityFunctionMod.F90:99-163 (constant and linear models)
//   flux + dflux                        src/mpp/ge/RichardsMod.F90:118-340
// Branch conditions (`pc < 0`, `p > p0`, `dphi >= 0`, seepage switch) are kept exactly; arithmetic is fp64.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "fastpow.hpp"

namespace mpp {

// x^y with a real exponent, x > 0, as the laws below need it: the table-driven fpow in device code (fastpow.hpp: ~100 instructions
// against the ~270 of the library pow, < 1 ulp), the math library on the host (set-up code and the host entry points for drivers)
__host__ __device__ inline double rpw(double x, double y, const PowTab &pt) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fpow(x, y, pt);
#else
  (void)pt;
  return pow(x, y);
#endif
}

constexpr double kPressureRef = 101325.0;     // MultiPhysicsProbConstants.F90:199
constexpr double kGravity = 9.80665;          // :201
constexpr double kFMWH2O = 18.01534;          // :202
constexpr double kDenH2O = 1.0e3;             // mpp_varcon.F90:17
constexpr double kViscosity = 8.904156e-4;    // EOSWaterMod.F90:582
// fp64 division is ~10 instructions around a quarter-rate v_rcp_f64 and the assembly kernels are issue-bound: divisions
// by constants are multiplications by the (correctly rounded) reciprocal, <= 1 ulp from the quotient
constexpr double kInvViscosity = 1.0 / kViscosity;
constexpr double kInvFMWH2O = 1.0 / kFMWH2O;

enum SatFunc : int { kSatVG = 0, kSatBC = 1, kSatSBC = 2,
                     kSatVGHalf = 3 /* compile-time specialisation only: van Genuchten with m == 1/2 in every cell */ };
enum DenType : int { kDenConstant = 1, kDenTGDPB01 = 2, kDenIFC67 = 3 };
enum CondType : int { kCondMassFlux = 504, kCondDirichlet = 505, kCondSeepage = 509 };

// IFC-67 density: everything that depends on the temperature alone.  In the VSFM the temperature is one number per problem
// (298.15 K unless set; RichardsODEPressureAuxType.F90:92), so five of the formulation's six real powers -- theta^18, theta^-6,
// theta^11, theta^19, (a6 - theta)^9 -- and the terms built from them are constants of the run: evaluated once on the host
// (make_eos), which leaves the march one real power per cell (zz^(-5/17)) instead of six.
struct Ifc67T {
  double theta, theta20, yy, x_t /* a3 yy^2 */, a4theta, u2, u7, t_poly /* theta (aa13 + aa14 theta) */, t_u8 /* u8 (a6 - theta) */,
         t_u4 /* aa16 u4 */, c21 /* 3 aa21 (a12 - theta) */, c21d /* 6 aa21 (a12 - theta) */;
};

// Problem-wide constants of the equation of state (temperature is uniform, 298.15 K unless set).
struct EosParams {
  int density_type;
  double dent;    // TGDPB01: a5*(1 - (t+a1)^2 (t+a2)/a3/(t+a4))
  double kcoef;   // TGDPB01: k0 + k1 t + k2 t^2
  double t_C;     // temperature in degC (IFC67)
  Ifc67T ifc;     // IFC67: the temperature-only part
};

// per-cell soil parameters as the kernels hold them in registers
struct Soil {
  double sat_res, alpha, m;        // m = vg_m (VG) or bc_lambda (BC/SBC)
  double pu, ps, b2, b3;           // SBC only
  double por;                      // porosity_base
  double por_pref, por_slope;      // POROSITY_LINEAR only (pressure_reference, lin_mod_slope)
  int sf;                          // this cell's saturation function (kernels that dispatch per cell)
};

// PorosityFunctionComputation (src/mpp/util/PorosityFunctionMod.F90:99-163): constant model, or linear in pressure
__host__ __device__ inline void porosity(const Soil &s, bool linear, double P, double &por, double &dpor) {
  if (linear) { por = s.por + (P - s.por_pref) * s.por_slope; dpor = s.por_slope; }
  else { por = s.por; dpor = 0.0; }
}

struct Aux {
  double P, kr, dkr, den, dden, sat, dsat;
};

__host__ __device__ inline EosParams make_eos(int density_type, double t_K) {
  EosParams e;
  e.density_type = density_type;
  const double a1 = -3.983035, a2 = 301.797, a3 = 522528.9, a4 = 69.34881, a5 = 999.974950;
  const double k0 = 50.74e-11, k1 = -0.326e-11, k2 = 0.00416e-11;
  double t_c = t_K - 273.15;
  double sq = (t_c + a1) * (t_c + a1);
  e.dent = a5 * (1.0 - sq * (t_c + a2) / a3 / (t_c + a4));
  e.kcoef = k0 + k1 * t_c + k2 * (t_c * t_c);
  e.t_C = t_c;
  {  // IFC-67, temperature-only terms (EOSWaterMod.F90:228-275; same names)
    const double aa13 = 1.522411790e-3, aa14 = 2.284279054e-2, aa15 = 2.421647003e02, aa16 = 1.269716088e-10, aa20 = 1.293441934e01,
                 aa21 = 1.308119072e-5;
    const double ia1 = 8.438375405e-1, ia2 = 5.362162162e-4, ia3 = 1.720000000e00, ia4 = 7.342278489e-2, ia6 = 6.537154300e-1,
                 ia7 = 1.150000000e-6, ia8 = 1.510800000e-5, ia9 = 1.418800000e-1, ia12 = 2.040000000e-1;
    const double tc1 = 647.3;
    Ifc67T &f = e.ifc;
    f.theta = (t_c + 273.15) * (1.0 / tc1);
    const double theta2x = f.theta * f.theta;
    const double theta18 = pow(f.theta, 18.0);
    f.theta20 = theta18 * theta2x;
    f.yy = 1.0 - ia1 * theta2x - ia2 * pow(f.theta, -6.0);
    f.x_t = ia3 * f.yy * f.yy;
    f.a4theta = ia4 * f.theta;
    f.u2 = 1.0 / (ia8 + pow(f.theta, 11.0));
    const double u4 = 1.0 / (ia7 + theta18 * f.theta);
    f.u7 = aa20 * theta18 * (ia9 + theta2x);
    const double u8 = aa15 * pow(ia6 - f.theta, 9.0);
    f.t_poly = f.theta * (aa13 + aa14 * f.theta);
    f.t_u8 = u8 * (ia6 - f.theta);
    f.t_u4 = aa16 * u4;
    f.c21 = 3.0 * aa21 * (ia12 - f.theta);
    f.c21d = 6.0 * aa21 * (ia12 - f.theta);
  }
  return e;
}

// IFC-67 liquid water density [kmol/m3] and d/dp at the problem's temperature (EOSWaterMod.F90:181-344 with the
// temperature-only terms taken from Ifc67T; same summation order as the formulation)
__host__ __device__ inline void density_ifc67_at(const Ifc67T &f, double p, double &dwmol, double &dwp, const PowTab &pt) {
  const double aa11 = 7.982692717e00, aa12 = -2.616571843e-2, aa17 = 2.074838328e-7, aa18 = 2.174020350e-8,
               aa19 = 1.105710498e-9, aa22 = 6.047626338e-14;
  const double a5 = 4.975858870e-2, a10 = 7.002753165e00, a11 = 2.995284926e-4;
  const double pc1 = 22.064e6, vc1 = 0.00317;
  const double upc1 = 1.0 / pc1, vc1mol = vc1 * kFMWH2O;
  const double beta = p * upc1;
  const double beta2x = beta * beta;
  double xx = f.x_t - 2.0 * (f.a4theta - a5 * beta);
  xx = sqrt(xx);
  const double zz = f.yy + xx;
  const double u0 = -5.0 / 17.0;
  const double u1 = aa11 * a5 * rpw(zz, u0, pt);
  const double u3 = aa17 + (2.0 * aa18 + 3.0 * aa19 * beta) * beta;
  const double ab = a10 + beta, ab2 = ab * ab;
  const double u5 = 1.0 / (ab2 * ab2);           // (a10 + beta)^(-4)
  const double u6 = a11 - 3.0 * u5;
  const double vr = u1 + aa12 + f.t_poly + f.t_u8 + f.t_u4 - f.u2 * u3 - u6 * f.u7 + (f.c21 + 4.0 * aa22 * beta / f.theta20) * beta2x;
  dwmol = 1.0 / (vr * vc1mol);
  const double zpp = a5 / xx;
  const double u9 = u0 * u1 / zz;
  const double vrpp = u9 * zpp - f.u2 * (2.0 * aa18 + 6.0 * aa19 * beta) - 12.0 * f.u7 * u5 / ab +
                      (f.c21d + 12.0 * aa22 * beta / f.theta20) * beta;
  const double cnv = -1.0 / (vc1mol * vr * vr);
  dwp = cnv * vrpp * upc1;
}

// DEN_T: -1 decide at run time; 0 the density type is not IFC-67 (constant / TGDPB01); 1 it is IFC-67.
// Compile-time pruning keeps the 60-register IFC-67 polynomial out of kernels that never need it.
template <int DEN_T = -1>
__host__ __device__ inline void density(const EosParams &e, double p, double &den, double &dden, const PowTab &pt) {
  if (DEN_T == 1) {
    density_ifc67_at(e.ifc, p, den, dden, pt);
    return;
  }
  if (e.density_type == kDenTGDPB01) {
    const double p0 = 101325.0;
    double kappa, dkappa_dp;
    if (p > p0) {  // strict: EOSWaterMod.F90:151
      kappa = 1.0 + e.kcoef * (p - p0);
      dkappa_dp = e.kcoef;
    } else {
      kappa = 1.0;
      dkappa_dp = 0.0;
    }
    den = e.dent * kappa * kInvFMWH2O;
    dden = (e.dent * dkappa_dp) * kInvFMWH2O;
  } else if (e.density_type == kDenConstant) {
    den = kDenH2O * kInvFMWH2O;
    dden = 0.0;
  } else if (DEN_T == -1) {
    density_ifc67_at(e.ifc, p, den, dden, pt);
  } else {
    den = dden = 0.0;  // unreachable: DEN_T == 0 promises constant or TGDPB01
  }
}

// Brooks-Corey relative permeability kr = Se^(2.5 + 2/lambda) on the branch where Se = (alpha pc)^(-lambda)
// (SaturationFunction.F90:960-975, 1100-1112): there Se^(2/lambda) = (alpha pc)^(-2), so the second real power of the
// reference is two squares, a square root and a division -- ~60 instructions instead of the ~250 of a pow(), within a
// few ulp of pow(Se, 2.5 + 2/lambda) (Se itself carries the rounding of the first power either way).
__host__ __device__ inline double bc_relperm(double Se, double pc_alpha) {
  return (Se * Se) * sqrt(Se) / (pc_alpha * pc_alpha);
}

// SatFunc_PressToSat + SatFunc_PressToRelPerm (Mualem) for one pressure; shares the common powers.
// SAT_T >= 0 fixes the saturation function at compile time (the run-time `sat_func_rt` is then ignored)
template <bool DERIV, int SAT_T = -1>
__host__ __device__ inline void sat_relperm(int sat_func_rt, const Soil &s, double press, double frac_liq, double &sat,
                                            double &dsat, double &kr, double &dkr, const PowTab &pt) {
  const int sat_func = SAT_T == kSatVGHalf ? (int)kSatVG : (SAT_T >= 0 ? SAT_T : sat_func_rt);
  const double pc = press - kPressureRef;
  const double sat_res = s.sat_res, alpha = s.alpha;
  if (sat_func == kSatVG) {
    if (pc < 0.0) {
      const double t = -alpha * pc;
      // m == 1/2 (n == 2, the reference's celia/vchannel soil): every real power below is a square, a square
      // root or their reciprocal -- exact or correctly rounded, so results stay within an ulp or two of pow();
      // 1/(1+t^2) is formed once and serves Se = (1+t^2)^(-1/2), AA = t^2/(1+t^2) and 1/Se = (1+t^2) Se
      const bool half = (SAT_T == kSatVGHalf) ? true : (s.m == 0.5);
      if (half) {
        const double a2 = t * t;
        const double one_plus = 1.0 + a2;
        const double r = 1.0 / one_plus;
        const double Se = sqrt(r);
        sat = sat_res + (1.0 - sat_res) * Se;
        const double AA = a2 * r;
        const double AAm = sqrt(AA);
        const double BB = 1.0 - AAm;
        const double q = sqrt(Se);
        kr = q * BB * BB;
        if (DERIV) {
          const double dSe_dpc = -(Se * AA) / pc;          // m n == 1
          dsat = (1.0 - sat_res) * dSe_dpc;
          const double p1 = Se * q;                        // Se^(1/m - 1/2)
          const double p2 = 1.0 / AAm;                     // AA^(m - 1)
          const double dkr_dSe = 0.5 * kr * (one_plus * Se) + 2.0 * p1 * p2 * BB;
          dkr = dkr_dSe * dSe_dpc;
        }
      } else {
        const double mm = s.m;
        const double nn = 1.0 / (1.0 - mm);  // SatFunc_Set_VG, SaturationFunction.F90:157
        const double pc_alpha_n = rpw(t, nn, pt);
        const double one_plus = 1.0 + pc_alpha_n;
        const double Se = rpw(one_plus, -mm, pt);
        sat = sat_res + (1.0 - sat_res) * Se;
        const double AA = pc_alpha_n / one_plus;
        const double AAm = rpw(AA, mm, pt);
        const double BB = 1.0 - AAm;
        const double q = sqrt(Se);
        kr = q * BB * BB;
        if (DERIV) {
          const double dSe_dpc = -mm * nn * Se * AA / pc;
          dsat = (1.0 - sat_res) * dSe_dpc;
          // Two of the reference's five real powers are quotients of values already at hand (a pow() is ~250 instructions
          // here, a division 19): Se^(1/m - 1/2) = Se^(1/m) / sqrt(Se) with Se^(1/m) = 1 / (1 + (alpha pc)^n), and
          // AA^(m - 1) = AA^m / AA.  Within a few ulp of the pow() results (SaturationFunction.F90:836-846 as written).
          const double p1 = 1.0 / (one_plus * q);
          const double p2 = AAm / AA;
          const double dkr_dSe = 0.5 * kr / Se + 2.0 * p1 * p2 * BB;
          dkr = dkr_dSe * dSe_dpc;
        }
      }
    } else {
      sat = 1.0; kr = 1.0;
      if (DERIV) { dsat = 0.0; dkr = 0.0; }
    }
  } else if (sat_func == kSatBC) {
    const double lambda = s.m;
    const double pc_alpha = -alpha * pc;
    if (pc_alpha > 1.0) {
      const double Se = rpw(pc_alpha, -lambda, pt);
      sat = sat_res + (1.0 - sat_res) * Se;
      const double ex = 2.5 + 2.0 / lambda;
      kr = bc_relperm(Se, pc_alpha);
      if (DERIV) {
        const double dSe_dpc = -lambda * Se / pc;
        dsat = (1.0 - sat_res) * dSe_dpc;
        dkr = (ex * kr / Se) * dSe_dpc;
      }
    } else {
      sat = 1.0; kr = 1.0;
      if (DERIV) { dsat = 0.0; dkr = 0.0; }
    }
    kr = frac_liq * kr;  // SaturationFunction.F90:987-988 (Brooks-Corey only)
    if (DERIV) dkr = frac_liq * dkr;
  } else {  // smoothed Brooks-Corey (SaturationFunction.F90:1027-1140): Brooks-Corey below p_u, a cubic in (pc - p_s) up to p_s
    // (one call site for the two branches' real powers, selected per lane, was measured: 2.29 / 2.58 ms against 2.21 / 2.34 ms
    // for the residual + Jacobian march with class-table / per-cell soils -- most waves sit on one branch; not kept)
    const double lambda = s.m;
    const double ex = 2.5 + 2.0 / lambda;
    if (pc <= s.pu) {
      const double pc_alpha = -alpha * pc;
      const double Se = rpw(pc_alpha, -lambda, pt);
      sat = sat_res + (1.0 - sat_res) * Se;
      kr = bc_relperm(Se, pc_alpha);
      if (DERIV) {
        const double dSe_dpc = -lambda * Se / pc;
        dsat = (1.0 - sat_res) * dSe_dpc;
        dkr = (ex * kr / Se) * dSe_dpc;
      }
    } else if (pc < s.ps) {
      const double deltaPc = pc - s.ps;
      const double Se = 1.0 + deltaPc * deltaPc * (s.b2 + deltaPc * s.b3);
      sat = sat_res + (1.0 - sat_res) * Se;
      kr = rpw(Se, ex, pt);
      if (DERIV) {
        const double dSe_dpc = deltaPc * (2 * s.b2 + 3 * deltaPc * s.b3);
        dsat = (1.0 - sat_res) * dSe_dpc;
        dkr = (ex * kr / Se) * dSe_dpc;
      }
    } else {
      sat = 1.0; kr = 1.0;
      if (DERIV) { dsat = 0.0; dkr = 0.0; }
    }
  }
}

// RichODEPressureAuxVarCompute (RichardsODEPressureAuxType.F90:237-294)
template <bool DERIV, int SAT_T = -1, int DEN_T = -1>
__host__ __device__ inline Aux auxvar_compute(int sat_func, const EosParams &e, const Soil &s, double press,
                                              double frac_liq, const PowTab &pt) {
  Aux a;
  a.P = press;
  a.dsat = 0.0; a.dkr = 0.0;
  sat_relperm<DERIV, SAT_T>(sat_func, s, press, frac_liq, a.sat, a.dsat, a.kr, a.dkr, pt);
  density<DEN_T>(e, press, a.den, a.dden, pt);
  return a;
}

// geometry of one connection as the kernels hold it (computed once on the host from
// area, dist_up, dist_dn, unit_vec and the two cells' permeabilities, RichardsMod.F90:236-272)
struct ConnGeom {
  double Dq, upweight, dist_gravity, area;
};

// RichardsFlux_Internal (RichardsMod.F90:262-336).  `bc_type` = 0 for internal connections.
// Returns flux; with DERIV also dflux_dP_up / dflux_dP_dn, which are the negatives of d(flux)/dP (:326-334).
template <bool DERIV>
__host__ __device__ inline double richards_flux(const Aux &up, const Aux &dn, const ConnGeom &g, int bc_type,
                                                double &dflux_up, double &dflux_dn) {
  const double upweight = g.upweight;
  const double den_ave = upweight * up.den + (1.0 - upweight) * dn.den;
  const double gravityterm = (upweight * up.den + (1.0 - upweight) * dn.den) * kFMWH2O * g.dist_gravity;
  double dphi = up.P - dn.P + gravityterm;
  bool seepage_update = false;
  if (bc_type == kCondSeepage) {
    if (dphi > 0.0 && up.P <= kPressureRef) seepage_update = true;
  }
  if (seepage_update) dphi = 0.0;
  const bool upwind_up = (dphi >= 0.0);
  const double ukvr = (upwind_up ? up.kr : dn.kr) * kInvViscosity;
  const bool massflux = (bc_type == kCondMassFlux);
  const double v_darcy = massflux ? 0.0 : -g.Dq * ukvr * dphi;
  const double q = v_darcy * g.area;
  const double flux = q * den_ave;
  if (DERIV) {
    const double dden_ave_up = upweight * up.dden;
    const double dden_ave_dn = (1.0 - upweight) * dn.dden;
    const double dgrav_up = upweight * g.dist_gravity * kFMWH2O;
    const double dgrav_dn = (1.0 - upweight) * g.dist_gravity * kFMWH2O;
    const double dphi_up = 1.0 + dgrav_up * up.dden;
    double dphi_dn = -1.0 + dgrav_dn * dn.dden;
    if (seepage_update) dphi_dn = 0.0;
    // dvis_dP == 0 (EOSWaterMod.F90:583): the kr/(vis*vis)*dvis_dP term is an exact zero
    const double dukvr_up = upwind_up ? up.dkr * kInvViscosity : 0.0;
    const double dukvr_dn = upwind_up ? 0.0 : dn.dkr * kInvViscosity;
    const double dq_up = g.Dq * (dukvr_up * dphi + ukvr * dphi_up) * g.area;
    const double dq_dn = g.Dq * (dukvr_dn * dphi + ukvr * dphi_dn) * g.area;
    if (massflux) {
      dflux_up = 0.0; dflux_dn = 0.0;
    } else {
      dflux_up = dq_up * den_ave - q * dden_ave_up;
      dflux_dn = dq_dn * den_ave - q * dden_ave_dn;
    }
  }
  return flux;
}

// host-side geometry precompute (same expression order as RichardsMod.F90:236-272)
inline ConnGeom make_conn_geom(bool internal_conn, double area, double dist_up, double dist_dn, const double *u,
                               const double *perm_up, const double *perm_dn) {
  ConnGeom g;
  double pu = std::fabs(u[0]) * perm_up[0] + std::fabs(u[1]) * perm_up[1] + std::fabs(u[2]) * perm_up[2];
  double pd = std::fabs(u[0]) * perm_dn[0] + std::fabs(u[1]) * perm_dn[1] + std::fabs(u[2]) * perm_dn[2];
  if (internal_conn) {
    g.upweight = dist_up / (dist_up + dist_dn);
    g.Dq = (pu * pd) / (dist_up * pd + dist_dn * pu);
  } else {  // COND_DIRICHLET, COND_MASS_FLUX, COND_SEEPAGE_BC (:249-252)
    g.upweight = 0.0;
    g.Dq = pd / (dist_up + dist_dn);
  }
  double grav_vec[3] = {0.0, 0.0, -kGravity};
  double udotg = u[0] * grav_vec[0] + u[1] * grav_vec[1] + u[2] * grav_vec[2];
  g.dist_gravity = (dist_up + dist_dn) * udotg;
  g.area = area;
  return g;
}

}  // namespace mpp
