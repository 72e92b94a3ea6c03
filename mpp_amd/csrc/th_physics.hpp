// th_physics.hpp -- constitutive laws and fluxes of the coupled thermal-hydrology system as inlined
// device functions: water density with temperature dependence, internal energy / enthalpy, the heat-equation
// auxvar, the temperature derivative of the Darcy flux and the advective + conductive energy flux.
//
// Behavioural contract -- reference file:line
//   Density (p, T) + d/dp + d/dT          src/mpp/util/EOSWaterMod.F90:38-344
//   InternalEnergyAndEnthalpy             src/mpp/util/EOSWaterMod.F90:589-707 (EnthalpyIFC67 :346-566)
//   ThermEnthalpyAuxVarCompute            src/mpp/auxvar/ThermalEnthalpySoilAuxType.F90:219-278
//   RichardsFluxDerivativeWrtTemperature  src/mpp/ge/RichardsMod.F90:343-583
//   ThermalEnthalpyFlux (+ d/dP)          src/mpp/ge/ThermalEnthalpyMod.F90:27-333
#pragma once

#include "physics.hpp"

namespace mpp {

enum IntEnergyType : int { kIEConstant = 1, kIEIFC67 = 2 };   // EOSWaterMod.F90:23-24

// x**m with an integer exponent, evaluated by binary exponentiation as a Fortran compiler does (not pow())
__host__ __device__ inline double powi(double x, int m) {
  unsigned n = (unsigned)m;
  double y = (n & 1u) ? x : 1.0;
  while (n >>= 1) {
    x = x * x;
    if (n & 1u) y *= x;
  }
  return y;
}

// Real powers with integer-valued exponents (theta**18.d0, theta**(-6.d0), ...): the kernels evaluate them as multiplication
// chains (LIB = false: a handful of multiplications instead of ~250 instructions of pow() each, within a few ulp of it; 29 -> 5
// pow calls per cell); LIB = true goes through the math library, as the reference's compiled code does -- what the host entry
// points for drivers use (mppgpu_host_*), whose callers may difference these functions numerically (th_mms: a 1e-6 central
// difference of the enthalpy turns a last-bit difference into 1e-9 of the manufactured heat source).
template <bool LIB>
__host__ __device__ inline double rpow(double x, int n) {
  if (LIB) return pow(x, (double)n);
  return n >= 0 ? powi(x, n) : 1.0 / powi(x, -n);
}
// IFC-67 density [kmol/m3] with both derivatives (EOSWaterMod.F90:181-344); t in degC
template <bool LIB = false>
__host__ __device__ inline void density_ifc67_pt(double t, double p, double &dwmol, double &dwp, double &dwt, const PowTab &pt) {
  const double aa11 = 7.982692717e00, aa12 = -2.616571843e-2, aa13 = 1.522411790e-3, aa14 = 2.284279054e-2,
               aa15 = 2.421647003e02, aa16 = 1.269716088e-10, aa17 = 2.074838328e-7, aa18 = 2.174020350e-8,
               aa19 = 1.105710498e-9, aa20 = 1.293441934e01, aa21 = 1.308119072e-5, aa22 = 6.047626338e-14;
  const double a1 = 8.438375405e-1, a2 = 5.362162162e-4, a3 = 1.720000000e00, a4 = 7.342278489e-2;
  const double a5 = 4.975858870e-2, a6 = 6.537154300e-1, a7 = 1.150000000e-6, a8 = 1.510800000e-5;
  const double a9 = 1.418800000e-1, a10 = 7.002753165e00, a11 = 2.995284926e-4, a12 = 2.040000000e-1;
  const double tc1 = 647.3, pc1 = 22.064e6, vc1 = 0.00317;
  const double utc1 = 1.0 / tc1, upc1 = 1.0 / pc1, vc1mol = vc1 * kFMWH2O;
  const double theta = (t + 273.15) * utc1;
  const double theta2x = theta * theta;
  const double theta18 = rpow<LIB>(theta, 18);
  const double theta20 = theta18 * theta2x;
  const double beta = p * upc1;
  const double beta2x = beta * beta;
  const double yy = 1.0 - a1 * theta2x - a2 * rpow<LIB>(theta, -6);
  double xx = a3 * yy * yy - 2.0 * (a4 * theta - a5 * beta);
  xx = sqrt(xx);
  const double zz = yy + xx;
  const double u0 = -5.0 / 17.0;
  const double u1 = aa11 * a5 * (LIB ? pow(zz, u0) : rpw(zz, u0, pt));
  const double u2 = 1.0 / (a8 + rpow<LIB>(theta, 11));
  const double u3 = aa17 + (2.0 * aa18 + 3.0 * aa19 * beta) * beta;
  const double u4 = 1.0 / (a7 + theta18 * theta);
  const double u5 = rpow<LIB>(a10 + beta, -4);
  const double u6 = a11 - 3.0 * u5;
  const double u7 = aa20 * theta18 * (a9 + theta2x);
  const double u8 = aa15 * rpow<LIB>(a6 - theta, 9);
  const double vr = u1 + aa12 + theta * (aa13 + aa14 * theta) + u8 * (a6 - theta) + aa16 * u4 - u2 * u3 - u6 * u7 +
                    (3.0 * aa21 * (a12 - theta) + 4.0 * aa22 * beta / theta20) * beta2x;
  dwmol = 1.0 / (vr * vc1mol);
  const double ypt = 6.0 * a2 * rpow<LIB>(theta, -7) - 2.0 * a1 * theta;
  const double zpt = ypt + (a3 * yy * ypt - a4) / xx;
  const double zpp = a5 / xx;
  const double u9 = u0 * u1 / zz;
  const double vrpt = u9 * zpt + aa13 + 2.0 * aa14 * theta - 10.0 * u8 - 19.0 * aa16 * u4 * u4 * theta18 +
                      11.0 * u2 * u2 * u3 * rpow<LIB>(theta, 10) - aa20 * u6 * (18.0 * a9 * theta18 + 20.0 * theta20) / theta -
                      (3.0 * aa21 + 80.0 * aa22 * beta / (theta20 * theta)) * beta2x;
  const double vrpp = u9 * zpp - u2 * (2.0 * aa18 + 6.0 * aa19 * beta) - 12.0 * u7 * u5 / (a10 + beta) +
                      (6.0 * aa21 * (a12 - theta) + 12.0 * aa22 * beta / theta20) * beta;
  const double cnv = -1.0 / (vc1mol * vr * vr);
  dwt = cnv * vrpt * utc1;
  dwp = cnv * vrpp * upc1;
}

// Density(p, t_K, type) -> den [kmol/m3], dden_dp, dden_dT (EOSWaterMod.F90:38-178)
template <bool LIB = false>
__host__ __device__ inline void density_pt(int density_type, double p, double t_K, double &den, double &dden_dp, double &dden_dT,
                                           const PowTab &pt) {
  if (density_type == kDenConstant) {
    den = kDenH2O / kFMWH2O; dden_dp = 0.0; dden_dT = 0.0;
  } else if (density_type == kDenTGDPB01) {
    const double a1 = -3.983035, a2 = 301.797, a3 = 522528.9, a4 = 69.34881, a5 = 999.974950;
    const double k0 = 50.74e-11, k1 = -0.326e-11, k2 = 0.00416e-11, p0 = 101325.0;
    const double t_c = t_K - 273.15;
    const double sq = (t_c + a1) * (t_c + a1);
    const double dent = a5 * (1.0 - sq * (t_c + a2) / a3 / (t_c + a4));
    double kappa, dkappa_dp, dkappa_dt;
    if (p > p0) {
      kappa = 1.0 + (k0 + k1 * t_c + k2 * (t_c * t_c)) * (p - p0);
      dkappa_dp = k0 + k1 * t_c + k2 * (t_c * t_c);
      dkappa_dt = (k1 + 2.0 * k2 * t_c) * (p - p0);
    } else { kappa = 1.0; dkappa_dp = 0.0; dkappa_dt = 0.0; }
    den = dent * kappa / kFMWH2O;
    const double d1 = -sq / a3 / (t_c + a4);
    const double d2 = -2.0 * (t_c + a1) * (t_c + a2) / a3 / (t_c + a4);
    const double d3 = sq * (t_c + a2) / a3 / ((t_c + a4) * (t_c + a4));
    const double ddent_dt = a5 * (d1 + d2 + d3);
    dden_dT = (ddent_dt * kappa + dent * dkappa_dt) / kFMWH2O;
    dden_dp = (0.0 * kappa + dent * dkappa_dp) / kFMWH2O;
  } else {
    density_ifc67_pt<LIB>(t_K - 273.15, p, den, dden_dp, dden_dT, pt);
  }
}

// EnthalpyIFC67 (EOSWaterMod.F90:346-566): hw [J/kmol], d/dp, d/dT; t in degC
template <bool LIB = false>
__host__ __device__ inline void enthalpy_ifc67(double t, double p, double &hw, double &hwp, double &hwt, const PowTab &pt) {
  const double aa[23] = {
      6.824687741e03, -5.422063673e02, -2.096666205e04, 3.941286787e04, -6.733277739e04, 9.902381028e04,
      -1.093911774e05, 8.590841667e04, -4.511168742e04, 1.418138926e04, -2.017271113e03, 7.982692717e00,
      -2.616571843e-2, 1.522411790e-3, 2.284279054e-2, 2.421647003e02, 1.269716088e-10, 2.074838328e-7,
      2.174020350e-8, 1.105710498e-9, 1.293441934e01, 1.308119072e-5, 6.047626338e-14};
  const double a1 = 8.438375405e-1, a2 = 5.362162162e-4, a3 = 1.720000000e00, a4 = 7.342278489e-2;
  const double a5 = 4.975858870e-2, a6 = 6.537154300e-1, a7 = 1.150000000e-6, a8 = 1.510800000e-5;
  const double a9 = 1.418800000e-1, a10 = 7.002753165e00, a11 = 2.995284926e-4, a12 = 2.040000000e-1;
  const double tc1 = 647.3, pc1 = 22.064e6, vc1 = 0.00317;
  const double utc1 = 1.0 / tc1, upc1 = 1.0 / pc1, vc1mol = vc1 * kFMWH2O;
  const double theta = (t + 273.15) * utc1;
  const double theta2x = theta * theta;
  const double theta18 = rpow<LIB>(theta, 18);
  const double theta20 = theta18 * theta2x;
  const double beta = p * upc1;
  const double beta2x = beta * beta;
  const double beta4 = beta2x * beta2x;
  const double yy = 1.0 - a1 * theta2x - a2 * rpow<LIB>(theta, -6);
  double xx = a3 * yy * yy - 2.0 * (a4 * theta - a5 * beta);
  xx = sqrt(xx);
  const double zz = yy + xx;
  const double u0 = -5.0 / 17.0;
  const double u1 = aa[11] * a5 * (LIB ? pow(zz, u0) : rpw(zz, u0, pt));
  const double ypt = 6.0 * a2 * rpow<LIB>(theta, -7) - 2.0 * a1 * theta;
  const double utheta = 1.0 / theta;
  const double term1 = aa[0] * theta;
  double term2 = -aa[1];
  double term2t = 0.0;
#pragma unroll
  for (int i = 3; i <= 10; i++) {
    const double tempreal = (double)(i - 2) * aa[i] * powi(theta, i - 1);
    term2t = term2t + tempreal * utheta * (double)(i - 1);
    term2 = term2 + tempreal;
  }
  const double v0_1 = u1 / a5;
  const double v2_1 = 17.0 * (zz / 29.0 - yy / 12.0) + 5.0 * theta * ypt / 12.0;
  const double v3_1 = a4 * theta - (a3 - 1.0) * theta * yy * ypt;
  const double v1_1 = zz * v2_1 + v3_1;
  const double term3 = v0_1 * v1_1;
  const double v1_2 = 9.0 * theta + a6;
  const double v20_2 = (a6 - theta);
  const double v2_2 = rpow<LIB>(v20_2, 9);
  const double theta19 = rpow<LIB>(theta, 19);
  const double v3_2 = a7 + 20.0 * theta19;
  const double v40_2 = a7 + theta19;
  const double v4_2 = 1.0 / (v40_2 * v40_2);
  const double term4p = aa[12] - aa[14] * theta2x + aa[15] * v1_2 * v2_2 + aa[16] * v3_2 * v4_2;
  const double term4 = term4p * beta;
  const double v1_3 = beta * (aa[17] + aa[18] * beta + aa[19] * beta2x);
  const double theta11 = rpow<LIB>(theta, 11);
  const double v2_3 = 12.0 * theta11 + a8;
  const double v4_3 = 1.0 / (a8 + theta11);
  const double v3_3 = v4_3 * v4_3;
  const double term5 = v1_3 * v2_3 * v3_3;
  const double v1_4 = rpow<LIB>(a10 + beta, -3) + a11 * beta;
  const double v3_4 = (17.0 * a9 + 19.0 * theta2x);
  const double v2_4 = aa[20] * theta18 * v3_4;
  const double term6 = v1_4 * v2_4;
  const double v1_5 = 21.0 * aa[22] / theta20 * beta4;
  const double v2_5 = aa[21] * a12 * beta2x * beta;
  const double term7 = v1_5 + v2_5;
  const double v1_6 = pc1 * vc1mol;
  hw = (term1 - term2 + term3 + term4 - term5 + term6 + term7) * v1_6;
  const double zpt = ypt + (a3 * yy * ypt - a4) / xx;
  const double zpp = a5 / xx;
  const double yptt = -2.0 * a1 - 42.0 * a2 / rpow<LIB>(theta, 8);
  const double dv2t = 17.0 * (zpt / 29.0 - ypt / 12.0) + 5.0 / 12.0 * (ypt + theta * yptt);
  const double dv3t = a4 - (a3 - 1.0) * (theta * yy * yptt + yy * ypt + theta * ypt * ypt);
  const double dv2p = 17.0 * zpp / 29.0;
  const double v4_1 = 5.0 * v1_1 / (17.0 * zz);
  const double term3t = v0_1 * (zz * dv2t + (v2_1 - v4_1) * zpt + dv3t);
  const double term3p = v0_1 * (zz * dv2p + (v2_1 - v4_1) * zpp);
  const double term4t = (-2.0 * aa[14] * theta + 9.0 * aa[15] * (v2_2 - v1_2 * v2_2 / v20_2) +
                         38.0 * theta18 * aa[16] * (10.0 * v4_2 - v3_2 * v4_2 / v40_2)) * beta;
  const double term5p = v3_3 * v2_3 * (aa[17] + 2.0 * aa[18] * beta + 3.0 * aa[19] * beta2x);
  const double theta10 = rpow<LIB>(theta, 10);
  const double term5t = v1_3 * (132.0 * v3_3 * theta10 - 22.0 * v2_3 * v3_3 * v4_3 * theta10);
  const double term6p = v2_4 * (a11 - 3.0 * rpow<LIB>(a10 + beta, -4));
  const double term6t = v1_4 * aa[20] * theta18 * (18.0 * v3_4 * utheta + 38.0 * theta);
  const double term7p = beta2x * (3.0 * aa[21] * a12 + 84.0 * aa[22] * beta / theta20);
  const double term7t = -420.0 * aa[22] * beta4 / (theta20 * theta);
  hwp = (term3p + term4p - term5p + term6p + term7p) * vc1mol;
  hwt = (aa[0] - term2t + term3t + term4t - term5t + term6t + term7t) * v1_6 * utc1;
}

struct IEOut { double U, H, dU_dT, dH_dT, dU_dP, dH_dP; };

// ---- what ONE CELL of the coupled system needs from the IFC-67 formulation, evaluated together (device kernels) ----
// ThermEnthalpyAuxVarCompute takes the density at P for the mass equation and density + enthalpy at Pc = max(P, PRESSURE_REF) for the
// heat equation, all at the cell's temperature: three evaluations of the same formulation that share every temperature-only term
// (the powers of theta, yy and its derivatives, u2, u4, u7, u8, the enthalpy's polynomials) and, at Pc, the pressure terms too.
// Same formulas as density_ifc67_pt / enthalpy_ifc67 above (EOSWaterMod.F90:181-566), term by term; what differs is that a quotient
// by a quantity that divides several terms is a product with its reciprocal formed once (1/theta, 1/theta^20, 1/xx, 1/zz,
// 1/(a10 + beta), 1/vr: 4 + 4 per pressure divisions instead of ~17 + ~17 + ~20 -- a double-precision division is ~10 instructions
// here, and the march that calls this is bound by instruction issue, round 3's PMC: profiles/r03_pmc_th_512x512x80.json), and
// negative integer powers are powers of that reciprocal.  Results agree with the functions above to a few units of the last place
// (tests/test_gpu_th.py compares the kernels with the oracle's libm evaluation).
struct Ifc67Temp {   // temperature-only part
  double theta, theta2x, theta10, theta11, theta18, theta19, theta20, itheta, it20, it21;
  double yy, ypt, yptt, a3yy2, a4theta, u2, u4, u7, u8, d6, p8, p9, tpoly;
  double h_term1, h_term2, h_term2t, h_term4p, h_t4t, h_v2_3, h_t5t, h_v2_4, h_t6t;
};
__host__ __device__ inline Ifc67Temp ifc67_temperature_terms(double t_c, bool enthalpy) {
  const double aa0 = 6.824687741e03, aa1 = -5.422063673e02;
  const double aah[8] = {3.941286787e04, -6.733277739e04, 9.902381028e04, -1.093911774e05, 8.590841667e04, -4.511168742e04, 1.418138926e04,
                         -2.017271113e03};   // aa[3 .. 10]
  const double aa12 = -2.616571843e-2, aa13 = 1.522411790e-3, aa14 = 2.284279054e-2, aa15 = 2.421647003e02, aa16 = 1.269716088e-10, aa20 = 1.293441934e01;
  const double a1 = 8.438375405e-1, a2 = 5.362162162e-4, a3 = 1.720000000e00, a4 = 7.342278489e-2;
  const double a6 = 6.537154300e-1, a7 = 1.150000000e-6, a8 = 1.510800000e-5, a9 = 1.418800000e-1;
  const double utc1 = 1.0 / 647.3;
  Ifc67Temp f;
  f.theta = (t_c + 273.15) * utc1;
  f.theta2x = f.theta * f.theta;
  const double th4 = f.theta2x * f.theta2x, th8 = th4 * th4;
  f.theta10 = th8 * f.theta2x;
  f.theta11 = f.theta10 * f.theta;
  f.theta18 = th8 * th8 * f.theta2x;
  f.theta19 = f.theta18 * f.theta;
  f.theta20 = f.theta18 * f.theta2x;
  f.itheta = 1.0 / f.theta;
  f.it20 = 1.0 / f.theta20;
  f.it21 = f.it20 * f.itheta;
  const double it2 = f.itheta * f.itheta, it6 = it2 * it2 * it2, it7 = it6 * f.itheta, it8 = it7 * f.itheta;
  f.yy = 1.0 - a1 * f.theta2x - a2 * it6;
  f.ypt = 6.0 * a2 * it7 - 2.0 * a1 * f.theta;
  f.yptt = -2.0 * a1 - 42.0 * a2 * it8;
  f.a3yy2 = a3 * f.yy * f.yy;
  f.a4theta = a4 * f.theta;
  f.u2 = 1.0 / (a8 + f.theta11);
  f.u4 = 1.0 / (a7 + f.theta19);
  f.u7 = aa20 * f.theta18 * (a9 + f.theta2x);
  f.d6 = a6 - f.theta;
  const double d2 = f.d6 * f.d6, d4 = d2 * d2;
  f.p8 = d4 * d4;
  f.p9 = f.p8 * f.d6;
  f.u8 = aa15 * f.p9;
  f.tpoly = aa12 + f.theta * (aa13 + aa14 * f.theta);
  if (enthalpy) {
    f.h_term1 = aa0 * f.theta;
    double term2 = -aa1, term2t = 0.0, pw = f.theta2x;   // theta^(i-1), i = 3 ..
#pragma unroll
    for (int i = 3; i <= 10; i++) {
      const double tempreal = (double)(i - 2) * aah[i - 3] * pw;
      term2t = term2t + tempreal * f.itheta * (double)(i - 1);
      term2 = term2 + tempreal;
      pw = pw * f.theta;
    }
    f.h_term2 = term2; f.h_term2t = term2t;
    const double v1_2 = 9.0 * f.theta + a6, v3_2 = a7 + 20.0 * f.theta19, v4_2 = f.u4 * f.u4;
    f.h_term4p = aa12 - aa14 * f.theta2x + aa15 * v1_2 * f.p9 + aa16 * v3_2 * v4_2;
    f.h_t4t = -2.0 * aa14 * f.theta + 9.0 * aa15 * (f.p9 - v1_2 * f.p8) + 38.0 * f.theta18 * aa16 * (10.0 * v4_2 - v3_2 * v4_2 * f.u4);
    f.h_v2_3 = 12.0 * f.theta11 + a8;
    const double v3_3 = f.u2 * f.u2;
    f.h_t5t = 132.0 * v3_3 * f.theta10 - 22.0 * f.h_v2_3 * v3_3 * f.u2 * f.theta10;
    const double v3_4 = 17.0 * a9 + 19.0 * f.theta2x;
    f.h_v2_4 = aa20 * f.theta18 * v3_4;
    f.h_t6t = aa20 * f.theta18 * (18.0 * v3_4 * f.itheta + 38.0 * f.theta);
  } else {
    f.h_term1 = f.h_term2 = f.h_term2t = f.h_term4p = f.h_t4t = f.h_v2_3 = f.h_t5t = f.h_v2_4 = f.h_t6t = 0.0;
  }
  return f;
}
// density [kmol/m3] with both derivatives at pressure p and, ENTH, the enthalpy [J/kmol] with both derivatives
template <bool ENTH>
__host__ __device__ inline void ifc67_at_pressure(const Ifc67Temp &f, double p, double &dwmol, double &dwp, double &dwt, double &hw,
                                                  double &hwp, double &hwt, const PowTab &pt) {
  const double aa11 = 7.982692717e00, aa13 = 1.522411790e-3, aa14 = 2.284279054e-2, aa16 = 1.269716088e-10, aa17 = 2.074838328e-7,
               aa18 = 2.174020350e-8, aa19 = 1.105710498e-9, aa20 = 1.293441934e01, aa21 = 1.308119072e-5, aa22 = 6.047626338e-14;
  const double a3 = 1.720000000e00, a4 = 7.342278489e-2, a5 = 4.975858870e-2, a9 = 1.418800000e-1, a10 = 7.002753165e00,
               a11 = 2.995284926e-4, a12 = 2.040000000e-1;
  const double pc1 = 22.064e6, vc1 = 0.00317, utc1 = 1.0 / 647.3, upc1 = 1.0 / pc1, vc1mol = vc1 * kFMWH2O;
  const double beta = p * upc1, beta2x = beta * beta;
  const double xx = sqrt(f.a3yy2 - 2.0 * (f.a4theta - a5 * beta));
  const double ixx = 1.0 / xx;
  const double zz = f.yy + xx;
  const double izz = 1.0 / zz;
  const double u0 = -5.0 / 17.0;
  const double w = rpw(zz, u0, pt);
  const double u1 = aa11 * a5 * w;
  const double u3 = aa17 + (2.0 * aa18 + 3.0 * aa19 * beta) * beta;
  const double iab = 1.0 / (a10 + beta), iab2 = iab * iab, u5 = iab2 * iab2;
  const double u6 = a11 - 3.0 * u5;
  const double vr = u1 + f.tpoly + f.u8 * f.d6 + aa16 * f.u4 - f.u2 * u3 - u6 * f.u7 + (3.0 * aa21 * (a12 - f.theta) + 4.0 * aa22 * beta * f.it20) * beta2x;
  const double ivr = 1.0 / (vr * vc1mol);
  dwmol = ivr;
  const double zpt = f.ypt + (a3 * f.yy * f.ypt - a4) * ixx;
  const double zpp = a5 * ixx;
  const double u9 = u0 * u1 * izz;
  const double vrpt = u9 * zpt + aa13 + 2.0 * aa14 * f.theta - 10.0 * f.u8 - 19.0 * aa16 * f.u4 * f.u4 * f.theta18 + 11.0 * f.u2 * f.u2 * u3 * f.theta10 -
                      aa20 * u6 * (18.0 * a9 * f.theta18 + 20.0 * f.theta20) * f.itheta - (3.0 * aa21 + 80.0 * aa22 * beta * f.it21) * beta2x;
  const double vrpp = u9 * zpp - f.u2 * (2.0 * aa18 + 6.0 * aa19 * beta) - 12.0 * f.u7 * u5 * iab +
                      (6.0 * aa21 * (a12 - f.theta) + 12.0 * aa22 * beta * f.it20) * beta;
  const double cnv = -(ivr * ivr) * vc1mol;   // -1 / (vc1mol vr^2)
  dwt = cnv * vrpt * utc1;
  dwp = cnv * vrpp * upc1;
  if (ENTH) {
    const double beta4 = beta2x * beta2x;
    const double v0_1 = aa11 * w;                                  // u1 / a5
    const double v2_1 = 17.0 * (zz * (1.0 / 29.0) - f.yy * (1.0 / 12.0)) + (5.0 / 12.0) * f.theta * f.ypt;
    const double v3_1 = f.a4theta - (a3 - 1.0) * f.theta * f.yy * f.ypt;
    const double v1_1 = zz * v2_1 + v3_1;
    const double term3 = v0_1 * v1_1;
    const double term4 = f.h_term4p * beta;
    const double v1_3 = beta * (aa17 + aa18 * beta + aa19 * beta2x);
    const double v3_3 = f.u2 * f.u2;
    const double term5 = v1_3 * f.h_v2_3 * v3_3;
    const double v1_4 = iab2 * iab + a11 * beta;
    const double term6 = v1_4 * f.h_v2_4;
    const double term7 = 21.0 * aa22 * f.it20 * beta4 + aa21 * a12 * beta2x * beta;
    const double v1_6 = pc1 * vc1mol;
    hw = (f.h_term1 - f.h_term2 + term3 + term4 - term5 + term6 + term7) * v1_6;
    const double dv2t = 17.0 * (zpt * (1.0 / 29.0) - f.ypt * (1.0 / 12.0)) + (5.0 / 12.0) * (f.ypt + f.theta * f.yptt);
    const double dv3t = a4 - (a3 - 1.0) * (f.theta * f.yy * f.yptt + f.yy * f.ypt + f.theta * f.ypt * f.ypt);
    const double dv2p = (17.0 / 29.0) * zpp;
    const double v4_1 = (5.0 / 17.0) * v1_1 * izz;
    const double term3t = v0_1 * (zz * dv2t + (v2_1 - v4_1) * zpt + dv3t);
    const double term3p = v0_1 * (zz * dv2p + (v2_1 - v4_1) * zpp);
    const double term4t = f.h_t4t * beta;
    const double term5p = v3_3 * f.h_v2_3 * (aa17 + 2.0 * aa18 * beta + 3.0 * aa19 * beta2x);
    const double term5t = v1_3 * f.h_t5t;
    const double term6p = f.h_v2_4 * u6;
    const double term6t = v1_4 * f.h_t6t;
    const double term7p = beta2x * (3.0 * aa21 * a12 + 84.0 * aa22 * beta * f.it20);
    const double term7t = -420.0 * aa22 * beta4 * f.it21;
    hwp = (term3p + f.h_term4p - term5p + term6p + term7p) * vc1mol;
    hwt = (6.824687741e03 - f.h_term2t + term3t + term4t - term5t + term6t + term7t) * v1_6 * utc1;
  }
}

// InternalEnergyAndEnthalpy (EOSWaterMod.F90:589-707); den and its derivatives in kg m^-3
template <bool LIB = false>
__host__ __device__ inline IEOut internal_energy_enthalpy(int itype, double P, double t_K, double den, double dden_dT, double dden_dP,
                                                          const PowTab &pt) {
  IEOut o;
  if (itype == kIEConstant) {
    const double u0 = (double)4.217f * 1.e3;   // `4.217 * 1.d3` in the reference: a single-precision literal
    double U = u0 * (t_K - 273.15);
    double dU_dT = u0, dU_dP = 0.0;
    double H = U + P / den;
    double dH_dT = dU_dT - P / (den * den) * dden_dT;
    double dH_dP = dU_dP + 1.0 / den - P / (den * den) * dden_dP;
    o.U = U * kFMWH2O; o.H = H * kFMWH2O; o.dU_dT = dU_dT * kFMWH2O; o.dH_dT = dH_dT * kFMWH2O;
    o.dU_dP = dU_dP; o.dH_dP = dH_dP * kFMWH2O;
  } else {
    double H, dH_dP, dH_dT;
    enthalpy_ifc67<LIB>(t_K - 273.15, P, H, dH_dP, dH_dT, pt);
    const double dm = den / kFMWH2O;
    o.H = H; o.dH_dP = dH_dP; o.dH_dT = dH_dT;
    o.U = H - P / dm;
    o.dU_dT = dH_dT + P / (dm * dm) * (dden_dT / kFMWH2O);
    o.dU_dP = dH_dP - 1.0 / dm + P / (dm * dm) * (dden_dP / kFMWH2O);
  }
  return o;
}

// the quantities of one cell the TH kernels exchange: `m` = the mass equation's view (density at P), `e` = the heat
// equation's view (density, enthalpy at max(P, PRESSURE_REF): ThermalEnthalpySoilAuxType.F90:250-251)
struct ThAux {
  double P, T, kr, dkr, sat, dsat;
  double den_m, ddenP_m, ddenT_m;
  double den_e, ddenP_e, ddenT_e;
  double ul, hl, dul_dP, dhl_dP, dul_dT, dhl_dT;
  double tc, dtc_dP;   // thermal conductivity and d/dP
};

struct ThSoil { double tc_wet, tc_dry, talpha; };

// RichODEPressureAuxVarCompute with the cell temperature + ThermEnthalpyAuxVarCompute
// EOS_T = 1: IFC-67 density and IFC-67 enthalpy, known at compile time (the three evaluations of a cell share their terms: see
// ifc67_temperature_terms; the march instantiated that way carries no other formulation); EOS_T = -1: as density_type / ie_type say
template <bool DERIV, int EOS_T = -1>
__host__ __device__ inline ThAux th_auxvar_compute(int sat_func, int density_type, int ie_type, const Soil &s, const ThSoil &ts,
                                                   double P, double T, bool mass_view, const PowTab &pt) {
  ThAux a;
  a.P = P; a.T = T;
  a.dsat = 0.0; a.dkr = 0.0;
  sat_relperm<true>(sat_func, s, P, 1.0, a.sat, a.dsat, a.kr, a.dkr, pt);
  const double Pc = (P < kPressureRef) ? kPressureRef : P;
  a.den_m = a.ddenP_m = a.ddenT_m = 0.0;
  if (EOS_T == 1) {
    // the three IFC-67 evaluations of a cell at once (see ifc67_temperature_terms)
    const Ifc67Temp f = ifc67_temperature_terms(T - 273.15, true);
    double H, dH_dP, dH_dT, unused;
    ifc67_at_pressure<true>(f, Pc, a.den_e, a.ddenP_e, a.ddenT_e, H, dH_dP, dH_dT, pt);
    if (mass_view) {
      if (Pc == P) { a.den_m = a.den_e; a.ddenP_m = a.ddenP_e; a.ddenT_m = a.ddenT_e; }
      else ifc67_at_pressure<false>(f, P, a.den_m, a.ddenP_m, a.ddenT_m, unused, unused, unused, pt);
    }
    // InternalEnergyAndEnthalpy, IFC-67 branch (EOSWaterMod.F90:660-707) on dm = den_e [kmol/m3]
    const double idm = 1.0 / a.den_e, pidm2 = Pc * (idm * idm);
    a.hl = H; a.dhl_dP = dH_dP; a.dhl_dT = dH_dT;
    a.ul = H - Pc * idm;
    a.dul_dT = dH_dT + pidm2 * a.ddenT_e;
    a.dul_dP = dH_dP - idm + pidm2 * a.ddenP_e;
  } else {
    density_pt(density_type, Pc, T, a.den_e, a.ddenP_e, a.ddenT_e, pt);
    if (mass_view) {
      if (Pc == P) { a.den_m = a.den_e; a.ddenP_m = a.ddenP_e; a.ddenT_m = a.ddenT_e; }   // same arguments, same value
      else density_pt(density_type, P, T, a.den_m, a.ddenP_m, a.ddenT_m, pt);
    }
    const IEOut ie = internal_energy_enthalpy(ie_type, Pc, T, a.den_e * kFMWH2O, a.ddenT_e * kFMWH2O, a.ddenP_e * kFMWH2O, pt);
    a.ul = ie.U; a.hl = ie.H; a.dul_dT = ie.dU_dT; a.dhl_dT = ie.dH_dT; a.dul_dP = ie.dU_dP; a.dhl_dP = ie.dH_dP;
  }
  // Kersten number (sat + 1e-6)^alpha and its derivative alpha (sat + 1e-6)^(alpha - 1) dsat/dP: the second real power of the
  // reference (ThermalEnthalpySoilAuxType.F90:268-272) is the first one divided by its base
  const double kbase = a.sat + 1.e-6;
  const double Kel = rpw(kbase, ts.talpha, pt);
  const double dKel_dp = ts.talpha * (Kel / kbase) * a.dsat;
  a.tc = ts.tc_wet * Kel + ts.tc_dry * (1.0 - Kel);
  a.dtc_dP = (ts.tc_wet - ts.tc_dry) * dKel_dp;
  (void)DERIV;
  return a;
}

// RichardsFluxDerivativeWrtTemperature_Internal (RichardsMod.F90:431-583): the TRUE d(flux)/dT_up, d(flux)/dT_dn
// (dvis_dT == 0).  `up`/`dn` carry the density view wanted (mass or heat equation) in den / dden (= dden_dT here).
__host__ __device__ inline void richards_flux_dT(double P_up, double kr_up, double den_up, double ddenT_up, double P_dn,
                                                 double kr_dn, double den_dn, double ddenT_dn, const ConnGeom &g,
                                                 int bc_type, double &dflux_dT_up, double &dflux_dT_dn) {
  const double upweight = g.upweight;
  const double den_ave = upweight * den_up + (1.0 - upweight) * den_dn;
  const double gravityterm = (upweight * den_up + (1.0 - upweight) * den_dn) * kFMWH2O * g.dist_gravity;
  double dphi = P_up - P_dn + gravityterm;
  bool seepage_update = false;
  if (bc_type == kCondSeepage && dphi > 0.0 && P_up <= kPressureRef) seepage_update = true;
  if (seepage_update) dphi = 0.0;
  const double ukvr = (dphi >= 0.0) ? kr_up / kViscosity : kr_dn / kViscosity;
  const bool massflux = (bc_type == kCondMassFlux);
  const double v_darcy = massflux ? 0.0 : -g.Dq * ukvr * dphi;
  const double q = v_darcy * g.area;
  const double dden_ave_dT_up = upweight * ddenT_up;
  const double dden_ave_dT_dn = (1.0 - upweight) * ddenT_dn;
  const double dphi_dT_up = (upweight * g.dist_gravity * kFMWH2O) * ddenT_up;
  double dphi_dT_dn = ((1.0 - upweight) * g.dist_gravity * kFMWH2O) * ddenT_dn;
  if (seepage_update) dphi_dT_dn = 0.0;
  const double dq_dT_up = g.Dq * (0.0 * dphi + ukvr * dphi_dT_up) * g.area;
  const double dq_dT_dn = g.Dq * (0.0 * dphi + ukvr * dphi_dT_dn) * g.area;
  if (massflux) { dflux_dT_up = -0.0; dflux_dT_dn = -0.0; }
  else {
    dflux_dT_up = -(dq_dT_up * den_ave - q * dden_ave_dT_up);
    dflux_dT_dn = -(dq_dT_dn * den_ave - q * dden_ave_dT_dn);
  }
}

// harmonic / one-sided conductance over distance (ThermalEnthalpyMod.F90:110-130): internal vs COND_DIRICHLET boundary
__host__ __device__ inline double th_cond_over_dist(bool internal_conn, double tc_up, double tc_dn, double dist_up, double dist_dn) {
  return internal_conn ? (tc_up * tc_dn) / (dist_up * tc_dn + dist_dn * tc_up) : tc_dn / (dist_up + dist_dn);
}

// ThermalEnthalpyFlux (ThermalEnthalpyMod.F90:27-165)
template <bool DERIV>
__host__ __device__ inline double thermal_enthalpy_flux(const ThAux &up, const ThAux &dn, double mflux, double dmflux_dT_up,
                                                        double dmflux_dT_dn, double area, double dist_up, double dist_dn,
                                                        bool internal_conn, double &deflux_dT_up, double &deflux_dT_dn) {
  const double D = th_cond_over_dist(internal_conn, up.tc, dn.tc, dist_up, dist_dn);
  const double h = (mflux <= 0.0) ? up.hl : dn.hl;
  const double eflux = mflux * h + (-D * (up.T - dn.T) * area);
  if (DERIV) {
    const double dh_dT_up = (mflux < 0.0) ? up.dhl_dT : 0.0;
    const double dh_dT_dn = (mflux < 0.0) ? 0.0 : dn.dhl_dT;
    deflux_dT_up = dmflux_dT_up * h + mflux * dh_dT_up + (-D * area);
    deflux_dT_dn = dmflux_dT_dn * h + mflux * dh_dT_dn + (+D * area);
  }
  return eflux;
}

// ThermalEnthalpyFluxDerivativeWrtPressure (ThermalEnthalpyMod.F90:168-333); dmflux_dP_* are TRUE derivatives
__host__ __device__ inline void thermal_enthalpy_flux_dP(const ThAux &up, const ThAux &dn, double mflux, double dmflux_dP_up,
                                                         double dmflux_dP_dn, double area, double dist_up, double dist_dn,
                                                         bool internal_conn, double &deflux_dP_up, double &deflux_dP_dn) {
  const double D = th_cond_over_dist(internal_conn, up.tc, dn.tc, dist_up, dist_dn);
  const double h = (mflux <= 0.0) ? up.hl : dn.hl;
  const double dh_dP_up = (mflux < 0.0) ? up.dhl_dP : 0.0;
  const double dh_dP_dn = (mflux < 0.0) ? 0.0 : dn.dhl_dP;
  double dDk_dp_up, dDk_dp_dn;
  if (internal_conn) {
    dDk_dp_up = (D * D) / (up.tc * up.tc) * dist_up * up.dtc_dP;
    dDk_dp_dn = (D * D) / (dn.tc * dn.tc) * dist_dn * dn.dtc_dP;
  } else {
    dDk_dp_up = 0.0;
    dDk_dp_dn = 1.0 / (dist_up + dist_dn) * dn.dtc_dP;
  }
  deflux_dP_up = dmflux_dP_up * h + mflux * dh_dP_up + (-dDk_dp_up * (up.T - dn.T) * area);
  deflux_dP_dn = dmflux_dP_dn * h + mflux * dh_dP_dn + (-dDk_dp_dn * (up.T - dn.T) * area);
}

}  // namespace mpp
