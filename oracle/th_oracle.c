/*
 * th_oracle.c -- CPU ORACLE (test infrastructure only; see vsfm_oracle.h) for the coupled thermal-hydrology
 * system of equations: one Richards (mass) governing equation and one enthalpy-based
 * heat transport governing equation on the same mesh, coupled through their internal auxiliary variables.
 *
 * Restates, expression by expression (paths relative to /root/reference):
 *   src/mpp/soe/SystemOfEquationsTHType.F90               SOETHPreSolve :119-272, SOETHResidual :736-850,
 *                                                          SOETHJacobian :853-1003, SOETHPostSolve :275-302
 *   src/mpp/ge/GoveqnThermalEnthalpySoilType.F90           residual :1060-1089, 1174-1219, 1299-1498;
 *                                                          Jacobian :1092-1120, 1223-1296, 1501-1716;
 *                                                          d(energy)/dP :2083-2376
 *   src/mpp/ge/GoveqnRichardsODEPressureType.F90           d(mass)/dT :2473-2612
 *   src/mpp/ge/ThermalEnthalpyMod.F90                      ThermalEnthalpyFlux :27-165, ...DerivativeWrtPressure :168-333
 *   src/mpp/ge/RichardsMod.F90                             RichardsFluxDerivativeWrtTemperature :343-583
 *   src/mpp/auxvar/ThermalEnthalpySoilAuxType.F90          ThermEnthalpyAuxVarCompute :219-278
 *   src/mpp/util/EOSWaterMod.F90                           InternalEnergyAndEnthalpy :589-707, EnthalpyIFC67 :346-566
 * The mass equation itself (residual, d(mass)/dP) is the VSFM restatement in vsfm_oracle.c with the cell
 * temperatures taken from the heat equation.  The global unknown vector is field-blocked [P(1..N); T(1..N)]
 * (DMComposite), the Jacobian one 2N x 2N AIJ matrix, solved by the SNES/KSP/PC restated in vsfm_oracle.c.
 *
 * Only coupling through internal auxvars on a shared mesh is restated (what the reference's th tests use);
 * COND_DIRICHLET_FRM_OTR_GOVEQ (coupling of different meshes through boundaries) is not.
 */
#include "th_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vsfm_oracle_internal.h"

#define FAILT(msg) do { vo__set_error(msg); return -1; } while (0)

/* ------------------------------------------------------------------ */
/* EOS: internal energy and enthalpy                                    */
/* ------------------------------------------------------------------ */
#define H2O_CRITICAL_TEMPERATURE 647.3   /* EOSWaterMod.F90 */
#define H2O_CRITICAL_PRESSURE 22.064e6

/* x**m with an INTEGER exponent as gfortran evaluates it (libgcc __powidf2: binary exponentiation) -- not pow() */
static double powi(double x, int m) {
  unsigned n = m < 0 ? -(unsigned)m : (unsigned)m;
  double y = (n % 2) ? x : 1.0;
  while (n >>= 1) {
    x = x * x;
    if (n % 2) y *= x;
  }
  return m < 0 ? 1.0 / y : y;
}

/* EnthalpyIFC67, EOSWaterMod.F90:346-566 (t in centigrade, p in Pa; hw J/kmol) */
static void enthalpy_ifc67(double t, double p, double *hw, double *hwp, double *hwt) {
  static const double aa[23] = {
      6.824687741e03, -5.422063673e02, -2.096666205e04, 3.941286787e04, -6.733277739e04, 9.902381028e04,
      -1.093911774e05, 8.590841667e04, -4.511168742e04, 1.418138926e04, -2.017271113e03, 7.982692717e00,
      -2.616571843e-2, 1.522411790e-3, 2.284279054e-2, 2.421647003e02, 1.269716088e-10, 2.074838328e-7,
      2.174020350e-8, 1.105710498e-9, 1.293441934e01, 1.308119072e-5, 6.047626338e-14};
  const double a1 = 8.438375405e-1, a2 = 5.362162162e-4, a3 = 1.720000000e00, a4 = 7.342278489e-2;
  const double a5 = 4.975858870e-2, a6 = 6.537154300e-1, a7 = 1.150000000e-6, a8 = 1.510800000e-5;
  const double a9 = 1.418800000e-1, a10 = 7.002753165e00, a11 = 2.995284926e-4, a12 = 2.040000000e-1;
  const double one = 1.0, two = 2.0, three = 3.0, five = 5.0, six = 6.0, nine = 9.0, ten = 10.0;
  double tc1 = H2O_CRITICAL_TEMPERATURE, pc1 = H2O_CRITICAL_PRESSURE, vc1 = 0.00317;
  double utc1 = one / tc1, upc1 = one / pc1, vc1mol = vc1 * FMWH2O;
  double theta = (t + 273.15) * utc1;
  double theta2x = theta * theta;
  double theta18 = pow(theta, 18.0);
  double theta20 = theta18 * theta2x;
  double beta = p * upc1;
  double beta2x = beta * beta;
  double beta4 = beta2x * beta2x;
  double yy = one - a1 * theta2x - a2 * pow(theta, -6.0);
  double xx = a3 * yy * yy - two * (a4 * theta - a5 * beta);
  if (xx > 0.0) xx = sqrt(xx); else xx = 1.e-6f;
  double zz = yy + xx;
  double u0 = -five / 17.0;
  double u1 = aa[11] * a5 * pow(zz, u0);
  double ypt = six * a2 * pow(theta, -7.0) - two * a1 * theta;
  double utheta = one / theta;
  double term1 = aa[0] * theta;
  double term2 = -aa[1];
  double term2t = 0.0;
  for (int i = 3; i <= 10; i++) {
    double tempreal = (double)(i - 2) * aa[i] * powi(theta, i - 1);  /* theta**(i-1): integer power */
    term2t = term2t + tempreal * utheta * (double)(i - 1);
    term2 = term2 + tempreal;
  }
  double v0_1 = u1 / a5;
  double v2_1 = 17.0 * (zz / 29.0 - yy / 12.0) + five * theta * ypt / 12.0;
  double v3_1 = a4 * theta - (a3 - one) * theta * yy * ypt;
  double v1_1 = zz * v2_1 + v3_1;
  double term3 = v0_1 * v1_1;
  double v1_2 = nine * theta + a6;
  double v20_2 = (a6 - theta);
  double v2_2 = pow(v20_2, 9.0);
  double v3_2 = a7 + 20.0 * pow(theta, 19.0);
  double v40_2 = a7 + pow(theta, 19.0);
  double v4_2 = one / (v40_2 * v40_2);
  double term4p = aa[12] - aa[14] * theta2x + aa[15] * v1_2 * v2_2 + aa[16] * v3_2 * v4_2;
  double term4 = term4p * beta;
  double v1_3 = beta * (aa[17] + aa[18] * beta + aa[19] * beta2x);
  double v2_3 = 12.0 * pow(theta, 11.0) + a8;
  double v4_3 = one / (a8 + pow(theta, 11.0));
  double v3_3 = v4_3 * v4_3;
  double term5 = v1_3 * v2_3 * v3_3;
  double v1_4 = pow(a10 + beta, -3.0) + a11 * beta;
  double v3_4 = (17.0 * a9 + 19.0 * theta2x);
  double v2_4 = aa[20] * theta18 * v3_4;
  double term6 = v1_4 * v2_4;
  double v1_5 = 21.0 * aa[22] / theta20 * beta4;
  double v2_5 = aa[21] * a12 * beta2x * beta;
  double term7 = v1_5 + v2_5;
  double v1_6 = pc1 * vc1mol;
  *hw = (term1 - term2 + term3 + term4 - term5 + term6 + term7) * v1_6;

  double zpt = ypt + (a3 * yy * ypt - a4) / xx;
  double zpp = a5 / xx;
  double yptt = -two * a1 - 42.0 * a2 / pow(theta, 8.0);
  double dv2t = 17.0 * (zpt / 29.0 - ypt / 12.0) + five / 12.0 * (ypt + theta * yptt);
  double dv3t = a4 - (a3 - one) * (theta * yy * yptt + yy * ypt + theta * ypt * ypt);
  double dv2p = 17.0 * zpp / 29.0;
  double v4_1 = five * v1_1 / (17.0 * zz);
  double term3t = v0_1 * (zz * dv2t + (v2_1 - v4_1) * zpt + dv3t);
  double term3p = v0_1 * (zz * dv2p + (v2_1 - v4_1) * zpp);
  double term4t = (-two * aa[14] * theta + nine * aa[15] * (v2_2 - v1_2 * v2_2 / v20_2) +
                   38.0 * theta18 * aa[16] * (ten * v4_2 - v3_2 * v4_2 / v40_2)) * beta;
  double term5p = v3_3 * v2_3 * (aa[17] + two * aa[18] * beta + three * aa[19] * beta2x);
  double term5t = v1_3 * (132.0 * v3_3 * pow(theta, 10.0) - 22.0 * v2_3 * v3_3 * v4_3 * pow(theta, 10.0));
  double term6p = v2_4 * (a11 - three * pow(a10 + beta, -4.0));
  double term6t = v1_4 * aa[20] * theta18 * (18.0 * v3_4 * utheta + 38.0 * theta);
  double term7p = beta2x * (three * aa[21] * a12 + 84.0 * aa[22] * beta / theta20);
  double term7t = -420.0 * aa[22] * beta4 / (theta20 * theta);
  *hwp = (term3p + term4p - term5p + term6p + term7p) * vc1mol;
  *hwt = (aa[0] - term2t + term3t + term4t - term5t + term6t + term7t) * v1_6 * utc1;
}

/* InternalEnergyAndEnthalpy, EOSWaterMod.F90:589-707; den, dden_* in kg m^-3 */
void tho_internal_energy_enthalpy(double P, double t_K, int itype, double den, double dden_dT, double dden_dP, double *out6) {
  double U, H, dU_dT, dH_dT, dU_dP, dH_dP;
  if (itype == THO_INT_ENERGY_ENTHALPY_CONSTANT) { /* :626-665 */
    const double u0 = (double)4.217f * 1.e3;       /* `4.217 * 1.d3`: the first literal is default (single) real */
    U = u0 * (t_K - 273.15);
    dU_dT = u0;
    dU_dP = 0.0;
    H = U + P / den;
    dH_dT = dU_dT - P / (den * den) * dden_dT;
    dH_dP = dU_dP + 1.0 / den - P / (den * den) * dden_dP;
    U = U * FMWH2O; H = H * FMWH2O; dU_dT = dU_dT * FMWH2O; dH_dT = dH_dT * FMWH2O; dH_dP = dH_dP * FMWH2O;
  } else {                                           /* IFC67 :668-705 */
    double T_C = t_K - 273.15;
    enthalpy_ifc67(T_C, P, &H, &dH_dP, &dH_dT);
    U = H - P / (den / FMWH2O);
    dU_dT = dH_dT + P / ((den / FMWH2O) * (den / FMWH2O)) * (dden_dT / FMWH2O);
    dU_dP = dH_dP - 1.0 / (den / FMWH2O) + P / ((den / FMWH2O) * (den / FMWH2O)) * (dden_dP / FMWH2O);
  }
  out6[0] = U; out6[1] = H; out6[2] = dU_dT; out6[3] = dH_dT; out6[4] = dU_dP; out6[5] = dH_dP;
}

/* ------------------------------------------------------------------ */
/* auxvar of the heat equation: therm_enthalpy_soil_auxvar_type (ThermalEnthalpySoilAuxType.F90:18-47) */
/* ------------------------------------------------------------------ */
typedef struct {
  auxvar a; /* extends rich_ode_pres_auxvar_type */
  double ul, hl, dul_dP, dhl_dP, dul_dT, dhl_dT, dsat_dT;
  double Kel, therm_cond_wet, therm_cond_dry, therm_cond, dtherm_cond_dP, dKel_dp, therm_alpha;
  double den_soil, heat_cap_soil;
} thaux;

static void thaux_init(thaux *t) { /* ThermEnthalpyAuxVarInit :58-100 */
  memset(t, 0, sizeof *t);
  vo__auxvar_init(&t->a);
  t->a.perm[0] = t->a.perm[1] = t->a.perm[2] = 8.3913e-12;
}

/* ThermEnthalpyAuxVarCompute :219-278 */
static void thaux_compute(thaux *t, int density_type, int ie_type) {
  auxvar *a = &t->a;
  double dvis_dT, e[6];
  vo__press_to_sat(&a->sp, a->pressure, &a->sat, &a->dsat_dP);
  vo__press_to_relperm(&a->sp, a->pressure, 1.0, &a->kr, &a->dkr_dP);
  /* PorosityFunctionComputation (PorosityFunctionMod.F90:99-163) at the cell pressure, ThermalEnthalpySoilAuxType.F90:246-249 */
  if (a->por_model == 2) {
    a->por = a->por_base + (a->pressure - a->por_pressure_reference) * a->por_lin_mod_slope;
    a->dpor_dP = a->por_lin_mod_slope;
  } else { a->por = a->por_base; a->dpor_dP = 0.0; }
  double pressure = a->pressure;
  if (a->pressure < PRESSURE_REF) pressure = PRESSURE_REF;
  vo_density(pressure, a->temperature, density_type, &a->den, &a->dden_dP, &a->dden_dT);
  vo_viscosity(pressure, a->temperature, &a->vis, &a->dvis_dP, &dvis_dT);
  tho_internal_energy_enthalpy(pressure, a->temperature, ie_type, a->den * FMWH2O, a->dden_dT * FMWH2O, a->dden_dP * FMWH2O, e);
  t->ul = e[0]; t->hl = e[1]; t->dul_dT = e[2]; t->dhl_dT = e[3]; t->dul_dP = e[4]; t->dhl_dP = e[5];
  t->Kel = pow(a->sat + 1.e-6, t->therm_alpha);
  t->dKel_dp = t->therm_alpha * pow(a->sat + 1.e-6, t->therm_alpha - 1.0) * a->dsat_dP;
  t->therm_cond = t->therm_cond_wet * t->Kel + t->therm_cond_dry * (1.0 - t->Kel);
  t->dtherm_cond_dP = (t->therm_cond_wet - t->therm_cond_dry) * t->dKel_dp;
}

/* RichardsFluxDerivativeWrtTemperature_Internal, RichardsMod.F90:431-583: returns the TRUE d(flux)/dT */
static void richards_flux_dT_internal(const auxvar *up, const auxvar *dn, const conn_geom *g, int internal_conn,
                                      int cond_type, double *flux, double *dflux_dT_up, double *dflux_dT_dn) {
  double area = g->area, dist_up = g->dist_up, dist_dn = g->dist_dn;
  const double *u = g->uvec;
  double dvis_dT_up = 0.0, dvis_dT_dn = 0.0; /* Viscosity :582-584 */
  double perm_up = fabs(u[0]) * up->perm[0] + fabs(u[1]) * up->perm[1] + fabs(u[2]) * up->perm[2];
  double perm_dn = fabs(u[0]) * dn->perm[0] + fabs(u[1]) * dn->perm[1] + fabs(u[2]) * dn->perm[2];
  double upweight, Dq;
  if (internal_conn || cond_type == VO_COND_DIRICHLET_FRM_OTR_GOVEQ) {
    upweight = dist_up / (dist_up + dist_dn);
    Dq = (perm_up * perm_dn) / (dist_up * perm_dn + dist_dn * perm_up);
  } else {
    upweight = 0.0;
    Dq = perm_dn / (dist_up + dist_dn);
  }
  double udist_dot_ugrav = u[0] * 0.0 + u[1] * 0.0 + u[2] * (-GRAVITY_CONSTANT);
  double dist_gravity = (dist_up + dist_dn) * udist_dot_ugrav;
  double den_ave = upweight * up->den + (1.0 - upweight) * dn->den;
  double gravityterm = (upweight * up->den + (1.0 - upweight) * dn->den) * FMWH2O * dist_gravity;
  double dphi = up->pressure - dn->pressure + gravityterm;
  int seepage_bc_update = 0;
  if (!internal_conn && cond_type == VO_COND_SEEPAGE_BC)
    if (dphi > 0.0 && up->pressure <= PRESSURE_REF) seepage_bc_update = 1;
  if (seepage_bc_update) dphi = 0.0;
  double ukvr = (dphi >= 0.0) ? up->kr / up->vis : dn->kr / dn->vis;
  int massflux_bc = (!internal_conn && cond_type == VO_COND_MASS_FLUX);
  double v_darcy = massflux_bc ? 0.0 : -Dq * ukvr * dphi;
  double q = v_darcy * area;
  *flux = q * den_ave;
  double dden_ave_dT_up = upweight * up->dden_dT;
  double dden_ave_dT_dn = (1.0 - upweight) * dn->dden_dT;
  double dgravityterm_dden_up = upweight * dist_gravity * FMWH2O;
  double dgravityterm_dden_dn = (1.0 - upweight) * dist_gravity * FMWH2O;
  double dphi_dT_up = dgravityterm_dden_up * up->dden_dT;
  double dphi_dT_dn = dgravityterm_dden_dn * dn->dden_dT;
  if (seepage_bc_update) dphi_dT_dn = 0.0;
  double dukvr_dT_up, dukvr_dT_dn;
  if (dphi >= 0) {
    dukvr_dT_up = -up->kr / (up->vis * up->vis) * dvis_dT_up;
    dukvr_dT_dn = 0.0;
  } else {
    dukvr_dT_up = 0.0;
    dukvr_dT_dn = -dn->kr / (dn->vis * dn->vis) * dvis_dT_dn;
  }
  double dq_dT_up = Dq * (dukvr_dT_up * dphi + ukvr * dphi_dT_up) * area;
  double dq_dT_dn = Dq * (dukvr_dT_dn * dphi + ukvr * dphi_dT_dn) * area;
  double dup, ddn;
  if (massflux_bc) { dup = 0.0; ddn = 0.0; }
  else {
    dup = (dq_dT_up * den_ave - q * dden_ave_dT_up);
    ddn = (dq_dT_dn * den_ave - q * dden_ave_dT_dn);
  }
  *dflux_dT_up = -dup;
  *dflux_dT_dn = -ddn;
}

/* ThermalEnthalpyFlux, ThermalEnthalpyMod.F90:27-165 (BC: COND_DIRICHLET only) */
static void thermal_enthalpy_flux(const thaux *up, const thaux *dn, double mflux, double dmflux_dT_up, double dmflux_dT_dn,
                                  const conn_geom *g, int compute_deriv, int internal_conn, double *eflux,
                                  double *deflux_dT_up, double *deflux_dT_dn) {
  double T_up = up->a.temperature, h_up = up->hl, dh_up_dT_up = up->dhl_dT, therm_cond_up = up->therm_cond;
  double T_dn = dn->a.temperature, h_dn = dn->hl, dh_dn_dT_dn = dn->dhl_dT, therm_cond_dn = dn->therm_cond;
  double area = g->area, dist_up = g->dist_up, dist_dn = g->dist_dn;
  double therm_cond_ave_over_dist;
  if (internal_conn)
    therm_cond_ave_over_dist = (therm_cond_up * therm_cond_dn) / (dist_up * therm_cond_dn + dist_dn * therm_cond_up);
  else
    therm_cond_ave_over_dist = therm_cond_dn / (dist_up + dist_dn);
  double h = (mflux <= 0.0) ? h_up : h_dn;
  *eflux = mflux * h + (-therm_cond_ave_over_dist * (T_up - T_dn) * area);
  if (compute_deriv) {
    double dh_dT_up, dh_dT_dn;
    if (mflux < 0.0) { dh_dT_up = dh_up_dT_up; dh_dT_dn = 0.0; }
    else { dh_dT_up = 0.0; dh_dT_dn = dh_dn_dT_dn; }
    *deflux_dT_up = dmflux_dT_up * h + mflux * dh_dT_up + (-therm_cond_ave_over_dist * area);
    *deflux_dT_dn = dmflux_dT_dn * h + mflux * dh_dT_dn + (+therm_cond_ave_over_dist * area);
  } else { *deflux_dT_up = 0.0; *deflux_dT_dn = 0.0; }
}

/* ThermalEnthalpyFluxDerivativeWrtPressure, ThermalEnthalpyMod.F90:168-333 */
static void thermal_enthalpy_flux_dP(const thaux *up, const thaux *dn, double mflux, double dmflux_dP_up, double dmflux_dP_dn,
                                     const conn_geom *g, int internal_conn, double *deflux_dP_up, double *deflux_dP_dn) {
  double T_up = up->a.temperature, h_up = up->hl, dh_up_dP_up = up->dhl_dP, therm_cond_up = up->therm_cond;
  double dtherm_cond_up_dP_up = up->dtherm_cond_dP;
  double T_dn = dn->a.temperature, h_dn = dn->hl, dh_dn_dP_dn = dn->dhl_dP, therm_cond_dn = dn->therm_cond;
  double dtherm_cond_dn_dP_dn = dn->dtherm_cond_dP;
  double area = g->area, dist_up = g->dist_up, dist_dn = g->dist_dn;
  double therm_cond_ave_over_dist;
  if (internal_conn)
    therm_cond_ave_over_dist = (therm_cond_up * therm_cond_dn) / (dist_up * therm_cond_dn + dist_dn * therm_cond_up);
  else
    therm_cond_ave_over_dist = therm_cond_dn / (dist_up + dist_dn);
  double h = (mflux <= 0.0) ? h_up : h_dn;
  double dh_dP_up, dh_dP_dn, dDk_dp_up, dDk_dp_dn;
  if (mflux < 0.0) { dh_dP_up = dh_up_dP_up; dh_dP_dn = 0.0; }
  else { dh_dP_up = 0.0; dh_dP_dn = dh_dn_dP_dn; }
  if (internal_conn) {
    dDk_dp_up = pow(therm_cond_ave_over_dist, 2.0) / pow(therm_cond_up, 2.0) * dist_up * dtherm_cond_up_dP_up;
    dDk_dp_dn = pow(therm_cond_ave_over_dist, 2.0) / pow(therm_cond_dn, 2.0) * dist_dn * dtherm_cond_dn_dP_dn;
  } else {
    dDk_dp_up = 0.0;
    dDk_dp_dn = 1.0 / (dist_up + dist_dn) * dtherm_cond_dn_dP_dn;
  }
  *deflux_dP_up = dmflux_dP_up * h + mflux * dh_dP_up + (-dDk_dp_up * (T_up - T_dn) * area);
  *deflux_dP_dn = dmflux_dP_dn * h + mflux * dh_dP_dn + (-dDk_dp_dn * (T_up - T_dn) * area);
}

/* ------------------------------------------------------------------ */
/* problem object                                                      */
/* ------------------------------------------------------------------ */
typedef struct {
  int kind, cond_type, n;
  int *id_dn;
  conn_geom *geom;
  thaux *aux;      /* aux_vars_bc */
  double *value;   /* condition values */
} econd;

struct tho_problem {
  int n;
  vo_problem *m;   /* the mass equation (goveqn_richards_ode_pressure_type) */
  vo_problem *g;   /* host of the 2N global system for the SNES restatement */
  thaux *aux;      /* aux_vars_in of the heat equation */
  int density_type, ie_type;
  int nbc, nss;
  econd *bc, *ss;
  double *accum_prev_e;
  /* block slot tables into the global CSR: for mass-pattern slot s of row i, global slots of the four blocks */
  int *gPP, *gPT, *gTP, *gTT;
  double *blk; /* scratch: one block in the mass pattern */
  int ready;
};

tho_problem *tho_create(int ncells) {
  tho_problem *t = (tho_problem *)calloc(1, sizeof *t);
  t->n = ncells;
  t->m = vo_create(ncells);
  t->g = vo_create(2 * ncells);
  t->aux = (thaux *)malloc(ncells * sizeof(thaux));
  for (int i = 0; i < ncells; i++) thaux_init(&t->aux[i]);
  t->density_type = VO_DENSITY_CONSTANT;
  t->ie_type = THO_INT_ENERGY_ENTHALPY_CONSTANT;
  t->accum_prev_e = (double *)calloc(ncells, sizeof(double));
  return t;
}

static void free_econd(econd *c) { free(c->id_dn); free(c->geom); free(c->aux); free(c->value); }

void tho_destroy(tho_problem *t) {
  if (!t) return;
  for (int i = 0; i < t->nbc; i++) free_econd(&t->bc[i]);
  for (int i = 0; i < t->nss; i++) free_econd(&t->ss[i]);
  free(t->bc); free(t->ss); free(t->aux); free(t->accum_prev_e);
  free(t->gPP); free(t->gPT); free(t->gTP); free(t->gTT); free(t->blk);
  /* the global CSR arrays live in g and are freed with it */
  vo_destroy(t->m);
  vo_destroy(t->g);
  free(t);
}

vo_problem *tho_mass(tho_problem *t) { return t->m; }
vo_problem *tho_global(tho_problem *t) { return t->g; }

int tho_set_eos(tho_problem *t, int density_type, int ie_type) {
  if (ie_type != THO_INT_ENERGY_ENTHALPY_CONSTANT && ie_type != THO_INT_ENERGY_ENTHALPY_IFC67) FAILT("unknown int_energy_enthalpy_type");
  t->density_type = density_type; t->ie_type = ie_type;
  return vo_set_density_type(t->m, density_type);
}

int tho_energy_add_condition(tho_problem *t, int kind, int cond_type, int nconn, const int *id_dn, const double *dist_up,
                             const double *dist_dn, const double *area, const double *unit_vec, int *cond_id) {
  if (kind == VO_COND_BC) {
    if (cond_type != VO_COND_DIRICHLET) FAILT("heat equation: only COND_DIRICHLET boundaries (ThermEnthalpySoilUpdateAuxVarsBC :964-1017 aborts on others)");
  } else if (kind == VO_COND_SS) {
    if (cond_type != VO_COND_HEAT_RATE) FAILT("heat equation: only COND_HEAT_RATE sources");
  } else FAILT("kind must be COND_BC or COND_SS");
  econd **list = kind == VO_COND_BC ? &t->bc : &t->ss;
  int *cnt = kind == VO_COND_BC ? &t->nbc : &t->nss;
  *list = (econd *)realloc(*list, (*cnt + 1) * sizeof(econd));
  econd *c = &(*list)[*cnt];
  memset(c, 0, sizeof *c);
  c->kind = kind; c->cond_type = cond_type; c->n = nconn;
  c->id_dn = (int *)malloc(nconn * sizeof(int));
  c->geom = (conn_geom *)calloc(nconn, sizeof(conn_geom));
  c->aux = (thaux *)malloc(nconn * sizeof(thaux));
  c->value = (double *)calloc(nconn, sizeof(double));
  for (int i = 0; i < nconn; i++) {
    if (id_dn[i] < 1 || id_dn[i] > t->n) FAILT("condition id_dn out of range");
    c->id_dn[i] = id_dn[i] - 1;
    thaux_init(&c->aux[i]);
    if (kind == VO_COND_BC) {
      c->geom[i].area = area[i]; c->geom[i].dist_up = dist_up[i]; c->geom[i].dist_dn = dist_dn[i];
      memcpy(c->geom[i].uvec, unit_vec + 3 * i, 3 * sizeof(double));
    }
  }
  ++*cnt;
  if (cond_id) *cond_id = *cnt;
  return 0;
}

/* The linear porosity model (PorosityFunctionMod.F90:75-96) for the heat equation's auxvars: porParams are per auxvar
 * (ThermalEnthalpySoilAuxType.F90:156-157); por given to tho_energy_set_soils is then porosity_base.  NULL, NULL: constant model.
 * No MPP-level routine of the reference selects the model: parity unpinned by the reference. */
int tho_energy_set_porosity_linear(tho_problem *t, const double *press_ref, const double *slope) {
  for (int i = 0; i < t->n; i++) {
    auxvar *a = &t->aux[i].a;
    a->por_model = press_ref ? 2 : 1;
    a->por_pressure_reference = press_ref ? press_ref[i] : 0.0;
    a->por_lin_mod_slope = slope ? slope[i] : 0.0;
  }
  for (int b = 0; b < t->nbc; b++)
    for (int i = 0; i < t->bc[b].n; i++) {
      auxvar *x = &t->bc[b].aux[i].a;
      const auxvar *s = &t->aux[t->bc[b].id_dn[i]].a;
      x->por_model = s->por_model; x->por_pressure_reference = s->por_pressure_reference; x->por_lin_mod_slope = s->por_lin_mod_slope;
    }
  return 0;
}

/* MPPTHSetSoilsForThermalEnthalpy (MultiPhysicsProbTH.F90:179-401) / the ThermEnthalpySet* setters (GE :2454-2700):
 * per-cell arrays; perm_* may be NULL (the auxvar default 8.3913e-12, ThermalEnthalpySoilAuxType.F90:93, stays --
 * MPPTHSetSoils never overwrites it, MultiPhysicsProbTH.F90:293).  Boundary auxvars copy from their id_dn cell. */
int tho_energy_set_soils(tho_problem *t, const double *por, const int *satfunc_type, const double *alpha, const double *lambda,
                         const double *sat_res, const double *perm_x, const double *perm_y, const double *perm_z,
                         const double *heat_cap, const double *tcond_wet, const double *tcond_dry, const double *talpha,
                         const double *den_soil) {
  for (int i = 0; i < t->n; i++) {
    thaux *x = &t->aux[i];
    x->a.por_base = por[i];
    if (vo__satparams_set(&x->a.sp, satfunc_type[i], alpha[i], lambda[i], sat_res[i])) FAILT("bad saturation function parameters");
    if (perm_x) { x->a.perm[0] = perm_x[i]; x->a.perm[1] = perm_y[i]; x->a.perm[2] = perm_z[i]; }
    x->heat_cap_soil = heat_cap[i]; x->therm_cond_wet = tcond_wet[i]; x->therm_cond_dry = tcond_dry[i];
    x->therm_alpha = talpha[i]; x->den_soil = den_soil[i];
  }
  for (int b = 0; b < t->nbc; b++)
    for (int i = 0; i < t->bc[b].n; i++) {
      thaux *x = &t->bc[b].aux[i];
      const thaux *s = &t->aux[t->bc[b].id_dn[i]];
      x->a.por_base = s->a.por_base; x->a.sp = s->a.sp;
      if (perm_x) memcpy(x->a.perm, s->a.perm, sizeof x->a.perm); /* the ThermEnthalpySetSoilPermeability path copies, MPPTHSetSoils does not */
      x->heat_cap_soil = s->heat_cap_soil; x->therm_cond_wet = s->therm_cond_wet; x->therm_cond_dry = s->therm_cond_dry;
      x->therm_alpha = s->therm_alpha; x->den_soil = s->den_soil;
    }
  return 0;
}

int tho_energy_set_condition_values(tho_problem *t, int kind, int cond_id, int n, const double *values) {
  int cnt = kind == VO_COND_BC ? t->nbc : t->nss;
  if (cond_id < 1 || cond_id > cnt) FAILT("no such condition");
  econd *c = kind == VO_COND_BC ? &t->bc[cond_id - 1] : &t->ss[cond_id - 1];
  if (n > c->n) FAILT("size(data_1d) > nauxvar");
  for (int i = 0; i < n; i++) { c->value[i] = values[i]; c->aux[i].a.condition_value = values[i]; }
  return 0;
}
/* aux_vars_bc(:)%pressure of the heat equation, assigned directly by the reference drivers
 * (mass_and_heat_model_problem.F90:583-640, th_mms_problem.F90:846-860) */
int tho_energy_set_bc_pressure(tho_problem *t, int cond_id, int n, const double *values) {
  if (cond_id < 1 || cond_id > t->nbc) FAILT("no such condition");
  econd *c = &t->bc[cond_id - 1];
  if (n > c->n) FAILT("size(data_1d) > nauxvar");
  for (int i = 0; i < n; i++) c->aux[i].a.pressure = values[i];
  return 0;
}
/* aux_vars_bc(:)%temperature of the mass equation (th_mms_problem.F90:862-875) */
int tho_mass_set_bc_temperature(tho_problem *t, int cond_id, int n, const double *values) {
  vo_problem *m = t->m;
  if (cond_id < 1 || cond_id > m->nbc) FAILT("no such condition");
  condition *c = &m->bc[cond_id - 1];
  if (n > c->n) FAILT("size(data_1d) > nauxvar");
  for (int i = 0; i < n; i++) c->aux[i].temperature = values[i];
  return 0;
}

int tho_restart(tho_problem *t, const double *pressure, const double *temperature) {
  int n = t->n;
  memcpy(t->g->soln, pressure, n * sizeof(double));
  memcpy(t->g->soln + n, temperature, n * sizeof(double));
  memcpy(t->g->soln_prev, t->g->soln, 2 * n * sizeof(double));
  return 0;
}

/* the 2N x 2N pattern: each of the four blocks has the mass equation's pattern (diagonal + internal connections) */
int tho_setup(tho_problem *t) {
  vo_problem *m = t->m, *g = t->g;
  if (!m->rowptr) FAILT("mass equation: connections not set");
  int n = t->n, nnzb = m->rowptr[n];
  free(g->rowptr); free(g->col); free(g->val);
  g->rowptr = (int *)malloc((2 * n + 1) * sizeof(int));
  g->col = (int *)malloc(4 * (size_t)nnzb * sizeof(int));
  g->val = (double *)calloc(4 * (size_t)nnzb, sizeof(double));
  t->gPP = (int *)malloc(nnzb * sizeof(int)); t->gPT = (int *)malloc(nnzb * sizeof(int));
  t->gTP = (int *)malloc(nnzb * sizeof(int)); t->gTT = (int *)malloc(nnzb * sizeof(int));
  t->blk = (double *)malloc(nnzb * sizeof(double));
  int k = 0;
  for (int r = 0; r < 2 * n; r++) {
    int i = r < n ? r : r - n;
    g->rowptr[r] = k;
    for (int s = m->rowptr[i]; s < m->rowptr[i + 1]; s++) { (r < n ? t->gPP : t->gTP)[s] = k; g->col[k++] = m->col[s]; }
    for (int s = m->rowptr[i]; s < m->rowptr[i + 1]; s++) { (r < n ? t->gPT : t->gTT)[s] = k; g->col[k++] = n + m->col[s]; }
  }
  g->rowptr[2 * n] = k;
  t->ready = 1;
  return 0;
}

/* SavePrimaryIndependentVar + SOETHGovEqnExchangeAuxVars + UpdateAuxVars of the heat equation
 * (SOETHResidual :776-803; GE :924-1017).  The mass equation's auxvars are refreshed by vo_residual / vo_pre_solve. */
static void th_update_aux(tho_problem *t, const double *x) {
  int n = t->n;
  vo_problem *m = t->m;
  for (int i = 0; i < n; i++) {
    m->aux[i].temperature = x[n + i];      /* mass equation needs VAR_TEMPERATURE from the heat equation */
    t->aux[i].a.pressure = x[i];           /* heat equation needs VAR_PRESSURE from the mass equation   */
    t->aux[i].a.temperature = x[n + i];
    thaux_compute(&t->aux[i], t->density_type, t->ie_type);
  }
  for (int b = 0; b < t->nbc; b++)
    for (int i = 0; i < t->bc[b].n; i++) {
      thaux *a = &t->bc[b].aux[i];
      a->a.temperature = a->a.condition_value; /* COND_DIRICHLET :994-996 */
      thaux_compute(a, t->density_type, t->ie_type);
    }
}

/* ThermalEnthalpySoilAccum :1174-1219 */
static void energy_accum(const tho_problem *t, double *f) {
  const vo_problem *m = t->m;
  double dtInv = 1.0 / m->dtime;
  for (int i = 0; i < t->n; i++) {
    if (!m->active[i]) continue;
    const thaux *a = &t->aux[i];
    f[i] = f[i] + (a->a.por * a->a.den * a->a.sat * a->ul +
                   (1.0 - a->a.por) * a->den_soil * a->heat_cap_soil * (a->a.temperature - 273.15)) * m->vol[i] * dtInv;
  }
}

/* ThermalEnthalpySoilDivergence :1299-1498 */
static void energy_divergence(const tho_problem *t, double *f) {
  const vo_problem *m = t->m;
  double mflux, d1, d2, eflux;
  for (int c = 0; c < m->nc; c++) {
    int u = m->up[c], d = m->dn[c];
    if (!m->active[u] || !m->active[d]) continue;
    vo__richards_flux(&t->aux[u].a, &t->aux[d].a, &m->geom[c], 0, 1, 0, VO_COND_NULL, &mflux, &d1, &d2);
    thermal_enthalpy_flux(&t->aux[u], &t->aux[d], mflux, 0.0, 0.0, &m->geom[c], 0, 1, &eflux, &d1, &d2);
    f[u] = f[u] - eflux;
    f[d] = f[d] + eflux;
  }
  for (int b = 0; b < t->nbc; b++) {
    const econd *cd = &t->bc[b];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!m->active[cell]) continue;
      vo__richards_flux(&cd->aux[i].a, &t->aux[cell].a, &cd->geom[i], 0, 0, 0, cd->cond_type, &mflux, &d1, &d2);
      thermal_enthalpy_flux(&cd->aux[i], &t->aux[cell], mflux, 0.0, 0.0, &cd->geom[i], 0, 0, &eflux, &d1, &d2);
      f[cell] = f[cell] + eflux;
    }
  }
  for (int s = 0; s < t->nss; s++) {
    const econd *cd = &t->ss[s];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!m->active[cell]) continue;
      f[cell] = f[cell] + cd->value[i]; /* COND_HEAT_RATE :1481 */
    }
  }
}

int tho_set_dtime(tho_problem *t, double dt) { t->m->dtime = dt; t->g->dtime = dt; return 0; }

/* SOETHPreSolve :119-272 */
int tho_pre_solve(tho_problem *t) {
  if (!t->ready) FAILT("tho_setup not called");
  int n = t->n;
  const double *xp = t->g->soln_prev;
  th_update_aux(t, xp);
  memcpy(t->m->soln_prev, xp, n * sizeof(double));
  vo_pre_solve(t->m);                                  /* RichardsODEPressurePreSolve */
  memset(t->accum_prev_e, 0, n * sizeof(double));      /* ThermEnthalpySoilPreSolve :2410-2432 */
  energy_accum(t, t->accum_prev_e);
  return 0;
}

/* SOETHResidual :736-850 */
int tho_residual(tho_problem *t, const double *x, double *f) {
  if (!t->ready) FAILT("tho_setup not called");
  int n = t->n;
  th_update_aux(t, x);
  if (vo_residual(t->m, x, f)) return -1;              /* mass equation, first N rows */
  double *fe = f + n;
  memset(fe, 0, n * sizeof(double));
  energy_accum(t, fe);
  for (int i = 0; i < n; i++) fe[i] = fe[i] - t->accum_prev_e[i];
  energy_divergence(t, fe);
  t->g->nfuncs++; t->g->cum_funcs++;
  return 0;
}

static void add4(tho_problem *t, int c, double uu, double ud, double du, double dd) {
  const vo_problem *m = t->m;
  t->blk[m->slot_uu[c]] += uu; t->blk[m->slot_ud[c]] += ud; t->blk[m->slot_du[c]] += du; t->blk[m->slot_dd[c]] += dd;
}
static void scatter(tho_problem *t, const int *gslot) {
  int nnzb = t->m->rowptr[t->n];
  for (int s = 0; s < nnzb; s++) t->g->val[gslot[s]] = t->blk[s];
}

/* SOETHJacobian :853-1003 at the state of the last residual evaluation */
static void th_jacobian_assemble(tho_problem *t) {
  vo_problem *m = t->m;
  int n = t->n, nnzb = m->rowptr[n];
  double dtInv = 1.0 / m->dtime;
  double mflux, Jup, Jdn, dmu, dmd, e;

  /* (1,1) d(mass)/dP: RichardsODEComputeJacobian */
  m->jacobian_fn(m);
  memcpy(t->blk, m->val, nnzb * sizeof(double));
  scatter(t, t->gPP);

  /* (1,2) d(mass)/dT: OffDiagJacobian_Temperature_ForInternalAuxVars, GoveqnRichardsODEPressureType.F90:2473-2612 */
  memset(t->blk, 0, nnzb * sizeof(double));
  for (int i = 0; i < n; i++) {
    double derivative;
    if (m->active[i]) {
      const auxvar *a = &m->aux[i];
      double dpor_dT = 0.0, dsat_dT = 0.0;
      derivative = (dpor_dT * a->den * a->sat + a->por * a->dden_dT * a->sat + a->por * a->den * dsat_dT) * m->vol[i] * dtInv;
    } else derivative = 1.0;
    t->blk[m->diag_slot[i]] += derivative;
  }
  for (int c = 0; c < m->nc; c++) {
    int u = m->up[c], d = m->dn[c];
    if (!m->active[u] || !m->active[d]) continue;
    richards_flux_dT_internal(&m->aux[u], &m->aux[d], &m->geom[c], 1, VO_COND_NULL, &mflux, &Jup, &Jdn);
    add4(t, c, -Jup, -Jdn, Jup, Jdn);
  }
  scatter(t, t->gPT);

  /* (2,1) d(energy)/dP: OffDiagJacobian_Pressure_ForInternalAuxVars, GoveqnThermalEnthalpySoilType.F90:2083-2376 */
  memset(t->blk, 0, nnzb * sizeof(double));
  for (int i = 0; i < n; i++) {
    double derivative;
    if (m->active[i]) {
      const thaux *a = &t->aux[i];
      double por = a->a.por, den = a->a.den, sat = a->a.sat, ul = a->ul, dpor_dP = a->a.dpor_dP;
      derivative = dpor_dP * den * sat * ul + por * a->a.dden_dP * sat * ul + por * den * a->a.dsat_dP * ul +
                   por * den * sat * a->dul_dP + (-dpor_dP * a->den_soil * a->heat_cap_soil * (a->a.temperature - 273.15));
      derivative = derivative * m->vol[i] * dtInv;
    } else derivative = 1.0;
    t->blk[m->diag_slot[i]] += derivative;
  }
  for (int c = 0; c < m->nc; c++) {
    int u = m->up[c], d = m->dn[c];
    if (!m->active[u] || !m->active[d]) continue;
    vo__richards_flux(&t->aux[u].a, &t->aux[d].a, &m->geom[c], 1, 1, 0, VO_COND_NULL, &mflux, &dmu, &dmd);
    dmu = -dmu; dmd = -dmd;
    thermal_enthalpy_flux_dP(&t->aux[u], &t->aux[d], mflux, dmu, dmd, &m->geom[c], 1, &Jup, &Jdn);
    add4(t, c, -Jup, -Jdn, Jup, Jdn);
  }
  for (int b = 0; b < t->nbc; b++) {
    const econd *cd = &t->bc[b];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i]; /* no is_active test here in the reference (:2283-2370) */
      vo__richards_flux(&cd->aux[i].a, &t->aux[cell].a, &cd->geom[i], 1, 0, 0, cd->cond_type, &mflux, &dmu, &dmd);
      dmu = -dmu; dmd = -dmd;
      thermal_enthalpy_flux_dP(&cd->aux[i], &t->aux[cell], mflux, dmu, dmd, &cd->geom[i], 0, &Jup, &Jdn);
      t->blk[m->diag_slot[cell]] += Jdn;
    }
  }
  scatter(t, t->gTP);

  /* (2,2) d(energy)/dT: ThermalEnthalpySoilDivergenceDeriv :1501-1716 then AccumDeriv :1223-1296 */
  memset(t->blk, 0, nnzb * sizeof(double));
  for (int c = 0; c < m->nc; c++) {
    int u = m->up[c], d = m->dn[c];
    if (!m->active[u] || !m->active[d]) continue;
    richards_flux_dT_internal(&t->aux[u].a, &t->aux[d].a, &m->geom[c], 1, VO_COND_NULL, &mflux, &dmu, &dmd);
    thermal_enthalpy_flux(&t->aux[u], &t->aux[d], mflux, dmu, dmd, &m->geom[c], 1, 1, &e, &Jup, &Jdn);
    add4(t, c, -Jup, -Jdn, Jup, Jdn);
  }
  for (int b = 0; b < t->nbc; b++) {
    const econd *cd = &t->bc[b];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!m->active[cell]) continue;
      richards_flux_dT_internal(&cd->aux[i].a, &t->aux[cell].a, &cd->geom[i], 0, cd->cond_type, &mflux, &dmu, &dmd);
      thermal_enthalpy_flux(&cd->aux[i], &t->aux[cell], mflux, dmu, dmd, &cd->geom[i], 1, 0, &e, &Jup, &Jdn);
      t->blk[m->diag_slot[cell]] += Jdn;
    }
  }
  for (int i = 0; i < n; i++) {
    double derivative;
    if (m->active[i]) {
      const thaux *a = &t->aux[i];
      double por = a->a.por, den = a->a.den, sat = a->a.sat, ul = a->ul;
      derivative = por * a->a.dden_dT * sat * ul + por * den * a->dsat_dT * ul + por * den * sat * a->dul_dT;
      derivative = (derivative + (1.0 - por) * a->den_soil * a->heat_cap_soil) * m->vol[i] * dtInv;
    } else derivative = 1.0;
    t->blk[m->diag_slot[i]] += derivative;
  }
  scatter(t, t->gTT);
}

int tho_jacobian_nnz(tho_problem *t) { return t->ready ? t->g->rowptr[2 * t->n] : -1; }

int tho_jacobian(tho_problem *t, const double *x, int *rowptr, int *col, double *val) {
  if (!t->ready) FAILT("tho_setup not called");
  double *f = (double *)malloc(2 * t->n * sizeof(double));
  int rc = tho_residual(t, x, f);
  free(f);
  if (rc) return rc;
  th_jacobian_assemble(t);
  int n2 = 2 * t->n, nnz = t->g->rowptr[n2];
  memcpy(rowptr, t->g->rowptr, (n2 + 1) * sizeof(int));
  memcpy(col, t->g->col, nnz * sizeof(int));
  memcpy(val, t->g->val, nnz * sizeof(double));
  return 0;
}

static int g_residual_cb(vo_problem *g, const double *x, double *f) { return tho_residual((tho_problem *)g->user, x, f); }
static void g_jacobian_cb(vo_problem *g) { th_jacobian_assemble((tho_problem *)g->user); }

/* SOEBaseStepDT_SNES (SystemOfEquationsBaseType.F90:368-552) with SOETHPreSolve / SOETHPostSolve */
int tho_step_dt(tho_problem *t, double dt, int *converged, int *converged_reason) {
  if (!t->ready) FAILT("tho_setup not called");
  vo_problem *g = t->g;
  g->residual_fn = g_residual_cb; g->jacobian_fn = g_jacobian_cb; g->user = t;
  const int max_num_time_cuts = 20;
  int num_time_cuts = 0, n2 = 2 * t->n;
  double target_time = dt, dt_iter = dt;
  g->time = 0.0;
  *converged = 0;
  for (;;) {
    tho_set_dtime(t, dt_iter);
    tho_pre_solve(t);
    int nit, lit;
    int reason = vo__snes_solve(g, g->soln, &nit, &lit);
    *converged_reason = reason;
    if (reason < 0) {
      num_time_cuts++; g->cum_cuts++;
      dt_iter = 0.5 * dt_iter;
      memcpy(g->soln, g->soln_prev, n2 * sizeof(double));
    } else {
      *converged = 1;
      g->time += dt_iter;
      g->cum_newton += nit; g->cum_linear += lit;
      memcpy(g->soln_prev, g->soln, n2 * sizeof(double)); /* SOETHPostSolve :299 */
    }
    if (num_time_cuts > max_num_time_cuts) { *converged = 0; return 0; }
    if (g->time >= target_time) break;
  }
  return 0;
}

/* SOETHGetDataForCLM(AUXVAR_INTERNAL, VAR_PRESSURE | VAR_TEMPERATURE): the SoE auxvars after PostSolve = soln */
int tho_get_solution(tho_problem *t, double *pressure, double *temperature) {
  int n = t->n;
  if (pressure) memcpy(pressure, t->g->soln, n * sizeof(double));
  if (temperature) memcpy(temperature, t->g->soln + n, n * sizeof(double));
  return 0;
}

/* auxvar-level probe for parity tests: [sat, dsat_dP, kr, dkr_dP, den, dden_dP, dden_dT, ul, hl, dul_dP, dhl_dP, dul_dT, dhl_dT,
 * therm_cond, dtherm_cond_dP] */
void tho_auxvar_compute(int sat_func_type, double sat_res, double alpha, double lambda, double por, int density_type,
                        int ie_type, double tcond_wet, double tcond_dry, double talpha, double pressure, double temperature,
                        double *out15) {
  thaux x;
  thaux_init(&x);
  vo__satparams_set(&x.a.sp, sat_func_type, alpha, lambda, sat_res);
  x.a.por_base = por; x.therm_cond_wet = tcond_wet; x.therm_cond_dry = tcond_dry; x.therm_alpha = talpha;
  x.a.pressure = pressure; x.a.temperature = temperature;
  thaux_compute(&x, density_type, ie_type);
  double o[15] = {x.a.sat, x.a.dsat_dP, x.a.kr, x.a.dkr_dP, x.a.den, x.a.dden_dP, x.a.dden_dT, x.ul, x.hl, x.dul_dP, x.dhl_dP,
                  x.dul_dT, x.dhl_dT, x.therm_cond, x.dtherm_cond_dP};
  memcpy(out15, o, sizeof o);
}
