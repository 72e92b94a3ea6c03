/*
 * vsfm_oracle.c -- CPU ORACLE (test infrastructure only; see vsfm_oracle.h).
 *
 * Plain-C restatement of the reference's VSFM algorithm.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).  Order of floating-point
 * operations follows the reference expressions so results agree to rounding.
 */
#include "vsfm_oracle_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


static char g_err[512] = "";
const char *vo_last_error(void) { return g_err; }
#define FAIL(...)                                   \
  do {                                              \
    snprintf(g_err, sizeof g_err, __VA_ARGS__);     \
    return -1;                                      \
  } while (0)

void vo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ */
/* EOS: src/mpp/util/EOSWaterMod.F90                                    */
/* ------------------------------------------------------------------ */

/* DensityConstant, EOSWaterMod.F90:80-99 */
static void density_constant(double *den, double *dden_dp, double *dden_dT) {
  *den = DENH2O / FMWH2O;
  *dden_dp = 0.0;
  *dden_dT = 0.0;
}

/* DensityTGDPB01 (Tanaka et al. 2001), EOSWaterMod.F90:102-178 */
static void density_tgdpb01(double p, double t_K, double *den, double *dden_dp, double *dden_dT) {
  const double a1 = -3.983035, a2 = 301.797, a3 = 522528.9, a4 = 69.34881, a5 = 999.974950;
  const double k0 = 50.74e-11, k1 = -0.326e-11, k2 = 0.00416e-11, p0 = 101325.0;
  double t_c = t_K - 273.15;
  double sq = pow(t_c + a1, 2.0);
  double dent = a5 * (1.0 - sq * (t_c + a2) / a3 / (t_c + a4));
  double kappa, dkappa_dp, dkappa_dt;
  if (p > p0) { /* strict '>' : EOSWaterMod.F90:151 */
    kappa = 1.0 + (k0 + k1 * t_c + k2 * pow(t_c, 2.0)) * (p - p0);
    dkappa_dp = k0 + k1 * t_c + k2 * pow(t_c, 2.0);
    dkappa_dt = (k1 + 2.0 * k2 * t_c) * (p - p0);
  } else {
    kappa = 1.0;
    dkappa_dp = 0.0;
    dkappa_dt = 0.0;
  }
  *den = dent * kappa / FMWH2O;
  double ddent_dp = 0.0;
  double d1 = -sq / a3 / (t_c + a4);
  double d2 = -2.0 * (t_c + a1) * (t_c + a2) / a3 / (t_c + a4);
  double d3 = sq * (t_c + a2) / a3 / pow(t_c + a4, 2.0);
  double ddent_dt = a5 * (d1 + d2 + d3);
  *dden_dT = (ddent_dt * kappa + dent * dkappa_dt) / FMWH2O;
  *dden_dp = (ddent_dp * kappa + dent * dkappa_dp) / FMWH2O;
}

/* DensityIFC67, EOSWaterMod.F90:181-344 (water branch; t in degC, p in Pa) */
static int density_ifc67(double t, double p, double *dwmol, double *dwp, double *dwt) {
  static const double aa[23] = {
      6.824687741e03,  -5.422063673e02, -2.096666205e04, 3.941286787e04,  -6.733277739e04, 9.902381028e04,
      -1.093911774e05, 8.590841667e04,  -4.511168742e04, 1.418138926e04,  -2.017271113e03, 7.982692717e00,
      -2.616571843e-2, 1.522411790e-3,  2.284279054e-2,  2.421647003e02,  1.269716088e-10, 2.074838328e-7,
      2.174020350e-8,  1.105710498e-9,  1.293441934e01,  1.308119072e-5,  6.047626338e-14};
  const double a1 = 8.438375405e-1, a2 = 5.362162162e-4, a3 = 1.720000000e00, a4 = 7.342278489e-2;
  const double a5 = 4.975858870e-2, a6 = 6.537154300e-1, a7 = 1.150000000e-6, a8 = 1.510800000e-5;
  const double a9 = 1.418800000e-1, a10 = 7.002753165e00, a11 = 2.995284926e-4, a12 = 2.040000000e-1;
  const double tc1 = 647.3, pc1 = 22.064e6, vc1 = 0.00317;
  double utc1 = 1.0 / tc1, upc1 = 1.0 / pc1, vc1mol = vc1 * FMWH2O;
  double theta = (t + 273.15) * utc1;
  double theta2x = theta * theta;
  double theta18 = pow(theta, 18.0);
  double theta20 = theta18 * theta2x;
  double beta = p * upc1;
  double beta2x = beta * beta;
  double yy = 1.0 - a1 * theta2x - a2 * pow(theta, -6.0);
  double xx = a3 * yy * yy - 2.0 * (a4 * theta - a5 * beta);
  if (xx > 0.0)
    xx = sqrt(xx);
  else
    return 1;
  double zz = yy + xx;
  double u0 = -5.0 / 17.0;
  double u1 = aa[11] * a5 * pow(zz, u0);
  double u2 = 1.0 / (a8 + pow(theta, 11.0));
  double u3 = aa[17] + (2.0 * aa[18] + 3.0 * aa[19] * beta) * beta;
  double u4 = 1.0 / (a7 + theta18 * theta);
  double u5 = pow(a10 + beta, -4.0);
  double u6 = a11 - 3.0 * u5;
  double u7 = aa[20] * theta18 * (a9 + theta2x);
  double u8 = aa[15] * pow(a6 - theta, 9.0);
  double vr = u1 + aa[12] + theta * (aa[13] + aa[14] * theta) + u8 * (a6 - theta) + aa[16] * u4 - u2 * u3 - u6 * u7 +
              (3.0 * aa[21] * (a12 - theta) + 4.0 * aa[22] * beta / theta20) * beta2x;
  *dwmol = 1.0 / (vr * vc1mol);
  double ypt = 6.0 * a2 * pow(theta, -7.0) - 2.0 * a1 * theta;
  double zpt = ypt + (a3 * yy * ypt - a4) / xx;
  double zpp = a5 / xx;
  double u9 = u0 * u1 / zz;
  double vrpt = u9 * zpt + aa[13] + 2.0 * aa[14] * theta - 10.0 * u8 - 19.0 * aa[16] * u4 * u4 * theta18 +
                11.0 * u2 * u2 * u3 * pow(theta, 10.0) - aa[20] * u6 * (18.0 * a9 * theta18 + 20.0 * theta20) / theta -
                (3.0 * aa[21] + 80.0 * aa[22] * beta / (theta20 * theta)) * beta2x;
  double vrpp = u9 * zpp - u2 * (2.0 * aa[18] + 6.0 * aa[19] * beta) - 12.0 * u7 * u5 / (a10 + beta) +
                (6.0 * aa[21] * (a12 - theta) + 12.0 * aa[22] * beta / theta20) * beta;
  double cnv = -1.0 / (vc1mol * vr * vr);
  *dwt = cnv * vrpt * utc1;
  *dwp = cnv * vrpp * upc1;
  return 0;
}

/* Density dispatch, EOSWaterMod.F90:38-77 */
void vo_density(double p, double t_K, int density_type, double *den, double *dden_dp, double *dden_dT) {
  switch (density_type) {
    case VO_DENSITY_CONSTANT: density_constant(den, dden_dp, dden_dT); break;
    case VO_DENSITY_TGDPB01: density_tgdpb01(p, t_K, den, dden_dp, dden_dT); break;
    case VO_DENSITY_IFC67:
      if (density_ifc67(t_K - 273.15, p, den, dden_dp, dden_dT)) { *den = *dden_dp = *dden_dT = NAN; }
      break;
    default: *den = *dden_dp = *dden_dT = NAN;
  }
}

/* Viscosity, EOSWaterMod.F90:568-586 */
void vo_viscosity(double p, double t_K, double *vis, double *dvis_dp, double *dvis_dT) {
  (void)p; (void)t_K;
  *vis = 8.904156e-4;
  *dvis_dp = 0.0;
  *dvis_dT = 0.0;
}

/* ------------------------------------------------------------------ */
/* Saturation functions: src/mpp/util/SaturationFunction.F90            */
/* ------------------------------------------------------------------ */
/* satparams: vsfm_oracle_internal.h */

/* findGu_SBC_zeroCoeff, SaturationFunction.F90:425-518 */
static double find_gu_sbc_zero_coeff(double lambda, int AA, double gs) {
  const double relTol = 1e-12;
  double gu = pow((double)AA / ((double)AA + lambda), -1.0 / lambda);
  if (gs > 0.0) {
    double guLeft = 1.0, guRight = gu;
    for (;;) {
      if (gu <= guLeft || gu >= guRight) gu = guLeft + 0.5 * (guRight - guLeft);
      double guInv = 1.0 / gu;
      double guToMinusLam = pow(gu, -lambda);
      double gsOnGu = gs * guInv;
      double resid = AA - guToMinusLam * (AA + lambda - lambda * gsOnGu);
      if (resid < 0.0) guLeft = gu; else guRight = gu;
      double dr_dGu = (1.0 + lambda) * (1.0 - gsOnGu) + (AA - 1);
      dr_dGu = lambda * guToMinusLam * guInv * dr_dGu;
      double deltaGu = resid / dr_dGu;
      gu = gu - deltaGu;
      if (fabs(deltaGu) < relTol * fabs(gu)) break;
    }
  }
  return gu;
}

/* SetSatFunc dispatch + SatFunc_Set_{VG,BC,SBC_bz2,SBC_bz3}, SaturationFunction.F90:127-372,1392-1428 */
static int satparams_set(satparams *s, int type, double alpha, double lambda, double sat_res) {
  memset(s, 0, sizeof *s);
  s->sat_res = sat_res;
  s->alpha = alpha;
  switch (type) {
    case VO_SAT_FUNC_VAN_GENUCHTEN:
      if (sat_res < 0.0 || sat_res > 0.5 || alpha <= 0.0 || alpha > 2.0 || lambda <= 0.0 || lambda >= 1.0) return -1;
      s->sat_func_type = VO_SAT_FUNC_VAN_GENUCHTEN;
      s->vg_m = lambda;
      s->vg_n = 1.0 / (1.0 - lambda);
      return 0;
    case VO_SAT_FUNC_BROOKS_COREY:
      if (sat_res < 0.0 || sat_res > 0.5 || alpha <= 0.0 || alpha > 2.0 || lambda <= 0.0 || lambda >= 2.0) return -1;
      s->sat_func_type = VO_SAT_FUNC_BROOKS_COREY;
      s->bc_lambda = lambda;
      return 0;
    case VO_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2:
    case VO_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3: {
      double ps = -0.9 / alpha; /* SaturationFunction.F90:1409,1412 */
      if (sat_res < 0.0 || sat_res > 0.5 || alpha <= 0.0 || alpha > 2.0 || lambda <= 0.0 || lambda >= 2.0 ||
          ps <= -1.0 / alpha || ps > 0.0)
        return -1;
      s->sat_func_type = VO_SAT_FUNC_SMOOTHED_BROOKS_COREY;
      s->bc_lambda = lambda;
      s->sbc_ps = ps;
      int bz2 = (type == VO_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2);
      double pu = find_gu_sbc_zero_coeff(lambda, bz2 ? 3 : 2, -alpha * ps) / (-alpha);
      s->sbc_pu = pu;
      double bcAtPu = pow(-alpha * pu, -lambda);
      double lambdaDeltaPuOnPu = lambda * (1.0 - ps / pu);
      double oneOnDeltaPu = 1.0 / (pu - ps);
      if (bz2) {
        s->sbc_b2 = 0.0;
        s->sbc_b3 = (2.0 - bcAtPu * (2.0 + lambdaDeltaPuOnPu)) * oneOnDeltaPu * oneOnDeltaPu * oneOnDeltaPu;
        if (s->sbc_b3 <= 0.0) return -1;
      } else {
        s->sbc_b2 = -(3.0 - bcAtPu * (3.0 + lambdaDeltaPuOnPu)) * oneOnDeltaPu * oneOnDeltaPu;
        if (s->sbc_b2 >= 0.0) return -1;
        s->sbc_b3 = 0.0;
      }
      return 0;
    }
    default: return -1;
  }
}

/* SatFunc_PressToSat -> PcToSat_{VG,BC,SBC}, SaturationFunction.F90:564-600,747-795,900-938,1027-1076 */
static void press_to_sat(const satparams *s, double press, double *sat, double *dsat_dP) {
  double pc = press - PRESSURE_REF;
  double sat_res = s->sat_res, alpha = s->alpha;
  switch (s->sat_func_type) {
    case VO_SAT_FUNC_VAN_GENUCHTEN: {
      double mm = s->vg_m, nn = s->vg_n;
      if (pc < 0.0) {
        double pc_alpha_n = pow(-alpha * pc, nn);
        double one_plus = 1.0 + pc_alpha_n;
        double Se = pow(one_plus, -mm);
        *sat = sat_res + (1.0 - sat_res) * Se;
        double AA = pc_alpha_n / one_plus;
        double dSe_dpc = -mm * nn * Se * AA / pc;
        *dsat_dP = (1.0 - sat_res) * dSe_dpc;
      } else {
        *sat = 1.0;
        *dsat_dP = 0.0;
      }
      break;
    }
    case VO_SAT_FUNC_BROOKS_COREY: {
      double lambda = s->bc_lambda;
      double pc_alpha = -alpha * pc;
      if (pc_alpha > 1.0) {
        double Se = pow(pc_alpha, -lambda);
        *sat = sat_res + (1.0 - sat_res) * Se;
        double dSe_dpc = -lambda * Se / pc;
        *dsat_dP = (1.0 - sat_res) * dSe_dpc;
      } else {
        *sat = 1.0;
        *dsat_dP = 0.0;
      }
      break;
    }
    case VO_SAT_FUNC_SMOOTHED_BROOKS_COREY: {
      double lambda = s->bc_lambda;
      if (pc <= s->sbc_pu) {
        double Se = pow(-alpha * pc, -lambda);
        *sat = sat_res + (1.0 - sat_res) * Se;
        double dSe_dpc = -lambda * Se / pc;
        *dsat_dP = (1.0 - sat_res) * dSe_dpc;
      } else if (pc < s->sbc_ps) {
        double deltaPc = pc - s->sbc_ps;
        double Se = 1.0 + deltaPc * deltaPc * (s->sbc_b2 + deltaPc * s->sbc_b3);
        *sat = sat_res + (1.0 - sat_res) * Se;
        double dSe_dpc = deltaPc * (2 * s->sbc_b2 + 3 * deltaPc * s->sbc_b3);
        *dsat_dP = (1.0 - sat_res) * dSe_dpc;
      } else {
        *sat = 1.0;
        *dsat_dP = 0.0;
      }
      break;
    }
    default: *sat = *dsat_dP = NAN;
  }
}

/* SatFunc_PressToRelPerm (Mualem) -> PcToRelPerm_{VG,BC,SBC}, SaturationFunction.F90:604-650,799-857,942-990,1080-1140 */
static void press_to_relperm(const satparams *s, double press, double frac_liq, double *kr, double *dkr_dP) {
  double pc = press - PRESSURE_REF;
  double alpha = s->alpha;
  switch (s->sat_func_type) {
    case VO_SAT_FUNC_VAN_GENUCHTEN: {
      double mm = s->vg_m, nn = s->vg_n;
      if (pc < 0.0) {
        double pc_alpha_n = pow(-alpha * pc, nn);
        double one_plus = 1.0 + pc_alpha_n;
        double Se = pow(one_plus, -mm);
        double AA = pc_alpha_n / one_plus;
        double dSe_dpc = -mm * nn * Se * AA / pc;
        double BB = 1.0 - pow(AA, mm);
        *kr = sqrt(Se) * BB * BB;
        double dkr_dSe = 0.5 * (*kr) / Se + 2.0 * pow(Se, 1.0 / mm - 0.5) * pow(AA, mm - 1.0) * BB;
        *dkr_dP = dkr_dSe * dSe_dpc;
      } else {
        *kr = 1.0;
        *dkr_dP = 0.0;
      }
      break;
    }
    case VO_SAT_FUNC_BROOKS_COREY: {
      double lambda = s->bc_lambda;
      double pc_alpha = -alpha * pc;
      if (pc_alpha > 1.0) {
        double Se = pow(pc_alpha, -lambda);
        double dSe_dpc = -lambda * Se / pc;
        *kr = pow(Se, 2.5 + 2.0 / lambda);
        double dkr_dSe = (2.5 + 2.0 / lambda) * (*kr) / Se;
        *dkr_dP = dkr_dSe * dSe_dpc;
      } else {
        *kr = 1.0;
        *dkr_dP = 0.0;
      }
      *kr = frac_liq * (*kr); /* SaturationFunction.F90:987-988 (BC only) */
      *dkr_dP = frac_liq * (*dkr_dP);
      break;
    }
    case VO_SAT_FUNC_SMOOTHED_BROOKS_COREY: {
      double lambda = s->bc_lambda;
      if (pc <= s->sbc_pu) {
        double Se = pow(-alpha * pc, -lambda);
        double dSe_dpc = -lambda * Se / pc;
        *kr = pow(Se, 2.5 + 2.0 / lambda);
        double dkr_dSe = (2.5 + 2.0 / lambda) * (*kr) / Se;
        *dkr_dP = dkr_dSe * dSe_dpc;
      } else if (pc < s->sbc_ps) {
        double deltaPc = pc - s->sbc_ps;
        double Se = 1.0 + deltaPc * deltaPc * (s->sbc_b2 + deltaPc * s->sbc_b3);
        double dSe_dpc = deltaPc * (2 * s->sbc_b2 + 3 * deltaPc * s->sbc_b3);
        *kr = pow(Se, 2.5 + 2.0 / lambda);
        double dkr_dSe = (2.5 + 2.0 / lambda) * (*kr) / Se;
        *dkr_dP = dkr_dSe * dSe_dpc;
      } else {
        *kr = 1.0;
        *dkr_dP = 0.0;
      }
      break;
    }
    default: *kr = *dkr_dP = NAN;
  }
}

/* ------------------------------------------------------------------ */
/* Auxiliary variable: src/mpp/auxvar/RichardsODEPressureAuxType.F90    */
/* ------------------------------------------------------------------ */
/* auxvar: vsfm_oracle_internal.h */

static void auxvar_init(auxvar *a) { /* RichODEPressureAuxVarInit :77-122 */
  memset(a, 0, sizeof *a);
  a->temperature = 273.15 + 25.0;
  a->frac_liq_sat = 1.0;
}

/* RichODEPressureAuxVarCompute, RichardsODEPressureAuxType.F90:237-294 */
static void auxvar_compute(auxvar *a, int density_type) {
  double dvis_dT;
  press_to_sat(&a->sp, a->pressure, &a->sat, &a->dsat_dP);
  press_to_relperm(&a->sp, a->pressure, a->frac_liq_sat, &a->kr, &a->dkr_dP);
  vo_density(a->pressure, a->temperature, density_type, &a->den, &a->dden_dP, &a->dden_dT);
  vo_viscosity(a->pressure, a->temperature, &a->vis, &a->dvis_dP, &dvis_dT);
  /* PorosityFunctionComputation, PorosityFunctionMod.F90:99-163 */
  if (a->por_model == 2) { /* PorosityFunctionComputationLinearModel :144-163 */
    a->por = a->por_base + (a->pressure - a->por_pressure_reference) * a->por_lin_mod_slope;
    a->dpor_dP = a->por_lin_mod_slope;
  } else { /* PorosityFunctionComputationConstantModel :126-141 */
    a->por = a->por_base;
    a->dpor_dP = 0.0;
  }
}

void vo_auxvar_compute(int sat_func_type, double sat_res, double alpha, double lambda_or_m, double por_base,
                       int density_type, double temperature, double frac_liq_sat, double pressure, double *out8) {
  auxvar a;
  auxvar_init(&a);
  if (satparams_set(&a.sp, sat_func_type, alpha, lambda_or_m, sat_res)) {
    for (int i = 0; i < 8; i++) out8[i] = NAN;
    return;
  }
  a.por_base = por_base;
  a.temperature = temperature;
  a.frac_liq_sat = frac_liq_sat;
  a.pressure = pressure;
  auxvar_compute(&a, density_type);
  out8[0] = a.sat; out8[1] = a.dsat_dP; out8[2] = a.kr; out8[3] = a.dkr_dP;
  out8[4] = a.den; out8[5] = a.dden_dP; out8[6] = a.por; out8[7] = a.dpor_dP;
}

/* ------------------------------------------------------------------ */
/* Two-point flux: src/mpp/ge/RichardsMod.F90                           */
/* ------------------------------------------------------------------ */
/* conn_geom: vsfm_oracle_internal.h */

/* RichardsFlux_Internal, RichardsMod.F90:118-340.  Note dflux_dP_* are the NEGATIVES of d(flux)/dP (:326-334). */
static void richards_flux_internal(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv,
                                   int internal_conn, int cond_type, double *flux, double *dflux_dP_up,
                                   double *dflux_dP_dn) {
  double area = g->area, dist_up = g->dist_up, dist_dn = g->dist_dn;
  const double *u = g->uvec;
  double perm_up = fabs(u[0]) * up->perm[0] + fabs(u[1]) * up->perm[1] + fabs(u[2]) * up->perm[2];
  double perm_dn = fabs(u[0]) * dn->perm[0] + fabs(u[1]) * dn->perm[1] + fabs(u[2]) * dn->perm[2];
  double upweight, Dq;
  if (internal_conn || cond_type == VO_COND_DIRICHLET_FRM_OTR_GOVEQ) {
    upweight = dist_up / (dist_up + dist_dn);
    Dq = (perm_up * perm_dn) / (dist_up * perm_dn + dist_dn * perm_up);
  } else { /* COND_DIRICHLET, COND_MASS_FLUX, COND_SEEPAGE_BC : :249-252 */
    upweight = 0.0;
    Dq = perm_dn / (dist_up + dist_dn);
  }
  double grav_vec[3] = {0.0, 0.0, -GRAVITY_CONSTANT};
  double udist_dot_ugrav = u[0] * grav_vec[0] + u[1] * grav_vec[1] + u[2] * grav_vec[2];
  double dist_gravity = (dist_up + dist_dn) * udist_dot_ugrav;
  double den_ave = upweight * up->den + (1.0 - upweight) * dn->den;
  double gravityterm = (upweight * up->den + (1.0 - upweight) * dn->den) * FMWH2O * dist_gravity;
  double dphi = up->pressure - dn->pressure + gravityterm;
  int seepage_bc_update = 0;
  if (!internal_conn && cond_type == VO_COND_SEEPAGE_BC) {
    if (dphi > 0.0 && up->pressure <= PRESSURE_REF) seepage_bc_update = 1;
  }
  if (seepage_bc_update) dphi = 0.0;
  double ukvr = (dphi >= 0.0) ? up->kr / up->vis : dn->kr / dn->vis;
  double v_darcy;
  int massflux_bc = (!internal_conn && cond_type == VO_COND_MASS_FLUX);
  if (massflux_bc) v_darcy = 0.0; else v_darcy = -Dq * ukvr * dphi;
  double q = v_darcy * area;
  *flux = q * den_ave;
  if (compute_deriv) {
    double dden_ave_dP_up = upweight * up->dden_dP;
    double dden_ave_dP_dn = (1.0 - upweight) * dn->dden_dP;
    double dgrav_dden_up = upweight * dist_gravity * FMWH2O;
    double dgrav_dden_dn = (1.0 - upweight) * dist_gravity * FMWH2O;
    double dphi_dP_up = 1.0 + dgrav_dden_up * up->dden_dP;
    double dphi_dP_dn = -1.0 + dgrav_dden_dn * dn->dden_dP;
    if (seepage_bc_update) dphi_dP_dn = 0.0;
    double dukvr_dP_up, dukvr_dP_dn;
    if (dphi >= 0) {
      dukvr_dP_up = up->dkr_dP / up->vis - up->kr / (up->vis * up->vis) * up->dvis_dP;
      dukvr_dP_dn = 0.0;
    } else {
      dukvr_dP_up = 0.0;
      dukvr_dP_dn = dn->dkr_dP / dn->vis - dn->kr / (dn->vis * dn->vis) * dn->dvis_dP;
    }
    double dq_dP_up = Dq * (dukvr_dP_up * dphi + ukvr * dphi_dP_up) * area;
    double dq_dP_dn = Dq * (dukvr_dP_dn * dphi + ukvr * dphi_dP_dn) * area;
    if (massflux_bc) {
      *dflux_dP_up = 0.0;
      *dflux_dP_dn = 0.0;
    } else {
      *dflux_dP_up = dq_dP_up * den_ave - q * dden_ave_dP_up;
      *dflux_dP_dn = dq_dP_dn * den_ave - q * dden_ave_dP_dn;
    }
  } else {
    *dflux_dP_up = 0.0;
    *dflux_dP_dn = 0.0;
  }
}

/* RichardsFlux (swap_order wrapper), RichardsMod.F90:29-114 */
static void richards_flux(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv, int internal_conn,
                          int swap_order, int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn) {
  double f, dup, ddn;
  if (!swap_order) {
    richards_flux_internal(up, dn, g, compute_deriv, internal_conn, cond_type, &f, &dup, &ddn);
    *flux = f; *dflux_dP_up = dup; *dflux_dP_dn = ddn;
  } else {
    richards_flux_internal(dn, up, g, compute_deriv, internal_conn, cond_type, &f, &dup, &ddn);
    *flux = -f; *dflux_dP_up = -ddn; *dflux_dP_dn = -dup;
  }
}

void vo_richards_flux(const double *aux_up, const double *perm_up, const double *aux_dn, const double *perm_dn,
                      double area, double dist_up, double dist_dn, const double *unitvec, int internal_conn,
                      int swap_order, int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn) {
  auxvar up, dn;
  conn_geom g;
  auxvar_init(&up); auxvar_init(&dn);
  double dT;
  up.pressure = aux_up[0]; up.kr = aux_up[1]; up.dkr_dP = aux_up[2]; up.den = aux_up[3]; up.dden_dP = aux_up[4];
  dn.pressure = aux_dn[0]; dn.kr = aux_dn[1]; dn.dkr_dP = aux_dn[2]; dn.den = aux_dn[3]; dn.dden_dP = aux_dn[4];
  vo_viscosity(0, 0, &up.vis, &up.dvis_dP, &dT);
  vo_viscosity(0, 0, &dn.vis, &dn.dvis_dP, &dT);
  memcpy(up.perm, perm_up, sizeof up.perm);
  memcpy(dn.perm, perm_dn, sizeof dn.perm);
  g.area = area; g.dist_up = dist_up; g.dist_dn = dist_dn;
  memcpy(g.uvec, unitvec, sizeof g.uvec);
  richards_flux(&up, &dn, &g, 1, internal_conn, swap_order, cond_type, flux, dflux_dP_up, dflux_dP_dn);
}

/* ------------------------------------------------------------------ */
/* Problem object                                                      */
/* ------------------------------------------------------------------ */
/* condition: vsfm_oracle_internal.h */

/* struct vo_problem: vsfm_oracle_internal.h */

static void jacobian_assemble(vo_problem *p);

vo_problem *vo_create(int ncells) {
  vo_problem *p = (vo_problem *)calloc(1, sizeof *p);
  p->n = ncells;
  p->residual_fn = vo_residual;
  p->jacobian_fn = jacobian_assemble;
  p->vol = (double *)calloc(ncells, sizeof(double));
  p->active = (int *)malloc(ncells * sizeof(int));
  p->aux = (auxvar *)malloc(ncells * sizeof(auxvar));
  for (int i = 0; i < ncells; i++) { p->active[i] = 1; auxvar_init(&p->aux[i]); }
  p->density_type = VO_DENSITY_CONSTANT;
  p->soln = (double *)calloc(ncells, sizeof(double));
  p->soln_prev = (double *)calloc(ncells, sizeof(double));
  p->accum_prev = (double *)calloc(ncells, sizeof(double));
  p->soln_prev_clm = (double *)calloc(ncells, sizeof(double));
  p->lat_mass_exc = (double *)calloc(ncells, sizeof(double));
  /* PETSc 3.16 defaults; stol from MultiPhysicsProbBaseType.F90:1112 */
  p->ksp_type = VO_KSP_GMRES; p->pc_type = VO_PC_ILU0;
  p->ksp_rtol = 1e-5; p->ksp_atol = 1e-50; p->ksp_dtol = 1e4; p->ksp_max_it = 10000; p->gmres_restart = 30;
  p->snes_rtol = 1e-8; p->snes_atol = 1e-50; p->snes_stol = 1e-10; p->snes_divtol = 1e4;
  p->snes_max_it = 50; p->snes_max_funcs = 10000;
  p->linesearch_type = 2; p->use_dynamic_linesearch = 0; p->ls_ynorm = -1.0;
  return p;
}

int vo_set_lateral_connections(vo_problem *p, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                               const double *dist_dn, const double *area, const double *unitvec) {
  free(p->lat_up); free(p->lat_dn); free(p->lat_geom);
  p->nlat = nconn;
  p->lat_up = (int *)malloc((nconn + 1) * sizeof(int)); p->lat_dn = (int *)malloc((nconn + 1) * sizeof(int));
  p->lat_geom = (conn_geom *)calloc(nconn + 1, sizeof(conn_geom));
  for (int c = 0; c < nconn; c++) {
    if (id_up[c] < 1 || id_up[c] > p->n || id_dn[c] < 1 || id_dn[c] > p->n) FAIL("lateral connection: cell id out of range");
    p->lat_up[c] = id_up[c] - 1; p->lat_dn[c] = id_dn[c] - 1;
    p->lat_geom[c].area = area[c]; p->lat_geom[c].dist_up = dist_up[c]; p->lat_geom[c].dist_dn = dist_dn[c];
    for (int k = 0; k < 3; k++) p->lat_geom[c].uvec[k] = unitvec[3 * c + k];
  }
  return 0;
}

/* VSFMComputeLateralFlux + RichardsODEComputeLateralFlux */
int vo_compute_lateral_flux(vo_problem *p, double dt, int ss_cond_id) {
  if (ss_cond_id < 1 || ss_cond_id > p->nss) FAIL("RichardsODEComputeLateralFlux: Lateral source-sink not found.");
  condition *lc = &p->ss[ss_cond_id - 1];
  if (lc->cond_type != VO_COND_MASS_RATE || lc->n != p->n) FAIL("the lateral-flux condition must be a COND_MASS_RATE over every cell");
  for (int i = 0; i < lc->n; i++) if (lc->id_dn[i] != i) FAIL("the lateral-flux condition must list the cells in order");
  vo_set_dtime(p, dt);
  vo_pre_solve(p);
  double flux, d1, d2;
  for (int i = 0; i < lc->n; i++) lc->value[i] = 0.0;
  for (int c = 0; c < p->nlat; c++) {
    int u = p->lat_up[c], d = p->lat_dn[c];
    if (!p->active[u] || !p->active[d]) continue;
    richards_flux(&p->aux[u], &p->aux[d], &p->lat_geom[c], 0, 1, 0, VO_COND_NULL, &flux, &d1, &d2);
    lc->value[u] = lc->value[u] + flux * FMWH2O;
    lc->value[d] = lc->value[d] - flux * FMWH2O;
  }
  return 0;
}

int vo_get_condition_values(vo_problem *p, int kind, int cond_id, int n, double *values) {
  int cnt = kind == VO_COND_BC ? p->nbc : p->nss;
  if (cond_id < 1 || cond_id > cnt) FAIL("bad condition id");
  condition *c = kind == VO_COND_BC ? &p->bc[cond_id - 1] : &p->ss[cond_id - 1];
  if (n > c->n) FAIL("size(data_1d) > nauxvar");
  for (int i = 0; i < n; i++) values[i] = c->value[i];
  return 0;
}

int vo_set_source_sink_auxvar(vo_problem *p, int var_type, int n, const double *values) {
  if (var_type != VO_VAR_POT_MASS_SINK_PRESSURE && var_type != VO_VAR_POT_MASS_SINK_EXPONENT)
    FAIL("VSFMMPPSetSourceSinkAuxVarRealValue: Unknown var_type");
  int sum_conn = 0;
  for (int s = 0; s < p->nss; s++)
    for (int i = 0; i < p->ss[s].n; i++, sum_conn++) {
      if (sum_conn >= n) FAIL("fewer values than source-sink connections");
      (var_type == VO_VAR_POT_MASS_SINK_PRESSURE ? p->ss[s].sink_pressure : p->ss[s].sink_exponent)[i] = values[sum_conn];
    }
  return 0;
}

static void free_cond(condition *c) {
  free(c->id_dn); free(c->geom); free(c->aux); free(c->value); free(c->flux);
  free(c->sink_pressure); free(c->sink_exponent); free(c->mass_exc);
}

void vo_destroy(vo_problem *p) {
  if (!p) return;
  free(p->vol); free(p->active); free(p->aux); free(p->up); free(p->dn); free(p->geom); free(p->internal_flux);
  for (int i = 0; i < p->nbc; i++) free_cond(&p->bc[i]);
  for (int i = 0; i < p->nss; i++) free_cond(&p->ss[i]);
  free(p->bc); free(p->ss);
  free(p->rowptr); free(p->col); free(p->diag_slot); free(p->slot_uu); free(p->slot_ud); free(p->slot_du);
  free(p->slot_dd); free(p->val); free(p->soln); free(p->soln_prev); free(p->accum_prev);
  free(p->block_ptr); free(p->block_cells);
  free(p->lat_up); free(p->lat_dn); free(p->lat_geom);
  free(p->horizontal); free(p->lat_mass_exc); free(p->soln_prev_clm);
  free(p);
}

int vo_set_mesh(vo_problem *p, const double *vol, const int *is_active) {
  memcpy(p->vol, vol, p->n * sizeof(double));
  for (int i = 0; i < p->n; i++) p->active[i] = is_active ? (is_active[i] != 0) : 1;
  return 0;
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

static int find_slot(const vo_problem *p, int row, int col) {
  int lo = p->rowptr[row], hi = p->rowptr[row + 1] - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    if (p->col[mid] == col) return mid;
    if (p->col[mid] < col) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/* CSR pattern = diagonal + both directions of every internal connection (the entries
 * MatSetValuesLocal creates at GoveqnRichardsODEPressureType.F90:1689,2051-2069) */
static int build_pattern(vo_problem *p) {
  int n = p->n, nc = p->nc;
  int *cnt = (int *)calloc(n + 1, sizeof(int));
  for (int i = 0; i < n; i++) cnt[i + 1] = 1;
  for (int c = 0; c < nc; c++) { cnt[p->up[c] + 1]++; cnt[p->dn[c] + 1]++; }
  for (int i = 0; i < n; i++) cnt[i + 1] += cnt[i];
  int cap = cnt[n];
  int *col = (int *)malloc(cap * sizeof(int));
  int *fill = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) { fill[i] = cnt[i]; col[fill[i]++] = i; }
  for (int c = 0; c < nc; c++) { col[fill[p->up[c]]++] = p->dn[c]; col[fill[p->dn[c]]++] = p->up[c]; }
  free(p->rowptr); free(p->col); free(p->val); free(p->diag_slot);
  p->rowptr = (int *)malloc((n + 1) * sizeof(int));
  p->col = (int *)malloc(cap * sizeof(int));
  int nnz = 0;
  for (int i = 0; i < n; i++) {
    p->rowptr[i] = nnz;
    int len = cnt[i + 1] - cnt[i];
    qsort(col + cnt[i], len, sizeof(int), cmp_int);
    for (int k = 0; k < len; k++) {
      int cc = col[cnt[i] + k];
      if (k == 0 || cc != col[cnt[i] + k - 1]) p->col[nnz++] = cc;
    }
  }
  p->rowptr[n] = nnz;
  free(col); free(fill); free(cnt);
  p->val = (double *)calloc(nnz, sizeof(double));
  p->diag_slot = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) p->diag_slot[i] = find_slot(p, i, i);
  free(p->slot_uu); free(p->slot_ud); free(p->slot_du); free(p->slot_dd);
  p->slot_uu = (int *)malloc((nc + 1) * sizeof(int)); p->slot_ud = (int *)malloc((nc + 1) * sizeof(int));
  p->slot_du = (int *)malloc((nc + 1) * sizeof(int)); p->slot_dd = (int *)malloc((nc + 1) * sizeof(int));
  for (int c = 0; c < nc; c++) {
    int u = p->up[c], d = p->dn[c];
    p->slot_uu[c] = p->diag_slot[u]; p->slot_ud[c] = find_slot(p, u, d);
    p->slot_du[c] = find_slot(p, d, u); p->slot_dd[c] = p->diag_slot[d];
  }
  return 0;
}

int vo_set_connections(vo_problem *p, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                       const double *dist_dn, const double *area, const double *unitvec) {
  free(p->up); free(p->dn); free(p->geom); free(p->internal_flux); free(p->horizontal);
  p->nc = nconn;
  p->horizontal = (int *)calloc(nconn + 1, sizeof(int));
  p->up = (int *)malloc((nconn + 1) * sizeof(int));
  p->dn = (int *)malloc((nconn + 1) * sizeof(int));
  p->geom = (conn_geom *)malloc((nconn + 1) * sizeof(conn_geom));
  p->internal_flux = (double *)calloc(nconn + 1, sizeof(double));
  for (int c = 0; c < nconn; c++) {
    if (id_up[c] < 1 || id_up[c] > p->n || id_dn[c] < 1 || id_dn[c] > p->n) FAIL("connection %d: cell id out of range", c + 1);
    p->up[c] = id_up[c] - 1; p->dn[c] = id_dn[c] - 1;
    p->geom[c].area = area[c]; p->geom[c].dist_up = dist_up[c]; p->geom[c].dist_dn = dist_dn[c];
    for (int k = 0; k < 3; k++) p->geom[c].uvec[k] = unitvec[3 * c + k];
    /* conn%GetType(): the mesh builders tag a connection CONN_HORIZONTAL / CONN_VERTICAL by its direction
     * (mpp_mesh_utils.F90:430-466); here from the unit vector */
    const double *u = p->geom[c].uvec;
    p->horizontal[c] = !(fabs(u[2]) > 0.5 * sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
  }
  return build_pattern(p);
}

int vo_add_condition(vo_problem *p, int kind, int cond_type, int nconn, const int *id_dn, const double *dist_up,
                     const double *dist_dn, const double *area, const double *unitvec, int swap_order) {
  condition **list; int *cnt;
  if (kind == VO_COND_BC) { list = &p->bc; cnt = &p->nbc; }
  else if (kind == VO_COND_SS) { list = &p->ss; cnt = &p->nss; }
  else FAIL("unknown condition kind %d", kind);
  if (kind == VO_COND_BC && cond_type != VO_COND_DIRICHLET && cond_type != VO_COND_SEEPAGE_BC &&
      cond_type != VO_COND_MASS_FLUX)
    FAIL("unsupported BC cond_type %d", cond_type);
  if (kind == VO_COND_SS && cond_type != VO_COND_MASS_RATE && cond_type != VO_COND_DOWNREG_MASS_RATE_CAMPBELL &&
      cond_type != VO_COND_DOWNREG_MASS_RATE_FETCH2)
    FAIL("unsupported SS cond_type %d", cond_type);
  *list = (condition *)realloc(*list, (*cnt + 1) * sizeof(condition));
  condition *c = &(*list)[*cnt];
  memset(c, 0, sizeof *c);
  c->kind = kind; c->cond_type = cond_type; c->n = nconn; c->swap_order = swap_order;
  c->id_dn = (int *)malloc((nconn + 1) * sizeof(int));
  c->geom = (conn_geom *)calloc(nconn + 1, sizeof(conn_geom));
  c->aux = (auxvar *)malloc((nconn + 1) * sizeof(auxvar));
  c->value = (double *)calloc(nconn + 1, sizeof(double));
  c->flux = (double *)calloc(nconn + 1, sizeof(double));
  c->mass_exc = (double *)calloc(nconn + 1, sizeof(double));
  c->sink_pressure = (double *)calloc(nconn + 1, sizeof(double));
  c->sink_exponent = (double *)calloc(nconn + 1, sizeof(double));
  for (int i = 0; i < nconn; i++) {
    if (id_dn[i] < 1 || id_dn[i] > p->n) FAIL("condition: cell id out of range");
    c->id_dn[i] = id_dn[i] - 1;
    auxvar_init(&c->aux[i]);
    if (kind == VO_COND_BC) {
      c->geom[i].area = area[i]; c->geom[i].dist_up = dist_up[i]; c->geom[i].dist_dn = dist_dn[i];
      for (int k = 0; k < 3; k++) c->geom[i].uvec[k] = unitvec[3 * i + k];
    }
  }
  (*cnt)++;
  return *cnt;
}

/* BC/SS auxvars take their soil parameters from the id_dn cell: MultiPhysicsProbVSFM.F90:422-471 */
static void copy_soil_to_conditions(vo_problem *p) {
  for (int b = 0; b < p->nbc; b++)
    for (int i = 0; i < p->bc[b].n; i++) {
      auxvar *a = &p->bc[b].aux[i]; const auxvar *s = &p->aux[p->bc[b].id_dn[i]];
      memcpy(a->perm, s->perm, sizeof a->perm); a->sp = s->sp;
      a->por_model = s->por_model; a->por_base = s->por_base;
      a->por_pressure_reference = s->por_pressure_reference; a->por_lin_mod_slope = s->por_lin_mod_slope;
    }
}

int vo_set_soil_permeability(vo_problem *p, const double *px, const double *py, const double *pz) {
  for (int i = 0; i < p->n; i++) { p->aux[i].perm[0] = px[i]; p->aux[i].perm[1] = py[i]; p->aux[i].perm[2] = pz[i]; }
  copy_soil_to_conditions(p);
  return 0;
}
int vo_set_soil_porosity(vo_problem *p, const double *por) { /* PorosityFunctionSetConstantModel :57-72 */
  for (int i = 0; i < p->n; i++) { p->aux[i].por_model = 1; p->aux[i].por_base = por[i]; }
  copy_soil_to_conditions(p);
  return 0;
}
/* PorosityFunctionSetLinearModel (PorosityFunctionMod.F90:75-96) for every cell; no MPP driver calls it (the porParams of
 * an auxvar are reachable through the auxvar only), so this entry point has no MPP-level counterpart */
int vo_set_soil_porosity_linear(vo_problem *p, const double *por_base, const double *press_ref, const double *slope) {
  for (int i = 0; i < p->n; i++) {
    p->aux[i].por_model = 2; p->aux[i].por_base = por_base[i];
    p->aux[i].por_pressure_reference = press_ref[i]; p->aux[i].por_lin_mod_slope = slope[i];
  }
  copy_soil_to_conditions(p);
  return 0;
}
int vo_set_saturation_function(vo_problem *p, const int *type, const double *alpha, const double *lambda,
                               const double *sat_res) {
  for (int i = 0; i < p->n; i++)
    if (satparams_set(&p->aux[i].sp, type[i], alpha[i], lambda[i], sat_res[i])) FAIL("bad saturation parameters at cell %d", i + 1);
  copy_soil_to_conditions(p);
  return 0;
}
int vo_set_density_type(vo_problem *p, int t) {
  if (t < 1 || t > 3) FAIL("unknown density type %d", t);
  p->density_type = t;
  return 0;
}

/* VSFMSOESetDataFromCLM(AUXVAR_BC|AUXVAR_SS, VAR_BC_SS_CONDITION, id, data), SystemOfEquationsVSFMType.F90:663-724 */
int vo_set_condition_values(vo_problem *p, int kind, int cond_id, int n, const double *values) {
  condition *c;
  if (kind == VO_COND_BC) { if (cond_id < 1 || cond_id > p->nbc) FAIL("bad bc id"); c = &p->bc[cond_id - 1]; }
  else if (kind == VO_COND_SS) { if (cond_id < 1 || cond_id > p->nss) FAIL("bad ss id"); c = &p->ss[cond_id - 1]; }
  else FAIL("bad kind");
  if (n > c->n) FAIL("size(data_1d) > nauxvar");
  for (int i = 0; i < n; i++) {
    if (kind == VO_COND_BC) c->aux[i].condition_value = values[i]; else c->value[i] = values[i];
  }
  return 0;
}

int vo_pre_step_dt(vo_problem *p) {
  memcpy(p->soln_prev, p->soln_prev_clm, p->n * sizeof(double));
  memcpy(p->soln, p->soln_prev_clm, p->n * sizeof(double));
  memset(p->lat_mass_exc, 0, p->n * sizeof(double));
  for (int b = 0; b < p->nbc; b++) memset(p->bc[b].mass_exc, 0, p->bc[b].n * sizeof(double));
  return 0;
}
int vo_post_step_dt(vo_problem *p) { memcpy(p->soln_prev_clm, p->soln_prev, p->n * sizeof(double)); return 0; }
int vo_set_frac_liq_sat(vo_problem *p, const double *frac) {
  for (int i = 0; i < p->n; i++) if (p->active[i]) p->aux[i].frac_liq_sat = frac[i];
  return 0;
}

/* VSFMMPPRestart, MultiPhysicsProbVSFM.F90:603-707 */
int vo_restart(vo_problem *p, const double *pressure) {
  memcpy(p->soln, pressure, p->n * sizeof(double));
  memcpy(p->soln_prev, pressure, p->n * sizeof(double));
  memcpy(p->soln_prev_clm, pressure, p->n * sizeof(double));   /* the drivers VecCopy soln into soln_prev_clm after Restart */
  return 0;
}

int vo_set_blocks(vo_problem *p, int nblocks, const int *block_ptr, const int *block_cells) {
  free(p->block_ptr); free(p->block_cells);
  p->nblocks = nblocks;
  p->block_ptr = (int *)malloc((nblocks + 1) * sizeof(int));
  memcpy(p->block_ptr, block_ptr, (nblocks + 1) * sizeof(int));
  int tot = block_ptr[nblocks];
  p->block_cells = (int *)malloc((tot + 1) * sizeof(int));
  for (int i = 0; i < tot; i++) p->block_cells[i] = block_cells[i] - 1;
  return 0;
}

int vo_set_solver(vo_problem *p, int ksp_type, int pc_type, double ksp_rtol, double snes_rtol, double snes_atol,
                  double snes_stol, int snes_max_it) {
  p->ksp_type = ksp_type; p->pc_type = pc_type;
  if (ksp_rtol > 0) p->ksp_rtol = ksp_rtol;
  if (snes_rtol > 0) p->snes_rtol = snes_rtol;
  if (snes_atol > 0) p->snes_atol = snes_atol;
  if (snes_stol > 0) p->snes_stol = snes_stol;
  if (snes_max_it > 0) p->snes_max_it = snes_max_it;
  return 0;
}

/* -ksp_gmres_restart (PETSc takes any positive restart length: KSPSetFromOptions, MultiPhysicsProbBaseType.F90:1205, 1317) */
int vo_set_gmres_restart(vo_problem *p, int restart) {
  if (restart < 1) return 1;
  p->gmres_restart = restart;
  return 0;
}

/* ------------------------------------------------------------------ */
/* Governing equation: GoveqnRichardsODEPressureType.F90                */
/* ------------------------------------------------------------------ */

/* SavePrimaryIndependentVar + UpdateAuxVarsIntrn + UpdateAuxVarsBC (:504-534, 1417-1552) */
static void update_auxvars(vo_problem *p, const double *x) {
  int n = p->n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    p->aux[i].pressure = x[i];
    if (p->active[i]) auxvar_compute(&p->aux[i], p->density_type);
  }
  for (int b = 0; b < p->nbc; b++) {
    condition *c = &p->bc[b];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->n; i++) {
      if (c->cond_type == VO_COND_DIRICHLET || c->cond_type == VO_COND_SEEPAGE_BC)
        c->aux[i].pressure = c->aux[i].condition_value;
      else /* COND_MASS_RATE, COND_MASS_FLUX :1521-1524 */
        c->aux[i].pressure = p->aux[c->id_dn[i]].pressure;
      auxvar_compute(&c->aux[i], p->density_type);
    }
  }
}

/* RichardsODEPressureAccum (:1603-1634) */
static void accumulation(const vo_problem *p, double *ff) {
  double dtInv = 1.0 / p->dtime;
  int n = p->n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    const auxvar *a = &p->aux[i];
    ff[i] = p->active[i] ? a->por * a->den * a->sat * p->vol[i] * dtInv : 0.0;
  }
}

/* RichardsODEPressureDivergence (:1696-1938): serial scatter-add in connection order, like the reference */
static void divergence(vo_problem *p, double *ff) {
  double d1, d2, flux;
  for (int c = 0; c < p->nc; c++) {
    int u = p->up[c], d = p->dn[c];
    if (!p->active[u] || !p->active[d]) continue;
    richards_flux(&p->aux[u], &p->aux[d], &p->geom[c], 0, 1, 0, VO_COND_NULL, &flux, &d1, &d2);
    ff[u] -= flux;
    ff[d] += flux;
    p->internal_flux[c] = flux * FMWH2O;
  }
  for (int b = 0; b < p->nbc; b++) {
    condition *cd = &p->bc[b];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->active[cell]) continue;
      richards_flux(&cd->aux[i], &p->aux[cell], &cd->geom[i], 0, 0, cd->swap_order, cd->cond_type, &flux, &d1, &d2);
      ff[cell] += flux;
      cd->flux[i] = flux * FMWH2O;
    }
  }
  for (int s = 0; s < p->nss; s++) {
    condition *cd = &p->ss[s];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->active[cell]) continue; /* active_conn_ids */
      if (cd->cond_type == VO_COND_MASS_RATE) {
        ff[cell] -= cd->value[i] / FMWH2O; /* :1896 */
        cd->flux[i] = cd->value[i];
      } else { /* down-regulated potential sinks :1900-1927 */
        double dP = p->aux[cell].pressure - PRESSURE_REF, Pc = cd->sink_pressure[i], n = cd->sink_exponent[i], factor;
        if (cd->cond_type == VO_COND_DOWNREG_MASS_RATE_CAMPBELL) {
          factor = (dP <= 0.0) ? 1.0 + pow(dP / Pc, n) : 1.0;
          ff[cell] = ff[cell] - cd->value[i] / factor / FMWH2O;
          cd->flux[i] = cd->value[i] / factor;
        } else {
          factor = (dP <= 0.0) ? exp(-pow(dP / Pc, n)) : 1.0;
          ff[cell] = ff[cell] - cd->value[i] * factor / FMWH2O;
          cd->flux[i] = cd->value[i] * factor;
        }
      }
    }
  }
}

int vo_set_dtime(vo_problem *p, double dt) { p->dtime = dt; return 0; }

/* VSFMSOEPreSolve + RichardsODEPressurePreSolve (SystemOfEquationsVSFMType.F90:506-563; GE:2747-2768) */
int vo_pre_solve(vo_problem *p) {
  update_auxvars(p, p->soln_prev);
  accumulation(p, p->accum_prev);
  return 0;
}

/* VSFMSOEResidual + RichardsODEComputeResidual (SoE VSFM :94-212; GE :388-421) */
int vo_residual(vo_problem *p, const double *x, double *f) {
  if (!p->rowptr) FAIL("connections not set");
  update_auxvars(p, x);
  accumulation(p, f);
  for (int i = 0; i < p->n; i++) f[i] = f[i] - p->accum_prev[i];
  divergence(p, f);
  p->nfuncs++; p->cum_funcs++;
  return 0;
}

int vo_jacobian_nnz(vo_problem *p) { return p->rowptr ? p->rowptr[p->n] : -1; }

/* RichardsODEComputeJacobian = DivergenceDeriv + AccumDeriv (:425-453, 1638-1693, 1941-2200), auxvars as left by the last residual */
static void jacobian_assemble(vo_problem *p) {
  int nnz = p->rowptr[p->n];
  double flux, Jup, Jdn;
  memset(p->val, 0, nnz * sizeof(double));
  for (int c = 0; c < p->nc; c++) {
    int u = p->up[c], d = p->dn[c];
    if (!p->active[u] || !p->active[d]) continue;
    richards_flux(&p->aux[u], &p->aux[d], &p->geom[c], 1, 1, 0, VO_COND_NULL, &flux, &Jup, &Jdn);
    p->val[p->slot_uu[c]] += Jup;
    p->val[p->slot_ud[c]] += Jdn;
    p->val[p->slot_du[c]] += -Jup;
    p->val[p->slot_dd[c]] += -Jdn;
  }
  for (int b = 0; b < p->nbc; b++) {
    condition *cd = &p->bc[b];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->active[cell]) continue;
      richards_flux(&cd->aux[i], &p->aux[cell], &cd->geom[i], 1, 0, cd->swap_order, cd->cond_type, &flux, &Jup, &Jdn);
      p->val[p->diag_slot[cell]] += -Jdn;
    }
  }
  for (int s = 0; s < p->nss; s++) { /* :2141-2195 */
    condition *cd = &p->ss[s];
    if (cd->cond_type == VO_COND_MASS_RATE) continue;
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->active[cell]) continue;
      double dP = p->aux[cell].pressure - PRESSURE_REF, Pc = cd->sink_pressure[i], n = cd->sink_exponent[i];
      if (dP <= 0.0) {
        double val;
        if (cd->cond_type == VO_COND_DOWNREG_MASS_RATE_CAMPBELL) {
          double factor = 1.0 + pow(dP / Pc, n);
          val = (cd->value[i] / FMWH2O) * (n * pow(dP / Pc, n)) / (dP * pow(factor, 2.0));
        } else {
          double factor = exp(-pow(dP / Pc, n));
          val = (cd->value[i] / FMWH2O) * (n * pow(dP / Pc, n)) * factor / dP;
        }
        p->val[p->diag_slot[cell]] += val;
      }
    }
  }
  double dtInv = 1.0 / p->dtime;
  for (int i = 0; i < p->n; i++) {
    const auxvar *a = &p->aux[i];
    double derivative;
    if (p->active[i])
      derivative = (a->dpor_dP * a->den * a->sat + a->por * a->dden_dP * a->sat + a->por * a->den * a->dsat_dP) * p->vol[i] * dtInv;
    else
      derivative = 1.0; /* :1682-1684 */
    p->val[p->diag_slot[i]] += derivative;
  }
}

int vo_jacobian(vo_problem *p, const double *x, int *rowptr, int *col, double *val) {
  if (!p->rowptr) FAIL("connections not set");
  update_auxvars(p, x);
  jacobian_assemble(p);
  int nnz = p->rowptr[p->n];
  memcpy(rowptr, p->rowptr, (p->n + 1) * sizeof(int));
  memcpy(col, p->col, nnz * sizeof(int));
  memcpy(val, p->val, nnz * sizeof(double));
  return 0;
}

/* ------------------------------------------------------------------ */
/* Linear algebra (PETSc v3.16.2 restated)                              */
/* ------------------------------------------------------------------ */
int vo_spmv(int n, const int *rowptr, const int *col, const double *val, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) s += val[k] * x[col[k]];
    y[i] = s;
  }
  return 0;
}

static double vdot(int n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
static double vnorm(int n, const double *a) { return sqrt(vdot(n, a, a)); }

typedef struct {
  int type, n;
  const int *rowptr, *col;
  const double *val;
  /* ILU(0): factored values in A's pattern, inverse pivots */
  double *lu, *idiag; int *dslot;
  /* bjacobi: per-cell tridiagonal factors in block order */
  const vo_problem *prob;
  double *tl, *tinvb, *tc; /* lower multiplier, inverse pivot, upper */
} pc_ctx;

/* ILU(0), natural ordering, IKJ on the sorted CSR rows (PETSc MatILUFactorSymbolic/Numeric_SeqAIJ, levels=0) */
static int pc_setup_ilu0(pc_ctx *pc) {
  int n = pc->n;
  const int *rp = pc->rowptr, *cj = pc->col;
  int nnz = rp[n];
  pc->lu = (double *)malloc(nnz * sizeof(double));
  pc->idiag = (double *)malloc(n * sizeof(double));
  pc->dslot = (int *)malloc(n * sizeof(int));
  memcpy(pc->lu, pc->val, nnz * sizeof(double));
  int *pos = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) pos[i] = -1;
  for (int i = 0; i < n; i++) {
    for (int k = rp[i]; k < rp[i + 1]; k++) pos[cj[k]] = k;
    int d = -1;
    for (int k = rp[i]; k < rp[i + 1]; k++) {
      int kk = cj[k];
      if (kk >= i) { if (kk == i) d = k; break; }
      double mult = pc->lu[k] * pc->idiag[kk];
      pc->lu[k] = mult;
      for (int j = pc->dslot[kk] + 1; j < rp[kk + 1]; j++) {
        int q = pos[cj[j]];
        if (q >= 0) pc->lu[q] -= mult * pc->lu[j];
      }
    }
    for (int k = rp[i]; k < rp[i + 1]; k++) pos[cj[k]] = -1;
    if (d < 0 || pc->lu[d] == 0.0) { free(pos); return -1; }
    pc->dslot[i] = d;
    pc->idiag[i] = 1.0 / pc->lu[d];
  }
  free(pos);
  return 0;
}

static void pc_apply_ilu0(const pc_ctx *pc, const double *r, double *z) {
  int n = pc->n;
  const int *rp = pc->rowptr, *cj = pc->col;
  for (int i = 0; i < n; i++) {
    double s = r[i];
    for (int k = rp[i]; k < pc->dslot[i]; k++) s -= pc->lu[k] * z[cj[k]];
    z[i] = s;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = z[i];
    for (int k = pc->dslot[i] + 1; k < rp[i + 1]; k++) s -= pc->lu[k] * z[cj[k]];
    z[i] = s * pc->idiag[i];
  }
}

static double csr_entry(const pc_ctx *pc, int row, int col) {
  for (int k = pc->rowptr[row]; k < pc->rowptr[row + 1]; k++)
    if (pc->col[k] == col) return pc->val[k];
  return 0.0;
}

/* block-Jacobi with one block per soil column = exact LU of the block's tridiagonal (Thomas) */
static int pc_setup_bjacobi(pc_ctx *pc) {
  const vo_problem *p = pc->prob;
  if (!p || !p->block_ptr) return -1;
  int tot = p->block_ptr[p->nblocks];
  pc->tl = (double *)malloc(tot * sizeof(double));
  pc->tinvb = (double *)malloc(tot * sizeof(double));
  pc->tc = (double *)malloc(tot * sizeof(double));
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (int b = 0; b < p->nblocks; b++) {
    int s = p->block_ptr[b], e = p->block_ptr[b + 1];
    double piv_prev = 0.0;
    for (int k = s; k < e; k++) {
      int cell = p->block_cells[k];
      double a = (k > s) ? csr_entry(pc, cell, p->block_cells[k - 1]) : 0.0;
      double bb = csr_entry(pc, cell, cell);
      double c = (k < e - 1) ? csr_entry(pc, cell, p->block_cells[k + 1]) : 0.0;
      double l = (k > s) ? a * piv_prev : 0.0; /* a / b'_{k-1} */
      double piv = bb - ((k > s) ? l * pc->tc[k - 1] : 0.0);
      if (piv == 0.0) bad = 1;
      pc->tl[k] = l; pc->tc[k] = c;
      pc->tinvb[k] = 1.0 / piv;
      piv_prev = pc->tinvb[k];
    }
  }
  return bad ? -1 : 0;
}

static void pc_apply_bjacobi(const pc_ctx *pc, const double *r, double *z) {
  const vo_problem *p = pc->prob;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < p->nblocks; b++) {
    int s = p->block_ptr[b], e = p->block_ptr[b + 1];
    double prev = 0.0;
    for (int k = s; k < e; k++) {
      int cell = p->block_cells[k];
      double d = r[cell] - pc->tl[k] * prev;
      z[cell] = d;
      prev = d;
    }
    double next = 0.0;
    for (int k = e - 1; k >= s; k--) {
      int cell = p->block_cells[k];
      double x = (z[cell] - pc->tc[k] * next) * pc->tinvb[k];
      z[cell] = x;
      next = x;
    }
  }
}

static int pc_setup(pc_ctx *pc, const vo_problem *p, int type, int n, const int *rowptr, const int *col, const double *val) {
  memset(pc, 0, sizeof *pc);
  pc->type = type; pc->n = n; pc->rowptr = rowptr; pc->col = col; pc->val = val; pc->prob = p;
  switch (type) {
    case VO_PC_NONE: return 0;
    case VO_PC_ILU0: return pc_setup_ilu0(pc);
    case VO_PC_BJACOBI: return pc_setup_bjacobi(pc);
    case VO_PC_JACOBI:
      pc->idiag = (double *)malloc(n * sizeof(double));
      for (int i = 0; i < n; i++) { double d = csr_entry(pc, i, i); if (d == 0.0) return -1; pc->idiag[i] = 1.0 / d; }
      return 0;
    default: return -1;
  }
}
static void pc_apply(const pc_ctx *pc, const double *r, double *z) {
  switch (pc->type) {
    case VO_PC_NONE: memcpy(z, r, pc->n * sizeof(double)); break;
    case VO_PC_ILU0: pc_apply_ilu0(pc, r, z); break;
    case VO_PC_BJACOBI: pc_apply_bjacobi(pc, r, z); break;
    case VO_PC_JACOBI: for (int i = 0; i < pc->n; i++) z[i] = r[i] * pc->idiag[i]; break;
  }
}
static void pc_free(pc_ctx *pc) { free(pc->lu); free(pc->idiag); free(pc->dslot); free(pc->tl); free(pc->tinvb); free(pc->tc); }

int vo_pc_setup_apply(vo_problem *p, int pc_type, const int *rowptr, const int *col, const double *val, const double *r, double *z) {
  pc_ctx pc;
  if (pc_setup(&pc, p, pc_type, p->n, rowptr, col, val)) { pc_free(&pc); FAIL("pc setup failed (zero pivot or no blocks)"); }
  pc_apply(&pc, r, z);
  pc_free(&pc);
  return 0;
}

/* KSPConvergedDefault: reasons 2=RTOL 3=ATOL, -3=ITS -4=DTOL -5=BREAKDOWN */
typedef struct { double rtol, atol, dtol, ttol, rnorm0; } ksp_conv;
static int ksp_converged(ksp_conv *c, int it, double rnorm) {
  if (it == 0) { c->rnorm0 = rnorm; c->ttol = fmax(c->rtol * rnorm, c->atol); }
  if (rnorm != rnorm) return -9; /* KSP_DIVERGED_NANORINF */
  if (rnorm <= c->ttol) return (rnorm < c->atol) ? 3 : 2;
  if (rnorm >= c->dtol * c->rnorm0) return -4;
  return 0;
}

/* t = M^{-1} A v  (left preconditioning, KSP_PCApplyBAorAB) */
static void apply_BA(const pc_ctx *pc, const double *v, double *t, double *work) {
  vo_spmv(pc->n, pc->rowptr, pc->col, pc->val, v, work);
  pc_apply(pc, work, t);
}

/* KSPSolve_BCGS (PETSc src/ksp/ksp/impls/bcgs/bcgs.c), zero initial guess, left PC, preconditioned norm */
static int ksp_bcgs(const pc_ctx *pc, ksp_conv *cv, int max_it, const double *b, double *x, int *its) {
  int n = pc->n, reason = 0;
  double *R = (double *)malloc(n * sizeof(double)), *RP = (double *)malloc(n * sizeof(double));
  double *V = (double *)calloc(n, sizeof(double)), *T = (double *)malloc(n * sizeof(double));
  double *S = (double *)malloc(n * sizeof(double)), *P = (double *)calloc(n, sizeof(double));
  double *W = (double *)malloc(n * sizeof(double));
  memset(x, 0, n * sizeof(double));
  pc_apply(pc, b, R);
  double dp = vnorm(n, R);
  *its = 0;
  reason = ksp_converged(cv, 0, dp);
  if (!reason) {
    memcpy(RP, R, n * sizeof(double));
    double rhoold = 1.0, alpha = 1.0, omegaold = 1.0;
    int i = 0;
    do {
      double rho = vdot(n, R, RP);
      double beta = (rho / rhoold) * (alpha / omegaold);
      for (int k = 0; k < n; k++) P[k] = R[k] - omegaold * beta * V[k] + beta * P[k];
      apply_BA(pc, P, V, W);
      double d1 = vdot(n, V, RP);
      if (d1 == 0.0) { reason = -5; break; }
      alpha = rho / d1;
      for (int k = 0; k < n; k++) S[k] = R[k] - alpha * V[k];
      apply_BA(pc, S, T, W);
      d1 = vdot(n, S, T);
      double d2 = vdot(n, T, T);
      if (d2 == 0.0) {
        double ss = vdot(n, S, S);
        if (ss != 0.0) { reason = -5; break; }
        for (int k = 0; k < n; k++) x[k] += alpha * P[k];
        (*its)++; reason = 2; break;
      }
      double omega = d1 / d2;
      for (int k = 0; k < n; k++) x[k] = alpha * P[k] + omega * S[k] + x[k];
      for (int k = 0; k < n; k++) R[k] = S[k] - omega * T[k];
      dp = vnorm(n, R);
      rhoold = rho; omegaold = omega;
      (*its)++;
      reason = ksp_converged(cv, i + 1, dp);
      if (reason) break;
      if (rho == 0.0) { reason = -5; break; }
      i++;
    } while (i < max_it);
    if (!reason) reason = -3;
  }
  free(R); free(RP); free(V); free(T); free(S); free(P); free(W);
  return reason;
}

/* KSPSolve_GMRES (restart m, classical Gram-Schmidt without refinement, Givens; PETSc gmres.c), left PC */
static int ksp_gmres(const pc_ctx *pc, ksp_conv *cv, int restart, int max_it, const double *b, double *x, int *its) {
  int n = pc->n, m = restart, reason = 0;
  double *VV = (double *)malloc((size_t)(m + 1) * n * sizeof(double));
  double *H = (double *)calloc((size_t)(m + 1) * m, sizeof(double)); /* column-major (m+1) x m */
  double *cs = (double *)malloc(m * sizeof(double)), *sn = (double *)malloc(m * sizeof(double));
  double *rs = (double *)malloc((m + 1) * sizeof(double)), *y = (double *)malloc(m * sizeof(double));
  double *W = (double *)malloc(n * sizeof(double)), *hh = (double *)malloc((m + 1) * sizeof(double));
  memset(x, 0, n * sizeof(double));
  *its = 0;
  int first = 1;
  while (!reason) {
    /* VV(0) = M^{-1}(b - A x) */
    if (first) pc_apply(pc, b, VV);
    else {
      vo_spmv(n, pc->rowptr, pc->col, pc->val, x, W);
      for (int k = 0; k < n; k++) W[k] = b[k] - W[k];
      pc_apply(pc, W, VV);
    }
    double res = vnorm(n, VV);
    if (first) { reason = ksp_converged(cv, 0, res); first = 0; if (reason) break; }
    else { reason = ksp_converged(cv, *its, res); if (reason) break; }
    if (res == 0.0) { reason = 3; break; }
    for (int k = 0; k < n; k++) VV[k] /= res;
    rs[0] = res;
    int it = 0;
    while (it < m && *its < max_it) {
      double *vnew = VV + (size_t)(it + 1) * n;
      apply_BA(pc, VV + (size_t)it * n, vnew, W);
      /* classical Gram-Schmidt: all dots first, then one update */
      for (int j = 0; j <= it; j++) hh[j] = vdot(n, vnew, VV + (size_t)j * n);
      for (int j = 0; j <= it; j++) {
        const double *vj = VV + (size_t)j * n;
        double h = hh[j];
        for (int k = 0; k < n; k++) vnew[k] -= h * vj[k];
      }
      double tt = vnorm(n, vnew);
      int hapend = 0;
      if (tt == 0.0) hapend = 1; else for (int k = 0; k < n; k++) vnew[k] /= tt;
      double *hcol = H + (size_t)it * (m + 1);
      for (int j = 0; j <= it; j++) hcol[j] = hh[j];
      hcol[it + 1] = tt;
      /* apply previous rotations, form new one (KSPGMRESUpdateHessenberg) */
      for (int j = 0; j < it; j++) {
        double t1 = hcol[j], t2 = hcol[j + 1];
        hcol[j] = cs[j] * t1 + sn[j] * t2;
        hcol[j + 1] = -sn[j] * t1 + cs[j] * t2;
      }
      double hyp = sqrt(hcol[it] * hcol[it] + hcol[it + 1] * hcol[it + 1]);
      if (hyp == 0.0) { reason = -5; break; }
      cs[it] = hcol[it] / hyp; sn[it] = hcol[it + 1] / hyp;
      rs[it + 1] = -sn[it] * rs[it];
      rs[it] = cs[it] * rs[it];
      hcol[it] = cs[it] * hcol[it] + sn[it] * hcol[it + 1];
      hcol[it + 1] = 0.0;
      res = fabs(rs[it + 1]);
      it++; (*its)++;
      reason = ksp_converged(cv, *its, res);
      if (reason || hapend) break;
    }
    /* build solution: solve upper-triangular H y = rs, x += V y */
    for (int i = it - 1; i >= 0; i--) {
      double s = rs[i];
      for (int j = i + 1; j < it; j++) s -= H[(size_t)j * (m + 1) + i] * y[j];
      y[i] = s / H[(size_t)i * (m + 1) + i];
    }
    for (int j = 0; j < it; j++) {
      const double *vj = VV + (size_t)j * n;
      double yj = y[j];
      for (int k = 0; k < n; k++) x[k] += yj * vj[k];
    }
    if (!reason && *its >= max_it) reason = -3;
  }
  free(VV); free(H); free(cs); free(sn); free(rs); free(y); free(W); free(hh);
  return reason;
}

static int ksp_solve(vo_problem *p, int ksp_type, int pc_type, double rtol, const int *rowptr, const int *col,
                     const double *val, const double *b, double *x, int *its) {
  pc_ctx pc;
  if (pc_setup(&pc, p, pc_type, p->n, rowptr, col, val)) { pc_free(&pc); *its = 0; return -11; /* KSP_DIVERGED_PC_FAILED */ }
  ksp_conv cv = {rtol, p->ksp_atol, p->ksp_dtol, 0, 0};
  int reason;
  switch (ksp_type) {
    case VO_KSP_PREONLY: pc_apply(&pc, b, x); *its = 1; reason = 4; /* KSP_CONVERGED_ITS */ break;
    case VO_KSP_BCGS: reason = ksp_bcgs(&pc, &cv, p->ksp_max_it, b, x, its); break;
    default: reason = ksp_gmres(&pc, &cv, p->gmres_restart, p->ksp_max_it, b, x, its); break;
  }
  pc_free(&pc);
  return reason;
}

int vo_ksp_solve(vo_problem *p, int ksp_type, int pc_type, double rtol, const int *rowptr, const int *col,
                 const double *val, const double *b, double *x, int *its, int *reason) {
  *reason = ksp_solve(p, ksp_type, pc_type, rtol > 0 ? rtol : p->ksp_rtol, rowptr, col, val, b, x, its);
  return 0;
}

/* ------------------------------------------------------------------ */
/* SNES newtonls + bt line search (PETSc v3.16.2 src/snes/impls/ls/ls.c, linesearch/impls/bt) */
/* ------------------------------------------------------------------ */
static int snes_converged_default(const vo_problem *p, int it, double xnorm, double snorm, double fnorm,
                                  double *ttol, double *rnorm0) {
  int reason = 0;
  if (!it) { *ttol = fnorm * p->snes_rtol; *rnorm0 = fnorm; }
  if (fnorm != fnorm) return -4;                              /* DIVERGED_FNORM_NAN */
  if (fnorm < p->snes_atol && it) return 2;                   /* CONVERGED_FNORM_ABS (forceiteration false: also at it 0) */
  if (fnorm < p->snes_atol) return 2;
  if (p->nfuncs >= p->snes_max_funcs) return -2;              /* DIVERGED_FUNCTION_COUNT */
  if (it) {
    if (fnorm <= *ttol) reason = 3;                           /* CONVERGED_FNORM_RELATIVE */
    else if (snorm < p->snes_stol * xnorm) reason = 4;        /* CONVERGED_SNORM_RELATIVE */
    else if (p->snes_divtol > 0 && fnorm > p->snes_divtol * (*rnorm0)) reason = -9; /* DIVERGED_DTOL */
  }
  return reason;
}

/* SNESLineSearchApply_BT, cubic order, alpha=1e-4, maxstep=1e8, steptol(minlambda)=1e-12, max_it=40.
 * X,F updated in place; Y is the Newton direction (X_new = X - lambda Y).  returns 0 ok, 1 failed */
static int linesearch_bt(vo_problem *p, double *X, double *F, double *Y, double *W, double *G, double *fnorm_io,
                         double *xnorm_o, double *ynorm_o) {
  const double alpha = 1e-4, maxstep = 1e8, minlambda = 1e-12;
  const int max_its = 40;
  int n = p->n;
  double fnorm = *fnorm_io, gnorm = 0.0, lambda = 1.0, lambdatemp, lambdaprev, gprev, g;
  double ynorm = vnorm(n, Y), xnorm = vnorm(n, X);
  if (ynorm == 0.0) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
  if (ynorm > maxstep) { double sc = maxstep / ynorm; for (int i = 0; i < n; i++) Y[i] *= sc; ynorm = maxstep; }
  double f = fnorm * fnorm;
  /* initslope = F . (J Y) */
  vo_spmv(n, p->rowptr, p->col, p->val, Y, W);
  double initslope = vdot(n, F, W);
  if (initslope > 0.0) initslope = -initslope;
  if (initslope == 0.0) initslope = -1.0;
  for (;;) {
    for (int i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
    if (p->nfuncs >= p->snes_max_funcs) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
    p->residual_fn(p, W, G);
    gnorm = vnorm(n, G);
    g = gnorm * gnorm;
    if (isfinite(g)) break;
    if (lambda <= minlambda) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
    lambda = .5 * lambda;
  }
  if (.5 * g <= .5 * f + lambda * alpha * initslope) {
    /* full step accepted */
  } else {
    if (p->snes_stol * xnorm > ynorm) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 0; } /* tiny step: quit, X,F unchanged */
    lambdatemp = -initslope / (g - f - 2.0 * lambda * initslope);
    lambdaprev = lambda; gprev = g;
    if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
    if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
    for (int i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
    if (p->nfuncs >= p->snes_max_funcs) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
    p->residual_fn(p, W, G);
    gnorm = vnorm(n, G); g = gnorm * gnorm;
    if (!isfinite(g)) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
    if (.5 * g < .5 * f + lambda * alpha * initslope) {
      /* quadratically determined step */
    } else {
      int count;
      for (count = 0; count < max_its; count++) {
        if (lambda <= minlambda) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
        double t1 = .5 * (g - f) - lambda * initslope;
        double t2 = .5 * (gprev - f) - lambdaprev * initslope;
        double a = (t1 / (lambda * lambda) - t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
        double b = (-lambdaprev * t1 / (lambda * lambda) + lambda * t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
        double d = b * b - 3 * a * initslope;
        if (d < 0.0) d = 0.0;
        if (a == 0.0) lambdatemp = -initslope / (2.0 * b); else lambdatemp = (-b + sqrt(d)) / (3.0 * a);
        lambdaprev = lambda; gprev = g;
        if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
        if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
        for (int i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
        if (p->nfuncs >= p->snes_max_funcs) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
        p->residual_fn(p, W, G);
        gnorm = vnorm(n, G); g = gnorm * gnorm;
        if (!isfinite(g)) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
        if (.5 * g < .5 * f + lambda * alpha * initslope) break;
      }
      if (count == max_its) { *xnorm_o = xnorm; *ynorm_o = ynorm; return 1; }
    }
  }
  memcpy(X, W, n * sizeof(double));
  memcpy(F, G, n * sizeof(double));
  *xnorm_o = vnorm(n, X);
  *ynorm_o = ynorm; /* norm of the unscaled direction, as SNESLineSearchSetNorms receives it in v3.16 */
  *fnorm_io = gnorm;
  return 0;
}

/* The other line searches SOEBaseStepDT_SNES can select (SystemOfEquationsBaseType.F90:432-475): PETSc v3.16.2
 * src/snes/linesearch/impls/{basic,l2,cp}; third-party, restated from the published algorithms.  Defaults of
 * SNESLineSearchCreate: damping (initial lambda) 1, steptol 1e-12, maxstep 1e8, rtol 1e-8, atol 1e-15, max_its 1 (l2, cp),
 * cp order linear.  All of them always move X (no sufficient-decrease test); l2 and cp report FAILED_REDUCT when the step
 * length ends at or below steptol.  No reference fixture exercises them: "parity unpinned by the reference".
 * Return value: 0 ok, 1 failed (the caller then applies newtonls' stol / maxFailures logic). */
static void ls_axpy(int n, double *W, const double *X, double lambda, const double *Y) {
  for (int i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];   /* VecCopy(X,W); VecAXPY(W,-lambda,Y) */
}
static void ls_finish(vo_problem *p, double *X, double *F, const double *Y, const double *W, double *fnorm_io,
                      double *xnorm_o, double *ynorm_o) {
  /* VecCopy(W,X); snesfunc(X,F); SNESLineSearchComputeNorms: ||X||, ||F||, ||Y|| (Y unscaled) */
  int n = p->n;
  memcpy(X, W, n * sizeof(double));
  p->residual_fn(p, X, F);
  *xnorm_o = vnorm(n, X); *fnorm_io = vnorm(n, F); *ynorm_o = vnorm(n, Y);
  p->ls_ynorm = *ynorm_o;
}
/* SNESLineSearchApply_Basic: X <- X - lambda Y with the damping lambda = 1 */
static int linesearch_basic(vo_problem *p, double *X, double *F, double *Y, double *W, double *G, double *fnorm_io,
                            double *xnorm_o, double *ynorm_o) {
  int n = p->n;
  ls_axpy(n, W, X, 1.0, Y);
  p->residual_fn(p, W, G);
  *fnorm_io = vnorm(n, G); *ynorm_o = vnorm(n, Y); *xnorm_o = vnorm(n, W);
  p->ls_ynorm = *ynorm_o;
  memcpy(X, W, n * sizeof(double));
  memcpy(F, G, n * sizeof(double));
  return 0;
}
/* SNESLineSearchApply_L2: secant iteration on ||F(X - lambda Y)||^2 through the points lambda_old, lambda_mid, lambda */
static int linesearch_l2(vo_problem *p, double *X, double *F, double *Y, double *W, double *G, double *fnorm_io,
                         double *xnorm_o, double *ynorm_o) {
  (void)G;
  const double steptol = 1e-12;
  double maxstep = 1e8;
  const int max_its = 1;
  int n = p->n;
  double lambda = 1.0, lambda_old = 0.0, lambda_mid, lambda_update, delLambda;
  double fnrm = 0.0, fnrm_old = (*fnorm_io) * (*fnorm_io), fnrm_mid = 0.0, delFnrm, delFnrm_old, del2Fnrm;
  lambda_mid = 0.5 * (lambda + lambda_old);
  for (int i = 0; i < max_its; i++) {
    for (;;) {
      ls_axpy(n, W, X, lambda_mid, Y);
      p->residual_fn(p, W, F);
      fnrm_mid = vnorm(n, F);
      ls_axpy(n, W, X, lambda, Y);
      p->residual_fn(p, W, F);
      fnrm = vnorm(n, F);
      fnrm_mid = fnrm_mid * fnrm_mid;
      fnrm = fnrm * fnrm;
      if (isfinite(fnrm)) break;
      if (lambda <= steptol) {   /* FAILED_REDUCT; X unchanged, F holds the last trial's residual as in PETSc */
        *xnorm_o = vnorm(n, X); *ynorm_o = vnorm(n, Y);
        return 1;
      }
      maxstep = .95 * lambda;
      lambda = .5 * (lambda + lambda_old);
      lambda_mid = .5 * (lambda + lambda_old);
    }
    delLambda = lambda - lambda_old;
    delFnrm = (3. * fnrm - 4. * fnrm_mid + 1. * fnrm_old) / delLambda;
    delFnrm_old = (-3. * fnrm_old + 4. * fnrm_mid - 1. * fnrm) / delLambda;
    del2Fnrm = (delFnrm - delFnrm_old) / delLambda;
    if (del2Fnrm > 0.) lambda_update = lambda - delFnrm / del2Fnrm;
    else if (del2Fnrm < 0.) lambda_update = lambda + delFnrm / del2Fnrm;
    else break;
    if (lambda_update < steptol) lambda_update = 0.5 * (lambda + lambda_old);
    if (!isfinite(lambda_update)) break;
    if (lambda_update > maxstep) break;
    lambda_old = lambda;
    lambda = lambda_update;
    fnrm_old = fnrm;
    lambda_mid = 0.5 * (lambda + lambda_old);
  }
  ls_axpy(n, W, X, lambda, Y);
  ls_finish(p, X, F, Y, W, fnorm_io, xnorm_o, ynorm_o);
  return lambda <= steptol ? 1 : 0;
}
/* SNESLineSearchApply_CP: secant iteration on F(X - lambda Y) . Y (order linear); *snes_reason receives
 * CONVERGED_FNORM_ABS (2) when |F.Y| < atol * ynorm at entry (ynorm: the previous application's) */
static int linesearch_cp(vo_problem *p, double *X, double *F, double *Y, double *W, double *G, double *fnorm_io,
                         double *xnorm_o, double *ynorm_o, int *snes_reason) {
  (void)G;
  const double steptol = 1e-12, maxstep = 1e8, rtol = 1e-8, atol = 1e-15;
  const int max_its = 1;
  int n = p->n;
  double lambda = 1.0, lambda_old = 0.0, lambda_update, delLambda, fty, fty_init, fty_old, s;
  const double ynorm_in = p->ls_ynorm;
  fty_old = vdot(n, F, Y);
  if (fabs(fty_old) < atol * ynorm_in) { *snes_reason = 2; *xnorm_o = vnorm(n, X); *ynorm_o = ynorm_in; return 0; }
  fty_init = fty_old;
  for (int i = 0; i < max_its; i++) {
    ls_axpy(n, W, X, lambda, Y);
    p->residual_fn(p, W, F);
    fty = vdot(n, F, Y);
    delLambda = lambda - lambda_old;
    if (fabs(delLambda) < steptol * lambda) break;
    if (fabs(fty) / fabs(fty_init) < rtol) break;
    if (fabs(fty) < atol * ynorm_in && i > 0) break;
    s = (fty - fty_old) / delLambda;
    if (s > 0.) s = -s;
    if (s == 0.0) break;
    lambda_update = lambda - fty / s;
    if (lambda_update < steptol) lambda_update = lambda + fty / s;
    if (!isfinite(lambda_update)) break;
    if (lambda_update > maxstep) break;
    lambda_old = lambda;
    lambda = lambda_update;
    fty_old = fty;
  }
  ls_axpy(n, W, X, lambda, Y);
  ls_finish(p, X, F, Y, W, fnorm_io, xnorm_o, ynorm_o);
  return lambda <= steptol ? 1 : 0;
}

/* SNESSolve_NEWTONLS; returns SNESConvergedReason */
static int snes_solve(vo_problem *p, double *X, int *newton_its, int *linear_its) {
  int n = p->n, reason = 0;
  double *F = (double *)malloc(n * sizeof(double)), *Y = (double *)malloc(n * sizeof(double));
  double *W = (double *)malloc(n * sizeof(double)), *G = (double *)malloc(n * sizeof(double));
  double ttol = 0, rnorm0 = 0, xnorm = 0, ynorm = 0;
  p->nfuncs = 0;
  *newton_its = 0; *linear_its = 0;
  p->residual_fn(p, X, F);
  double fnorm = vnorm(n, F);
  reason = snes_converged_default(p, 0, 0.0, 0.0, fnorm, &ttol, &rnorm0);
  int i;
  for (i = 0; !reason && i < p->snes_max_it; i++) {
    p->jacobian_fn(p); /* auxvars are those of the last residual evaluation, at X */
    int its;
    int kreason = ksp_solve(p, p->ksp_type, p->pc_type, p->ksp_rtol, p->rowptr, p->col, p->val, F, Y, &its);
    *linear_its += its;
    if (kreason < 0) { reason = -3; break; } /* DIVERGED_LINEAR_SOLVE (maxLinearSolveFailures = 1) */
    int lsfail, ls_reason = 0;
    switch (p->linesearch_type) {
      case 1: lsfail = linesearch_basic(p, X, F, Y, W, G, &fnorm, &xnorm, &ynorm); break;
      case 3: lsfail = linesearch_l2(p, X, F, Y, W, G, &fnorm, &xnorm, &ynorm); break;
      case 4: lsfail = linesearch_cp(p, X, F, Y, W, G, &fnorm, &xnorm, &ynorm, &ls_reason); break;
      default: lsfail = linesearch_bt(p, X, F, Y, W, G, &fnorm, &xnorm, &ynorm); p->ls_ynorm = ynorm; break;
    }
    if (ls_reason) { reason = ls_reason; break; }              /* if (snes->reason) break; */
    if (!isfinite(fnorm)) { reason = -4; break; }              /* SNESCheckFunctionNorm */
    if (lsfail) {
      if (p->snes_stol * xnorm > ynorm) { reason = 4; break; }
      reason = -6; break; /* DIVERGED_LINE_SEARCH (maxFailures = 1) */
    }
    *newton_its = i + 1;
    reason = snes_converged_default(p, i + 1, xnorm, ynorm, fnorm, &ttol, &rnorm0);
  }
  if (!reason) reason = -5; /* DIVERGED_MAX_IT */
  free(F); free(Y); free(W); free(G);
  return reason;
}

/* SOEBaseStepDT_SNES (SystemOfEquationsBaseType.F90:368-552) with VSFMSOEPreSolve/PostSolve, including the list of line
 * searches tried before a time-step cut when solver%use_dynamic_linesearch is set (:432-453, 462-475, 486-506, 539-550) */
int vo_step_dt(vo_problem *p, double dt, int nstep, int *converged, int *converged_reason) {
  (void)nstep;
  if (!p->rowptr) FAIL("connections not set");
  const int max_num_time_cuts = 20;
  int num_time_cuts = 0, linesearch_iter = 0;
  double target_time = dt, dt_iter = dt;
  const int ls_default = p->linesearch_type;
  int types[3], max_linesearch_iter;
  if (ls_default == 2 || ls_default == 3) {   /* bt or l2: the other one is the alternative */
    max_linesearch_iter = 2;
    types[0] = ls_default; types[1] = ls_default == 2 ? 3 : 2; types[2] = 0;
  } else {                                    /* basic or cp: then l2, then bt */
    max_linesearch_iter = 3;
    types[0] = ls_default; types[1] = 3; types[2] = 2;
  }
  p->time = 0.0;
  *converged = 0;
  for (;;) {
    vo_set_dtime(p, dt_iter);
    vo_pre_solve(p);
    p->linesearch_type = types[linesearch_iter];
    int nit, lit;
    int reason = snes_solve(p, p->soln, &nit, &lit);
    *converged_reason = reason;
    if (reason < 0) {
      linesearch_iter++;
      if (p->use_dynamic_linesearch && linesearch_iter < max_linesearch_iter) {
        p->cum_linesearch_switches++;         /* try another line search at the same dt */
      } else {
        linesearch_iter = 0;
        num_time_cuts++; p->cum_cuts++;
        dt_iter = 0.5 * dt_iter;
      }
      memcpy(p->soln, p->soln_prev, p->n * sizeof(double));
    } else {
      *converged = 1;
      p->time += dt_iter;
      p->cum_newton += nit; p->cum_linear += lit;
      memcpy(p->soln_prev, p->soln, p->n * sizeof(double)); /* VSFMSOEPostSolve :600 */
      /* SetDataInSOEAuxVar in PostSolve (GE:1150-1166, 1338-1342): exchanged mass from the fluxes of the last residual */
      for (int c = 0; c < p->nc; c++)
        if (p->horizontal[c]) {
          p->lat_mass_exc[p->up[c]] = p->lat_mass_exc[p->up[c]] + p->internal_flux[c] * p->dtime;
          p->lat_mass_exc[p->dn[c]] = p->lat_mass_exc[p->dn[c]] - p->internal_flux[c] * p->dtime;
        }
      for (int b = 0; b < p->nbc; b++)
        for (int i = 0; i < p->bc[b].n; i++) p->bc[b].mass_exc[i] = p->bc[b].mass_exc[i] + p->bc[b].flux[i] * p->dtime;
    }
    if (num_time_cuts > max_num_time_cuts) {
      /* The SoE auxvars GetDataForCLM reads are filled in PostSolve only (VSFMSOEPostSolve :566-660), so after a step that
       * gave up they still hold the state of the last sub-step that converged = soln (restored above); the auxvars and
       * fluxes of this restatement, left at the diverged iterate by the last residual, are brought back to that state */
      double *scratch = (double *)malloc(p->n * sizeof(double));
      vo_residual(p, p->soln, scratch);
      p->nfuncs--; p->cum_funcs--;   /* (not one of SNES's function evaluations) */
      free(scratch);
      *converged = 0; p->linesearch_type = ls_default; return 0;
    }
    if (p->time >= target_time) break;
  }
  p->linesearch_type = ls_default;   /* "Set the linsearch type to be the default setting" :539-550 */
  return 0;
}
/* -snes_linesearch_type basic|bt|l2|cp (1..4) and solver%use_dynamic_linesearch */
int vo_set_linesearch(vo_problem *p, int type, int use_dynamic) {
  if (type < 1 || type > 4) FAIL("Unknown default linesearch option");
  p->linesearch_type = type; p->use_dynamic_linesearch = use_dynamic != 0;
  return 0;
}
long long vo_linesearch_switches(const vo_problem *p) { return p->cum_linesearch_switches; }

/* VSFMSOEGetDataForCLM after RichardsODEPressureSetDataInSOEAuxVar (GE:1170-1195); auxvars as of the last residual */
int vo_get(vo_problem *p, int auxvar_type, int var_type, int cond_id, int n, double *out) {
  if (auxvar_type == VO_AUXVAR_INTERNAL) {
    if (n > p->n) FAIL("size(data_1d) > nauxvar");
    for (int i = 0; i < n; i++) {
      const auxvar *a = &p->aux[i];
      switch (var_type) {
        case VO_VAR_PRESSURE: out[i] = a->pressure; break;
        /* inactive cells: nothing is copied for them (GE:1172), the SoE auxvar keeps the zeros of VSFMSOEAuxVarInit */
        case VO_VAR_LIQ_SAT: out[i] = p->active[i] ? a->sat : 0.0; break;
        case VO_VAR_MASS: out[i] = p->active[i] ? a->por * a->den * FMWH2O * a->sat * p->vol[i] : 0.0; break;
        case VO_VAR_SOIL_MATRIX_POT: out[i] = p->active[i] ? (a->pressure - PRESSURE_REF) / (a->den * FMWH2O * GRAVITY_CONSTANT) : 0.0; break;
        case VO_VAR_LATERAL_MASS_EXCHANGED: out[i] = p->lat_mass_exc[i]; break;
        case VO_VAR_FRAC_LIQ_SAT: out[i] = a->frac_liq_sat; break;
        default: FAIL("unknown var_type %d", var_type);
      }
    }
    return 0;
  }
  if (auxvar_type == VO_AUXVAR_BC || auxvar_type == VO_AUXVAR_SS) {
    int nb = auxvar_type == VO_AUXVAR_BC ? p->nbc : p->nss;
    if (cond_id < 1 || cond_id > nb) FAIL("bad condition id");
    condition *c = auxvar_type == VO_AUXVAR_BC ? &p->bc[cond_id - 1] : &p->ss[cond_id - 1];
    if (n > c->n) FAIL("size(data_1d) > nauxvar");
    if (var_type == VO_VAR_BC_MASS_EXCHANGED && auxvar_type == VO_AUXVAR_BC) { for (int i = 0; i < n; i++) out[i] = c->mass_exc[i]; return 0; }
    if (var_type != VO_VAR_MASS_FLUX) FAIL("unknown var_type %d", var_type);
    for (int i = 0; i < n; i++) out[i] = c->flux[i];
    return 0;
  }
  if (auxvar_type == VO_AUXVAR_CONN_INTERNAL) {
    if (n > p->nc) FAIL("size(data_1d) > nauxvar");
    if (var_type != VO_VAR_MASS_FLUX) FAIL("unknown var_type %d", var_type);
    for (int i = 0; i < n; i++) out[i] = p->internal_flux[i];
    return 0;
  }
  FAIL("unknown auxvar_type %d", auxvar_type);
}

void vo_get_counters(vo_problem *p, long *newton, long *linear, long *time_cuts, long *func_evals) {
  if (newton) *newton = p->cum_newton;
  if (linear) *linear = p->cum_linear;
  if (time_cuts) *time_cuts = p->cum_cuts;
  if (func_evals) *func_evals = p->cum_funcs;
}

/* ------------------------------------------------------------------ */
/* internals exported to th_oracle.c (vsfm_oracle_internal.h)           */
/* ------------------------------------------------------------------ */
int vo__satparams_set(satparams *s, int type, double alpha, double lambda, double sat_res) { return satparams_set(s, type, alpha, lambda, sat_res); }
void vo__press_to_sat(const satparams *s, double press, double *sat, double *dsat_dP) { press_to_sat(s, press, sat, dsat_dP); }
void vo__press_to_relperm(const satparams *s, double press, double frac_liq, double *kr, double *dkr_dP) { press_to_relperm(s, press, frac_liq, kr, dkr_dP); }
void vo__auxvar_init(auxvar *a) { auxvar_init(a); }
void vo__auxvar_compute(auxvar *a, int density_type) { auxvar_compute(a, density_type); }
void vo__richards_flux_internal(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv, int internal_conn,
                                int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn) {
  richards_flux_internal(up, dn, g, compute_deriv, internal_conn, cond_type, flux, dflux_dP_up, dflux_dP_dn);
}
void vo__richards_flux(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv, int internal_conn,
                       int swap_order, int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn) {
  richards_flux(up, dn, g, compute_deriv, internal_conn, swap_order, cond_type, flux, dflux_dP_up, dflux_dP_dn);
}
int vo__density_ifc67(double t_C, double p, double *dwmol, double *dwp, double *dwt) { return density_ifc67(t_C, p, dwmol, dwp, dwt); }
int vo__snes_solve(vo_problem *p, double *X, int *newton_its, int *linear_its) { return snes_solve(p, X, newton_its, linear_its); }
void vo__set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
