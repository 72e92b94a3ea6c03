/*
 * vsfm_oracle_internal.h -- CPU ORACLE internals shared between vsfm_oracle.c and th_oracle.c
 * (test infrastructure only; see vsfm_oracle.h).
 */
#ifndef VSFM_ORACLE_INTERNAL_H
#define VSFM_ORACLE_INTERNAL_H
#include "vsfm_oracle.h"

/* MultiPhysicsProbConstants.F90:199-202 */
#define PRESSURE_REF 101325.0
#define GRAVITY_CONSTANT 9.80665
#define FMWH2O 18.01534
/* mpp_varcon.F90:17 */
#define DENH2O 1.0e3

/* saturation function parameters: src/mpp/util/SaturationFunction.F90:31-48 */
typedef struct {
  int sat_func_type;  /* VG / BC / SBC (bz2,bz3 are stored as SBC: SaturationFunction.F90:291,348) */
  double sat_res, alpha, vg_m, vg_n, bc_lambda, sbc_pu, sbc_ps, sbc_b2, sbc_b3;
} satparams;

/* src/mpp/auxvar/RichardsODEPressureAuxType.F90 */
typedef struct {
  /* state + parameters (RichardsODEPressureAuxType.F90:18-60) */
  double pressure, temperature, frac_liq_sat, condition_value;
  double perm[3];
  /* porosity_params_type (PorosityFunctionMod.F90:18-29): POROSITY_CONSTANT (1) or POROSITY_LINEAR (2) */
  int por_model;
  double por_base, por_pressure_reference, por_lin_mod_slope;
  satparams sp;
  /* derived */
  double vis, kr, sat, den, por;
  double dpor_dP, dvis_dP, dkr_dP, dsat_dP, dden_dP, dden_dT;
} auxvar;

typedef struct {
  double area, dist_up, dist_dn, uvec[3];
} conn_geom;

typedef struct {
  int kind, cond_type, n, swap_order;
  int *id_dn;      /* 0-based */
  conn_geom *geom;
  auxvar *aux;     /* BC auxvars (aux_vars_bc) */
  double *value;   /* SS condition values (cur_cond%value) */
  double *sink_pressure, *sink_exponent; /* aux_vars_ss%pot_mass_sink_pressure / _exponent (down-regulated sinks) */
  double *flux;    /* boundary_flux / ss_flux [kg/s] */
  double *mass_exc; /* bnd_mass_exc [kg] (GoveqnRichardsODEPressureType.F90:34, 1338) */
} condition;

struct vo_problem {
  int n;
  double *vol;
  int *active;
  auxvar *aux;          /* aux_vars_in */
  int density_type;
  int nc;
  int *up, *dn;         /* 0-based */
  conn_geom *geom;
  double *internal_flux;
  int *horizontal;      /* conn%GetType() == CONN_HORIZONTAL */
  double *lat_mass_exc; /* [kg] (GE:33, 1159-1163) */
  double *soln_prev_clm;
  int nlat;             /* mesh%lateral_conn_set_list (operator split, GE:2615-2743) */
  int *lat_up, *lat_dn;
  conn_geom *lat_geom;
  int nbc, nss;
  condition *bc, *ss;
  /* CSR pattern (sorted columns) + assembly slot tables */
  int *rowptr, *col, *diag_slot;
  int *slot_uu, *slot_ud, *slot_du, *slot_dd; /* per internal connection */
  double *val;
  /* vectors */
  double *soln, *soln_prev, *accum_prev;
  double dtime, time;
  /* block-Jacobi blocks */
  int nblocks, *block_ptr, *block_cells;
  /* solver options (PETSc defaults) */
  int ksp_type, pc_type;
  double ksp_rtol, ksp_atol, ksp_dtol; int ksp_max_it, gmres_restart;
  double snes_rtol, snes_atol, snes_stol, snes_divtol; int snes_max_it, snes_max_funcs;
  /* counters (SolverType.F90:40-41,140-169) */
  long cum_newton, cum_linear, cum_cuts, cum_funcs;
  int nfuncs;
  /* SNESLineSearch: 1 basic, 2 bt (PETSc's default for newtonls), 3 l2, 4 cp; solver%use_dynamic_linesearch (SolverType.F90:43) */
  int linesearch_type, use_dynamic_linesearch;
  double ls_ynorm;   /* linesearch->ynorm as the previous application left it (cp reads it before computing its own) */
  long long cum_linesearch_switches;
  /* the nonlinear system SNES solves: VSFM by default; th_oracle.c installs the coupled TH system */
  int (*residual_fn)(vo_problem *p, const double *x, double *f);
  void (*jacobian_fn)(vo_problem *p); /* at the state of the last residual evaluation, into rowptr/col/val */
  void *user;
};

/* internals exported to th_oracle.c */
int vo__satparams_set(satparams *s, int type, double alpha, double lambda, double sat_res);
void vo__press_to_sat(const satparams *s, double press, double *sat, double *dsat_dP);
void vo__press_to_relperm(const satparams *s, double press, double frac_liq, double *kr, double *dkr_dP);
void vo__auxvar_init(auxvar *a);
void vo__auxvar_compute(auxvar *a, int density_type);
void vo__richards_flux_internal(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv,
                                int internal_conn, int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn);
void vo__richards_flux(const auxvar *up, const auxvar *dn, const conn_geom *g, int compute_deriv, int internal_conn,
                       int swap_order, int cond_type, double *flux, double *dflux_dP_up, double *dflux_dP_dn);
int vo__density_ifc67(double t_C, double p, double *dwmol, double *dwp, double *dwt);
int vo__snes_solve(vo_problem *p, double *X, int *newton_its, int *linear_its);
void vo__set_error(const char *msg);
#endif
