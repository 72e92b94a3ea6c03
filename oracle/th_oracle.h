/*
 * th_oracle.h -- CPU ORACLE (test infrastructure only) for the coupled thermal-hydrology system
 * (SURVEY.md section 8 f-2).  See th_oracle.c for the reference lines each function restates.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 */
#ifndef TH_ORACLE_H
#define TH_ORACLE_H
#include "vsfm_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { VO_COND_HEAT_FLUX = 507, VO_COND_HEAT_RATE = 511 };   /* MultiPhysicsProbConstants.F90 */
enum { THO_INT_ENERGY_ENTHALPY_CONSTANT = 1, THO_INT_ENERGY_ENTHALPY_IFC67 = 2 }; /* EOSWaterMod.F90:23-24 */

typedef struct tho_problem tho_problem;

void tho_internal_energy_enthalpy(double P, double t_K, int itype, double den, double dden_dT, double dden_dP, double *out6);
void tho_auxvar_compute(int sat_func_type, double sat_res, double alpha, double lambda, double por, int density_type,
                        int ie_type, double tcond_wet, double tcond_dry, double talpha, double pressure, double temperature,
                        double *out15);

tho_problem *tho_create(int ncells);
void tho_destroy(tho_problem *t);
/* the mass equation (mesh, internal connections, its conditions and soils are set on it with the vo_* calls; the heat
 * equation shares its mesh and internal connections) and the host of the 2N global system (solver options, counters) */
vo_problem *tho_mass(tho_problem *t);
vo_problem *tho_global(tho_problem *t);
int tho_set_eos(tho_problem *t, int density_type, int int_energy_enthalpy_type);
int tho_energy_add_condition(tho_problem *t, int kind, int cond_type, int nconn, const int *id_dn, const double *dist_up,
                             const double *dist_dn, const double *area, const double *unit_vec, int *cond_id);
int tho_energy_set_porosity_linear(tho_problem *t, const double *press_ref, const double *slope);
int tho_energy_set_soils(tho_problem *t, const double *por, const int *satfunc_type, const double *alpha, const double *lambda,
                         const double *sat_res, const double *perm_x, const double *perm_y, const double *perm_z,
                         const double *heat_cap, const double *tcond_wet, const double *tcond_dry, const double *talpha,
                         const double *den_soil);
int tho_energy_set_condition_values(tho_problem *t, int kind, int cond_id, int n, const double *values);
int tho_energy_set_bc_pressure(tho_problem *t, int cond_id, int n, const double *values);
int tho_mass_set_bc_temperature(tho_problem *t, int cond_id, int n, const double *values);
int tho_restart(tho_problem *t, const double *pressure, const double *temperature);
int tho_setup(tho_problem *t);
int tho_set_dtime(tho_problem *t, double dt);
int tho_pre_solve(tho_problem *t);
int tho_residual(tho_problem *t, const double *x, double *f);                 /* x, f: [P(1..N); T(1..N)] */
int tho_jacobian_nnz(tho_problem *t);
int tho_jacobian(tho_problem *t, const double *x, int *rowptr, int *col, double *val); /* 2N x 2N CSR, 0-based */
int tho_step_dt(tho_problem *t, double dt, int *converged, int *converged_reason);
int tho_get_solution(tho_problem *t, double *pressure, double *temperature);

#ifdef __cplusplus
}
#endif
#endif
