/*
 * vsfm_oracle.h -- CPU ORACLE for the VSFM Newton-Krylov hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; the shipped library (mpp_amd/csrc) never does.
 *
 * It is a plain-C restatement (written from the reference's published algorithm, no source
 * copied) of MPP's variably-saturated-flow model:
 *   constitutive laws   src/mpp/util/{SaturationFunction,EOSWaterMod,PorosityFunctionMod}.F90
 *   two-point flux      src/mpp/ge/RichardsMod.F90:118-340
 *   residual/Jacobian   src/mpp/ge/GoveqnRichardsODEPressureType.F90:388-453,1603-2200
 *   time step driver    src/mpp/soe/SystemOfEquationsBaseType.F90:368-552
 * and of the PETSc v3.16.2 pieces the reference delegates to (third-party, not under
 * /root/reference; pinned by README.md:35 and .ci-scripts/install-petsc.sh:5):
 *   SNES newtonls + bt line search, KSP GMRES(30)/BiCGStab, PC ILU(0)/block-Jacobi.
 *
 * Pinning: the reference cannot be built here (needs PETSc headers + library, absent), so the
 * oracle is pinned by the reference's own fixtures: the three EOS known-answer tests
 * (src/tests/test_eos_*.F90:19-25) and the end-to-end regression baselines
 * regression_tests/vsfm/vsfm_{celia1990,vchannel}.regression.baseline -- see tests/.
 * The Krylov/PC pieces have no fixture in the reference ("parity unpinned by the reference");
 * they are cross-checked against dense solves in tests/.
 *
 * Cell and connection ids at this API are 1-based, exactly as the Fortran callers hold them.
 */
#ifndef VSFM_ORACLE_H
#define VSFM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* integer ids: MultiPhysicsProbConstants.F90:82-94,102-109,141,172-177 */
enum {
  VO_COND_NULL = 500, VO_COND_BC = 501, VO_COND_SS = 502, VO_COND_MASS_RATE = 503,
  VO_COND_MASS_FLUX = 504, VO_COND_DIRICHLET = 505, VO_COND_DIRICHLET_FRM_OTR_GOVEQ = 506,
  VO_COND_SEEPAGE_BC = 509, VO_COND_DOWNREG_MASS_RATE_CAMPBELL = 512, VO_COND_DOWNREG_MASS_RATE_FETCH2 = 513
};
enum { VO_VAR_FRAC_LIQ_SAT = 612, VO_VAR_LATERAL_MASS_EXCHANGED = 613, VO_VAR_BC_MASS_EXCHANGED = 614 };
enum { VO_VAR_POT_MASS_SINK_PRESSURE = 638, VO_VAR_POT_MASS_SINK_EXPONENT = 639 };
enum { VO_VAR_PRESSURE = 604, VO_VAR_LIQ_SAT = 608, VO_VAR_MASS = 610,
       VO_VAR_SOIL_MATRIX_POT = 611, VO_VAR_MASS_FLUX = 644 };
enum { VO_AUXVAR_INTERNAL = 701, VO_AUXVAR_BC = 702, VO_AUXVAR_SS = 703, VO_AUXVAR_CONN_INTERNAL = 704 };
/* SaturationFunction.F90:19-28 */
enum { VO_SAT_FUNC_VAN_GENUCHTEN = 1301, VO_SAT_FUNC_BROOKS_COREY = 1302,
       VO_SAT_FUNC_SMOOTHED_BROOKS_COREY = 1303, VO_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2 = 1304,
       VO_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3 = 1305 };
/* EOSWaterMod.F90:19-21 */
enum { VO_DENSITY_CONSTANT = 1, VO_DENSITY_TGDPB01 = 2, VO_DENSITY_IFC67 = 3 };
/* solver selections (PETSc option names) */
enum { VO_KSP_PREONLY = 0, VO_KSP_GMRES = 1, VO_KSP_BCGS = 2 };
enum { VO_PC_NONE = 0, VO_PC_ILU0 = 1, VO_PC_BJACOBI = 2, VO_PC_JACOBI = 3 };

typedef struct vo_problem vo_problem;

/* ---- scalar constitutive functions (for unit pins) ---- */
void vo_density(double p, double t_K, int density_type, double *den, double *dden_dp, double *dden_dT);
void vo_viscosity(double p, double t_K, double *vis, double *dvis_dp, double *dvis_dT);
/* one cell's AuxVarCompute with given soil parameters; out[8] = sat,dsat_dP,kr,dkr_dP,den,dden_dP,por,dpor_dP */
void vo_auxvar_compute(int sat_func_type, double sat_res, double alpha, double lambda_or_m,
                       double por_base, int density_type, double temperature, double frac_liq_sat,
                       double pressure, double *out8);
/* RichardsFlux (RichardsMod.F90:29-114): aux[5] = P,kr,dkr_dP,den,dden_dP ; perm[3]; returns flux and the two "derivatives" */
void vo_richards_flux(const double *aux_up, const double *perm_up, const double *aux_dn, const double *perm_dn,
                      double area, double dist_up, double dist_dn, const double *unitvec,
                      int internal_conn, int swap_order, int cond_type,
                      double *flux, double *dflux_dP_up, double *dflux_dP_dn);

/* ---- problem set-up ---- */
vo_problem *vo_create(int ncells);
void vo_destroy(vo_problem *p);
const char *vo_last_error(void);
int vo_set_mesh(vo_problem *p, const double *vol, const int *is_active /* may be NULL = all active */);
int vo_set_connections(vo_problem *p, int nconn, const int *id_up, const int *id_dn,
                       const double *dist_up, const double *dist_dn, const double *area,
                       const double *unitvec /* nconn x 3, row-major */);
/* returns 1-based condition id within its kind (the reference's soe_auxvar_id), <0 on error */
int vo_add_condition(vo_problem *p, int kind /* COND_BC|COND_SS */, int cond_type, int nconn, const int *id_dn,
                     const double *dist_up, const double *dist_dn, const double *area, const double *unitvec,
                     int swap_order);
int vo_set_soil_permeability(vo_problem *p, const double *perm_x, const double *perm_y, const double *perm_z);
int vo_set_soil_porosity(vo_problem *p, const double *por);
int vo_set_soil_porosity_linear(vo_problem *p, const double *por_base, const double *press_ref, const double *slope);
int vo_set_saturation_function(vo_problem *p, const int *satfunc_type, const double *alpha,
                               const double *lambda, const double *sat_res);
int vo_set_density_type(vo_problem *p, int density_type);
int vo_set_condition_values(vo_problem *p, int kind, int cond_id, int n, const double *values);
/* lateral connection set of the mesh + VSFMComputeLateralFlux (SystemOfEquationsVSFMType.F90:848-889,
 * GoveqnRichardsODEPressureType.F90:2615-2743): SetDtime, PreSolve, then the Darcy flux of every lateral connection at
 * soln_prev is accumulated [kg/s] into the mass-rate source-sink condition `ss_cond_id` (the reference finds it by its name
 * 'Lateral_flux'; it must list every cell once, in cell order) */
int vo_set_lateral_connections(vo_problem *p, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                               const double *dist_dn, const double *area, const double *unitvec);
int vo_compute_lateral_flux(vo_problem *p, double dt, int ss_cond_id);
int vo_get_condition_values(vo_problem *p, int kind, int cond_id, int n, double *values);
/* VSFMMPPSetSourceSinkAuxVarRealValue (MultiPhysicsProbVSFM.F90:1437-1520): values run over the concatenation of all
 * source-sink connections of the equation */
int vo_set_source_sink_auxvar(vo_problem *p, int var_type, int n, const double *values);
int vo_restart(vo_problem *p, const double *pressure);
int vo_set_gmres_restart(vo_problem *p, int restart);
/* soe%PreStepDT / PostStepDT of ELM's calling sequence (SystemOfEquationsVSFMType.F90:892-940; GE:2773-2785): soln_prev_clm ->
 * soln_prev, soln and reset of the exchanged-mass accumulators; soln_prev -> soln_prev_clm */
int vo_pre_step_dt(vo_problem *p);
int vo_post_step_dt(vo_problem *p);
/* per-cell frac_liq_sat (SetDataFromCLM(AUXVAR_INTERNAL, VAR_FRAC_LIQ_SAT), copied into the GE auxvars in PreSolve, GE:536-582) */
int vo_set_frac_liq_sat(vo_problem *p, const double *frac);
/* column blocks for PC bjacobi: block b holds cells block_cells[block_ptr[b]..block_ptr[b+1]) (1-based ids, bottom->top) */
int vo_set_blocks(vo_problem *p, int nblocks, const int *block_ptr, const int *block_cells);
/* -snes_linesearch_type: 1 basic, 2 bt (default), 3 l2, 4 cp; use_dynamic: solver%use_dynamic_linesearch */
int vo_set_linesearch(vo_problem *p, int type, int use_dynamic);
long long vo_linesearch_switches(const vo_problem *p);
int vo_set_solver(vo_problem *p, int ksp_type, int pc_type, double ksp_rtol, double snes_rtol,
                  double snes_atol, double snes_stol, int snes_max_it);
void vo_set_num_threads(int n);

/* ---- operator level (GoverningEquation API) ---- */
int vo_set_dtime(vo_problem *p, double dt);
int vo_pre_solve(vo_problem *p);                                   /* accum_prev from soln_prev */
int vo_residual(vo_problem *p, const double *x, double *f);
int vo_jacobian_nnz(vo_problem *p);
int vo_jacobian(vo_problem *p, const double *x, int *rowptr, int *col, double *val); /* 0-based CSR, sorted cols */
int vo_spmv(int n, const int *rowptr, const int *col, const double *val, const double *x, double *y);
int vo_pc_setup_apply(vo_problem *p, int pc_type, const int *rowptr, const int *col, const double *val,
                      const double *r, double *z);
int vo_ksp_solve(vo_problem *p, int ksp_type, int pc_type, double rtol, const int *rowptr, const int *col,
                 const double *val, const double *b, double *x, int *its, int *reason);

/* ---- time step (SystemOfEquations API) ---- */
int vo_step_dt(vo_problem *p, double dt, int nstep, int *converged, int *converged_reason);
int vo_get(vo_problem *p, int auxvar_type, int var_type, int cond_id, int n, double *out);
void vo_get_counters(vo_problem *p, long *newton, long *linear, long *time_cuts, long *func_evals);

#ifdef __cplusplus
}
#endif
#endif
