/*
 * mppgpu.h -- C ABI of the MI355X-native VSFM Newton-Krylov solve path.
 *
 * Drop-in boundary for MPP's variably-saturated-flow model: a Fortran (iso_c_binding), C, or
 * ctypes host keeps the reference's SystemOfEquations / GoverningEquation call sequence and
 * binds these entry points instead of the PETSc-backed implementations.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error (PETSc `ierr` convention,
 *     src/mpp/soe/SystemOfEquationsBaseType.F90:334-365); the message is available from
 *     mppgpu_last_error().  Nothing aborts the process.
 *   - cell / connection / condition ids are 1-based exactly as the Fortran callers hold them
 *     (natural ordering of the caller's mesh, any numbering).
 *   - all array arguments are caller-owned host memory, copied during the call; the library
 *     retains no host pointer.  Device memory is owned by the handle.
 *   - one handle = one GPU + one HIP stream; calls on a handle must be serialised by the caller.
 *   - the library computes on the GPU only: there is no CPU fallback.  mppgpu_create fails when
 *     no HIP device is usable.
 */
#ifndef MPPGPU_H
#define MPPGPU_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mppgpu_handle_s *mppgpu_handle;

/* ---- integer ids shared with the reference (MultiPhysicsProbConstants.F90:82-94,102-109,141,172-177) ---- */
#define MPPGPU_COND_BC 501
#define MPPGPU_COND_SS 502
#define MPPGPU_COND_MASS_RATE 503
#define MPPGPU_COND_MASS_FLUX 504
#define MPPGPU_COND_DIRICHLET 505
#define MPPGPU_COND_HEAT_FLUX 507
#define MPPGPU_COND_SEEPAGE_BC 509
#define MPPGPU_COND_HEAT_RATE 511
#define MPPGPU_COND_DOWNREG_MASS_RATE_CAMPBELL 512
#define MPPGPU_COND_DOWNREG_MASS_RATE_FETCH2 513
#define MPPGPU_VAR_PRESSURE 604
#define MPPGPU_VAR_TEMPERATURE 605
#define MPPGPU_VAR_BC_SS_CONDITION 607
#define MPPGPU_VAR_LIQ_SAT 608
#define MPPGPU_VAR_MASS 610
#define MPPGPU_VAR_SOIL_MATRIX_POT 611
#define MPPGPU_VAR_FRAC_LIQ_SAT 612
#define MPPGPU_VAR_LATERAL_MASS_EXCHANGED 613
#define MPPGPU_VAR_BC_MASS_EXCHANGED 614
#define MPPGPU_VAR_LIQ_AREAL_DEN 615
#define MPPGPU_VAR_ICE_AREAL_DEN 617
#define MPPGPU_VAR_FRAC 618
#define MPPGPU_VAR_SNOW_WATER 619
#define MPPGPU_VAR_NUM_SNOW_LYR 620
#define MPPGPU_VAR_DHS_DT 621
#define MPPGPU_VAR_ACTIVE 624
#define MPPGPU_VAR_TUNING_FACTOR 630
#define MPPGPU_VAR_POT_MASS_SINK_PRESSURE 638
#define MPPGPU_VAR_POT_MASS_SINK_EXPONENT 639
#define MPPGPU_VAR_MASS_FLUX 644
#define MPPGPU_AUXVAR_INTERNAL 701
#define MPPGPU_AUXVAR_BC 702
#define MPPGPU_AUXVAR_SS 703
#define MPPGPU_AUXVAR_CONN_INTERNAL 704
/* EOSWaterMod.F90:23-24 */
#define MPPGPU_INT_ENERGY_ENTHALPY_CONSTANT 1
#define MPPGPU_INT_ENERGY_ENTHALPY_IFC67 2
/* SaturationFunction.F90:19-23 */
#define MPPGPU_SAT_FUNC_VAN_GENUCHTEN 1301
#define MPPGPU_SAT_FUNC_BROOKS_COREY 1302
#define MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2 1304
#define MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3 1305
/* EOSWaterMod.F90:19-21 */
#define MPPGPU_DENSITY_CONSTANT 1
#define MPPGPU_DENSITY_TGDPB01 2
#define MPPGPU_DENSITY_IFC67 3

/* ------------------------------------------------------------------------------------------------
 * life cycle -- replaces vsfm_mpp%Init / SetName / SetID(MPP_VSFM_SNES_CLM) / SetMPIRank
 * (src/driver/standalone/vsfm/vsfm_celia1990_problem.F90:140-168)
 * ---------------------------------------------------------------------------------------------- */
int mppgpu_create(int device_id, mppgpu_handle *out);
int mppgpu_destroy(mppgpu_handle h);
const char *mppgpu_last_error(mppgpu_handle h); /* h may be NULL: error of the last failed create */
int mppgpu_synchronize(mppgpu_handle h);

/* ------------------------------------------------------------------------------------------------
 * problem set-up
 * ---------------------------------------------------------------------------------------------- */
/* mpp%MeshSetDimensions / MeshSetGridCellFilter / MeshComputeVolume
 * (src/mpp/mpp/MultiPhysicsProbBaseType.F90:214-349; src/mpp/dtypes/MeshType.F90:1244-1264).
 * is_active may be NULL (= all cells active). */
int mppgpu_set_mesh(mppgpu_handle h, int ncells, int nlev, const double *vol, const int *is_active);

/* mpp%CreateAndAddConnectionSet(imesh, CONN_SET_INTERNAL, nconn, id_up, id_dn, dist_up, dist_dn, area, itype, unit_vec)
 * (src/mpp/mpp/MultiPhysicsProbBaseType.F90:352-377).  unit_vec is nconn x 3 row-major (up -> dn). */
int mppgpu_set_connections(mppgpu_handle h, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                           const double *dist_dn, const double *area, const double *unit_vec);

/* soe%AddConditionInGovEqn(igoveqn, COND_BC|COND_SS, name, unit, cond_type, conn_set)
 * (src/mpp/soe/SystemOfEquationsBaseType.F90:1260-1290).  For COND_SS only id_dn is read.
 * *cond_id receives the 1-based id within its kind (the reference's soe_auxvar_id). */
int mppgpu_add_condition(mppgpu_handle h, int kind, int cond_type, int nconn, const int *id_dn, const double *dist_up,
                         const double *dist_dn, const double *area, const double *unit_vec, int swap_order,
                         int *cond_id);

/* VSFMMPPSetSoilPermeability / SetSoilPorosity / SetSaturationFunction / SetDensityType
 * (src/mpp/mpp/MultiPhysicsProbVSFM.F90:1115-1182, 1295-1361, 1364-1434, 1247-1292).
 * Boundary-condition auxvars take their soil parameters from their id_dn cell (:422-471). */
int mppgpu_set_soil_permeability(mppgpu_handle h, const double *perm_x, const double *perm_y, const double *perm_z);
int mppgpu_set_soil_porosity(mppgpu_handle h, const double *por);
/* PorosityFunctionSetLinearModel for every cell (src/mpp/util/PorosityFunctionMod.F90:75-96, evaluated :144-163):
 * por = por_base + (P - pressure_reference) * slope, dpor_dP = slope (enters RichardsODEPressureAccumDeriv,
 * src/mpp/ge/GoveqnRichardsODEPressureType.F90:1675-1677).  No MPP-level routine sets it in the reference (the
 * porParams of an auxvar are reachable through the auxvar only); mppgpu_set_soil_porosity returns to the constant model. */
int mppgpu_set_soil_porosity_linear(mppgpu_handle h, const double *por_base, const double *pressure_reference,
                                    const double *slope);
/* satfunc_type is per cell (satParams of every auxvar, src/mpp/auxvar/RichardsODEPressureAuxType.F90:18-67; dispatch
 * src/mpp/util/SaturationFunction.F90:564-650): one mesh may mix van Genuchten, Brooks-Corey and smoothed Brooks-Corey cells */
int mppgpu_set_saturation_function(mppgpu_handle h, const int *satfunc_type, const double *alpha, const double *lambda,
                                   const double *sat_res);
int mppgpu_set_density_type(mppgpu_handle h, int density_type);

/* soe%PreStepDT / soe%PostStepDT of ELM's calling sequence (src/mpp/soe/SystemOfEquationsVSFMType.F90:892-940):
 * PreStepDT copies soln_prev_clm into soln_prev and soln and resets the exchanged-mass accumulators
 * (GoveqnRichardsODEPressureType.F90:2773-2785); PostStepDT copies soln_prev into soln_prev_clm.  mppgpu_restart also
 * fills soln_prev_clm.  With option "track_mass_exchange" = 1 every successful sub-step of mppgpu_step_dt adds flux*dt of
 * the horizontal internal connections and of the boundary connections to the accumulators read back with
 * mppgpu_get_data(AUXVAR_INTERNAL, VAR_LATERAL_MASS_EXCHANGED) / (AUXVAR_BC, VAR_BC_MASS_EXCHANGED, id) (GE:1150-1166,
 * 1338-1342).  mppgpu_set_data(AUXVAR_INTERNAL, VAR_FRAC_LIQ_SAT, -1, ...) sets the per-cell frac_liq_sat the reference
 * copies into the auxvars in PreSolve (GE:536-582). */
int mppgpu_pre_step_dt(mppgpu_handle h);
int mppgpu_post_step_dt(mppgpu_handle h);

/* 3-D meshes split across ranks ("one-plane halo exchange per neighbour per SpMV").  mppgpu_set_ghost_cells:
 * the LAST nghost cells of the mesh (mesh%ncells_ghost in the reference, src/mpp/dtypes/MeshType.F90:42) are copies of
 * cells that another rank owns; connections may join an owned and a ghost cell (never two ghosts); conditions sit on
 * owned cells.  mppgpu_set_halo: for neighbour k the values of the owned cells send_cells[send_ptr[k] .. send_ptr[k+1])
 * (1-based) go to rank peer_rank[k], and recv_count[k] values arrive from it into the next ghost cells -- ghosts are
 * grouped by owner in the order of that owner's send list.  The library refreshes ghost values before every residual /
 * Jacobian / SpMV (RCCL grouped send/recv on the solver's stream, or device copies inside an in-process group) and applies
 * the preconditioner rank-locally (PETSc's parallel default: block-Jacobi with ILU(0) per rank).  Vectors passed through
 * the ABI keep their ghost entries (ignored on input, unspecified on output).  Call both before mppgpu_setup, and
 * mppgpu_comm_init(_local) before the first solve.  The coupled thermal-hydrology system ([P; T] vectors) takes the same
 * two calls: both fields of a ghost cell are refreshed. */
int mppgpu_set_ghost_cells(mppgpu_handle h, int nghost);
int mppgpu_set_halo(mppgpu_handle h, int nneigh, const int *peer_rank, const int *send_ptr, const int *send_cells,
                    const int *recv_count);

/* Operator split of lateral flow (ELM mode).  mppgpu_set_lateral_connections: the mesh's lateral_conn_set_list
 * (src/mpp/dtypes/MeshType.F90:53, 635) -- connections that are NOT part of the implicit system.
 * mppgpu_compute_lateral_flux = soe%ComputeLateralFlux(dt) (src/mpp/soe/SystemOfEquationsVSFMType.F90:848-889 ->
 * RichardsODEComputeLateralFlux, src/mpp/ge/GoveqnRichardsODEPressureType.F90:2615-2743): SetDtime, PreSolve, then the
 * Darcy flux of every lateral connection at soln_prev is accumulated [kg/s] (+ at id_up, - at id_dn) into the
 * COND_MASS_RATE source-sink `ss_cond_id` (the reference locates it by its name 'Lateral_flux'), which must list every
 * cell once in cell order.  mppgpu_get_condition_values reads a condition's current values back
 * (SetDataInSOEAuxVar(AUXVAR_SS, ...) in the reference). */
int mppgpu_set_lateral_connections(mppgpu_handle h, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                                   const double *dist_dn, const double *area, const double *unit_vec);
int mppgpu_compute_lateral_flux(mppgpu_handle h, double dt, int ss_cond_id);
int mppgpu_get_condition_values(mppgpu_handle h, int kind, int cond_id, int n, double *values);

/* VSFMMPPSetSourceSinkAuxVarRealValue (src/mpp/mpp/MultiPhysicsProbVSFM.F90:1437-1520): var_type
 * VAR_POT_MASS_SINK_PRESSURE | VAR_POT_MASS_SINK_EXPONENT of the down-regulated potential sinks
 * (COND_DOWNREG_MASS_RATE_CAMPBELL: rate = value / (1 + (dP/Pc)^n); _FETCH2: value * exp(-(dP/Pc)^n), dP = P - P_ref <= 0;
 * GoveqnRichardsODEPressureType.F90:1900-1927, 2158-2188).  `values` runs over the concatenation of all source-sink
 * connections of the equation, in the order the conditions were added. */
int mppgpu_set_source_sink_auxvar(mppgpu_handle h, int var_type, int n, const double *values);

/* solver selection: the PETSc options database the reference reads through SNESSetFromOptions
 * (src/mpp/mpp/MultiPhysicsProbBaseType.F90:1205).  Defaults as PETSc's: GMRES(30) + ILU(0).  Keys: "ksp_type" (bcgs|gmres|preonly),
 * "pc_type" (bjacobi|ilu|jacobi|none; ilu = PETSc's default ILU(0) in the caller's natural ordering, the default here too since
 * round 5 -- on vertical-only grids it coincides with the exact per-column LU and is served by the same kernels),
 * "ksp_rtol", "ksp_atol", "ksp_dtol", "ksp_max_it", "snes_rtol", "snes_atol", "snes_stol", "snes_max_it",
 * "profile" (0 | 1: HIP-event timing of every kernel class | 2: of the CSR SpMV launches and the SpMV fused with the column solve only),
 * "problem_type" (vsfm|thermal|th),
 * "cnfac" (thermal).
 * Tuning keys (results unchanged): "spmv_variant", "assembly" (auto|general|columns), "nseg", "fused_assembly"
 * (0|1), "fused_trials" (line-search trial points whose Jacobian is assembled together with the residual),
 * "fused_predict", "fused_rows" (0|1: column march that folds the boundary fluxes in and also leaves ||F||, ||X|| and
 * the column LU), "fused_ls_dots" (0|1: line-search reductions inside the SpMV), "ilu_graph", "ilu_chain" (0 one launch
 * per level | 1 chained tiles where they pay | 2 always chained), "ilu_box" (ILU(0) of a structured grid in natural ordering
 * factorised and solved box by box: 0 never | 1 when the pattern is such a grid | 2 or fail), "ilu_box_len" (box extent along the
 * fastest direction), "ilu_box_factor" (0: factorise by the general elimination program, solve by boxes), "ilu_box_stream"
 * (1: the box solves as two streams of pencils per application, faces handed over inside the launch; 0: one launch per box
 * level), "ilu_box_workers", "ilu_box_fused_io", "ilu_box_queue", "th_spmv_cols" / "th_stage" (coupled system on soil columns:
 * index-free SpMV, matrix rows staged through LDS), "ksp_gmres_restart" (1..4096, default 30), "ell_assembly", "defer_test0",
 * "ls_ahead", "wide_xr", "track_mass_exchange".
 * Set-up keys (before mppgpu_setup): "setup_device" (1: column detection, numberings, incidence lists and the CSR pattern are
 * built by kernels; 0: on the host's cores), "setup_verify" (1: both, every array compared -- an error if they differ).
 * Environment: MPPGPU_HOST_THREADS (threads of the set-up passes on the host; default: the cores the process may run on, at
 * most 32), MPPGPU_TIMING (set: wall-clock of the set-up phases on stderr). */
int mppgpu_set_option(mppgpu_handle h, const char *key, const char *value);

/* mpp%AllocateAuxVars + mpp%SetupProblem (src/mpp/mpp/MultiPhysicsProbVSFM.F90:710-1015): builds the CSR
 * pattern and the device layout.  Must follow mesh/connections/conditions and precede everything below. */
int mppgpu_setup(mppgpu_handle h);

/* mpp%Restart(data_1d) (src/mpp/mpp/MultiPhysicsProbVSFM.F90:603-707) */
int mppgpu_restart(mppgpu_handle h, int n, const double *pressure);

/* ------------------------------------------------------------------------------------------------
 * time stepping -- the SystemOfEquations API
 * ---------------------------------------------------------------------------------------------- */
/* soe%SetDataFromCLM(soe_auxvar_type, var_type, soe_auxvar_id, data_1d)
 * (src/mpp/soe/SystemOfEquationsVSFMType.F90:663-724); AUXVAR_BC / AUXVAR_SS with VAR_BC_SS_CONDITION. */
int mppgpu_set_data(mppgpu_handle h, int auxvar_type, int var_type, int cond_id, int n, const double *data);

/* soe%StepDT(dt, nstep, converged, converged_reason, ierr) (src/mpp/soe/SystemOfEquationsBaseType.F90:334-552):
 * Newton (newtonls + cubic backtracking), linear solve, dt halving (<= 20 cuts).  converged_reason uses
 * PETSc's SNESConvergedReason numbering (>0 converged, <0 diverged). */
int mppgpu_step_dt(mppgpu_handle h, double dt, int nstep, int *converged, int *converged_reason);

/* soe%GetDataForCLM(soe_auxvar_type, var_type, soe_auxvar_id, data_1d) (SystemOfEquationsVSFMType.F90:781-845):
 * AUXVAR_INTERNAL: VAR_PRESSURE, VAR_LIQ_SAT, VAR_MASS, VAR_SOIL_MATRIX_POT;
 * AUXVAR_BC / AUXVAR_SS / AUXVAR_CONN_INTERNAL: VAR_MASS_FLUX. */
int mppgpu_get_data(mppgpu_handle h, int auxvar_type, int var_type, int cond_id, int n, double *data);

/* The sub-steps the most recent mppgpu_step_dt accepted, in order: {dt} without a time-step cut, else the halved steps the
 * loop of SOEBaseStepDT_SNES went on with (src/mpp/soe/SystemOfEquationsBaseType.F90:454-536: after a cut the loop keeps
 * the halved dt_iter until time >= target, so the steps may add up to more than dt).  *n = their number (all of them,
 * also when max_n is smaller); the reference only logs the cuts (write(iulog,*) ... 'cutting dt to'). */
int mppgpu_get_step_history(mppgpu_handle h, int max_n, double *dt_sub, int *n);
/* solver_type counters (src/mpp/dtypes/SolverType.F90:40-41,140-169) */
int mppgpu_get_counters(mppgpu_handle h, long long *newton_its, long long *linear_its, long long *time_cuts,
                        long long *function_evals);

/* ------------------------------------------------------------------------------------------------
 * operator level -- the GoverningEquation API, for PETSc-hosted runs and parity tests
 * ---------------------------------------------------------------------------------------------- */
/* soe%SetDtime + soe%PreSolve (SystemOfEquationsVSFMType.F90:506-563; GoveqnRichardsODEPressureType.F90:2747-2768) */
int mppgpu_set_dtime(mppgpu_handle h, double dt);
int mppgpu_pre_solve(mppgpu_handle h);
/* goveqn%ComputeResidual(X, F, ierr) behind SOEResidual (GoveqnRichardsODEPressureType.F90:388-421) */
int mppgpu_residual(mppgpu_handle h, const double *x, double *f);
/* goveqn%ComputeJacobian(X, A, B, ierr) behind SOEJacobian (:425-453).  CSR, 0-based, sorted columns, caller ordering. */
int mppgpu_jacobian_nnz(mppgpu_handle h, int *nnz);
int mppgpu_jacobian(mppgpu_handle h, const double *x, int *rowptr, int *col, double *val);
/* PETSc MatMult / PCApply / KSPSolve on the Jacobian most recently assembled by mppgpu_jacobian or mppgpu_step_dt */
int mppgpu_spmv(mppgpu_handle h, const double *x, double *y);
int mppgpu_pc_apply(mppgpu_handle h, const double *r, double *z);
int mppgpu_ksp_solve(mppgpu_handle h, const double *b, double *x, int *its, int *reason);

/* ------------------------------------------------------------------------------------------------
 * coupled thermal-hydrology: sysofeqns_th_type with goveqn_richards_ode_pressure_type (equation 1) and
 * goveqn_thermal_enthalpy_soil_type (equation 2) on one mesh, coupled through their internal auxvars
 * (src/mpp/soe/SystemOfEquationsTHType.F90:736-1003; src/mpp/ge/GoveqnThermalEnthalpySoilType.F90:1060-2376;
 * src/mpp/ge/GoveqnRichardsODEPressureType.F90:2473-2612).  Select with mppgpu_set_option(h, "problem_type", "th")
 * before conditions are added.  Unknowns, residuals and the Jacobian are field-blocked [P(1..N); T(1..N)] (the
 * reference's DMComposite order): mppgpu_restart, mppgpu_residual, mppgpu_jacobian, mppgpu_spmv, mppgpu_pc_apply and
 * mppgpu_ksp_solve take / return 2N values and a 2N x 2N CSR; mppgpu_step_dt solves the coupled system;
 * mppgpu_get_data(AUXVAR_INTERNAL, VAR_PRESSURE | VAR_TEMPERATURE) returns one field.  Mesh, connections, mass-equation
 * soils (mppgpu_set_soil_*, shared by both equations as in MPPTHSetSoils, MultiPhysicsProbTH.F90:75-176) and
 * mppgpu_set_density_type are the VSFM calls; "int_energy_enthalpy_type" (1 constant | 2 IFC67) is an option.
 * Preconditioners: ilu (PETSc's default), jacobi, none.
 * ---------------------------------------------------------------------------------------------- */
/* soe%AddConditionInGovEqn(igoveqn, ...) (SystemOfEquationsBaseType.F90:1260-1290): igoveqn 1 = mass, 2 = heat.
 * Heat equation: COND_DIRICHLET boundaries and COND_HEAT_RATE sources (the reference aborts on any other boundary type,
 * GoveqnThermalEnthalpySoilType.F90:1008-1012).  *cond_id is the SoE-level id (mass-equation conditions are numbered
 * before heat-equation ones, as sysofeqns_th_type does) -- an id returned earlier for a heat-equation condition moves up
 * by one when a mass-equation condition of the same kind is added after it: add the mass equation's conditions first. */
int mppgpu_th_add_condition(mppgpu_handle h, int igoveqn, int kind, int cond_type, int nconn, const int *id_dn,
                            const double *dist_up, const double *dist_dn, const double *area, const double *unit_vec,
                            int swap_order, int *cond_id);
/* heat-equation material: ThermEnthalpySetHeatCapacity / ThermalCondWet / ThermalCondDry / ThermalAlpha /
 * SetSoilDensity / SetSoilPermeability (GoveqnThermalEnthalpySoilType.F90:2454-2700) or the constants MPPTHSetSoils
 * writes (MultiPhysicsProbTH.F90:331-335).  perm_* may be NULL: the auxvar default 8.3913e-12 m^2 stays
 * (ThermalEnthalpySoilAuxType.F90:93), which is what MPPTHSetSoils leaves in place. */
int mppgpu_th_set_soils(mppgpu_handle h, const double *heat_cap, const double *therm_cond_wet, const double *therm_cond_dry,
                        const double *therm_alpha, const double *den_soil, const double *perm_x, const double *perm_y,
                        const double *perm_z);
/* The other field of a boundary auxvar is set with mppgpu_set_data(h, AUXVAR_BC, VAR_TEMPERATURE, id, ...) for a
 * mass-equation boundary (default 298.15 K) and (AUXVAR_BC, VAR_PRESSURE, id, ...) for a heat-equation boundary --
 * the reference drivers assign aux_vars_bc(:)%temperature / %pressure directly (th_mms_problem.F90:846-875). */

/* ------------------------------------------------------------------------------------------------
 * thermal KSP path: soil heat diffusion, linear solve per step -- sysofeqns_thermal_type with one
 * goveqn_thermal_ksp_temp_soil_type.  Select with mppgpu_set_option(h, "problem_type", "thermal") before conditions
 * are added; mesh, connections, conditions (COND_DIRICHLET / COND_HEAT_FLUX boundaries, COND_HEAT_RATE sources),
 * mppgpu_setup and mppgpu_set_data / mppgpu_set_option ("cnfac", "ksp_type", "pc_type" incl. "ilu") are shared.
 * ---------------------------------------------------------------------------------------------- */
/* MPPThermalSetSoils (src/mpp/mpp/MultiPhysicsProbThermal.F90:76-208) as per-cell arrays; itype 1 soil/crop,
 * 2 wetland, 3 land ice; dz = mesh%dz (layer thickness used by ThermKSPTempSoilAuxVarCompute) */
int mppgpu_thermal_set_soils(mppgpu_handle h, const int *itype, const int *is_soil_shallow, const double *watsat,
                             const double *csol, const double *tkmg, const double *tkdry, const double *dz);
/* soe%SetSolnPrevCLM / PreStepDT / StepDT / GetSoln (src/mpp/soe/SystemOfEquationsThermalType.F90:171-230, 391-409;
 * src/mpp/soe/SystemOfEquationsBaseType.F90:555-647).  mppgpu_set_data carries SetRDataFromCLM:
 * AUXVAR_INTERNAL x {VAR_LIQ_AREAL_DEN, VAR_ICE_AREAL_DEN, VAR_SNOW_WATER, VAR_NUM_SNOW_LYR, VAR_TUNING_FACTOR},
 * AUXVAR_BC x {VAR_BC_SS_CONDITION, VAR_FRAC, VAR_DHS_DT, VAR_ACTIVE}, AUXVAR_SS x VAR_BC_SS_CONDITION.
 * VAR_FRAC / VAR_DHS_DT / VAR_ACTIVE stand for ThermKSPTempSoilAuxVarSetRValues on the equation's boundary auxvars; as in the
 * reference (ThermalKSPTemperatureBaseAuxType.F90:56-60) these start inactive with frac = 0: a boundary contributes nothing until
 * the caller sets VAR_ACTIVE (and VAR_FRAC) for it, as thermal_mms_problem.F90:664-668 does.
 * Option "thermal_reference_defects" = 1 reproduces GoveqnThermalKSPTemperatureSoilType.F90:906-908 (stale area in the
 * COND_DIRICHLET right-hand side) and :1215 (dhsdT**area) as written; the default computes what they intend. */
int mppgpu_thermal_set_soln_prev_clm(mppgpu_handle h, int n, const double *temperature);
int mppgpu_thermal_pre_step_dt(mppgpu_handle h);
int mppgpu_thermal_step_dt(mppgpu_handle h, double dt, int *converged, int *linear_its);
int mppgpu_thermal_get_soln(mppgpu_handle h, int n, double *temperature);
/* soe%ComputeRHS / ComputeOperators behind SOEComputeRHS / SOEComputeOperators (SystemOfEquationsThermalType.F90:546-759;
 * src/mpp/ge/GoveqnThermalKSPTemperatureSoilType.F90:646-1229): for PETSc-hosted runs and parity tests */
int mppgpu_thermal_compute_rhs(mppgpu_handle h, double dt, double *b);
int mppgpu_thermal_compute_operators(mppgpu_handle h, double dt, int *rowptr, int *col, double *val);

/* ------------------------------------------------------------------------------------------------
 * measurement
 * ---------------------------------------------------------------------------------------------- */
/* HIP-event statistics per kernel class since the last reset ("profile" option must be 1).
 * names: "spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce".
 * bytes = algorithmic bytes per launch as defined in DESIGN.md. */
int mppgpu_kernel_stats(mppgpu_handle h, const char *name, long long *launches, double *total_ms,
                        double *algorithmic_bytes_per_launch);
/* ... and how the timed launches were sampled: `calls` = launches of the class since the last reset (timed or not), `stride` = every
 * stride-th of them was bracketed by a HIP-event pair (option "profile_stride"): total_ms * calls / launches estimates the class's
 * total.  No counterpart in the reference (PETSc's -log_view counts calls per event the same way). */
int mppgpu_kernel_calls(mppgpu_handle h, const char *name, long long *calls, int *stride);
int mppgpu_reset_stats(mppgpu_handle h);
/* collectives this handle issued since the last mppgpu_reset_stats: all-reduces (RCCL ncclAllReduce or the in-process
 * group) and halo exchanges; how many chained ILU(0) applications had to be redone with per-level launches because a
 * dependency wait exhausted its spin budget (since mppgpu_setup); and how many times the host waited for device scalars
 * (norms, dot products, Krylov recurrences) since the last mppgpu_reset_stats.  No counterpart in the reference
 * (PETSC_COMM_SELF, host-resident Vec). */
int mppgpu_get_comm_stats(mppgpu_handle h, long long *allreduces, long long *halo_exchanges, long long *chain_fallbacks,
                          long long *host_readbacks);
/* the halo plan of a rank's piece of a split mesh (mppgpu_set_ghost_cells / mppgpu_set_halo; the reference's counterpart is the
 * VecScatter of ugrid_type, src/mpp/dtypes/UnstructuredGridType.F90, sized by ncells_ghost = ncols_ghost*nlevgrnd,
 * src/mpp/dtypes/MeshType.F90:362): neighbours = ranks this one exchanges with, cells_sent / cells_received = owned cells copied out /
 * ghost cells filled per exchange and field, bytes_received_per_exchange = 8 * cells_received * fields (fields = 2 for the coupled
 * thermal-hydrology system: P and T of a ghost cell travel), max_cells_from_one_neighbour = the largest receive count -- for a 7-point
 * nx x ny x nz grid split into x-y blocks of whole soil columns this is nz * nx (nz * nx * 8 B per neighbour and SpMV).
 * Zeros without ghost cells.  Any output pointer may be NULL. */
int mppgpu_get_halo_stats(mppgpu_handle h, long long *neighbours, long long *cells_sent, long long *cells_received,
                          long long *bytes_received_per_exchange, long long *max_cells_from_one_neighbour);
/* how the ILU(0) preconditioner (PETSc PCILU, MatSolve_SeqAIJ) applies its triangular solves on this handle, known once the
 * preconditioner has been set up: path = 0 no ILU(0) in use (or not set up yet), 1 one launch per dependency level,
 * 2 one launch per triangle with the tiles of a level chained by flags, 3 one launch per box level of a structured grid;
 * launches = kernel launches per application of the preconditioner; box_extents[0..2] = cells per box in the caller's
 * fastest, second and third direction (path 3; zeros otherwise).  Any output pointer may be NULL. */
int mppgpu_get_pc_info(mppgpu_handle h, int *path, int *launches, int *box_extents);
/* which data layout and kernels the assembly and Krylov passes of this handle use (known after the first time step; no
 * counterpart in the reference): info[0] soil classes in use by the column march (0: per-cell soil arrays are streamed),
 * [1] per-level connection geometry (1 / 0), [2] per-level porosity x volume, [3] complete-row column march (boundary fluxes,
 * norms, column LU and forward elimination inside the assembly pass), [4] column segments per thread-block row, [5] SpMV
 * variant, [6] column solve through LDS, [7] speculative line-search trial points that had to be discarded since
 * mppgpu_setup, [8] the level / box launches of the ILU(0) solves are replayed from hipGraphs (option ilu_graph), [9] graphs
 * instantiated for them so far, [10] set-up structures built on the device that option setup_verify found equal to the host-built
 * ones, [11] the line search's product J Y (with its sums) is formed from the column tridiagonals at their computed places, without an
 * index (soil-column grids), [12] line searches since mppgpu_setup whose F . (J Y) and ||Y|| came out of the kernels of the Krylov solve
 * that produced Y (one-iteration BiCGStab solves on soil-column grids, option ls_fold) instead of a product pass of their own, [13] the
 * assembly of a 3-D mesh of complete soil columns runs as column march + lateral pass (option assembly, round 5).  n = entries of info
 * the caller provides (at most 14 are written). */
int mppgpu_get_assembly_info(mppgpu_handle h, int n, int *info);
/* device-resident micro-benchmark of one kernel class on the current state: `reps` launches, average ms */
int mppgpu_bench_kernel(mppgpu_handle h, const char *name, int variant, int reps, double *avg_ms,
                        double *algorithmic_bytes);

/* ------------------------------------------------------------------------------------------------
 * The constitutive laws evaluated on the HOST, array by array -- for drivers that need them to build their inputs, as the
 * reference's drivers do when they call the EOS and saturation-function modules directly: th_mms_problem.F90:1262-1452
 * forms its manufactured mass and heat sources from Density (src/mpp/util/EOSWaterMod.F90:38-77), InternalEnergyAndEnthalpy
 * (:589-707), SatFunc_PressToSat / SatFunc_PressToRelPerm (src/mpp/util/SaturationFunction.F90:564-650).  No handle, no GPU,
 * not part of any solve; same arithmetic as the device kernels (mpp_amd/csrc/physics.hpp, th_physics.hpp).  Outputs may be
 * NULL.  Units as in the reference: density kmol/m^3, energies J/kmol, temperatures K.  Errors: mppgpu_last_error(NULL).
 * ---------------------------------------------------------------------------------------------- */
int mppgpu_host_density(int density_type, int n, const double *p, const double *t_K, double *den, double *dden_dp, double *dden_dt);
/* den, dden_dt, dden_dp in kg/m^3 (the reference passes den * FMWH2O: ThermalEnthalpySoilAuxType.F90:262-266) */
int mppgpu_host_internal_energy_enthalpy(int ie_type, int n, const double *p, const double *t_K, const double *den,
                                         const double *dden_dt, const double *dden_dp, double *u, double *h, double *du_dt,
                                         double *dh_dt, double *du_dp, double *dh_dp);
/* satfunc_type: MPPGPU_SAT_FUNC_* per entry; lambda = vg_m or bc_lambda */
int mppgpu_host_sat_relperm(int n, const int *satfunc_type, const double *p, const double *sat_res, const double *alpha,
                            const double *lambda, double *sat, double *dsat_dp, double *kr, double *dkr_dp);

/* ------------------------------------------------------------------------------------------------
 * multi-GPU: one handle per rank, columns sharded across ranks, RCCL all-reduce for the global
 * Newton/Krylov norms and dot products (no counterpart in the reference, whose solver objects are
 * all PETSC_COMM_SELF: src/mpp/mpp/MultiPhysicsProbBaseType.F90:1132,1154,1193).
 * The 128-byte unique id is produced on rank 0 and broadcast by the host (MPI_Bcast / torch.distributed).
 * ---------------------------------------------------------------------------------------------- */
/* Loads RCCL and runs the guard every communicator creation runs: the library's ncclGetVersion must be 2.x and of the same
 * major version as the rccl.h the hand-written prototypes and enumerator values were checked against at build time
 * (static_asserts in mpp_amd/csrc/runtime.cpp).  version = major * 10000 + minor * 100 + patch.  Needs no GPU. */
int mppgpu_comm_runtime_version(int *version);
int mppgpu_comm_get_unique_id(void *id128);
int mppgpu_comm_init(mppgpu_handle h, int nranks, int rank, const void *id128);
/* The same global solve across several handles of ONE process: rank r = handles[r], every handle driven by its own host
 * thread (the calls block in the reductions until all ranks have arrived).  Reductions go through host memory in rank
 * order.  For hosts that drive several GPUs from one process, and for exercising the multi-rank path on a single GPU
 * (RCCL accepts one rank per device). */
int mppgpu_comm_init_local(mppgpu_handle *handles, int nranks);

#ifdef __cplusplus
}
#endif
#endif /* MPPGPU_H */
