!> mppgpu_mod -- thin iso_c_binding shim over the C ABI of include/mppgpu.h.
!!
!! It keeps the reference's call shapes so that an MPP driver changes only its `use` lines:
!!   soe%SetDataFromCLM / StepDT / GetDataForCLM   (src/mpp/soe/SystemOfEquationsVSFMType.F90:663-845,
!!                                                  src/mpp/soe/SystemOfEquationsBaseType.F90:334-365)
!!   goveqn%ComputeResidual / ComputeJacobian      (src/mpp/ge/GoveqnRichardsODEPressureType.F90:388-453)
!!   VSFMMPPSetSoil* / Restart                     (src/mpp/mpp/MultiPhysicsProbVSFM.F90:603-707, 1115-1434)
!! Arrays are default-kind integer / real(8), 1-based ids, exactly as the reference holds them.
module mppgpu_mod
  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer(c_int), parameter, public :: COND_BC = 501, COND_SS = 502, COND_MASS_RATE = 503, COND_MASS_FLUX = 504
  integer(c_int), parameter, public :: COND_DIRICHLET = 505, COND_HEAT_FLUX = 507, COND_SEEPAGE_BC = 509, COND_HEAT_RATE = 511
  integer(c_int), parameter, public :: COND_DOWNREG_MASS_RATE_CAMPBELL = 512, COND_DOWNREG_MASS_RATE_FETCH2 = 513
  integer(c_int), parameter, public :: VAR_TEMPERATURE = 605, VAR_POT_MASS_SINK_PRESSURE = 638, VAR_POT_MASS_SINK_EXPONENT = 639
  integer(c_int), parameter, public :: VAR_PRESSURE = 604, VAR_BC_SS_CONDITION = 607, VAR_LIQ_SAT = 608
  integer(c_int), parameter, public :: VAR_MASS = 610, VAR_SOIL_MATRIX_POT = 611, VAR_MASS_FLUX = 644
  integer(c_int), parameter, public :: AUXVAR_INTERNAL = 701, AUXVAR_BC = 702, AUXVAR_SS = 703, AUXVAR_CONN_INTERNAL = 704
  integer(c_int), parameter, public :: SAT_FUNC_VAN_GENUCHTEN = 1301, SAT_FUNC_BROOKS_COREY = 1302
  integer(c_int), parameter, public :: DENSITY_CONSTANT = 1, DENSITY_TGDPB01 = 2, DENSITY_IFC67 = 3

  ! the raw bindings are public too: one per entry point of include/mppgpu.h
  public :: mppgpu_create, mppgpu_destroy, mppgpu_last_error, mppgpu_set_mesh, mppgpu_set_connections, &
        mppgpu_add_condition, mppgpu_set_soil_permeability, mppgpu_set_soil_porosity, mppgpu_set_soil_porosity_linear, &
        mppgpu_set_saturation_function, &
        mppgpu_set_density_type, mppgpu_set_option, mppgpu_setup, mppgpu_restart, mppgpu_set_data, mppgpu_step_dt, &
        mppgpu_get_data, mppgpu_get_counters, mppgpu_get_step_history, mppgpu_set_dtime, mppgpu_pre_solve, mppgpu_residual, &
        mppgpu_jacobian_nnz, mppgpu_jacobian, mppgpu_synchronize, mppgpu_spmv, mppgpu_pc_apply, mppgpu_ksp_solve, &
        mppgpu_set_lateral_connections, mppgpu_compute_lateral_flux, mppgpu_get_condition_values, &
        mppgpu_set_source_sink_auxvar, mppgpu_thermal_set_soils, mppgpu_thermal_set_soln_prev_clm, &
        mppgpu_thermal_pre_step_dt, mppgpu_thermal_step_dt, mppgpu_thermal_get_soln, mppgpu_thermal_compute_rhs, &
        mppgpu_thermal_compute_operators, mppgpu_th_add_condition, mppgpu_th_set_soils, mppgpu_kernel_stats, mppgpu_kernel_calls, &
        mppgpu_reset_stats, mppgpu_get_comm_stats, mppgpu_get_halo_stats, mppgpu_get_pc_info, mppgpu_get_assembly_info, mppgpu_bench_kernel, &
        mppgpu_comm_get_unique_id, mppgpu_comm_runtime_version, mppgpu_comm_init, &
        mppgpu_comm_init_local, mppgpu_set_ghost_cells, mppgpu_set_halo, &
        mppgpu_pre_step_dt, mppgpu_post_step_dt, &
        mppgpu_host_density, mppgpu_host_internal_energy_enthalpy, mppgpu_host_sat_relperm

  interface
     integer(c_int) function mppgpu_create(device_id, h) bind(C, name='mppgpu_create')
       import :: c_int, c_ptr
       integer(c_int), value :: device_id
       type(c_ptr) :: h
     end function
     integer(c_int) function mppgpu_destroy(h) bind(C, name='mppgpu_destroy')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     type(c_ptr) function mppgpu_last_error(h) bind(C, name='mppgpu_last_error')
       import :: c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_set_mesh(h, ncells, nlev, vol, is_active) bind(C, name='mppgpu_set_mesh')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: ncells, nlev
       real(c_double) :: vol(*)
       integer(c_int) :: is_active(*)
     end function
     integer(c_int) function mppgpu_set_connections(h, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec) &
          bind(C, name='mppgpu_set_connections')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nconn
       integer(c_int) :: id_up(*), id_dn(*)
       real(c_double) :: dist_up(*), dist_dn(*), area(*), unit_vec(*)
     end function
     integer(c_int) function mppgpu_add_condition(h, kind, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, &
          swap_order, cond_id) bind(C, name='mppgpu_add_condition')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: kind, cond_type, nconn, swap_order
       integer(c_int) :: id_dn(*)
       real(c_double) :: dist_up(*), dist_dn(*), area(*), unit_vec(*)
       integer(c_int) :: cond_id
     end function
     integer(c_int) function mppgpu_set_soil_permeability(h, px, py, pz) bind(C, name='mppgpu_set_soil_permeability')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: px(*), py(*), pz(*)
     end function
     integer(c_int) function mppgpu_set_soil_porosity(h, por) bind(C, name='mppgpu_set_soil_porosity')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: por(*)
     end function
     integer(c_int) function mppgpu_set_soil_porosity_linear(h, por_base, pressure_reference, slope) &
          bind(C, name='mppgpu_set_soil_porosity_linear')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: por_base(*), pressure_reference(*), slope(*)
     end function
     integer(c_int) function mppgpu_set_saturation_function(h, satfunc_type, alpha, lambda, sat_res) &
          bind(C, name='mppgpu_set_saturation_function')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int) :: satfunc_type(*)
       real(c_double) :: alpha(*), lambda(*), sat_res(*)
     end function
     integer(c_int) function mppgpu_set_density_type(h, density_type) bind(C, name='mppgpu_set_density_type')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int), value :: density_type
     end function
     integer(c_int) function mppgpu_set_option(h, key, val) bind(C, name='mppgpu_set_option')
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: h
       character(kind=c_char) :: key(*), val(*)
     end function
     integer(c_int) function mppgpu_setup(h) bind(C, name='mppgpu_setup')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_restart(h, n, pressure) bind(C, name='mppgpu_restart')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: n
       real(c_double) :: pressure(*)
     end function
     integer(c_int) function mppgpu_set_data(h, auxvar_type, var_type, cond_id, n, data) bind(C, name='mppgpu_set_data')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: auxvar_type, var_type, cond_id, n
       real(c_double) :: data(*)
     end function
     integer(c_int) function mppgpu_step_dt(h, dt, nstep, converged, converged_reason) bind(C, name='mppgpu_step_dt')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
       integer(c_int), value :: nstep
       integer(c_int) :: converged, converged_reason
     end function
     integer(c_int) function mppgpu_get_data(h, auxvar_type, var_type, cond_id, n, data) bind(C, name='mppgpu_get_data')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: auxvar_type, var_type, cond_id, n
       real(c_double) :: data(*)
     end function
     integer(c_int) function mppgpu_get_counters(h, newton, linear, cuts, funcs) bind(C, name='mppgpu_get_counters')
       import :: c_int, c_ptr, c_long_long
       type(c_ptr), value :: h
       integer(c_long_long) :: newton, linear, cuts, funcs
     end function
     integer(c_int) function mppgpu_set_dtime(h, dt) bind(C, name='mppgpu_set_dtime')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
     end function
     integer(c_int) function mppgpu_pre_solve(h) bind(C, name='mppgpu_pre_solve')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_residual(h, x, f) bind(C, name='mppgpu_residual')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: x(*), f(*)
     end function
     integer(c_int) function mppgpu_jacobian_nnz(h, nnz) bind(C, name='mppgpu_jacobian_nnz')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int) :: nnz
     end function
     integer(c_int) function mppgpu_jacobian(h, x, rowptr, col, val) bind(C, name='mppgpu_jacobian')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: x(*), val(*)
       integer(c_int) :: rowptr(*), col(*)
     end function
     integer(c_int) function mppgpu_synchronize(h) bind(C, name='mppgpu_synchronize')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_spmv(h, x, y) bind(C, name='mppgpu_spmv')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: x(*), y(*)
     end function
     integer(c_int) function mppgpu_pc_apply(h, r, z) bind(C, name='mppgpu_pc_apply')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: r(*), z(*)
     end function
     integer(c_int) function mppgpu_ksp_solve(h, b, x, its, reason) bind(C, name='mppgpu_ksp_solve')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: b(*), x(*)
       integer(c_int) :: its, reason
     end function
     ! ---- operator split of lateral flow, down-regulated sinks (ELM coupling surface) ----
     integer(c_int) function mppgpu_pre_step_dt(h) bind(C, name='mppgpu_pre_step_dt')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_post_step_dt(h) bind(C, name='mppgpu_post_step_dt')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_set_ghost_cells(h, nghost) bind(C, name='mppgpu_set_ghost_cells')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int), value :: nghost
     end function
     integer(c_int) function mppgpu_set_halo(h, nneigh, peer_rank, send_ptr, send_cells, recv_count) bind(C, name='mppgpu_set_halo')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int), value :: nneigh
       integer(c_int) :: peer_rank(*), send_ptr(*), send_cells(*), recv_count(*)
     end function
     integer(c_int) function mppgpu_set_lateral_connections(h, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec) &
          bind(C, name='mppgpu_set_lateral_connections')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: nconn
       integer(c_int) :: id_up(*), id_dn(*)
       real(c_double) :: dist_up(*), dist_dn(*), area(*), unit_vec(*)
     end function
     integer(c_int) function mppgpu_compute_lateral_flux(h, dt, ss_cond_id) bind(C, name='mppgpu_compute_lateral_flux')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
       integer(c_int), value :: ss_cond_id
     end function
     integer(c_int) function mppgpu_get_condition_values(h, kind, cond_id, n, values) bind(C, name='mppgpu_get_condition_values')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: kind, cond_id, n
       real(c_double) :: values(*)
     end function
     integer(c_int) function mppgpu_set_source_sink_auxvar(h, var_type, n, values) bind(C, name='mppgpu_set_source_sink_auxvar')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: var_type, n
       real(c_double) :: values(*)
     end function
     ! ---- thermal KSP path (sysofeqns_thermal_type) ----
     ! NOTE: by default the thermal operator uses a Dirichlet connection's OWN area in the right-hand side and dhsdT*area in the
     ! heat-flux term; the reference has a stale `area` (GoveqnThermalKSPTemperatureSoilType.F90:906-911) and dhsdT**area (:1215).
     ! mppgpu_set_option(h, 'thermal_reference_defects', '1') reproduces the reference to the letter (INTEGRATION.md section 7).
     integer(c_int) function mppgpu_thermal_set_soils(h, itype, is_soil_shallow, watsat, csol, tkmg, tkdry, dz) &
          bind(C, name='mppgpu_thermal_set_soils')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int) :: itype(*), is_soil_shallow(*)
       real(c_double) :: watsat(*), csol(*), tkmg(*), tkdry(*), dz(*)
     end function
     integer(c_int) function mppgpu_thermal_set_soln_prev_clm(h, n, temperature) bind(C, name='mppgpu_thermal_set_soln_prev_clm')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: n
       real(c_double) :: temperature(*)
     end function
     integer(c_int) function mppgpu_thermal_pre_step_dt(h) bind(C, name='mppgpu_thermal_pre_step_dt')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_thermal_step_dt(h, dt, converged, linear_its) bind(C, name='mppgpu_thermal_step_dt')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
       integer(c_int) :: converged, linear_its
     end function
     integer(c_int) function mppgpu_thermal_get_soln(h, n, temperature) bind(C, name='mppgpu_thermal_get_soln')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: n
       real(c_double) :: temperature(*)
     end function
     integer(c_int) function mppgpu_thermal_compute_rhs(h, dt, b) bind(C, name='mppgpu_thermal_compute_rhs')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
       real(c_double) :: b(*)
     end function
     integer(c_int) function mppgpu_thermal_compute_operators(h, dt, rowptr, col, val) bind(C, name='mppgpu_thermal_compute_operators')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double), value :: dt
       integer(c_int) :: rowptr(*), col(*)
       real(c_double) :: val(*)
     end function
     ! ---- coupled thermal-hydrology (sysofeqns_th_type) ----
     integer(c_int) function mppgpu_th_add_condition(h, igoveqn, kind, cond_type, nconn, id_dn, dist_up, dist_dn, area, &
          unit_vec, swap_order, cond_id) bind(C, name='mppgpu_th_add_condition')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: igoveqn, kind, cond_type, nconn, swap_order
       integer(c_int) :: id_dn(*)
       real(c_double) :: dist_up(*), dist_dn(*), area(*), unit_vec(*)
       integer(c_int) :: cond_id
     end function
     integer(c_int) function mppgpu_th_set_soils(h, heat_cap, therm_cond_wet, therm_cond_dry, therm_alpha, den_soil, &
          perm_x, perm_y, perm_z) bind(C, name='mppgpu_th_set_soils')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       real(c_double) :: heat_cap(*), therm_cond_wet(*), therm_cond_dry(*), therm_alpha(*), den_soil(*)
       type(c_ptr), value :: perm_x, perm_y, perm_z   ! c_loc(array) or c_null_ptr (keep the auxvar default)
     end function
     ! ---- measurement, multi-GPU ----
     integer(c_int) function mppgpu_kernel_stats(h, kclass, launches, total_ms, bytes_per_launch) bind(C, name='mppgpu_kernel_stats')
       import :: c_int, c_ptr, c_double, c_char, c_long_long
       type(c_ptr), value :: h
       character(kind=c_char) :: kclass(*)
       integer(c_long_long) :: launches
       real(c_double) :: total_ms, bytes_per_launch
     end function
     integer(c_int) function mppgpu_kernel_calls(h, kclass, calls, stride) bind(C, name='mppgpu_kernel_calls')
       import :: c_int, c_ptr, c_char, c_long_long
       type(c_ptr), value :: h
       character(kind=c_char) :: kclass(*)
       integer(c_long_long) :: calls
       integer(c_int) :: stride
     end function
     integer(c_int) function mppgpu_reset_stats(h) bind(C, name='mppgpu_reset_stats')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
     end function
     integer(c_int) function mppgpu_get_step_history(h, max_n, dt_sub, n) bind(C, name='mppgpu_get_step_history')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: max_n
       real(c_double) :: dt_sub(*)
       integer(c_int) :: n
     end function
     integer(c_int) function mppgpu_get_comm_stats(h, allreduces, halo_exchanges, chain_fallbacks, host_readbacks) &
          bind(C, name='mppgpu_get_comm_stats')
       import :: c_int, c_ptr, c_long_long
       type(c_ptr), value :: h
       integer(c_long_long) :: allreduces, halo_exchanges, chain_fallbacks, host_readbacks
     end function
     integer(c_int) function mppgpu_get_halo_stats(h, neighbours, cells_sent, cells_received, bytes_received_per_exchange, &
          max_cells_from_one_neighbour) bind(C, name='mppgpu_get_halo_stats')
       import :: c_int, c_ptr, c_long_long
       type(c_ptr), value :: h
       integer(c_long_long) :: neighbours, cells_sent, cells_received, bytes_received_per_exchange, max_cells_from_one_neighbour
     end function
     integer(c_int) function mppgpu_get_assembly_info(h, n, info) bind(C, name='mppgpu_get_assembly_info')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int), value :: n
       integer(c_int) :: info(*)
     end function
     integer(c_int) function mppgpu_get_pc_info(h, path, launches, box_extents) bind(C, name='mppgpu_get_pc_info')
       import :: c_int, c_ptr
       type(c_ptr), value :: h
       integer(c_int) :: path, launches, box_extents(3)
     end function
     integer(c_int) function mppgpu_bench_kernel(h, kernel, variant, reps, avg_ms, bytes) bind(C, name='mppgpu_bench_kernel')
       import :: c_int, c_ptr, c_double, c_char
       type(c_ptr), value :: h
       character(kind=c_char) :: kernel(*)
       integer(c_int), value :: variant, reps
       real(c_double) :: avg_ms, bytes
     end function
     ! the constitutive laws on the host (what a driver calls where the reference's drivers call Density,
     ! InternalEnergyAndEnthalpy, SatFunc_PressToSat / _PressToRelPerm directly, e.g. th_mms_problem.F90:1262-1452)
     integer(c_int) function mppgpu_host_density(density_type, n, p, t_K, den, dden_dp, dden_dt) bind(C, name='mppgpu_host_density')
       import :: c_int, c_double
       integer(c_int), value :: density_type, n
       real(c_double) :: p(*), t_K(*), den(*), dden_dp(*), dden_dt(*)
     end function
     integer(c_int) function mppgpu_host_internal_energy_enthalpy(ie_type, n, p, t_K, den, dden_dt, dden_dp, u, h, du_dt, dh_dt, &
          du_dp, dh_dp) bind(C, name='mppgpu_host_internal_energy_enthalpy')
       import :: c_int, c_double
       integer(c_int), value :: ie_type, n
       real(c_double) :: p(*), t_K(*), den(*), dden_dt(*), dden_dp(*), u(*), h(*), du_dt(*), dh_dt(*), du_dp(*), dh_dp(*)
     end function
     integer(c_int) function mppgpu_host_sat_relperm(n, satfunc_type, p, sat_res, alpha, lambda, sat, dsat_dp, kr, dkr_dp) &
          bind(C, name='mppgpu_host_sat_relperm')
       import :: c_int, c_double
       integer(c_int), value :: n
       integer(c_int) :: satfunc_type(*)
       real(c_double) :: p(*), sat_res(*), alpha(*), lambda(*), sat(*), dsat_dp(*), kr(*), dkr_dp(*)
     end function
     integer(c_int) function mppgpu_comm_runtime_version(version) bind(C, name='mppgpu_comm_runtime_version')
       import :: c_int
       integer(c_int) :: version
     end function
     integer(c_int) function mppgpu_comm_get_unique_id(id128) bind(C, name='mppgpu_comm_get_unique_id')
       import :: c_int, c_char
       character(kind=c_char) :: id128(128)
     end function
     integer(c_int) function mppgpu_comm_init_local(handles, nranks) bind(C, name='mppgpu_comm_init_local')
       import :: c_int, c_ptr
       type(c_ptr) :: handles(*)
       integer(c_int), value :: nranks
     end function
     integer(c_int) function mppgpu_comm_init(h, nranks, rank, id128) bind(C, name='mppgpu_comm_init')
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: h
       integer(c_int), value :: nranks, rank
       character(kind=c_char) :: id128(128)
     end function
  end interface

  integer(c_int), parameter, public :: VAR_LIQ_AREAL_DEN = 615, VAR_TUNING_FACTOR = 630, VAR_FRAC = 618, VAR_ACTIVE = 624

  !> sysofeqns_vsfm_type stand-in: same method names as the reference's SoE and governing equation
  !! (SoE: SetDataFromCLM / GetDataForCLM / PreStepDT / StepDT / PostStepDT / ComputeLateralFlux,
  !!  src/mpp/soe/SystemOfEquationsVSFMType.F90:663-940; goveqn_base_type: ComputeResidual / ComputeJacobian / SetDtime /
  !!  PreSolve, src/mpp/dtypes/GoverningEquationBaseType.F90:46-71)
  type, public :: sysofeqns_vsfm_gpu_type
     type(c_ptr) :: h = c_null_ptr
     integer     :: ncells = 0
   contains
     procedure :: Create, Destroy, SetMesh, SetConnections, AddConditionInGovEqn, SetOption, SetupProblem
     procedure :: SetSoilPermeability, SetSoilPorosity, SetSoilPorosityLinear, SetSaturationFunction, SetDensityType, Restart
     procedure :: SetDataFromCLM, StepDT, GetDataForCLM, SetDtime, PreSolve, ComputeResidual, LastError
     procedure :: JacobianNNZ, ComputeJacobian, MatMult, PCApply, KSPSolve
     procedure :: PreStepDT, PostStepDT, SetLateralConnections, ComputeLateralFlux, SetSourceSinkAuxVarRealValue
     procedure :: GetCounters, GetStepHistory
  end type

  !> sysofeqns_thermal_type stand-in (soil heat diffusion, KSP): src/mpp/soe/SystemOfEquationsThermalType.F90:234-759,
  !! goveqn ComputeRHS / ComputeOperatorsDiag (src/mpp/ge/GoveqnThermalKSPTemperatureSoilType.F90:646-1229)
  type, public, extends(sysofeqns_vsfm_gpu_type) :: sysofeqns_thermal_gpu_type
   contains
     procedure :: Create => ThermalCreate
     procedure :: SetSoils => ThermalSetSoils
     procedure :: SetSolnPrevCLM => ThermalSetSolnPrevCLM
     procedure :: PreStepDT => ThermalPreStepDT
     procedure :: StepDTKSP => ThermalStepDT
     procedure :: GetSoln => ThermalGetSoln
     procedure :: ComputeRHS => ThermalComputeRHS
     procedure :: ComputeOperatorsDiag => ThermalComputeOperators
  end type

  !> sysofeqns_th_type stand-in (Richards mass equation + enthalpy-based heat equation, unknowns [P; T]):
  !! src/mpp/soe/SystemOfEquationsTHType.F90:736-1003
  type, public, extends(sysofeqns_vsfm_gpu_type) :: sysofeqns_th_gpu_type
   contains
     procedure :: Create => THCreate
     procedure :: AddConditionInGovEqnTH
     procedure :: SetSoilsTH
  end type

contains

  function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char) :: c(len_trim(s) + 1)
    integer :: i
    do i = 1, len_trim(s)
       c(i) = s(i:i)
    end do
    c(len_trim(s) + 1) = c_null_char
  end function

  subroutine Create(this, device_id, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: device_id
    integer, intent(out) :: ierr
    ierr = mppgpu_create(int(device_id, c_int), this%h)
  end subroutine
  subroutine Destroy(this)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer :: ierr
    if (c_associated(this%h)) ierr = mppgpu_destroy(this%h)
    this%h = c_null_ptr
  end subroutine
  function LastError(this) result(msg)
    class(sysofeqns_vsfm_gpu_type) :: this
    character(len=512) :: msg
    character(kind=c_char), pointer :: p(:)
    integer :: i
    msg = ''
    call c_f_pointer(mppgpu_last_error(this%h), p, [512])
    do i = 1, 512
       if (p(i) == c_null_char) exit
       msg(i:i) = p(i)
    end do
  end function
  subroutine SetMesh(this, ncells, nlev, vol, is_active, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: ncells, nlev
    real(8), intent(in) :: vol(:)
    integer, intent(in) :: is_active(:)
    integer, intent(out) :: ierr
    this%ncells = ncells
    ierr = mppgpu_set_mesh(this%h, ncells, nlev, vol, is_active)
  end subroutine
  subroutine SetConnections(this, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: nconn, id_up(:), id_dn(:)
    real(8), intent(in) :: dist_up(:), dist_dn(:), area(:), unit_vec(:)   ! unit_vec: 3*nconn, (x,y,z) per connection
    integer, intent(out) :: ierr
    ierr = mppgpu_set_connections(this%h, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec)
  end subroutine
  subroutine AddConditionInGovEqn(this, ss_or_bc_type, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, cond_id, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: ss_or_bc_type, cond_type, nconn, id_dn(:)
    real(8), intent(in) :: dist_up(:), dist_dn(:), area(:), unit_vec(:)
    integer, intent(out) :: cond_id, ierr
    ierr = mppgpu_add_condition(this%h, ss_or_bc_type, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, 0, cond_id)
  end subroutine
  subroutine SetOption(this, key, val, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    character(len=*), intent(in) :: key, val
    integer, intent(out) :: ierr
    ierr = mppgpu_set_option(this%h, cstr(key), cstr(val))
  end subroutine
  subroutine SetupProblem(this, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(out) :: ierr
    ierr = mppgpu_setup(this%h)
  end subroutine
  subroutine SetSoilPermeability(this, perm_x, perm_y, perm_z, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: perm_x(:), perm_y(:), perm_z(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_soil_permeability(this%h, perm_x, perm_y, perm_z)
  end subroutine
  subroutine SetSoilPorosity(this, por, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: por(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_soil_porosity(this%h, por)
  end subroutine
  subroutine SetSaturationFunction(this, satfunc_type, alpha, lambda, sat_res, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: satfunc_type(:)
    real(8), intent(in) :: alpha(:), lambda(:), sat_res(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_saturation_function(this%h, satfunc_type, alpha, lambda, sat_res)
  end subroutine
  subroutine SetDensityType(this, density_type, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: density_type
    integer, intent(out) :: ierr
    ierr = mppgpu_set_density_type(this%h, density_type)
  end subroutine
  subroutine Restart(this, data_1d, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: data_1d(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_restart(this%h, size(data_1d), data_1d)
  end subroutine
  subroutine SetDataFromCLM(this, soe_auxvar_type, var_type, soe_auxvar_id, data_1d, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: soe_auxvar_type, var_type, soe_auxvar_id
    real(8), intent(in) :: data_1d(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_data(this%h, soe_auxvar_type, var_type, soe_auxvar_id, size(data_1d), data_1d)
  end subroutine
  subroutine StepDT(this, dt, nstep, converged, converged_reason, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: dt
    integer, intent(in) :: nstep
    logical, intent(out) :: converged
    integer, intent(out) :: converged_reason, ierr
    integer(c_int) :: c
    ierr = mppgpu_step_dt(this%h, dt, nstep, c, converged_reason)
    converged = (c /= 0)
  end subroutine
  subroutine GetDataForCLM(this, soe_auxvar_type, var_type, soe_auxvar_id, data_1d, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: soe_auxvar_type, var_type, soe_auxvar_id
    real(8), intent(inout) :: data_1d(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_get_data(this%h, soe_auxvar_type, var_type, soe_auxvar_id, size(data_1d), data_1d)
  end subroutine
  subroutine SetDtime(this, dt, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: dt
    integer, intent(out) :: ierr
    ierr = mppgpu_set_dtime(this%h, dt)
  end subroutine
  subroutine PreSolve(this, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(out) :: ierr
    ierr = mppgpu_pre_solve(this%h)
  end subroutine
  subroutine ComputeResidual(this, X, F, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: X(:)
    real(8), intent(inout) :: F(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_residual(this%h, X, F)
  end subroutine

  subroutine SetSoilPorosityLinear(this, por_base, pressure_reference, slope, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: por_base(:), pressure_reference(:), slope(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_soil_porosity_linear(this%h, por_base, pressure_reference, slope)
  end subroutine
  subroutine JacobianNNZ(this, nnz, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(out) :: nnz, ierr
    ierr = mppgpu_jacobian_nnz(this%h, nnz)
  end subroutine
  !> goveqn%ComputeJacobian(X, A, B): the CSR matrix in the caller's ordering, 0-based rowptr / col (PETSc's convention)
  subroutine ComputeJacobian(this, X, rowptr, col, val, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: X(:)
    integer, intent(inout) :: rowptr(:), col(:)
    real(8), intent(inout) :: val(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_jacobian(this%h, X, rowptr, col, val)
  end subroutine
  subroutine MatMult(this, x, y, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: x(:)
    real(8), intent(inout) :: y(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_spmv(this%h, x, y)
  end subroutine
  subroutine PCApply(this, r, z, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: r(:)
    real(8), intent(inout) :: z(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_pc_apply(this%h, r, z)
  end subroutine
  subroutine KSPSolve(this, b, x, its, reason, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: b(:)
    real(8), intent(inout) :: x(:)
    integer, intent(out) :: its, reason, ierr
    ierr = mppgpu_ksp_solve(this%h, b, x, its, reason)
  end subroutine
  subroutine PreStepDT(this, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(out) :: ierr
    ierr = mppgpu_pre_step_dt(this%h)
  end subroutine
  subroutine PostStepDT(this, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(out) :: ierr
    ierr = mppgpu_post_step_dt(this%h)
  end subroutine
  subroutine SetLateralConnections(this, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: nconn, id_up(:), id_dn(:)
    real(8), intent(in) :: dist_up(:), dist_dn(:), area(:), unit_vec(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_lateral_connections(this%h, nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec)
  end subroutine
  subroutine ComputeLateralFlux(this, dt, ss_cond_id, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(in) :: dt
    integer, intent(in) :: ss_cond_id
    integer, intent(out) :: ierr
    ierr = mppgpu_compute_lateral_flux(this%h, dt, ss_cond_id)
  end subroutine
  subroutine SetSourceSinkAuxVarRealValue(this, var_type, values, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer, intent(in) :: var_type
    real(8), intent(in) :: values(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_set_source_sink_auxvar(this%h, var_type, size(values), values)
  end subroutine
  subroutine GetCounters(this, newton, linear, cuts, funcs, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    integer(8), intent(out) :: newton, linear, cuts, funcs
    integer, intent(out) :: ierr
    integer(c_long_long) :: a, b, c, d
    ierr = mppgpu_get_counters(this%h, a, b, c, d)
    newton = a; linear = b; cuts = c; funcs = d
  end subroutine
  subroutine GetStepHistory(this, dt_sub, n, ierr)
    class(sysofeqns_vsfm_gpu_type) :: this
    real(8), intent(inout) :: dt_sub(:)
    integer, intent(out) :: n, ierr
    ierr = mppgpu_get_step_history(this%h, size(dt_sub), dt_sub, n)
  end subroutine

  ! ---- thermal (KSP) ----
  subroutine ThermalCreate(this, device_id, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    integer, intent(in) :: device_id
    integer, intent(out) :: ierr
    ierr = mppgpu_create(int(device_id, c_int), this%h)
    if (ierr == 0) ierr = mppgpu_set_option(this%h, cstr('problem_type'), cstr('thermal'))
  end subroutine
  subroutine ThermalSetSoils(this, itype, is_soil_shallow, watsat, csol, tkmg, tkdry, dz, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    integer, intent(in) :: itype(:), is_soil_shallow(:)
    real(8), intent(in) :: watsat(:), csol(:), tkmg(:), tkdry(:), dz(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_set_soils(this%h, itype, is_soil_shallow, watsat, csol, tkmg, tkdry, dz)
  end subroutine
  subroutine ThermalSetSolnPrevCLM(this, temperature, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    real(8), intent(in) :: temperature(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_set_soln_prev_clm(this%h, size(temperature), temperature)
  end subroutine
  subroutine ThermalPreStepDT(this, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_pre_step_dt(this%h)
  end subroutine
  !> soe%StepDT of the KSP path (SOEBaseStepDT_KSP, SystemOfEquationsBaseType.F90:555-647)
  subroutine ThermalStepDT(this, dt, converged, linear_its, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    real(8), intent(in) :: dt
    logical, intent(out) :: converged
    integer, intent(out) :: linear_its, ierr
    integer(c_int) :: c
    ierr = mppgpu_thermal_step_dt(this%h, dt, c, linear_its)
    converged = (c /= 0)
  end subroutine
  subroutine ThermalGetSoln(this, temperature, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    real(8), intent(inout) :: temperature(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_get_soln(this%h, size(temperature), temperature)
  end subroutine
  subroutine ThermalComputeRHS(this, dt, b, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    real(8), intent(in) :: dt
    real(8), intent(inout) :: b(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_compute_rhs(this%h, dt, b)
  end subroutine
  subroutine ThermalComputeOperators(this, dt, rowptr, col, val, ierr)
    class(sysofeqns_thermal_gpu_type) :: this
    real(8), intent(in) :: dt
    integer, intent(inout) :: rowptr(:), col(:)
    real(8), intent(inout) :: val(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_thermal_compute_operators(this%h, dt, rowptr, col, val)
  end subroutine

  ! ---- coupled thermal-hydrology ----
  subroutine THCreate(this, device_id, ierr)
    class(sysofeqns_th_gpu_type) :: this
    integer, intent(in) :: device_id
    integer, intent(out) :: ierr
    ierr = mppgpu_create(int(device_id, c_int), this%h)
    if (ierr == 0) ierr = mppgpu_set_option(this%h, cstr('problem_type'), cstr('th'))
  end subroutine
  subroutine AddConditionInGovEqnTH(this, igoveqn, ss_or_bc_type, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, &
                                    cond_id, ierr)
    class(sysofeqns_th_gpu_type) :: this
    integer, intent(in) :: igoveqn, ss_or_bc_type, cond_type, nconn, id_dn(:)
    real(8), intent(in) :: dist_up(:), dist_dn(:), area(:), unit_vec(:)
    integer, intent(out) :: cond_id, ierr
    ierr = mppgpu_th_add_condition(this%h, igoveqn, ss_or_bc_type, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, &
                                   0, cond_id)
  end subroutine
  !> MPPTHSetSoils (src/mpp/mpp/MultiPhysicsProbTH.F90:211-300): the heat equation keeps its default permeability
  subroutine SetSoilsTH(this, heat_cap, therm_cond_wet, therm_cond_dry, therm_alpha, den_soil, ierr)
    class(sysofeqns_th_gpu_type) :: this
    real(8), intent(in) :: heat_cap(:), therm_cond_wet(:), therm_cond_dry(:), therm_alpha(:), den_soil(:)
    integer, intent(out) :: ierr
    ierr = mppgpu_th_set_soils(this%h, heat_cap, therm_cond_wet, therm_cond_dry, therm_alpha, den_soil, c_null_ptr, c_null_ptr, &
                               c_null_ptr)
  end subroutine

end module mppgpu_mod
