!> The reference's vsfm_mms problem (src/driver/standalone/vsfm/vsfm_mms_problem.F90 with
!! vsfm_mms_steady_state_soil_only_1D.F90) driven through the iso_c_binding shim: a steady horizontal column x in [0, 10] m of
!! nx cells, porosity 0, manufactured pressure P(x) = 15000 sin(pi x / 10) -+ 20000 + 101325 (:43-63), permeability
!! 1e-11 (2 + cos(pi x / 10)) (:66-79), van Genuchten alpha = 1/4000, m = 1/2, sr = 1/2 (:89-121), TGDPB01 density, Dirichlet ends,
!! mass source -d/dx(rho k kr / mu dP/dx) dx per cell (DATA_MASS_SOURCE :263-302) with the constitutive laws evaluated on the
!! host through mppgpu_host_density / mppgpu_host_sat_relperm where the reference's driver calls Density / SatFunc_* itself,
!! the per-cell-array setters (:405-418), one StepDT(1.0) (vsfm_mms_problem.F90:97-102).
!! usage: vsfm_mms_gpu [nx] [fully_saturated 0|1]; prints the regression block of the pressure and the max-norm error.
program vsfm_mms_gpu
  use mppgpu_mod
  implicit none
  real(8), parameter :: x_min = 0.d0, x_max = 10.d0, a0 = 15000.d0, a1 = -20000.d0, PRESSURE_REF = 101325.d0
  real(8), parameter :: mu = 8.904156d-4, FMWH2O = 18.01534d0                  ! EOSWaterMod.F90:582, mpp_varcon
  type(sysofeqns_vsfm_gpu_type) :: soe
  integer :: nx, fully_saturated, ierr, i, reason, bc_id, ss_id
  logical :: converged
  real(8) :: dx, xlim, pi, P0
  real(8), allocatable :: xc(:), vol(:), k(:), dk_dx(:), P(:), dP_dx(:), d2P_dx2(:), tK(:), den(:), dden_dp(:), dden_dt(:)
  real(8), allocatable :: sr(:), alpha(:), lam(:), se(:), dse_dp(:), kr(:), dkr_dp(:), src(:), sol(:), zero(:), one(:), uv0(:)
  real(8), allocatable :: dist(:), area(:), uvec(:)
  integer, allocatable :: is_active(:), id_up(:), id_dn(:), satf(:), cells(:)
  integer :: bc_cells(2)
  real(8) :: bc_up(2), bc_dn(2), bc_area(2), bc_uvec(6), bc_val(2), dummy, dummy2
  character(len=32) :: arg

  nx = 20; fully_saturated = 0
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) nx
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg, *) fully_saturated
  end if
  pi = 4.d0*atan(1.d0); xlim = x_max - x_min; dx = xlim/nx
  allocate(xc(nx), vol(nx), k(nx), dk_dx(nx), P(nx), dP_dx(nx), d2P_dx2(nx), tK(nx), den(nx), dden_dp(nx), dden_dt(nx))
  allocate(sr(nx), alpha(nx), lam(nx), se(nx), dse_dp(nx), kr(nx), dkr_dp(nx), src(nx), sol(nx), zero(nx), one(nx), uv0(3*nx))
  allocate(is_active(nx), id_up(nx-1), id_dn(nx-1), satf(nx), cells(nx), dist(nx-1), area(nx-1), uvec(3*(nx-1)))

  do i = 1, nx
     xc(i) = x_min + (i - 0.5d0)*dx
     call pressure(xc(i), P(i), dP_dx(i), d2P_dx2(i))
     k(i)     = 1.d-11*(2.d0 + cos((xc(i) - x_min)/xlim*pi))
     dk_dx(i) = 1.d-11*pi/xlim*(-sin((xc(i) - x_min)/xlim*pi))
     cells(i) = i
  end do
  vol = dx; is_active = 1; zero = 0.d0; one = 1.d0; uv0 = 0.d0
  do i = 1, nx-1
     id_up(i) = i; id_dn(i) = i+1; dist(i) = 0.5d0*dx; area(i) = 1.d0
     uvec(3*i-2:3*i) = [1.d0, 0.d0, 0.d0]
  end do

  ! the manufactured source, the laws on the host
  tK = 298.15d0; satf = SAT_FUNC_VAN_GENUCHTEN; sr = 0.5d0; alpha = 1.d0/4000.d0; lam = 0.5d0
  ierr = mppgpu_host_density(DENSITY_TGDPB01, nx, P, tK, den, dden_dp, dden_dt);            call chk('host density')
  ierr = mppgpu_host_sat_relperm(nx, satf, P, sr, alpha, lam, se, dse_dp, kr, dkr_dp);      call chk('host sat')
  do i = 1, nx
     src(i) = (-((k(i)*kr(i)/mu)*(dden_dp(i)*FMWH2O*dP_dx(i)) + (den(i)*FMWH2O*kr(i)/mu)*dk_dx(i) &
                 + (den(i)*FMWH2O*k(i)/mu)*(dkr_dp(i)*dP_dx(i)))*dP_dx(i) - (den(i)*FMWH2O*k(i)*kr(i)/mu)*d2P_dx2(i))*dx
  end do
  P0 = 0.d0
  do i = 1, nx
     P0 = P0 + 1.d0/nx*P(i)                                              ! DATA_INITIAL_PRESSURE :196-203
  end do

  call soe%Create(0, ierr);                                              call chk('create')
  call soe%SetMesh(nx, 1, vol, is_active, ierr);                         call chk('mesh')
  call soe%SetConnections(nx-1, id_up, id_dn, dist, dist, area, uvec, ierr); call chk('conn')
  ! ComputeXBoundaryDomainConnection (mpp_mesh_utils.F90:783-812): dist_up 0, dist_dn dx/2, +x at x_min and -x at x_max
  bc_cells = [1, nx]; bc_up = 0.d0; bc_dn = 0.5d0*dx; bc_area = 1.d0
  bc_uvec = [1.d0, 0.d0, 0.d0, -1.d0, 0.d0, 0.d0]
  call soe%AddConditionInGovEqn(COND_BC, COND_DIRICHLET, 2, bc_cells, bc_up, bc_dn, bc_area, bc_uvec, bc_id, ierr); call chk('bc')
  call soe%AddConditionInGovEqn(COND_SS, COND_MASS_RATE, nx, cells, zero, zero, one, uv0, ss_id, ierr);            call chk('ss')
  call soe%SetOption('ksp_type', 'gmres', ierr);                         call chk('opt')
  call soe%SetOption('pc_type', 'ilu', ierr);                            call chk('opt')
  call soe%SetupProblem(ierr);                                           call chk('setup')
  call soe%SetSoilPermeability(k, k, k, ierr);                           call chk('perm')
  call soe%SetSoilPorosity(zero, ierr);                                  call chk('por')
  call soe%SetSaturationFunction(satf, alpha, lam, sr, ierr);            call chk('satfunc')
  call soe%SetDensityType(DENSITY_TGDPB01, ierr);                        call chk('density')
  call pressure(xc(1) - 0.5d0*dx, bc_val(1), dummy, dummy2)
  call pressure(xc(nx) + 0.5d0*dx, bc_val(2), dummy, dummy2)
  call soe%SetDataFromCLM(AUXVAR_BC, VAR_BC_SS_CONDITION, bc_id, bc_val, ierr); call chk('set bc')
  call soe%SetDataFromCLM(AUXVAR_SS, VAR_BC_SS_CONDITION, ss_id, src, ierr);    call chk('set ss')
  sol = P0
  call soe%Restart(sol, ierr);                                           call chk('restart')
  call soe%PreStepDT(ierr);                                              call chk('PreStepDT')
  call soe%StepDT(1.d0, 1, converged, reason, ierr);                     call chk('StepDT')
  if (.not. converged) then
     write(*,*) 'StepDT did not converge, reason = ', reason
     stop 1
  end if
  call soe%GetDataForCLM(AUXVAR_INTERNAL, VAR_PRESSURE, -1, sol, ierr);  call chk('get p')
  call write_data('liquid_pressure', 'pressure', sol)
  write(*, '("max_error = ",e21.13)') maxval(abs(sol - P))
  call soe%Destroy()

contains
  subroutine pressure(x, val, ddx, d2dx2)                                  ! compute_pressure_or_deriv :43-63
    real(8), intent(in) :: x
    real(8), intent(out) :: val, ddx, d2dx2
    real(8) :: shift
    shift = a1
    if (fully_saturated /= 0) shift = -a1
    val   = a0*sin((x - x_min)/xlim*pi) + shift + PRESSURE_REF
    ddx   = a0*pi/xlim*cos((x - x_min)/xlim*pi)
    d2dx2 = -a0*pi*pi/xlim/xlim*sin((x - x_min)/xlim*pi)
  end subroutine
  subroutine chk(what)
    character(len=*), intent(in) :: what
    if (ierr /= 0) then
       write(*,*) 'mppgpu error in ', what, ': ', trim(soe%LastError())
       stop 2
    end if
  end subroutine
  subroutine write_data(name, category, data)
    character(len=*), intent(in) :: name, category
    real(8), intent(in) :: data(:)
    integer :: cell
    write(*, '("[",a,"]")') trim(name)
    write(*, '("category = ",a)') trim(category)
    write(*, '("min = ",e21.13)') minval(data)
    write(*, '("max = ",e21.13)') maxval(data)
    write(*, '("mean = ",e21.13)') sum(data)/size(data)
    do cell = 1, size(data), max(1, size(data)/5)
       write(*, '("cell ", i6, " = ", e21.13)') cell, data(cell)
    end do
    write(*, '(a)')
  end subroutine
end program
