!> The reference's vsfm_celia1990 problem (src/driver/standalone/vsfm/vsfm_celia1990_problem.F90) driven through the
!! iso_c_binding shim: nz = 100 cells, 24 steps of 3600 s, van Genuchten, TGDPB01, Dirichlet top/bottom.
!! Prints the regression block of regression_mod%WriteData (src/driver/standalone/util/regression.F90:76-124).
program vsfm_celia1990_gpu
  use mppgpu_mod
  implicit none
  integer, parameter :: nz = 100, nstep = 24
  real(8), parameter :: dtime = 3600.d0, z_column = 1.d0
  real(8), parameter :: porosity = 0.368d0, lambda = 0.5d0, alpha0 = 3.4257d-4, perm0 = 8.3913d-12
  real(8), parameter :: vish2o = 0.001002d0, denh2o = 1.0d3, grav = 9.80616d0, GRAVITY_CONSTANT = 9.80665d0
  type(sysofeqns_vsfm_gpu_type) :: soe
  real(8) :: vol(nz), dz, hksat, sucsat, perm, alpha
  integer :: is_active(nz), id_up(nz-1), id_dn(nz-1), satf(nz), ierr, istep, i, reason, bc_top, bc_bot
  real(8) :: dist(nz-1), area(nz-1), uvec(3*(nz-1)), p(nz), sat(nz), arr(nz)
  integer :: idtop(1), idbot(1)
  real(8) :: one(1), zero(1), half(1), utop(3), ubot(3), vtop(1), vbot(1)
  logical :: converged

  dz = z_column/nz
  vol = 1.d0*1.d0*dz
  is_active = 1
  do i = 1, nz-1
     id_up(i) = i; id_dn(i) = i+1
     dist(i) = 0.5d0*dz; area(i) = 1.d0
     uvec(3*i-2:3*i) = [0.d0, 0.d0, 1.d0]
  end do
  ! CLM unit round trip (celia1990_problem.F90:325-327; MultiPhysicsProbVSFM.F90:374-381)
  hksat  = perm0/vish2o*(denh2o*grav)/0.001d0
  sucsat = 1.d0/(alpha0*GRAVITY_CONSTANT)
  perm   = hksat*vish2o/(denh2o*grav)*0.001d0
  alpha  = 1.d0/(sucsat*grav)

  call soe%Create(0, ierr);                                              call chk('create')
  call soe%SetMesh(nz, nz, vol, is_active, ierr);                        call chk('mesh')
  call soe%SetConnections(nz-1, id_up, id_dn, dist, dist, area, uvec, ierr); call chk('conn')
  idtop = nz; idbot = 1; one = 1.d0; zero = 0.d0; half = 0.5d0*dz
  utop = [0.d0, 0.d0, -1.d0]; ubot = [0.d0, 0.d0, 1.d0]
  call soe%AddConditionInGovEqn(COND_BC, COND_DIRICHLET, 1, idtop, zero, half, one, utop, bc_top, ierr); call chk('bc top')
  call soe%AddConditionInGovEqn(COND_BC, COND_DIRICHLET, 1, idbot, zero, half, one, ubot, bc_bot, ierr); call chk('bc bot')
  call soe%SetOption('ksp_type', 'bcgs', ierr);                          call chk('opt')
  call soe%SetOption('pc_type', 'bjacobi', ierr);                        call chk('opt')
  call soe%SetupProblem(ierr);                                           call chk('setup')
  arr = perm
  call soe%SetSoilPermeability(arr, arr, arr, ierr);                     call chk('perm')
  arr = porosity
  call soe%SetSoilPorosity(arr, ierr);                                   call chk('por')
  satf = SAT_FUNC_VAN_GENUCHTEN
  block
    real(8) :: a(nz), l(nz), sr(nz)
    a = alpha; l = lambda; sr = 0.2772d0
    call soe%SetSaturationFunction(satf, a, l, sr, ierr);                call chk('satfunc')
  end block
  call soe%SetDensityType(DENSITY_TGDPB01, ierr);                        call chk('density')
  p = 3.5355d3
  call soe%Restart(p, ierr);                                             call chk('restart')

  vtop = 9.3991d4; vbot = 3.5355d3
  do istep = 1, nstep
     call soe%SetDataFromCLM(AUXVAR_BC, VAR_BC_SS_CONDITION, bc_top, vtop, ierr); call chk('set bc')
     call soe%SetDataFromCLM(AUXVAR_BC, VAR_BC_SS_CONDITION, bc_bot, vbot, ierr); call chk('set bc')
     call soe%StepDT(dtime, istep, converged, reason, ierr);             call chk('StepDT')
     if (.not. converged) then
        write(*,*) 'StepDT did not converge, reason = ', reason
        stop 1
     end if
  end do
  call soe%GetDataForCLM(AUXVAR_INTERNAL, VAR_PRESSURE, -1, p, ierr);    call chk('get p')
  call soe%GetDataForCLM(AUXVAR_INTERNAL, VAR_LIQ_SAT, -1, sat, ierr);   call chk('get sat')
  call write_data('liquid_pressure', 'pressure', p)
  call write_data('liquid_saturation', 'general', sat)
  call soe%Destroy()

contains
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
    do cell = 1, size(data), size(data)/5
       write(*, '("cell ", i4, " = ", e21.13)') cell, data(cell)
    end do
    write(*, '(a)')
  end subroutine
end program
