! Drop-in bodies for SPEEDY's adiabatic time stepping, forwarding to the MI355X library (csrc/dynamics.hip).
! Same names and argument meaning as the reference's external subroutines:
!     impint(dt, alph)                     src/ini_impint.f90
!     step(j1, j2, dt, alph, rob, wil)     src/dyn_step.f90:1-128   (grtend's phypar call included once dyn_hip_physics_init ran)
!     stepone                              src/ini_stepone.f90
! plus the device-resident form of the hybrid window (stepone + the stloop inner loop, src/dyn_stloop.f90:28-43):
!     dyn_hip_window(nsteps)
! The reference keeps the prognostic variables in module mod_dynvar (vor, div, t, ps, tr, phis) and the diffusion
! corrections in mod_hdifcon (tcorh, qcorh); in the patched tree these routines `use` those modules (INTEGRATION.md).  To
! build stand-alone this file carries a module `speedy_state` with exactly those arrays in the reference's shapes
! (src/mod_dynvar.f90:14-27, src/mod_hdifcon.f90:19, src/mod_atparam.f90:9-14).
module speedy_state
  use iso_c_binding
  implicit none
  integer, parameter :: mx = 31, nx = 32, kx = 8, ntr = 1
  complex(c_double_complex) :: vor(mx,nx,kx,2), div(mx,nx,kx,2), t(mx,nx,kx,2), ps(mx,nx,2), tr(mx,nx,kx,2,ntr)
  complex(c_double_complex) :: phis(mx,nx), tcorh(mx,nx), qcorh(mx,nx)
  logical :: lradsw = .true.               ! src/mod_lflags.f90:22
end module

module speedy_dyn_hip
  use iso_c_binding
  use speedyml_hip
  use speedy_state
  implicit none
  type(c_ptr), save :: sp_h = c_null_ptr, dyn_h = c_null_ptr, state_dev = c_null_ptr, phys_h = c_null_ptr
  ! src/mod_tsteps.f90:19,84-96
  real(c_double), parameter :: delt = 86400.0_c_double/96, delt2 = 2*delt, rob = 0.05_c_double, wil = 0.53_c_double
  real(c_double), save :: alph = 0.5_c_double
contains

  ! once, after parmtr/indyns of the reference (the library builds its own copies of their tables)
  subroutine dyn_hip_init(rearth)
    real(c_double), intent(in) :: rearth
    call sml_check(sml_spectral_create(rearth, sp_h), 'sml_spectral_create')
    call sml_check(sml_dyn_create(sp_h, dyn_h), 'sml_dyn_create')
    call sml_check(sml_dyn_state_dev(dyn_h, state_dev), 'sml_dyn_state_dev')
  end subroutine

  ! once, after inphys: the column physics joins every later time step (src/dyn_grtend.f90:222-225).  hsg(9): half sigma levels
  ! (src/ini_indyns.f90:38-41); radang(48): latitudes in radians, south to north (:72-80); nstrad: src/mod_tsteps.f90:65
  subroutine dyn_hip_physics_init(hsg, radang, nstrad)
    real(c_double), intent(in) :: hsg(9), radang(48)
    integer, intent(in) :: nstrad
    call sml_check(sml_phys_create(hsg, radang, phys_h), 'sml_phys_create')
    call sml_check(sml_dyn_attach_physics(dyn_h, phys_h, int(nstrad, c_int)), 'sml_dyn_attach_physics')
  end subroutine

  ! after fordate (daily) / the coupler: surface boundary fields as phypar reads them (fmask1, phis0 of mod_surfcon; stl_am,
  ! soilw_am of mod_var_land; sst_am of mod_var_sea; alb_l, alb_s, albsfc, snowc of mod_radcon) and sol_oz(tyear)
  subroutine dyn_hip_surface(fmask1, phis0, stl_am, sst_am, soilw_am, alb_l, alb_s, albsfc, snowc, tyear)
    real(c_double), intent(in) :: fmask1(96,48), phis0(96,48), stl_am(96,48), sst_am(96,48), soilw_am(96,48)
    real(c_double), intent(in) :: alb_l(96,48), alb_s(96,48), albsfc(96,48), snowc(96,48), tyear
    call sml_check(sml_phys_set_surface(phys_h, fmask1, phis0, stl_am, sst_am, soilw_am, alb_l, alb_s, albsfc, snowc), 'sml_phys_set_surface')
    call sml_check(sml_phys_sol_oz(phys_h, tyear), 'sml_phys_sol_oz')
  end subroutine

  ! after fordate (daily): surface geopotential and the diffusion correction terms
  subroutine dyn_hip_boundary()
    call sml_check(sml_dyn_set_boundary_host(dyn_h, phis, tcorh, qcorh), 'sml_dyn_set_boundary_host')
  end subroutine

  ! stepone + nsteps leapfrog steps with the state resident on the device: ONE upload, ONE download per 6-hour window
  subroutine dyn_hip_window(nsteps)
    integer, intent(in) :: nsteps
    call sml_check(sml_dyn_set_state_host(dyn_h, vor, div, t, ps, tr), 'sml_dyn_set_state_host')
    call sml_check(sml_dyn_set_lradsw(dyn_h, merge(1_c_int, 0_c_int, lradsw)), 'sml_dyn_set_lradsw')
    call sml_check(sml_dyn_window(dyn_h, state_dev, 1_c_int, int(nsteps, c_int), delt, alph, rob, wil, c_null_ptr), 'sml_dyn_window')
    call sml_check(sml_dyn_get_state_host(dyn_h, vor, div, t, ps, tr), 'sml_dyn_get_state_host')
    if (nsteps > 0) lradsw = (mod(nsteps, 3) == 1)      ! what stloop leaves in the module flag (src/dyn_stloop.f90:39, nstrad = 3)
  end subroutine

end module speedy_dyn_hip

! ---- external subroutines with the reference's names (link in place of ini_impint.o, dyn_step.o, ini_stepone.o) ----
subroutine impint(dt, alph_in)
  use speedy_dyn_hip
  implicit none
  real(c_double), intent(in) :: dt, alph_in
  call sml_check(sml_dyn_impint(dyn_h, dt, alph_in), 'sml_dyn_impint')
end subroutine

subroutine step(j1, j2, dt, alph_in, rob_in, wil_in)
  ! one time step on the host arrays: upload, five launches, download (use dyn_hip_window to keep the state on the device)
  use speedy_dyn_hip
  implicit none
  integer, intent(in) :: j1, j2
  real(c_double), intent(in) :: dt, alph_in, rob_in, wil_in
  call sml_check(sml_dyn_set_state_host(dyn_h, vor, div, t, ps, tr), 'sml_dyn_set_state_host')
  call sml_check(sml_dyn_set_lradsw(dyn_h, merge(1_c_int, 0_c_int, lradsw)), 'sml_dyn_set_lradsw')
  call sml_check(sml_dyn_step(dyn_h, state_dev, int(j1, c_int), int(j2, c_int), dt, alph_in, rob_in, wil_in, c_null_ptr), 'sml_dyn_step')
  call sml_check(sml_dyn_get_state_host(dyn_h, vor, div, t, ps, tr), 'sml_dyn_get_state_host')
end subroutine

subroutine stepone
  ! src/ini_stepone.f90 for istart = 0 or 2, statement for statement
  use speedy_dyn_hip
  implicit none
  real(c_double) :: delth
  external :: impint, step
  delth = 0.5_c_double*delt
  call impint(delth, alph)
  call step(1, 1, delth, alph, rob, wil)
  call impint(delt, alph)
  call step(1, 2, delt, alph, rob, wil)
  call impint(delt2, alph)
end subroutine
