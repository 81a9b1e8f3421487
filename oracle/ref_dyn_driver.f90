! TEST INFRASTRUCTURE ONLY (oracle/_ref harness) -- never linked into the product.
!
! C-callable harness around the *compiled reference* SPEEDY dynamical-core routines that have no dependency outside
! the reference's own pure-Fortran sources (compiled in place by oracle/build_ref.sh):
!   tables : indyns (src/ini_indyns.f90), impint (src/ini_impint.f90) + inv/ludcmp/lubksb (src/spe_matinv.f90)
!   kernels: geop (src/dyn_geop.f90), sptend (src/dyn_sptend.f90), implic (src/dyn_implic.f90),
!            hordif, timint (src/dyn_step.f90)
! src/dyn_step.f90 also defines step(), which calls grtend -> phypar (column physics: out of scope and not built).
! Those two references are left WEAK/undefined by build_ref.sh and are never called; nothing stands in for them.
! Complex spectral arrays cross the boundary as interleaved (re,im) doubles = Fortran complex(8) storage.
module ref_dyn_driver
  use iso_c_binding
  use mod_atparam
  use mod_dynvar
  use mod_dyncon1
  use mod_dyncon2
  use mod_hdifcon
  use mod_tsteps, only: alph
  implicit none
contains

  subroutine refd_init() bind(C, name="refd_init")
    call inifft()
    call indyns()          ! sets alph = 0.5, calls parmtr(rearth)
    vor = (0.,0.); div = (0.,0.); t = (0.,0.); ps = (0.,0.); tr = (0.,0.); phi = (0.,0.); phis = (0.,0.)
    tcorh = (0.,0.); qcorh = (0.,0.)
  end subroutine

  subroutine refd_impint(dt, a_) bind(C, name="refd_impint")
    real(c_double), value :: dt, a_
    call impint(dt, a_)
  end subroutine

  subroutine refd_get(which, out, n) bind(C, name="refd_get")
    integer(c_int), value :: which, n
    real(c_double), intent(out) :: out(n)
    select case (which)
    case (1);  out(1:kxp) = hsg
    case (2);  out(1:kx) = dhs
    case (3);  out(1:kx) = fsg
    case (4);  out(1:kx) = dhsr
    case (5);  out(1:kx) = fsgr
    case (6);  out(1:il) = coriol
    case (7);  out(1:kx) = xgeop1
    case (8);  out(1:kx) = xgeop2
    case (9);  out(1:mx*nx) = reshape(dmp, (/mx*nx/))
    case (10); out(1:mx*nx) = reshape(dmpd, (/mx*nx/))
    case (11); out(1:mx*nx) = reshape(dmps, (/mx*nx/))
    case (12); out(1:mx*nx) = reshape(dmp1, (/mx*nx/))
    case (13); out(1:mx*nx) = reshape(dmp1d, (/mx*nx/))
    case (14); out(1:mx*nx) = reshape(dmp1s, (/mx*nx/))
    case (15); out(1:kx) = tcorv
    case (16); out(1:kx) = qcorv
    case (17); out(1:kx) = tref
    case (18); out(1:kx) = tref1
    case (19); out(1:kx) = tref2
    case (20); out(1:kx) = tref3
    case (21); out(1:kx*kx) = reshape(xc, (/kx*kx/))
    case (22); out(1:kx*kx) = reshape(xd, (/kx*kx/))
    case (23); out(1:kx*kx*lmax) = reshape(xj, (/kx*kx*lmax/))
    case (24); out(1:kx) = dhsx
    case (25); out(1:mx*nx) = reshape(elz, (/mx*nx/))
    case (26); out(1) = alph
    end select
  end subroutine

  ! state <-> flat interleaved arrays: vor/div/t/tr (2,mx,nx,kx,2), ps (2,mx,nx,2), phis (2,mx,nx)
  subroutine refd_set_state(vor_, div_, t_, ps_, tr_, phis_, tcorh_, qcorh_) bind(C, name="refd_set_state")
    real(c_double), intent(in) :: vor_(2,mx,nx,kx,2), div_(2,mx,nx,kx,2), t_(2,mx,nx,kx,2), ps_(2,mx,nx,2), tr_(2,mx,nx,kx,2)
    real(c_double), intent(in) :: phis_(2,mx,nx), tcorh_(2,mx,nx), qcorh_(2,mx,nx)
    vor = cmplx(vor_(1,:,:,:,:), vor_(2,:,:,:,:), kind=8)
    div = cmplx(div_(1,:,:,:,:), div_(2,:,:,:,:), kind=8)
    t = cmplx(t_(1,:,:,:,:), t_(2,:,:,:,:), kind=8)
    ps = cmplx(ps_(1,:,:,:), ps_(2,:,:,:), kind=8)
    tr(:,:,:,:,1) = cmplx(tr_(1,:,:,:,:), tr_(2,:,:,:,:), kind=8)
    phis = cmplx(phis_(1,:,:), phis_(2,:,:), kind=8)
    tcorh = cmplx(tcorh_(1,:,:), tcorh_(2,:,:), kind=8)
    qcorh = cmplx(qcorh_(1,:,:), qcorh_(2,:,:), kind=8)
  end subroutine

  subroutine refd_get_state(vor_, div_, t_, ps_, tr_, phi_) bind(C, name="refd_get_state")
    real(c_double), intent(out) :: vor_(2,mx,nx,kx,2), div_(2,mx,nx,kx,2), t_(2,mx,nx,kx,2), ps_(2,mx,nx,2), tr_(2,mx,nx,kx,2)
    real(c_double), intent(out) :: phi_(2,mx,nx,kx)
    vor_(1,:,:,:,:) = real(vor);  vor_(2,:,:,:,:) = aimag(vor)
    div_(1,:,:,:,:) = real(div);  div_(2,:,:,:,:) = aimag(div)
    t_(1,:,:,:,:) = real(t);      t_(2,:,:,:,:) = aimag(t)
    ps_(1,:,:,:) = real(ps);      ps_(2,:,:,:) = aimag(ps)
    tr_(1,:,:,:,:) = real(tr(:,:,:,:,1)); tr_(2,:,:,:,:) = aimag(tr(:,:,:,:,1))
    phi_(1,:,:,:) = real(phi);    phi_(2,:,:,:) = aimag(phi)
  end subroutine

  subroutine refd_geop(jj) bind(C, name="refd_geop")
    integer(c_int), value :: jj
    call geop(jj)
  end subroutine

  ! tendencies as interleaved arrays (2,mx,nx,kx) / (2,mx,nx), in/out
  subroutine refd_sptend(divdt_, tdt_, psdt_, j4) bind(C, name="refd_sptend")
    real(c_double), intent(inout) :: divdt_(2,mx,nx,kx), tdt_(2,mx,nx,kx), psdt_(2,mx,nx)
    integer(c_int), value :: j4
    complex(8) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
    divdt = cmplx(divdt_(1,:,:,:), divdt_(2,:,:,:), kind=8); tdt = cmplx(tdt_(1,:,:,:), tdt_(2,:,:,:), kind=8)
    psdt = cmplx(psdt_(1,:,:), psdt_(2,:,:), kind=8)
    call sptend(divdt, tdt, psdt, j4)
    divdt_(1,:,:,:) = real(divdt); divdt_(2,:,:,:) = aimag(divdt)
    tdt_(1,:,:,:) = real(tdt); tdt_(2,:,:,:) = aimag(tdt)
    psdt_(1,:,:) = real(psdt); psdt_(2,:,:) = aimag(psdt)
  end subroutine

  subroutine refd_implic(divdt_, tdt_, psdt_) bind(C, name="refd_implic")
    real(c_double), intent(inout) :: divdt_(2,mx,nx,kx), tdt_(2,mx,nx,kx), psdt_(2,mx,nx)
    complex(8) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
    divdt = cmplx(divdt_(1,:,:,:), divdt_(2,:,:,:), kind=8); tdt = cmplx(tdt_(1,:,:,:), tdt_(2,:,:,:), kind=8)
    psdt = cmplx(psdt_(1,:,:), psdt_(2,:,:), kind=8)
    call implic(divdt, tdt, psdt)
    divdt_(1,:,:,:) = real(divdt); divdt_(2,:,:,:) = aimag(divdt)
    tdt_(1,:,:,:) = real(tdt); tdt_(2,:,:,:) = aimag(tdt)
    psdt_(1,:,:) = real(psdt); psdt_(2,:,:) = aimag(psdt)
  end subroutine

  ! hordif(nlev, field, fdt, dmp, dmp1) with the coefficient pair chosen by `which`: 1 dmp/dmp1, 2 dmpd/dmp1d, 3 dmps/dmp1s
  subroutine refd_hordif(nlev, field_, fdt_, which) bind(C, name="refd_hordif")
    integer(c_int), value :: nlev, which
    real(c_double), intent(in) :: field_(2,mx,nx,kx)
    real(c_double), intent(inout) :: fdt_(2,mx,nx,kx)
    complex(8) :: field(mx,nx,kx), fdt(mx,nx,kx)
    field = cmplx(field_(1,:,:,:), field_(2,:,:,:), kind=8); fdt = cmplx(fdt_(1,:,:,:), fdt_(2,:,:,:), kind=8)
    if (which == 1) call hordif(nlev, field, fdt, dmp, dmp1)
    if (which == 2) call hordif(nlev, field, fdt, dmpd, dmp1d)
    if (which == 3) call hordif(nlev, field, fdt, dmps, dmp1s)
    fdt_(1,:,:,:) = real(fdt); fdt_(2,:,:,:) = aimag(fdt)
  end subroutine

  subroutine refd_timint(j1, dt, eps, wil_, nlev, field_, fdt_) bind(C, name="refd_timint")
    integer(c_int), value :: j1, nlev
    real(c_double), value :: dt, eps, wil_
    real(c_double), intent(inout) :: field_(2,mx,nx,nlev,2), fdt_(2,mx,nx,nlev)
    complex(8) :: field(mx,nx,nlev,2), fdt(mx,nx,nlev)
    field = cmplx(field_(1,:,:,:,:), field_(2,:,:,:,:), kind=8); fdt = cmplx(fdt_(1,:,:,:), fdt_(2,:,:,:), kind=8)
    call timint(j1, dt, eps, wil_, nlev, field, fdt)
    field_(1,:,:,:,:) = real(field); field_(2,:,:,:,:) = aimag(field)
    fdt_(1,:,:,:) = real(fdt); fdt_(2,:,:,:) = aimag(fdt)
  end subroutine

end module
