! TEST INFRASTRUCTURE ONLY (oracle/_ref harness) -- never linked into the product.
!
! C-callable harness around the *compiled reference* spectral core.  The reference
! sources are compiled where they lie under /root/reference/src (see build_ref.sh):
!   mod_atparam.f90, mod_spectral.f90, mod_fft.f90, spe_spectral.f90,
!   spe_subfft_fftpack.f90 (which includes spe_subfft_fftpack2.f90)
! with `amdflang -fdefault-real-8` (the reference's own -r8 promotion, src/makefile:6,12).
! Nothing here re-implements reference arithmetic: every routine below only
! forwards to the reference's external F77-style subroutines or copies module data out.
module ref_spectral_driver
  use iso_c_binding
  use mod_atparam
  use mod_spectral
  use mod_fft, only: wsave
  implicit none
contains

  subroutine ref_init(a) bind(C, name="ref_init")
    real(c_double), value :: a
    call inifft()
    call parmtr(a)
  end subroutine

  ! copy a named table out (flat, Fortran order).  which: see tests/_ref.py
  subroutine ref_get_table(which, out, n) bind(C, name="ref_get_table")
    integer(c_int), value :: which, n
    real(c_double), intent(out) :: out(n)
    select case (which)
    case (1);  out(1:iy) = sia
    case (2);  out(1:iy) = coa
    case (3);  out(1:iy) = wt
    case (4);  out(1:iy) = wght
    case (5);  out(1:il) = cosg
    case (6);  out(1:il) = cosgr
    case (7);  out(1:il) = cosgr2
    case (8);  out(1:mx*nx) = reshape(el2, (/mx*nx/))
    case (9);  out(1:mx*nx) = reshape(elm2, (/mx*nx/))
    case (10); out(1:mx*nx) = reshape(el4, (/mx*nx/))
    case (11); out(1:mx*nx) = reshape(trfilt, (/mx*nx/))
    case (12); out(1:nx) = real(nsh2, c_double)
    case (13); out(1:mxp*nxp) = reshape(epsi, (/mxp*nxp/))
    case (14); out(1:mxp*nxp) = reshape(repsi, (/mxp*nxp/))
    case (15); out(1:mxp) = consq
    case (16); out(1:mx) = gradx
    case (17); out(1:mx*nx) = reshape(gradym, (/mx*nx/))
    case (18); out(1:mx*nx) = reshape(gradyp, (/mx*nx/))
    case (19); out(1:mx*nx) = reshape(uvdx, (/mx*nx/))
    case (20); out(1:mx*nx) = reshape(uvdym, (/mx*nx/))
    case (21); out(1:mx*nx) = reshape(uvdyp, (/mx*nx/))
    case (22); out(1:mx*nx) = reshape(vddym, (/mx*nx/))
    case (23); out(1:mx*nx) = reshape(vddyp, (/mx*nx/))
    case (24); out(1:mx2*nx*iy) = reshape(cpol, (/mx2*nx*iy/))
    case (25); out(1:2*ix+15) = wsave
    case (26); out(1) = sqrhlf
    end select
  end subroutine

  subroutine ref_grid(vorm, vorg, kcos) bind(C, name="ref_grid")
    real(c_double), intent(inout) :: vorm(mx2,nx), vorg(ix,il)
    integer(c_int), value :: kcos
    call grid(vorm, vorg, kcos)
  end subroutine

  subroutine ref_spec(vorg, vorm) bind(C, name="ref_spec")
    real(c_double), intent(inout) :: vorg(ix,il), vorm(mx2,nx)
    call spec(vorg, vorm)
  end subroutine

  subroutine ref_vdspec(ug, vg, vorm, divm, kcos) bind(C, name="ref_vdspec")
    real(c_double), intent(inout) :: ug(ix,il), vg(ix,il), vorm(mx2,nx), divm(mx2,nx)
    integer(c_int), value :: kcos
    call vdspec(ug, vg, vorm, divm, kcos)
  end subroutine

  subroutine ref_gridy(v, varm) bind(C, name="ref_gridy")
    real(c_double), intent(inout) :: v(mx2,nx), varm(mx2,il)
    call gridy(v, varm)
  end subroutine

  subroutine ref_gridx(varm, vorg, kcos) bind(C, name="ref_gridx")
    real(c_double), intent(inout) :: varm(mx2,il), vorg(ix,il)
    integer(c_int), value :: kcos
    call gridx(varm, vorg, kcos)
  end subroutine

  subroutine ref_specx(vorg, varm) bind(C, name="ref_specx")
    real(c_double), intent(inout) :: vorg(ix,il), varm(mx2,il)
    call specx(vorg, varm)
  end subroutine

  subroutine ref_specy(varm, vorm) bind(C, name="ref_specy")
    real(c_double), intent(inout) :: varm(mx2,il), vorm(mx2,nx)
    call specy(varm, vorm)
  end subroutine

  subroutine ref_uvspec(vorm, divm, ucosm, vcosm) bind(C, name="ref_uvspec")
    real(c_double), intent(inout) :: vorm(mx2,nx), divm(mx2,nx), ucosm(mx2,nx), vcosm(mx2,nx)
    call uvspec(vorm, divm, ucosm, vcosm)
  end subroutine

  subroutine ref_vds(ucosm, vcosm, vorm, divm) bind(C, name="ref_vds")
    real(c_double), intent(inout) :: ucosm(mx2,nx), vcosm(mx2,nx), vorm(mx2,nx), divm(mx2,nx)
    call vds(ucosm, vcosm, vorm, divm)
  end subroutine

  subroutine ref_grad(psi, psdx, psdy) bind(C, name="ref_grad")
    real(c_double), intent(inout) :: psi(mx2,nx), psdx(mx2,nx), psdy(mx2,nx)
    call grad(psi, psdx, psdy)
  end subroutine

  subroutine ref_lap(strm, vorm) bind(C, name="ref_lap")
    real(c_double), intent(inout) :: strm(mx2,nx), vorm(mx2,nx)
    call lap(strm, vorm)
  end subroutine

  subroutine ref_invlap(vorm, strm) bind(C, name="ref_invlap")
    real(c_double), intent(inout) :: vorm(mx2,nx), strm(mx2,nx)
    call invlap(vorm, strm)
  end subroutine

  subroutine ref_trunct(vor) bind(C, name="ref_trunct")
    real(c_double), intent(inout) :: vor(mx2,nx)
    call trunct(vor)
  end subroutine

  subroutine ref_rfftf(r) bind(C, name="ref_rfftf")
    real(c_double), intent(inout) :: r(ix)
    call rfftf(ix, r, wsave)
  end subroutine

  subroutine ref_rfftb(r) bind(C, name="ref_rfftb")
    real(c_double), intent(inout) :: r(ix)
    call rfftb(ix, r, wsave)
  end subroutine

end module
