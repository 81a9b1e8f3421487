! TEST INFRASTRUCTURE ONLY -- C-callable harness over the reference's own column-physics routines (phy_convmf.f90, phy_lscond.f90,
! phy_shtorh.f90, phy_radiat.f90, phy_suflux.f90, phy_vdifsc.f90, ini_inphys.f90, ini_fordate.f90), compiled IN PLACE from /root/reference/src by
! oracle/build_ref.sh into oracle/_ref/libref_phy.so.  Nothing here restates physics: every routine below forwards to the
! reference.  All reals are 8 bytes (-fdefault-real-8, as the reference builds).  ngp = 96*48 columns, nlev = 8.
module ref_phy_driver
  use iso_c_binding
  use mod_atparam
  use mod_physcon
  use mod_radcon
  use mod_sflcon, only: forog
  implicit none
  integer, parameter :: ngp = ix*il, nlev = kx
contains

  subroutine refp_init(hsg, rlat) bind(C, name="refp_init")
    real(c_double), intent(in) :: hsg(0:nlev), rlat(il)
    real(c_double) :: ppl(nlev)
    ppl = 0.0
    call inphys(hsg, ppl, rlat)
    call radset
    ablco2_ref = ablco2
  end subroutine

  ! the reference's own fordate(0) (src/ini_fordate.f90, compiled in place): the per-window forcing set-up the hybrid re-runs every
  ! step through agcm_init -- surface albedos from snow depth / sea ice (:54-61), tcorh = spec(gamlat phis0) (:72-86) and
  ! qcorh = spec(refrh1 (q_sat(tref, 1) - q_sat(tsfc, psfc))) from the land / sea surface temperatures (:88-113).  Inputs are the
  ! module variables fordate reads; nothing is restated here.
  subroutine refp_fordate(tyear_, phis0_, fmask_l_, fmask_s_, stl_am_, sst_am_, alb0_, snowd_am_, sice_am_, tcorh_, qcorh_, &
                          snowc_, alb_l_, alb_s_, albsfc_) bind(C, name="refp_fordate")
    use mod_date, only: tyear, iyear
    use mod_surfcon, only: phis0, alb0
    use mod_cli_land, only: fmask_l
    use mod_cli_sea, only: fmask_s
    use mod_var_land, only: stl_am, snowd_am
    use mod_var_sea, only: sst_am, sice_am
    use mod_hdifcon, only: tcorh, qcorh
    real(c_double), value :: tyear_
    real(c_double), intent(in) :: phis0_(ix,il), fmask_l_(ix,il), fmask_s_(ix,il), stl_am_(ngp), sst_am_(ngp), alb0_(ix,il)
    real(c_double), intent(in) :: snowd_am_(ngp), sice_am_(ngp)
    real(c_double), intent(out) :: tcorh_(2,mx,nx), qcorh_(2,mx,nx), snowc_(ngp), alb_l_(ngp), alb_s_(ngp), albsfc_(ngp)
    logical, save :: spectral_ready = .false.
    if (.not. spectral_ready) then
      call inifft()
      call parmtr(6.371d6)
      spectral_ready = .true.
    end if
    tyear = tyear_; iyear = 1981
    phis0 = phis0_; alb0 = alb0_; fmask_l = fmask_l_; fmask_s = fmask_s_
    stl_am = stl_am_; snowd_am = snowd_am_; sst_am = sst_am_; sice_am = sice_am_
    call fordate(0)
    tcorh_(1,:,:) = real(tcorh); tcorh_(2,:,:) = aimag(tcorh)
    qcorh_(1,:,:) = real(qcorh); qcorh_(2,:,:) = aimag(qcorh)
    snowc_ = snowc; alb_l_ = alb_l; alb_s_ = alb_s; albsfc_ = albsfc
  end subroutine

  subroutine refp_set_surface(phi0, alb_l_, alb_s_, albsfc_, snowc_) bind(C, name="refp_set_surface")
    real(c_double), intent(in) :: phi0(ngp), alb_l_(ngp), alb_s_(ngp), albsfc_(ngp), snowc_(ngp)
    alb_l = alb_l_; alb_s = alb_s_; albsfc = albsfc_; snowc = snowc_
    call sflset(phi0)
  end subroutine

  subroutine refp_sol_oz(tyear) bind(C, name="refp_sol_oz")
    real(c_double), value :: tyear
    call sol_oz(tyear)
  end subroutine

  subroutine refp_get_fields(fsol_, ozone_, ozupp_, zenit_, stratz_, forog_, fband_, sig_, dsig_, sigh_, grdsig_, grdscp_, wvi_) &
      bind(C, name="refp_get_fields")
    real(c_double), intent(out) :: fsol_(ngp), ozone_(ngp), ozupp_(ngp), zenit_(ngp), stratz_(ngp), forog_(ngp), fband_(301,4)
    real(c_double), intent(out) :: sig_(nlev), dsig_(nlev), sigh_(0:nlev), grdsig_(nlev), grdscp_(nlev), wvi_(nlev,2)
    fsol_ = fsol; ozone_ = ozone; ozupp_ = ozupp; zenit_ = zenit; stratz_ = stratz; forog_ = forog; fband_ = fband
    sig_ = sig; dsig_ = dsig; sigh_ = sigh; grdsig_ = grdsig; grdscp_ = grdscp; wvi_ = wvi
  end subroutine

  subroutine refp_get_radstate(tau2_, stratc_, qcloud_) bind(C, name="refp_get_radstate")
    real(c_double), intent(out) :: tau2_(ngp,nlev,4), stratc_(ngp,2), qcloud_(ngp)
    tau2_ = tau2; stratc_ = stratc; qcloud_ = qcloud
  end subroutine

  subroutine refp_shtorh(imode, ta, ps, sigv, qa, rh, qsat) bind(C, name="refp_shtorh")
    integer(c_int), value :: imode
    real(c_double), value :: sigv
    real(c_double), intent(in) :: ta(ngp), ps(ngp)
    real(c_double), intent(inout) :: qa(ngp), rh(ngp), qsat(ngp)
    call shtorh(imode, ngp, ta, ps, sigv, qa, rh, qsat)
  end subroutine

  subroutine refp_convmf(psa, se, qa, qsat, itop, cbmf, precnv, dfse, dfqa) bind(C, name="refp_convmf")
    real(c_double), intent(in) :: psa(ngp), se(ngp,nlev), qa(ngp,nlev), qsat(ngp,nlev)
    integer(c_int), intent(inout) :: itop(ngp)
    real(c_double), intent(inout) :: cbmf(ngp), precnv(ngp), dfse(ngp,nlev), dfqa(ngp,nlev)
    call convmf(psa, se, qa, qsat, itop, cbmf, precnv, dfse, dfqa)
  end subroutine

  subroutine refp_lscond(psa, qa, qsat, itop, precls, dtlsc, dqlsc) bind(C, name="refp_lscond")
    real(c_double), intent(in) :: psa(ngp), qa(ngp,nlev), qsat(ngp,nlev)
    integer(c_int), intent(inout) :: itop(ngp)
    real(c_double), intent(inout) :: precls(ngp), dtlsc(ngp,nlev), dqlsc(ngp,nlev)
    call lscond(psa, qa, qsat, itop, precls, dtlsc, dqlsc)
  end subroutine

  subroutine refp_cloud(qa, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr) bind(C, name="refp_cloud")
    real(c_double), intent(in) :: qa(ngp,nlev), rh(ngp,nlev), precnv(ngp), precls(ngp), gse(ngp), fmask(ngp)
    integer(c_int), intent(inout) :: iptop(ngp), icltop(ngp)
    real(c_double), intent(inout) :: cloudc(ngp), clstr(ngp)
    call cloud(qa, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr)
  end subroutine

  subroutine refp_radsw(psa, qa, icltop, cloudc, clstr, fsfcd, fsfc, ftop, dfabs) bind(C, name="refp_radsw")
    real(c_double), intent(in) :: psa(ngp), qa(ngp,nlev), cloudc(ngp), clstr(ngp)
    integer(c_int), intent(in) :: icltop(ngp)
    real(c_double), intent(inout) :: fsfcd(ngp), fsfc(ngp), ftop(ngp), dfabs(ngp,nlev)
    call radsw(psa, qa, icltop, cloudc, clstr, fsfcd, fsfc, ftop, dfabs)
  end subroutine

  subroutine refp_radlw(imode, ta, ts, fsfcd, fsfcu, fsfc, ftop, dfabs) bind(C, name="refp_radlw")
    integer(c_int), value :: imode
    real(c_double), intent(in) :: ta(ngp,nlev), ts(ngp)
    real(c_double), intent(inout) :: fsfcd(ngp), fsfcu(ngp), fsfc(ngp), ftop(ngp), dfabs(ngp,nlev)
    call radlw(imode, ta, ts, fsfcd, fsfcu, fsfc, ftop, dfabs)
  end subroutine

  subroutine refp_suflux(psa, ua, va, ta, qa, rh, phi, phi0, fmask, tland, tsea, swav, ssrd, slrd, ustr, vstr, shf, evap, slru, &
                         hfluxn, tsfc, tskin, u0, v0, t0, q0, lfluxland) bind(C, name="refp_suflux")
    real(c_double), intent(in) :: ua(ngp,nlev), va(ngp,nlev), ta(ngp,nlev), qa(ngp,nlev), rh(ngp,nlev), phi(ngp,nlev)
    real(c_double), intent(in) :: phi0(ngp), fmask(ngp), tland(ngp), tsea(ngp), swav(ngp), ssrd(ngp), slrd(ngp)
    real(c_double), intent(inout) :: psa(ngp), ustr(ngp,3), vstr(ngp,3), shf(ngp,3), evap(ngp,3), slru(ngp,3), hfluxn(ngp,2)
    real(c_double), intent(inout) :: tsfc(ngp), tskin(ngp), u0(ngp), v0(ngp), t0(ngp), q0(ngp)
    integer(c_int), value :: lfluxland
    call suflux(psa, ua, va, ta, qa, rh, phi, phi0, fmask, tland, tsea, swav, ssrd, slrd, ustr, vstr, shf, evap, slru, hfluxn, &
                tsfc, tskin, u0, v0, t0, q0, lfluxland /= 0)
  end subroutine

  subroutine refp_vdifsc(ua, va, se, rh, qa, qsat, phi, icnv, ut, vt, tt, qt) bind(C, name="refp_vdifsc")
    real(c_double), intent(in) :: ua(ngp,nlev), va(ngp,nlev), se(ngp,nlev), rh(ngp,nlev), qa(ngp,nlev), qsat(ngp,nlev), phi(ngp,nlev)
    integer(c_int), intent(in) :: icnv(ngp)
    real(c_double), intent(inout) :: ut(ngp,nlev), vt(ngp,nlev), tt(ngp,nlev), qt(ngp,nlev)
    call vdifsc(ua, va, se, rh, qa, qsat, phi, icnv, ut, vt, tt, qt)
  end subroutine

end module
