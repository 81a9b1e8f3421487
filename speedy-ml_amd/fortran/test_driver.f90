! Fortran host test: a synthetic region-954-shaped reservoir is stepped (a) by the drop-in module on the MI355X and
! (b) by the reference's own statements written out in plain Fortran (COO loop, matmul, tanh, matmul, un-standardise);
! the spectral externals grid_/spec_ are exercised with the reference's F77 calling convention (array-element
! actual arguments, sequence association).  Exit status 0 = parity within tolerance.
program test_driver
  use iso_c_binding
  use speedyml_hip
  use test_percall
  use speedy_dyn_hip
  implicit none

  type(reservoir_type) :: res
  integer, parameter :: n = 1152, d = 576, n_model = 132, n_out = 136
  integer :: k, i, j, e, q, nfail, length
  real(kind=dp), allocatable :: x(:), x_ref(:), y(:), temp(:), x_aug(:), out_ref(:), input(:,:)
  integer(c_int), allocatable :: gidx(:)
  integer(c_int) :: nmap
  real(kind=dp) :: seed, errx, erro
  real(kind=dp) :: vorm(62,32,2), vorg(96,48), back(62,32), a
  integer :: kcos
  external :: parmtr, inifft, grid, spec, stepone
  complex(c_double_complex) :: vor0(mx,nx,kx,2), div0(mx,nx,kx,2), t0(mx,nx,kx,2), ps0(mx,nx,2), tr0(mx,nx,kx,2,ntr)
  complex(c_double_complex) :: vorA(mx,nx,kx,2), tA(mx,nx,kx,2), psA(mx,nx,2)
  real(kind=dp) :: fsg(kx), rgam
  integer :: kk, m2, n2
  integer, parameter :: mb = 48
  real(kind=dp), allocatable :: tstates(:,:), tmodel(:,:), ty(:,:), aug(:,:), c_host(:,:), b_host(:,:), resid(:,:)
  integer :: ib, naug

  if (sml_device_count() < 1) then
    print *, 'test_driver: no HIP device visible'
    stop 2
  end if
  nfail = 0
  seed = 0.314159_dp
  k = int((6.0_dp/6000.0_dp)*n*n)
  res%assigned_region = 954
  res%n = n; res%k = k; res%reservoir_numinputs = d
  res%chunk_size_speedy = n_model; res%chunk_size_prediction = n_out
  res%leakage = 1.0_dp
  allocate(res%rows(k), res%cols(k), res%vals(k), res%win(n,d), res%wout(n_out,n+n_model))
  allocate(res%feedback(d), res%local_model(n_model), res%outvec(n_out), res%mean(36), res%std(36), res%out_stat_idx(n_out))
  do e = 1, k
    res%rows(e) = 1 + int(rnd()*n); res%cols(e) = 1 + int(rnd()*n); res%vals(e) = rnd()*0.25_dp
  end do
  res%win = 0.0_dp
  q = n/d
  do i = 1, d
    do j = (i-1)*q+1, i*q
      res%win(j,i) = 0.5_dp*(2.0_dp*rnd()-1.0_dp)        ! src/mod_reservoir.f90:272-280
    end do
  end do
  do j = 1, n+n_model
    do i = 1, n_out
      res%wout(i,j) = 0.02_dp*(rnd()-0.5_dp)
    end do
  end do
  do i = 1, 36
    res%mean(i) = 2.0_dp*rnd()-1.0_dp; res%std(i) = 0.5_dp+1.5_dp*rnd()
  end do
  do i = 1, d
    res%feedback(i) = 2.0_dp*rnd()-1.0_dp
  end do
  do i = 1, n_model
    res%local_model(i) = 2.0_dp*rnd()-1.0_dp
  end do
  allocate(gidx(n_out))
  nmap = sml_domain_out_map(1152_c_int, 954_c_int, 1_c_int, 1_c_int, 0_c_int, 1_c_int, gidx, res%out_stat_idx, int(n_out, c_int))
  if (nmap /= n_out) then
    print *, 'sml_domain_out_map returned', nmap
    stop 3
  end if

  call mklsparse(res)

  ! ---- synchronize: 5 teacher-forced steps ----
  length = 5
  allocate(x(n), x_ref(n), y(n), temp(n), x_aug(n+n_model), out_ref(n_out), input(d,length))
  do j = 1, length
    do i = 1, d
      input(i,j) = 2.0_dp*rnd()-1.0_dp
    end do
  end do
  x = 0.0_dp; x_ref = 0.0_dp
  call synchronize(res, input, x, length)
  do j = 1, length
    call coo_mv(x_ref, y)
    temp = matmul(res%win, input(:,j))
    x_ref = (1.0_dp-res%leakage)*x_ref + res%leakage*tanh(y+temp)
  end do
  errx = maxval(abs(x - x_ref))
  print '(a,es10.3)', 'synchronize max|dx| = ', errx
  if (errx > 1.0e-12_dp) nfail = nfail + 1

  ! ---- predict: reference statements (src/mod_reservoir.f90:1444-1471) vs the drop-in ----
  x = x_ref
  call predict(res, x, res%local_model)
  call coo_mv(x_ref, y)
  temp = matmul(res%win, res%feedback)
  x_ref = (1.0_dp-res%leakage)*x_ref + res%leakage*tanh(y+temp)
  x_aug(1:n_model) = res%local_model
  x_aug(n_model+1:) = x_ref
  x_aug(n_model+2:n_model+n:2) = x_aug(n_model+2:n_model+n:2)**2
  out_ref = matmul(res%wout, x_aug)
  do i = 1, n_out
    out_ref(i) = out_ref(i)*res%std(res%out_stat_idx(i)+1)
    out_ref(i) = out_ref(i) + res%mean(res%out_stat_idx(i)+1)
  end do
  errx = maxval(abs(x - x_ref))
  erro = maxval(abs(res%outvec - out_ref))/maxval(abs(out_ref))
  print '(a,es10.3,a,es10.3)', 'predict max|dx| = ', errx, '   rel outvec err = ', erro
  if (errx > 1.0e-13_dp .or. erro > 1.0e-11_dp) nfail = nfail + 1

  ! ---- spectral externals with the reference's calling convention ----
  a = 6.371e6_dp
  call parmtr(a)
  call inifft()
  vorm = 0.0_dp
  do j = 1, 31
    do i = 1, 62
      if ((i-1)/2 + j - 1 <= 30 .and. i /= 2) vorm(i,j,1) = 2.0_dp*rnd()-1.0_dp
    end do
  end do
  kcos = 1
  call grid(vorm(1,1,1), vorg, kcos)            ! array-element actual argument, as src/dyn_grtend.f90:62
  call spec(vorg, back)
  errx = maxval(abs(back - vorm(:,:,1)))/maxval(abs(vorm(:,:,1)))
  print '(a,es10.3)', 'spec(grid(v)) rel err = ', errx
  if (errx > 1.0e-12_dp) nfail = nfail + 1

  ! ---- training: chunking_matmul (two batches) + fit_chunk_hybrid on the device against the normal equations formed with
  ! Fortran matmul on the host:  wout (C + reg) = B + prior  (src/mod_reservoir.f90:1261-1313) ----
  naug = n + n_model
  allocate(tstates(n,mb), tmodel(n_model,mb), ty(n_out,mb), aug(naug,mb), c_host(naug,naug), b_host(n_out,naug), resid(n_out,naug))
  c_host = 0.0_dp; b_host = 0.0_dp
  do ib = 1, 2
    do j = 1, mb
      do i = 1, n
        tstates(i,j) = 2.0_dp*rnd()-1.0_dp
      end do
      do i = 1, n_model
        tmodel(i,j) = 2.0_dp*rnd()-1.0_dp
      end do
      do i = 1, n_out
        ty(i,j) = 2.0_dp*rnd()-1.0_dp
      end do
    end do
    call chunking_matmul(res, tstates, tmodel, ty)
    aug(1:n_model,:) = tmodel
    aug(n_model+1:naug,:) = tstates
    c_host = c_host + matmul(aug, transpose(aug))
    b_host = b_host + matmul(ty, transpose(aug))
  end do
  res%beta_res = 0.001_dp; res%beta_model = 1.0_dp; res%prior_val = 0.5_dp; res%using_prior = .true.
  call fit_chunk_hybrid(res)
  do i = 1, naug                                             ! with a prior the betas enter squared (quirk Q8, :1271-1290)
    if (i <= n_model) then
      c_host(i,i) = c_host(i,i) + res%beta_model**2
    else
      c_host(i,i) = c_host(i,i) + res%beta_res**2
    end if
  end do
  do i = 1, n_model
    b_host(i,i) = b_host(i,i) + res%prior_val*res%beta_model**2
  end do
  resid = matmul(res%wout, c_host) - b_host
  erro = maxval(abs(resid))/maxval(abs(b_host))
  print '(a,es10.3)', 'training: max|wout (C+reg) - (B+prior)| / max|B| = ', erro
  if (.not. (erro < 1.0e-9_dp)) nfail = nfail + 1

  ! ---- SPEEDY adiabatic time stepping: the external subroutines impint/step/stepone (host arrays in the reference's
  ! shapes, one round trip per call) against the device-resident window; same kernels -> identical bits ----
  call dyn_hip_init(a)
  fsg = (/0.025_dp, 0.095_dp, 0.20_dp, 0.34_dp, 0.51_dp, 0.685_dp, 0.835_dp, 0.95_dp/)
  rgam = (2.0_dp/7.0_dp*1004.0_dp)*6.0_dp/(1000.0_dp*9.81_dp)
  vor = (0.0_dp,0.0_dp); div = (0.0_dp,0.0_dp); t = (0.0_dp,0.0_dp); ps = (0.0_dp,0.0_dp); tr = (0.0_dp,0.0_dp)
  phis = (0.0_dp,0.0_dp); tcorh = (0.0_dp,0.0_dp); qcorh = (0.0_dp,0.0_dp)
  do kk = 1, kx
    t(1,1,kk,1) = cmplx(288.0_dp*max(0.2_dp, fsg(kk))**rgam*sqrt(2.0_dp), 0.0_dp, kind=dp)
    do n2 = 1, 6
      do m2 = 1, 6
        if (m2 == 1 .and. n2 == 1) cycle
        vor(m2,n2,kk,1) = 2.0e-6_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
        div(m2,n2,kk,1) = 1.0e-7_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
        t(m2,n2,kk,1) = 0.5_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
        tr(m2,n2,kk,1,1) = 1.0e-4_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
      end do
    end do
  end do
  do n2 = 1, 6
    do m2 = 1, 6
      ps(m2,n2,1) = 2.0e-3_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
      phis(m2,n2) = 300.0_dp*cmplx(rnd()-0.5_dp, merge(0.0_dp, rnd()-0.5_dp, m2 == 1), kind=dp)
    end do
  end do
  tcorh = phis*(6.0_dp/(1000.0_dp*9.81_dp))
  vor(:,:,:,2) = vor(:,:,:,1); div(:,:,:,2) = div(:,:,:,1); t(:,:,:,2) = t(:,:,:,1); ps(:,:,2) = ps(:,:,1); tr(:,:,:,2,:) = tr(:,:,:,1,:)
  vor0 = vor; div0 = div; t0 = t; ps0 = ps; tr0 = tr
  call dyn_hip_boundary()
  call dyn_hip_window(0)                        ! stepone on the device-resident state
  vorA = vor; tA = t; psA = ps
  vor = vor0; div = div0; t = t0; ps = ps0; tr = tr0
  call stepone                                  ! src/ini_stepone.f90 through the external impint/step
  errx = maxval(abs(vor - vorA)) + maxval(abs(t - tA)) + maxval(abs(ps - psA))
  print '(a,es10.3)', 'stepone: host-array externals vs device-resident window, sum of max|diff| = ', errx
  if (errx /= 0.0_dp) nfail = nfail + 1
  if (ps(1,1,1) /= ps0(1,1,1) .or. ps(1,1,2) /= ps0(1,1,1)) then
    print *, 'global mean of log(ps) not conserved'
    nfail = nfail + 1
  end if
  if (maxval(abs(vor(:,:,:,2) - vor0(:,:,:,1))) == 0.0_dp .or. .not. (maxval(abs(t)) < 1.0e3_dp)) then
    print *, 'stepone left the state unchanged or produced non-finite values'
    nfail = nfail + 1
  end if
  call dyn_hip_window(24)                       ! a whole 6-hour hybrid window
  print '(a,es10.3,a,es10.3)', '6-hour window: max|vor| = ', maxval(abs(vor)), '   max|T| = ', maxval(abs(t))
  if (.not. (maxval(abs(vor)) < 1.0e-3_dp .and. maxval(abs(t)) < 1.0e3_dp)) nfail = nfail + 1

  ! ---- the same with the column physics attached (grtend's phypar call): externals vs device-resident window, identical bits ----
  block
    real(kind=dp) :: hsg(9), radang(48), sfc(96,48,9), lat
    integer :: jl, il2
    hsg = (/0.0_dp, 0.05_dp, 0.14_dp, 0.26_dp, 0.42_dp, 0.60_dp, 0.77_dp, 0.90_dp, 1.0_dp/)
    do jl = 1, 48
      radang(jl) = (-87.159_dp + (jl - 1)*(2.0_dp*87.159_dp/47.0_dp))*3.14159265358979_dp/180.0_dp
    end do
    call dyn_hip_physics_init(hsg, radang, 3)
    do jl = 1, 48
      lat = radang(jl)
      do il2 = 1, 96
        sfc(il2,jl,1) = merge(1.0_dp, 0.0_dp, mod(il2/12 + jl/8, 3) == 0)       ! land fraction
        sfc(il2,jl,2) = 0.0_dp                                                  ! phis0
        sfc(il2,jl,3) = 288.0_dp - 30.0_dp*sin(lat)**2                          ! land temperature
        sfc(il2,jl,4) = max(272.0_dp, 300.0_dp - 30.0_dp*sin(lat)**2)           ! sea temperature
        sfc(il2,jl,5) = 0.5_dp                                                  ! soil wetness
        sfc(il2,jl,6) = 0.2_dp; sfc(il2,jl,7) = 0.07_dp                         ! land / sea albedo
        sfc(il2,jl,8) = 0.07_dp + sfc(il2,jl,1)*0.13_dp; sfc(il2,jl,9) = 0.0_dp ! surface albedo, snow cover
      end do
    end do
    call dyn_hip_surface(sfc(:,:,1), sfc(:,:,2), sfc(:,:,3), sfc(:,:,4), sfc(:,:,5), sfc(:,:,6), sfc(:,:,7), sfc(:,:,8), sfc(:,:,9), 0.37_dp)
    do kk = 1, kx           ! some moisture so that condensation and convection have something to do
      tr0(1,1,kk,:,1) = cmplx(8.0_dp*fsg(kk)**3*sqrt(2.0_dp), 0.0_dp, kind=dp)
    end do
    vor = vor0; div = div0; t = t0; ps = ps0; tr = tr0
    lradsw = .true.
    call dyn_hip_window(0)
    vorA = vor; tA = t; psA = ps
    vor = vor0; div = div0; t = t0; ps = ps0; tr = tr0
    lradsw = .true.
    call stepone
    errx = maxval(abs(vor - vorA)) + maxval(abs(t - tA)) + maxval(abs(ps - psA))
    print '(a,es10.3)', 'stepone with column physics: externals vs device-resident window, sum of max|diff| = ', errx
    if (errx /= 0.0_dp) nfail = nfail + 1
    vorA = vor; tA = t
    lradsw = .true.
    call dyn_hip_window(24)
    print '(a,es10.3,a,es10.3)', '6-hour window with physics: max|vor| = ', maxval(abs(vor)), '   max|T| = ', maxval(abs(t))
    if (.not. (maxval(abs(vor)) < 1.0e-3_dp .and. maxval(abs(t)) < 1.0e3_dp)) nfail = nfail + 1
  end block

  if (nfail == 0) then
    print *, 'FORTRAN HOST PARITY OK'
  else
    print *, 'FORTRAN HOST PARITY FAILED', nfail
    stop 1
  end if

contains

  function rnd() result(r)          ! small deterministic generator (the compiler RNG is not reproducible, SURVEY H5)
    real(kind=dp) :: r
    seed = seed*997.0_dp + 0.1234567_dp
    seed = seed - int(seed)
    r = seed
  end function

  subroutine coo_mv(xin, yout)      ! MKL_SPARSE_D_MV semantics: y = A x, COO in storage order
    real(kind=dp), intent(in) :: xin(:)
    real(kind=dp), intent(out) :: yout(:)
    integer :: ee
    yout = 0.0_dp
    do ee = 1, res%k
      yout(res%rows(ee)) = yout(res%rows(ee)) + res%vals(ee)*xin(res%cols(ee))
    end do
  end subroutine

end program test_driver
