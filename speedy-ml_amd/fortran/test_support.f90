! Test support for the module-API drop-ins: synthetic stand-ins of exactly the reference procedures the drop-ins call for DATA --
!   speedyml_data_source :: source_read_era, source_read_model_states -- what speedy_res_interface::read_era / read_model_states
!                           (src/speedy_res_interface.f90:439-723: ERA5 / SPEEDY NetCDF readers) forward to in the drop-in
!   mod_io               :: read_trained_res, read_trained_ocean_res, read_3d_file_parallel, write_netcdf_2d_non_met_data,
!                           write_netcdf_1d_non_met_data_int / _real (src/mod_io.f90)
! -- with the reference's argument lists, plus hybrid_boundary_fields, the one procedure a host adds to hand SPEEDY's boundary
! arrays to the engine (INTEGRATION.md).  A maintainer links the reference's modules instead of this file.  The fields are
! deterministic, smooth and ERA5-shaped (SURVEY 8d); nothing here is part of the product.
module speedyml_data_source
  use iso_c_binding
  use mod_utilities, only : dp, reservoir_type, grid_type, model_parameters_type, era_data_type, speedy_data_type, xgrid, ygrid, zgrid
  use mod_calendar
  implicit none
  real(kind=dp), parameter :: pi = 3.14159265358979323846_dp
  real(kind=dp), parameter :: sig(8) = [0.025_dp, 0.095_dp, 0.20_dp, 0.34_dp, 0.51_dp, 0.685_dp, 0.835_dp, 0.95_dp]
  real(kind=dp), parameter :: latd(48) = [ -87.159_dp, -83.479_dp, -79.777_dp, -76.070_dp, -72.362_dp, -68.652_dp, -64.942_dp, &
      -61.232_dp, -57.521_dp, -53.810_dp, -50.099_dp, -46.389_dp, -42.678_dp, -38.967_dp, -35.256_dp, -31.545_dp, -27.833_dp, -24.122_dp, &
      -20.411_dp, -16.700_dp, -12.989_dp, -9.278_dp, -5.567_dp, -1.856_dp, 1.856_dp, 5.567_dp, 9.278_dp, 12.989_dp, 16.700_dp, 20.411_dp, &
      24.122_dp, 27.833_dp, 31.545_dp, 35.256_dp, 38.967_dp, 42.678_dp, 46.389_dp, 50.099_dp, 53.810_dp, 57.521_dp, 61.232_dp, 64.942_dp, &
      68.652_dp, 72.362_dp, 76.070_dp, 79.777_dp, 83.479_dp, 87.159_dp ]
contains

  pure function wrapx(ix) result(x)
    integer, intent(in) :: ix
    integer :: x
    x = modulo(ix - 1, xgrid) + 1
  end function

  ! value of variable v (1 T, 2 u, 3 v, 4 q [kg/kg]) at global point (x,y,z), hour h of the year; bias > 0: the imperfect model
  pure function field3d(v, x, y, z, h, bias) result(f)
    integer, intent(in) :: v, x, y, z, h
    real(kind=dp), intent(in) :: bias
    real(kind=dp) :: f, lat, lon, wave
    lat = latd(y) * pi / 180.0_dp; lon = (x - 1) * 2.0_dp * pi / xgrid
    wave = sin(2.0_dp * pi * h / 240.0_dp + 3.0_dp * lon) * cos(lat)
    select case (v)
    case (1); f = max(216.0_dp, 288.0_dp * sig(z)**0.19_dp) - 22.0_dp * sin(lat)**2 * sig(z) + 2.0_dp * wave + bias
    case (2); f = 25.0_dp * sin(2.0_dp * lat)**2 * (1.0_dp - sig(z)) + 4.0_dp * wave + bias
    case (3); f = 3.0_dp * cos(2.0_dp * pi * h / 240.0_dp + 3.0_dp * lon) * cos(lat) + 0.5_dp * bias
    case default; f = max(1.0e-9_dp, 0.012_dp * sig(z)**3 * exp(-(latd(y) / 40.0_dp)**2) * (1.0_dp + 0.1_dp * wave) * (1.0_dp + 0.02_dp * bias))
    end select
  end function

  pure function field2d(which, x, y, h) result(f)          ! 1 logp, 2 tisr, 3 sst, 4 precip (hourly)
    integer, intent(in) :: which, x, y, h
    real(kind=dp) :: f, lat, lon, decl, cosz
    lat = latd(y) * pi / 180.0_dp; lon = (x - 1) * 2.0_dp * pi / xgrid
    select case (which)
    case (1); f = 0.01_dp * sin(lon) * cos(lat) - 0.02_dp * sin(lat)**2 + 0.003_dp * sin(2.0_dp * pi * h / 240.0_dp + 2.0_dp * lon)
    case (2)
      decl = 23.44_dp * pi / 180.0_dp * sin(2.0_dp * pi * (h / 24.0_dp - 80.0_dp) / 365.0_dp)
      cosz = sin(lat) * sin(decl) + cos(lat) * cos(decl) * cos(2.0_dp * pi * h / 24.0_dp + lon - pi)
      f = max(0.0_dp, cosz) * 1361.0_dp * 3600.0_dp
    case (3); f = max(272.0_dp, 300.0_dp - 30.0_dp * sin(lat)**2 + 0.5_dp * sin(lon * 2.0_dp)) + merge(0.4_dp * sin(2.0_dp * pi * h / 8760.0_dp), 0.0_dp, abs(latd(y)) < 60.0_dp)
    case default; f = 1.0e-4_dp * (1.0_dp + sin(2.0_dp * pi * h / 96.0_dp + 5.0_dp * lon))**2 * cos(lat)**2
    end select
  end function

  ! hours at the end of the returned window that carry data (SML_TEST_ERA_HOURS; 2300 covers the 1440-hour training window of
  ! test_train_batch, a forecast needs its synchronisation window only)
  integer function filled_hours()
    character(len=16) :: v
    integer :: n, st
    filled_hours = 2300
    call get_environment_variable('SML_TEST_ERA_HOURS', v, n, st)
    if (st == 0 .and. n > 0) read(v(1:n), *) filled_hours
  end function

  subroutine hours_covered(start_year, end_year, nh)
    integer, intent(in) :: start_year, end_year
    integer, intent(out) :: nh
    integer :: into, whole
    ! the reference's readers return whole calendar years (leap years with 8784 hours); the stand-in stops just after the last hour
    ! the caller can index
    call numof_hours_into_year(calendar%currentyear, calendar%currentmonth, calendar%currentday, calendar%currenthour, into)
    whole = 0
    if (end_year > start_year) call numof_hours(start_year, end_year - 1, whole)
    nh = whole + into + 8
  end subroutine

  ! read_era (src/speedy_res_interface.f90): the region's INPUT patch, hourly, periodic in x
  subroutine source_read_era(reservoir, grid, model_parameters, start_year, end_year, era_data, timestep_arg)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer, intent(in) :: start_year, end_year
    type(era_data_type), intent(inout) :: era_data
    integer, intent(in), optional :: timestep_arg
    integer :: nh, ix, iy, iz, v, h, x, y, h0, hfirst
    call hours_covered(start_year, end_year, nh)
    h0 = 0
    hfirst = max(1, nh - filled_hours())                         ! (earlier hours stay zero: no caller of this test reads them)
    allocate(era_data%eravariables(4, grid%inputxchunk, grid%inputychunk, grid%inputzchunk, nh), era_data%era_logp(grid%inputxchunk, grid%inputychunk, nh), &
             era_data%era_tisr(grid%inputxchunk, grid%inputychunk, nh), era_data%era_sst(grid%inputxchunk, grid%inputychunk, nh), &
             era_data%era_precip(grid%inputxchunk, grid%inputychunk, nh))
    era_data%eravariables = 0.0_dp; era_data%era_logp = 0.0_dp; era_data%era_tisr = 0.0_dp; era_data%era_sst = 0.0_dp; era_data%era_precip = 0.0_dp
    do h = hfirst, nh
      do iy = 1, grid%inputychunk
        y = grid%input_ystart + iy - 1
        do ix = 1, grid%inputxchunk
          x = wrapx(grid%input_xstart + ix - 1)
          do iz = 1, grid%inputzchunk
            do v = 1, 4
              era_data%eravariables(v, ix, iy, iz, h) = field3d(v, x, y, grid%input_zstart + iz - 1, h0 + h, 0.0_dp)
            end do
          end do
          era_data%era_logp(ix, iy, h) = field2d(1, x, y, h0 + h)
          era_data%era_tisr(ix, iy, h) = field2d(2, x, y, h0 + h)
          era_data%era_sst(ix, iy, h) = field2d(3, x, y, h0 + h)
          era_data%era_precip(ix, iy, h) = field2d(4, x, y, h0 + h)
        end do
      end do
    end do
  end subroutine

  ! read_model_states: SPEEDY's forecast of the region's RES patch (an imperfect copy of the truth)
  subroutine source_read_model_states(reservoir, grid, model_parameters, start_year, end_year, speedy_data, timestep_arg)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer, intent(in) :: start_year, end_year
    type(speedy_data_type), intent(inout) :: speedy_data
    integer, intent(in), optional :: timestep_arg
    integer :: nh, ix, iy, iz, v, h
    call hours_covered(start_year, end_year, nh)
    allocate(speedy_data%speedyvariables(4, grid%resxchunk, grid%resychunk, grid%reszchunk, nh), speedy_data%speedy_logp(grid%resxchunk, grid%resychunk, nh))
    speedy_data%speedyvariables = 0.0_dp; speedy_data%speedy_logp = 0.0_dp
    do h = max(1, nh - filled_hours()), nh
      do iy = 1, grid%resychunk
        do ix = 1, grid%resxchunk
          do iz = 1, grid%reszchunk
            do v = 1, 4
              speedy_data%speedyvariables(v, ix, iy, iz, h) = field3d(v, grid%res_xstart + ix - 1, grid%res_ystart + iy - 1, grid%res_zstart + iz - 1, h, 0.6_dp)
            end do
            if (speedy_data%speedyvariables(4, ix, iy, iz, h) > 0) speedy_data%speedyvariables(4, ix, iy, iz, h) = speedy_data%speedyvariables(4, ix, iy, iz, h) * 1000.0_dp
          end do
          speedy_data%speedy_logp(ix, iy, h) = field2d(1, grid%res_xstart + ix - 1, grid%res_ystart + iy - 1, h) + 0.001_dp
        end do
      end do
    end do
  end subroutine

  ! SPEEDY's boundary data and the hybrid's start state for the engine (what agcm_init / the boundary files give the reference):
  ! g = grid4d(4,96,48,8) | logp | precip | sst | tisr at the first prediction hour; phi0 [m2/s2]; the TISR table of an hourly year
  subroutine hybrid_boundary_fields(model_parameters, g, phi0, tisr, hsg, radang, fmask, tland, swav, alb_l, alb_s, albsfc, snowc)
    type(model_parameters_type), intent(in) :: model_parameters
    real(kind=dp), intent(out) :: g(:), phi0(:,:), tisr(:,:,:), hsg(9), radang(48), fmask(:,:), tland(:,:), swav(:,:), alb_l(:,:), alb_s(:,:), albsfc(:,:), snowc(:,:)
    integer :: x, y, z, v, h, k
    real(kind=dp) :: lat, lon
    hsg = [0.0_dp, 0.05_dp, 0.14_dp, 0.26_dp, 0.42_dp, 0.60_dp, 0.77_dp, 0.90_dp, 1.0_dp]
    radang = latd * pi / 180.0_dp
    h = model_parameters%traininglength + model_parameters%synclength
    k = 0
    do z = 1, zgrid
      do y = 1, ygrid
        do x = 1, xgrid
          do v = 1, 4
            k = k + 1
            g(k) = field3d(v, x, y, z, h, 0.0_dp)
            if (v == 4) g(k) = g(k) * 1000.0_dp
          end do
        end do
      end do
    end do
    do y = 1, ygrid
      do x = 1, xgrid
        lat = latd(y) * pi / 180.0_dp; lon = (x - 1) * 2.0_dp * pi / xgrid
        g(147456 + (y-1)*xgrid + x) = field2d(1, x, y, h)
        g(152064 + (y-1)*xgrid + x) = log(1.0_dp + 6.0_dp * field2d(4, x, y, h) / 0.001_dp)
        g(156672 + (y-1)*xgrid + x) = field2d(3, x, y, h)
        g(161280 + (y-1)*xgrid + x) = field2d(2, x, y, h)
        phi0(x, y) = 9.81_dp * 1500.0_dp * max(0.0_dp, sin(2.0_dp * lon) * cos(lat)**2 * sin(lat + 0.3_dp))
        fmask(x, y) = merge(1.0_dp, 0.0_dp, phi0(x, y) > 0.0_dp)
        tland(x, y) = field3d(1, x, y, 8, h, 0.0_dp)
        swav(x, y) = 0.5_dp; alb_l(x, y) = 0.2_dp; alb_s(x, y) = 0.07_dp; snowc(x, y) = 0.0_dp
        albsfc(x, y) = alb_s(x, y) + fmask(x, y) * (alb_l(x, y) - alb_s(x, y))
      end do
    end do
    do h = 1, 8760
      do y = 1, ygrid
        do x = 1, xgrid
          tisr(x, y, h) = field2d(2, x, y, h - 1)
        end do
      end do
    end do
  end subroutine
end module speedyml_data_source

module mod_io
  use iso_c_binding
  use speedyml_hip
  use mod_utilities, only : dp, reservoir_type, grid_type, model_parameters_type
  implicit none
  type cached_adjacency
    integer :: n = 0, k = 0, seed = 0
    integer(c_int), allocatable :: rows(:), cols(:)
    real(c_double), allocatable :: vals(:)
  end type
  type(cached_adjacency), save :: cache(64)
  integer, save :: ncached = 0
contains

  ! read_trained_res (src/mod_io.f90:2938-2983): win, wout, rows, cols, vals, mean, std of worker_RRRR_level_L_<trial>.nc -- here a
  ! synthetic trained reservoir whose W_out passes the imperfect model's forecast through with a small reservoir correction
  subroutine read_trained_res(reservoir, model_parameters, grid)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    type(sml_region) :: g
    type(sml_res_sizes) :: s
    real(kind=dp), allocatable :: r(:)
    real(c_double) :: eigs
    character(len=32) :: env
    integer :: i, q, m, mlen, stat, nl
    m = 6000
    call get_environment_variable('SML_RES_M', env, mlen, stat)
    if (stat == 0 .and. mlen > 0) read(env(1:mlen), *) m
    reservoir%sst_bool_input = reservoir%sst_bool .and. abs(grid%res_ystart - 24.5) < 14 .and. mod(grid%res_xstart / 8, 3) /= 0      ! "sea" regions
    call sml_check(sml_domain_region(int(grid%number_of_regions, c_int), int(reservoir%assigned_region, c_int), int(grid%overlap, c_int), &
                                     int(grid%num_vert_levels, c_int), int(grid%level_index, c_int), int(grid%vert_overlap, c_int), g), 'sml_domain_region')
    call sml_check(sml_domain_sizes(g, int(m, c_int), 6_c_int, 4_c_int, 1_c_int, merge(1_c_int, 0_c_int, model_parameters%precip_bool), &
                                    merge(1_c_int, 0_c_int, reservoir%sst_bool_input), 1_c_int, merge(1_c_int, 0_c_int, model_parameters%ml_only), s), 'sml_domain_sizes')
    if (allocated(reservoir%win)) deallocate(reservoir%win, reservoir%wout, reservoir%rows, reservoir%cols, reservoir%vals)
    allocate(reservoir%win(s%n, s%reservoir_numinputs), reservoir%wout(s%chunk_size_prediction, s%n + s%chunk_size_speedy), &
             reservoir%rows(s%k), reservoir%cols(s%k), reservoir%vals(s%k))
    ! (28 distinct synthetic adjacency matrices -- 7 seeds x 4 size classes -- for 1152 regions: generated once each)
    block
      integer :: c, hit
      hit = 0
      do c = 1, ncached
        if (cache(c)%n == s%n .and. cache(c)%k == s%k .and. cache(c)%seed == 777 + mod(reservoir%assigned_region, 7)) hit = c
      end do
      if (hit == 0) then
        call sml_check(sml_gen_res(s%n, s%k, 0.6_c_double, int(777 + mod(reservoir%assigned_region, 7), c_int64_t), reservoir%rows, reservoir%cols, reservoir%vals, eigs), 'sml_gen_res')
        if (ncached < size(cache)) then
          ncached = ncached + 1
          cache(ncached)%n = s%n; cache(ncached)%k = s%k; cache(ncached)%seed = 777 + mod(reservoir%assigned_region, 7)
          cache(ncached)%rows = reservoir%rows; cache(ncached)%cols = reservoir%cols; cache(ncached)%vals = reservoir%vals
        end if
      else
        reservoir%rows = cache(hit)%rows; reservoir%cols = cache(hit)%cols; reservoir%vals = cache(hit)%vals
      end if
    end block
    q = s%n / s%reservoir_numinputs
    allocate(r(q))
    reservoir%win = 0.0_dp
    do i = 1, s%reservoir_numinputs
      r = [(0.5_dp * sin(0.37_dp * (i * q + stat) + 1.1_dp), stat = 1, q)]
      reservoir%win((i-1)*q+1:i*q, i) = r
    end do
    reservoir%wout = 0.0_dp
    do i = 1, s%chunk_size_speedy
      reservoir%wout(i, i) = 1.0_dp
    end do
    do i = 1, s%chunk_size_prediction
      reservoir%wout(i, s%chunk_size_speedy + 1 + mod(7 * i, s%n)) = 1.0e-3_dp
      if (i > s%chunk_size_speedy) reservoir%wout(i, s%chunk_size_speedy + 1 + mod(11 * i, s%n)) = 0.05_dp
    end do
    ! SML_TEST_F32_WEIGHTS=1: values as a real weights file delivers them -- the reference writes and reads them as NF90_REAL
    ! (src/mod_io.f90), so every entry is exactly a float; the bank then reads its compact copies (sml_bank_storage)
    call get_environment_variable('SML_TEST_F32_WEIGHTS', env, mlen, stat)
    if (stat == 0 .and. mlen > 0 .and. env(1:1) == '1') then
      reservoir%vals = real(real(reservoir%vals, 4), dp)
      reservoir%win = real(real(reservoir%win, 4), dp)
      reservoir%wout = real(real(reservoir%wout, 4), dp)
    end if
    call synthetic_statistics(grid)
  end subroutine

  ! the statistics a weights file carries (mean / std per variable and level, then logp, tisr, precip, sst): the same synthetic climate
  ! for every region
  subroutine synthetic_statistics(grid)
    type(grid_type), intent(inout) :: grid
    integer :: i, nl
    nl = 4 * 8
    if (allocated(grid%mean)) deallocate(grid%mean, grid%std)
    allocate(grid%mean(nl + 4), grid%std(nl + 4))
    do i = 1, 8
      grid%mean(i) = 288.0_dp * (0.025_dp + 0.13_dp * (i - 1))**0.19_dp - 8.0_dp; grid%std(i) = 12.0_dp            ! T
      grid%mean(8 + i) = 8.0_dp; grid%std(8 + i) = 9.0_dp                                                             ! u
      grid%mean(16 + i) = 0.0_dp; grid%std(16 + i) = 3.0_dp                                                           ! v
      grid%mean(24 + i) = 4.0_dp * (0.025_dp + 0.13_dp * (i - 1))**3; grid%std(24 + i) = 1.0_dp + 3.0_dp * (0.13_dp * i)**3   ! q [g/kg]
    end do
    grid%mean(33) = -0.01_dp; grid%std(33) = 0.02_dp                  ! logp
    grid%mean(34) = 1.2e6_dp; grid%std(34) = 1.5e6_dp                 ! tisr
    grid%mean(35) = 0.5_dp; grid%std(35) = 0.6_dp                     ! precip (log-transformed)
    grid%mean(36) = 290.0_dp; grid%std(36) = 8.0_dp                   ! sst
  end subroutine

  ! read_trained_ocean_res (src/mod_io.f90:2985-3036): worker_RRRR_ocean_<trial>.nc -- here a synthetic trained slab reservoir for the
  ! regions whose atmosphere reservoir takes SST input ("the file exists"), with a small W_out (SST anomalies around the region's mean)
  subroutine read_trained_ocean_res(reservoir, model_parameters, grid)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    type(sml_region) :: g
    type(sml_res_sizes) :: s
    real(c_double) :: eigs
    character(len=32) :: env
    integer :: i, j, q, m, mlen, stat
    reservoir%sst_bool_input = abs(grid%res_ystart - 24.5) < 14 .and. mod(grid%res_xstart / 8, 3) /= 0            ! the "sea" regions of read_trained_res
    reservoir%sst_bool_prediction = reservoir%sst_bool_input
    if (.not. reservoir%sst_bool_input) return
    m = 4000
    call get_environment_variable('SML_SLAB_M', env, mlen, stat)
    if (stat == 0 .and. mlen > 0) read(env(1:mlen), *) m
    g%resxchunk = grid%resxchunk; g%resychunk = grid%resychunk; g%inputxchunk = grid%inputxchunk; g%inputychunk = grid%inputychunk
    call sml_check(sml_slab_sizes(g, int(m, c_int), 6_c_int, 4_c_int, s), 'sml_slab_sizes')
    if (allocated(reservoir%win)) deallocate(reservoir%win, reservoir%wout, reservoir%rows, reservoir%cols, reservoir%vals)
    allocate(reservoir%win(s%n, s%reservoir_numinputs), reservoir%wout(s%chunk_size_prediction, s%n), reservoir%rows(s%k), reservoir%cols(s%k), &
             reservoir%vals(s%k))
    block                                                             ! (cached as in read_trained_res)
      integer :: c, hit
      hit = 0
      do c = 1, ncached
        if (cache(c)%n == s%n .and. cache(c)%k == s%k .and. cache(c)%seed == 555 + mod(reservoir%assigned_region, 5)) hit = c
      end do
      if (hit == 0) then
        call sml_check(sml_gen_res(s%n, s%k, 0.6_c_double, int(555 + mod(reservoir%assigned_region, 5), c_int64_t), reservoir%rows, reservoir%cols, reservoir%vals, eigs), &
                       'sml_gen_res')
        if (ncached < size(cache)) then
          ncached = ncached + 1
          cache(ncached)%n = s%n; cache(ncached)%k = s%k; cache(ncached)%seed = 555 + mod(reservoir%assigned_region, 5)
          cache(ncached)%rows = reservoir%rows; cache(ncached)%cols = reservoir%cols; cache(ncached)%vals = reservoir%vals
        end if
      else
        reservoir%rows = cache(hit)%rows; reservoir%cols = cache(hit)%cols; reservoir%vals = cache(hit)%vals
      end if
    end block
    q = s%n / s%reservoir_numinputs
    reservoir%win = 0.0_dp
    do i = 1, s%reservoir_numinputs
      do j = 1, q
        reservoir%win((i-1)*q + j, i) = 0.6_dp * sin(0.29_dp * (i * q + j) + 0.7_dp)
      end do
    end do
    reservoir%wout = 0.0_dp
    do i = 1, s%chunk_size_prediction
      reservoir%wout(i, 1 + mod(13 * i, s%n)) = 0.02_dp
      reservoir%wout(i, 1 + mod(29 * i + 5, s%n)) = -0.015_dp
    end do
    call synthetic_statistics(grid)                    ! (a slab weights file carries the atmosphere reservoir's statistics, :366-367)
  end subroutine

  ! read_3d_file_parallel (src/mod_io.f90:2731-2812): (x, y, t) of the region's input patch from a NetCDF file, hourly from
  ! start_time_arg -- here the synthetic ocean heat content [J m-2] (and, for other variable names, zeros)
  subroutine read_3d_file_parallel(filename, varname, mpi_res, grid, var3d, start_time_arg, stride_arg, time_length)
    use mod_utilities, only : mpi_type
    character(len=*), intent(in) :: filename, varname
    type(mpi_type), intent(in) :: mpi_res
    type(grid_type), intent(in) :: grid
    real(kind=dp), allocatable, intent(inout) :: var3d(:,:,:)
    integer, intent(in), optional :: start_time_arg, stride_arg, time_length
    integer :: nt, t0, ix, iy, t, x, y
    real(kind=dp) :: lat, lon
    real(kind=dp), parameter :: pi = 3.14159265358979323846_dp
    nt = 8760; t0 = 1
    if (present(time_length)) nt = time_length
    if (present(start_time_arg)) t0 = start_time_arg
    if (allocated(var3d)) deallocate(var3d)
    allocate(var3d(grid%inputxchunk, grid%inputychunk, nt))
    var3d = 0.0_dp
    if (varname /= 'sohtc300') return
    do t = 1, nt
      do iy = 1, grid%inputychunk
        y = grid%input_ystart + iy - 1
        do ix = 1, grid%inputxchunk
          x = modulo(grid%input_xstart + ix - 2, 96) + 1
          lat = (y - 24.5_dp) * 3.71_dp * pi / 180.0_dp; lon = (x - 1) * 2.0_dp * pi / 96
          var3d(ix, iy, t) = 1.1e10_dp * cos(lat)**2 + 4.0e8_dp * sin(2.0_dp * lon) + 6.0e8_dp * sin(2.0_dp * pi * (t0 + t) / 8760.0_dp)
        end do
      end do
    end do
  end subroutine

  subroutine write_netcdf_2d_non_met_data(array, varname, filename, units, x_dim, y_dim)
    real(kind=dp), intent(in) :: array(:,:)
    character(len=*), intent(in) :: varname, filename, units, x_dim, y_dim
  end subroutine

  subroutine write_netcdf_1d_non_met_data_int(array, varname, filename, units, x_dim)
    integer, intent(in) :: array(:)
    character(len=*), intent(in) :: varname, filename, units, x_dim
  end subroutine

  subroutine write_netcdf_1d_non_met_data_real(array, varname, filename, units, x_dim)
    real(kind=dp), intent(in) :: array(:)
    character(len=*), intent(in) :: varname, filename, units, x_dim
  end subroutine
end module mod_io
