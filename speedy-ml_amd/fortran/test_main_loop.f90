! Parity driver for the module-API drop-ins: the trained-model part of the reference's program main (src/parallelmain.f90:140-272) --
! trained_reservoir_prediction / trained_ocean_reservoir_prediction, initialize_prediction / initialize_prediction_slab for every
! region of the rank, then the forecast loop with start_prediction / start_prediction_slab, predict, predict_slab_ml every
! timestep_slab / timestep-th step and sendrecievegrid(res, t, slab_model) -- written against the SAME module names, procedure names
! and argument lists (mpires, mod_reservoir, mod_slab_ocean_reservoir, resdomain, mod_utilities, mod_calendar).  The reference's own
! program main compiles and links against these modules too (make reference_main, tests/test_fortran_boundary.py); this driver is the
! part of it that can RUN here, with checks added.
!
! Environment: SML_TEST_SLAB (default 1: slab_ocean_model_bool as shipped), SML_TEST_STEPS (default 2; 30 reaches the slab step),
! SML_TEST_PREDICTIONS (default 1), SML_TEST_DUMP (a file that receives G, F and every resident region's next inputs, for the
! comparison of a 2-rank run with the 1-rank run), SML_RANK / SML_NRANKS / SML_COMM_* (mpires::startmpi), SML_RES_M / SML_SLAB_M.
! Checks: (1) the batched predict behind the per-region predict calls equals a per-region sml_bank_predict_one of the same reservoir
! on the same inputs, bit for bit; (2) the feedback sendrecievegrid leaves on the device for a region equals the host-side tiling +
! standardisation of the engine's global state through sml_domain_in_map; (3) run_speedy stays .true. on a physical state and the
! forecast moved; (4) the batched predict_slab_ml equals the slab step written out on the host (COO product, tanh, squared even
! entries, W_out, SST statistics), and its SST reaches the hybrid state; (5) a second forecast restarts the engine's step counter
! (TISR slice of its first step).  Data come from the synthetic stand-ins of test_support.f90.
program test_main_loop
  use iso_c_binding
  use mpires, only : mpi_res, startmpi, sendrecievegrid, killmpi
  use mod_reservoir, only : initialize_model_parameters, start_prediction, initialize_prediction, predict, trained_reservoir_prediction, predict_ml, hip_fetch
  use mod_slab_ocean_reservoir, only : initialize_prediction_slab, start_prediction_slab, predict_slab, predict_slab_ml, trained_ocean_reservoir_prediction
  use resdomain, only : processor_decomposition, initializedomain
  use mod_utilities, only : main_type, dp, init_random_marker
  use mod_calendar
  use speedyml_hip
  use speedyml_state
  implicit none
  interface
    function sml_tisr_index(startyear, hours_elapsed) bind(C, name="sml_tisr_index") result(idx)
      import :: c_int
      integer(c_int), value :: startyear, hours_elapsed
      integer(c_int) :: idx
    end function
  end interface
  integer :: i, j, t, prediction_num, nfail, probe, nsteps, npred, slab_fired
  logical :: slab_model, have_probe
  type(main_type) :: res
  real(kind=dp), allocatable :: x0(:), fb0(:), lm0(:), out_one(:), g(:), f(:), want(:), fb_dev(:), sst_first(:)
  real(kind=dp), allocatable :: sx(:), sfb(:), swant(:), sy(:)
  integer(c_int), allocatable :: gidx(:), stat(:)
  integer(c_int) :: cnt
  nfail = 0; slab_fired = 0

  call startmpi()
  call initialize_model_parameters(res%model_parameters, mpi_res%proc_num, mpi_res%numprocs)
  res%model_parameters%slab_ocean_model_bool = env_default('SML_TEST_SLAB', 1) /= 0
  res%model_parameters%outvec_component_contribs = env_default('SML_TEST_CONTRIBS', 0) /= 0        ! predict also fills v_p / v_ml
  nsteps = env_default('SML_TEST_STEPS', 2)
  npred = env_default('SML_TEST_PREDICTIONS', 1)
  if (npred /= res%model_parameters%num_predictions) then
    res%model_parameters%num_predictions = npred
    call redistribute_markers(res%model_parameters)
  end if
  call processor_decomposition(res%model_parameters)
  call init_random_marker(33)
  allocate(res%reservoir(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_vert_levels))
  allocate(res%grid(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_vert_levels))
  if (res%model_parameters%slab_ocean_model_bool) then
    res%model_parameters%special_reservoirs = .true.
    res%model_parameters%num_special_reservoirs = 1
  end if
  if (res%model_parameters%special_reservoirs) then
    allocate(res%reservoir_special(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_special_reservoirs))
    allocate(res%grid_special(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_special_reservoirs))
  end if

  ! ---- "if(trained_model)" (src/parallelmain.f90:140-183) ----
  do i = 1, res%model_parameters%num_of_regions_on_proc
    do j = 1, res%model_parameters%num_vert_levels
      call initializedomain(res%model_parameters%number_of_regions, res%model_parameters%region_indices(i), &
                            res%model_parameters%overlap, res%model_parameters%num_vert_levels, j, res%model_parameters%vert_loc_overlap, &
                            res%grid(i,j))
      res%reservoir(i,j)%assigned_region = res%model_parameters%region_indices(i)
      res%grid(i,j)%level_index = j
      call initialize_calendar(calendar, 1981, 1, 1, 0)
      call trained_reservoir_prediction(res%reservoir(i,j), res%model_parameters, res%grid(i,j))
    end do
    if (res%model_parameters%slab_ocean_model_bool) then
      call initializedomain(res%model_parameters%number_of_regions, res%model_parameters%region_indices(i), &
                            res%model_parameters%overlap, res%model_parameters%num_vert_levels, j-1, res%model_parameters%vert_loc_overlap, &
                            res%grid_special(i,1))
      res%grid_special(i,1)%level_index = j-1
      res%reservoir_special(i,1)%assigned_region = res%model_parameters%region_indices(i)
      call trained_ocean_reservoir_prediction(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir(i,j-1), res%grid(i,j-1))
    end if
  end do
  print *, 'rank', mpi_res%proc_num, 'of', mpi_res%numprocs, ': loaded', hip_loaded, 'atmosphere and', slab_loaded, 'slab reservoirs'

  ! ---- initialize prediction (:185-200) ----
  do i = 1, res%model_parameters%num_of_regions_on_proc
    do j = 1, res%model_parameters%num_vert_levels
      call initialize_prediction(res%reservoir(i,j), res%model_parameters, res%grid(i,j))
    end do
    if (res%model_parameters%slab_ocean_model_bool) &
      call initialize_prediction_slab(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir(i,j-1), res%grid(i,j-1))
  end do

  ! the probe: region 954 (interior), if this rank owns it -- and, for the slab check, the first sea region of the rank
  probe = 0
  do i = 1, res%model_parameters%num_of_regions_on_proc
    if (res%model_parameters%region_indices(i) == 954) probe = i
  end do
  have_probe = probe > 0

  ! ---- the forecast loop (:206-273) ----
  do prediction_num = 1, res%model_parameters%num_predictions
    do t = 1, nsteps
      if (t == 1) then
        do i = 1, res%model_parameters%num_of_regions_on_proc
          do j = 1, res%model_parameters%num_vert_levels
            call start_prediction(res%reservoir(i,j), res%model_parameters, res%grid(i,j), prediction_num)
            res%reservoir(i,j)%current_state = res%reservoir(i,j)%saved_state
          end do
          if (res%model_parameters%slab_ocean_model_bool) then
            call start_prediction_slab(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir(i,j-1), res%grid(i,j-1), prediction_num)
            if (res%reservoir_special(i,1)%sst_bool_prediction) res%reservoir_special(i,1)%current_state = res%reservoir_special(i,1)%saved_state
          end if
        end do
      end if
      ! ---- check (1), set-up: the probe's state and inputs before the step ----
      if (have_probe) then
        x0 = res%reservoir(probe,1)%current_state
        if (t == 1) then
          fb0 = res%reservoir(probe,1)%feedback; lm0 = res%reservoir(probe,1)%local_model
        else
          call fetch_inputs(res%reservoir(probe,1)%hip_slot, fb0, lm0)
          call sml_check(sml_bank_get_state(hip_bank, res%reservoir(probe,1)%hip_slot, x0), 'sml_bank_get_state')
        end if
      end if
      do i = 1, res%model_parameters%num_of_regions_on_proc
        do j = 1, res%model_parameters%num_vert_levels
          if (res%model_parameters%ml_only) then
            call predict_ml(res%reservoir(i,j), res%model_parameters, res%grid(i,j), res%reservoir(i,j)%current_state)
            res%model_parameters%run_speedy = .true.
          else
            call predict(res%reservoir(i,j), res%model_parameters, res%grid(i,j), res%reservoir(i,j)%current_state, res%reservoir(i,j)%local_model)
          end if
        end do
        if (res%model_parameters%slab_ocean_model_bool) then
          if (mod(t * res%model_parameters%timestep, res%model_parameters%timestep_slab) == 0 .and. res%reservoir_special(i,1)%sst_bool_prediction &
              .and. .not. res%model_parameters%non_stationary_ocn_climo) then
            if (slab_fired == 0) call slab_step_setup(i)                ! check (4), set-up: the first slab reservoir, before the batched step
            if (res%model_parameters%ml_only_ocean) then
              call predict_slab_ml(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir_special(i,1)%current_state)
            else
              call predict_slab(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir_special(i,1)%current_state, &
                                res%reservoir_special(i,1)%local_model)
            end if
            if (slab_fired == 0) call slab_step_check(i, nfail)
            slab_fired = slab_fired + 1
          end if
        end if
      end do
      ! ---- check (1): per-region predict of the same reservoir on a one-slot bank ----
      if (have_probe) then
        call hip_fetch(res%reservoir(probe,1), res%reservoir(probe,1)%current_state)
        call one_slot_predict(res%reservoir(probe,1), res%grid(probe,1), x0, fb0, lm0, out_one)
        if (any(out_one /= res%reservoir(probe,1)%outvec) .or. any(x0 /= res%reservoir(probe,1)%current_state)) then
          print *, 'FAIL (1) step', t, maxval(abs(out_one - res%reservoir(probe,1)%outvec)); nfail = nfail + 1
        end if
        ! ---- check (6): outvec_component_contribs -- v_p is the physics-model block of the readout on the inputs of this step ----
        if (res%model_parameters%outvec_component_contribs) then
          if (.not. allocated(res%reservoir(probe,1)%v_p) .or. .not. allocated(res%reservoir(probe,1)%v_ml)) then
            print *, 'FAIL (6): predict left v_p / v_ml unallocated'; nfail = nfail + 1
          else
            want = matmul(res%reservoir(probe,1)%wout(:, 1:res%reservoir(probe,1)%chunk_size_speedy), lm0(1:res%reservoir(probe,1)%chunk_size_speedy))
            if (maxval(abs(want - res%reservoir(probe,1)%v_p)) > 1.0e-12_dp * max(1.0_dp, maxval(abs(want))) .or. &
                .not. all(res%reservoir(probe,1)%v_ml == res%reservoir(probe,1)%v_ml) .or. maxval(abs(res%reservoir(probe,1)%v_ml)) == 0.0_dp) then
              print *, 'FAIL (6) step', t, maxval(abs(want - res%reservoir(probe,1)%v_p)); nfail = nfail + 1
            else if (t == 1) then
              print *, 'split readout of region 954: |v_p| max', maxval(abs(res%reservoir(probe,1)%v_p)), ' |v_ml| max', maxval(abs(res%reservoir(probe,1)%v_ml))
            end if
            deallocate(want)
          end if
        end if
      end if

      slab_model = res%model_parameters%slab_ocean_model_bool
      call sendrecievegrid(res, t, slab_model)
      if (res%model_parameters%run_speedy .eqv. .false.) then
        print *, 'FAIL (3): the range guard tripped at step', t; nfail = nfail + 1
        exit
      end if
      allocate(g(165888), f(165888))
      call sml_check(sml_hybrid_get_state(hip_engine, g, f), 'sml_hybrid_get_state')
      ! ---- check (2): the probe's next feedback against the host-side tiling of G ----
      if (have_probe) then
        allocate(gidx(res%reservoir(probe,1)%reservoir_numinputs), stat(res%reservoir(probe,1)%reservoir_numinputs))
        cnt = sml_domain_in_map(1152_c_int, int(res%reservoir(probe,1)%assigned_region, c_int), 1_c_int, 1_c_int, 1_c_int, 0_c_int, 1_c_int, &
                                merge(1_c_int, 0_c_int, res%reservoir(probe,1)%sst_bool_input), 1_c_int, gidx, stat, int(size(gidx), c_int))
        call sml_check(cnt, 'sml_domain_in_map')
        allocate(want(cnt))
        do i = 1, cnt
          want(i) = (g(gidx(i) + 1) - res%grid(probe,1)%mean(stat(i) + 1)) / res%grid(probe,1)%std(stat(i) + 1)
        end do
        call fetch_inputs(res%reservoir(probe,1)%hip_slot, fb_dev, lm0)
        if (any(want /= fb_dev(1:cnt))) then
          print *, 'FAIL (2) step', t, maxval(abs(want - fb_dev(1:cnt))); nfail = nfail + 1
        end if
        deallocate(gidx, stat, want)
      end if
      if (t == 2 .and. .not. (maxval(abs(f(1:147456))) > 0.0_dp .and. all(f(1:147456) == f(1:147456)))) then
        print *, 'FAIL (3): forecast empty or NaN'; nfail = nfail + 1
      end if
      ! ---- check (5): the TISR slice of the step is the one of the forecast's own calendar (restart of a second forecast) ----
      if (t == 1) call tisr_check(prediction_num, g, nfail)
      if (slab_model) then
        if (t == 1 .and. prediction_num == 1) sst_first = g(156673:161280)
        if (slab_fired > 0 .and. prediction_num == 1 .and. t == nsteps) then
          if (all(g(156673:161280) == sst_first)) then
            print *, 'FAIL (4): the slab reservoirs fired but the SST of the hybrid state never moved'; nfail = nfail + 1
          end if
        end if
      end if
      if (t <= 2 .or. t == nsteps) print *, 'prediction', prediction_num, 'step', t, ' T range of the forecast', minval(f(1:147456:4)), maxval(f(1:147456:4)), &
                                            ' SST range', minval(g(156673:161280)), maxval(g(156673:161280))
      if (t == nsteps .and. prediction_num == res%model_parameters%num_predictions) call dump_state(g, f)
      deallocate(g, f)
    end do
  end do
  ! ---- the same loop body, bare and timed: what a step costs through the Fortran host (SML_TEST_TIMED_STEPS iterations of
  ! src/parallelmain.f90:207-272 -- predict for every region of the rank, then sendrecievegrid -- between two system_clock readings) ----
  call timed_main_loop(env_default('SML_TEST_TIMED_STEPS', 0), nsteps)
  if (res%model_parameters%slab_ocean_model_bool .and. nsteps >= 28 .and. slab_loaded > 0 .and. slab_fired == 0) then
    print *, 'FAIL (4): 28 steps taken and no slab reservoir was stepped'; nfail = nfail + 1
  end if
  if (nfail == 0) then
    print *, 'main loop parity OK'
  else
    print *, 'main loop parity FAILED', nfail
    stop 1
  end if
  call killmpi()

contains

  subroutine timed_main_loop(ntimed, t_done)
    integer, intent(in) :: ntimed, t_done
    integer(kind=8) :: c0, c1, rate
    integer :: tt, ii, jj
    logical :: ocean
    if (ntimed <= 0) return
    ocean = res%model_parameters%slab_ocean_model_bool
    call sml_check(sml_device_synchronize(), 'sml_device_synchronize')
    call system_clock(c0, rate)
    do tt = t_done + 1, t_done + ntimed
      do ii = 1, res%model_parameters%num_of_regions_on_proc
        do jj = 1, res%model_parameters%num_vert_levels
          call predict(res%reservoir(ii,jj), res%model_parameters, res%grid(ii,jj), res%reservoir(ii,jj)%current_state, res%reservoir(ii,jj)%local_model)
        end do
        if (ocean) then
          if (mod(tt * res%model_parameters%timestep, res%model_parameters%timestep_slab) == 0 .and. res%reservoir_special(ii,1)%sst_bool_prediction) &
            call predict_slab_ml(res%reservoir_special(ii,1), res%model_parameters, res%grid_special(ii,1), res%reservoir_special(ii,1)%current_state)
        end if
      end do
      call sendrecievegrid(res, tt, ocean)
      if (.not. res%model_parameters%run_speedy) exit
    end do
    call sml_check(sml_device_synchronize(), 'sml_device_synchronize')
    call system_clock(c1)
    print '(a,i0,a,f9.4,a,i0,a)', ' timed main loop: ', ntimed, ' steps through the Fortran host, ', 1.0d3 * dble(c1 - c0) / dble(rate) / dble(ntimed), &
          ' ms per step (', res%model_parameters%num_of_regions_on_proc, ' regions on this rank; predict x regions + sendrecievegrid, system_clock)'
  end subroutine

  integer function env_default(name, default)
    character(len=*), intent(in) :: name
    integer, intent(in) :: default
    character(len=32) :: v
    integer :: n, st
    env_default = default
    call get_environment_variable(name, v, n, st)
    if (st == 0 .and. n > 0) read(v(1:n), *) env_default
  end function

  subroutine redistribute_markers(model_parameters)
    use mpires, only : distribute_prediction_marker
    use mod_utilities, only : model_parameters_type
    type(model_parameters_type), intent(inout) :: model_parameters
    call distribute_prediction_marker(model_parameters)
  end subroutine

  ! ---- check (4): predict_slab_ml (src/mod_slab_ocean_reservoir.f90:1318-1363) of slab reservoir `i` against the same step written out on
  ! the host: x <- tanh(A x + W_in u), even entries squared, W_out x~, every output un-standardised with the SST statistics ----
  subroutine slab_step_setup(i)
    integer, intent(in) :: i
    integer :: e, n, d
    n = res%reservoir_special(i,1)%n; d = res%reservoir_special(i,1)%reservoir_numinputs
    if (allocated(sx)) deallocate(sx, sfb, swant, sy)
    allocate(sx(n), sfb(slab_max_d), sy(n), swant(res%reservoir_special(i,1)%chunk_size_prediction))
    call sml_check(sml_bank_get_state(hip_slab_bank, res%reservoir_special(i,1)%hip_slot, sx), 'sml_bank_get_state')
    call sml_check(sml_dev_download_off(sfb, sml_bank_feedback_dev(hip_slab_bank), 8_c_int64_t * slab_max_d * res%reservoir_special(i,1)%hip_slot, &
                                        8_c_int64_t * slab_max_d), 'download slab feedback')
    sy = 0.0_dp
    do e = 1, res%reservoir_special(i,1)%k
      sy(res%reservoir_special(i,1)%rows(e)) = sy(res%reservoir_special(i,1)%rows(e)) + res%reservoir_special(i,1)%vals(e) * sx(res%reservoir_special(i,1)%cols(e))
    end do
    sy = tanh(sy + matmul(res%reservoir_special(i,1)%win, sfb(1:d)))
    sx = sy
    sx(2:n:2) = sx(2:n:2) ** 2
    swant = matmul(res%reservoir_special(i,1)%wout, sx) * res%grid_special(i,1)%std(res%grid_special(i,1)%sst_mean_std_idx) &
            + res%grid_special(i,1)%mean(res%grid_special(i,1)%sst_mean_std_idx)
  end subroutine

  subroutine slab_step_check(i, nfail)
    integer, intent(in) :: i
    integer, intent(inout) :: nfail
    real(kind=dp), allocatable :: xs(:), os(:)
    allocate(xs(res%reservoir_special(i,1)%n), os(res%reservoir_special(i,1)%chunk_size_prediction))
    call sml_check(sml_bank_get_state(hip_slab_bank, res%reservoir_special(i,1)%hip_slot, xs), 'sml_bank_get_state')
    call sml_check(sml_bank_get_outvec(hip_slab_bank, res%reservoir_special(i,1)%hip_slot, os), 'sml_bank_get_outvec')
    if (maxval(abs(xs - sy)) > 1.0e-13_dp .or. maxval(abs(os - swant)) > 1.0e-11_dp * maxval(abs(swant)) .or. maxval(abs(sfb)) == 0.0_dp) then
      print *, 'FAIL (4): predict_slab_ml', maxval(abs(xs - sy)), maxval(abs(os - swant)), maxval(abs(sfb)); nfail = nfail + 1
    else
      print *, 'slab predict_slab_ml of region', res%reservoir_special(i,1)%assigned_region, ': state', maxval(abs(xs - sy)), ' outvec', &
               maxval(abs(os - swant)) / maxval(abs(swant)), ' SST', os(1:2)
    end if
  end subroutine

  ! G's TISR segment after step 1 of a forecast = the slice get_tisr_by_date(timestep - 1) picks for that forecast's start hour
  subroutine tisr_check(prediction_num, g, nfail)
    use speedyml_data_source, only : field2d
    integer, intent(in) :: prediction_num
    real(kind=dp), intent(in) :: g(:)
    integer, intent(inout) :: nfail
    integer :: start_hours, idx, x, y
    real(kind=dp) :: worst
    start_hours = res%model_parameters%traininglength + res%model_parameters%prediction_markers(prediction_num) + res%model_parameters%synclength
    idx = sml_tisr_index(1981_c_int, int(start_hours, c_int))
    worst = 0.0_dp
    do y = 1, 48
      do x = 1, 96
        worst = max(worst, abs(g(161280 + (y-1)*96 + x) - field2d(2, x, y, idx - 1)))
      end do
    end do
    if (worst /= 0.0_dp) then
      print *, 'FAIL (5): TISR slice of forecast', prediction_num, 'step 1 is not table slice', idx, worst; nfail = nfail + 1
    end if
  end subroutine

  ! G, F and every resident region's next inputs, for the comparison of runs with different rank counts
  subroutine dump_state(g, f)
    real(kind=dp), intent(in) :: g(:), f(:)
    character(len=256) :: path
    real(kind=dp), allocatable :: fb(:), lm(:)
    integer :: n, st, u, i
    call get_environment_variable('SML_TEST_DUMP', path, n, st)
    if (st /= 0 .or. n <= 0) return
    open(newunit=u, file=path(1:n), access='stream', form='unformatted', status='replace')
    write(u) int(hip_loaded, c_int)
    write(u) g, f(1:152064)
    do i = 1, hip_loaded
      call fetch_inputs(int(i - 1, c_int), fb, lm)
      write(u) region_of_slot(i), fb, lm
    end do
    close(u)
  end subroutine

  subroutine fetch_inputs(slot, fb, lm)
    integer(c_int), intent(in) :: slot
    real(kind=dp), allocatable, intent(inout) :: fb(:), lm(:)
    type(c_ptr) :: pf, pl
    if (allocated(fb)) deallocate(fb)
    if (allocated(lm)) deallocate(lm)
    allocate(fb(576), lm(132))
    pf = sml_bank_feedback_dev(hip_bank); pl = sml_bank_local_model_dev(hip_bank)
    call sml_check(sml_dev_download_off(fb, pf, int(slot, c_int64_t) * 576 * 8, 576_c_int64_t * 8), 'download feedback')
    call sml_check(sml_dev_download_off(lm, pl, int(slot, c_int64_t) * 132 * 8, 132_c_int64_t * 8), 'download local_model')
  end subroutine

  subroutine one_slot_predict(reservoir, grid, x, fb, lm, out)
    use mod_utilities, only : reservoir_type, grid_type
    type(reservoir_type), intent(in) :: reservoir
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(inout) :: x(:)
    real(kind=dp), intent(in) :: fb(:), lm(:)
    real(kind=dp), allocatable, intent(inout) :: out(:)
    integer(c_int), allocatable :: gi(:), st(:)
    integer(c_int) :: c
    type(c_ptr) :: b
    allocate(gi(reservoir%chunk_size_prediction), st(reservoir%chunk_size_prediction))
    c = sml_domain_out_map(1152_c_int, int(reservoir%assigned_region, c_int), 1_c_int, 1_c_int, 0_c_int, 1_c_int, gi, st, int(size(gi), c_int))
    call sml_check(c, 'sml_domain_out_map')
    call sml_check(sml_bank_create(1_c_int, 576_c_int, 132_c_int, 136_c_int, b), 'sml_bank_create')
    call sml_check(sml_bank_load(b, 0_c_int, int(reservoir%n, c_int), int(reservoir%reservoir_numinputs, c_int), int(reservoir%k, c_int), &
                                 int(reservoir%chunk_size_speedy, c_int), int(reservoir%chunk_size_prediction, c_int), reservoir%rows, reservoir%cols, &
                                 reservoir%vals, reservoir%win, reservoir%wout, reservoir%leakage, grid%mean, grid%std, int(size(grid%mean), c_int), st), 'sml_bank_load')
    call sml_check(sml_bank_set_feedback(b, 0_c_int, fb), 'sml_bank_set_feedback')
    if (allocated(out)) deallocate(out)
    allocate(out(reservoir%chunk_size_prediction))
    call sml_check(sml_bank_predict_one(b, 0_c_int, x, lm, out), 'sml_bank_predict_one')
    call sml_check(sml_bank_destroy(b), 'sml_bank_destroy')
  end subroutine
end program test_main_loop
