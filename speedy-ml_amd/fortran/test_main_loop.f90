! Parity driver for the module-API drop-ins: the prediction part of the reference's program main (src/parallelmain.f90:140-272) --
! trained_reservoir_prediction / initialize_prediction for every region of the rank, then the forecast loop with start_prediction,
! predict and sendrecievegrid -- written against the SAME module names, procedure names and argument lists
! (mpires, mod_reservoir, resdomain, mod_utilities, mod_calendar), for two time steps with all 1152 regions on one rank.
! Checks: (1) the batched predict behind the per-region predict calls equals a per-region sml_bank_predict_one of the same reservoir
! on the same inputs, bit for bit; (2) the feedback sendrecievegrid leaves on the device for a region equals the host-side tiling +
! standardisation of the engine's global state through sml_domain_in_map; (3) run_speedy stays .true. on a physical state and the
! forecast moved.  Data come from the synthetic stand-ins of test_support.f90 (SML_RES_M keeps the reservoirs small).
program test_main_loop
  use iso_c_binding
  use mpires, only : mpi_res, startmpi, sendrecievegrid
  use mod_reservoir, only : initialize_model_parameters, start_prediction, initialize_prediction, predict, trained_reservoir_prediction, predict_ml, hip_fetch
  use resdomain, only : processor_decomposition, initializedomain
  use mod_utilities, only : main_type, dp, init_random_marker
  use mod_calendar
  use speedyml_hip
  use speedyml_state
  implicit none
  integer :: i, j, t, prediction_num, nfail, probe
  logical :: slab_model
  type(main_type) :: res
  real(kind=dp), allocatable :: x0(:), fb0(:), lm0(:), out_one(:), g(:), f(:), want(:), fb_dev(:)
  integer(c_int), allocatable :: gidx(:), stat(:)
  integer(c_int) :: cnt
  type(c_ptr) :: one
  nfail = 0

  call startmpi()
  call initialize_model_parameters(res%model_parameters, mpi_res%proc_num, mpi_res%numprocs)
  res%model_parameters%slab_ocean_model_bool = .false.          ! (the atmosphere's loop body; the slab calls are exercised from Python)
  call processor_decomposition(res%model_parameters)
  call init_random_marker(33)
  allocate(res%reservoir(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_vert_levels))
  allocate(res%grid(res%model_parameters%num_of_regions_on_proc, res%model_parameters%num_vert_levels))

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
  end do
  print *, 'loaded', hip_loaded, 'reservoirs; region 954: n, d =', res%reservoir(955,1)%n, res%reservoir(955,1)%reservoir_numinputs

  do i = 1, res%model_parameters%num_of_regions_on_proc
    do j = 1, res%model_parameters%num_vert_levels
      call initialize_prediction(res%reservoir(i,j), res%model_parameters, res%grid(i,j))
    end do
  end do

  probe = 955                                                     ! region 954 (0-based), interior
  do prediction_num = 1, res%model_parameters%num_predictions
    do t = 1, 2
      if (t == 1) then
        do i = 1, res%model_parameters%num_of_regions_on_proc
          do j = 1, res%model_parameters%num_vert_levels
            call start_prediction(res%reservoir(i,j), res%model_parameters, res%grid(i,j), prediction_num)
            res%reservoir(i,j)%current_state = res%reservoir(i,j)%saved_state
          end do
        end do
      end if
      ! ---- check (1), set-up: the probe's state and inputs before the step ----
      x0 = res%reservoir(probe,1)%current_state
      if (t == 1) then
        fb0 = res%reservoir(probe,1)%feedback; lm0 = res%reservoir(probe,1)%local_model
      else
        call fetch_inputs(res%reservoir(probe,1)%hip_slot, fb0, lm0)
        call sml_check(sml_bank_get_state(hip_bank, res%reservoir(probe,1)%hip_slot, x0), 'sml_bank_get_state')
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
      end do
      ! ---- check (1): per-region predict of the same reservoir on a one-slot bank ----
      call hip_fetch(res%reservoir(probe,1), res%reservoir(probe,1)%current_state)
      call one_slot_predict(res%reservoir(probe,1), res%grid(probe,1), x0, fb0, lm0, out_one)
      if (any(out_one /= res%reservoir(probe,1)%outvec) .or. any(x0 /= res%reservoir(probe,1)%current_state)) then
        print *, 'FAIL (1) step', t, maxval(abs(out_one - res%reservoir(probe,1)%outvec)); nfail = nfail + 1
      end if

      slab_model = res%model_parameters%slab_ocean_model_bool
      call sendrecievegrid(res, t, slab_model)
      if (res%model_parameters%run_speedy .eqv. .false.) then
        print *, 'FAIL (3): the range guard tripped at step', t; nfail = nfail + 1
        exit
      end if
      ! ---- check (2): the probe's next feedback against the host-side tiling of G ----
      allocate(g(165888), f(165888))
      call sml_check(sml_hybrid_get_state(hip_engine, g, f), 'sml_hybrid_get_state')
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
      if (t == 2 .and. .not. (maxval(abs(f(1:147456))) > 0.0_dp .and. all(f(1:147456) == f(1:147456)))) then
        print *, 'FAIL (3): forecast empty or NaN'; nfail = nfail + 1
      end if
      print *, 'step', t, ' probe outvec(1:3) =', res%reservoir(probe,1)%outvec(1:3), ' T range of the forecast', minval(f(1:147456:4)), maxval(f(1:147456:4))
      deallocate(g, f, gidx, stat, want)
    end do
  end do
  call slab_check(nfail)
  if (nfail == 0) then
    print *, 'main loop parity OK'
  else
    print *, 'main loop parity FAILED', nfail
    stop 1
  end if

contains

  ! ---- check (4): mod_slab_ocean_reservoir's predict_slab_ml (src/mod_slab_ocean_reservoir.f90:1318-1363) on a small ML-only ocean
  ! reservoir against the same step written out on the host: x <- tanh(A x + W_in u), even entries squared, W_out x~, every output
  ! un-standardised with the SST statistics ----
  subroutine slab_check(nfail)
    use mod_utilities, only : reservoir_type, grid_type
    use mod_slab_ocean_reservoir, only : load_slab_reservoir, predict_slab_ml
    integer, intent(inout) :: nfail
    type(reservoir_type) :: r
    type(grid_type) :: g
    integer, parameter :: n = 256, d = 16, no = 8, kk = 1536
    real(kind=dp), allocatable :: x(:), y(:), xa(:), want(:), u(:)
    real(kind=dp) :: rnd(kk)
    integer :: e
    r%n = n; r%reservoir_numinputs = d; r%k = kk; r%chunk_size_speedy = 0; r%chunk_size_prediction = no; r%leakage = 1.0_dp
    r%hip_slot = 0
    allocate(r%rows(kk), r%cols(kk), r%vals(kk), r%win(n, d), r%wout(no, n), r%feedback(d), r%outvec(no), x(n), u(kk))
    call random_number(rnd); r%rows = 1 + int(rnd * n); r%rows = min(r%rows, n)
    call random_number(rnd); r%cols = 1 + int(rnd * n); r%cols = min(r%cols, n)
    call random_number(r%vals); r%vals = (r%vals - 0.5_dp) * 0.3_dp
    call random_number(r%win); r%win = (r%win - 0.5_dp) * 0.6_dp
    call random_number(r%wout); r%wout = (r%wout - 0.5_dp) * 0.1_dp
    call random_number(r%feedback); r%feedback = r%feedback - 0.5_dp
    call random_number(x); x = (x - 0.5_dp) * 0.4_dp
    allocate(g%mean(36), g%std(36))
    call random_number(g%mean); call random_number(g%std); g%std = 0.5_dp + g%std
    g%sst_mean_std_idx = 36
    ! the host's own step
    allocate(y(n), xa(n), want(no))
    y = 0.0_dp
    do e = 1, kk
      y(r%rows(e)) = y(r%rows(e)) + r%vals(e) * x(r%cols(e))
    end do
    y = tanh(y + matmul(r%win, r%feedback))
    xa = y
    xa(2:n:2) = xa(2:n:2) ** 2
    want = matmul(r%wout, xa) * g%std(36) + g%mean(36)
    call load_slab_reservoir(r, g, 1, .false.)
    call predict_slab_ml(r, res%model_parameters, g, x)
    if (maxval(abs(x - y)) > 1.0e-13_dp .or. maxval(abs(r%outvec - want)) > 1.0e-11_dp * maxval(abs(want))) then
      print *, 'FAIL (4): predict_slab_ml', maxval(abs(x - y)), maxval(abs(r%outvec - want)); nfail = nfail + 1
    else
      print *, 'slab predict_slab_ml: state', maxval(abs(x - y)), ' outvec', maxval(abs(r%outvec - want)) / maxval(abs(want))
    end if
  end subroutine

  subroutine fetch_inputs(slot, fb, lm)
    integer(c_int), intent(in) :: slot
    real(kind=dp), allocatable, intent(inout) :: fb(:), lm(:)
    type(c_ptr) :: pf, pl
    real(kind=dp), pointer :: dummy
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
