! TEST SUPPORT ONLY (used by test_driver.f90): per-call wrappers over the C-ABI -- one reservoir in a private one-slot bank, x and outvec
! on the host -- so that the parity driver can put single library calls (mklsparse, synchronize, predict, chunking_matmul,
! fit_chunk_hybrid) next to the reference's own statements written out in Fortran.  This is NOT a binding to integrate against: the
! drop-in with the reference's module names, argument lists and derived types, which program main compiles against unchanged, is
! mod_reservoir.f90 / mpires.f90 / resdomain.f90 / mod_utilities.f90 / mod_slab_ocean_reservoir.f90 (INTEGRATION.md section 2).
! The reduced reservoir_type below holds exactly the fields these wrappers touch.
module test_percall
  use iso_c_binding
  use speedyml_hip
  implicit none
  integer, parameter :: dp = c_double

  type reservoir_type
    integer :: assigned_region = 0
    integer :: n = 0, k = 0, reservoir_numinputs = 0
    integer :: chunk_size_speedy = 0, chunk_size_prediction = 0
    real(kind=dp) :: leakage = 1.0_dp
    integer, allocatable       :: rows(:), cols(:)
    real(kind=dp), allocatable :: vals(:)
    real(kind=dp), allocatable :: win(:,:), wout(:,:)
    real(kind=dp), allocatable :: feedback(:), local_model(:), outvec(:)
    real(kind=dp), allocatable :: mean(:), std(:)          ! grid%mean / grid%std
    integer, allocatable       :: out_stat_idx(:)          ! from sml_domain_out_map
    type(c_ptr) :: hip_bank = c_null_ptr                   ! replaces cooA / descrA
    integer(c_int) :: hip_slot = 0
    ! training (src/mod_utilities.f90:262-266): the Gram matrices stay on the device between batches
    real(kind=dp) :: beta_res = 0.001_dp, beta_model = 1.0_dp, prior_val = 0.0_dp
    logical :: using_prior = .true.
    type(c_ptr) :: hip_c = c_null_ptr, hip_b = c_null_ptr  ! states_x_states_aug (n_aug,n_aug), states_x_trainingdata_aug (n_out,n_aug)
  end type

contains

  subroutine mklsparse(reservoir)
    ! reference: mkl_sparse_d_create_coo on rows/cols/vals.  Here: upload A, W_in, W_out and the statistics once.
    type(reservoir_type), intent(inout) :: reservoir
    integer(c_int) :: rc
    if (.not. c_associated(reservoir%hip_bank)) then
      rc = sml_bank_create(1_c_int, int(reservoir%reservoir_numinputs, c_int), int(max(reservoir%chunk_size_speedy, 1), c_int), &
                           int(reservoir%chunk_size_prediction, c_int), reservoir%hip_bank)
      call sml_check(rc, 'sml_bank_create')
      reservoir%hip_slot = 0
    end if
    rc = sml_bank_load(reservoir%hip_bank, reservoir%hip_slot, int(reservoir%n, c_int), int(reservoir%reservoir_numinputs, c_int), &
                       int(reservoir%k, c_int), int(reservoir%chunk_size_speedy, c_int), int(reservoir%chunk_size_prediction, c_int), &
                       reservoir%rows, reservoir%cols, reservoir%vals, reservoir%win, reservoir%wout, reservoir%leakage, &
                       reservoir%mean, reservoir%std, int(size(reservoir%mean), c_int), reservoir%out_stat_idx)
    call sml_check(rc, 'sml_bank_load')
  end subroutine

  subroutine synchronize(reservoir, input, x, length)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(in)    :: input(:,:)
    real(kind=dp), intent(inout) :: x(:)
    integer, intent(in)          :: length
    integer :: i
    integer(c_int) :: rc
    rc = sml_bank_set_state(reservoir%hip_bank, reservoir%hip_slot, x)
    call sml_check(rc, 'sml_bank_set_state')
    do i = 1, length
      rc = sml_bank_set_feedback(reservoir%hip_bank, reservoir%hip_slot, input(:, i))
      call sml_check(rc, 'sml_bank_set_feedback')
      rc = sml_bank_advance_all(reservoir%hip_bank, c_null_ptr)
      call sml_check(rc, 'sml_bank_advance_all')
    end do
    rc = sml_bank_get_state(reservoir%hip_bank, reservoir%hip_slot, x)
    call sml_check(rc, 'sml_bank_get_state')
  end subroutine

  subroutine predict(reservoir, x, local_model_in)
    ! reference signature: predict(reservoir,model_parameters,grid,x,local_model_in); model_parameters and grid only feed
    ! the un-standardisation, whose statistics were uploaded by mklsparse.  reservoir%outvec is un-standardised on return.
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(inout) :: x(:)
    real(kind=dp), intent(inout) :: local_model_in(:)
    integer(c_int) :: rc
    rc = sml_bank_set_feedback(reservoir%hip_bank, reservoir%hip_slot, reservoir%feedback)
    call sml_check(rc, 'sml_bank_set_feedback')
    rc = sml_bank_predict_one(reservoir%hip_bank, reservoir%hip_slot, x, reservoir%local_model, reservoir%outvec)
    call sml_check(rc, 'sml_bank_predict_one')
  end subroutine

  ! chunking_matmul (src/mod_reservoir.f90:1645-1701): one batch of (squared-even) reservoir states, imperfect-model
  ! forecasts and targets is folded into the device-resident Gram matrices:  C += aug aug^T (tiles on/below the diagonal),
  ! B += Y aug^T with aug = [model ; states].  The reference slices model/targets out of imperfect_model_states / trainingdata
  ! by batch number; here the caller passes the three blocks.
  subroutine chunking_matmul(reservoir, states, model_block, target_block)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(in) :: states(:,:), model_block(:,:), target_block(:,:)     ! (n,m), (n_model,m), (n_out,m)
    integer(c_int) :: n, nm, no, m
    integer(c_int64_t) :: naug
    type(c_ptr) :: ds, dm, dy
    n = size(states, 1); m = size(states, 2); nm = size(model_block, 1); no = size(target_block, 1)
    naug = int(n + nm, c_int64_t)
    if (.not. c_associated(reservoir%hip_c)) then               ! initialize_chunk_training (:1561-1592): zeroed accumulators
      call sml_check(sml_dev_alloc(8_c_int64_t*naug*naug, reservoir%hip_c), 'sml_dev_alloc')
      call sml_check(sml_dev_alloc(8_c_int64_t*naug*no, reservoir%hip_b), 'sml_dev_alloc')
      call sml_check(sml_dev_zero(reservoir%hip_c, 8_c_int64_t*naug*naug), 'sml_dev_zero')
      call sml_check(sml_dev_zero(reservoir%hip_b, 8_c_int64_t*naug*no), 'sml_dev_zero')
    end if
    call sml_check(sml_dev_alloc(8_c_int64_t*n*m, ds), 'sml_dev_alloc')
    call sml_check(sml_dev_alloc(8_c_int64_t*max(nm, 1)*m, dm), 'sml_dev_alloc')
    call sml_check(sml_dev_alloc(8_c_int64_t*no*m, dy), 'sml_dev_alloc')
    call sml_check(sml_dev_upload(ds, states, 8_c_int64_t*n*m), 'sml_dev_upload')
    if (nm > 0) call sml_check(sml_dev_upload(dm, model_block, 8_c_int64_t*nm*m), 'sml_dev_upload')
    call sml_check(sml_dev_upload(dy, target_block, 8_c_int64_t*no*m), 'sml_dev_upload')
    call sml_check(sml_train_accumulate(ds, dm, dy, n, nm, no, m, reservoir%hip_c, reservoir%hip_b, c_null_ptr), 'sml_train_accumulate')
    call sml_check(sml_dev_free(ds), 'sml_dev_free')            ! hipFree waits for the accumulation
    call sml_check(sml_dev_free(dm), 'sml_dev_free')
    call sml_check(sml_dev_free(dy), 'sml_dev_free')
  end subroutine

  ! fit_chunk_hybrid (src/mod_reservoir.f90:1235-1334): ridge regularisation, dgesv on the transposed system, wout = Z^T.
  ! reservoir%wout (n_out, n_aug) is overwritten; the accumulators stay on the device (sml_train_fit does not destroy them).
  subroutine fit_chunk_hybrid(reservoir)
    type(reservoir_type), intent(inout) :: reservoir
    integer(c_int) :: n, nm, no
    type(c_ptr) :: dw
    n = reservoir%n; nm = reservoir%chunk_size_speedy; no = reservoir%chunk_size_prediction
    call sml_check(sml_dev_alloc(8_c_int64_t*no*(n + nm), dw), 'sml_dev_alloc')
    call sml_check(sml_train_fit(reservoir%hip_c, reservoir%hip_b, n, nm, no, reservoir%beta_res, reservoir%beta_model, &
                                 reservoir%prior_val, merge(1_c_int, 0_c_int, reservoir%using_prior), dw, c_null_ptr), 'sml_train_fit')
    call sml_check(sml_dev_download(reservoir%wout, dw, 8_c_int64_t*no*(n + nm)), 'sml_dev_download')
    call sml_check(sml_dev_free(dw), 'sml_dev_free')
  end subroutine

end module test_percall
