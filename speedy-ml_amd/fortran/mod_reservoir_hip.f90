! Drop-in bodies for the prediction hot path of the reference's mod_reservoir / mod_linalg, forwarding to the
! MI355X library.  Same subroutine names and argument meaning as the reference:
!     mklsparse(reservoir)                      src/mod_linalg.f90:10-25     (build the device-resident operator)
!     synchronize(reservoir, input, x, length)  src/mod_reservoir.f90:1354-1381
!     predict(reservoir, x, local_model_in)     src/mod_reservoir.f90:1418-1489
! The reference's reservoir_type (src/mod_utilities.f90:168-330) carries MKL handles (cooA, descrA); the patch shown
! in INTEGRATION.md replaces them by the two fields hip_bank / hip_slot below.  This module defines a reduced
! reservoir_type with exactly the fields those three routines touch so that it builds stand-alone (the full type
! needs MKL_SPBLAS / mpi modules that are not part of this repository).
module mod_reservoir_hip
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

end module mod_reservoir_hip
