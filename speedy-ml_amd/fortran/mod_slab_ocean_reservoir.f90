! Module mod_slab_ocean_reservoir of the drop-in (src/mod_slab_ocean_reservoir.f90): the slab-ocean prediction calls of program main
! with the reference's argument lists.  predict_slab_ml (:1318-1363) and predict_slab (:1268-1316) are the same device kernels as
! predict with the slab reservoir's shapes (SST statistics for every output, sml_bank_load's out_stat map); the coupling schedule
! (every timestep_slab / timestep-th step, running mean of the atmosphere inputs) is sml_slab_* behind mpires::sendrecievegrid.
module mod_slab_ocean_reservoir
  use iso_c_binding
  use speedyml_hip
  use mod_utilities, only : dp, reservoir_type, grid_type, model_parameters_type
  implicit none
  type(c_ptr), save :: slab_bank = c_null_ptr
contains

  subroutine predict_slab_ml(reservoir, model_parameters, grid, x)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(inout) :: x(:)
    call sml_check(sml_bank_set_feedback(slab_bank, reservoir%hip_slot, reservoir%feedback), 'sml_bank_set_feedback')
    call sml_check(sml_bank_predict_one(slab_bank, reservoir%hip_slot, x, reservoir%feedback, reservoir%outvec), 'sml_bank_predict_one')
  end subroutine

  subroutine predict_slab(reservoir, model_parameters, grid, x, local_model_in)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(inout) :: x(:)
    real(kind=dp), intent(in) :: local_model_in(:)
    call sml_check(sml_bank_set_feedback(slab_bank, reservoir%hip_slot, reservoir%feedback), 'sml_bank_set_feedback')
    call sml_check(sml_bank_predict_one(slab_bank, reservoir%hip_slot, x, local_model_in, reservoir%outvec), 'sml_bank_predict_one')
  end subroutine

  ! load a trained slab reservoir (trained_ocean_reservoir_prediction :1389-1511 reads it from worker_RRRR_ocean_<trial>.nc): n_model =
  ! 0 for the ML-only ocean, every output un-standardised with the SST statistics (:1354)
  subroutine load_slab_reservoir(reservoir, grid, capacity, hybrid_ocean)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(in) :: grid
    integer, intent(in) :: capacity
    logical, intent(in) :: hybrid_ocean
    integer(c_int), allocatable :: stat(:)
    integer :: i
    if (.not. c_associated(slab_bank)) call sml_check(sml_bank_create(int(capacity, c_int), 192_c_int, 8_c_int, 8_c_int, slab_bank), 'sml_bank_create')
    allocate(stat(reservoir%chunk_size_prediction))
    stat = int(grid%sst_mean_std_idx - 1, c_int)
    call sml_check(sml_bank_load(slab_bank, reservoir%hip_slot, int(reservoir%n, c_int), int(reservoir%reservoir_numinputs, c_int), int(reservoir%k, c_int), &
                                 merge(int(reservoir%chunk_size_speedy, c_int), 0_c_int, hybrid_ocean), int(reservoir%chunk_size_prediction, c_int), &
                                 reservoir%rows, reservoir%cols, reservoir%vals, reservoir%win, reservoir%wout, reservoir%leakage, &
                                 grid%mean, grid%std, int(size(grid%mean), c_int), stat), 'sml_bank_load')
  end subroutine

end module mod_slab_ocean_reservoir
