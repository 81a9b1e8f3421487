! Module mpires of the MI355X drop-in (src/mpires.f90): what program main imports (src/parallelmain.f90:6).  The reference gathers every
! region's outvec to the root over MPI, tiles the global grids there, runs SPEEDY on the root and sends every region its next inputs
! (sendrecievegrid, :218-804).  Here every rank holds its reservoirs in HBM and runs the global, deterministic SPEEDY replica itself,
! so the step's only exchange is the all-gather of the outvec slab (RCCL through sml_comm_* for a multi-rank host; a single rank needs
! none) and sendrecievegrid is one call into the library's hybrid engine (sml_hybrid_*: scatter + clamps, iogrid(30), the 6-hour
! window with the column physics, iogrid(31), get_tisr_by_date, the next feedback / local_model of every resident reservoir).
!
! Ranks: startmpi takes the rank and the rank count from the launcher's environment (SML_RANK / SML_NRANKS, or the usual
! OMPI_COMM_WORLD_* / PMI_* variables) -- the image has no Fortran MPI module; a maintainer's MPI build fills mpi_res from
! mpi_comm_rank / mpi_comm_size and broadcasts the 128-byte RCCL id of sml_comm_unique_id.
module mpires
  use iso_c_binding
  use speedyml_hip
  use speedyml_state
  use mod_utilities, only : dp, main_type, mpi_type, model_parameters_type, state_vector_type, xgrid, ygrid
  implicit none
  type(mpi_type) :: mpi_res
  type(state_vector_type) :: internal_state_vector
  integer, parameter :: leapfrog_steps_per_window = 24          ! nsteps / 4 after stepone's two starters (src/dyn_stloop.f90:26-43)
contains

  subroutine startmpi()
    character(len=32) :: v
    integer :: n, stat
    mpi_res%proc_num = 0; mpi_res%numprocs = 1; mpi_res%ierr = 0
    call get_environment_variable('SML_RANK', v, n, stat)
    if (stat == 0 .and. n > 0) read(v(1:n), *) mpi_res%proc_num
    call get_environment_variable('SML_NRANKS', v, n, stat)
    if (stat == 0 .and. n > 0) read(v(1:n), *) mpi_res%numprocs
    mpi_res%is_root = mpi_res%proc_num == 0
    mpi_res%is_serial = mpi_res%numprocs == 1
    call sml_check(sml_set_device(0_c_int), 'sml_set_device')
  end subroutine

  subroutine killmpi()
    stop
  end subroutine

  ! prediction markers: one forecast every synclength hours (src/mpires.f90:928-948)
  subroutine distribute_prediction_marker(model_parameters)
    type(model_parameters_type), intent(inout) :: model_parameters
    integer :: i
    if (allocated(model_parameters%prediction_markers)) deallocate(model_parameters%prediction_markers)
    allocate(model_parameters%prediction_markers(model_parameters%num_predictions))
    do i = 1, model_parameters%num_predictions
      model_parameters%prediction_markers(i) = model_parameters%synclength * (i - 1)
    end do
  end subroutine

  subroutine predictionmpicontroller(res, timestep)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
    call sendrecievegrid(res, timestep, .false.)
  end subroutine

  ! the engine is built at the first exchange: by then every reservoir of the rank is resident (speedyml_state)
  subroutine build_engine(res)
    use speedy_res_interface, only : hybrid_boundary_fields
    type(main_type), intent(inout) :: res
    real(kind=dp), allocatable :: g(:), phi0(:,:), tisr(:,:,:), fmask(:,:), tland(:,:), swav(:,:), alb_l(:,:), alb_s(:,:), albsfc(:,:), snowc(:,:)
    real(kind=dp) :: hsg(9), radang(48)
    integer :: start_hours
    call sml_check(sml_hybrid_create(hip_bank, int(res%model_parameters%number_of_regions, c_int), region_of_slot, int(hip_loaded, c_int), &
                                     int(res%model_parameters%overlap, c_int), merge(1_c_int, 0_c_int, res%model_parameters%precip_bool), &
                                     sst_input_of_slot, hip_engine), 'sml_hybrid_create')
    allocate(g(165888), phi0(xgrid, ygrid), tisr(xgrid, ygrid, 8760), fmask(xgrid, ygrid), tland(xgrid, ygrid), swav(xgrid, ygrid), &
             alb_l(xgrid, ygrid), alb_s(xgrid, ygrid), albsfc(xgrid, ygrid), snowc(xgrid, ygrid))
    ! SPEEDY's boundary data (mod_surfcon phi0 / fmask1, the land and albedo fields phypar reads, the sigma half levels and Gaussian
    ! latitudes) and the hybrid's start state and TISR table: the reference's SPEEDY initialisation owns them (agcm_init, out of scope)
    call hybrid_boundary_fields(res%model_parameters, g, phi0, tisr, hsg, radang, fmask, tland, swav, alb_l, alb_s, albsfc, snowc)
    call sml_check(sml_hybrid_set_state(hip_engine, g), 'sml_hybrid_set_state')
    call sml_check(sml_hybrid_set_orography(hip_engine, phi0), 'sml_hybrid_set_orography')
    start_hours = res%model_parameters%traininglength + res%model_parameters%prediction_markers(max(res%model_parameters%current_trial_number, 1)) &
                  + res%model_parameters%synclength
    call sml_check(sml_hybrid_set_tisr_table(hip_engine, tisr, int(start_hours, c_int), int(res%model_parameters%timestep, c_int)), 'sml_hybrid_set_tisr_table')
    call sml_check(sml_hybrid_attach_physics(hip_engine, hsg, radang, fmask, max(phi0, 0.0_dp), tland, swav, alb_l, alb_s, albsfc, snowc, 3_c_int), &
                   'sml_hybrid_attach_physics')
  end subroutine

  ! sendrecievegrid(res,timestep,ocean_model) (src/mpires.f90:218-804)
  subroutine sendrecievegrid(res, timestep, ocean_model)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
    logical, intent(in) :: ocean_model
    integer(c_int) :: safe
    if (.not. c_associated(hip_engine)) call build_engine(res)
    if (res%model_parameters%ml_only) then
      call sml_check(sml_hybrid_exchange_and_speedy(hip_engine, c_null_ptr, -1_c_int, c_null_ptr), 'sml_hybrid_exchange_and_speedy')
    else
      call sml_check(sml_hybrid_exchange_and_speedy(hip_engine, c_null_ptr, int(leapfrog_steps_per_window, c_int), c_null_ptr), &
                     'sml_hybrid_exchange_and_speedy')
    end if
    ! run_speedy: the range guard of iogrid(30); the reference broadcasts it to every rank (:744)
    call sml_check(sml_hybrid_safe(hip_engine, safe), 'sml_hybrid_safe')
    res%model_parameters%run_speedy = safe /= 0
    internal_state_vector%is_safe_to_run_speedy = safe /= 0
  end subroutine

  ! diagnostics of the split outvec (outvec_component_contribs = .false. in the shipped configuration)
  subroutine send_outvec_ml_contrib(res, timestep)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
  end subroutine

  subroutine send_outvec_speedy_contrib(res, timestep)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
  end subroutine

end module mpires
