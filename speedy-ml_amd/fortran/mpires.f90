! Module mpires of the MI355X drop-in (src/mpires.f90): what program main imports (src/parallelmain.f90:6).  The reference gathers every
! region's outvec to the root over MPI, tiles the global grids there, runs SPEEDY on the root and sends every region its next inputs
! (sendrecievegrid, :218-804).  Here every rank holds its reservoirs in HBM and runs the global, deterministic SPEEDY replica itself,
! so the step's only exchange is the all-gather of the outvec slabs (atmosphere, and slab ocean when `ocean_model`), done INSIDE the
! library's hybrid engine (sml_hybrid_set_comm -> sml_comm_allgather_outvec: RCCL over xGMI), and sendrecievegrid is one call into
! that engine (sml_hybrid_*: SST assembly from the slab reservoirs, scatter + clamps, iogrid(30), the 6-hour window with the column
! physics, iogrid(31), get_tisr_by_date, the next feedback / local_model of every resident reservoir, the slab inputs' averaging ring).
!
! Ranks.  The image has no Fortran MPI module, so startmpi takes the rank and the rank count from the launcher's environment
! (SML_RANK / SML_NRANKS, else OMPI_COMM_WORLD_RANK / _SIZE, else PMI_RANK / PMI_SIZE; SML_LOCAL_RANK selects the GPU, default = rank
! modulo the visible devices) and the communicator comes from sml_comm_bootstrap (the RCCL id travels through /dev/shm/<SML_COMM_NAME>.id;
! SML_COMM_TRANSPORT=shm is the host-staged rehearsal transport for several ranks on one GPU).  A maintainer's MPI build fills mpi_res
! from mpi_comm_rank / mpi_comm_size and hands MPI_Bcast's copy of sml_comm_unique_id to sml_comm_create instead.
module mpires
  use iso_c_binding
  use speedyml_hip
  use speedyml_state
  use mod_utilities, only : dp, main_type, mpi_type, model_parameters_type, state_vector_type, xgrid, ygrid
  implicit none
  type(mpi_type) :: mpi_res
  type(state_vector_type) :: internal_state_vector
  integer, parameter :: leapfrog_steps_per_window = 24          ! nsteps / 4 after stepone's two starters (src/dyn_stloop.f90:26-43)
  logical, save :: engine_has_slab = .false.
contains

  integer function env_int(names, default)
    character(len=*), intent(in) :: names(:)
    integer, intent(in) :: default
    character(len=32) :: v
    integer :: i, n, stat
    env_int = default
    do i = 1, size(names)
      call get_environment_variable(trim(names(i)), v, n, stat)
      if (stat == 0 .and. n > 0) then
        read(v(1:n), *) env_int
        return
      end if
    end do
  end function

  subroutine startmpi()
    character(len=64) :: name
    integer :: n, stat, local, ndev
    mpi_res%ierr = 0
    mpi_res%proc_num = env_int([character(len=24) :: 'SML_RANK', 'OMPI_COMM_WORLD_RANK', 'PMI_RANK'], 0)
    mpi_res%numprocs = env_int([character(len=24) :: 'SML_NRANKS', 'OMPI_COMM_WORLD_SIZE', 'PMI_SIZE'], 1)
    mpi_res%is_root = mpi_res%proc_num == 0
    mpi_res%is_serial = mpi_res%numprocs == 1
    ndev = max(sml_device_count(), 1)
    local = env_int([character(len=24) :: 'SML_LOCAL_RANK', 'OMPI_COMM_WORLD_LOCAL_RANK'], mod(mpi_res%proc_num, ndev))
    call sml_check(sml_set_device(int(mod(local, ndev), c_int)), 'sml_set_device')
    if (mpi_res%numprocs > 1) then
      call get_environment_variable('SML_COMM_NAME', name, n, stat)
      if (stat /= 0 .or. n <= 0) then
        name = 'speedyml_comm'; n = len_trim(name)
      end if
      call sml_check(sml_comm_bootstrap(int(mpi_res%numprocs, c_int), int(mpi_res%proc_num, c_int), name(1:n) // c_null_char, 0_c_int64_t, hip_comm), &
                     'sml_comm_bootstrap')
    end if
  end subroutine

  ! (program main ends with MPI_Barrier + mpi_finalize; killmpi is the orderly exit of the reference's error paths)
  subroutine killmpi()
    integer(c_int) :: rc
    if (c_associated(hip_engine)) rc = sml_hybrid_destroy(hip_engine)
    hip_engine = c_null_ptr
    if (c_associated(hip_comm)) rc = sml_comm_destroy(hip_comm)
    hip_comm = c_null_ptr
    rc = sml_train_release_workspace()
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

  ! The engine is built at the first exchange: by then every reservoir of the rank is resident (speedyml_state).  Every forecast
  ! (timestep == 1) starts from the analysis of its own start hour: traininglength + prediction marker + synclength.
  subroutine start_forecast(res, ocean_model)
    use speedyml_data_source, only : hybrid_boundary_fields
    type(main_type), intent(inout) :: res
    logical, intent(in) :: ocean_model
    real(kind=dp), allocatable :: g(:), phi0(:,:), tisr(:,:,:), fmask(:,:), tland(:,:), swav(:,:), alb_l(:,:), alb_s(:,:), albsfc(:,:), snowc(:,:)
    real(kind=dp), allocatable :: fill(:), phis0(:,:)
    integer(c_int), allocatable :: mask(:), sea_of_region(:)
    real(kind=dp) :: hsg(9), radang(48)
    integer :: start_hours, s
    logical :: first
    first = .not. c_associated(hip_engine)
    if (first) then
      call sml_check(sml_hybrid_create(hip_bank, int(res%model_parameters%number_of_regions, c_int), region_of_slot, int(hip_loaded, c_int), &
                                       int(res%model_parameters%overlap, c_int), merge(1_c_int, 0_c_int, res%model_parameters%precip_bool), &
                                       sst_input_of_slot, hip_engine), 'sml_hybrid_create')
    end if
    allocate(g(165888), phi0(xgrid, ygrid), tisr(xgrid, ygrid, 8760), fmask(xgrid, ygrid), tland(xgrid, ygrid), swav(xgrid, ygrid), &
             alb_l(xgrid, ygrid), alb_s(xgrid, ygrid), albsfc(xgrid, ygrid), snowc(xgrid, ygrid))
    ! SPEEDY's boundary data (mod_surfcon phi0 / fmask1, the land and albedo fields phypar reads, the sigma half levels and Gaussian
    ! latitudes) and the hybrid's start state and TISR table: the reference's SPEEDY initialisation owns them (agcm_init, out of scope)
    call hybrid_boundary_fields(res%model_parameters, g, phi0, tisr, hsg, radang, fmask, tland, swav, alb_l, alb_s, albsfc, snowc)
    start_hours = res%model_parameters%traininglength + res%model_parameters%prediction_markers(max(res%model_parameters%current_trial_number, 1)) &
                  + res%model_parameters%synclength
    if (.not. first) call sml_check(sml_hybrid_restart(hip_engine, int(start_hours, c_int)), 'sml_hybrid_restart')
    call sml_check(sml_hybrid_set_state(hip_engine, g), 'sml_hybrid_set_state')
    if (.not. first) return
    call sml_check(sml_hybrid_set_orography(hip_engine, phi0), 'sml_hybrid_set_orography')
    call sml_check(sml_hybrid_set_tisr_table(hip_engine, tisr, int(start_hours, c_int), int(res%model_parameters%timestep, c_int)), 'sml_hybrid_set_tisr_table')
    ! mod_surfcon's phis0 = grid(trunct(spec(phi0))) (src/ini_invars.f90:31-34): what fordate and the physics read.  With the physics
    ! attached every window starts with fordate(0) on the device (tcorh, and qcorh from the hybrid SST; fmask_s = 1 - fmask1)
    allocate(phis0(xgrid, ygrid))
    call sml_check(sml_hybrid_get_phis0(hip_engine, phis0), 'sml_hybrid_get_phis0')
    call sml_check(sml_hybrid_attach_physics(hip_engine, hsg, radang, fmask, phis0, tland, swav, alb_l, alb_s, albsfc, snowc, 3_c_int), &
                   'sml_hybrid_attach_physics')
    if (c_associated(hip_comm)) call sml_check(sml_hybrid_set_comm(hip_engine, hip_comm), 'sml_hybrid_set_comm')
    if (ocean_model) then
      ! base_sst_grid / sea_mask (src/mod_reservoir.f90:846-884: the root reads them from the SST analysis; here every rank runs the
      ! replica and takes the start state's SST and the land mask of the boundary fields unless the host filled them)
      if (.not. allocated(res%model_parameters%base_sst_grid)) then
        allocate(res%model_parameters%base_sst_grid(xgrid, ygrid), res%model_parameters%sea_mask(xgrid, ygrid))
        res%model_parameters%base_sst_grid = reshape(g(156673:161280), [xgrid, ygrid])
        res%model_parameters%sea_mask = merge(1.0_dp, 0.0_dp, fmask >= 0.5_dp)
      end if
      allocate(mask(xgrid * ygrid), sea_of_region(res%model_parameters%number_of_regions))
      mask = merge(1_c_int, 0_c_int, reshape(res%model_parameters%sea_mask, [xgrid * ygrid]) > 0.0_dp)
      call sml_check(sml_hybrid_set_base_sst(hip_engine, reshape(res%model_parameters%base_sst_grid, [xgrid * ygrid]), mask), 'sml_hybrid_set_base_sst')
      call ensure_slab_bank()
      ! a region without a slab reservoir sends 272 K (src/mpires.f90:322-327,380-392): its row of the slab bank's output buffer holds
      ! that value from the start, so every row of the gathered slab is "the region's SST" whoever owns it
      allocate(fill(slab_max_out))
      fill = 272.0_dp
      do s = 1, hip_loaded
        if (slab_sea_of_slot(s) == 0) &
          call sml_check(sml_dev_upload_off(sml_bank_outvec_dev(hip_slab_bank), 8_c_int64_t * slab_max_out * (s - 1), fill, 8_c_int64_t * slab_max_out), 'sml_dev_upload')
      end do
      sea_of_region = 1
      call sml_check(sml_hybrid_attach_slab(hip_engine, hip_slab_bank, slab_sea_of_slot, sea_of_region, int(res%model_parameters%timestep_slab, c_int)), &
                     'sml_hybrid_attach_slab')
      engine_has_slab = .true.
    end if
  end subroutine

  ! (a rank whose regions are all land still takes part in the slab exchange)
  subroutine ensure_slab_bank()
    if (c_associated(hip_slab_bank)) return
    slab_max_d = 192; slab_max_out = 8
    call sml_check(sml_bank_create(int(hip_capacity, c_int), int(slab_max_d, c_int), int(slab_max_out, c_int), int(slab_max_out, c_int), hip_slab_bank), &
                   'sml_bank_create')
    allocate(slab_sea_of_slot(hip_capacity), slab_predicted(hip_capacity))
    slab_sea_of_slot = 0; slab_predicted = .false.; slab_loaded = 0; slab_done = 0
  end subroutine

  ! sendrecievegrid(res,timestep,ocean_model) (src/mpires.f90:218-804)
  subroutine sendrecievegrid(res, timestep, ocean_model)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
    logical, intent(in) :: ocean_model
    integer(c_int) :: safe
    if (timestep == 1 .or. .not. c_associated(hip_engine)) call start_forecast(res, ocean_model)
    if (ocean_model .neqv. engine_has_slab) stop 'mpires: sendrecievegrid was first called with another ocean_model setting'
    if (res%model_parameters%ml_only) then
      call sml_check(sml_hybrid_exchange_and_speedy(hip_engine, c_null_ptr, -1_c_int, c_null_ptr), 'sml_hybrid_exchange_and_speedy')
    else
      call sml_check(sml_hybrid_exchange_and_speedy(hip_engine, c_null_ptr, int(leapfrog_steps_per_window, c_int), c_null_ptr), &
                     'sml_hybrid_exchange_and_speedy')
    end if
    ! run_speedy: the range guard of iogrid(30); the reference broadcasts the root's flag (:744) -- here every rank's replica sees the same one
    call sml_check(sml_hybrid_safe(hip_engine, safe), 'sml_hybrid_safe')
    res%model_parameters%run_speedy = safe /= 0
    internal_state_vector%is_safe_to_run_speedy = safe /= 0
  end subroutine

  ! send_outvec_ml_contrib / send_outvec_speedy_contrib (src/mpires.f90:1076-1330): the root tiles every region's v_ml / v_p into global
  ! grids and writes them with write_netcdf, only reached with outvec_component_contribs = .true. (off as shipped,
  ! src/mod_reservoir.f90:71).  The two products themselves are part of predict and ARE computed (reservoir%v_p, reservoir%v_ml,
  ! mod_reservoir::predict -> sml_bank_outvec_contribs); what stops here is the NetCDF writer (out of scope, SURVEY section 2.1 #4).
  subroutine send_outvec_ml_contrib(res, timestep)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
    stop 'mpires: the NetCDF writer behind send_outvec_*_contrib is not part of the MI355X drop-in (reservoir%v_p / v_ml are filled by predict)'
  end subroutine

  subroutine send_outvec_speedy_contrib(res, timestep)
    type(main_type), intent(inout) :: res
    integer, intent(in) :: timestep
    stop 'mpires: the NetCDF writer behind send_outvec_*_contrib is not part of the MI355X drop-in (reservoir%v_p / v_ml are filled by predict)'
  end subroutine

end module mpires
