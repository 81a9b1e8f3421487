! Module mod_reservoir of the MI355X drop-in: the procedures program main imports (src/parallelmain.f90:7) with the reference's
! argument lists --
!     initialize_model_parameters(model_parameters,processor,num_of_procs)      src/mod_reservoir.f90:16-78
!     allocate_res_new(reservoir,grid,model_parameters)                         :80-180
!     train_reservoir(reservoir,grid,model_parameters)                          :214-320
!     trained_reservoir_prediction(reservoir,model_parameters,grid)             :1783-1886
!     initialize_prediction(reservoir,model_parameters,grid)                    :791-887
!     start_prediction(reservoir,model_parameters,grid,prediction_number)       :940-961
!     predict(reservoir,model_parameters,grid,x,local_model_in)                 :1418-1489
!     predict_ml(reservoir,model_parameters,grid,x)                             :1491-1535
!     synchronize(reservoir,input,x,length) / synchronize_print(...)            :1354-1416
! -- and the module variable global_time_step (src/dyn_stloop.f90:15,23).  The arithmetic runs in libspeedyml_hip.so.
!
! Device residency.  All reservoirs of the rank live in ONE bank in HBM (speedyml_state%hip_bank; reservoir%hip_slot replaces
! the MKL handles cooA / descrA).  program main calls predict once per region and time step; the FIRST such call of a time step
! launches the batched predict of every resident reservoir (two kernels), the others find their work done.  A reservoir's
! feedback and local_model are written on the device by mpires::sendrecievegrid; reservoir%outvec / x on the host are refreshed
! only when speedyml_state%host_mirror is set (the reference's per-call semantics, at the price of a device synchronisation per
! region) or through hip_fetch(reservoir).
!
! What stays with the reference: file I/O.  The data readers (speedy_res_interface::read_era / read_model_states) and the NetCDF
! helpers of mod_io are called with the reference's signatures; the repository ships synthetic stand-ins of exactly the called
! procedures for its own test (fortran/test_support.f90), a maintainer links the reference's modules instead.
module mod_reservoir
  use iso_c_binding
  use speedyml_hip
  use speedyml_state
  use mod_utilities, only : dp, main_type, reservoir_type, grid_type, model_parameters_type, era_data_type, speedy_data_type, &
                            standardize_data_given_pars5d, standardize_data_given_pars_5d_logp, standardize_data_given_pars_5d_logp_tisr, &
                            standardize_data_given_pars3d, total_precip_over_a_period
  implicit none
  integer :: global_time_step

contains

  subroutine initialize_model_parameters(model_parameters, processor, num_of_procs)
    use mpires, only : distribute_prediction_marker
    type(model_parameters_type), intent(inout) :: model_parameters
    integer, intent(in) :: processor, num_of_procs
    ! the shipped configuration (src/mod_reservoir.f90:24-77)
    model_parameters%ml_only = .false.; model_parameters%ml_only_ocean = .true.
    model_parameters%num_predictions = 1
    model_parameters%trial_name = '6000_20_20_20_sigma0.5_beta_res0.001_beta_model_1.0_prior_0.0_overlap1_vertlevel_1_precip_epsilon0.001_ohtc_test'
    model_parameters%trial_name_extra_end = ''
    model_parameters%discardlength = 24 * 10
    model_parameters%traininglength = 12000
    model_parameters%predictionlength = 8760 * 20
    model_parameters%synclength = 24 * 14
    model_parameters%timestep = 6
    model_parameters%timestep_slab = 24 * 7
    global_time_step = model_parameters%timestep
    model_parameters%slab_ocean_model_bool = .true.
    model_parameters%train_on_sst_anomalies = .false.
    model_parameters%ohtc_bool_input = .true.
    model_parameters%non_stationary_ocn_climo = .false.
    model_parameters%final_sst_bias = 2.0_dp
    model_parameters%precip_bool = .true.
    model_parameters%precip_epsilon = 0.001
    model_parameters%timeofday_bool = .false.
    model_parameters%full_predictvars = 4
    model_parameters%full_heightlevels = 8
    model_parameters%num_vert_levels = 1
    model_parameters%vert_loc_overlap = 0
    model_parameters%overlap = 1
    model_parameters%irank = processor
    model_parameters%numprocs = num_of_procs
    model_parameters%noisy = .true.
    model_parameters%regional_vary = .true.
    model_parameters%using_prior = .true.
    model_parameters%model_noise = 0.0_dp
    model_parameters%outvec_component_contribs = .false.
    model_parameters%special_reservoirs = .false.
    model_parameters%num_special_reservoirs = 0
    model_parameters%run_speedy = .true.
    call distribute_prediction_marker(model_parameters)
  end subroutine

  ! the flags train_reservoir / trained_reservoir_prediction set before sizing (:228-246, :1796-1812)
  subroutine set_level_flags(reservoir, grid, model_parameters)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    reservoir%tisr_input_bool = .true.
    reservoir%sst_climo_bool = .false.
    if (grid%bottom) then
      reservoir%logp_bool = .true.; grid%logp_bool = .true.
      reservoir%sst_bool = model_parameters%slab_ocean_model_bool
      reservoir%precip_input_bool = model_parameters%precip_bool
      reservoir%precip_bool = model_parameters%precip_bool
    else
      reservoir%logp_bool = .false.; grid%logp_bool = .false.
      reservoir%sst_bool = .false.
      reservoir%precip_input_bool = .false.
      reservoir%precip_bool = .false.
    end if
    reservoir%local_predictvars = model_parameters%full_predictvars
    reservoir%local_heightlevels_input = grid%inputzchunk
    reservoir%local_heightlevels_res = grid%reszchunk
  end subroutine

  subroutine allocate_res_new(reservoir, grid, model_parameters)
    use resdomain, only : set_reservoir_by_region
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    type(sml_region) :: g
    type(sml_res_sizes) :: s
    character(len=32) :: env
    integer :: mlen, stat
    reservoir%m = 6000
    call get_environment_variable('SML_RES_M', env, mlen, stat)          ! tests: smaller reservoirs (not a reference parameter)
    if (stat == 0 .and. mlen > 0) read(env(1:mlen), *) reservoir%m
    reservoir%deg = 6; reservoir%radius = 0.9_dp
    reservoir%beta_res = 0.001_dp; reservoir%beta_model = 1.0_dp
    reservoir%sigma = 0.5_dp; reservoir%leakage = 1.0_dp; reservoir%prior_val = 0.0_dp
    reservoir%density = reservoir%deg / real(reservoir%m, kind=dp)
    call set_reservoir_by_region(reservoir, grid)
    g%res_xstart = grid%res_xstart; g%res_xend = grid%res_xend; g%res_ystart = grid%res_ystart; g%res_yend = grid%res_yend
    g%resxchunk = grid%resxchunk; g%resychunk = grid%resychunk; g%res_zstart = grid%res_zstart; g%res_zend = grid%res_zend
    g%reszchunk = grid%reszchunk; g%input_xstart = grid%input_xstart; g%input_xend = grid%input_xend
    g%input_ystart = grid%input_ystart; g%input_yend = grid%input_yend; g%inputxchunk = grid%inputxchunk; g%inputychunk = grid%inputychunk
    g%input_zstart = grid%input_zstart; g%input_zend = grid%input_zend; g%inputzchunk = grid%inputzchunk
    g%pole = merge(1, 0, grid%pole); g%periodicboundary = merge(1, 0, grid%periodicboundary)
    g%top = merge(1, 0, grid%top); g%bottom = merge(1, 0, grid%bottom)
    g%tdata_xstart = grid%tdata_xstart; g%tdata_xend = grid%tdata_xend; g%tdata_ystart = grid%tdata_ystart; g%tdata_yend = grid%tdata_yend
    g%tdata_zstart = grid%tdata_zstart; g%tdata_zend = grid%tdata_zend
    call sml_check(sml_domain_sizes(g, int(reservoir%m, c_int), int(reservoir%deg, c_int), int(reservoir%local_predictvars, c_int), &
                                    merge(1_c_int, 0_c_int, reservoir%logp_bool), merge(1_c_int, 0_c_int, reservoir%precip_input_bool), &
                                    merge(1_c_int, 0_c_int, reservoir%sst_bool_input), merge(1_c_int, 0_c_int, reservoir%tisr_input_bool), &
                                    merge(1_c_int, 0_c_int, model_parameters%ml_only), s), 'sml_domain_sizes')
    reservoir%logp_size_input = merge(grid%inputxchunk * grid%inputychunk, 0, reservoir%logp_bool)
    reservoir%logp_size_res = merge(grid%resxchunk * grid%resychunk, 0, reservoir%logp_bool)
    reservoir%sst_size_input = merge(grid%inputxchunk * grid%inputychunk, 0, reservoir%sst_bool_input)
    reservoir%sst_size_res = merge(grid%resxchunk * grid%resychunk, 0, reservoir%sst_bool_input)
    reservoir%precip_size_input = merge(grid%inputxchunk * grid%inputychunk, 0, reservoir%precip_input_bool)
    reservoir%precip_size_res = merge(grid%resxchunk * grid%resychunk, 0, reservoir%precip_input_bool)
    reservoir%tisr_size_input = merge(grid%inputxchunk * grid%inputychunk, 0, reservoir%tisr_input_bool)
    reservoir%tisr_size_res = merge(grid%resxchunk * grid%resychunk, 0, reservoir%tisr_input_bool)
    reservoir%chunk_size = s%chunk_size; reservoir%chunk_size_prediction = s%chunk_size_prediction
    reservoir%chunk_size_speedy = s%chunk_size_speedy; reservoir%locality = s%locality
    reservoir%n = s%n; reservoir%k = s%k; reservoir%reservoir_numinputs = s%reservoir_numinputs
    grid%atmo3d_start = s%atmo3d_start; grid%atmo3d_end = s%atmo3d_end; grid%logp_start = s%logp_start; grid%logp_end = s%logp_end
    grid%precip_start = s%precip_start; grid%precip_end = s%precip_end; grid%sst_start = s%sst_start; grid%sst_end = s%sst_end
    grid%tisr_start = s%tisr_start; grid%tisr_end = s%tisr_end
    grid%predict_start = 1
    grid%predict_end = merge(merge(s%precip_end, s%logp_end, reservoir%precip_bool), s%atmo3d_end, reservoir%logp_bool)
    ! the statistics' slots (src/mod_reservoir.f90:1851-1885): 3-d variables first, then logp, tisr, precip, sst
    grid%logp_mean_std_idx = reservoir%local_predictvars * reservoir%local_heightlevels_input + 1
    grid%tisr_mean_std_idx = grid%logp_mean_std_idx + 1
    grid%precip_mean_std_idx = grid%logp_mean_std_idx + 2
    grid%sst_mean_std_idx = grid%logp_mean_std_idx + 3
    if (.not. allocated(reservoir%vals)) allocate(reservoir%vals(reservoir%k))
    if (.not. allocated(reservoir%win)) allocate(reservoir%win(reservoir%n, reservoir%reservoir_numinputs))
    if (.not. allocated(reservoir%wout)) allocate(reservoir%wout(reservoir%chunk_size_prediction, reservoir%n + reservoir%chunk_size_speedy))
    if (.not. allocated(reservoir%rows)) allocate(reservoir%rows(reservoir%k))
    if (.not. allocated(reservoir%cols)) allocate(reservoir%cols(reservoir%k))
  end subroutine

  ! gen_res (:182-212): random sparse adjacency rescaled to the spectral radius
  subroutine gen_res(reservoir)
    type(reservoir_type), intent(inout) :: reservoir
    real(c_double) :: eigs
    call sml_check(sml_gen_res(int(reservoir%n, c_int), int(reservoir%k, c_int), reservoir%radius, &
                               int(20240000 + reservoir%assigned_region, c_int64_t), reservoir%rows, reservoir%cols, reservoir%vals, eigs), 'sml_gen_res')
  end subroutine

  ! ---- the rank's bank: created at the first reservoir, one slot per (region, level) of this rank ----
  subroutine bank_slot_for(reservoir, model_parameters)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    if (.not. c_associated(hip_bank)) then
      hip_capacity = model_parameters%num_of_regions_on_proc * model_parameters%num_vert_levels
      call sml_check(sml_bank_create(int(hip_capacity, c_int), 576_c_int, 132_c_int, 136_c_int, hip_bank), 'sml_bank_create')
      allocate(region_of_slot(hip_capacity), sst_input_of_slot(hip_capacity), slot_predicted(hip_capacity))
      slot_predicted = .false.
      hip_loaded = 0
    end if
    if (reservoir%hip_slot < 0) then
      if (hip_loaded >= hip_capacity) stop 'mod_reservoir: more reservoirs than num_of_regions_on_proc * num_vert_levels'
      reservoir%hip_slot = hip_loaded
      hip_loaded = hip_loaded + 1
    end if
    region_of_slot(reservoir%hip_slot + 1) = reservoir%assigned_region
    sst_input_of_slot(reservoir%hip_slot + 1) = merge(1, 0, reservoir%sst_bool_input)
  end subroutine

  ! mklsparse (src/mod_linalg.f90:10-25) + the upload of W_in, W_out and the statistics: the reservoir becomes resident
  subroutine load_into_bank(reservoir, grid, model_parameters)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(in) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer(c_int), allocatable :: gidx(:), stat(:)
    integer(c_int) :: cnt
    call bank_slot_for(reservoir, model_parameters)
    allocate(gidx(reservoir%chunk_size_prediction), stat(reservoir%chunk_size_prediction))
    cnt = sml_domain_out_map(int(grid%number_of_regions, c_int), int(reservoir%assigned_region, c_int), int(grid%num_vert_levels, c_int), &
                             int(grid%level_index, c_int), int(grid%vert_overlap, c_int), merge(1_c_int, 0_c_int, reservoir%precip_bool), &
                             gidx, stat, int(size(gidx), c_int))
    call sml_check(cnt, 'sml_domain_out_map')
    call sml_check(sml_bank_load(hip_bank, reservoir%hip_slot, int(reservoir%n, c_int), int(reservoir%reservoir_numinputs, c_int), &
                                 int(reservoir%k, c_int), int(reservoir%chunk_size_speedy, c_int), int(reservoir%chunk_size_prediction, c_int), &
                                 reservoir%rows, reservoir%cols, reservoir%vals, reservoir%win, reservoir%wout, reservoir%leakage, &
                                 grid%mean, grid%std, int(size(grid%mean), c_int), stat), 'sml_bank_load')
  end subroutine

  ! ---- data: get_prediction_data (:622-790) / get_training_data (:322-605) over the reference's readers ----
  subroutine fill_from_era(reservoir, model_parameters, grid, start_index, length, inputs, model_states, compute_stats)
    use mod_calendar
    use speedy_res_interface, only : read_era, read_model_states
    use resdomain, only : standardize_speedy_data
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    integer, intent(in) :: start_index, length
    real(kind=dp), allocatable, intent(inout) :: inputs(:,:), model_states(:,:)
    logical, intent(in), optional :: compute_stats
    type(era_data_type) :: era
    type(speedy_data_type) :: spd
    integer :: hours0, start_year, t0, t1, step, ncol, natm
    call get_current_time_delta_hour(calendar, start_index)
    call numof_hours_into_year(calendar%currentyear, calendar%currentmonth, calendar%currentday, calendar%currenthour, hours0)
    start_year = calendar%currentyear
    call get_current_time_delta_hour(calendar, start_index + length)
    call read_era(reservoir, grid, model_parameters, start_year, calendar%currentyear, era, 1)
    t0 = hours0; t1 = t0 + length; step = model_parameters%timestep; ncol = length / step
    if (t1 > size(era%eravariables, 5)) then
      write(*,'(a,i0,a,i0,a)') ' mod_reservoir: the window ends at hour ', t1, ' of the data read_era returned, which hold ', size(era%eravariables, 5), ' hours'
      stop 1
    end if
    ! units and floors as the reference applies them (:660-690): q in g/kg with a floor, no negative radiation or rain, SST >= 272 K,
    ! precipitation accumulated over a time step and log-transformed
    era%eravariables(4,:,:,:,:) = max(era%eravariables(4,:,:,:,:) * 1000.0_dp, 0.000001_dp)
    if (reservoir%tisr_input_bool) era%era_tisr = max(era%era_tisr, 0.0_dp)
    if (reservoir%sst_bool .and. .not. model_parameters%train_on_sst_anomalies) era%era_sst = max(era%era_sst, 272.0_dp)
    if (reservoir%precip_bool) then
      era%era_precip = max(era%era_precip, 0.0_dp)
      call total_precip_over_a_period(era%era_precip, step)
      era%era_precip = log(1 + era%era_precip / model_parameters%precip_epsilon)
    end if
    if (present(compute_stats)) then
      if (compute_stats) call region_statistics(reservoir, grid, era)
    end if
    if (reservoir%tisr_input_bool .and. reservoir%logp_bool) then
      call standardize_data_given_pars_5d_logp_tisr(grid%mean, grid%std, era%eravariables, era%era_logp, era%era_tisr)
    else if (reservoir%logp_bool) then
      call standardize_data_given_pars_5d_logp(grid%mean, grid%std, era%eravariables, era%era_logp)
    else if (reservoir%tisr_input_bool) then
      call standardize_data_given_pars_5d_logp(grid%mean, grid%std, era%eravariables, era%era_tisr)
    else
      call standardize_data_given_pars5d(grid%mean, grid%std, era%eravariables)
    end if
    if (reservoir%sst_bool_input) call standardize_data_given_pars3d(era%era_sst, grid%mean(grid%sst_mean_std_idx), grid%std(grid%sst_mean_std_idx))
    if (reservoir%precip_bool) call standardize_data_given_pars3d(era%era_precip, grid%mean(grid%precip_mean_std_idx), grid%std(grid%precip_mean_std_idx))
    if (allocated(inputs)) deallocate(inputs)
    allocate(inputs(reservoir%reservoir_numinputs, ncol))
    inputs(grid%atmo3d_start:grid%atmo3d_end, :) = reshape(era%eravariables(:,:,:,:,t0:t1:step), [grid%atmo3d_end, ncol])
    if (reservoir%logp_bool) inputs(grid%logp_start:grid%logp_end, :) = reshape(era%era_logp(:,:,t0:t1:step), [reservoir%logp_size_input, ncol])
    if (reservoir%precip_bool) inputs(grid%precip_start:grid%precip_end, :) = reshape(era%era_precip(:,:,t0:t1:step), [reservoir%precip_size_input, ncol])
    if (reservoir%sst_bool_input) inputs(grid%sst_start:grid%sst_end, :) = reshape(era%era_sst(:,:,t0:t1:step), [reservoir%sst_size_input, ncol])
    if (reservoir%tisr_input_bool) inputs(grid%tisr_start:grid%tisr_end, :) = reshape(era%era_tisr(:,:,t0:t1:step), [reservoir%tisr_size_input, ncol])
    if (.not. model_parameters%ml_only) then
      call read_model_states(reservoir, grid, model_parameters, start_year, calendar%currentyear, spd, 1)
      spd%speedyvariables(4,:,:,:,:) = max(spd%speedyvariables(4,:,:,:,:), 0.000001_dp)
      call standardize_speedy_data(reservoir, grid, spd)
      if (allocated(model_states)) deallocate(model_states)
      allocate(model_states(reservoir%chunk_size_speedy, ncol))
      model_states = 0.0_dp
      natm = reservoir%local_predictvars * grid%resxchunk * grid%resychunk * grid%reszchunk
      model_states(1:natm, :) = reshape(spd%speedyvariables(:,:,:,:,t0:t1:step), [natm, ncol])
      if (reservoir%logp_bool) model_states(natm+1:reservoir%chunk_size_speedy, :) = reshape(spd%speedy_logp(:,:,t0:t1:step), [grid%resxchunk * grid%resychunk, ncol])
      end if
  end subroutine

  ! the statistics standardize_data leaves in grid%mean / grid%std (src/mod_reservoir.f90:443-470): per 3-d variable and level,
  ! then logp, tisr, precip, sst; a region takes SST as an input when it varies there (std > 0.2, :1843-1847)
  subroutine region_statistics(reservoir, grid, era)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(era_data_type), intent(in) :: era
    integer :: v, z, l, nl
    nl = size(era%eravariables, 1) * size(era%eravariables, 4)
    if (allocated(grid%mean)) deallocate(grid%mean, grid%std)
    allocate(grid%mean(nl + 4), grid%std(nl + 4))
    grid%mean = 0.0_dp; grid%std = 1.0_dp
    l = 0
    do v = 1, size(era%eravariables, 1)
      do z = 1, size(era%eravariables, 4)
        l = l + 1
        call mean_std(reshape(era%eravariables(v,:,:,z,:), [size(era%eravariables(v,:,:,z,:))]), grid%mean(l), grid%std(l))
      end do
    end do
    if (allocated(era%era_logp)) call mean_std(reshape(era%era_logp, [size(era%era_logp)]), grid%mean(nl+1), grid%std(nl+1))
    if (allocated(era%era_tisr)) call mean_std(reshape(era%era_tisr, [size(era%era_tisr)]), grid%mean(nl+2), grid%std(nl+2))
    if (allocated(era%era_precip)) call mean_std(reshape(era%era_precip, [size(era%era_precip)]), grid%mean(nl+3), grid%std(nl+3))
    if (allocated(era%era_sst)) then
      call mean_std(reshape(era%era_sst, [size(era%era_sst)]), grid%mean(nl+4), grid%std(nl+4))
      reservoir%sst_bool_input = reservoir%sst_bool .and. grid%std(nl+4) > 0.2_dp
      if (.not. reservoir%sst_bool_input) grid%std(nl+4) = 1.0_dp
    end if
  contains
    subroutine mean_std(a, m, s)
      real(kind=dp), intent(in) :: a(:)
      real(kind=dp), intent(out) :: m, s
      m = sum(a) / size(a)
      s = sqrt(sum((a - m)**2) / size(a))
      if (s <= 0.0_dp) s = 1.0_dp
    end subroutine
  end subroutine

  subroutine get_prediction_data(reservoir, model_parameters, grid, start_index, length)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    integer, intent(in) :: start_index, length
    call fill_from_era(reservoir, model_parameters, grid, start_index, length, reservoir%predictiondata, reservoir%imperfect_model_states)
  end subroutine

  ! get_training_data (:322-605): the hourly training window of the region, its statistics, the sizes that follow from them
  subroutine get_training_data(reservoir, model_parameters, grid, loop_index)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    integer, intent(in) :: loop_index
    real(kind=dp), allocatable :: hourly(:,:), hourly_model(:,:)
    integer :: keep
    logical :: had_sst
    ! sizes with every optional input present, so that the segments exist while the data are laid out; SST may drop out below
    reservoir%sst_bool_input = reservoir%sst_bool
    had_sst = reservoir%sst_bool_input
    call allocate_res_new(reservoir, grid, model_parameters)
    keep = model_parameters%timestep
    model_parameters%timestep = 1                                      ! the training arrays are hourly (train_reservoir strides them)
    call fill_from_era(reservoir, model_parameters, grid, 0, model_parameters%traininglength, hourly, hourly_model, compute_stats=.true.)
    if (had_sst .and. .not. reservoir%sst_bool_input) then            ! a land region: no SST segment after all (:1843-1847)
      deallocate(reservoir%vals, reservoir%win, reservoir%wout, reservoir%rows, reservoir%cols)
      call allocate_res_new(reservoir, grid, model_parameters)
      call fill_from_era(reservoir, model_parameters, grid, 0, model_parameters%traininglength, hourly, hourly_model)
    end if
    model_parameters%timestep = keep
    call move_alloc(hourly, reservoir%trainingdata)
    if (allocated(hourly_model)) call move_alloc(hourly_model, reservoir%imperfect_model_states)
  end subroutine

  ! ---- training ----
  ! noisy copy of one input column: gaussian_noise_1d_function / gaussian_noise_1d_function_precip (src/mod_utilities.f90:1387-1464) --
  ! g ~ N(0,1) per entry in entry order (Box-Muller on RANDOM_NUMBER), x + g noisemag x; with precip_bool the precipitation segment is
  ! un-standardised, taken out of log space, perturbed, made non-negative, and taken back
  subroutine noisy_column(x, noisemag, grid, model_parameters, with_precip, out)
    use mod_utilities, only : box_muller
    real(kind=dp), intent(in) :: x(:), noisemag
    type(grid_type), intent(in) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    logical, intent(in) :: with_precip
    real(kind=dp), intent(out) :: out(:)
    real(kind=dp), allocatable :: g(:), t(:)
    real(kind=dp), parameter :: e_constant = 2.7182818284590452353602874_dp
    integer :: i, a, b
    allocate(g(size(x)))
    do i = 1, size(x)
      g(i) = box_muller()
    end do
    out = x + g * noisemag * x
    if (with_precip) then
      a = grid%precip_start; b = grid%precip_end
      t = x(a:b)
      t = t * grid%std(grid%precip_mean_std_idx) + grid%mean(grid%precip_mean_std_idx)
      t = model_parameters%precip_epsilon * (e_constant**t - 1)
      t = t + g(a:b) * noisemag * t
      t = abs(t)
      t = log(1 + t / model_parameters%precip_epsilon)
      t = t - grid%mean(grid%precip_mean_std_idx)
      out(a:b) = t / grid%std(grid%precip_mean_std_idx)
    end if
  end subroutine

  ! train_reservoir (:214-320).  The reservoir is built here (data, A, W_in) and ENQUEUED for training with everything the device
  ! needs (speedyml_train): the recurrences of up to SML_TRAIN_RESIDENTS reservoirs share their per-column launches and the ridge
  ! systems of a size class are solved in lockstep.  The queue runs when it is full, when the rank's last reservoir has arrived, or
  ! when a result is needed (finish_training, called by every procedure that comes after training in program main).
  subroutine train_reservoir(reservoir, grid, model_parameters)
    use speedyml_train
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(inout) :: model_parameters
    type(train_job) :: job
    real(kind=dp), allocatable :: rand(:), pass_in(:,:)
    integer(c_int), allocatable :: tpos(:)
    integer(c_int) :: ntarg
    integer :: q, i, c, ncol, d, nm, no, step
    call set_level_flags(reservoir, grid, model_parameters)
    call get_training_data(reservoir, model_parameters, grid, 1)
    call gen_res(reservoir)
    q = reservoir%n / reservoir%reservoir_numinputs
    allocate(rand(q))
    reservoir%win = 0.0_dp
    do i = 1, reservoir%reservoir_numinputs                             ! W_in: q nodes per input (:262-283)
      call random_number(rand)
      reservoir%win((i-1)*q+1:i*q, i) = reservoir%sigma * (-1.0_dp + 2.0_dp * rand)
    end do
    d = reservoir%reservoir_numinputs; nm = reservoir%chunk_size_speedy; no = reservoir%chunk_size_prediction
    if (model_parameters%ml_only) nm = 0
    step = model_parameters%timestep
    ! initialize_chunk_training (:1561-1592): 20 batches per pass, batch size the closest divisor
    ncol = model_parameters%traininglength / step
    reservoir%batch_size = sml_find_closest_divisor(int((model_parameters%traininglength - model_parameters%discardlength) / (20 * step), c_int), &
                                                    int((model_parameters%traininglength - model_parameters%discardlength) / step, c_int))
    reservoir%wout = 0.0_dp
    call load_into_bank(reservoir, grid, model_parameters)               ! resident for the prediction that follows; W_out arrives with the flush
    allocate(tpos(no))
    ntarg = sml_domain_target_map(int(grid%number_of_regions, c_int), int(reservoir%assigned_region, c_int), int(grid%overlap, c_int), &
                                  int(grid%num_vert_levels, c_int), int(grid%level_index, c_int), int(grid%vert_overlap, c_int), &
                                  merge(1_c_int, 0_c_int, reservoir%precip_bool), tpos, int(no, c_int))
    call sml_check(ntarg, 'sml_domain_target_map')
    job%n = reservoir%n; job%d = d; job%k = reservoir%k; job%n_model = nm; job%n_out = no
    job%discard = model_parameters%discardlength / step; job%batch = reservoir%batch_size
    job%ml_variant = merge(1, 0, model_parameters%ml_only); job%using_prior = merge(1, 0, model_parameters%using_prior)
    job%leakage = reservoir%leakage; job%beta_res = reservoir%beta_res; job%beta_model = reservoir%beta_model; job%prior_val = reservoir%prior_val
    job%rows = reservoir%rows; job%cols = reservoir%cols; job%vals = reservoir%vals; job%win = reservoir%win
    job%mean = grid%mean; job%std = grid%std
    job%bank = hip_bank; job%slot = reservoir%hip_slot
    call train_job_passes(job, step, ncol)
    do i = 1, step                                                       ! the interleaved passes (:298-305)
      pass_in = reservoir%trainingdata(:, i:model_parameters%traininglength:step)
      job%ncol(i) = ncol
      job%noisy(:, :, i) = pass_in
      if (model_parameters%noisy) then                                   ! one noisy copy per column the recurrence reads (:1091-1166)
        do c = 1, ncol - 1
          call noisy_column(pass_in(:, c), reservoir%noisemag, grid, model_parameters, model_parameters%precip_bool .and. reservoir%precip_bool, &
                            job%noisy(:, c, i))
        end do
      end if
      job%targ(:, :, i) = pass_in(tpos(1:no) + 1, :)                     ! chunking_matmul's targets (tile_full_input_to_target_data)
      if (nm > 0) job%mdl(:, :, i) = reservoir%imperfect_model_states(:, i:model_parameters%traininglength:step)
    end do
    reservoir%hip_train_job = train_enqueue(job)
    if (.not. (model_parameters%slab_ocean_model_bool .and. grid%bottom)) deallocate(reservoir%trainingdata)
    if (allocated(reservoir%imperfect_model_states)) deallocate(reservoir%imperfect_model_states)
    ! the rank's last reservoir: nothing else will join the queue
    if (hip_loaded == hip_capacity) call train_flush()
  end subroutine

  ! W_out of a reservoir that went through the training queue: into reservoir%wout and the weights file (write_trained_res, :1330)
  subroutine finish_training(reservoir, model_parameters, grid)
    use speedyml_train
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    if (reservoir%hip_train_job <= 0) return
    call train_take(reservoir%hip_train_job, reservoir%wout)
    reservoir%hip_train_job = 0
    call write_trained_res(reservoir, model_parameters, grid)
  end subroutine

  ! write_trained_res (:1703-1737) through the reference's NetCDF helpers
  subroutine write_trained_res(reservoir, model_parameters, grid)
    use mod_io, only : write_netcdf_2d_non_met_data, write_netcdf_1d_non_met_data_int, write_netcdf_1d_non_met_data_real
    type(reservoir_type), intent(in) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    character(len=4) :: worker_char
    character(len=1) :: height_char
    character(len=:), allocatable :: fname
    write(worker_char, '(i0.4)') reservoir%assigned_region
    write(height_char, '(i0.1)') grid%level_index
    fname = 'worker_' // worker_char // '_level_' // height_char // '_' // trim(model_parameters%trial_name) // '.nc'
    call write_netcdf_2d_non_met_data(reservoir%win, 'win', fname, 'unitless', 'win_x', 'win_y')
    call write_netcdf_2d_non_met_data(reservoir%wout, 'wout', fname, 'unitless', 'wout_x', 'wout_y')
    call write_netcdf_1d_non_met_data_int(reservoir%rows, 'rows', fname, 'unitless', 'rows_x')
    call write_netcdf_1d_non_met_data_int(reservoir%cols, 'cols', fname, 'unitless', 'cols_x')
    call write_netcdf_1d_non_met_data_real(reservoir%vals, 'vals', fname, 'unitless', 'vals_x')
    call write_netcdf_1d_non_met_data_real(grid%mean, 'mean', fname, 'unitless', 'mean_x')
    call write_netcdf_1d_non_met_data_real(grid%std, 'std', fname, 'unitless', 'std_x')
  end subroutine

  ! trained_reservoir_prediction (:1783-1886): sizes, the trained arrays from their file, residency
  subroutine trained_reservoir_prediction(reservoir, model_parameters, grid)
    use mod_io, only : read_trained_res
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    call set_level_flags(reservoir, grid, model_parameters)
    call read_trained_res(reservoir, model_parameters, grid)            ! win, wout, rows, cols, vals, grid%mean, grid%std; sets sst_bool_input
    call allocate_res_new(reservoir, grid, model_parameters)
    call load_into_bank(reservoir, grid, model_parameters)
  end subroutine

  ! ---- prediction ----
  subroutine synchronize(reservoir, input, x, length)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(in) :: input(:,:)
    real(kind=dp), intent(inout) :: x(:)
    integer, intent(in) :: length
    real(kind=dp), allocatable :: cols(:,:)
    cols = input(:, 1:length)                                            ! contiguous d x length block
    call sml_check(sml_bank_synchronize_one(hip_bank, reservoir%hip_slot, cols, int(length, c_int), x), 'sml_bank_synchronize_one')
  end subroutine

  subroutine synchronize_print(reservoir, grid, input, x, length)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(in) :: input(:,:)
    real(kind=dp), intent(inout) :: x(:)
    integer, intent(in) :: length
    call synchronize(reservoir, input, x, length)
  end subroutine

  subroutine initialize_prediction(reservoir, model_parameters, grid)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    integer, parameter :: un_noisy_sync = 2160
    call finish_training(reservoir, model_parameters, grid)
    if (.not. allocated(reservoir%saved_state)) allocate(reservoir%saved_state(reservoir%n))
    reservoir%saved_state = 0
    call get_prediction_data(reservoir, model_parameters, grid, model_parameters%traininglength - un_noisy_sync, un_noisy_sync)
    call synchronize(reservoir, reservoir%predictiondata, reservoir%saved_state, un_noisy_sync / model_parameters%timestep - 1)
    if (.not. (model_parameters%slab_ocean_model_bool .and. grid%bottom)) deallocate(reservoir%predictiondata)
    if (.not. allocated(reservoir%local_model)) allocate(reservoir%local_model(reservoir%chunk_size_speedy))
    if (.not. allocated(reservoir%outvec)) allocate(reservoir%outvec(reservoir%chunk_size_prediction))
    if (.not. allocated(reservoir%feedback)) allocate(reservoir%feedback(reservoir%reservoir_numinputs))
    if (.not. allocated(reservoir%current_state)) allocate(reservoir%current_state(reservoir%n))
  end subroutine

  subroutine start_prediction(reservoir, model_parameters, grid, prediction_number)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    integer, intent(in) :: prediction_number
    integer :: nsync
    model_parameters%current_trial_number = prediction_number
    call get_prediction_data(reservoir, model_parameters, grid, &
                             model_parameters%traininglength + model_parameters%prediction_markers(prediction_number), model_parameters%synclength + 100)
    nsync = model_parameters%synclength / model_parameters%timestep
    call synchronize_print(reservoir, grid, reservoir%predictiondata(:, 1:nsync-1), reservoir%saved_state, nsync - 1)
    reservoir%feedback = reservoir%predictiondata(:, nsync)
    call sml_check(sml_bank_set_feedback(hip_bank, reservoir%hip_slot, reservoir%feedback), 'sml_bank_set_feedback')
    if (.not. model_parameters%ml_only) then
      reservoir%local_model = reservoir%imperfect_model_states(:, nsync + 1)
      call sml_check(sml_bank_set_local_model(hip_bank, reservoir%hip_slot, reservoir%local_model), 'sml_bank_set_local_model')
    end if
    ! (program main copies saved_state into current_state right after this call; the device copy is what predict advances)
    call sml_check(sml_bank_set_state(hip_bank, reservoir%hip_slot, reservoir%saved_state), 'sml_bank_set_state')
  end subroutine

  ! predict / predict_ml: the first call of a time step advances and reads out EVERY resident reservoir
  subroutine batched_predict(reservoir, x, contribs)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(inout) :: x(:)
    logical, intent(in) :: contribs
    integer :: s
    s = reservoir%hip_slot + 1
    if (slot_predicted(s) .or. hip_predicted == 0) then                  ! a slot seen twice, or nobody yet: a new time step begins
      slot_predicted = .false.
      hip_predicted = 0
      call sml_check(sml_bank_predict_all(hip_bank, 0_c_int, c_null_ptr), 'sml_bank_predict_all')
      ! outvec_component_contribs (src/mod_reservoir.f90:1458-1461): the two column blocks of the readout apart, for every slot at once
      if (contribs) call sml_check(sml_bank_outvec_contribs(hip_bank, c_null_ptr), 'sml_bank_outvec_contribs')
    end if
    slot_predicted(s) = .true.
    hip_predicted = hip_predicted + 1
    if (hip_predicted == hip_loaded) hip_predicted = 0
    if (host_mirror) call hip_fetch(reservoir, x)
    if (contribs) then          ! reservoir%v_p, reservoir%v_ml (standardised, as the reference leaves them)
      if (.not. allocated(reservoir%v_p)) allocate(reservoir%v_p(reservoir%chunk_size_prediction), reservoir%v_ml(reservoir%chunk_size_prediction))
      call sml_check(sml_bank_get_contribs(hip_bank, reservoir%hip_slot, reservoir%v_p, reservoir%v_ml), 'sml_bank_get_contribs')
    end if
  end subroutine

  ! reservoir%outvec (un-standardised) and the state x of this reservoir from the device
  subroutine hip_fetch(reservoir, x)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(inout) :: x(:)
    call sml_check(sml_bank_get_outvec(hip_bank, reservoir%hip_slot, reservoir%outvec), 'sml_bank_get_outvec')
    call sml_check(sml_bank_get_state(hip_bank, reservoir%hip_slot, x), 'sml_bank_get_state')
  end subroutine

  subroutine predict(reservoir, model_parameters, grid, x, local_model_in)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    real(kind=dp), intent(inout) :: x(:)
    real(kind=dp), intent(inout) :: local_model_in(:)
    call batched_predict(reservoir, x, model_parameters%outvec_component_contribs)
  end subroutine

  subroutine predict_ml(reservoir, model_parameters, grid, x)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    real(kind=dp), intent(inout) :: x(:)
    call batched_predict(reservoir, x, .false.)          ! (predict_ml has no model block: src/mod_reservoir.f90:1491-1535)
  end subroutine

end module mod_reservoir
