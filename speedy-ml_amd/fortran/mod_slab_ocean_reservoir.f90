! Module mod_slab_ocean_reservoir of the MI355X drop-in (src/mod_slab_ocean_reservoir.f90): every procedure program main imports
! (src/parallelmain.f90:8) with the reference's argument lists --
!     initialize_slab_ocean_model(reservoir,grid,model_parameters)                                            :9-133
!     train_slab_ocean_model(reservoir,grid,model_parameters)                                                 :172-269
!     get_training_data_from_atmo(reservoir,model_parameters,grid,reservoir_atmo,grid_atmo)                   :271-399
!     initialize_prediction_slab(reservoir,model_parameters,grid,atmo_reservoir,atmo_grid)                    :764-815
!     start_prediction_slab(reservoir,model_parameters,grid,atmo_reservoir,atmo_grid,prediction_number)       :833-867
!     predict_slab(reservoir,model_parameters,grid,x,local_model_in) / predict_slab_ml(reservoir,...,x)       :1268-1363
!     trained_ocean_reservoir_prediction(reservoir,model_parameters,grid,reservoir_atmo,grid_atmo)            :1561-1647
! The slab reservoirs of the rank live in a second bank in HBM (speedyml_state%hip_slab_bank), slot i beside slot i of the atmosphere
! bank (the same region), so that the coupling -- SST assembly from the slab outputs, the 27-step averaging ring of the slab inputs,
! src/mpires.f90:286-330,470-484,756-790 -- runs inside the hybrid engine (sml_hybrid_attach_slab, called by mpires).  The per-region
! predict_slab_ml calls of a slab step are served by ONE batched predict of the whole slab bank, as predict is for the atmosphere.
! What stays with the reference: the data files (mod_io::read_trained_ocean_res, read_3d_file_parallel for the ocean heat content).
module mod_slab_ocean_reservoir
  use iso_c_binding
  use speedyml_hip
  use speedyml_state
  use mod_utilities, only : dp, reservoir_type, grid_type, model_parameters_type
  implicit none
  character(len=*), parameter :: ohtc_file = '/scratch/user/troyarcomano/ORAS5/regridded_sohtc300_control_monthly_highres_2D_CONS_v0.1_hourly_gcc.nc'
contains

  ! initialize_slab_ocean_model (:9-133): the shipped hyper-parameters and every derived size (sml_slab_sizes, bit-exact integers)
  subroutine initialize_slab_ocean_model(reservoir, grid, model_parameters)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(inout) :: model_parameters
    type(sml_region) :: g
    type(sml_res_sizes) :: s
    character(len=32) :: env
    integer :: mlen, stat, in2d, res2d
    reservoir%local_predictvars = model_parameters%full_predictvars
    reservoir%local_heightlevels_input = grid%inputzchunk
    reservoir%local_heightlevels_res = grid%reszchunk
    model_parameters%ml_only_ocean = .true.
    reservoir%m = 4000
    call get_environment_variable('SML_SLAB_M', env, mlen, stat)         ! tests: smaller reservoirs (not a reference parameter)
    if (stat == 0 .and. mlen > 0) read(env(1:mlen), *) reservoir%m
    reservoir%deg = 6; reservoir%radius = 0.9_dp
    reservoir%beta_res = 0.0001_dp; reservoir%beta_model = 1.0_dp
    reservoir%sigma = 0.6_dp; reservoir%prior_val = 0.0_dp
    reservoir%density = reservoir%deg / reservoir%m
    reservoir%noisemag = 0.10_dp; reservoir%leakage = 1.0_dp
    reservoir%sst_bool = .true.; reservoir%sst_bool_prediction = .true.; reservoir%sst_bool_input = .true.
    reservoir%tisr_input_bool = .true.; reservoir%atmo_to_ocean_coupled = .true.
    reservoir%ohtc_prediction = .true.
    reservoir%sst_climo_input = .false.
    reservoir%precip_input_bool = .false.
    reservoir%num_atmo_levels = 1
    in2d = grid%inputxchunk * grid%inputychunk; res2d = grid%resxchunk * grid%resychunk
    reservoir%sst_size_res = res2d; reservoir%sst_size_input = in2d; reservoir%sst_climo_res = 0
    reservoir%tisr_size_res = res2d; reservoir%tisr_size_input = in2d
    reservoir%atmo_size_input = in2d * reservoir%local_predictvars + in2d
    reservoir%ohtc_input_size = in2d; reservoir%ohtc_res_size = res2d
    g%resxchunk = grid%resxchunk; g%resychunk = grid%resychunk; g%inputxchunk = grid%inputxchunk; g%inputychunk = grid%inputychunk
    call sml_check(sml_slab_sizes(g, int(reservoir%m, c_int), int(reservoir%deg, c_int), int(reservoir%local_predictvars, c_int), s), 'sml_slab_sizes')
    reservoir%chunk_size_speedy = 0
    reservoir%chunk_size = s%chunk_size; reservoir%chunk_size_prediction = s%chunk_size_prediction; reservoir%locality = s%locality
    reservoir%n = s%n; reservoir%k = s%k; reservoir%reservoir_numinputs = s%reservoir_numinputs
    if (.not. allocated(reservoir%vals)) allocate(reservoir%vals(reservoir%k))
    if (.not. allocated(reservoir%win)) allocate(reservoir%win(reservoir%n, reservoir%reservoir_numinputs))
    if (.not. allocated(reservoir%wout)) allocate(reservoir%wout(reservoir%chunk_size_prediction, reservoir%n + reservoir%chunk_size_speedy))
    if (.not. allocated(reservoir%rows)) allocate(reservoir%rows(reservoir%k))
    if (.not. allocated(reservoir%cols)) allocate(reservoir%cols(reservoir%k))
  end subroutine

  ! the u(t) segments of a slab reservoir and the entries of the atmosphere reservoir's input they are taken from
  ! (get_training_data_from_atmo :330-378, trained_ocean_reservoir_prediction :1603-1645)
  subroutine slab_segments(reservoir, grid, reservoir_atmo, grid_atmo, with_sst_tisr_idx)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(reservoir_type), intent(in) :: reservoir_atmo
    type(grid_type), intent(in) :: grid_atmo
    logical, intent(in) :: with_sst_tisr_idx
    integer :: in2d, total, i, counter
    in2d = grid_atmo%inputxchunk * grid_atmo%inputychunk
    total = in2d * reservoir_atmo%local_predictvars + 3 * in2d
    if (reservoir%ohtc_prediction) total = total + in2d
    grid%atmo3d_start = grid_atmo%atmo3d_start
    grid%atmo3d_end = in2d * reservoir_atmo%local_predictvars
    grid%logp_start = grid%atmo3d_end + 1
    grid%logp_end = in2d * reservoir_atmo%local_predictvars + in2d
    grid%sst_start = grid%logp_end + 1
    grid%sst_end = grid%sst_start + in2d - 1
    grid%tisr_start = grid%sst_end + 1
    grid%tisr_end = grid%tisr_start + in2d - 1
    if (reservoir%ohtc_prediction) then
      grid%ohtc_start = grid%tisr_end + 1
      grid%ohtc_end = grid%ohtc_start + in2d - 1
    end if
    grid%sst_mean_std_idx = grid_atmo%sst_mean_std_idx
    if (allocated(reservoir%atmo_training_data_idx)) deallocate(reservoir%atmo_training_data_idx)
    allocate(reservoir%atmo_training_data_idx(total))
    reservoir%atmo_training_data_idx = 0
    counter = 0
    do i = grid_atmo%atmo3d_end - in2d * reservoir_atmo%local_predictvars + 1, grid_atmo%logp_end       ! lowest level + logp
      counter = counter + 1
      reservoir%atmo_training_data_idx(counter) = i
    end do
    if (with_sst_tisr_idx) then
      do i = grid_atmo%sst_start, grid_atmo%sst_end
        counter = counter + 1
        reservoir%atmo_training_data_idx(counter) = i
      end do
      do i = grid_atmo%tisr_start, grid_atmo%tisr_end
        counter = counter + 1
        reservoir%atmo_training_data_idx(counter) = i
      end do
    end if
  end subroutine

  ! ocean heat content of the region's input patch, hourly (read_ohtc_parallel_training / _prediction, :1649-1745): the reference's
  ! reader with its arguments; the file starts on 16 January 1979
  subroutine read_ohtc(model_parameters, grid, ohtc_var, hours_from_1981, length)
    use mpires, only : mpi_res
    use mod_io, only : read_3d_file_parallel
    use mod_utilities, only : calendar_type
    use mod_calendar
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    real(kind=dp), allocatable, intent(out) :: ohtc_var(:,:,:)
    integer, intent(in) :: hours_from_1981, length
    type(calendar_type) :: ohtc_calendar
    integer :: start_index
    call initialize_calendar(ohtc_calendar, 1979, 1, 16, 0)
    call get_current_time_delta_hour(ohtc_calendar, 0)
    call get_current_time_delta_hour(calendar, hours_from_1981)
    call time_delta_between_two_dates_datetime_type(ohtc_calendar, calendar, start_index)
    call read_3d_file_parallel(ohtc_file, 'sohtc300', mpi_res, grid, ohtc_var, start_index, 1, length)
    where (abs(ohtc_var) > 10.0_dp**13) ohtc_var = 0.0_dp
  end subroutine

  ! get_training_data_from_atmo (:271-399): the slab reservoir's training inputs are rows of the atmosphere reservoir's (lowest level,
  ! logp, SST, TISR, running-mean over a slab step for the atmosphere part) plus the ocean heat content
  subroutine get_training_data_from_atmo(reservoir, model_parameters, grid, reservoir_atmo, grid_atmo)
    use mod_utilities, only : rolling_average_over_a_period_2d, standardize_data_3d
    type(reservoir_type), intent(inout) :: reservoir, reservoir_atmo
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid, grid_atmo
    real(kind=dp), allocatable :: ohtc_var(:,:,:)
    integer :: a0, ncol
    grid%mean = grid_atmo%mean
    grid%std = grid_atmo%std
    reservoir%hip_slot = reservoir_atmo%hip_slot                       ! the slab reservoir sits beside its atmosphere reservoir
    reservoir%sst_bool_input = reservoir_atmo%sst_bool_input
    reservoir%sst_bool_prediction = reservoir_atmo%sst_bool_input
    reservoir%ohtc_prediction = reservoir_atmo%sst_bool_input .and. model_parameters%ohtc_bool_input
    ncol = size(reservoir_atmo%trainingdata, 2)
    ! (the reference's parallel reader must be called by every rank, sea or not)
    if (model_parameters%ohtc_bool_input) call read_ohtc(model_parameters, grid, ohtc_var, model_parameters%traininglength + model_parameters%synclength, ncol)
    if (reservoir%sst_bool_prediction) then
      reservoir%sst_climo_input = .false.
      reservoir%precip_input_bool = .false.
      reservoir%local_predictvars = reservoir_atmo%local_predictvars
      call slab_segments(reservoir, grid, reservoir_atmo, grid_atmo, .true.)
      if (reservoir%ohtc_prediction) then
        grid%ohtc_mean_std_idx = 1                                        ! (overwrites the first atmosphere statistic: top-level temperature)
        call standardize_data_3d(ohtc_var, grid%mean(grid%ohtc_mean_std_idx), grid%std(grid%ohtc_mean_std_idx))
      end if
      allocate(reservoir%trainingdata(size(reservoir%atmo_training_data_idx), ncol))
      a0 = grid_atmo%atmo3d_end - grid_atmo%inputxchunk * grid_atmo%inputychunk * reservoir_atmo%local_predictvars + 1
      reservoir%trainingdata(grid%atmo3d_start:grid%logp_end, :) = reservoir_atmo%trainingdata(a0:grid_atmo%logp_end, :)
      reservoir%trainingdata(grid%sst_start:grid%sst_end, :) = reservoir_atmo%trainingdata(grid_atmo%sst_start:grid_atmo%sst_end, :)
      reservoir%trainingdata(grid%tisr_start:grid%tisr_end, :) = reservoir_atmo%trainingdata(grid_atmo%tisr_start:grid_atmo%tisr_end, :)
      if (reservoir%ohtc_prediction) &
        reservoir%trainingdata(grid%ohtc_start:grid%ohtc_end, :) = reshape(ohtc_var(:,:,1:ncol), [grid_atmo%inputxchunk * grid_atmo%inputychunk, ncol])
      call rolling_average_over_a_period_2d(reservoir%trainingdata(grid%atmo3d_start:grid%logp_end, :), model_parameters%timestep_slab)
    end if
    deallocate(reservoir_atmo%trainingdata)
  end subroutine

  ! the 0-based rows of a slab input column that are its targets: SST, then ocean heat content, of the res patch inside the input patch
  ! (tile_full_input_to_target_data_ocean_model, src/res_domain.f90:691-764)
  subroutine slab_target_rows(reservoir, grid, rows)
    type(reservoir_type), intent(in) :: reservoir
    type(grid_type), intent(in) :: grid
    integer, allocatable, intent(out) :: rows(:)
    integer :: x, y, c, seg, nseg, start
    allocate(rows(reservoir%chunk_size_prediction))
    nseg = merge(2, 1, reservoir%ohtc_prediction)
    c = 0
    do seg = 1, nseg
      start = merge(grid%sst_start, grid%ohtc_start, seg == 1)
      do y = grid%tdata_ystart, grid%tdata_yend
        do x = grid%tdata_xstart, grid%tdata_xend
          c = c + 1
          rows(c) = start - 1 + (y - 1) * grid%inputxchunk + (x - 1)
        end do
      end do
    end do
  end subroutine

  ! train_slab_ocean_model (:172-269): gen_res, W_in, timestep_slab interleaved passes of reservoir_layer_chunking_ml (:869-957; the
  ! hybrid-ocean variant with the persistence forecast as imperfect model, :211-232, when ml_only_ocean is off), fit_chunk_ml (:1061-1101)
  subroutine train_slab_ocean_model(reservoir, grid, model_parameters)
    use speedyml_train
    use mod_utilities, only : box_muller
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid
    type(train_job) :: job
    real(kind=dp), allocatable :: rand(:), pass_in(:,:), persistence(:,:)
    integer, allocatable :: trows(:)
    real(c_double) :: eigs
    integer :: q, i, c, r, ncol, d, no, nm, step, total
    call sml_check(sml_gen_res(int(reservoir%n, c_int), int(reservoir%k, c_int), reservoir%radius, &
                               int(30240000 + reservoir%assigned_region, c_int64_t), reservoir%rows, reservoir%cols, reservoir%vals, eigs), 'sml_gen_res')
    q = reservoir%n / reservoir%reservoir_numinputs
    allocate(rand(q))
    reservoir%win = 0.0_dp
    do i = 1, reservoir%reservoir_numinputs
      call random_number(rand)
      reservoir%win((i-1)*q+1:i*q, i) = reservoir%sigma * (-1.0_dp + 2.0_dp * rand)
    end do
    d = reservoir%reservoir_numinputs; no = reservoir%chunk_size_prediction
    nm = merge(0, no, model_parameters%ml_only_ocean)
    step = model_parameters%timestep_slab
    total = model_parameters%traininglength
    ! initialize_chunk_training (:1389-1418): ONE batch per pass, its size the closest divisor
    reservoir%batch_size = sml_find_closest_divisor(int((total - model_parameters%discardlength) / step, c_int), &
                                                    int((total - model_parameters%discardlength) / step, c_int))
    call slab_target_rows(reservoir, grid, trows)
    if (nm > 0) then                                                     ! persistence: the targets one slab step earlier (:211-218)
      allocate(persistence(no, size(reservoir%trainingdata, 2)))
      persistence(:, 1:step) = reservoir%trainingdata(trows + 1, 1:step)
      persistence(:, step+1:) = reservoir%trainingdata(trows + 1, 1:size(reservoir%trainingdata, 2) - step)
    end if
    job%n = reservoir%n; job%d = d; job%k = reservoir%k; job%n_model = nm; job%n_out = no
    job%discard = model_parameters%discardlength / step; job%batch = reservoir%batch_size
    job%ml_variant = merge(1, 0, model_parameters%ml_only_ocean)
    job%using_prior = merge(0, merge(1, 0, model_parameters%using_prior), model_parameters%ml_only_ocean)     ! fit_chunk_ml has no prior: beta_res as it is
    job%leakage = reservoir%leakage; job%beta_res = reservoir%beta_res; job%beta_model = reservoir%beta_model; job%prior_val = reservoir%prior_val
    job%rows = reservoir%rows; job%cols = reservoir%cols; job%vals = reservoir%vals; job%win = reservoir%win
    job%mean = grid%mean; job%std = grid%std
    call train_job_passes(job, step, (total - 1) / step + 1)
    do i = 1, step
      pass_in = reservoir%trainingdata(:, i:total:step)
      ncol = size(pass_in, 2)
      job%ncol(i) = ncol
      job%noisy(:, 1:ncol, i) = pass_in
      do c = 1, ncol - 1                                                  ! gaussian_noise_1d_function per column read (:893,:918)
        do r = 1, d
          job%noisy(r, c, i) = pass_in(r, c) + box_muller() * reservoir%noisemag * pass_in(r, c)
        end do
      end do
      job%targ(:, 1:ncol, i) = pass_in(trows + 1, :)
      if (nm > 0) job%mdl(:, 1:ncol, i) = persistence(:, i:total:step)
    end do
    ! resident beside its atmosphere reservoir from now on; W_out arrives when the training queue runs (speedyml_train)
    reservoir%wout = 0.0_dp
    call load_slab_reservoir(reservoir, grid, model_parameters, reservoir%hip_slot)
    job%bank = hip_slab_bank; job%slot = reservoir%hip_slot
    if (.not. allocated(reservoir%saved_state)) allocate(reservoir%saved_state(reservoir%n))
    reservoir%saved_state = 0.0_dp
    reservoir%hip_train_job = train_enqueue(job)
  end subroutine

  ! W_out of a slab reservoir that went through the training queue: into reservoir%wout and its weights file (fit_chunk_ml :1097)
  subroutine finish_slab_training(reservoir, model_parameters, grid)
    use speedyml_train
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    if (reservoir%hip_train_job <= 0) return
    call train_take(reservoir%hip_train_job, reservoir%wout)
    reservoir%hip_train_job = 0
    call write_trained_res(reservoir, model_parameters, grid)
  end subroutine

  ! write_trained_res (:1531-1559): worker_RRRR_ocean_<trial>.nc through the reference's NetCDF helpers
  subroutine write_trained_res(reservoir, model_parameters, grid)
    use mod_io, only : write_netcdf_2d_non_met_data, write_netcdf_1d_non_met_data_int, write_netcdf_1d_non_met_data_real
    type(reservoir_type), intent(in) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    character(len=4) :: worker_char
    character(len=:), allocatable :: fname
    write(worker_char, '(i0.4)') reservoir%assigned_region
    fname = 'worker_' // worker_char // '_ocean_' // trim(model_parameters%trial_name) // '.nc'
    call write_netcdf_2d_non_met_data(reservoir%win, 'win', fname, 'unitless', 'win_x', 'win_y')
    call write_netcdf_2d_non_met_data(reservoir%wout, 'wout', fname, 'unitless', 'wout_x', 'wout_y')
    call write_netcdf_1d_non_met_data_int(reservoir%rows, 'rows', fname, 'unitless', 'rows_x')
    call write_netcdf_1d_non_met_data_int(reservoir%cols, 'cols', fname, 'unitless', 'cols_x')
    call write_netcdf_1d_non_met_data_real(reservoir%vals, 'vals', fname, 'unitless', 'vals_x')
    call write_netcdf_1d_non_met_data_real(grid%mean, 'mean', fname, 'unitless', 'mean_x')
    call write_netcdf_1d_non_met_data_real(grid%std, 'std', fname, 'unitless', 'std_x')
  end subroutine

  ! residency: the slab bank of the rank (created at the first slab reservoir), slot = the region's slot in the atmosphere bank;
  ! every output is un-standardised with the SST statistics (predict_slab_ml :1354)
  subroutine load_slab_reservoir(reservoir, grid, model_parameters, slot)
    use mpires, only : ensure_slab_bank
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(in) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer(c_int), intent(in) :: slot
    integer(c_int), allocatable :: stat(:)
    integer :: nm
    if (slot < 0 .or. slot >= hip_capacity) stop 'mod_slab_ocean_reservoir: the atmosphere reservoir of the region must be resident first'
    call ensure_slab_bank()
    reservoir%hip_slot = slot
    nm = merge(0, reservoir%chunk_size_prediction, model_parameters%ml_only_ocean)
    slab_hybrid_ocean = nm > 0
    allocate(stat(reservoir%chunk_size_prediction))
    stat = int(grid%sst_mean_std_idx - 1, c_int)
    call sml_check(sml_bank_load(hip_slab_bank, slot, int(reservoir%n, c_int), int(reservoir%reservoir_numinputs, c_int), int(reservoir%k, c_int), &
                                 int(nm, c_int), int(reservoir%chunk_size_prediction, c_int), reservoir%rows, reservoir%cols, reservoir%vals, &
                                 reservoir%win, reservoir%wout, reservoir%leakage, grid%mean, grid%std, int(size(grid%mean), c_int), stat), 'sml_bank_load')
    if (slab_sea_of_slot(slot + 1) == 0) slab_loaded = slab_loaded + 1
    slab_sea_of_slot(slot + 1) = 1
  end subroutine

  ! trained_ocean_reservoir_prediction (:1561-1647): the trained arrays from worker_RRRR_ocean_<trial>.nc, sizes, residency
  subroutine trained_ocean_reservoir_prediction(reservoir, model_parameters, grid, reservoir_atmo, grid_atmo)
    use mod_io, only : read_trained_ocean_res
    type(reservoir_type), intent(inout) :: reservoir, reservoir_atmo
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid, grid_atmo
    call read_trained_ocean_res(reservoir, model_parameters, grid)       ! win wout rows cols vals, grid%mean / std; sets sst_bool_input / _prediction
    reservoir%sst_bool_prediction = reservoir%sst_bool_input
    reservoir%ohtc_prediction = reservoir%sst_bool_input .and. model_parameters%ohtc_bool_input
    if (.not. reservoir%sst_bool_prediction) return
    reservoir%sst_climo_input = .false.
    reservoir%precip_input_bool = .false.
    ! only the atmosphere entries are averaged over a slab step during prediction (:1624-1640; SST and TISR entries are left out)
    call slab_segments(reservoir, grid, reservoir_atmo, grid_atmo, .false.)
    if (reservoir%ohtc_prediction) grid%ohtc_mean_std_idx = 1
    call initialize_slab_ocean_model(reservoir, grid, model_parameters)
    call load_slab_reservoir(reservoir, grid, model_parameters, reservoir_atmo%hip_slot)
  end subroutine

  ! get_prediction_data_from_atmo (:401-482): slab inputs every timestep_slab / timestep-th column of the atmosphere reservoir's
  ! prediction data.  (As shipped the running mean is applied to rows atmo3d_start:logp_end of the ATMOSPHERE array -- its first rows,
  ! the top model level -- while the rows then taken are the lowest level's: reproduced.)
  subroutine get_prediction_data_from_atmo(reservoir, model_parameters, grid, reservoir_atmo, grid_atmo, start_idx, delete_atmo_data)
    use mod_utilities, only : rolling_average_over_a_period_2d, standardize_data_given_pars3d
    type(reservoir_type), intent(inout) :: reservoir, reservoir_atmo
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid, grid_atmo
    integer, intent(in) :: start_idx
    logical, intent(in), optional :: delete_atmo_data
    real(kind=dp), allocatable :: temp(:,:), ohtc_var(:,:,:)
    integer :: ratio, num_syncs, a0, ncol
    ncol = size(reservoir_atmo%predictiondata, 2)
    if (model_parameters%ohtc_bool_input) call read_ohtc(model_parameters, grid, ohtc_var, start_idx, ncol * model_parameters%timestep)
    if (.not. reservoir%sst_bool_prediction) return
    ratio = model_parameters%timestep_slab / model_parameters%timestep
    num_syncs = (ncol - 1) / ratio + 1
    if (allocated(reservoir%predictiondata)) then
      if (size(reservoir%predictiondata, 2) /= num_syncs) deallocate(reservoir%predictiondata)
    end if
    if (.not. allocated(reservoir%predictiondata)) allocate(reservoir%predictiondata(reservoir%reservoir_numinputs, num_syncs))
    temp = reservoir_atmo%predictiondata
    call rolling_average_over_a_period_2d(temp(grid%atmo3d_start:grid%logp_end, :), ratio)
    a0 = grid_atmo%atmo3d_end - grid_atmo%inputxchunk * grid_atmo%inputychunk * reservoir_atmo%local_predictvars + 1
    reservoir%predictiondata(grid%atmo3d_start:grid%logp_end, :) = temp(a0:grid_atmo%logp_end, 1:ncol:ratio)
    reservoir%predictiondata(grid%sst_start:grid%sst_end, :) = temp(grid_atmo%sst_start:grid_atmo%sst_end, 1:ncol:ratio)
    reservoir%predictiondata(grid%tisr_start:grid%tisr_end, :) = temp(grid_atmo%tisr_start:grid_atmo%tisr_end, 1:ncol:ratio)
    if (reservoir%ohtc_prediction) then
      call standardize_data_given_pars3d(ohtc_var, grid%mean(grid%ohtc_mean_std_idx), grid%std(grid%ohtc_mean_std_idx))
      reservoir%predictiondata(grid%ohtc_start:grid%ohtc_end, :) = &
        reshape(ohtc_var(:,:,1:size(ohtc_var, 3):model_parameters%timestep_slab), [reservoir%ohtc_input_size, num_syncs])
    end if
    if (.not. present(delete_atmo_data)) deallocate(reservoir_atmo%predictiondata)
  end subroutine

  ! synchronize (:1237-1266) for this slab reservoir: `length` teacher-forced steps, x in and out
  subroutine synchronize(reservoir, input, x, length)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(in) :: input(:,:)
    real(kind=dp), intent(inout) :: x(:)
    integer, intent(in) :: length
    real(kind=dp), allocatable :: cols(:,:)
    if (length <= 0) return
    cols = input(:, 1:length)
    call sml_check(sml_bank_synchronize_one(hip_slab_bank, reservoir%hip_slot, cols, int(length, c_int), x), 'sml_bank_synchronize_one')
  end subroutine

  ! initialize_prediction_slab (:764-815)
  subroutine initialize_prediction_slab(reservoir, model_parameters, grid, atmo_reservoir, atmo_grid)
    type(reservoir_type), intent(inout) :: reservoir, atmo_reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid, atmo_grid
    integer, parameter :: un_noisy_sync = 2160
    call finish_slab_training(reservoir, model_parameters, grid)
    call get_prediction_data_from_atmo(reservoir, model_parameters, grid, atmo_reservoir, atmo_grid, model_parameters%traininglength - un_noisy_sync)
    if (.not. reservoir%sst_bool_prediction) return
    if (.not. allocated(reservoir%saved_state)) allocate(reservoir%saved_state(reservoir%n))
    reservoir%saved_state = 0
    call synchronize(reservoir, reservoir%predictiondata, reservoir%saved_state, un_noisy_sync / model_parameters%timestep_slab - 1)
    if (reservoir%tisr_input_bool .and. allocated(atmo_reservoir%full_tisr)) reservoir%full_tisr = atmo_reservoir%full_tisr
    deallocate(reservoir%predictiondata)
    if (.not. allocated(reservoir%local_model)) allocate(reservoir%local_model(reservoir%chunk_size_prediction))
    if (.not. allocated(reservoir%outvec)) allocate(reservoir%outvec(reservoir%chunk_size_prediction))
    if (.not. allocated(reservoir%feedback)) allocate(reservoir%feedback(reservoir%reservoir_numinputs))
    if (.not. allocated(reservoir%current_state)) allocate(reservoir%current_state(reservoir%n))
    if (.not. allocated(reservoir%averaged_atmo_input_vec)) &
      allocate(reservoir%averaged_atmo_input_vec(grid%logp_end, model_parameters%timestep_slab / model_parameters%timestep - 1))
    reservoir%averaged_atmo_input_vec = 0.0_dp
  end subroutine

  ! start_prediction_slab (:833-867): the sync of the forecast window; the slab reservoir's first feedback and -- until its first
  ! own prediction, timestep_slab hours in -- the SST of the analysis as its output
  subroutine start_prediction_slab(reservoir, model_parameters, grid, atmo_reservoir, atmo_grid, prediction_number)
    type(reservoir_type), intent(inout) :: reservoir, atmo_reservoir
    type(model_parameters_type), intent(inout) :: model_parameters
    type(grid_type), intent(inout) :: grid, atmo_grid
    integer, intent(in) :: prediction_number
    integer, allocatable :: trows(:)
    integer :: nsync
    integer(c_int64_t) :: off
    model_parameters%current_trial_number = prediction_number
    call get_prediction_data_from_atmo(reservoir, model_parameters, grid, atmo_reservoir, atmo_grid, &
                                       model_parameters%traininglength + model_parameters%prediction_markers(prediction_number), .false.)
    if (.not. reservoir%sst_bool_prediction) return
    nsync = model_parameters%synclength / model_parameters%timestep_slab
    call synchronize(reservoir, reservoir%predictiondata(:, 1:nsync), reservoir%saved_state, nsync)
    reservoir%feedback = reservoir%predictiondata(:, nsync)
    call slab_target_rows(reservoir, grid, trows)
    reservoir%outvec = reservoir%predictiondata(trows + 1, nsync)
    if (.not. model_parameters%ml_only_ocean) reservoir%local_model = reservoir%outvec
    reservoir%outvec = reservoir%outvec * grid%std(grid%sst_mean_std_idx) + grid%mean(grid%sst_mean_std_idx)
    ! device copies: state, feedback, the output row the engine's SST assembly reads, the hybrid ocean's local_model
    call sml_check(sml_bank_set_state(hip_slab_bank, reservoir%hip_slot, reservoir%saved_state), 'sml_bank_set_state')
    call sml_check(sml_bank_set_feedback(hip_slab_bank, reservoir%hip_slot, reservoir%feedback), 'sml_bank_set_feedback')
    off = 8_c_int64_t * slab_max_out * reservoir%hip_slot
    call sml_check(sml_dev_upload_off(sml_bank_outvec_dev(hip_slab_bank), off, reservoir%outvec, 8_c_int64_t * size(reservoir%outvec)), 'sml_dev_upload')
    if (.not. model_parameters%ml_only_ocean) &
      call sml_check(sml_bank_set_local_model(hip_slab_bank, reservoir%hip_slot, reservoir%local_model), 'sml_bank_set_local_model')
    slab_predicted = .false.; slab_done = 0
  end subroutine

  ! predict_slab_ml / predict_slab: the first call of a slab step advances and reads out EVERY resident slab reservoir
  subroutine batched_slab_predict(reservoir, x)
    type(reservoir_type), intent(inout) :: reservoir
    real(kind=dp), intent(inout) :: x(:)
    integer :: s
    s = reservoir%hip_slot + 1
    if (slab_predicted(s) .or. slab_done == 0) then
      slab_predicted = .false.
      slab_done = 0
      if (slab_hybrid_ocean) then
        call sml_check(sml_slab_predict_hybrid(hip_slab_bank, c_null_ptr), 'sml_slab_predict_hybrid')
      else
        call sml_check(sml_bank_predict_all(hip_slab_bank, 0_c_int, c_null_ptr), 'sml_bank_predict_all')
      end if
    end if
    slab_predicted(s) = .true.
    slab_done = slab_done + 1
    if (slab_done == slab_loaded) slab_done = 0
    if (host_mirror) then
      call sml_check(sml_bank_get_outvec(hip_slab_bank, reservoir%hip_slot, reservoir%outvec), 'sml_bank_get_outvec')
      call sml_check(sml_bank_get_state(hip_slab_bank, reservoir%hip_slot, x), 'sml_bank_get_state')
    end if
  end subroutine

  subroutine predict_slab_ml(reservoir, model_parameters, grid, x)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(inout) :: x(:)
    call batched_slab_predict(reservoir, x)
  end subroutine

  subroutine predict_slab(reservoir, model_parameters, grid, x, local_model_in)
    type(reservoir_type), intent(inout) :: reservoir
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(in) :: grid
    real(kind=dp), intent(inout) :: x(:)
    real(kind=dp), intent(inout) :: local_model_in(:)
    call batched_slab_predict(reservoir, x)
  end subroutine

end module mod_slab_ocean_reservoir
