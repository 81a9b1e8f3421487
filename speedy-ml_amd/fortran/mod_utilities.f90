! Module mod_utilities of the MI355X drop-in: the derived types that are the data contract of the reference's module API
! (src/mod_utilities.f90:32-598) with the field names program main and the module procedures touch, and the host-side
! helpers those procedures use.  Differences from the reference's types, all in reservoir_type: the MKL handles cooA / descrA
! (:188-189) are replaced by the slot of the device-resident bank (hip_slot) and the device Gram matrices (hip_c, hip_b);
! model_parameters_type has no opened_netcdf_files (NetCDF stays with the reference's mod_io).
module mod_utilities
  use iso_c_binding
  implicit none
  integer, parameter :: dp = c_double
  integer, parameter :: xgrid = 96, ygrid = 48, zgrid = 8         ! T30L8 (src/mod_utilities.f90:12-14)

  type grid_type
    ! where the reservoir's prediction (res) patch and its input patch sit in the global grid, 1-based inclusive
    integer :: res_xstart, res_xend, res_ystart, res_yend, res_zstart, res_zend, resxchunk, resychunk, reszchunk
    integer :: tdata_xstart, tdata_xend, tdata_ystart, tdata_yend, tdata_zstart, tdata_zend
    integer :: input_xstart, input_xend, input_ystart, input_yend, input_zstart, input_zend, inputxchunk, inputychunk, inputzchunk
    logical :: pole, periodicboundary, top_vert_level, bottom_vert_level
    integer :: overlap, num_vert_levels, vert_overlap
    character(len=:), allocatable :: region_char
    real(kind=dp), allocatable :: mean(:), std(:)                 ! per physical component: (var-1)*8+level, logp, tisr, precip, sst
    integer :: tisr_mean_std_idx, logp_mean_std_idx, sst_mean_std_idx, precip_mean_std_idx, ohtc_mean_std_idx
    integer :: number_of_regions
    logical :: top, bottom
    integer :: level_index
    logical :: logp_bool
    ! segments of u(t) = (atmosphere | logp | precip | sst | tisr | ohtc), 1-based inclusive
    integer :: atmo3d_start, atmo3d_end, sst_start, sst_end, logp_start, logp_end, precip_start, precip_end
    integer :: tisr_start, tisr_end, predict_start, predict_end, ohtc_start, ohtc_end
  end type grid_type

  type reservoir_type
    integer :: assigned_region
    integer, allocatable :: vert_indices_res(:), vert_indices_input(:)
    real(kind=dp), allocatable :: trainingdata(:,:)
    ! device residency (in place of cooA / descrA): the slot of this reservoir in the rank's bank, the Gram matrices of training
    integer(c_int) :: hip_slot = -1
    integer :: hip_train_job = 0                                  ! its entry in the training queue while W_out is still on its way
    type(c_ptr) :: hip_c = c_null_ptr, hip_b = c_null_ptr
    real(kind=dp) :: deg, radius, beta_res, beta_model, density, sigma, leakage
    integer, allocatable :: rows(:), cols(:)
    real(kind=dp), allocatable :: vals(:)
    integer :: k, reservoir_numinputs, locality, m, n
    real(kind=dp), allocatable :: win(:,:), wout(:,:), states(:,:), augmented_states(:,:)
    integer :: batch_size
    real(kind=dp), allocatable :: states_x_states(:,:), states_x_trainingdata(:,:), states_x_states_aug(:,:), states_x_trainingdata_aug(:,:)
    integer :: local_heightlevels_res, local_heightlevels_input, local_predictvars, logp_size_res, logp_size_input
    logical :: logp_bool
    real(kind=dp), allocatable :: saved_state(:), current_state(:)
    logical :: tisr_input_bool
    integer :: tisr_size_input, tisr_size_res
    logical :: precip_bool, precip_input_bool
    integer :: precip_size_res, precip_size_input
    logical :: sst_bool, sst_bool_input, sst_bool_prediction
    integer :: sst_size_res, sst_size_input
    logical :: sst_climo_bool, sst_climo_input
    integer :: sst_climo_res
    logical :: atmo_to_ocean_coupled
    integer :: atmo_size_input, num_atmo_levels
    logical :: ohtc_prediction
    integer :: ohtc_res_size, ohtc_input_size
    integer, allocatable :: atmo_training_data_idx(:)
    real(kind=dp), allocatable :: averaged_atmo_input_vec(:,:)
    integer :: chunk_size, chunk_size_prediction, chunk_size_speedy
    real(kind=dp), allocatable :: imperfect_model_states(:,:), predictiondata(:,:)
    real(kind=dp) :: noisemag, prior_val
    real(kind=dp), allocatable :: local_model(:), outvec(:), v_ml(:), v_p(:), feedback(:)
    real(kind=dp), allocatable :: full_tisr(:,:,:), full_sst(:,:,:)
    integer :: predictvars2d
  end type reservoir_type

  type model_parameters_type
    logical :: ml_only, ml_only_ocean
    integer :: num_vert_levels, vert_loc_overlap, number_of_regions, num_of_regions_on_proc
    integer, allocatable :: region_indices(:)
    integer :: full_heightlevels, full_predictvars
    integer :: traininglength, discardlength, synclength, predictionlength, overlap
    integer, allocatable :: prediction_markers(:)
    integer :: num_predictions, current_trial_number, irank, numprocs
    real(kind=dp), allocatable :: prediction(:,:)
    logical :: specific_humidity_log_bool
    real(kind=dp) :: specific_humidity_epsilon = 0.3_dp
    logical :: pole_only
    character(len=3) :: trial_number
    character(len=10) :: trial_date
    character(len=:), allocatable :: trial_name, trial_name_extra_end
    logical :: run_speedy, timeofday_bool, regional_vary, using_prior
    real(kind=dp) :: model_noise
    integer :: timestep, timestep_slab
    logical :: toa_isr_bool, precip_bool
    real :: precip_epsilon
    logical :: noisy
    character(len=:), allocatable :: prediction_file
    logical :: special_reservoirs
    integer :: num_special_reservoirs
    logical :: slab_ocean_model_bool, ohtc_bool_input, train_on_sst_anomalies
    real(kind=dp), allocatable :: base_sst_grid(:,:), sea_mask(:,:)
    logical :: non_stationary_ocn_climo
    real(kind=dp) :: final_sst_bias, current_sst_bias
    logical :: outvec_component_contribs
  end type model_parameters_type

  type main_type
    type(grid_type), allocatable :: grid(:,:)
    type(reservoir_type), allocatable :: reservoir(:,:)
    type(grid_type), allocatable :: grid_special(:,:)
    type(reservoir_type), allocatable :: reservoir_special(:,:)
    type(model_parameters_type) :: model_parameters
  end type main_type

  type speedy_data_type
    real(kind=dp), allocatable :: speedyvariables(:,:,:,:,:), speedy_logp(:,:,:)
  end type speedy_data_type

  type era_data_type
    real(kind=dp), allocatable :: eravariables(:,:,:,:,:), era_logp(:,:,:), era_tisr(:,:,:), era_sst(:,:,:), era_sst_climo(:,:,:)
    real(kind=dp), allocatable :: era_precip(:,:,:), era_ohtc(:,:,:)
  end type era_data_type

  type state_vector_type
    real(kind=dp), allocatable :: variables3d(:,:,:,:), logp(:,:)
    integer :: istart, era_start
    character(len=100) :: era_file
    integer :: era_hour, era_hour_plus_one, iyear0, imont0, iday, ihour
    logical :: is_safe_to_run_speedy, hybrid_slab
    real(kind=dp), allocatable :: sst_hybrid(:,:)
    real(kind=dp) :: sst_bias = 0.0_dp
  end type state_vector_type

  type calendar_type                                              ! src/mod_utilities.f90 calendar_type
    integer :: startyear, startmonth, startday, starthour
    integer :: currentyear, currentmonth, currentday, currenthour
  end type calendar_type

  type opened_netcdf_type                                         ! src/mod_utilities.f90:631-638 (an argument type of read_era_netcdf_opened)
    logical :: is_opened
    logical :: is_closed
    integer :: ncid
    character(len=:), allocatable :: filename
  end type opened_netcdf_type

  type mpi_type
    integer :: ierr, numprocs, proc_num
    logical :: is_root = .false., is_serial = .false.
    integer :: mpi_world = 0
  end type mpi_type

  ! the generic names program main imports (src/mod_utilities.f90:641-663)
  interface standardize_data
    module procedure standardize_data_1d, standardize_data_2d, standardize_data_3d, standardize_data_4d, standardize_data_5d, &
                     standardize_data_5d_logp, standardize_data_5d_logp_tisr
  end interface
  interface gaussian_noise
    module procedure gaussian_noise_2d, gaussian_noise_1d
  end interface

contains

  ! a different random seed on every worker (src/mod_utilities.f90 init_random_marker)
  ! seed word i = (3 + 2 input)(i - 1): the same on every run (src/mod_utilities.f90:1553-1567)
  subroutine init_random_marker(input)
    integer, intent(in) :: input
    integer :: nseed, i
    integer, allocatable :: seed(:)
    call random_seed(size=nseed)
    allocate(seed(nseed))
    seed = (3 + input * 2) * [(i - 1, i = 1, nseed)]
    call random_seed(put=seed)
  end subroutine

  ! seed word i = clock + (18 + 12 worker)(i - 1): different on every run and worker (src/mod_utilities.f90:1535-1551)
  subroutine init_random_seed(worker)
    integer, intent(in) :: worker
    integer :: nseed, clock, i
    integer, allocatable :: seed(:)
    call random_seed(size=nseed)
    allocate(seed(nseed))
    call system_clock(count=clock)
    seed = clock + (18 + worker * 12) * [(i - 1, i = 1, nseed)]
    call random_seed(put=seed)
  end subroutine

  ! ---- standardize_data: statistics of the data itself, returned, then (x - mean) / std as two statements ----
  ! 1-d .. 4-d (src/mod_utilities.f90:833-983): ONE statistic over the whole array, the variance in its one-pass form
  ! (sum x^2 - (sum x)^2 / N) / N;  5-d forms (:934-1193): one statistic per (variable, level) in two-pass form, then the 2-d fields
  subroutine one_pass_stats(total, total_sq, count, mean, std)
    real(kind=dp), intent(in) :: total, total_sq
    integer, intent(in) :: count
    real(kind=dp), intent(out) :: mean, std
    mean = total / count
    std = sqrt((total_sq - total**2 / count) / count)
  end subroutine

  subroutine standardize_data_1d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:)
    real(kind=dp), intent(out) :: mean, std
    call one_pass_stats(sum(inputdata), sum(inputdata**2), size(inputdata), mean, std)
    call standardize_data_given_pars1d(inputdata, mean, std)
  end subroutine

  subroutine standardize_data_2d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:,:)
    real(kind=dp), intent(out) :: mean, std
    call one_pass_stats(sum(inputdata), sum(inputdata**2), size(inputdata), mean, std)
    call standardize_data_given_pars2d(inputdata, mean, std)
  end subroutine

  subroutine standardize_data_3d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:,:,:)
    real(kind=dp), intent(out) :: mean, std
    call one_pass_stats(sum(inputdata), sum(inputdata**2), size(inputdata), mean, std)
    call standardize_data_given_pars3d(inputdata, mean, std)
  end subroutine

  subroutine standardize_data_4d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:,:,:,:)
    real(kind=dp), intent(out) :: mean, std
    call one_pass_stats(sum(inputdata), sum(inputdata**2), size(inputdata), mean, std)
    call standardize_data_given_pars4d(inputdata, mean, std)
  end subroutine

  subroutine two_pass_field(field, mean, std)
    real(kind=dp), intent(inout) :: field(:,:,:)
    real(kind=dp), intent(out) :: mean, std
    mean = sum(field) / size(field)
    std = sqrt(sum((field - mean)**2) / size(field))
    call standardize_data_given_pars3d(field, mean, std)
  end subroutine

  subroutine standardize_data_5d(reservoir, inputdata, mean, std)
    type(reservoir_type), intent(in) :: reservoir
    real(kind=dp), intent(inout) :: inputdata(:,:,:,:,:)
    real(kind=dp), intent(out) :: mean(:), std(:)
    real(kind=dp), allocatable :: field(:,:,:)
    integer :: v, z, l
    l = 0
    do v = 1, size(inputdata, 1)
      do z = 1, size(inputdata, 4)
        l = l + 1
        field = inputdata(v,:,:,z,:)
        call two_pass_field(field, mean(l), std(l))
        inputdata(v,:,:,z,:) = field
      end do
    end do
  end subroutine

  subroutine standardize_data_5d_logp(reservoir, inputdata, logp, mean, std)
    type(reservoir_type), intent(in) :: reservoir
    real(kind=dp), intent(inout) :: inputdata(:,:,:,:,:), logp(:,:,:)
    real(kind=dp), intent(out) :: mean(:), std(:)
    integer :: l
    call standardize_data_5d(reservoir, inputdata, mean, std)
    l = size(inputdata, 1) * size(inputdata, 4) + 1
    call two_pass_field(logp, mean(l), std(l))
  end subroutine

  subroutine standardize_data_5d_logp_tisr(reservoir, inputdata, logp, tisr, mean, std)
    type(reservoir_type), intent(in) :: reservoir
    real(kind=dp), intent(inout) :: inputdata(:,:,:,:,:), logp(:,:,:), tisr(:,:,:)
    real(kind=dp), intent(out) :: mean(:), std(:)
    integer :: l
    call standardize_data_5d_logp(reservoir, inputdata, logp, mean, std)
    l = size(inputdata, 1) * size(inputdata, 4) + 2
    call two_pass_field(tisr, mean(l), std(l))
  end subroutine

  ! standardize_data_given_pars{1,2,4}d (src/mod_utilities.f90:1283-1329): subtract, then divide (two roundings)
  subroutine standardize_data_given_pars1d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:)
    real(kind=dp), intent(in) :: mean, std
    inputdata = inputdata - mean
    inputdata = inputdata / std
  end subroutine

  subroutine standardize_data_given_pars2d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:,:)
    real(kind=dp), intent(in) :: mean, std
    inputdata = inputdata - mean
    inputdata = inputdata / std
  end subroutine

  subroutine standardize_data_given_pars4d(inputdata, mean, std)
    real(kind=dp), intent(inout) :: inputdata(:,:,:,:)
    real(kind=dp), intent(in) :: mean, std
    inputdata = inputdata - mean
    inputdata = inputdata / std
  end subroutine

  ! gaussian_noise_{1d,2d} (src/mod_utilities.f90:1345-1385): x <- x + g noisemag x, g ~ N(0,1) drawn element by element with the
  ! Box-Muller pair of gaussian_noise_maker (:1519-1533; the 2-d form walks the FIRST index outermost, :1481-1495)
  function box_muller() result(g)
    real(kind=dp) :: g, u1, u2
    call random_number(u1)
    call random_number(u2)
    g = sqrt(-2.0_dp * log(u1)) * cos(8.0_dp * atan(1.0_dp) * u2)
  end function

  subroutine gaussian_noise_1d(inputdata, noisemag)
    real(kind=dp), intent(inout) :: inputdata(:)
    real(kind=dp), intent(in) :: noisemag
    real(kind=dp), allocatable :: g(:)
    integer :: i
    allocate(g(size(inputdata)))
    do i = 1, size(g)
      g(i) = box_muller()
    end do
    inputdata = inputdata + g * noisemag * inputdata
  end subroutine

  subroutine gaussian_noise_2d(inputdata, noisemag)
    real(kind=dp), intent(inout) :: inputdata(:,:)
    real(kind=dp), intent(in) :: noisemag
    real(kind=dp), allocatable :: g(:,:)
    integer :: i, j
    allocate(g(size(inputdata, 1), size(inputdata, 2)))
    do i = 1, size(g, 1)
      do j = 1, size(g, 2)
        g(i, j) = box_muller()
      end do
    end do
    inputdata = inputdata + g * noisemag * inputdata
  end subroutine

  ! rolling_average_over_a_period_2d (src/mod_utilities.f90:1773-1813): running mean along the second index over the last
  ! `period` + 1 entries divided by `period` (the first `period` entries: mean of what there is); a window whose sum is within 1e-7
  ! of zero leaves the entry as it was
  subroutine rolling_average_over_a_period_2d(grid, period)
    real(kind=dp), intent(inout) :: grid(:,:)
    integer, intent(in) :: period
    real(kind=dp), allocatable :: copy(:,:)
    real(kind=dp) :: window
    integer :: i, t
    copy = grid
    do i = 1, size(grid, 1)
      do t = 1, size(grid, 2)
        if (t - period < 1) then
          grid(i, t) = sum(copy(i, 1:t)) / t
        else
          window = sum(copy(i, t-period:t))
          if (abs(window) > 0.0000001_dp) grid(i, t) = window / period
        end if
      end do
    end do
  end subroutine

  ! standardize_data_given_pars5d / _5d_logp / _5d_logp_tisr / 3d (src/mod_utilities.f90:1195-1329): (v - mean_l) / std_l with
  ! l = (var-1)*heightlevels + level for the 3-d variables, then the 2-d fields with the statistics that follow them
  subroutine standardize_data_given_pars5d(mean, std, input_data)
    real(kind=dp), intent(in) :: mean(:), std(:)
    real(kind=dp), intent(inout) :: input_data(:,:,:,:,:)
    integer :: v, z, l
    l = 0
    do v = 1, size(input_data, 1)
      do z = 1, size(input_data, 4)
        l = l + 1
        input_data(v,:,:,z,:) = (input_data(v,:,:,z,:) - mean(l)) / std(l)
      end do
    end do
  end subroutine

  subroutine standardize_data_given_pars_5d_logp(mean, std, input_data, input_data2d)
    real(kind=dp), intent(in) :: mean(:), std(:)
    real(kind=dp), intent(inout) :: input_data(:,:,:,:,:), input_data2d(:,:,:)
    integer :: l
    call standardize_data_given_pars5d(mean, std, input_data)
    l = size(input_data, 1) * size(input_data, 4) + 1
    input_data2d = (input_data2d - mean(l)) / std(l)
  end subroutine

  subroutine standardize_data_given_pars_5d_logp_tisr(mean, std, input_data, input_data2d, input_data2d_2)
    real(kind=dp), intent(in) :: mean(:), std(:)
    real(kind=dp), intent(inout) :: input_data(:,:,:,:,:), input_data2d(:,:,:), input_data2d_2(:,:,:)
    integer :: l
    call standardize_data_given_pars_5d_logp(mean, std, input_data, input_data2d)
    l = size(input_data, 1) * size(input_data, 4) + 2
    input_data2d_2 = (input_data2d_2 - mean(l)) / std(l)
  end subroutine

  subroutine standardize_data_given_pars3d(input_data, mean, std)
    real(kind=dp), intent(inout) :: input_data(:,:,:)
    real(kind=dp), intent(in) :: mean, std
    input_data = (input_data - mean) / std
  end subroutine

  ! total_precip_over_a_period (src/mod_utilities.f90): hourly precipitation -> running totals over `period` hours
  subroutine total_precip_over_a_period(precip, period)
    real(kind=dp), intent(inout) :: precip(:,:,:)
    integer, intent(in) :: period
    real(kind=dp), allocatable :: copy(:,:,:)
    integer :: t, t0
    copy = precip
    do t = 1, size(precip, 3)
      t0 = merge(1, t - period, t - period < 1)             ! (the reference's window holds period + 1 hours, :1720-1724)
      precip(:,:,t) = sum(copy(:,:,t0:t), dim=3)
    end do
  end subroutine

end module mod_utilities
