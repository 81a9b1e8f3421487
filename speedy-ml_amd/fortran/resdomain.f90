! Module resdomain of the drop-in (src/res_domain.f90): the domain bookkeeping program main and mod_reservoir call, forwarded to the
! library's integer routines (sml_domain_*: bit-exact against the oracle's restatement of res_domain.f90, tests/test_domain_product.py).
module resdomain
  use iso_c_binding
  use speedyml_hip
  use mod_utilities, only : dp, grid_type, reservoir_type, model_parameters_type, speedy_data_type
  implicit none
  ! Gaussian latitudes of the T30 grid (src/mod_utilities.f90:18-29)
  real(kind=dp), parameter :: speedylat(48) = [ -87.159_dp, -83.479_dp, -79.777_dp, -76.070_dp, -72.362_dp, -68.652_dp, -64.942_dp, &
      -61.232_dp, -57.521_dp, -53.810_dp, -50.099_dp, -46.389_dp, -42.678_dp, -38.967_dp, -35.256_dp, -31.545_dp, -27.833_dp, -24.122_dp, &
      -20.411_dp, -16.700_dp, -12.989_dp, -9.278_dp, -5.567_dp, -1.856_dp, 1.856_dp, 5.567_dp, 9.278_dp, 12.989_dp, 16.700_dp, 20.411_dp, &
      24.122_dp, 27.833_dp, 31.545_dp, 35.256_dp, 38.967_dp, 42.678_dp, 46.389_dp, 50.099_dp, 53.810_dp, 57.521_dp, 61.232_dp, 64.942_dp, &
      68.652_dp, 72.362_dp, 76.070_dp, 79.777_dp, 83.479_dp, 87.159_dp ]
contains

  ! processor_decomposition (src/res_domain.f90:31-62): the regions of this rank, remainder rule included
  subroutine processor_decomposition(model_parameters)
    type(model_parameters_type), intent(inout) :: model_parameters
    integer(c_int), allocatable :: idx(:)
    integer(c_int) :: cnt
    model_parameters%number_of_regions = 1152
    allocate(idx(model_parameters%number_of_regions))
    cnt = sml_domain_decompose(int(model_parameters%irank, c_int), int(model_parameters%numprocs, c_int), &
                               int(model_parameters%number_of_regions, c_int), idx, int(size(idx), c_int))
    call sml_check(cnt, 'sml_domain_decompose')
    model_parameters%num_of_regions_on_proc = cnt
    if (allocated(model_parameters%region_indices)) deallocate(model_parameters%region_indices)
    allocate(model_parameters%region_indices(cnt))
    model_parameters%region_indices = idx(1:cnt)
  end subroutine

  ! initializedomain (src/res_domain.f90:96-121)
  subroutine initializedomain(num_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, grid)
    integer, intent(in) :: num_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap
    type(grid_type), intent(inout) :: grid
    type(sml_region) :: g
    call sml_check(sml_domain_region(int(num_regions, c_int), int(region_num, c_int), int(overlap, c_int), int(num_vert_levels, c_int), &
                                     int(vert_level, c_int), int(vert_overlap, c_int), g), 'sml_domain_region')
    grid%res_xstart = g%res_xstart; grid%res_xend = g%res_xend; grid%res_ystart = g%res_ystart; grid%res_yend = g%res_yend
    grid%resxchunk = g%resxchunk; grid%resychunk = g%resychunk
    grid%res_zstart = g%res_zstart; grid%res_zend = g%res_zend; grid%reszchunk = g%reszchunk
    grid%input_xstart = g%input_xstart; grid%input_xend = g%input_xend; grid%input_ystart = g%input_ystart; grid%input_yend = g%input_yend
    grid%inputxchunk = g%inputxchunk; grid%inputychunk = g%inputychunk
    grid%input_zstart = g%input_zstart; grid%input_zend = g%input_zend; grid%inputzchunk = g%inputzchunk
    grid%pole = g%pole /= 0; grid%periodicboundary = g%periodicboundary /= 0; grid%top = g%top /= 0; grid%bottom = g%bottom /= 0
    grid%tdata_xstart = g%tdata_xstart; grid%tdata_xend = g%tdata_xend; grid%tdata_ystart = g%tdata_ystart; grid%tdata_yend = g%tdata_yend
    grid%tdata_zstart = g%tdata_zstart; grid%tdata_zend = g%tdata_zend
    grid%overlap = overlap; grid%num_vert_levels = num_vert_levels; grid%vert_overlap = vert_overlap
    grid%number_of_regions = num_regions
  end subroutine

  ! set_region + set_reservoir_by_region (src/res_domain.f90:1564-1661): latitude class, noise magnitude, spectral radius
  subroutine set_reservoir_by_region(reservoir, grid)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    real(kind=dp) :: lat0, lat1
    lat0 = speedylat(grid%res_ystart); lat1 = speedylat(grid%res_yend)
    if (lat0 <= -60.0_dp .or. lat1 >= 60.0_dp) then
      grid%region_char = 'polar'
    else if ((lat0 <= -30.0_dp .and. lat0 > -60.0_dp) .or. (lat0 >= 30.0_dp .and. lat0 < 60.0_dp)) then
      grid%region_char = 'extratropic'
    else
      grid%region_char = 'tropic'
    end if
    reservoir%noisemag = 0.20_dp
    if (abs(min(lat0, lat1)) >= 45.0_dp) then
      reservoir%radius = 0.7_dp
    else
      reservoir%radius = (0.7_dp - 0.3_dp) / 45.0_dp + 0.3_dp
    end if
  end subroutine

  ! standardize_speedy_data (src/res_domain.f90): SPEEDY's forecast of the res patch with the same statistics as the ERA fields
  subroutine standardize_speedy_data(reservoir, grid, speedy_data)
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(speedy_data_type), intent(inout) :: speedy_data
    integer :: v, z, l
    l = 0
    do v = 1, size(speedy_data%speedyvariables, 1)
      do z = 1, size(speedy_data%speedyvariables, 4)
        l = l + 1
        speedy_data%speedyvariables(v,:,:,z,:) = (speedy_data%speedyvariables(v,:,:,z,:) - grid%mean(l)) / grid%std(l)
      end do
    end do
    if (reservoir%logp_bool) speedy_data%speedy_logp = (speedy_data%speedy_logp - grid%mean(grid%logp_mean_std_idx)) / grid%std(grid%logp_mean_std_idx)
  end subroutine

end module resdomain
