! Module speedy_res_interface of the MI355X drop-in: the names the reference tree imports from it --
!     startspeedy(model_parameters,grid,runspeedy)                                         src/speedy_res_interface.f90:20-39     (program main, :9)
!     write_restart_new(filename,timestep,grid4d,grid2d)                                   :51-61                                 (ppo_iogrid, :26)
!     getspeedyvariable()                                                                  :63-90                                 (dyn_stloop, :14)
!     read_era_netcdf_opened(reservoir,grid,model_parameters,start_year,end_year,era_data,netcdf_files,timestep_arg)    :248-437
!     read_era(reservoir,grid,model_parameters,start_year,end_year,era_data,timestep_arg)  :439-635   (mod_reservoir, mod_slab_ocean_reservoir)
!     read_model_states(reservoir,grid,model_parameters,start_year,end_year,speedy_data,timestep_arg)                    :637-723
!     truncate_letkf_code_version(field_orig, trunc_twn)                                   :820-839                               (ppo_iogrid, :26)
! -- and the module variable internal_state_vector.
!
! What is compute here is done here; what is file I/O stays with the host.  The three readers are NetCDF code in the reference
! (mod_io, out of scope: SURVEY 2): they forward, with the reference's argument lists, to module speedyml_data_source, which a host
! supplies -- in the reference tree a thin file around its own mod_io readers, in this repository's tests the synthetic generator
! of fortran/test_support.f90.  startspeedy does what the reference's does (the rank's domain and the calendar; SPEEDY itself is
! initialised inside the device-resident engine when the first forecast starts, mpires::start_forecast).  getspeedyvariable and
! write_restart_new are the reference's own no-ops (their bodies are commented out there but for a progress print; SPEEDY's stloop
! does not run on the host in the drop-in, so there is no step counter to print).
module speedy_res_interface
  use mod_utilities, only : dp, speedy_data_type, era_data_type, state_vector_type, reservoir_type, grid_type, model_parameters_type, &
                            opened_netcdf_type
  use mod_calendar, only : calendar, initialize_calendar
  implicit none
  type(state_vector_type) :: internal_state_vector

contains

  subroutine startspeedy(model_parameters, grid, runspeedy)
    use mpires, only : mpi_res
    use resdomain, only : initializedomain
    type(model_parameters_type), intent(in) :: model_parameters
    type(grid_type), intent(inout) :: grid
    logical, intent(in) :: runspeedy
    ! (the reference passes a vertical level it never sets, :34-36; the one-level layout it ships with makes that level 1)
    call initializedomain(mpi_res%numprocs, mpi_res%proc_num, model_parameters%overlap, grid%num_vert_levels, 1, grid%vert_overlap, grid)
    call initialize_calendar(calendar, 1981, 1, 1, 0)
  end subroutine

  subroutine write_restart_new(filename, timestep, grid4d, grid2d)
    character(len=*), intent(in) :: filename
    integer, intent(in) :: timestep
    real(kind=dp), intent(in) :: grid4d(:,:,:,:), grid2d(:,:)
  end subroutine

  subroutine getspeedyvariable()
  end subroutine

  subroutine read_era(reservoir, grid, model_parameters, start_year, end_year, era_data, timestep_arg)
    use speedyml_data_source, only : source_read_era
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer, intent(in) :: start_year, end_year
    type(era_data_type), intent(inout) :: era_data
    integer, intent(in), optional :: timestep_arg
    call source_read_era(reservoir, grid, model_parameters, start_year, end_year, era_data, timestep_arg)
  end subroutine

  ! the same window through files the caller keeps open between calls: the open handles are the host reader's business
  subroutine read_era_netcdf_opened(reservoir, grid, model_parameters, start_year, end_year, era_data, netcdf_files, timestep_arg)
    use speedyml_data_source, only : source_read_era
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(opened_netcdf_type), intent(inout) :: netcdf_files(:)
    type(model_parameters_type), intent(in) :: model_parameters
    integer, intent(in) :: start_year, end_year
    type(era_data_type), intent(inout) :: era_data
    integer, intent(in), optional :: timestep_arg
    call source_read_era(reservoir, grid, model_parameters, start_year, end_year, era_data, timestep_arg)
  end subroutine

  subroutine read_model_states(reservoir, grid, model_parameters, start_year, end_year, speedy_data, timestep_arg)
    use speedyml_data_source, only : source_read_model_states
    type(reservoir_type), intent(inout) :: reservoir
    type(grid_type), intent(inout) :: grid
    type(model_parameters_type), intent(in) :: model_parameters
    integer, intent(in) :: start_year, end_year
    type(speedy_data_type), intent(inout) :: speedy_data
    integer, intent(in), optional :: timestep_arg
    call source_read_model_states(reservoir, grid, model_parameters, start_year, end_year, speedy_data, timestep_arg)
  end subroutine

  ! triangular truncation of one spectral field: coefficients with total wavenumber m + n - 2 above trunc_twn are set to zero
  function truncate_letkf_code_version(field_orig, trunc_twn) result(field_new)
    complex, intent(in) :: field_orig(:,:)
    integer, intent(in) :: trunc_twn
    complex, allocatable :: field_new(:,:)
    integer :: m, n
    allocate(field_new(size(field_orig, 1), size(field_orig, 2)))
    do n = 1, size(field_orig, 2)
      do m = 1, size(field_orig, 1)
        field_new(m, n) = merge(field_orig(m, n), (0.0, 0.0), m + n - 2 <= trunc_twn)
      end do
    end do
  end function

end module speedy_res_interface
