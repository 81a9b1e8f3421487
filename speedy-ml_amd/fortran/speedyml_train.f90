! The training queue of the drop-in modules: train_reservoir (src/mod_reservoir.f90:214-320) and train_slab_ocean_model
! (src/mod_slab_ocean_reservoir.f90:172-269) are called once per region by program main (src/parallelmain.f90:82-128), but the device
! trains whole groups at once -- one recurrence launch per time column shared by every resident reservoir (sml_bank_train_pass) and
! the ridge solves of a size class in lockstep (sml_train_fit_batched).  So a call only ENQUEUES its reservoir, with everything the
! training needs (the interleaved passes of noisy inputs, targets and imperfect-model columns, drawn and laid out at enqueue time, in
! the caller's order: the random draws are those of the one-at-a-time path); the queue runs when it holds `group` jobs
! (SML_TRAIN_RESIDENTS, default 64), when the rank's last reservoir arrives, or when somebody needs a result (train_take).  A group of
! one IS the one-at-a-time path, and larger groups give the same W_out bit for bit (fortran/test_train_batch.f90).
module speedyml_train
  use iso_c_binding
  use speedyml_hip
  implicit none
  private
  public :: train_job, pass_data, train_enqueue, train_flush, train_take, train_pending, train_group_size, train_last_seconds, train_last_count

  type pass_data                                 ! one interleaved pass: columns i, i + step, i + 2 step ... of the hourly arrays
    real(c_double), allocatable :: noisy(:,:), targ(:,:), mdl(:,:)      ! (d, ncol), (n_out, ncol), (n_model, ncol)
  end type

  type train_job
    integer :: n = 0, d = 0, k = 0, n_model = 0, n_out = 0, discard = 0, batch = 0, ml_variant = 0, using_prior = 0
    real(c_double) :: leakage = 1.0_c_double, beta_res = 0.0_c_double, beta_model = 0.0_c_double, prior_val = 0.0_c_double
    integer(c_int), allocatable :: rows(:), cols(:)
    real(c_double), allocatable :: vals(:), win(:,:), mean(:), std(:)
    type(pass_data), allocatable :: pass(:)
    type(c_ptr) :: bank = c_null_ptr             ! the prediction bank that receives W_out (sml_bank_set_wout), or null
    integer(c_int) :: slot = -1
    real(c_double), allocatable :: wout(:,:)     ! (n_out, n_model + n): the result
    logical :: done = .false.
  end type

  type(train_job), allocatable, save :: jobs(:)
  integer, save :: njobs = 0, first_pending = 1
  real(c_double), save :: train_last_seconds = 0.0_c_double
  integer, save :: train_last_count = 0

contains

  integer function train_group_size()
    character(len=32) :: env
    integer :: n, stat
    train_group_size = 64
    call get_environment_variable('SML_TRAIN_RESIDENTS', env, n, stat)
    if (stat == 0 .and. n > 0) read(env(1:n), *) train_group_size
    train_group_size = max(train_group_size, 1)
  end function

  integer function train_pending()
    train_pending = njobs - first_pending + 1
  end function

  ! takes over `job` (its allocatable components are moved); returns its id
  integer function train_enqueue(job)
    type(train_job), intent(inout) :: job
    type(train_job), allocatable :: grown(:)
    integer :: i
    if (.not. allocated(jobs)) allocate(jobs(16))
    if (njobs == size(jobs)) then
      allocate(grown(2 * size(jobs)))
      do i = 1, njobs
        call move_job(jobs(i), grown(i))
      end do
      call move_alloc(grown, jobs)
    end if
    njobs = njobs + 1
    call move_job(job, jobs(njobs))
    train_enqueue = njobs
    if (train_pending() >= train_group_size()) call train_flush()
  end function

  subroutine move_job(a, b)
    type(train_job), intent(inout) :: a, b
    b%n = a%n; b%d = a%d; b%k = a%k; b%n_model = a%n_model; b%n_out = a%n_out; b%discard = a%discard; b%batch = a%batch
    b%ml_variant = a%ml_variant; b%using_prior = a%using_prior; b%leakage = a%leakage; b%beta_res = a%beta_res
    b%beta_model = a%beta_model; b%prior_val = a%prior_val; b%bank = a%bank; b%slot = a%slot; b%done = a%done
    if (allocated(a%rows)) call move_alloc(a%rows, b%rows)
    if (allocated(a%cols)) call move_alloc(a%cols, b%cols)
    if (allocated(a%vals)) call move_alloc(a%vals, b%vals)
    if (allocated(a%win)) call move_alloc(a%win, b%win)
    if (allocated(a%mean)) call move_alloc(a%mean, b%mean)
    if (allocated(a%std)) call move_alloc(a%std, b%std)
    if (allocated(a%pass)) call move_alloc(a%pass, b%pass)
    if (allocated(a%wout)) call move_alloc(a%wout, b%wout)
  end subroutine

  ! W_out of job `id` (running the queue first if the job is still pending); the job's buffers are released
  subroutine train_take(id, wout)
    integer, intent(in) :: id
    real(c_double), intent(out) :: wout(:,:)
    if (id < 1 .or. id > njobs) stop 'speedyml_train: unknown job'
    if (.not. jobs(id)%done) call train_flush()
    wout = jobs(id)%wout
    deallocate(jobs(id)%wout)
  end subroutine

  logical function same_schedule(a, b)
    type(train_job), intent(in) :: a, b
    integer :: p
    same_schedule = a%discard == b%discard .and. a%batch == b%batch .and. a%ml_variant == b%ml_variant .and. size(a%pass) == size(b%pass)
    if (.not. same_schedule) return
    do p = 1, size(a%pass)
      if (size(a%pass(p)%noisy, 2) /= size(b%pass(p)%noisy, 2)) same_schedule = .false.
    end do
  end function

  logical function same_system(a, b)
    type(train_job), intent(in) :: a, b
    same_system = a%n == b%n .and. a%n_model == b%n_model .and. a%n_out == b%n_out .and. a%beta_res == b%beta_res .and. &
                  a%beta_model == b%beta_model .and. a%prior_val == b%prior_val .and. a%using_prior == b%using_prior
  end function

  ! train every pending job: groups that share a pass schedule go through one bank, their ridge solves by size class
  subroutine train_flush()
    integer :: i, j, cnt
    integer, allocatable :: members(:)
    logical, allocatable :: taken(:)
    integer(c_int64_t) :: c0, c1, rate
    if (train_pending() <= 0) return
    call system_clock(c0, rate)
    allocate(taken(njobs), members(njobs))
    taken = .false.
    do i = first_pending, njobs
      if (taken(i)) cycle
      cnt = 0
      do j = i, njobs
        if (.not. taken(j)) then
          if (same_schedule(jobs(i), jobs(j))) then
            cnt = cnt + 1; members(cnt) = j; taken(j) = .true.
          end if
        end if
      end do
      call train_group(members(1:cnt))
    end do
    call system_clock(c1)
    train_last_count = train_pending()
    train_last_seconds = real(c1 - c0, c_double) / real(rate, c_double)
    write(*,'(a,i0,a,f9.3,a,f9.2,a)') ' speedyml_train: trained ', train_last_count, ' reservoir(s) in ', train_last_seconds, ' s (', &
          1.0d3 * train_last_seconds / train_last_count, ' ms each: recurrence + Gram accumulation + ridge solve)'
    first_pending = njobs + 1
  end subroutine

  subroutine train_group(m)
    integer, intent(in) :: m(:)
    type(c_ptr) :: tbank, dnoisy
    type(c_ptr), allocatable :: dmodel(:), dtarg(:), dc(:), db(:), dw(:), cc(:), bb(:), ww(:)
    real(c_double), allocatable :: stage(:,:,:), zero_wout(:,:)
    integer(c_int), allocatable :: nostat(:)
    logical, allocatable :: solved(:)
    integer :: cap, s, p, i, j, max_d, max_nm, max_no, max_col, ncol, n_aug, cnt
    integer(c_int) :: nb
    integer(c_int64_t) :: b8
    cap = size(m)
    max_d = 0; max_nm = 1; max_no = 0; max_col = 0
    do s = 1, cap
      max_d = max(max_d, jobs(m(s))%d); max_nm = max(max_nm, jobs(m(s))%n_model); max_no = max(max_no, jobs(m(s))%n_out)
    end do
    do p = 1, size(jobs(m(1))%pass)
      max_col = max(max_col, size(jobs(m(1))%pass(p)%noisy, 2))
    end do
    ! a bank of `cap` slots for the recurrences (W_out plays no part in training: zeros)
    call sml_check(sml_bank_create(int(cap, c_int), int(max_d, c_int), int(max_nm, c_int), int(max_no, c_int), tbank), 'sml_bank_create')
    allocate(dmodel(cap), dtarg(cap), dc(cap), db(cap), dw(cap))
    b8 = 8
    do s = 1, cap
      associate (q => jobs(m(s)))
        n_aug = q%n + q%n_model
        allocate(zero_wout(q%n_out, n_aug), nostat(q%n_out))
        zero_wout = 0.0_c_double; nostat = -1
        call sml_check(sml_bank_load(tbank, int(s - 1, c_int), int(q%n, c_int), int(q%d, c_int), int(q%k, c_int), int(q%n_model, c_int), &
                                     int(q%n_out, c_int), q%rows, q%cols, q%vals, q%win, zero_wout, q%leakage, q%mean, q%std, &
                                     int(size(q%mean), c_int), nostat), 'sml_bank_load')
        deallocate(zero_wout, nostat)
        call sml_check(sml_dev_alloc(b8 * n_aug * n_aug, dc(s)), 'sml_dev_alloc'); call sml_check(sml_dev_zero(dc(s), b8 * n_aug * n_aug), 'sml_dev_zero')
        call sml_check(sml_dev_alloc(b8 * q%n_out * n_aug, db(s)), 'sml_dev_alloc'); call sml_check(sml_dev_zero(db(s), b8 * q%n_out * n_aug), 'sml_dev_zero')
        call sml_check(sml_dev_alloc(b8 * q%n_out * n_aug, dw(s)), 'sml_dev_alloc')
        call sml_check(sml_dev_alloc(b8 * q%n_out * max_col, dtarg(s)), 'sml_dev_alloc')
        call sml_check(sml_dev_alloc(b8 * max(q%n_model, 1) * max_col, dmodel(s)), 'sml_dev_alloc')
      end associate
    end do
    call sml_check(sml_dev_alloc(b8 * max_d * cap * max_col, dnoisy), 'sml_dev_alloc')
    do p = 1, size(jobs(m(1))%pass)
      ncol = size(jobs(m(1))%pass(p)%noisy, 2)
      allocate(stage(max_d, cap, ncol))              ! = [ncol][capacity][max_d] as the library reads it
      stage = 0.0_c_double
      do s = 1, cap
        associate (q => jobs(m(s)))
          stage(1:q%d, s, :) = q%pass(p)%noisy
          call sml_check(sml_dev_upload(dtarg(s), q%pass(p)%targ, b8 * q%n_out * ncol), 'sml_dev_upload')
          if (q%n_model > 0) call sml_check(sml_dev_upload(dmodel(s), q%pass(p)%mdl, b8 * q%n_model * ncol), 'sml_dev_upload')
        end associate
      end do
      call sml_check(sml_dev_upload(dnoisy, stage, b8 * max_d * cap * ncol), 'sml_dev_upload')
      deallocate(stage)
      nb = sml_bank_train_pass(tbank, dnoisy, int(ncol, c_int), int(jobs(m(1))%discard, c_int), int(jobs(m(1))%batch, c_int), dmodel, dtarg, dc, db, &
                               int(jobs(m(1))%ml_variant, c_int), c_null_ptr)
      call sml_check(nb, 'sml_bank_train_pass')
    end do
    ! fit_chunk_hybrid / fit_chunk_ml: the systems of one size class in lockstep
    allocate(solved(cap), cc(cap), bb(cap), ww(cap))
    solved = .false.
    do i = 1, cap
      if (solved(i)) cycle
      cnt = 0
      do j = i, cap
        if (.not. solved(j)) then
          if (same_system(jobs(m(i)), jobs(m(j)))) then
            cnt = cnt + 1; cc(cnt) = dc(j); bb(cnt) = db(j); ww(cnt) = dw(j); solved(j) = .true.
          end if
        end if
      end do
      associate (q => jobs(m(i)))
        call sml_check(sml_train_fit_batched(int(cnt, c_int), cc, bb, int(q%n, c_int), int(q%n_model, c_int), int(q%n_out, c_int), q%beta_res, &
                                             q%beta_model, q%prior_val, int(q%using_prior, c_int), ww, c_null_ptr), 'sml_train_fit_batched')
      end associate
    end do
    do s = 1, cap
      associate (q => jobs(m(s)))
        n_aug = q%n + q%n_model
        allocate(q%wout(q%n_out, n_aug))
        call sml_check(sml_dev_download(q%wout, dw(s), b8 * q%n_out * n_aug), 'sml_dev_download')
        if (c_associated(q%bank)) call sml_check(sml_bank_set_wout(q%bank, q%slot, q%wout), 'sml_bank_set_wout')
        q%done = .true.
        deallocate(q%pass, q%rows, q%cols, q%vals, q%win)
      end associate
      call sml_check(sml_dev_free(dc(s)), 'sml_dev_free'); call sml_check(sml_dev_free(db(s)), 'sml_dev_free'); call sml_check(sml_dev_free(dw(s)), 'sml_dev_free')
      call sml_check(sml_dev_free(dtarg(s)), 'sml_dev_free'); call sml_check(sml_dev_free(dmodel(s)), 'sml_dev_free')
    end do
    call sml_check(sml_dev_free(dnoisy), 'sml_dev_free')
    call sml_check(sml_bank_destroy(tbank), 'sml_bank_destroy')
  end subroutine

end module speedyml_train
