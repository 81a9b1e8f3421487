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
  public :: train_job, train_job_passes, train_enqueue, train_flush, train_take, train_pending, train_group_size, train_last_seconds, train_last_count

  type train_job
    integer :: n = 0, d = 0, k = 0, n_model = 0, n_out = 0, discard = 0, batch = 0, ml_variant = 0, using_prior = 0
    real(c_double) :: leakage = 1.0_c_double, beta_res = 0.0_c_double, beta_model = 0.0_c_double, prior_val = 0.0_c_double
    integer(c_int), allocatable :: rows(:), cols(:)
    real(c_double), allocatable :: vals(:), win(:,:), mean(:), std(:)
    ! the interleaved passes (columns i, i + step, i + 2 step ... of the hourly arrays): pass p has ncol(p) columns
    integer, allocatable :: ncol(:)
    real(c_double), allocatable :: noisy(:,:,:), targ(:,:,:), mdl(:,:,:)      ! (d, max ncol, npass), (n_out, ., .), (n_model, ., .)
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
    ! a Gram matrix of non-finite columns fails much later, inside the ridge solve: refuse them here, where the region is still known
    if (.not. (all(abs(job%noisy) <= huge(1.0_c_double)) .and. all(abs(job%targ) <= huge(1.0_c_double)) .and. all(abs(job%mdl) <= huge(1.0_c_double)))) then
      write(*,'(a,i0,a,3l2)') ' speedyml_train: job ', njobs + 1, ': training columns are not finite (inputs, targets, model states finite =', &
            all(abs(job%noisy) <= huge(1.0_c_double)), all(abs(job%targ) <= huge(1.0_c_double)), all(abs(job%mdl) <= huge(1.0_c_double))
      stop 1
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
    if (allocated(a%ncol)) call move_alloc(a%ncol, b%ncol)
    if (allocated(a%noisy)) call move_alloc(a%noisy, b%noisy)
    if (allocated(a%targ)) call move_alloc(a%targ, b%targ)
    if (allocated(a%mdl)) call move_alloc(a%mdl, b%mdl)
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

  ! storage of a job's passes: npass passes of at most max_col columns
  subroutine train_job_passes(job, npass, max_col)
    type(train_job), intent(inout) :: job
    integer, intent(in) :: npass, max_col
    allocate(job%ncol(npass), job%noisy(job%d, max_col, npass), job%targ(job%n_out, max_col, npass), job%mdl(max(job%n_model, 1), max_col, npass))
    job%ncol = 0; job%noisy = 0.0_c_double; job%targ = 0.0_c_double; job%mdl = 0.0_c_double
  end subroutine

  logical function same_schedule(a, b)
    type(train_job), intent(in) :: a, b
    same_schedule = a%discard == b%discard .and. a%batch == b%batch .and. a%ml_variant == b%ml_variant .and. size(a%ncol) == size(b%ncol)
    if (same_schedule) same_schedule = all(a%ncol == b%ncol)
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
    real(c_double), allocatable :: stage(:,:,:), zero_wout(:,:), buf(:,:)
    integer(c_int), allocatable :: nostat(:)
    logical, allocatable :: solved(:)
    integer :: cap, s, p, i, j, k, max_d, max_nm, max_no, max_col, ncol, n_aug, cnt, npass
    integer(c_int) :: nb
    integer(c_int64_t) :: b8
    cap = size(m)
    k = m(1)
    npass = size(jobs(k)%ncol)
    max_col = maxval(jobs(k)%ncol)
    max_d = 0; max_nm = 1; max_no = 0
    do s = 1, cap
      max_d = max(max_d, jobs(m(s))%d); max_nm = max(max_nm, jobs(m(s))%n_model); max_no = max(max_no, jobs(m(s))%n_out)
    end do
    ! a bank of `cap` slots for the recurrences (W_out plays no part in training: zeros)
    call sml_check(sml_bank_create(int(cap, c_int), int(max_d, c_int), int(max_nm, c_int), int(max_no, c_int), tbank), 'sml_bank_create')
    allocate(dmodel(cap), dtarg(cap), dc(cap), db(cap), dw(cap))
    b8 = 8
    do s = 1, cap
      k = m(s)
      n_aug = jobs(k)%n + jobs(k)%n_model
      allocate(zero_wout(jobs(k)%n_out, n_aug), nostat(jobs(k)%n_out))
      zero_wout = 0.0_c_double; nostat = -1
      call sml_check(sml_bank_load(tbank, int(s - 1, c_int), int(jobs(k)%n, c_int), int(jobs(k)%d, c_int), int(jobs(k)%k, c_int), int(jobs(k)%n_model, c_int), &
                                   int(jobs(k)%n_out, c_int), jobs(k)%rows, jobs(k)%cols, jobs(k)%vals, jobs(k)%win, zero_wout, jobs(k)%leakage, jobs(k)%mean, &
                                   jobs(k)%std, int(size(jobs(k)%mean), c_int), nostat), 'sml_bank_load')
      deallocate(zero_wout, nostat)
      call sml_check(sml_dev_alloc(b8 * n_aug * n_aug, dc(s)), 'sml_dev_alloc'); call sml_check(sml_dev_zero(dc(s), b8 * n_aug * n_aug), 'sml_dev_zero')
      call sml_check(sml_dev_alloc(b8 * jobs(k)%n_out * n_aug, db(s)), 'sml_dev_alloc'); call sml_check(sml_dev_zero(db(s), b8 * jobs(k)%n_out * n_aug), 'sml_dev_zero')
      call sml_check(sml_dev_alloc(b8 * jobs(k)%n_out * n_aug, dw(s)), 'sml_dev_alloc')
      call sml_check(sml_dev_alloc(b8 * jobs(k)%n_out * max_col, dtarg(s)), 'sml_dev_alloc')
      call sml_check(sml_dev_alloc(b8 * max(jobs(k)%n_model, 1) * max_col, dmodel(s)), 'sml_dev_alloc')
    end do
    call sml_check(sml_dev_alloc(b8 * max_d * cap * max_col, dnoisy), 'sml_dev_alloc')
    do p = 1, npass
      ncol = jobs(m(1))%ncol(p)
      allocate(stage(max_d, cap, ncol))              ! = [ncol][capacity][max_d] as the library reads it
      stage = 0.0_c_double
      do s = 1, cap
        k = m(s)
        stage(1:jobs(k)%d, s, 1:ncol) = jobs(k)%noisy(:, 1:ncol, p)
        buf = jobs(k)%targ(:, 1:ncol, p)               ! (contiguous copies: the library reads (rows, ncol) column-major blocks)
        call sml_check(sml_dev_upload(dtarg(s), buf, b8 * jobs(k)%n_out * ncol), 'sml_dev_upload')
        if (jobs(k)%n_model > 0) then
          buf = jobs(k)%mdl(:, 1:ncol, p)
          call sml_check(sml_dev_upload(dmodel(s), buf, b8 * jobs(k)%n_model * ncol), 'sml_dev_upload')
        end if
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
      k = m(i)
      write(*,'(a,i0,a,i0,a,i0,a,i0,a,i0,a)') ' speedyml_train: ridge solves of ', cnt, ' system(s) with n = ', jobs(k)%n, ', n_model = ', jobs(k)%n_model, &
            ', n_out = ', jobs(k)%n_out, ' (', npass, ' passes accumulated)'
      call sml_check(sml_train_fit_batched(int(cnt, c_int), cc, bb, int(jobs(k)%n, c_int), int(jobs(k)%n_model, c_int), int(jobs(k)%n_out, c_int), &
                                           jobs(k)%beta_res, jobs(k)%beta_model, jobs(k)%prior_val, int(jobs(k)%using_prior, c_int), ww, c_null_ptr), &
                     'sml_train_fit_batched')
    end do
    do s = 1, cap
      k = m(s)
      n_aug = jobs(k)%n + jobs(k)%n_model
      allocate(jobs(k)%wout(jobs(k)%n_out, n_aug))
      call sml_check(sml_dev_download(jobs(k)%wout, dw(s), b8 * jobs(k)%n_out * n_aug), 'sml_dev_download')
      if (c_associated(jobs(k)%bank)) call sml_check(sml_bank_set_wout(jobs(k)%bank, jobs(k)%slot, jobs(k)%wout), 'sml_bank_set_wout')
      jobs(k)%done = .true.
      deallocate(jobs(k)%ncol, jobs(k)%noisy, jobs(k)%targ, jobs(k)%mdl, jobs(k)%rows, jobs(k)%cols, jobs(k)%vals, jobs(k)%win)
      call sml_check(sml_dev_free(dc(s)), 'sml_dev_free'); call sml_check(sml_dev_free(db(s)), 'sml_dev_free'); call sml_check(sml_dev_free(dw(s)), 'sml_dev_free')
      call sml_check(sml_dev_free(dtarg(s)), 'sml_dev_free'); call sml_check(sml_dev_free(dmodel(s)), 'sml_dev_free')
    end do
    call sml_check(sml_dev_free(dnoisy), 'sml_dev_free')
    call sml_check(sml_bank_destroy(tbank), 'sml_bank_destroy')
  end subroutine

end module speedyml_train
