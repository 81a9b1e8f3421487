! iso_c_binding interfaces to libspeedyml_hip.so (include/speedyml_hip.h) -- the host language the reference is
! written in.  Only plain pointers, sizes and opaque handles cross the boundary.
module speedyml_hip
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: SML_OK = 0

  interface
    function sml_last_error() bind(C, name="sml_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
    function sml_device_count() bind(C, name="sml_device_count") result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function sml_set_device(ordinal) bind(C, name="sml_set_device") result(rc)
      import :: c_int
      integer(c_int), value :: ordinal
      integer(c_int) :: rc
    end function

    ! ---- resdomain ----
    function sml_domain_decompose(rank, nranks, number_of_regions, region_indices, capacity) bind(C, name="sml_domain_decompose") result(n)
      import :: c_int
      integer(c_int), value :: rank, nranks, number_of_regions, capacity
      integer(c_int), intent(out) :: region_indices(*)
      integer(c_int) :: n
    end function
    function sml_domain_out_map(number_of_regions, region_num, num_vert_levels, vert_level, vert_overlap, precip_bool, &
                                g_index, stat_idx, capacity) bind(C, name="sml_domain_out_map") result(n)
      import :: c_int
      integer(c_int), value :: number_of_regions, region_num, num_vert_levels, vert_level, vert_overlap, precip_bool, capacity
      integer(c_int), intent(out) :: g_index(*), stat_idx(*)
      integer(c_int) :: n
    end function

    ! ---- reservoir bank ----
    function sml_bank_create(capacity, max_d, max_n_model, max_n_out, bank) bind(C, name="sml_bank_create") result(rc)
      import :: c_int, c_ptr
      integer(c_int), value :: capacity, max_d, max_n_model, max_n_out
      type(c_ptr), intent(out) :: bank
      integer(c_int) :: rc
    end function
    function sml_bank_destroy(bank) bind(C, name="sml_bank_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank
      integer(c_int) :: rc
    end function
    function sml_bank_load(bank, slot, n, d, k, n_model, n_out, rows, cols, vals, win, wout, leakage, mean, std, nstat, &
                           out_stat_idx) bind(C, name="sml_bank_load") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot, n, d, k, n_model, n_out, nstat
      integer(c_int), intent(in) :: rows(*), cols(*), out_stat_idx(*)
      real(c_double), intent(in) :: vals(*), win(*), wout(*), mean(*), std(*)
      real(c_double), value :: leakage
      integer(c_int) :: rc
    end function
    function sml_bank_set_feedback(bank, slot, u) bind(C, name="sml_bank_set_feedback") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(in) :: u(*)
      integer(c_int) :: rc
    end function
    function sml_bank_set_state(bank, slot, x) bind(C, name="sml_bank_set_state") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(in) :: x(*)
      integer(c_int) :: rc
    end function
    function sml_bank_get_state(bank, slot, x) bind(C, name="sml_bank_get_state") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(out) :: x(*)
      integer(c_int) :: rc
    end function
    function sml_bank_predict_one(bank, slot, x, local_model, outvec) bind(C, name="sml_bank_predict_one") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(inout) :: x(*)
      real(c_double), intent(in) :: local_model(*)
      real(c_double), intent(out) :: outvec(*)
      integer(c_int) :: rc
    end function
    function sml_bank_advance_all(bank, stream) bind(C, name="sml_bank_advance_all") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank, stream
      integer(c_int) :: rc
    end function
    ! ---- device memory helpers and the training kernels (include/speedyml_hip.h section 5) ----
    function sml_dev_alloc(bytes, dev) bind(C, name="sml_dev_alloc") result(rc)
      import :: c_int, c_int64_t, c_ptr
      integer(c_int64_t), value :: bytes
      type(c_ptr), intent(out) :: dev
      integer(c_int) :: rc
    end function
    function sml_dev_free(dev) bind(C, name="sml_dev_free") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: dev
      integer(c_int) :: rc
    end function
    function sml_dev_zero(dev, bytes) bind(C, name="sml_dev_zero") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: dev
      integer(c_int64_t), value :: bytes
      integer(c_int) :: rc
    end function
    function sml_dev_upload(dst_dev, src_host, bytes) bind(C, name="sml_dev_upload") result(rc)
      import :: c_int, c_int64_t, c_ptr, c_double
      type(c_ptr), value :: dst_dev
      real(c_double), intent(in) :: src_host(*)
      integer(c_int64_t), value :: bytes
      integer(c_int) :: rc
    end function
    function sml_dev_download(dst_host, src_dev, bytes) bind(C, name="sml_dev_download") result(rc)
      import :: c_int, c_int64_t, c_ptr, c_double
      real(c_double), intent(out) :: dst_host(*)
      type(c_ptr), value :: src_dev
      integer(c_int64_t), value :: bytes
      integer(c_int) :: rc
    end function
    function sml_train_accumulate(states_dev, model_dev, y_dev, n, n_model, n_out, m, c_dev, b_dev, stream) &
             bind(C, name="sml_train_accumulate") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: states_dev, model_dev, y_dev, c_dev, b_dev, stream
      integer(c_int), value :: n, n_model, n_out, m
      integer(c_int) :: rc
    end function
    function sml_train_fit(c_dev, b_dev, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout_dev, stream) &
             bind(C, name="sml_train_fit") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: c_dev, b_dev, wout_dev, stream
      integer(c_int), value :: n, n_model, n_out, using_prior
      real(c_double), value :: beta_res, beta_model, prior_val
      integer(c_int) :: rc
    end function
    ! ---- spectral handle + SPEEDY adiabatic time step (include/speedyml_hip.h sections 4 and 4a) ----
    function sml_spectral_create(a, sp) bind(C, name="sml_spectral_create") result(rc)
      import :: c_int, c_double, c_ptr
      real(c_double), value :: a
      type(c_ptr), intent(out) :: sp
      integer(c_int) :: rc
    end function
    function sml_dyn_create(sp, dyn) bind(C, name="sml_dyn_create") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: sp
      type(c_ptr), intent(out) :: dyn
      integer(c_int) :: rc
    end function
    function sml_dyn_impint(dyn, dt, alph) bind(C, name="sml_dyn_impint") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: dyn
      real(c_double), value :: dt, alph
      integer(c_int) :: rc
    end function
    function sml_dyn_state_dev(dyn, state_dev) bind(C, name="sml_dyn_state_dev") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: dyn
      type(c_ptr), intent(out) :: state_dev
      integer(c_int) :: rc
    end function
    function sml_dyn_set_state_host(dyn, vor, div, t, ps, tr) bind(C, name="sml_dyn_set_state_host") result(rc)
      import :: c_int, c_double_complex, c_ptr
      type(c_ptr), value :: dyn
      complex(c_double_complex), intent(in) :: vor(*), div(*), t(*), ps(*), tr(*)
      integer(c_int) :: rc
    end function
    function sml_dyn_get_state_host(dyn, vor, div, t, ps, tr) bind(C, name="sml_dyn_get_state_host") result(rc)
      import :: c_int, c_double_complex, c_ptr
      type(c_ptr), value :: dyn
      complex(c_double_complex), intent(out) :: vor(*), div(*), t(*), ps(*), tr(*)
      integer(c_int) :: rc
    end function
    function sml_dyn_set_boundary_host(dyn, phis, tcorh, qcorh) bind(C, name="sml_dyn_set_boundary_host") result(rc)
      import :: c_int, c_double_complex, c_ptr
      type(c_ptr), value :: dyn
      complex(c_double_complex), intent(in) :: phis(*), tcorh(*), qcorh(*)
      integer(c_int) :: rc
    end function
    function sml_dyn_step(dyn, state_dev, j1, j2, dt, alph, rob, wil, stream) bind(C, name="sml_dyn_step") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: dyn, state_dev, stream
      integer(c_int), value :: j1, j2
      real(c_double), value :: dt, alph, rob, wil
      integer(c_int) :: rc
    end function
    function sml_dyn_window(dyn, state_dev, start, nsteps, delt, alph, rob, wil, stream) bind(C, name="sml_dyn_window") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: dyn, state_dev, stream
      integer(c_int), value :: start, nsteps
      real(c_double), value :: delt, alph, rob, wil
      integer(c_int) :: rc
    end function
    ! ---- SPEEDY column physics (src/phy_phypar.f90 grid-point part), attached to the time step like grtend's phypar call ----
    function sml_phys_create(hsg9, rlat48, phys) bind(C, name="sml_phys_create") result(rc)
      import :: c_int, c_double, c_ptr
      real(c_double), intent(in) :: hsg9(*), rlat48(*)
      type(c_ptr), intent(out) :: phys
      integer(c_int) :: rc
    end function
    function sml_phys_destroy(phys) bind(C, name="sml_phys_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: phys
      integer(c_int) :: rc
    end function
    ! fmask1, phis0, stl_am, sst_am, soilw_am, alb_l, alb_s, albsfc, snowc: real(ix,il) host arrays
    function sml_phys_set_surface(phys, fmask, phis0, tland, tsea, swav, alb_l, alb_s, albsfc, snowc) &
        bind(C, name="sml_phys_set_surface") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: phys
      real(c_double), intent(in) :: fmask(*), phis0(*), tland(*), tsea(*), swav(*), alb_l(*), alb_s(*), albsfc(*), snowc(*)
      integer(c_int) :: rc
    end function
    function sml_phys_sol_oz(phys, tyear) bind(C, name="sml_phys_sol_oz") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: phys
      real(c_double), value :: tyear
      integer(c_int) :: rc
    end function
    function sml_phys_diag(phys, which, out_host) bind(C, name="sml_phys_diag") result(rc)
      import :: c_int, c_double, c_ptr
      type(c_ptr), value :: phys
      integer(c_int), value :: which
      real(c_double), intent(out) :: out_host(*)
      integer(c_int) :: rc
    end function
    function sml_dyn_attach_physics(dyn, phys, nstrad) bind(C, name="sml_dyn_attach_physics") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: dyn, phys
      integer(c_int), value :: nstrad
      integer(c_int) :: rc
    end function
    function sml_dyn_set_lradsw(dyn, lradsw) bind(C, name="sml_dyn_set_lradsw") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: dyn
      integer(c_int), value :: lradsw
      integer(c_int) :: rc
    end function
  end interface

contains

  ! The reference prints library status and stops (src/mod_linalg.f90:18-22,147-150); same here.
  subroutine sml_check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    integer :: i
    if (rc >= 0) return
    call c_f_pointer(sml_last_error(), msg, [512])
    write(*,'(a,a,a,i0,a)', advance='no') 'speedyml_hip: ', where, ' failed (status ', rc, '): '
    do i = 1, 512
      if (msg(i) == c_null_char) exit
      write(*,'(a)', advance='no') msg(i)
    end do
    write(*,*)
    stop 1
  end subroutine

end module speedyml_hip
