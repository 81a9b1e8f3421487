! iso_c_binding interfaces to libspeedyml_hip.so (include/speedyml_hip.h) -- the host language the reference is
! written in.  Only plain pointers, sizes and opaque handles cross the boundary.
module speedyml_hip
  use iso_c_binding
  implicit none
  ! sml_region / sml_res_sizes of include/speedyml_hip.h (interoperable mirrors of grid_type's extents and allocate_res_new's sizes)
  type, bind(C) :: sml_region
    integer(c_int32_t) :: res_xstart, res_xend, res_ystart, res_yend, resxchunk, resychunk
    integer(c_int32_t) :: res_zstart, res_zend, reszchunk
    integer(c_int32_t) :: input_xstart, input_xend, input_ystart, input_yend, inputxchunk, inputychunk
    integer(c_int32_t) :: input_zstart, input_zend, inputzchunk
    integer(c_int32_t) :: pole, periodicboundary, top, bottom
    integer(c_int32_t) :: tdata_xstart, tdata_xend, tdata_ystart, tdata_yend, tdata_zstart, tdata_zend
  end type
  type, bind(C) :: sml_res_sizes
    integer(c_int32_t) :: chunk_size, chunk_size_prediction, chunk_size_speedy, locality
    integer(c_int32_t) :: nodes_per_input, n, k, reservoir_numinputs
    integer(c_int32_t) :: atmo3d_start, atmo3d_end, logp_start, logp_end, precip_start, precip_end
    integer(c_int32_t) :: sst_start, sst_end, tisr_start, tisr_end
  end type


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
    function sml_device_synchronize() bind(C, name="sml_device_synchronize") result(rc)
      import :: c_int
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
    ! ---- added for the module-API drop-ins (mod_reservoir / resdomain / mpires) ----
    function sml_domain_region(number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, out) &
        bind(C, name="sml_domain_region") result(rc)
      import :: c_int, sml_region
      integer(c_int), value :: number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap
      type(sml_region), intent(out) :: out
      integer(c_int) :: rc
    end function
    function sml_domain_sizes(g, m, deg, local_predictvars, logp_bool, precip_bool, sst_bool_input, tisr_input_bool, ml_only, out) &
        bind(C, name="sml_domain_sizes") result(rc)
      import :: c_int, sml_region, sml_res_sizes
      type(sml_region), intent(in) :: g
      integer(c_int), value :: m, deg, local_predictvars, logp_bool, precip_bool, sst_bool_input, tisr_input_bool, ml_only
      type(sml_res_sizes), intent(out) :: out
      integer(c_int) :: rc
    end function
    function sml_domain_target_map(number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, precip_bool, in_pos, capacity) &
        bind(C, name="sml_domain_target_map") result(n)
      import :: c_int
      integer(c_int), value :: number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, precip_bool, capacity
      integer(c_int), intent(out) :: in_pos(*)
      integer(c_int) :: n
    end function
    function sml_find_closest_divisor(target, number) bind(C, name="sml_find_closest_divisor") result(d)
      import :: c_int
      integer(c_int), value :: target, number
      integer(c_int) :: d
    end function
    function sml_gen_res(n, k, radius, seed, rows, cols, vals, eigs) bind(C, name="sml_gen_res") result(rc)
      import :: c_int, c_double, c_int64_t
      integer(c_int), value :: n, k
      real(c_double), value :: radius
      integer(c_int64_t), value :: seed
      integer(c_int), intent(out) :: rows(*), cols(*)
      real(c_double), intent(out) :: vals(*), eigs
      integer(c_int) :: rc
    end function
    function sml_bank_set_wout(bank, slot, wout) bind(C, name="sml_bank_set_wout") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(in) :: wout(*)
      integer(c_int) :: rc
    end function
    function sml_bank_set_local_model(bank, slot, lm) bind(C, name="sml_bank_set_local_model") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(in) :: lm(*)
      integer(c_int) :: rc
    end function
    function sml_bank_get_outvec(bank, slot, out) bind(C, name="sml_bank_get_outvec") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(out) :: out(*)
      integer(c_int) :: rc
    end function
    function sml_bank_outvec_contribs(bank, stream) bind(C, name="sml_bank_outvec_contribs") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank, stream
      integer(c_int) :: rc
    end function
    function sml_bank_get_contribs(bank, slot, v_p, v_ml) bind(C, name="sml_bank_get_contribs") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot
      real(c_double), intent(out) :: v_p(*), v_ml(*)
      integer(c_int) :: rc
    end function
    function sml_bank_predict_all(bank, flags, stream) bind(C, name="sml_bank_predict_all") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank, stream
      integer(c_int), value :: flags
      integer(c_int) :: rc
    end function
    function sml_bank_synchronize_one(bank, slot, inputs, length, x) bind(C, name="sml_bank_synchronize_one") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: bank
      integer(c_int), value :: slot, length
      real(c_double), intent(in) :: inputs(*)
      real(c_double), intent(inout) :: x(*)
      integer(c_int) :: rc
    end function
    function sml_bank_train_pass(bank, noisy_inputs_dev, T, discard, batch, model_dev, targets_dev, c_dev, b_dev, ml_variant, stream) &
        bind(C, name="sml_bank_train_pass") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank, noisy_inputs_dev, stream
      integer(c_int), value :: T, discard, batch, ml_variant
      type(c_ptr), intent(in) :: model_dev(*), targets_dev(*), c_dev(*), b_dev(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_create(bank, number_of_regions, region_of_slot, nslots, overlap, precip_bool, sst_input_of_slot, h) &
        bind(C, name="sml_hybrid_create") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: bank
      integer(c_int), value :: number_of_regions, nslots, overlap, precip_bool
      integer(c_int), intent(in) :: region_of_slot(*), sst_input_of_slot(*)
      type(c_ptr), intent(out) :: h
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_state(h, g) bind(C, name="sml_hybrid_set_state") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: g(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_get_state(h, g, f) bind(C, name="sml_hybrid_get_state") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: g(*), f(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_orography(h, phi0) bind(C, name="sml_hybrid_set_orography") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: phi0(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_update_surface(h, stl_am, soilw_am, snowd_am, sice_am) bind(C, name="sml_hybrid_update_surface") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h, stl_am, soilw_am, snowd_am, sice_am          ! c_loc of (96,48) real(8) arrays, or c_null_ptr = unchanged
      integer(c_int) :: rc
    end function
    function sml_hybrid_get_phis0(h, phis0) bind(C, name="sml_hybrid_get_phis0") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: phis0(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_fordate_fields(h, fmask_s, alb0, snowd_am, sice_am) bind(C, name="sml_hybrid_set_fordate_fields") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: fmask_s(*), alb0(*), snowd_am(*), sice_am(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_tisr_table(h, tisr, start_hours, timestep_hours) bind(C, name="sml_hybrid_set_tisr_table") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: tisr(*)
      integer(c_int), value :: start_hours, timestep_hours
      integer(c_int) :: rc
    end function
    function sml_hybrid_attach_physics(h, hsg9, radang48, fmask, phis0, tland, swav, alb_l, alb_s, albsfc, snowc, nstrad) &
        bind(C, name="sml_hybrid_attach_physics") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: hsg9(*), radang48(*), fmask(*), phis0(*), tland(*), swav(*), alb_l(*), alb_s(*), albsfc(*), snowc(*)
      integer(c_int), value :: nstrad
      integer(c_int) :: rc
    end function
    function sml_hybrid_initial_inputs(h, stream) bind(C, name="sml_hybrid_initial_inputs") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h, stream
      integer(c_int) :: rc
    end function
    function sml_hybrid_exchange_and_speedy(h, all_outvec_dev, leapfrog_steps, stream) bind(C, name="sml_hybrid_exchange_and_speedy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h, all_outvec_dev, stream
      integer(c_int), value :: leapfrog_steps
      integer(c_int) :: rc
    end function
    function sml_hybrid_safe(h, safe) bind(C, name="sml_hybrid_safe") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), intent(out) :: safe
      integer(c_int) :: rc
    end function
    function sml_domain_in_map(number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, precip_bool, sst_bool_input, &
                               tisr_input_bool, g_index, stat_idx, capacity) bind(C, name="sml_domain_in_map") result(n)
      import :: c_int
      integer(c_int), value :: number_of_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, precip_bool, sst_bool_input, &
                               tisr_input_bool, capacity
      integer(c_int), intent(out) :: g_index(*), stat_idx(*)
      integer(c_int) :: n
    end function
    function sml_bank_feedback_dev(bank) bind(C, name="sml_bank_feedback_dev") result(p)
      import :: c_ptr
      type(c_ptr), value :: bank
      type(c_ptr) :: p
    end function
    function sml_bank_local_model_dev(bank) bind(C, name="sml_bank_local_model_dev") result(p)
      import :: c_ptr
      type(c_ptr), value :: bank
      type(c_ptr) :: p
    end function
    function sml_hybrid_attach_slab(h, slab_bank, sea_of_slot, sea_of_region, timestep_slab_hours) bind(C, name="sml_hybrid_attach_slab") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h, slab_bank
      integer(c_int), intent(in) :: sea_of_slot(*), sea_of_region(*)
      integer(c_int), value :: timestep_slab_hours
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_base_sst(h, base_sst, sea_mask) bind(C, name="sml_hybrid_set_base_sst") result(rc)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: base_sst(*)
      integer(c_int), intent(in) :: sea_mask(*)
      integer(c_int) :: rc
    end function
    function sml_hybrid_set_comm(h, comm) bind(C, name="sml_hybrid_set_comm") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h, comm
      integer(c_int) :: rc
    end function
    function sml_hybrid_restart(h, start_hours) bind(C, name="sml_hybrid_restart") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: start_hours
      integer(c_int) :: rc
    end function
    function sml_hybrid_slab_due(h) bind(C, name="sml_hybrid_slab_due") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function sml_hybrid_destroy(h) bind(C, name="sml_hybrid_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function sml_comm_bootstrap(nranks, rank, name, max_doubles_per_rank, comm) bind(C, name="sml_comm_bootstrap") result(rc)
      import :: c_int, c_ptr, c_char, c_int64_t
      integer(c_int), value :: nranks, rank
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), value :: max_doubles_per_rank
      type(c_ptr), intent(out) :: comm
      integer(c_int) :: rc
    end function
    function sml_comm_destroy(comm) bind(C, name="sml_comm_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
      integer(c_int) :: rc
    end function
    function sml_slab_sizes(g, m, deg, local_predictvars, out) bind(C, name="sml_slab_sizes") result(rc)
      import :: c_int, sml_region, sml_res_sizes
      type(sml_region), intent(in) :: g
      integer(c_int), value :: m, deg, local_predictvars
      type(sml_res_sizes), intent(out) :: out
      integer(c_int) :: rc
    end function
    function sml_slab_predict_hybrid(slab_bank, stream) bind(C, name="sml_slab_predict_hybrid") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: slab_bank, stream
      integer(c_int) :: rc
    end function
    function sml_bank_outvec_dev(bank) bind(C, name="sml_bank_outvec_dev") result(p)
      import :: c_ptr
      type(c_ptr), value :: bank
      type(c_ptr) :: p
    end function
    function sml_train_fit_batched(count, c_dev, b_dev, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout_dev, stream) &
        bind(C, name="sml_train_fit_batched") result(rc)
      import :: c_int, c_ptr, c_double
      integer(c_int), value :: count, n, n_model, n_out, using_prior
      type(c_ptr), intent(in) :: c_dev(*), b_dev(*), wout_dev(*)
      real(c_double), value :: beta_res, beta_model, prior_val
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function
    function sml_train_release_workspace() bind(C, name="sml_train_release_workspace") result(rc)
      import :: c_int
      integer(c_int) :: rc
    end function
    function sml_dev_upload_raw(dst_dev, src_host, bytes) bind(C, name="sml_dev_upload") result(rc)
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: dst_dev, src_host
      integer(c_int64_t), value :: bytes
      integer(c_int) :: rc
    end function
    function sml_dev_download_raw(dst_host, src_dev, bytes) bind(C, name="sml_dev_download") result(rc)
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: dst_host, src_dev
      integer(c_int64_t), value :: bytes
      integer(c_int) :: rc
    end function
  end interface

contains

  ! download `bytes` bytes starting `off` bytes into a device buffer
  function sml_dev_download_off(dst, src_dev, off, bytes) result(rc)
    real(c_double), intent(out), target :: dst(*)
    type(c_ptr), intent(in) :: src_dev
    integer(c_int64_t), intent(in) :: off, bytes
    integer(c_int) :: rc
    rc = sml_dev_download_raw(c_loc(dst), transfer(transfer(src_dev, 0_c_intptr_t) + off, src_dev), bytes)
  end function

  ! upload `bytes` bytes to `off` bytes into a device buffer
  function sml_dev_upload_off(dst_dev, off, src, bytes) result(rc)
    type(c_ptr), intent(in) :: dst_dev
    real(c_double), intent(in), target :: src(*)
    integer(c_int64_t), intent(in) :: off, bytes
    integer(c_int) :: rc
    rc = sml_dev_upload_raw(transfer(transfer(dst_dev, 0_c_intptr_t) + off, dst_dev), c_loc(src), bytes)
  end function

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
