/* speedyml_hip.h -- C-ABI of libspeedyml_hip.so, the MI355X (gfx950) implementation of the SPEEDY-ML
 * hybrid-step hot path.  Plain C: pointers, sizes and opaque handles only (no C++/torch types), so it
 * binds from Fortran (iso_c_binding -- see speedy-ml_amd/fortran/ and INTEGRATION.md), ctypes, cgo...
 *
 * Each entry point names the reference interface it replaces (file:line relative to the reference
 * repository root).  Conventions:
 *   - every function returns 0 on success, a negative sml_status otherwise; sml_last_error() gives text.
 *     (The reference has no error propagation -- MKL/LAPACK `info` is printed, then `stop`,
 *     src/mod_linalg.f90:18-22,147-150 -- the Fortran wrappers print sml_last_error() and `stop`.)
 *   - "host" pointers are ordinary CPU memory (what the Fortran driver owns); "dev" pointers are HIP
 *     device memory.  Arrays keep the reference's Fortran layout (column-major, 1-based index VALUES
 *     inside rows/cols) at the boundary; the library re-lays them out in HBM once, at load.
 *   - all floating point is IEEE fp64, all indices int32 (the reference's default integer).
 *   - stream arguments are hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef SPEEDYML_HIP_H
#define SPEEDYML_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sml_status {
    SML_OK = 0,
    SML_ERR_ARG = -1,      /* bad argument / shape mismatch */
    SML_ERR_HIP = -2,      /* a HIP runtime call failed (no device, OOM, launch failure) */
    SML_ERR_STATE = -3,    /* call order violated (e.g. predict before load) */
    SML_ERR_NUMERIC = -4   /* singular matrix in the ridge solve (dgesv info > 0) */
} sml_status;

const char *sml_last_error(void);
int sml_version(void);
/* number of visible HIP devices (0 on a CPU-only box; never initialises a context) */
int sml_device_count(void);
/* selects the GPU of this process.  One process drives one GPU (one rank per GPU): a later call with another ordinal returns
 * SML_ERR_STATE -- the library's cached streams, launch attributes and solver workspaces belong to the first device. */
int sml_set_device(int ordinal);
/* hipDeviceSynchronize for hosts without a HIP binding of their own (a Fortran host timing its loop) */
int sml_device_synchronize(void);
/* device memory for hosts without a HIP binding of their own (the Fortran drop-ins of speedy-ml_amd/fortran/): hipMalloc,
 * hipFree, hipMemset(0), synchronous hipMemcpy in either direction */
int sml_dev_alloc(uint64_t bytes, void **out_dev);
/* A HIP stream whose kernels run on a subset of the compute units (hipExtStreamCreateWithCUMask): mask bit i = CU i, nwords
 * 32-bit words (8 words cover the 256 CUs).  Used by the pipelined hybrid step to keep the HBM-streaming reservoir readout and the
 * latency-bound SPEEDY window off each other's CUs.  sml_stream_destroy is for streams made here only. */
int sml_stream_create_cu_mask(const uint32_t *mask, int nwords, void **stream_out);
int sml_stream_destroy(void *stream);
int sml_dev_free(void *dev);
int sml_dev_zero(void *dev, uint64_t bytes);
int sml_dev_upload(void *dst_dev, const void *src_host, uint64_t bytes);
int sml_dev_download(void *dst_host, const void *src_dev, uint64_t bytes);

/* ===================================================================================================
 * 1. resdomain: integer bookkeeping (host only, no GPU needed) -- replaces src/res_domain.f90
 * =================================================================================================== */
typedef struct sml_region {
    /* 1-based inclusive extents exactly as grid_type holds them (src/mod_utilities.f90 grid_type) */
    int32_t res_xstart, res_xend, res_ystart, res_yend, resxchunk, resychunk;
    int32_t res_zstart, res_zend, reszchunk;
    int32_t input_xstart, input_xend, input_ystart, input_yend, inputxchunk, inputychunk;
    int32_t input_zstart, input_zend, inputzchunk;
    int32_t pole, periodicboundary, top, bottom;
    int32_t tdata_xstart, tdata_xend, tdata_ystart, tdata_yend, tdata_zstart, tdata_zend;
} sml_region;

typedef struct sml_res_sizes {
    /* integer results of allocate_res_new (src/mod_reservoir.f90:80-180) and the u(t) segment offsets of
     * trained_reservoir_prediction (:1851-1885); offsets 1-based inclusive, 0/0 when the segment is absent */
    int32_t chunk_size, chunk_size_prediction, chunk_size_speedy, locality;
    int32_t nodes_per_input, n, k, reservoir_numinputs;
    int32_t atmo3d_start, atmo3d_end, logp_start, logp_end, precip_start, precip_end;
    int32_t sst_start, sst_end, tisr_start, tisr_end;
} sml_res_sizes;

/* processor_decomposition / processor_decomposition_manual (src/res_domain.f90:31-94).
 * Writes the regions owned by `rank`; returns their count (or <0). */
int sml_domain_decompose(int rank, int nranks, int number_of_regions, int32_t *region_indices, int capacity);
/* its inverse: the rank that owns `region` and the region's position in that rank's list (the remainder rule of
 * src/res_domain.f90:53-60 puts the tail regions at position number_of_regions / nranks of ranks 1..left_over) */
int sml_domain_region_owner(int nranks, int number_of_regions, int region, int *rank_out, int *slot_out);
/* initializedomain (src/res_domain.f90:96-121) and everything it calls (:123-292, :547-600) */
int sml_domain_region(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                      int vert_overlap, sml_region *out);
/* allocate_res_new sizing (src/mod_reservoir.f90:80-180) */
int sml_domain_sizes(const sml_region *g, int m, int deg, int local_predictvars, int logp_bool, int precip_bool,
                     int sst_bool_input, int tisr_input_bool, int ml_only, sml_res_sizes *out);

/* getsend_receive_size_{res,speedy,input,res_slab,input_slab} (src/mpires.f90:806-925) -> sizes5[0..4] */
int sml_domain_message_sizes(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                             int vert_overlap, int precip_bool, int ohtc_bool_input, int32_t *sizes5);
/* tile_full_input_to_target_data (src/res_domain.f90:602-689) as an index map: 0-based positions in the region's input
 * vector u(t) of the entries that form its target vector (chunk_size_prediction of them: (var,x,y,z) of the res patch, var
 * fastest, then logp, then precip when predicted).  Returns the count. */
int sml_domain_target_map(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                          int vert_overlap, int precip_bool, int32_t *in_pos, int capacity);
/* find_closest_divisor (src/mod_utilities.f90:1598-1636); returns the divisor (>0) or <0 on bad arguments */
int sml_find_closest_divisor(int target, int number);

/* Global state buffer ("G") layout used by the device-resident step loop (doubles):
 *   [0, 147456)            grid4d(4,96,48,8)  Fortran order: ((z*48+y)*96+x)*4+v
 *   [147456, +4608)        logp(96,48)        y*96+x
 *   [152064, +4608)        precip(96,48)
 *   [156672, +4608)        sst(96,48)
 *   [161280, +4608)        tisr(96,48)        (current hour's slice)
 */
enum { SML_G4_OFF = 0, SML_G2_OFF = 147456, SML_GP_OFF = 152064, SML_GS_OFF = 156672, SML_GT_OFF = 161280,
       SML_G_SIZE = 165888 };

/* Index maps that replace the slice-and-reshape tilers by precomputed gathers/scatters:
 *  - out map: outvec element i of region r -> index into G.  Same ordering as
 *    tile_full_grid_with_local_state_vec_res1d (src/res_domain.f90:791-826); its first chunk_size_speedy
 *    entries are also the gather map of tile_4d_and_logp_full_grid_to_local_res_vec (:1022-1053).
 *    stat_idx[i] = 0-based slot of mean/std used by (un)standardize_state_vec_res (:1270-1315,:1424-1475).
 *  - in map: input element j of region r -> index into G, ordering of
 *    tile_4d_and_logp_to_local_state_input (:1081-1125) followed by the sst and tisr segments
 *    (src/mpires.f90:586-599,752-773); stat_idx as used by standardize_state_vec_input (:1211-1268) and the
 *    precip/sst/tisr standardisation (src/mpires.f90:765-773).
 * Both return the number of entries written (or <0). */
int sml_domain_out_map(int number_of_regions, int region_num, int num_vert_levels, int vert_level, int vert_overlap,
                       int precip_bool, int32_t *g_index, int32_t *stat_idx, int capacity);
int sml_domain_in_map(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                      int vert_overlap, int precip_bool, int sst_bool_input, int tisr_input_bool,
                      int32_t *g_index, int32_t *stat_idx, int capacity);

/* ===================================================================================================
 * 2. reservoir bank: all reservoirs of one rank resident in HBM -- replaces the per-reservoir state of
 *    reservoir_type (src/mod_utilities.f90) and predict/synchronize (src/mod_reservoir.f90)
 * =================================================================================================== */
typedef struct sml_bank sml_bank;

/* capacity = number of reservoir slots; strides bound the per-slot feedback / local_model / outvec vectors */
int sml_bank_create(int capacity, int max_d, int max_n_model, int max_n_out, sml_bank **out);
int sml_bank_destroy(sml_bank *bank);

/* Load one trained reservoir into `slot`: what read_trained_res + allocate_res_new + mklsparse leave in
 * reservoir_type (src/mod_reservoir.f90:1783-1886, src/mod_linalg.f90:10-25).
 *   rows/cols/vals : COO, 1-based, unsorted, duplicates allowed (they accumulate, as in MKL_SPARSE_D_MV)
 *   win            : dense (n,d) column-major as the reference stores it; exact zeros are dropped at load
 *   wout           : (n_out, n_model+n) column-major
 *   mean/std       : nstat entries (36 for the atmosphere reservoirs)
 *   out_stat_idx   : n_out 0-based slots into mean/std for the fused un-standardisation (from sml_domain_out_map)
 */
int sml_bank_load(sml_bank *bank, int slot, int n, int d, int k, int n_model, int n_out,
                  const int32_t *rows, const int32_t *cols, const double *vals,
                  const double *win, const double *wout, double leakage,
                  const double *mean, const double *std, int nstat, const int32_t *out_stat_idx);
/* Same, with W_in given as COO triplets (1-based) instead of the 26.5 MB dense array. */
int sml_bank_load_sparse_win(sml_bank *bank, int slot, int n, int d, int k, int n_model, int n_out,
                             const int32_t *rows, const int32_t *cols, const double *vals,
                             int win_nnz, const int32_t *win_rows, const int32_t *win_cols, const double *win_vals,
                             const double *wout, double leakage,
                             const double *mean, const double *std, int nstat, const int32_t *out_stat_idx);
/* replace W_out of a slot (after training) */
int sml_bank_set_wout(sml_bank *bank, int slot, const double *wout);

/* host <-> device state access (x = reservoir state, in/out of predict; mod_reservoir.f90:1418) */
int sml_bank_set_state(sml_bank *bank, int slot, const double *x_host);
int sml_bank_get_state(sml_bank *bank, int slot, double *x_host);
int sml_bank_set_feedback(sml_bank *bank, int slot, const double *u_host);          /* reservoir%feedback(d) */
int sml_bank_set_local_model(sml_bank *bank, int slot, const double *lm_host);     /* reservoir%local_model */
int sml_bank_get_outvec(sml_bank *bank, int slot, double *out_host);               /* reservoir%outvec */

/* device views for the resident loop: [capacity][stride] row-major */
double *sml_bank_feedback_dev(sml_bank *bank);    /* stride max_d */
double *sml_bank_local_model_dev(sml_bank *bank); /* stride max_n_model */
double *sml_bank_outvec_dev(sml_bank *bank);      /* stride max_n_out */

/* predict (src/mod_reservoir.f90:1418-1489) for EVERY loaded slot in one batched pass:
 *   x <- (1-leak) x + leak tanh(A x + Win u);  outvec <- unstandardize(Wout [local_model ; x with even entries squared])
 * flags: bit0 = skip the un-standardisation (raw W_out product, for tests). */
int sml_bank_predict_all(sml_bank *bank, int flags, void *stream);
/* predict for one slot with the reference's calling shape (x in/out on the host) -- the drop-in for a
 * per-region call from program main (src/parallelmain.f90:233). */
int sml_bank_predict_one(sml_bank *bank, int slot, double *x_inout_host, const double *local_model_host,
                         double *outvec_host);
/* synchronize (src/mod_reservoir.f90:1354-1381) for every loaded slot: `length` teacher-forced steps.
 * inputs_dev: [length][capacity][max_d] (step-major) device array. */
/* synchronize for ONE slot of a shared bank with the reference's host arrays: inputs(d, length) column-major, x(n) in/out.  Only
 * that slot is stepped (the per-region calls of initialize_prediction / start_prediction, src/mod_reservoir.f90:822-824,949-951). */
int sml_bank_synchronize_one(sml_bank *bank, int slot, const double *inputs_host, int length, double *x_inout);
int sml_bank_synchronize_all(sml_bank *bank, const double *inputs_dev, int length, void *stream);
/* one advance without readout (K1-K3) for every slot, feedback taken from the bank */
int sml_bank_advance_all(sml_bank *bank, void *stream);
/* The readout product of predict (src/mod_reservoir.f90:1450-1471) in two column blocks, because in the hybrid loop the
 * reservoir state is known a whole SPEEDY window before the forecast is (DESIGN.md "Software pipeline"):
 *   part 1: W_out[:, n_model:] x~  (99 % of the bytes) -> an internal buffer, needs only the advanced state;
 *   part 2: + W_out[:, :n_model] local_model, un-standardise -> outvec.  part 1 then part 2 == the readout of
 * sml_bank_predict_all up to the association of the column sum.  flags: bit0 as for sml_bank_predict_all.
 * Part 1 only: bit3 (8) runs the block as a persistent work-queue kernel with a bounded footprint (one 8-wave workgroup
 * per CU, <= 128 VGPRs, no LDS) that leaves room for kernels of other streams -- use it when part 1 is overlapped with
 * the SPEEDY window; bit4 (16) launches a full-occupancy kernel that drains the SAME work queue (call it, on any stream,
 * once the overlapped work is done; part 1 is complete when both launches have finished). */
int sml_bank_readout_part(sml_bank *bank, int part, int flags, void *stream);
/* predict's split readout, `outvec_component_contribs` (src/mod_reservoir.f90:1458-1461): after a predict of every slot (state
 * advanced, local_model still the step's), v_ml = W_out[:, n_model:] x~ and v_p = W_out[:, :n_model] local_model of every slot, both
 * left standardised as the reference leaves them; v_p + v_ml = the readout before unstandardize_state_vec_res (<= 1e-13: the column
 * sum is associated differently).  get_contribs: host copies of one slot's two vectors (either may be NULL). */
int sml_bank_outvec_contribs(sml_bank *bank, void *stream);
int sml_bank_get_contribs(sml_bank *bank, int slot, double *v_p, double *v_ml);

/* byte accounting for the roofline (algorithmic bytes as defined in DESIGN.md) */
int sml_bank_algorithmic_bytes(sml_bank *bank, uint64_t *update_bytes, uint64_t *readout_bytes);
/* Compact storage.  The reference's weight files hold win / wout / vals / mean / std as NF90_REAL (src/mod_io.f90, written by
 * write_trained_res, src/mod_reservoir.f90:1727-1736): a reservoir loaded from them has weights that are exactly floats.  sml_bank_load /
 * sml_bank_set_wout detect that (every value of W_out and of the operator must survive the round trip through float) and keep a second,
 * 4-byte copy; when EVERY loaded reservoir of a bank has one, the predict kernels read those copies and convert back on the fly -- the
 * same numbers, the same double-precision arithmetic, half the bytes of the two HBM-bound kernels.  The readout then sums four columns
 * per 16-byte load instead of two, i.e. in another (fixed) association: results agree with the 8-byte kernels to ~1e-13, not bit for
 * bit.  *compact = 1 when the bank is in that mode.  SML_BANK_COMPACT=0 disables it; sml_bank_algorithmic_bytes counts the bytes of
 * the copies actually read. */
int sml_bank_storage(sml_bank *bank, int *compact);
/* allow = 0: this bank's predict kernels keep to the 8-byte copies whatever the weights are; 1 (default): automatic */
int sml_bank_use_compact(sml_bank *bank, int allow);
/* the same accounting for one column block of sml_bank_readout_part (the partial sums count as traffic of both parts) */
int sml_bank_readout_part_bytes(sml_bank *bank, int part, uint64_t *bytes);
/* Per-kernel timing with HIP events recorded on the launch stream (for bench.py's roofline block).
 * enable!=0 starts recording an event pair around every k_update / k_readout launch (enable == 2: around k_readout only -- an event
 * record is a packet the dependent launches queue behind); collect synchronises the
 * recorded events, returns the summed milliseconds and launch counts since the last collect, and clears them. */
int sml_bank_timing(sml_bank *bank, int enable);
int sml_bank_timing_collect(sml_bank *bank, double *update_ms, int *update_launches, double *readout_ms, int *readout_launches);

/* ===================================================================================================
 * 3. exchange: the device-resident form of sendrecievegrid (src/mpires.f90:218-804) without MPI/NetCDF
 * =================================================================================================== */
typedef struct sml_exchange sml_exchange;

/* Build the per-slot index maps for the regions resident in `bank` (region_of_slot[capacity]). */
int sml_exchange_create(sml_bank *bank, int number_of_regions, const int32_t *region_of_slot, int nslots,
                        int overlap, int precip_bool, const int32_t *sst_input_of_slot /* 0/1 per slot */,
                        sml_exchange **out);
int sml_exchange_destroy(sml_exchange *ex);
/* outvec slab of this rank -> global slab position (for the all-gather) then G: scatter + clamps (Appendix G 2-3,
 * src/mpires.f90:309-330,460-490).  all_outvec_dev: [number_of_regions][max_n_out] in REGION order. */
int sml_exchange_scatter(sml_exchange *ex, const double *all_outvec_dev, double *g_dev,
                         const double *base_sst_dev /* 4608 or NULL */, const int32_t *sea_mask_dev /* 4608 or NULL */,
                         void *stream);
/* G (+ current TISR slice already in G) -> standardised feedback of every slot; forecast F (same layout as the
 * first two parts of G) -> standardised local_model of every slot (Appendix G 5-6, src/mpires.f90:580-775). */
/* Either pointer may be NULL to skip that half (the feedback is available before the forecast is). */
int sml_exchange_gather(sml_exchange *ex, const double *g_dev, const double *f_dev, void *stream);
/* where each slot's outvec goes in the region-ordered slab: offset = region*max_n_out */
int sml_exchange_pack_outvec(sml_exchange *ex, double *all_outvec_dev, void *stream);

/* Hybrid <-> SPEEDY hand-off around iogrid modes 30/31 (src/ppo_iogrid.f90:497-601):
 *  to_fields  : G -> 33 grid fields [T(8) | u(8) | v(8) | q(8) | ps(1)][48][96], rounded through real(4) exactly as the
 *               reference's ugr4..psgr4 staging does (quirk Q3, :43-44,500-517) and with q < 0 -> 0 (:511-513);
 *  from_fields: 33 grid fields -> F (layout of the first two parts of G), with SPEEDY's output floor q < 1e-6 -> 1e-6
 *               (src/mpires.f90:1648-1650);
 *  check      : the physical-range guard |u|<=150, |v|<=120, 160<=T<=330, -6<=q<=30 (:563-577); *safe_dev (int32 on
 *               the device) is set to 0 when the state is unsafe, left untouched otherwise. */
int sml_handoff_to_fields(const double *g_dev, double *fields_dev, void *stream);
int sml_handoff_from_fields(const double *fields_dev, double *f_dev, void *stream);
int sml_handoff_check(const double *fields_dev, int32_t *safe_dev, void *stream);

/* ---- RCCL from the C-ABI, for a multi-rank host that is not Python (Fortran under MPI): the step's one data-path collective.
 * One rank calls sml_comm_unique_id and distributes the 128 bytes (MPI_Bcast); every rank then calls sml_comm_create.
 * sml_comm_allgather_outvec: every rank ends up with all_outvec_dev [number_of_regions][max_n_out], the region-ordered slab of
 * sml_exchange_scatter (the gather-to-root + root-side tiling of src/mpires.f90:347-454).  When the region count divides by the
 * rank count and the bank holds exactly its share, this is ONE ncclAllGather of the bank's outvec slab straight into the result
 * (no packing).  With the remainder of processor_decomposition (src/res_domain.f90:53-60: ranks 1..left_over own one extra
 * region from the tail) every rank contributes number_of_regions / nranks + 1 slots through a padded staging slab and
 * sml_comm_unpack_regions puts the rows in region order.  The bank's slot i must hold the rank's i-th region
 * (sml_domain_decompose order).  librccl.so is resolved at the first call (dlopen), not at link time.
 * sml_comm_unpack_regions: the reordering alone (stage_dev [nranks][slots_per_rank][max_n_out] -> region order); no RCCL. */
typedef struct sml_comm sml_comm;
int sml_comm_unique_id(char *id128);
int sml_comm_create(int nranks, int rank, const char *id128, sml_comm **out);
/* For a host without MPI (this image has no Fortran MPI module; the drop-in's startmpi takes rank and rank count from the launcher's
 * environment): every rank of ONE node calls this with the same `name`.  SML_COMM_TRANSPORT=rccl (default): rank 0 draws the RCCL id
 * and leaves it in /dev/shm/<name>.id for the others -- what MPI_Bcast does in an MPI build -- then sml_comm_create.
 * SML_COMM_TRANSPORT=shm: a host-staged all-gather through a POSIX shared-memory segment, for REHEARSING the multi-rank path with
 * several ranks on one GPU (RCCL refuses two ranks on one device); never used for numbers.  max_doubles_per_rank bounds one rank's
 * contribution to a collective (shm only; 0 = 512 Ki doubles).
 * Leftovers of an earlier run under the same name are never picked up: the id file / segment carries a per-launch token --
 * SML_COMM_NONCE when the launcher exports one (a job id), else the ranks' common parent process id -- and a peer waits for one with its
 * own token (SML_COMM_TIMEOUT_S, default 120); rank 0 removes the name as soon as the collective init / first barrier has returned, and
 * an exit handler removes whatever a dying process had published. */
int sml_comm_bootstrap(int nranks, int rank, const char *name, uint64_t max_doubles_per_rank, sml_comm **out);
int sml_comm_destroy(sml_comm *comm);
int sml_comm_allgather_outvec(sml_comm *comm, sml_bank *bank, int number_of_regions, double *all_outvec_dev, void *stream);
int sml_comm_unpack_regions(const double *stage_dev, int nranks, int slots_per_rank, int number_of_regions, int max_n_out,
                            double *all_outvec_dev, void *stream);

/* ---- the device-resident body of sendrecievegrid as one native engine (for hosts that are not Python: the Fortran drop-in
 * speedy-ml_amd/fortran/mpires.f90).  Everything the reference's root does between the predict calls of two consecutive steps
 * (src/mpires.f90:218-804): tile every region's outvec into the global grids + clamps, run_model -> agcm_main (iogrid(30), stepone +
 * the leapfrog steps of one window with phypar inside grtend, iogrid(31)), get_tisr_by_date, tile + standardise the next feedback /
 * local_model of every reservoir resident in `bank` (slot i holds region_of_slot[i]).  Global state G = grid4d(4,96,48,8) | logp |
 * precip | sst | tisr (SML_G4_OFF ...); F = SPEEDY's forecast in the same layout.
 *   set_orography : phi0(96,48) -> phis = trunct(spec(phi0)), phis0 = grid(phis) (src/ini_invars.f90:31-34; get_phis0 returns it for
 *                   the physics' phis0), tcorh = spec(gamlat phis0) (src/ini_fordate.f90:72-86)
 *   set_tisr_table: full_tisr [8760][48][96] (src/mod_reservoir.f90:890-909) + hours since 1 Jan 1981 00h of the first step
 *   attach_physics: phypar inside every time step with these surface fields ((96,48) each); sst_am = G's SST grid.  From then on
 *                   every window starts with fordate(0) on the device (sml_phys_fordate: tcorh, and qcorh from the hybrid SST;
 *                   fmask_s = 1 - fmask unless set_fordate_fields gives it; with alb0 / snowd_am / sice_am also the albedos)
 *   initial_inputs: TISR slice 0 into G, feedback and local_model of every slot gathered from G
 *   exchange_and_speedy: all_outvec_dev = region-ordered slab [number_of_regions][max_n_out] after the all-gather, or NULL to take
 *                   the bank's own outvec buffer; leapfrog_steps = 24 for the 6-hour window (< 0: hand-off only)
 *   safe          : run_speedy (src/mpires.f90:744): 1 while iogrid(30)'s range guard has not tripped; synchronises */
typedef struct sml_hybrid sml_hybrid;
int sml_hybrid_create(sml_bank *bank, int number_of_regions, const int32_t *region_of_slot, int nslots, int overlap, int precip_bool,
                      const int32_t *sst_input_of_slot, sml_hybrid **out);
int sml_hybrid_destroy(sml_hybrid *h);
int sml_hybrid_set_state(sml_hybrid *h, const double *g_host);
int sml_hybrid_get_state(sml_hybrid *h, double *g_host, double *f_host);
int sml_hybrid_set_base_sst(sml_hybrid *h, const double *base_sst, const int32_t *sea_mask);
int sml_hybrid_set_orography(sml_hybrid *h, const double *phi0_grid);
int sml_hybrid_set_tisr_table(sml_hybrid *h, const double *tisr_8760x48x96, int start_hours, int timestep_hours);
int sml_hybrid_attach_physics(sml_hybrid *h, const double *hsg9, const double *radang48, const double *fmask, const double *phis0,
                              const double *tland, const double *swav, const double *alb_l, const double *alb_s, const double *albsfc,
                              const double *snowc, int nstrad);
int sml_hybrid_get_phis0(sml_hybrid *h, double *phis0_host);
int sml_hybrid_set_fordate_fields(sml_hybrid *h, const double *fmask_s, const double *alb0, const double *snowd_am, const double *sice_am);
/* the coupler's daily output between two windows (land temperature, soil wetness, snow depth, sea-ice fraction; (96,48) each, NULL =
 * unchanged): the coupler stays with the host, its results enter here and the next window's fordate and physics read them */
int sml_hybrid_update_surface(sml_hybrid *h, const double *stl_am, const double *soilw_am, const double *snowd_am, const double *sice_am);
int sml_hybrid_initial_inputs(sml_hybrid *h, void *stream);
int sml_hybrid_exchange_and_speedy(sml_hybrid *h, const double *all_outvec_dev, int leapfrog_steps, void *stream);
int sml_hybrid_safe(sml_hybrid *h, int *safe_out);
/* Device time of each phase of the steps taken through sml_hybrid_step while timing is on (HIP events on the step's stream), summed
 * over *steps: ms5 = predict (+ predict_slab_ml when due) | rank exchange | scatter + clamps (+ SST assembly) | SPEEDY leg (iogrid(30),
 * fordate, the window, iogrid(31)) | TISR slice + gather + standardise (+ slab inputs).  collect synchronises and resets the sums. */
int sml_hybrid_timing(sml_hybrid *h, int on);
int sml_hybrid_timing_collect(sml_hybrid *h, double *ms5, int *steps);
/* ---- the `ocean_model` branches and the rank exchange inside the engine (src/mpires.f90:286-330,347-454,470-484,756-790) ----
 *   attach_slab : the rank's slab-ocean bank beside its atmosphere bank (slot i of both = region_of_slot[i]; slab slots are loaded
 *                 for SST-predicting regions only).  sea_of_slot[nslots] / sea_of_region[number_of_regions] = sst_bool_prediction;
 *                 timestep_slab_hours = 168 as shipped.  Every later exchange assembles wholegrid_sst from the slab reservoirs' last
 *                 outputs (272 K where a region has none), applies the mask / floor, and keeps the slab inputs' averaging ring.
 *   set_comm    : with a communicator (sml_comm_create / sml_comm_bootstrap) an exchange whose all_outvec_dev is NULL all-gathers the
 *                 banks' outvec buffers (sml_comm_allgather_outvec) instead of placing this rank's rows only.  Not owned.
 *   restart     : a new forecast (program main's prediction_num loop): step counter, forcing calendar and range guard start over
 *   slab_due    : 1 when mod(t * timestep, timestep_slab) == 0 for the step about to be taken (src/parallelmain.f90:238)
 *   step        : one whole iteration of program main's t loop for this rank: predict of every resident reservoir, predict_slab_ml
 *                 of the slab bank when due, then exchange_and_speedy(NULL, leapfrog_steps) */
int sml_hybrid_attach_slab(sml_hybrid *h, sml_bank *slab_bank, const int32_t *sea_of_slot, const int32_t *sea_of_region, int timestep_slab_hours);
int sml_hybrid_set_comm(sml_hybrid *h, sml_comm *comm);
int sml_hybrid_restart(sml_hybrid *h, int start_hours);
int sml_hybrid_slab_due(sml_hybrid *h);
int sml_hybrid_step(sml_hybrid *h, int leapfrog_steps, void *stream);
/* the same iteration in two calls for a host that owns the rank exchange (its own all-gather of the banks' outvec buffers in between);
 * all_outvec_dev = the gathered region-ordered slab [number_of_regions][max_n_out], NULL = the engine's own exchange */
int sml_hybrid_step_predict(sml_hybrid *h, void *stream);
int sml_hybrid_step_finish(sml_hybrid *h, const double *all_outvec_dev, int leapfrog_steps, void *stream);
double *sml_hybrid_g_dev(sml_hybrid *h);
double *sml_hybrid_f_dev(sml_hybrid *h);

/* ---- slab-ocean coupling (config 5): the `ocean_model` branches of sendrecievegrid, src/mpires.f90:286-330, 470-484,
 * 756-790; sizes of initialize_slab_ocean_model, src/mod_slab_ocean_reservoir.f90:9-133.  The slab reservoirs live in a second
 * sml_bank (n_model = 0, every output un-standardised with the SST statistics as predict_slab_ml does, :1318-1363) and are
 * stepped with sml_bank_predict_all every timestep_slab/timestep-th atmosphere step (src/parallelmain.f90:237-249). */
typedef struct sml_slab sml_slab;
/* d, n, k, chunk sizes and the input segment offsets of a region's slab reservoir (m = 4000, deg = 6 as shipped) */
int sml_slab_sizes(const sml_region *g, int m, int deg, int local_predictvars, sml_res_sizes *out);
/* sea_of_slot: sst_bool_prediction per slot; atmo_sst_input_of_slot as given to sml_exchange_create; ring = timestep_slab/timestep - 1 */
int sml_slab_create(sml_bank *atmo_bank, sml_bank *slab_bank, int number_of_regions, const int32_t *region_of_slot, int nslots,
                    const int32_t *sea_of_slot, const int32_t *atmo_sst_input_of_slot, int ring, sml_slab **out);
int sml_slab_destroy(sml_slab *slab);
/* wholegrid_sst before the mask and floor: region r writes its res patch from all_slab_out[r][0:resx*resy] (REGION order,
 * stride out_stride) if sea_of_region[r], 272 K otherwise (src/mpires.f90:309-330).  Call it BEFORE sml_exchange_scatter,
 * whose SST kernel then restores base_sst where sea_mask > 0 and applies the 272 K floor (:470-484). */
/* predict_slab (src/mod_slab_ocean_reservoir.f90:1268-1316), the hybrid slab ocean (ml_only_ocean = .false.): slots loaded with
 * n_model = n_out; advance + readout of every loaded slot, local_model <- the raw (standardised) output, outvec un-standardised */
int sml_slab_predict_hybrid(sml_bank *slab_bank, void *stream);
int sml_slab_scatter_sst(sml_slab *slab, const double *all_slab_out_dev, int out_stride, const int32_t *sea_of_region_dev, double *g_dev,
                         void *stream);
/* after sml_exchange_gather: ring column (timestep-1) mod ring <- atmo_training_data_idx entries of the atmosphere feedback;
 * slab feedback <- mean of the ring columns (src/mpires.f90:776-781).  timestep is sendrecievegrid's 1-based step. */
int sml_slab_update_inputs(sml_slab *slab, int timestep, void *stream);

/* The hybrid's calendar (src/mod_calendar.f90:24-175) and the TISR slice it selects (get_tisr_by_date,
 * src/mpires.f90:1676-1708): integer bookkeeping, quirks included.  sml_tisr_index returns the 1-based slice (1..8760) of
 * the hour-of-365-day-year table for `hours_elapsed` since 1 January `startyear` 00h (the reference starts at 1981). */
int sml_calendar_date(int startyear, int hours_elapsed, int32_t *date_out /* year, month, day, hour */);
int sml_hours_into_year(int year, int month, int day, int hour);
int sml_tisr_index(int startyear, int hours_elapsed);

/* ===================================================================================================
 * 4. spectral transforms -- replaces src/spe_spectral.f90 + src/spe_subfft_fftpack.f90 (FFTPACK)
 * =================================================================================================== */
typedef struct sml_spectral sml_spectral;
/* parmtr(a) + inifft (src/spe_spectral.f90:45-192, src/spe_subfft_fftpack.f90:1-12): tables built on the host in
 * fp64 and uploaded once. */
int sml_spectral_create(double a, sml_spectral **out);
int sml_spectral_destroy(sml_spectral *sp);
/* copy a table back to the host (numbering as in oracle/ref_spectral_driver.f90: 1 sia .. 24 cpol) */
int sml_spectral_get_table(sml_spectral *sp, int which, double *out_host, int capacity);

/* Batched device entry points: nf fields per launch (never one transform per launch).
 *   spec arrays: [nf][32][62] (= Fortran vorm(mx2,nx) per field), grid arrays: [nf][48][96] (= vorg(ix,il)).
 *   kcos: per call (1 or 2), as in grid(vorm,vorg,kcos) (src/spe_spectral.f90:389-401). */
int sml_spectral_grid(sml_spectral *sp, const double *vorm_dev, double *vorg_dev, int nf, int kcos, void *stream);
int sml_spectral_spec(sml_spectral *sp, const double *vorg_dev, double *vorm_dev, int nf, void *stream);      /* :403-414 */
/* One launch for a whole transform SET of a SPEEDY time step, where fields differ in kcos (grid(.,.,1) next to
 * grid(.,.,2), src/dyn_grtend.f90:61-99) or in the forward pre-scaling (plain spec next to the two specx of a vdspec,
 * :237-277): per-field int32 flags on the device.  kcos: 1|2.  scale: 0 none, 1 *cosgr(j) (vdspec kcos=2), 2 *cosgr2(j). */
int sml_spectral_grid_mixed(sml_spectral *sp, const double *vorm_dev, double *vorg_dev, int nf, const int32_t *kcos_dev, void *stream);
int sml_spectral_spec_mixed(sml_spectral *sp, const double *vorg_dev, double *vorm_dev, int nf, const int32_t *scale_dev, void *stream);
/* Inverse transforms of DERIVED fields in one launch, without materialising uvspec / grad first (the transform set of
 * grtend, src/dyn_grtend.f90:61-99, and of iogrid(31), src/ppo_iogrid.f90:582-593).  desc_dev: int32 [nf][4] on the device,
 * (type, src0, src1, kcos) per output field, src = field index into spec_base_dev ([.][32][62]):
 *   type 0: field src0;  1 | 2: ucos | vcos of uvspec(vor = src0, div = src1);  3 | 4: d/dx | d/dy of grad(src0). */
int sml_spectral_grid_derived(sml_spectral *sp, const double *spec_base_dev, const int32_t *desc_dev, double *vorg_dev, int nf, void *stream);
/* ... plus type 7: the geopotential of level src1 (0-based) from the 8 temperature levels that start at field src0,
 * src/dyn_geop.f90:19-35, for phypar's grid(phi1) (src/phy_phypar.f90:61-65) without a separate geop pass.
 * aux_dev: xgeop1(8) | xgeop2(8) | corf(8) | phis(62x32) on the device (needed only when a type-7 row is present). */
int sml_spectral_grid_derived_aux(sml_spectral *sp, const double *spec_base_dev, const int32_t *desc_dev, const double *aux_dev,
                                  double *vorg_dev, int nf, void *stream);
/* The forward counterpart: after a (mixed) forward transform, form the output fields in one launch.  desc_dev: int32
 * [nf_out][4] = (type, src0, src1, truncate): type 0 = field src0; 5 | 6 = vor | div of vds(ucos = src0, vcos = src1)
 * (:307-349); truncate != 0 applies trunct (:540-551).  iogrid(30)'s vdspec/spec/trunct (src/ppo_iogrid.f90:530-547) is
 * sml_spectral_spec_mixed + this. */
int sml_spectral_spec_post(sml_spectral *sp, const double *spec_in_dev, const int32_t *desc_dev, double *spec_out_dev, int nf_out, void *stream);
/* the same with the last nf_out2 output fields written to a second array (desc_dev has nf_out + nf_out2 rows): the hybrid engine
 * transforms fordate's two correction fields (spec(corh, tcorh), spec(corh, qcorh), src/ini_fordate.f90:86,113) in iogrid(30)'s
 * launch and wants them in the time steps' boundary arrays */
int sml_spectral_spec_post_split(sml_spectral *sp, const double *spec_in_dev, const int32_t *desc_dev, double *spec_out_dev, int nf_out,
                                 double *spec_out2_dev, int nf_out2, void *stream);
int sml_spectral_vdspec(sml_spectral *sp, const double *ug_dev, const double *vg_dev, double *vorm_dev,
                        double *divm_dev, int nf, int kcos, void *stream);                                    /* :416-452 */
int sml_spectral_uvspec(sml_spectral *sp, const double *vorm_dev, const double *divm_dev, double *ucosm_dev,
                        double *vcosm_dev, int nf, void *stream);                                             /* :351-387 */
int sml_spectral_vds(sml_spectral *sp, const double *ucosm_dev, const double *vcosm_dev, double *vorm_dev,
                     double *divm_dev, int nf, void *stream);                                                 /* :307-349 */
int sml_spectral_grad(sml_spectral *sp, const double *psi_dev, double *psdx_dev, double *psdy_dev, int nf, void *stream); /* :271-305 */
int sml_spectral_lap(sml_spectral *sp, const double *strm_dev, double *vorm_dev, int nf, void *stream);      /* :244-254 */
int sml_spectral_invlap(sml_spectral *sp, const double *vorm_dev, double *strm_dev, int nf, void *stream);   /* :256-269 */
int sml_spectral_trunct(sml_spectral *sp, double *vor_dev, int nf, void *stream);                             /* :540-551 */

/* Link-level drop-ins with the reference's external F77 symbols (trailing underscore, everything by
 * reference, host arrays): src/spe_spectral.f90:244-551.  They use a process-global sml_spectral created
 * by parmtr_()/inifft_(). */
void parmtr_(const double *a);
void inifft_(void);
void grid_(const double *vorm, double *vorg, const int *kcos);
void spec_(const double *vorg, double *vorm);
void vdspec_(const double *ug, const double *vg, double *vorm, double *divm, const int *kcos);
void uvspec_(const double *vorm, const double *divm, double *ucosm, double *vcosm);
void vds_(const double *ucosm, const double *vcosm, double *vorm, double *divm);
void grad_(const double *psi, double *psdx, double *psdy);
void lap_(const double *strm, double *vorm);
void invlap_(const double *vorm, double *strm);
void trunct_(double *vor);

/* ===================================================================================================
 * 4a. SPEEDY time step on the device -- replaces, inside the hybrid window between iogrid(30) and
 *     iogrid(31), src/dyn_step.f90 (step, hordif, timint), src/dyn_grtend.f90 (its phypar call included once a physics
 *     handle of section 4c is attached; the adiabatic core otherwise), src/dyn_sptend.f90, src/dyn_geop.f90, src/dyn_implic.f90, and the set-up routines
 *     src/ini_indyns.f90, src/ini_impint.f90 (+ src/spe_matinv.f90), src/ini_stepone.f90, src/dyn_stloop.f90:28-43.
 *     State (device, caller-owned): double state[2][33][32][62] = time level (mod_dynvar.f90's last index), then the
 *     fields vor(8) | div(8) | t(8) | tr(:,:,:,1)(8) | ps, each a Fortran complex (mx,nx) array.
 *     Tendencies: double tend[33][32][62] in the same field order.
 * =================================================================================================== */
typedef struct sml_dyn sml_dyn;
typedef struct sml_phys sml_phys;      /* column physics, declared further down */
/* indyns (src/ini_indyns.f90): level and diffusion tables from the spectral handle's Gaussian latitudes */
int sml_dyn_create(sml_spectral *sp, sml_dyn **out);
int sml_dyn_destroy(sml_dyn *dyn);
/* impint(dt, alph) (src/ini_impint.f90): the semi-implicit tables become current for the following steps; tables of
 * every (dt, alph) seen are kept on the device, so alternating dt (stepone) costs no upload after the first window. */
int sml_dyn_impint(sml_dyn *dyn, double dt, double alph);
/* host copy of a table, numbered as oracle/ref_dyn_driver.f90: 1 hsg .. 11 dmps (indyns), 12 dmp1 .. 25 elz (current
 * impint), 26 alph */
int sml_dyn_get_table(sml_dyn *dyn, int which, double *out_host, int capacity);
/* phis (mod_surfcon / mod_dynvar), tcorh, qcorh (mod_hdifcon.f90:19, set by ini_fordate.f90:86,113): spectral [32][62] */
int sml_dyn_set_boundary(sml_dyn *dyn, const double *phis_dev, const double *tcorh_dev, const double *qcorh_dev, void *stream);
/* the handle's own device copy, [3][32][62] = phis | tcorh | qcorh, for producers that write it in place (sml_phys_fordate) */
double *sml_dyn_boundary_dev(sml_dyn *dyn);
/* For a Fortran host that keeps the prognostic variables in mod_dynvar's arrays (src/mod_dynvar.f90:14-27): the handle owns a
 * device state; *_host copy complex vor/div/t(mx,nx,kx,2), ps(mx,nx,2) and the first tracer tr(mx,nx,kx,2) in and out, and
 * phis/tcorh/qcorh(mx,nx).  Synchronous.  sml_dyn_state_dev returns the device state to pass to step/window. */
int sml_dyn_state_dev(sml_dyn *dyn, double **state_dev);
int sml_dyn_set_state_host(sml_dyn *dyn, const double *vor, const double *div, const double *t, const double *ps, const double *tr);
int sml_dyn_get_state_host(sml_dyn *dyn, double *vor, double *div, double *t, double *ps, double *tr);
int sml_dyn_set_boundary_host(sml_dyn *dyn, const double *phis, const double *tcorh, const double *qcorh);
/* grtend(vordt,divdt,tdt,psdt,trdt,1,j2) without physics (src/dyn_grtend.f90): tend_dev receives the 33 tendencies */
int sml_dyn_grtend(sml_dyn *dyn, const double *state_dev, int j2, double *tend_dev, void *stream);
/* the rest of step() from given grid-point tendencies (src/dyn_step.f90:45-127): sptend [+ implic when alph != 0],
 * hordif, and when dt > 0 timint on both time levels.  tend_dev is updated in place to the diffused tendencies. */
int sml_dyn_spectral_step(sml_dyn *dyn, double *state_dev, double *tend_dev, int j1, int j2, double dt, double alph, double rob,
                          double wil, void *stream);
/* step(j1,j2,dt,alph,rob,wil) (src/dyn_step.f90:1-128): four launches, state updated in place */
int sml_dyn_step(sml_dyn *dyn, double *state_dev, int j1, int j2, double dt, double alph, double rob, double wil, void *stream);
/* start != 0: stepone (src/ini_stepone.f90: impint(delt/2), step(1,1,delt/2), impint(delt), step(1,2,delt)); then
 * impint(2 delt) and nsteps leapfrog steps step(2,2,2 delt) (src/dyn_stloop.f90:28-43) */
int sml_dyn_window(sml_dyn *dyn, double *state_dev, int start, int nsteps, double delt, double alph, double rob, double wil,
                   void *stream);
/* Column physics inside grtend (src/dyn_grtend.f90:222-225: geop(j1); phypar(...)): once attached, every time step also
 * transforms time level 1 to the 27 of phypar's 41 grids that the parametrisations read (same launch as grtend's 50; the
 * winds above the lowest level are never used) and adds sml_phys_tendencies_sfcwind to the grid-point tendencies before the
 * forward transforms.  phys == NULL detaches.  nstrad: short-wave radiation every nstrad-th leapfrog
 * step of a window, lradsw = (mod(istep, nstrad) == 1) (src/dyn_stloop.f90:39); sml_dyn_set_lradsw sets the flag that single
 * sml_dyn_step / sml_dyn_grtend calls and a window's stepone use (the module variable lradsw, src/mod_lflags.f90:22). */
int sml_dyn_attach_physics(sml_dyn *dyn, sml_phys *phys, int nstrad);
int sml_dyn_set_lradsw(sml_dyn *dyn, int lradsw);
/* iogrid(30)'s physical-range guard (src/ppo_iogrid.f90:563-577) without an inverse set of its own: a window that starts with
 * stepone (start != 0) checks the grids of its first time step -- T, q, u, v of the state just handed over -- and clears
 * *safe_dev (int32 on the device, set to 1 by the caller) when a value is outside the range or NaN.  NULL switches it off. */
int sml_dyn_set_range_guard(sml_dyn *dyn, int32_t *safe_dev);
/* whether time steps keep the physics' 2-D diagnostics (sml_phys_diag: precipitation, fluxes, cloud cover ...) up to date: on by
 * default; a host that does not read them between windows can switch the 18 stores per column and step off */
int sml_dyn_physics_diag(sml_dyn *dyn, int on);
/* how a time step runs grtend's grid-point part with physics attached: 3 = one fused launch of three wavefronts per 64 columns --
 * grid-point dynamics | convection, condensation, vertical diffusion, final sums | radiation and surface fluxes (default since round 4),
 * 1 = the two-wavefront launch of rounds 2-3, 0 = the grid-point dynamics and sml_phys_tendencies_sfcwind as two launches, 2 = one
 * fused one-wavefront launch (same arithmetic and bits in all four; kept so that tests can compare them) */
int sml_dyn_select_physics_form(int fused);
/* how sml_dyn_window runs a time step: 0 = four launches over whole fields (default), 1 = two kernels (zonal-wavenumber
 * space <-> latitude space; bit-identical results, measured slower on MI355X, see csrc/dynamics.hip), -1 = default /
 * environment SML_DYN_TWO_KERNEL */
int sml_dyn_select_window_form(int form);

/* ===================================================================================================
 * 4c. SPEEDY column physics on the device -- replaces the grid-point part of phypar (src/phy_phypar.f90:80-230) and the
 *     parametrisations it calls: shtorh, convmf, lscond, cloud, radsw, radlw, suflux, vdifsc (src/phy_*.f90), with the
 *     set-up routines inphys, radset, sflset, sol_oz.  One kernel, one thread per grid column.  The surface state (land-sea
 *     mask, orography, land/sea temperatures, soil wetness, albedos, snow cover: the daily output of the reference's coupler
 *     and of fordate) is an input.  Grids are [nf][48][96] as everywhere.
 * =================================================================================================== */
/* inphys(hsg, ., rlat) + radset: hsg9 = sigma half levels 0..8 (src/ini_indyns.f90:38-41), rlat48 = Gaussian latitudes in
 * radians, south to north (src/ini_indyns.f90:72-80) */
int sml_phys_create(const double *hsg9, const double *rlat48, sml_phys **out);
int sml_phys_destroy(sml_phys *phys);
/* host arrays [48][96]: fmask1, phis0 (mod_surfcon), stl_am, sst_am, soilw_am (mod_var_land / mod_var_sea), alb_l, alb_s,
 * albsfc, snowc (mod_radcon, set by fordate); sflset(phis0) is applied here */
int sml_phys_set_surface(sml_phys *phys, const double *fmask, const double *phis0, const double *tland, const double *tsea,
                         const double *swav, const double *alb_l, const double *alb_s, const double *albsfc, const double *snowc);
/* the coupler's daily output (stl_am, soilw_am; snowd_am, sice_am for fordate's albedos), host [48][96], each optional */
int sml_phys_update_surface(sml_phys *phys, const double *tland, const double *swav, const double *snowd_am, const double *sice_am);
/* the hybrid model's SST grid (G's SST segment, device) becomes sst_am */
int sml_phys_set_sst_dev(sml_phys *phys, const double *tsea_dev, void *stream);
/* ... or read in place: later launches take the sea temperature straight from tsea_dev ([48][96] doubles on the device, e.g. the
 * SST segment of the hybrid state), no copy per step.  NULL returns to the handle's own copy. */
int sml_phys_bind_sst_dev(sml_phys *phys, const double *tsea_dev);
/* fordate(0)'s per-window work (src/ini_fordate.f90), which the hybrid repeats at the start of every 6-hour window through
 * agcm_init (src/ini_agcm_init.f90:86) with sst_am = the ML-predicted SST:
 *   :54-61   snowc, alb_l, alb_s, albsfc from alb0 (mod_surfcon), snowd_am, sice_am -- when those three are given (all or none;
 *            otherwise the albedos of sml_phys_set_surface stay);
 *   :72-86   tcorh = spec(gamlat phis0), gamlat = gamma / (1000 g) (setgam :116-136);
 *   :88-113  qcorh = spec(refrh1 (q_sat(tref, 1) - q_sat(tsfc, psfc))), tsfc = fmask_l stl_am + fmask_s sst_am, tref = tsfc + gamlat phis0,
 *            psfc = (tsfc / tref)^(1 / (rd gamlat)); psfc_dummy = 1.0 and the absence of trunct as in the reference.
 * fmask_l, phis0, stl_am, sst_am are the handle's fmask, phis0, tland and (bound) tsea.  set_fordate_fields: host arrays (96,48),
 * fmask_s = mod_cli_sea's sea fraction (src/ini_inbcon.f90:148-157).  sml_phys_fordate: one grid-point launch + one two-field
 * spec; corh_spec_dev [2][32][62] receives tcorh | qcorh (e.g. sml_dyn_boundary_dev(dyn) + 32*62). */
int sml_phys_set_fordate_fields(sml_phys *phys, const double *fmask_s, const double *alb0, const double *snowd_am, const double *sice_am);
int sml_phys_fordate(sml_phys *phys, sml_spectral *sp, double *corh_spec_dev, void *stream);
/* host copy of one surface field [48][96]: 0 fmask 1 phis0 2 tland 3 tsea (the handle's own copy) 4 swav 5 alb_l 6 alb_s 7 albsfc
 * 8 snowc 9 forog; 10, 11 = fordate's grid fields corh (temperature, humidity).  Synchronises. */
int sml_phys_get_surface(sml_phys *phys, int which, double *out_host);
/* sol_oz(tyear) (src/phy_radiat.f90:1-83): zonal solar / ozone fields for the day, tyear = fraction of the year */
int sml_phys_sol_oz(sml_phys *phys, double tyear);
/* the same enqueued on a stream (the values travel as kernel arguments): no host synchronisation -- what the hybrid engine calls once per model day */
int sml_phys_sol_oz_async(sml_phys *phys, double tyear, void *stream);
/* host copies for tests: zonal [6][48] = fsol ozone ozupp zenit stratz sqrt(clat); fband [301][4]; levels [9][9] = sig sigl
 * dsig sigh grdsig grdscp wvi(:,2) wvi(:,1) entr with level index 1..8 (NULL to skip any) */
int sml_phys_get_tables(sml_phys *phys, double *zonal_host, double *fband_host, double *levels_host);
/* phypar's grid-point sequence for every column.  grids_dev [41][48][96]: ug1 vg1 tg1 qg1 phig1 (8 levels each) and pslg1 of
 * time level 1 (src/phy_phypar.f90:54-66).  The tendencies utend vtend ttend qtend are 8 consecutive fields each of tend_dev,
 * starting at fields off_u off_v off_t off_q; accumulate != 0 adds the physics to what is there (the dynamical tendencies),
 * in the reference's order.  lradsw: this is a short-wave step (every nstrad = 3rd, src/dyn_stloop.f90:36). */
int sml_phys_tendencies(sml_phys *phys, const double *grids_dev, int lradsw, double *tend_dev, int off_u, int off_v, int off_t, int off_q,
                        int accumulate, void *stream);
/* The same with the inputs the parametrisations actually read: of the winds only the lowest level enters (suflux,
 * src/phy_suflux.f90:104-116), so grids_dev is [27][48][96] = ug1(:,kx) vg1(:,kx) tg1(8) qg1(8) phig1(8) pslg1 and 14 of
 * phypar's 41 inverse transforms need not be done at all.  want_diag == 0 skips the 2-D diagnostics (sml_phys_diag). */
int sml_phys_tendencies_sfcwind(sml_phys *phys, const double *grids_dev, int lradsw, double *tend_dev, int off_u, int off_v, int off_t,
                                int off_q, int accumulate, int want_diag, void *stream);
/* per-column diagnostics of the last call, host [48][96]: 0 precnv 1 precls 2 cbmf 3 ts 4 tskin 5 ssrd 6 slrd 7 olr 8 shf 9 evap
 * 10 ustr 11 vstr 12 cloudc 13 clstr 14 tsr 15 ssr 16 slr 17 hfluxn(land) 18 hfluxn(sea) 19 t0 20 q0 21 iptop 22 icltop */
int sml_phys_diag(sml_phys *phys, int which, double *out_host);

/* ===================================================================================================
 * 4b. reservoir construction (host, set-up time) -- replaces gen_res / makesparse / shuffle / sparse_eigen
 *     (src/mod_reservoir.f90:182-212, src/mod_linalg.f90:180-514, src/mod_utilities.f90:1569-1596)
 * =================================================================================================== */
/* makesparse: k COO entries, rows and cols each a concatenation of random permutations of 1..n, vals ~ U(0,1) */
int sml_makesparse(int n, int k, uint64_t seed, int32_t *rows, int32_t *cols, double *vals);
/* the same construction on uniform deviates in [0,1) the caller supplies, consumed in the reference's order: RANDOM_NUMBER(vals)
 * first (k of them), then one per iteration of every shuffle call (rows before cols inside each block of n) --
 * sml_makesparse_draws(n, k) of them in all.  A host with the reference's own RANDOM_NUMBER stream gets its own matrices. */
long sml_makesparse_draws(int n, int k);
int sml_makesparse_from_draws(int n, int k, const double *draws, long ndraws, int32_t *rows, int32_t *cols, double *vals);
/* largest-magnitude eigenvalue of the (non-negative) COO matrix by power iteration: what sparse_eigen asks ARPACK-NG for
 * (dnaupd / dneupd, 'LM', src/mod_linalg.f90:351,405; the library is not in the image).  Quirk Q4 -- the reference's
 * eigs = maxval(d) also scans the imaginary-part and residual columns of a partly uninitialised d(30,3), :246,511 -- reads
 * undefined memory and is not reproduced: the Perron root is what that code intends, and what tests/test_genres.py checks against
 * scipy's ARPACK to 1e-8. */
int sml_spectral_radius(int n, int k, const int32_t *rows, const int32_t *cols, const double *vals, double tol, int maxit,
                        double *lambda, int *iterations);
/* gen_res: makesparse, then vals <- vals / lambda_max * radius */
int sml_gen_res(int n, int k, double radius, uint64_t seed, int32_t *rows, int32_t *cols, double *vals, double *eigs);

/* ===================================================================================================
 * 5. training -- replaces chunking_matmul / fit_chunk_hybrid / mldivide
 *    (src/mod_reservoir.f90:1645-1701, 1235-1334; src/mod_linalg.f90:109-151)
 * =================================================================================================== */
/* C(n_aug,n_aug) += aug*aug^T and B(n_out,n_aug) += Y*aug^T with aug = [model ; states], fp64 MFMA.
 * All device, column-major as in the reference: states (n,m), model (n_model,m), y (n_out,m).
 * Long products (m >= 256, even n / n_model / n_out, 16-byte aligned arrays) run as ONE launch of 256 x 128 tiles over all five
 * products; its workgroups add into C and B with device-memory fp64 atomics, so c_dev and b_dev must be ordinary (coarse-grained)
 * hipMalloc allocations -- as every buffer of this library is -- not host-pinned or managed memory.  Calls share one scratch for the
 * K-split tail, so concurrent calls must be on ONE stream (as the reference's single-threaded rank issues them). */
int sml_train_accumulate(const double *states_dev, const double *model_dev, const double *y_dev,
                         int n, int n_model, int n_out, int m, double *c_dev, double *b_dev, void *stream);
/* reservoir_layer_chunking_hybrid (src/mod_reservoir.f90:1067-1175) for EVERY loaded slot of a bank, one pass over T
 * input columns: x <- 0; `discard` warm-up steps; then one advance per column, the (squared-even)
 * state stored as a column of the slot's states(n, batch) buffer; every `batch` columns C += aug aug^T, B += Y aug^T.
 *   noisy_inputs_dev : [T][capacity][max_d]   inputs with the training noise already applied (the reference's noise
 *                      comes from the compiler RNG and is not reproducible -- SURVEY H5)
 *   model_dev[slot]  : (n_model, T) column-major "imperfect model" forecasts;  targets_dev[slot] : (n_out, T)
 *   c_dev[slot]      : (n_aug, n_aug), b_dev[slot] : (n_out, n_aug), accumulated (lower-triangle tiles of C)
 * The four pointer tables are HOST arrays of device pointers, one entry per slot (NULL entries are skipped).
 * ml_variant != 0: reservoir_layer_chunking_ml (:963-1065), whose step after a batch flush multiplies A by the column with
 * the squared even entries while the leak term keeps the state (quirk Q6); the hybrid loop restarts from saved_state.
 * Returns the number of batches flushed (or <0). */
int sml_bank_train_pass(sml_bank *bank, const double *noisy_inputs_dev, int T, int discard, int batch,
                        const double *const *model_dev, const double *const *targets_dev,
                        double *const *c_dev, double *const *b_dev, int ml_variant, void *stream);

/* sml_train_accumulate updates only the tiles of C on or below the diagonal (half the flops and half the C traffic of
 * the reference's full DGEMM); this mirrors them into the upper triangle (sml_train_fit calls it itself). */
int sml_train_symmetrize(double *c_dev, int n_aug, void *stream);
/* fit_chunk_hybrid (src/mod_reservoir.f90:1235-1334): regularise the diagonal, solve C^T Z = (B + prior)^T, wout = Z^T.
 * wout_dev: (n_out, n_aug) column-major.  Synchronises the stream.
 * Solver.  The regularised Gram matrix is symmetric positive definite by construction (C = sum aug aug^T plus a positive diagonal), so
 * the default is a blocked Cholesky of C's lower triangle -- half the flops and half the traffic of an LU, no pivot search, no row
 * interchanges (SURVEY 2.2 lists it as an allowed form; W_out parity is defined by backward error, H4).  A pivot that is not positive
 * (an indefinite matrix, or one singular to working precision) sends that system through the pivoted LU instead, which is dgesv's
 * algorithm (first maximum per column, row interchange), as mldivide is (src/mod_linalg.f90:109-151); sml_train_select_solver(1) makes
 * the LU the only solver.  c_dev is symmetrised in place either way (sml_train_accumulate fills the lower-triangle tiles only).
 * No size limits: the LU's register-resident panel kernel holds 7168 rows and taller panels go through a slower in-memory leaf; any
 * n_out (the back substitution runs in groups of 136 right-hand sides).
 * Returns SML_ERR_NUMERIC when a pivot of the LU is exactly zero (dgesv info > 0).
 * Reproducibility: under the Cholesky this one-system entry (latency form: the back substitution's far rows by vector FMAs in the
 * solving launch) and sml_train_fit_batched (far rows by an MFMA product through the mirrored factor) are DIFFERENT arithmetics for
 * n_aug > 256 -- both at a backward error of 3-4e-17, last bits apart.  Only the batched entry gives the same bits whatever the number
 * of systems per call; a host that needs W_out reproducible across group sizes or rank counts (the Fortran training queue, training.py)
 * uses the batched entry for every system, also for a single one, and leaves SML_CHOL_BACKSUB_GEMM / SML_CHOL_GROUP / SML_CHOL_FUSED unset. */
int sml_train_fit(double *c_dev, const double *b_dev, int n, int n_model, int n_out, double beta_res, double beta_model,
                  double prior_val, int using_prior, double *wout_dev, void *stream);

/* Several ridge solves of equal size at once (host arrays of device pointers): up to 16 systems advance in lockstep through ONE chain
 * of launches (the panel chain of a single factorisation is latency-bound; its trailing updates fill the chip only together).
 * One arithmetic for every count: W_out has the same bits whether the systems come one per call or all together.  sml_train_fit is
 * the latency form of ONE solve: the same factorisation, but its back substitution fuses the far rows' update into the solve launch
 * (vector FMAs) where this entry point runs it as a matrix-core product; the two W_out agree to rounding (backward error of both
 * <= 4e-17 on the 5892-row system of config 4). */
int sml_train_fit_batched(int count, double *const *c_dev, const double *const *b_dev, int n, int n_model, int n_out,
                          double beta_res, double beta_model, double prior_val, int using_prior, double *const *wout_dev,
                          void *stream);
/* 0 = Cholesky, LU where it breaks down (default; environment SML_FIT_SOLVER=auto|lu|chol presets it); 1 = pivoted LU only;
 * 2 = Cholesky only (SML_ERR_NUMERIC for a system that is not positive definite); < 0 = query.  Returns the previous setting. */
int sml_train_select_solver(int solver);
/* sml_train_fit[_batched] keep their device scratch (the row-major system, panel buffers, streams: ~310 MB per factorisation in
 * flight at n_aug = 5892) between calls; this frees it. */
int sml_train_release_workspace(void);

#ifdef __cplusplus
}
#endif
#endif
