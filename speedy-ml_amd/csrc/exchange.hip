// Region exchange, device-resident: the part of sendrecievegrid (src/mpires.f90:218-804) that is neither MPI
// plumbing nor NetCDF output, re-designed as gathers/scatters through precomputed int32 index maps.
//
// Reference flow per step:
//   2. every region's outvec is tiled into wholegrid4d / wholegrid2d / wholegrid_precip
//      (tile_full_grid_with_local_state_vec_res1d, src/res_domain.f90:791-826; root loop src/mpires.f90:309-454)
//   3. clamps on the assembled grids: q < 1e-6 -> 1e-6 (:460-462), SST < 272 -> 272 (:480-484), precip < 1e-5 -> 0 (:486-490),
//      SST <- base_sst_grid where sea_mask > 0 (:470-478)
//   5. feedback(1:d-n_tisr-n_sst) <- tile_4d_and_logp_to_local_state_input(G4,G2,GP) (:586);
//      local_model <- standardize_state_vec_res(tile_4d_and_logp_full_grid_to_local_res_vec(F4,F2)) (:589-591)
//   6. tisr / sst segments, then (v-mean)/std of every segment (:752-773)
// Here: one scatter kernel over all regions' outvec elements with the clamps applied to the value being written
// (every grid cell is written by exactly one region, so clamping at the source equals clamping the assembled
// grid), one SST kernel, and one gather kernel per target (feedback, local_model) that standardises on the fly:
// subtract, then divide -- two roundings, as standardize_data_given_pars1d (src/mod_utilities.f90:1319-1329).
#include <vector>

#include "bank.h"

struct sml_exchange {
    sml_bank *bank = nullptr;
    int number_of_regions = 0, nslots = 0, out_stride = 0, in_stride = 0, lm_stride = 0;
    int32_t *d_out_map = nullptr;      // [number_of_regions][out_stride]  G index, -1 = none
    int32_t *d_in_map = nullptr;       // [nslots][in_stride]              G index, -1 = none
    int32_t *d_in_stat = nullptr;      // [nslots][in_stride]              mean/std slot, -1 = copy
    int32_t *d_lm_map = nullptr;       // [nslots][lm_stride]              F index
    int32_t *d_lm_stat = nullptr;
    int32_t *d_region_of_slot = nullptr;
    std::vector<int32_t> region_of_slot;
};

namespace {

using sml::ResDesc;

__global__ void k_scatter(const double *__restrict__ all_out, int out_stride, const int32_t *__restrict__ out_map, int total,
                          double *__restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int gi = out_map[t];
    if (gi < 0) return;
    double v = all_out[t];
    if (gi < SML_G2_OFF) {
        if ((gi & 3) == 3 && v < 0.000001) v = 0.000001;                 // specific humidity floor (mpires.f90:460-462)
    } else if (gi >= SML_GP_OFF && gi < SML_GS_OFF) {
        if (v < 0.00001) v = 0.0;                                       // precip (:486-490)
    }
    g[gi] = v;
}

__global__ void k_sst(double *__restrict__ g, const double *__restrict__ base_sst, const int32_t *__restrict__ sea_mask)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 4608) return;
    double v = g[SML_GS_OFF + c];
    // wholegrid_sst starts as base_sst_grid (:290); cells of regions with a slab model are overwritten by the caller
    // before this kernel; sea_mask > 0 restores the base value (:470-478); then the 272 K floor (:480-484)
    if (base_sst && (!sea_mask || sea_mask[c] > 0)) v = base_sst[c];
    if (v < 272.0) v = 272.0;
    g[SML_GS_OFF + c] = v;
}

// dst[slot][j] = (src[map[slot][j]] - mean[stat]) / std[stat]
__global__ void k_gather(const ResDesc *__restrict__ descs, const double *__restrict__ src, const int32_t *__restrict__ map,
                         const int32_t *__restrict__ stat, int stride, int nslots, double *__restrict__ dst, int dst_stride)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y;
    if (j >= stride || slot >= nslots) return;
    const int gi = map[(size_t)slot * stride + j];
    if (gi < 0) return;
    const ResDesc &D = descs[slot];
    double v = src[gi];
    const int si = stat[(size_t)slot * stride + j];
    if (si >= 0) {
        v = __dsub_rn(v, D.mean[si]);
        v = v / D.stdv[si];
    }
    dst[(size_t)slot * dst_stride + j] = v;
}

// both gathers of a step (feedback from G, local_model from F) in one launch: blockIdx.z picks the pair of maps
struct GatherArgs { const double *src; const int32_t *map, *stat; double *dst; int stride; };
__global__ void k_gather2(const ResDesc *__restrict__ descs, GatherArgs a0, GatherArgs a1, int nslots)
{
    const GatherArgs &a = blockIdx.z == 0 ? a0 : a1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y;
    if (j >= a.stride || slot >= nslots) return;
    const int gi = a.map[(size_t)slot * a.stride + j];
    if (gi < 0) return;
    const ResDesc &D = descs[slot];
    double v = a.src[gi];
    const int si = a.stat[(size_t)slot * a.stride + j];
    if (si >= 0) {
        v = __dsub_rn(v, D.mean[si]);
        v = v / D.stdv[si];
    }
    a.dst[(size_t)slot * a.stride + j] = v;
}

__global__ void k_pack(const double *__restrict__ slab, int stride, const int32_t *__restrict__ region_of_slot, int nslots,
                       double *__restrict__ all_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y;
    if (j >= stride || slot >= nslots) return;
    all_out[(size_t)region_of_slot[slot] * stride + j] = slab[(size_t)slot * stride + j];
}

// ---- hybrid <-> SPEEDY hand-off (src/ppo_iogrid.f90:497-601) ----
constexpr int NGP = 96 * 48, NFIELD = 33;

// field f: 0..7 T(k), 8..15 u(k), 16..23 v(k), 24..31 q(k), 32 ps ; G4 variable order is (T,u,v,q) (:499-505)
__global__ void k_to_fields(const double *__restrict__ g, double *__restrict__ fields)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NFIELD * NGP) return;
    const int f = t / NGP, p = t % NGP;
    double v;
    if (f < 32) {
        const int var = f >> 3, k = f & 7;
        v = g[SML_G4_OFF + ((size_t)k * NGP + p) * 4 + var];
    } else {
        v = g[SML_G2_OFF + p];
    }
    float v4 = (float)v;                           // ugr4..psgr4 are real(4): quirk Q3
    if (f >= 24 && f < 32 && v4 < 0.0f) v4 = 0.0f;  // where(qgr4 < 0.0) qgr4 = 0.0
    fields[t] = (double)v4;
}

__global__ void k_from_fields(const double *__restrict__ fields, double *__restrict__ fo)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NFIELD * NGP) return;
    const int f = t / NGP, p = t % NGP;
    double v = fields[t];
    if (f < 32) {
        const int var = f >> 3, k = f & 7;
        if (var == 3 && v < 0.000001) v = 0.000001;          // run_model, src/mpires.f90:1648-1650
        fo[SML_G4_OFF + ((size_t)k * NGP + p) * 4 + var] = v;
    } else {
        fo[SML_G2_OFF + p] = v;
    }
}

__global__ void k_check(const double *__restrict__ fields, int32_t *__restrict__ safe)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * NGP) return;
    const int var = (t / NGP) >> 3;
    const double v = fields[t];
    bool bad;
    if (var == 0) bad = v < 160.0 || v > 330.0;
    else if (var == 1) bad = v < -150.0 || v > 150.0;
    else if (var == 2) bad = v < -120.0 || v > 120.0;
    else bad = v < -6.0 || v > 30.0;
    if (bad || v != v) *safe = 0;
}

// The end of a hybrid step in ONE launch (the engine's form of k_from_fields + the TISR copy + k_gather2, three launches of 4-9 us
// each with their boundaries): blocks [0, nblk_fields) write SPEEDY's forecast F from iogrid(31)'s 33 grid fields (k_from_fields) and
// the TISR slice of the next step into G; the blocks after them, one per resident reservoir, gather + standardise its next feedback
// (from G) and local_model (from F).  The gathering blocks do not wait for the first group: what they would read from F or from G's TISR
// segment they form themselves from the fields / the slice, with the same expressions, so both groups hold the same values.
__global__ __launch_bounds__(256) void k_egress(const ResDesc *__restrict__ descs, const double *__restrict__ fields, const double *__restrict__ tisr_slice,
                                                 double *__restrict__ g, double *__restrict__ fo, GatherArgs a0, GatherArgs a1, int nblk_fields, int nslots)
{
    if ((int)blockIdx.x < nblk_fields) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        if (t < NFIELD * NGP) {
            const int f = t / NGP, p = t % NGP;
            double v = fields[t];
            if (f < 32) {
                const int var = f >> 3, k = f & 7;
                if (var == 3 && v < 0.000001) v = 0.000001;          // run_model, src/mpires.f90:1648-1650
                fo[SML_G4_OFF + ((size_t)k * NGP + p) * 4 + var] = v;
            } else {
                fo[SML_G2_OFF + p] = v;
            }
        } else if (t < (NFIELD + 1) * NGP && tisr_slice) {
            g[SML_GT_OFF + t - NFIELD * NGP] = tisr_slice[t - NFIELD * NGP];
        }
        return;
    }
    const int slot = blockIdx.x - nblk_fields;
    if (slot >= nslots) return;
    const ResDesc &D = descs[slot];
    for (int j = threadIdx.x; j < a0.stride + a1.stride; j += 256) {
        const bool first = j < a0.stride;
        const GatherArgs &a = first ? a0 : a1;
        const int jj = first ? j : j - a0.stride;
        const int gi = a.map[(size_t)slot * a.stride + jj];
        if (gi < 0) continue;
        double v;
        if (first) {
            v = (gi >= SML_GT_OFF && tisr_slice) ? tisr_slice[gi - SML_GT_OFF] : a.src[gi];
        } else if (gi < SML_G2_OFF) {
            const int var = gi & 3, rest = gi >> 2, k = rest / NGP, p = rest % NGP;
            v = fields[(size_t)(var * 8 + k) * NGP + p];
            if (var == 3 && v < 0.000001) v = 0.000001;
        } else if (gi < SML_GP_OFF) {
            v = fields[(size_t)32 * NGP + gi - SML_G2_OFF];
        } else {
            v = a.src[gi];                                           // (segments of F this launch does not write)
        }
        const int si = a.stat[(size_t)slot * a.stride + jj];
        if (si >= 0) {
            v = __dsub_rn(v, D.mean[si]);
            v = v / D.stdv[si];
        }
        a.dst[(size_t)slot * a.stride + jj] = v;
    }
}

}  // namespace

extern "C" {

int sml_handoff_to_fields(const double *g_dev, double *fields_dev, void *stream)
{
    SML_REQUIRE(g_dev && fields_dev, "sml_handoff_to_fields: null pointer");
    hipLaunchKernelGGL(k_to_fields, dim3((NFIELD * NGP + 255) / 256), dim3(256), 0, sml::as_stream(stream), g_dev, fields_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_handoff_from_fields(const double *fields_dev, double *f_dev, void *stream)
{
    SML_REQUIRE(f_dev && fields_dev, "sml_handoff_from_fields: null pointer");
    hipLaunchKernelGGL(k_from_fields, dim3((NFIELD * NGP + 255) / 256), dim3(256), 0, sml::as_stream(stream), fields_dev, f_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_handoff_check(const double *fields_dev, int32_t *safe_dev, void *stream)
{
    SML_REQUIRE(safe_dev && fields_dev, "sml_handoff_check: null pointer");
    hipLaunchKernelGGL(k_check, dim3((32 * NGP + 255) / 256), dim3(256), 0, sml::as_stream(stream), fields_dev, safe_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_exchange_create(sml_bank *bank, int number_of_regions, const int32_t *region_of_slot, int nslots,
                        int overlap, int precip_bool, const int32_t *sst_input_of_slot, sml_exchange **out)
{
    SML_REQUIRE(bank && region_of_slot && out && nslots > 0 && nslots <= bank->capacity, "sml_exchange_create: bad arguments");
    sml_exchange *ex = new sml_exchange;
    ex->bank = bank; ex->number_of_regions = number_of_regions; ex->nslots = nslots;
    ex->out_stride = bank->max_n_out; ex->in_stride = bank->max_d; ex->lm_stride = bank->max_n_model;
    ex->region_of_slot.assign(region_of_slot, region_of_slot + nslots);

    std::vector<int32_t> omap((size_t)number_of_regions * ex->out_stride, -1);
    std::vector<int32_t> tmp_g(8 * 96 * 48 * 8), tmp_s(8 * 96 * 48 * 8);
    for (int r = 0; r < number_of_regions; ++r) {
        int n = sml_domain_out_map(number_of_regions, r, 1, 1, 0, precip_bool, tmp_g.data(), tmp_s.data(), (int)tmp_g.size());
        if (n < 0) { delete ex; return n; }
        if (n > ex->out_stride) { delete ex; return sml::fail(SML_ERR_ARG, "sml_exchange_create: region %d has %d outputs > bank stride %d", r, n, ex->out_stride); }
        for (int i = 0; i < n; ++i) omap[(size_t)r * ex->out_stride + i] = tmp_g[i];
    }
    std::vector<int32_t> imap((size_t)nslots * ex->in_stride, -1), istat((size_t)nslots * ex->in_stride, -1);
    std::vector<int32_t> lmap((size_t)nslots * ex->lm_stride, -1), lstat((size_t)nslots * ex->lm_stride, -1);
    for (int s = 0; s < nslots; ++s) {
        const int r = region_of_slot[s];
        const sml::ResDesc &D = bank->res[s].desc;
        if (!D.loaded) { delete ex; return sml::fail(SML_ERR_STATE, "sml_exchange_create: slot %d has no reservoir", s); }
        int n = sml_domain_in_map(number_of_regions, r, overlap, 1, 1, 0, precip_bool, sst_input_of_slot ? sst_input_of_slot[s] : 1, 1,
                                  tmp_g.data(), tmp_s.data(), (int)tmp_g.size());
        if (n < 0) { delete ex; return n; }
        if (n != D.d) { delete ex; return sml::fail(SML_ERR_ARG, "sml_exchange_create: slot %d (region %d): input map has %d entries, reservoir d=%d", s, r, n, D.d); }
        for (int j = 0; j < n; ++j) { imap[(size_t)s * ex->in_stride + j] = tmp_g[j]; istat[(size_t)s * ex->in_stride + j] = tmp_s[j]; }
        n = sml_domain_out_map(number_of_regions, r, 1, 1, 0, precip_bool, tmp_g.data(), tmp_s.data(), (int)tmp_g.size());
        if (D.n_model > n) { delete ex; return sml::fail(SML_ERR_ARG, "sml_exchange_create: slot %d n_model=%d > outputs %d", s, D.n_model, n); }
        for (int i = 0; i < D.n_model; ++i) { lmap[(size_t)s * ex->lm_stride + i] = tmp_g[i]; lstat[(size_t)s * ex->lm_stride + i] = tmp_s[i]; }
    }
    int rc;
    if ((rc = sml::dev_upload(&ex->d_out_map, omap.data(), omap.size())) || (rc = sml::dev_upload(&ex->d_in_map, imap.data(), imap.size())) ||
        (rc = sml::dev_upload(&ex->d_in_stat, istat.data(), istat.size())) || (rc = sml::dev_upload(&ex->d_lm_map, lmap.data(), lmap.size())) ||
        (rc = sml::dev_upload(&ex->d_lm_stat, lstat.data(), lstat.size())) ||
        (rc = sml::dev_upload(&ex->d_region_of_slot, ex->region_of_slot.data(), ex->region_of_slot.size()))) {
        sml_exchange_destroy(ex);
        return rc;
    }
    *out = ex;
    return SML_OK;
}

int sml_exchange_destroy(sml_exchange *ex)
{
    if (!ex) return SML_OK;
    (void)hipFree(ex->d_out_map); (void)hipFree(ex->d_in_map); (void)hipFree(ex->d_in_stat);
    (void)hipFree(ex->d_lm_map); (void)hipFree(ex->d_lm_stat); (void)hipFree(ex->d_region_of_slot);
    delete ex;
    return SML_OK;
}

int sml_exchange_pack_outvec(sml_exchange *ex, double *all_outvec_dev, void *stream)
{
    SML_REQUIRE(ex && all_outvec_dev, "sml_exchange_pack_outvec: bad arguments");
    dim3 grid((ex->out_stride + 127) / 128, ex->nslots);
    hipLaunchKernelGGL(k_pack, grid, dim3(128), 0, sml::as_stream(stream), ex->bank->d_outvec, ex->out_stride, ex->d_region_of_slot,
                       ex->nslots, all_outvec_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_exchange_scatter(sml_exchange *ex, const double *all_outvec_dev, double *g_dev, const double *base_sst_dev,
                         const int32_t *sea_mask_dev, void *stream)
{
    SML_REQUIRE(ex && all_outvec_dev && g_dev, "sml_exchange_scatter: bad arguments");
    const int total = ex->number_of_regions * ex->out_stride;
    hipLaunchKernelGGL(k_scatter, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), all_outvec_dev, ex->out_stride,
                       ex->d_out_map, total, g_dev);
    SML_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sst, dim3(18), dim3(256), 0, sml::as_stream(stream), g_dev, base_sst_dev, sea_mask_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_exchange_gather(sml_exchange *ex, const double *g_dev, const double *f_dev, void *stream)
{
    SML_REQUIRE(ex && (g_dev || f_dev), "sml_exchange_gather: bad arguments");
    int rc = sml::bank_sync_descs(ex->bank);
    if (rc) return rc;
    hipStream_t st = sml::as_stream(stream);
    if (g_dev && f_dev) {       // the usual case: one launch for both
        GatherArgs a0{g_dev, ex->d_in_map, ex->d_in_stat, ex->bank->d_feedback, ex->in_stride};
        GatherArgs a1{f_dev, ex->d_lm_map, ex->d_lm_stat, ex->bank->d_local_model, ex->lm_stride};
        dim3 grid((std::max(ex->in_stride, ex->lm_stride) + 127) / 128, ex->nslots, 2);
        hipLaunchKernelGGL(k_gather2, grid, dim3(128), 0, st, ex->bank->d_descs, a0, a1, ex->nslots);
        SML_HIP(hipGetLastError());
        return SML_OK;
    }
    if (g_dev) {
        dim3 gin((ex->in_stride + 127) / 128, ex->nslots);
        hipLaunchKernelGGL(k_gather, gin, dim3(128), 0, st, ex->bank->d_descs, g_dev, ex->d_in_map, ex->d_in_stat, ex->in_stride, ex->nslots,
                           ex->bank->d_feedback, ex->in_stride);
        SML_HIP(hipGetLastError());
    }
    if (f_dev) {
        dim3 glm((ex->lm_stride + 127) / 128, ex->nslots);
        hipLaunchKernelGGL(k_gather, glm, dim3(128), 0, st, ex->bank->d_descs, f_dev, ex->d_lm_map, ex->d_lm_stat, ex->lm_stride, ex->nslots,
                           ex->bank->d_local_model, ex->lm_stride);
        SML_HIP(hipGetLastError());
    }
    return SML_OK;
}

}  // extern "C"

namespace sml {
// engine-internal (csrc/hybrid.hip): iogrid(31)'s copy-out, the next TISR slice and both gathers in one launch (k_egress)
int exchange_egress(sml_exchange *ex, const double *fields_out_dev, const double *tisr_slice_dev, double *g_dev, double *f_dev, hipStream_t st)
{
    SML_REQUIRE(ex && fields_out_dev && g_dev && f_dev, "exchange_egress: bad arguments");
    int rc = sml::bank_sync_descs(ex->bank);
    if (rc) return rc;
    GatherArgs a0{g_dev, ex->d_in_map, ex->d_in_stat, ex->bank->d_feedback, ex->in_stride};
    GatherArgs a1{f_dev, ex->d_lm_map, ex->d_lm_stat, ex->bank->d_local_model, ex->lm_stride};
    const int nblk_fields = ((NFIELD + 1) * NGP + 255) / 256;
    hipLaunchKernelGGL(k_egress, dim3(nblk_fields + ex->nslots), dim3(256), 0, st, ex->bank->d_descs, fields_out_dev, tisr_slice_dev, g_dev, f_dev, a0, a1,
                       nblk_fields, ex->nslots);
    SML_HIP(hipGetLastError());
    return SML_OK;
}
}  // namespace sml
