// Slab-ocean coupling of the region exchange, device-resident: the `ocean_model` branches of sendrecievegrid
// (src/mpires.f90:286-330 SST assembly, :470-484 mask + floor, :756-790 input averaging) and the sizes of
// initialize_slab_ocean_model (src/mod_slab_ocean_reservoir.f90:9-133).  The slab reservoirs themselves live in a second
// sml_bank and are stepped by sml_bank_predict_all (predict_slab_ml, :1318-1363: no physics-model rows, every output
// un-standardised with the SST statistics) every timestep_slab/timestep-th atmosphere step (src/parallelmain.f90:237-249).
//
// Per atmosphere step:
//   sml_slab_scatter_sst   wholegrid_sst: every region writes its 2x2 res patch -- the slab reservoir's first resx*resy outputs
//                          where the region predicts SST, 272 K elsewhere (mpires.f90:309-330); the mask/floor kernel of
//                          sml_exchange_scatter then restores base_sst where sea_mask > 0 and applies the 272 K floor.
//   sml_slab_update_inputs ring column mod(timestep-1, R) <- the lowest-level atmosphere inputs, logp, SST and TISR entries of
//                          the region's atmosphere feedback (atmo_training_data_idx, mod_slab_ocean_reservoir.f90:364-378);
//                          slab feedback <- sum over the R ring columns / R (R = timestep_slab/timestep - 1 = 27, :778-781).
// Where the reference is undefined the library takes the defined reading and says so: atmo_training_data_idx is allocated
// with 16 (one input patch) more entries than are ever assigned when OHTC is predicted (:326-329 vs :366-378); those trailing
// entries -- the OHTC input segment of the slab reservoir -- keep the value the exchange gives them (the standardised
// all-zero wholegrid_ohtc, mpires.f90:294-295,730-733) instead of being gathered through uninitialised indices.
#include <vector>

#include "bank.h"

struct sml_slab {
    sml_bank *atmo = nullptr, *slab = nullptr;
    int number_of_regions = 0, nslots = 0, ring = 0, stride = 0, res_cells = 4;
    int32_t *d_idx = nullptr;          // [nslots][stride]: 0-based position in the atmosphere feedback, -1 = not gathered
    int32_t *d_res_cell = nullptr;     // [number_of_regions][4]: cell (y*96+x) of each output of the res patch
    int32_t *d_sea_of_slot = nullptr;  // [nslots]
    double *d_ring = nullptr;          // [ring][nslots][stride]
};

namespace {

using sml::ResDesc;

__global__ void k_slab_sst(const double *__restrict__ all_slab_out, int out_stride, const int32_t *__restrict__ sea_of_region,
                           const int32_t *__restrict__ res_cell, int total, double *__restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int r = t >> 2, j = t & 3;
    const int cell = res_cell[t];
    if (cell < 0) return;
    g[SML_GS_OFF + cell] = sea_of_region[r] ? all_slab_out[(size_t)r * out_stride + j] : 272.0;
}

__global__ void k_slab_inputs(const double *__restrict__ atmo_fb, int atmo_stride, const int32_t *__restrict__ idx,
                              const int32_t *__restrict__ sea_of_slot, double *__restrict__ ring, int nring, int col, int nslots,
                              int stride, double *__restrict__ slab_fb)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (j >= stride || s >= nslots || !sea_of_slot[s]) return;
    const int p = idx[(size_t)s * stride + j];
    if (p < 0) return;
    const size_t plane = (size_t)nslots * stride, o = (size_t)s * stride + j;
    ring[(size_t)col * plane + o] = atmo_fb[(size_t)s * atmo_stride + p];
    double sum = 0.0;
    for (int c = 0; c < nring; ++c) sum = sum + ring[(size_t)c * plane + o];        // sum(..., dim=2)
    slab_fb[o] = sum / (double)nring;
}

// predict_slab's tail (src/mod_slab_ocean_reservoir.f90:1303-1309): the raw readout becomes the next call's local_model (the
// reservoir's own previous output is its "imperfect model"), then the outputs are un-standardised -- multiply, then add
__global__ void k_slab_hybrid_tail(const ResDesc *__restrict__ descs, int capacity, double *__restrict__ outvec, int out_stride,
                                   double *__restrict__ local_model, int lm_stride)
{
    const int s = blockIdx.x, i = threadIdx.x;
    if (s >= capacity) return;
    const ResDesc D = descs[s];
    if (!D.loaded || i >= D.n_out) return;
    const double raw = outvec[(size_t)s * out_stride + i];
    if (i < D.n_model) local_model[(size_t)s * lm_stride + i] = raw;
    const int st = D.out_stat[i];
    if (st >= 0) {
        const double scaled = __dmul_rn(raw, D.stdv[st]);
        outvec[(size_t)s * out_stride + i] = __dadd_rn(scaled, D.mean[st]);
    }
}

}  // namespace

extern "C" {

int sml_slab_sizes(const sml_region *g, int m, int deg, int local_predictvars, sml_res_sizes *out)
{   // initialize_slab_ocean_model (src/mod_slab_ocean_reservoir.f90:57-124) with its fixed flags: SST, TISR, OHTC and
    // atmosphere-to-ocean inputs on, precip input off, ml_only_ocean -> chunk_size_speedy = 0
    SML_REQUIRE(g && out && m > 0 && deg > 0, "sml_slab_sizes: bad arguments");
    const int in2d = g->inputxchunk * g->inputychunk, res2d = g->resxchunk * g->resychunk;
    const int atmo_size_input = in2d * local_predictvars + in2d;
    memset(out, 0, sizeof *out);
    out->chunk_size = res2d + res2d;                       // sst_size_res + ohtc_res_size
    out->chunk_size_prediction = out->chunk_size;
    out->chunk_size_speedy = 0;
    out->locality = atmo_size_input + in2d /*sst*/ + 0 /*sst climo*/ + in2d /*tisr*/ + in2d /*ohtc*/ - out->chunk_size;
    const double per = (double)m / ((double)out->chunk_size + (double)out->locality);
    out->nodes_per_input = (int)(per + 0.5);               // NINT of a positive number
    out->reservoir_numinputs = out->chunk_size + out->locality;
    out->n = out->nodes_per_input * out->reservoir_numinputs;
    // reservoir%k = reservoir%density*reservoir%n*reservoir%n with density = deg/m in double, truncated on assignment
    out->k = (int)(((double)deg / (double)m) * (double)out->n * (double)out->n);
    out->atmo3d_start = 1; out->atmo3d_end = in2d * local_predictvars;                       // :330-346 (get_training_data_from_atmo)
    out->logp_start = out->atmo3d_end + 1; out->logp_end = out->atmo3d_end + in2d;
    out->sst_start = out->logp_end + 1; out->sst_end = out->sst_start + in2d - 1;
    out->tisr_start = out->sst_end + 1; out->tisr_end = out->tisr_start + in2d - 1;
    return SML_OK;
}

int sml_slab_create(sml_bank *atmo_bank, sml_bank *slab_bank, int number_of_regions, const int32_t *region_of_slot, int nslots,
                    const int32_t *sea_of_slot, const int32_t *atmo_sst_input_of_slot, int ring, sml_slab **out)
{
    SML_REQUIRE(atmo_bank && slab_bank && region_of_slot && sea_of_slot && out && nslots > 0 && ring > 0 && nslots <= atmo_bank->capacity &&
                nslots <= slab_bank->capacity, "sml_slab_create: bad arguments");
    sml_slab *sl = new sml_slab;
    sl->atmo = atmo_bank; sl->slab = slab_bank; sl->number_of_regions = number_of_regions; sl->nslots = nslots; sl->ring = ring;
    sl->stride = slab_bank->max_d;
    std::vector<int32_t> idx((size_t)nslots * sl->stride, -1), cells((size_t)number_of_regions * 4, -1);
    std::vector<int32_t> tmp_g(8 * 96 * 48 * 8), tmp_s(8 * 96 * 48 * 8);
    int rc = SML_OK;
    for (int r = 0; r < number_of_regions && !rc; ++r) {
        // tile_full_2d_grid_with_local_res (src/res_domain.f90:828-850) addresses the same cells, in the same order, as the
        // logp segment of the region's output map
        sml_region g;
        if ((rc = sml_domain_region(number_of_regions, r, 1, 1, 1, 0, &g))) break;
        const int n = sml_domain_out_map(number_of_regions, r, 1, 1, 0, 0, tmp_g.data(), tmp_s.data(), (int)tmp_g.size());
        if (n < 0) { rc = n; break; }
        const int res2d = g.resxchunk * g.resychunk;
        if (res2d > 4) { rc = sml::fail(SML_ERR_ARG, "sml_slab_create: res patch of %d cells > 4", res2d); break; }
        for (int j = 0; j < res2d; ++j) cells[(size_t)r * 4 + j] = tmp_g[n - res2d + j] - SML_G2_OFF;
    }
    for (int s = 0; s < nslots && !rc; ++s) {
        if (!sea_of_slot[s]) continue;
        const ResDesc &A = atmo_bank->res[s].desc, &S = slab_bank->res[s].desc;
        if (!A.loaded || !S.loaded) { rc = sml::fail(SML_ERR_STATE, "sml_slab_create: slot %d needs an atmosphere and a slab reservoir", s); break; }
        sml_region g;
        sml_res_sizes a;
        if ((rc = sml_domain_region(number_of_regions, region_of_slot[s], 1, 1, 1, 0, &g))) break;
        const int sst_in = atmo_sst_input_of_slot ? atmo_sst_input_of_slot[s] : 1;
        if ((rc = sml_domain_sizes(&g, 6000, 6, 4, 1, 1, sst_in, 1, 0, &a))) break;
        if (!sst_in) { rc = sml::fail(SML_ERR_ARG, "sml_slab_create: slot %d predicts SST but its atmosphere reservoir has no SST input", s); break; }
        if (a.reservoir_numinputs != A.d) { rc = sml::fail(SML_ERR_ARG, "sml_slab_create: slot %d: atmosphere d=%d, expected %d", s, A.d, a.reservoir_numinputs); break; }
        const int in2d = g.inputxchunk * g.inputychunk;
        // atmo_training_data_idx (src/mod_slab_ocean_reservoir.f90:364-378), 1-based -> 0-based
        int c = 0;
        int32_t *row = &idx[(size_t)s * sl->stride];
        for (int i = a.atmo3d_end - in2d * 4 + 1; i <= a.logp_end; ++i) row[c++] = i - 1;
        for (int i = a.sst_start; i <= a.sst_end; ++i) row[c++] = i - 1;
        for (int i = a.tisr_start; i <= a.tisr_end; ++i) row[c++] = i - 1;
        if (c + in2d != S.d) { rc = sml::fail(SML_ERR_ARG, "sml_slab_create: slot %d: slab d=%d, expected %d", s, S.d, c + in2d); break; }
    }
    if (!rc) rc = sml::dev_upload(&sl->d_idx, idx.data(), idx.size());
    if (!rc) rc = sml::dev_upload(&sl->d_res_cell, cells.data(), cells.size());
    if (!rc) rc = sml::dev_upload(&sl->d_sea_of_slot, sea_of_slot, (size_t)nslots);
    if (!rc) rc = sml::dev_zeros(&sl->d_ring, (size_t)ring * nslots * sl->stride);
    if (rc) { sml_slab_destroy(sl); return rc; }
    *out = sl;
    return SML_OK;
}

int sml_slab_destroy(sml_slab *sl)
{
    if (!sl) return SML_OK;
    (void)hipFree(sl->d_idx); (void)hipFree(sl->d_res_cell); (void)hipFree(sl->d_sea_of_slot); (void)hipFree(sl->d_ring);
    delete sl;
    return SML_OK;
}

int sml_slab_scatter_sst(sml_slab *sl, const double *all_slab_out_dev, int out_stride, const int32_t *sea_of_region_dev, double *g_dev,
                         void *stream)
{
    SML_REQUIRE(sl && all_slab_out_dev && sea_of_region_dev && g_dev && out_stride >= 4, "sml_slab_scatter_sst: bad arguments");
    const int total = sl->number_of_regions * 4;
    hipLaunchKernelGGL(k_slab_sst, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), all_slab_out_dev, out_stride,
                       sea_of_region_dev, sl->d_res_cell, total, g_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_slab_update_inputs(sml_slab *sl, int timestep, void *stream)
{
    SML_REQUIRE(sl && timestep >= 1, "sml_slab_update_inputs: timestep is 1-based");
    const int col = (timestep - 1) % sl->ring;            // mod(timestep-1, R) + 1, 0-based here
    dim3 grid((sl->stride + 127) / 128, sl->nslots);
    hipLaunchKernelGGL(k_slab_inputs, grid, dim3(128), 0, sml::as_stream(stream), sl->atmo->d_feedback, sl->atmo->max_d, sl->d_idx,
                       sl->d_sea_of_slot, sl->d_ring, sl->ring, col, sl->nslots, sl->stride, sl->slab->d_feedback);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

/* predict_slab (src/mod_slab_ocean_reservoir.f90:1268-1316), the hybrid slab ocean: the bank's slots were loaded with n_model = n_out
 * (the physics-model rows are the reservoir's own previous, standardised output).  Advances and reads out every loaded slot, feeds
 * the raw readout back as the next local_model and leaves the un-standardised output in the bank's outvec buffer. */
int sml_slab_predict_hybrid(sml_bank *slab_bank, void *stream)
{
    SML_REQUIRE(slab_bank, "sml_slab_predict_hybrid: null bank");
    int rc = sml_bank_predict_all(slab_bank, 1, stream);           // flags bit0: raw readout
    if (rc) return rc;
    SML_REQUIRE(slab_bank->max_n_out <= 1024, "sml_slab_predict_hybrid: too many outputs");
    hipLaunchKernelGGL(k_slab_hybrid_tail, dim3(slab_bank->capacity), dim3(((slab_bank->max_n_out + 63) / 64) * 64), 0, sml::as_stream(stream),
                       (const ResDesc *)slab_bank->d_descs, slab_bank->capacity, slab_bank->d_outvec, slab_bank->max_n_out, slab_bank->d_local_model,
                       slab_bank->max_n_model);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

}  // extern "C"
