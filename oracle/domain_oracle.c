/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the `resdomain` index bookkeeping and tilers
 * (src/res_domain.f90) and of the integer sizing in allocate_res_new / trained_reservoir_prediction
 * (src/mod_reservoir.f90:80-180, 1783-1886).  See sml_oracle.h for scope.
 *
 * PARITY: pinned only by (a) the reference's own known answer tests/mod_unit_test.f90:63-96
 * (getxyresextent(288,145): x 49-52, chunk 4x4; its y expectation 9-12 contradicts the shipped code,
 * which yields 5-8 -- SURVEY.md section 4) and (b) the reference-run facts recorded in SURVEY.md
 * Appendix A (tests/golden/survey_appendix_a.json).  res_domain.f90 itself cannot be compiled here
 * without stand-ins for the MKL_SPBLAS and mpi Fortran modules that mod_utilities.f90:3,5 imports.
 *
 * The tilers below operate on real arrays exactly like the Fortran ones (slice + reshape), so the
 * product's precomputed int32 gather/scatter maps are checked against an independent formulation.
 * Storage: grid4d(4,96,48,8) -> [((z*48+y)*96+x)*4+v], grid2d(96,48) -> [y*96+x]  (0-based x,y,z,v).
 */
#include "sml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G4(g, v, x, y, z) (g)[((((z) - 1) * RD_YGRID + ((y) - 1)) * RD_XGRID + ((x) - 1)) * 4 + ((v) - 1)]
#define G2(g, x, y) (g)[((y) - 1) * RD_XGRID + ((x) - 1)]

/* src/res_domain.f90:64-94 (identical logic to processor_decomposition :31-62) */
int rd_processor_decomposition(int proc, int numprocs, int number_of_regions, int *region_indices)
{
    int per = number_of_regions / numprocs;
    int left_over = number_of_regions % numprocs;
    int i;
    if (proc >= left_over + 1 && proc > 0) {
        for (i = 1; i <= per; ++i) region_indices[i - 1] = per * proc + i - 1;
        return per;
    } else if (proc == 0) {
        for (i = 1; i <= per; ++i) region_indices[i - 1] = i - 1;
        return per;
    } else {
        for (i = 1; i <= per; ++i) region_indices[i - 1] = per * proc + i - 1;
        /* Fortran DO leaves i == per+1 */
        region_indices[i - 1] = number_of_regions - left_over + proc - 1;
        return per + 1;
    }
}

/* src/res_domain.f90:258-280 */
void rd_domaindecomposition(int numregions, int *factorx, int *factory)
{
    int n = RD_GRIDNUM / numregions;
    int factorMax = (int)floor(sqrt((double)(float)n));
    *factorx = 0; *factory = 0;
    for (int i = factorMax; i >= 1; --i) {        /* i=0 would be MOD(ygrid,0); never reached for valid inputs */
        if (RD_YGRID % i == 0) {
            *factory = i;
            if (n % *factory == 0) {
                *factorx = n / *factory;
                if (RD_XGRID % *factorx == 0) break;
            }
        }
    }
}

/* src/res_domain.f90:282-292 */
void rd_getworkerlower_leftcorner(int region_num, int factory, int *row, int *col)
{
    *col = region_num % (RD_YGRID / factory);
    /* floor(real(region_num)/(real(ygrid)/real(factory))) in promoted (double) reals */
    *row = (int)floor((double)region_num / ((double)RD_YGRID / (double)factory));
}

/* src/res_domain.f90:123-141 */
void rd_getxyresextent(int num_regions, int region_num, int *xs, int *xe, int *ys, int *ye, int *xchunk, int *ychunk)
{
    int cornerx, cornery;
    rd_domaindecomposition(num_regions, xchunk, ychunk);
    rd_getworkerlower_leftcorner(region_num, *ychunk, &cornerx, &cornery);
    *xs = cornerx * *xchunk + 1;
    *xe = (cornerx + 1) * *xchunk;
    *ys = cornery * *ychunk + 1;
    *ye = (cornery + 1) * *ychunk;
}

/* src/res_domain.f90:143-153 */
static void get_z_res_extent(int num_vert_levels, int vert_level, int *zs, int *ze, int *zchunk)
{
    *zchunk = RD_ZGRID / num_vert_levels;
    *zs = (vert_level - 1) * *zchunk + 1;
    *ze = vert_level * *zchunk;
}

/* src/res_domain.f90:155-203 */
static void getoverlapindices(int numregions, int region_num, int overlap, int *ixs, int *ixe, int *iys, int *iye,
                              int *ixchunk, int *iychunk, int *pole, int *periodic)
{
    int rxs, rxe, rys, rye, xchunk, ychunk;
    rd_getxyresextent(numregions, region_num, &rxs, &rxe, &rys, &rye, &xchunk, &ychunk);
    *ixchunk = xchunk + 2 * overlap;
    *iychunk = ychunk + 2 * overlap;
    *periodic = 0; *pole = 0;
    if (rxs - overlap < 1) { *ixs = RD_XGRID - overlap + 1; *periodic = 1; }
    else *ixs = rxs - overlap;
    if (rxe + overlap > RD_XGRID) { *ixe = overlap; *periodic = 1; }
    else *ixe = overlap + rxe;
    if (rys - overlap < 1) { *iys = 1; *iychunk = ychunk + overlap + (rys - 1); *pole = 1; }
    else *iys = rys - overlap;
    if (rye + overlap > RD_YGRID) { *iye = RD_YGRID; *iychunk = ychunk + overlap + (RD_YGRID - rye); *pole = 1; }
    else *iye = overlap + rye;
}

/* src/res_domain.f90:205-256 */
static void getoverlapindices_vert(int num_vert_levels, int vert_level, int vert_overlap, int *izs, int *ize, int *izchunk,
                                   int *top, int *bottom)
{
    int zs, ze, zchunk;
    get_z_res_extent(num_vert_levels, vert_level, &zs, &ze, &zchunk);
    *top = (zs == 1);
    *bottom = (ze == RD_ZGRID);
    if (zs - vert_overlap >= 1 && ze + vert_overlap <= RD_ZGRID) {
        *izs = zs - vert_overlap; *ize = ze + vert_overlap; *izchunk = zchunk + 2 * vert_overlap;
    } else if (zs - vert_overlap < 1) {
        *izs = 1; *ize = ze + vert_overlap; *izchunk = zchunk + vert_overlap + (zs - 1);
    } else {
        *izs = zs - vert_overlap; *ize = RD_ZGRID; *izchunk = zchunk + vert_overlap + (RD_ZGRID - ze);
    }
}

/* src/res_domain.f90:547-575 */
static void get_trainingdataindices(int num_regions, int region_num, int overlap, int *xs, int *xe, int *ys, int *ye)
{
    int rxs, rxe, rys, rye, rxc, ryc, ixs, ixe, iys, iye, ixc, iyc, pole, per;
    rd_getxyresextent(num_regions, region_num, &rxs, &rxe, &rys, &rye, &rxc, &ryc);
    getoverlapindices(num_regions, region_num, overlap, &ixs, &ixe, &iys, &iye, &ixc, &iyc, &pole, &per);
    *xs = 1 + overlap;
    *xe = ixc - overlap;
    if (rys - overlap < 1) { *ys = 1 + (rys - 1); *ye = iyc - overlap; }
    else if (rye + overlap > RD_YGRID) { *ys = 1 + overlap; *ye = iyc - (RD_YGRID - rye); }
    else { *ys = 1 + overlap; *ye = iyc - overlap; }
}

/* src/res_domain.f90:577-600 */
static void get_trainingdataindices_vert(int num_vert_levels, int vert_level, int vert_overlap, int *zs, int *ze)
{
    int rzs, rze, rzc, izs, ize, izc, top, bottom;
    get_z_res_extent(num_vert_levels, vert_level, &rzs, &rze, &rzc);
    getoverlapindices_vert(num_vert_levels, vert_level, vert_overlap, &izs, &ize, &izc, &top, &bottom);
    if (rzs - vert_overlap < 1) { *zs = 1 + (rzs - 1); *ze = izc - vert_overlap; }
    else if (rze + vert_overlap > RD_ZGRID) { *zs = 1 + vert_overlap; *ze = izc - (RD_ZGRID - rze); }
    else { *zs = 1 + vert_overlap; *ze = izc - vert_overlap; }
}

/* src/res_domain.f90:96-121 */
void rd_initializedomain(int num_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                         int vert_overlap, rd_grid *g)
{
    memset(g, 0, sizeof *g);
    rd_getxyresextent(num_regions, region_num, &g->res_xstart, &g->res_xend, &g->res_ystart, &g->res_yend, &g->resxchunk, &g->resychunk);
    get_z_res_extent(num_vert_levels, vert_level, &g->res_zstart, &g->res_zend, &g->reszchunk);
    getoverlapindices(num_regions, region_num, overlap, &g->input_xstart, &g->input_xend, &g->input_ystart, &g->input_yend,
                      &g->inputxchunk, &g->inputychunk, &g->pole, &g->periodicboundary);
    getoverlapindices_vert(num_vert_levels, vert_level, vert_overlap, &g->input_zstart, &g->input_zend, &g->inputzchunk, &g->top, &g->bottom);
    get_trainingdataindices(num_regions, region_num, overlap, &g->tdata_xstart, &g->tdata_xend, &g->tdata_ystart, &g->tdata_yend);
    get_trainingdataindices_vert(num_vert_levels, vert_level, vert_overlap, &g->tdata_zstart, &g->tdata_zend);
    g->overlap = overlap; g->num_vert_levels = num_vert_levels; g->vert_overlap = vert_overlap; g->number_of_regions = num_regions;
}

/* src/mod_reservoir.f90:80-180 (sizes) and :1851-1885 (segment offsets) */
void rd_allocate_sizes(const rd_grid *g, int m, int deg, int local_predictvars, int logp_bool, int precip_bool,
                       int sst_bool_input, int tisr_input_bool, int ml_only, rd_sizes *s)
{
    memset(s, 0, sizeof *s);
    double density = (double)deg / (double)m;                   /* :103 reservoir%deg/real(m) */
    int in2d = g->inputxchunk * g->inputychunk, res2d = g->resxchunk * g->resychunk;
    s->logp_size_input = logp_bool ? in2d : 0;
    s->sst_size_input = sst_bool_input ? in2d : 0;
    s->logp_size_res = logp_bool ? res2d : 0;
    s->precip_size_res = precip_bool ? res2d : 0;
    s->precip_size_input = precip_bool ? in2d : 0;
    s->tisr_size_input = tisr_input_bool ? in2d : 0;
    s->chunk_size = res2d * local_predictvars * g->reszchunk + s->logp_size_res + s->precip_size_res;
    s->chunk_size_prediction = s->chunk_size;
    s->chunk_size_speedy = res2d * local_predictvars * g->reszchunk + s->logp_size_res;
    if (ml_only) s->chunk_size_speedy = 0;
    s->locality = in2d * g->inputzchunk * local_predictvars + s->logp_size_input + s->precip_size_input +
                  s->tisr_size_input + s->sst_size_input - s->chunk_size;
    /* NINT(dble(m)/(dble(chunk)+dble(locality))) : round half away from zero */
    double q = (double)m / ((double)s->chunk_size + (double)s->locality);
    s->nodes_per_input = (int)(q >= 0 ? floor(q + 0.5) : -floor(-q + 0.5));
    s->n = s->nodes_per_input * (s->chunk_size + s->locality);
    s->k = (int)(density * (double)s->n * (double)s->n);       /* :172 truncation toward zero */
    s->reservoir_numinputs = s->chunk_size + s->locality;

    s->atmo3d_start = 1;
    s->atmo3d_end = local_predictvars * in2d * g->inputzchunk;
    if (logp_bool) { s->logp_start = s->atmo3d_end + 1; s->logp_end = s->atmo3d_end + s->logp_size_input; }
    if (precip_bool) { s->precip_start = s->atmo3d_end + s->logp_size_input + 1; s->precip_end = s->precip_start + s->precip_size_input - 1; }
    if (sst_bool_input) { s->sst_start = s->atmo3d_end + s->logp_size_input + s->precip_size_input + 1; s->sst_end = s->sst_start + s->sst_size_input - 1; }
    if (tisr_input_bool) { s->tisr_start = s->atmo3d_end + s->logp_size_input + s->precip_size_input + s->sst_size_input + 1; s->tisr_end = s->tisr_start + s->tisr_size_input - 1; }
}

/* tileoverlapgrid4d (src/res_domain.f90:348-420): localgrid(4, xchunk, ychunk, zchunk), Fortran order, returned flat.
 * All four branches reduce to: copy x-range [ixs..xgrid] then [1..ixe] when the region wraps, else [ixs..ixe]. */
static void tileoverlapgrid4d(const double *grid, int numregions, int region_num, int overlap, int num_vert_levels,
                              int vert_level, int vert_overlap, double *local, int *xc, int *yc, int *zc)
{
    int rxs, rxe, rys, rye, rxc, ryc, ixs, ixe, iys, iye, ixc, iyc, pole, per, izs, ize, izc, top, bottom;
    rd_getxyresextent(numregions, region_num, &rxs, &rxe, &rys, &rye, &rxc, &ryc);
    getoverlapindices(numregions, region_num, overlap, &ixs, &ixe, &iys, &iye, &ixc, &iyc, &pole, &per);
    getoverlapindices_vert(num_vert_levels, vert_level, vert_overlap, &izs, &ize, &izc, &top, &bottom);
    *xc = ixc; *yc = iyc; *zc = izc;
    int wrap = per && (rxe > ixe || ixs > rxs);
    int nfirst = wrap ? RD_XGRID - (ixs - 1) : ixc;
    for (int z = izs; z <= ize; ++z)
        for (int y = iys; y <= iye; ++y)
            for (int lx = 1; lx <= ixc; ++lx) {
                int gx = (lx <= nfirst) ? ixs + lx - 1 : lx - nfirst;
                for (int v = 1; v <= 4; ++v)
                    local[((((z - izs) * iyc + (y - iys)) * ixc) + (lx - 1)) * 4 + (v - 1)] = G4(grid, v, gx, y, z);
            }
}

/* tileoverlapgrid2d (src/res_domain.f90:484-545) */
static void tileoverlapgrid2d(const double *grid, int numregions, int region_num, int overlap, double *local, int *xc, int *yc)
{
    int rxs, rxe, rys, rye, rxc, ryc, ixs, ixe, iys, iye, ixc, iyc, pole, per;
    rd_getxyresextent(numregions, region_num, &rxs, &rxe, &rys, &rye, &rxc, &ryc);
    getoverlapindices(numregions, region_num, overlap, &ixs, &ixe, &iys, &iye, &ixc, &iyc, &pole, &per);
    *xc = ixc; *yc = iyc;
    int wrap = per && (rxe > ixe || ixs > rxs);
    int nfirst = wrap ? RD_XGRID - (ixs - 1) : ixc;
    for (int y = iys; y <= iye; ++y)
        for (int lx = 1; lx <= ixc; ++lx) {
            int gx = (lx <= nfirst) ? ixs + lx - 1 : lx - nfirst;
            local[(y - iys) * ixc + (lx - 1)] = G2(grid, gx, y);
        }
}

void rd_tile_input2d(int num_regions, int region_num, int overlap, const double *grid2d, double *out)
{
    int xc, yc;
    tileoverlapgrid2d(grid2d, num_regions, region_num, overlap, out, &xc, &yc);
}

/* tile_4d_and_logp_to_local_state_input (src/res_domain.f90:1081-1125) */
void rd_tile_input(int num_regions, int region_num, int overlap, int num_vert_levels, int vert_level, int vert_overlap,
                   int precip_bool, const double *grid4d, const double *grid2d, const double *precip, double *inputvec)
{
    int zs, ze, zchunk, x, y, z;
    get_z_res_extent(num_vert_levels, vert_level, &zs, &ze, &zchunk);
    tileoverlapgrid4d(grid4d, num_regions, region_num, overlap, num_vert_levels, vert_level, vert_overlap, inputvec, &x, &y, &z);
    if (ze == RD_ZGRID) {
        int x2, y2;
        tileoverlapgrid2d(grid2d, num_regions, region_num, overlap, inputvec + 4 * x * y * z, &x2, &y2);
        if (precip_bool) tileoverlapgrid2d(precip, num_regions, region_num, overlap, inputvec + 4 * x * y * z + x * y, &x2, &y2);
    }
}

/* tile_full_grid_with_local_state_vec_res1d (src/res_domain.f90:791-826) */
void rd_scatter_res(int num_regions, int num_vert_levels, int region_num, int vert_level, int precip_bool, int length,
                    const double *statevec, double *grid4d, double *grid2d, double *precip)
{
    int xs, xe, ys, ye, xc, yc, zs, ze, zc;
    rd_getxyresextent(num_regions, region_num, &xs, &xe, &ys, &ye, &xc, &yc);
    get_z_res_extent(num_vert_levels, vert_level, &zs, &ze, &zc);
    int p = 0;
    for (int z = zs; z <= ze; ++z)
        for (int y = ys; y <= ye; ++y)
            for (int x = xs; x <= xe; ++x)
                for (int v = 1; v <= 4; ++v) G4(grid4d, v, x, y, z) = statevec[p++];
    if (ze == RD_ZGRID) {
        for (int y = ys; y <= ye; ++y)
            for (int x = xs; x <= xe; ++x) G2(grid2d, x, y) = statevec[p++];
        if (precip_bool) {
            /* statevec(atmo3d_length + xc*yc + 1 : length) */
            for (int y = ys; y <= ye; ++y)
                for (int x = xs; x <= xe; ++x) G2(precip, x, y) = statevec[p++];
            (void)length;
        }
    } else {
        for (int y = ys; y <= ye; ++y)
            for (int x = xs; x <= xe; ++x) G2(grid2d, x, y) = 0;
    }
}

/* tile_4d_and_logp_full_grid_to_local_res_vec (src/res_domain.f90:1022-1053) */
void rd_tile_res(int num_regions, int num_vert_levels, int region_num, int vert_level,
                 const double *grid4d, const double *grid2d, double *statevec)
{
    int xs, xe, ys, ye, xc, yc, zs, ze, zc;
    rd_getxyresextent(num_regions, region_num, &xs, &xe, &ys, &ye, &xc, &yc);
    get_z_res_extent(num_vert_levels, vert_level, &zs, &ze, &zc);
    int p = 0;
    for (int z = zs; z <= ze; ++z)
        for (int y = ys; y <= ye; ++y)
            for (int x = xs; x <= xe; ++x)
                for (int v = 1; v <= 4; ++v) statevec[p++] = G4(grid4d, v, x, y, z);
    if (ze == RD_ZGRID)
        for (int y = ys; y <= ye; ++y)
            for (int x = xs; x <= xe; ++x) statevec[p++] = G2(grid2d, x, y);
}

/* standardize_data_given_pars*: subtract, then divide (src/mod_utilities.f90:1283-1329) */
/* tile_full_input_to_target_data2d (src/res_domain.f90:602-689): statevec(reservoir_numinputs, T) column-major ->
 * tiled(chunk_size_prediction, T).  The reshape/slice/reshape of the Fortran is spelt out element by element in column-major
 * order: temp5d(var, x, y, z, t) = statevec(var + nv*((x-1) + nxi*((y-1) + nyi*(z-1))), t). */
void rd_tile_target(const rd_grid *g, const rd_sizes *s, int local_predictvars, int logp_bool, int precip_bool,
                    const double *statevec, int ld_in, int T, double *tiled, int ld_out)
{
    const int nv = local_predictvars, nxi = g->inputxchunk, nyi = g->inputychunk;
    for (int t = 0; t < T; ++t) {
        const double *u = statevec + (size_t)t * ld_in;
        double *o = tiled + (size_t)t * ld_out;
        int q = 0;
        for (int z = g->tdata_zstart; z <= g->tdata_zend; ++z)
            for (int y = g->tdata_ystart; y <= g->tdata_yend; ++y)
                for (int x = g->tdata_xstart; x <= g->tdata_xend; ++x)
                    for (int v = 1; v <= nv; ++v) o[q++] = u[(v - 1) + nv * ((x - 1) + nxi * ((y - 1) + nyi * (z - 1)))];
        if (logp_bool)              /* :624-627: temp3d = reshape(statevec(logp_start:logp_end,:)) */
            for (int y = g->tdata_ystart; y <= g->tdata_yend; ++y)
                for (int x = g->tdata_xstart; x <= g->tdata_xend; ++x) o[q++] = u[(s->logp_start - 1) + (x - 1) + nxi * (y - 1)];
        if (precip_bool)            /* :641-647 */
            for (int y = g->tdata_ystart; y <= g->tdata_yend; ++y)
                for (int x = g->tdata_xstart; x <= g->tdata_xend; ++x) o[q++] = u[(s->precip_start - 1) + (x - 1) + nxi * (y - 1)];
    }
}

static inline double stdz(double v, double mean, double std) { double t = v - mean; return t / std; }
/* unstandardize_data_*: multiply, then add as two statements (src/mod_utilities.f90:667-831) */
static inline double unstdz(double v, double mean, double std) { double t = v * std; return t + mean; }

/* standardize_state_vec_input (src/res_domain.f90:1211-1268): l runs var-major, level-minor; then logp */
void rd_standardize_input(const rd_grid *g, const rd_sizes *s, int local_predictvars, int logp_bool,
                          const double *mean, const double *std, double *state_vec)
{
    int xc = g->inputxchunk, yc = g->inputychunk, zc = g->inputzchunk;
    int l = 0;
    for (int i = 0; i < local_predictvars; ++i)
        for (int j = 0; j < zc; ++j, ++l)
            for (int y = 0; y < yc; ++y)
                for (int x = 0; x < xc; ++x) {
                    int idx = (s->atmo3d_start - 1) + (((j * yc + y) * xc) + x) * local_predictvars + i;
                    state_vec[idx] = stdz(state_vec[idx], mean[l], std[l]);
                }
    if (logp_bool)
        for (int p = s->logp_start - 1; p < s->logp_end; ++p) state_vec[p] = stdz(state_vec[p], mean[l], std[l]);
}

/* standardize_state_vec_res (src/res_domain.f90:1270-1315) */
void rd_standardize_res(const rd_grid *g, int local_predictvars, int heightlevels_input, int logp_bool,
                        const double *mean, const double *std, double *state_vec)
{
    int xc = g->resxchunk, yc = g->resychunk, zc = g->reszchunk;
    int l = 0;
    for (int i = 0; i < local_predictvars; ++i) {
        int data_height = 0;
        for (int j = 1; j <= heightlevels_input; ++j, ++l) {
            if (j >= g->tdata_zstart && j <= g->tdata_zend) {
                for (int y = 0; y < yc; ++y)
                    for (int x = 0; x < xc; ++x) {
                        int idx = (((data_height * yc + y) * xc) + x) * local_predictvars + i;
                        state_vec[idx] = stdz(state_vec[idx], mean[l], std[l]);
                    }
                ++data_height;
            }
        }
    }
    if (logp_bool) {
        int base = local_predictvars * xc * yc * zc;
        for (int p = 0; p < xc * yc; ++p) state_vec[base + p] = stdz(state_vec[base + p], mean[l], std[l]);
    }
}

/* unstandardize_state_vec_res (src/res_domain.f90:1424-1475); logp_idx/precip_idx are 1-based mean/std slots */
void rd_unstandardize_res(const rd_grid *g, int local_predictvars, int heightlevels_input, int logp_bool, int precip_bool,
                          int logp_idx, int precip_idx, const double *mean, const double *std, double *state_vec)
{
    int xc = g->resxchunk, yc = g->resychunk, zc = g->reszchunk;
    int l = 0;
    for (int i = 0; i < local_predictvars; ++i) {
        int data_height = 0;
        for (int j = 1; j <= heightlevels_input; ++j, ++l) {
            if (j >= g->tdata_zstart && j <= g->tdata_zend) {
                for (int y = 0; y < yc; ++y)
                    for (int x = 0; x < xc; ++x) {
                        int idx = (((data_height * yc + y) * xc) + x) * local_predictvars + i;
                        state_vec[idx] = unstdz(state_vec[idx], mean[l], std[l]);
                    }
                ++data_height;
            }
        }
    }
    int base = local_predictvars * xc * yc * zc;
    if (logp_bool)
        for (int p = 0; p < xc * yc; ++p) state_vec[base + p] = unstdz(state_vec[base + p], mean[logp_idx - 1], std[logp_idx - 1]);
    if (precip_bool)
        for (int p = 0; p < xc * yc; ++p)
            state_vec[base + xc * yc + p] = unstdz(state_vec[base + xc * yc + p], mean[precip_idx - 1], std[precip_idx - 1]);
}

/* src/res_domain.f90:1630-1660 (quirk Q5: a constant, not a ramp, equatorward of 45 deg) */
double rd_get_radius_by_lat(double startlat, double endlat)
{
    const double highest_lat = 45.0, max_radius = 0.7, min_radius = 0.3;
    double smallest_lat = fabs(startlat < endlat ? startlat : endlat);
    if (smallest_lat >= highest_lat) return max_radius;
    return (max_radius - min_radius) / highest_lat + min_radius;
}

/* ---- the hybrid's calendar (src/mod_calendar.f90) ---- */
static int rd_leap(int y) { return ((y % 4 == 0) && (y % 100 != 0)) || (y % 400 == 0); }      /* leap_year_check :93-105 */

/* get_current_time_delta_hour :24-91 -> date[4] = year, month, day, hour */
void rd_calendar_date(int startyear, int hours_elapsed, int *date)
{
    int ncal365[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    int years_elasped = hours_elapsed / 8760;
    int currentyear = years_elasped + startyear;
    int leap_days = 0;
    for (int i = 0; i <= years_elasped - 1; ++i) if (rd_leap(startyear + i)) leap_days = leap_days + 1;
    int day_of_year = ((hours_elapsed % 8760) / 24) - leap_days;
    if (rd_leap(currentyear)) ncal365[1] = 29;
    int day_while_counter = day_of_year, month = 1;
    while (day_while_counter > 0) {
        day_while_counter = day_while_counter - ncal365[month - 1];
        month = month + 1;
    }
    month = month - 1;
    if (month <= 0) { month = 12; currentyear = currentyear - 1; }
    date[0] = currentyear; date[1] = month; date[2] = ncal365[month - 1] + day_while_counter; date[3] = hours_elapsed % 24;
}

/* numof_hours_into_year :133-175 */
int rd_hours_into_year(int year, int month, int day, int hour)
{
    static const int ncal365[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    static const int ncal_leap[12] = {31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    int numofhours = 0;
    if (month > 1) for (int i = 1; i <= month - 1; ++i) numofhours += 24 * (rd_leap(year) ? ncal_leap[i - 1] : ncal365[i - 1]);
    if (day > 1) for (int i = 1; i <= day - 1; ++i) numofhours += 24;
    numofhours = numofhours + hour;
    if (numofhours == 0) numofhours = 1;
    return numofhours;
}

/* get_tisr_by_date, src/mpires.f90:1695-1704 */
int rd_tisr_index(int startyear, int hours_elapsed)
{
    int d[4];
    rd_calendar_date(startyear, hours_elapsed, d);
    int idx = rd_hours_into_year(d[0], d[1], d[2], d[3]);
    if (idx > 24 * 365) idx = idx - 24 * 365;
    return idx;
}

/* ---- slab-ocean coupling (src/mpires.f90:286-330, 470-484, 776-781) ---- */
/* wholegrid_sst (96 x 48, [y][x] as G stores it): base, then every region's res patch (slab outvec or 272), then the mask
 * rule and the 272 K floor.  res_cell[r*4+j] = y*96+x of output j of region r (tile_full_2d_grid_with_local_res). */
void rd_slab_sst(int nreg, const double *base_sst, const int *sea_mask_gt0, const int *sea_of_region, const int *res_cell,
                 const double *all_slab_out, int out_stride, double *sst)
{
    for (int c = 0; c < RD_GRIDNUM; ++c) sst[c] = base_sst[c];
    for (int r = 0; r < nreg; ++r)
        for (int j = 0; j < 4; ++j) {
            int cell = res_cell[r * 4 + j];
            if (cell < 0) continue;
            sst[cell] = sea_of_region[r] ? all_slab_out[(size_t)r * out_stride + j] : 272.0;
        }
    for (int c = 0; c < RD_GRIDNUM; ++c) {
        if (sea_mask_gt0[c]) sst[c] = base_sst[c];
        if (sst[c] < 272.0) sst[c] = 272.0;
    }
}

/* averaged_atmo_input_vec(:, mod(timestep-1, R)+1) = feedback_atmo(atmo_training_data_idx); feedback_slab = sum(ring, dim=2)/R.
 * idx: nidx 0-based positions; ring: [R][nidx]; slab feedback entries [0, nidx) are overwritten, the rest is left alone. */
void rd_slab_ring_update(int timestep, int R, int nidx, const int *idx, const double *feedback_atmo, double *ring, double *feedback_slab)
{
    int col = (timestep - 1) % R;
    for (int j = 0; j < nidx; ++j) ring[(size_t)col * nidx + j] = feedback_atmo[idx[j]];
    for (int j = 0; j < nidx; ++j) {
        double s = 0.0;
        for (int c = 0; c < R; ++c) s = s + ring[(size_t)c * nidx + j];
        feedback_slab[j] = s / (double)R;
    }
}
