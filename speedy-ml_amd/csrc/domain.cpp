// resdomain bookkeeping as closed-form integer index maps (host only; no GPU needed).
//
// Replaces src/res_domain.f90: processor_decomposition(_manual) :31-94, initializedomain :96-121,
// getxyresextent :123-141, get_z_res_extent :143-153, getoverlapindices(_vert) :155-256,
// domaindecomposition :258-280, getworkerlower_leftcorner :282-292, get_trainingdataindices(_vert) :547-600,
// and turns the slice+reshape tilers (:294-545, :791-826, :1022-1125) into int32 gather/scatter maps into the
// device-resident global state buffer G (layout in include/speedyml_hip.h).  The reference recomputes the
// extents and allocates temporaries on every tiler call; here each region's maps are built once.
//
// Grid constants: xgrid=96, ygrid=48, zgrid=8 (src/mod_utilities.f90:17-20).
#include <cmath>
#include <vector>

#include "common.h"

namespace {

constexpr int XG = 96, YG = 48, ZG = 8, NV = 4;

struct Factors { int fx, fy; };

// Sub-domain shape: n = 4608/numregions grid points per region, fy = largest divisor of 48 that is
// <= floor(sqrt(n)) with n % fy == 0 and 96 % (n/fy) == 0  (domaindecomposition, :258-280).
bool factors(int numregions, Factors &f)
{
    if (numregions <= 0) return false;
    const int n = (XG * YG) / numregions;
    if (n <= 0) return false;
    const int fmax = (int)std::floor(std::sqrt((double)(float)n));
    for (int i = fmax; i >= 1; --i) {
        if (YG % i) continue;
        if (n % i) continue;
        const int fx = n / i;
        if (XG % fx) continue;
        f.fx = fx; f.fy = i;
        return true;
    }
    return false;
}

// horizontal geometry of one region
struct Patch {
    int rxs, rxe, rys, rye, rxc, ryc;          // res patch, 1-based inclusive
    int ixs, ixe, iys, iye, ixc, iyc;          // input patch (ixs > ixe when it wraps)
    bool pole, periodic, wraps;
    int tdxs, tdxe, tdys, tdye;
};

bool patch(int numregions, int region, int overlap, Patch &p)
{
    Factors f;
    if (!factors(numregions, f) || region < 0 || region >= numregions) return false;
    const int per_col = YG / f.fy;                    // regions stacked along latitude per longitude strip
    const int cornery = region % per_col;             // "col" in getworkerlower_leftcorner
    const int cornerx = region / per_col;             // floor(real(region)/(real(ygrid)/real(factory)))
    p.rxc = f.fx; p.ryc = f.fy;
    p.rxs = cornerx * f.fx + 1; p.rxe = (cornerx + 1) * f.fx;
    p.rys = cornery * f.fy + 1; p.rye = (cornery + 1) * f.fy;
    p.ixc = f.fx + 2 * overlap; p.iyc = f.fy + 2 * overlap;
    p.pole = p.periodic = false;
    if (p.rxs - overlap < 1) { p.ixs = XG - overlap + 1; p.periodic = true; } else p.ixs = p.rxs - overlap;
    if (p.rxe + overlap > XG) { p.ixe = overlap; p.periodic = true; } else p.ixe = p.rxe + overlap;
    if (p.rys - overlap < 1) { p.iys = 1; p.iyc = f.fy + overlap + (p.rys - 1); p.pole = true; } else p.iys = p.rys - overlap;
    if (p.rye + overlap > YG) { p.iye = YG; p.iyc = f.fy + overlap + (YG - p.rye); p.pole = true; } else p.iye = p.rye + overlap;
    p.wraps = p.periodic && (p.rxe > p.ixe || p.ixs > p.rxs);
    p.tdxs = 1 + overlap; p.tdxe = p.ixc - overlap;
    if (p.rys - overlap < 1) { p.tdys = 1 + (p.rys - 1); p.tdye = p.iyc - overlap; }
    else if (p.rye + overlap > YG) { p.tdys = 1 + overlap; p.tdye = p.iyc - (YG - p.rye); }
    else { p.tdys = 1 + overlap; p.tdye = p.iyc - overlap; }
    return true;
}

struct Vert { int rzs, rze, rzc, izs, ize, izc; bool top, bottom; int tdzs, tdze; };

bool vert(int nlev, int lev, int vo, Vert &v)
{
    if (nlev <= 0 || lev < 1 || lev > nlev) return false;
    v.rzc = ZG / nlev; v.rzs = (lev - 1) * v.rzc + 1; v.rze = lev * v.rzc;
    v.top = v.rzs == 1; v.bottom = v.rze == ZG;
    if (v.rzs - vo >= 1 && v.rze + vo <= ZG) { v.izs = v.rzs - vo; v.ize = v.rze + vo; v.izc = v.rzc + 2 * vo; }
    else if (v.rzs - vo < 1) { v.izs = 1; v.ize = v.rze + vo; v.izc = v.rzc + vo + (v.rzs - 1); }
    else { v.izs = v.rzs - vo; v.ize = ZG; v.izc = v.rzc + vo + (ZG - v.rze); }
    if (v.rzs - vo < 1) { v.tdzs = 1 + (v.rzs - 1); v.tdze = v.izc - vo; }
    else if (v.rze + vo > ZG) { v.tdzs = 1 + vo; v.tdze = v.izc - (ZG - v.rze); }
    else { v.tdzs = 1 + vo; v.tdze = v.izc - vo; }
    return true;
}

// global x (1-based) of local input column lx (1-based)
inline int input_gx(const Patch &p, int lx)
{
    if (!p.wraps) return p.ixs + lx - 1;
    const int nfirst = XG - (p.ixs - 1);
    return lx <= nfirst ? p.ixs + lx - 1 : lx - nfirst;
}

inline int g4_index(int v, int x, int y, int z) { return SML_G4_OFF + (((z - 1) * YG + (y - 1)) * XG + (x - 1)) * NV + (v - 1); }
inline int g2_index(int base, int x, int y) { return base + (y - 1) * XG + (x - 1); }

}  // namespace

extern "C" {

int sml_domain_decompose(int rank, int nranks, int number_of_regions, int32_t *region_indices, int capacity)
{
    SML_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks && number_of_regions > 0, "sml_domain_decompose: bad rank/nranks/regions");
    const int per = number_of_regions / nranks, left = number_of_regions % nranks;
    // ranks 1..left_over own one extra region taken from the tail (src/res_domain.f90:53-60)
    const bool extra = rank >= 1 && rank <= left;
    const int count = per + (extra ? 1 : 0);
    SML_REQUIRE(capacity >= count, "sml_domain_decompose: capacity %d < %d", capacity, count);
    for (int i = 0; i < per; ++i) region_indices[i] = per * rank + i;
    if (extra) region_indices[per] = number_of_regions - left + rank - 1;
    return count;
}

// the inverse of sml_domain_decompose: which rank owns `region`, and at which position of that rank's list
int sml_domain_region_owner(int nranks, int number_of_regions, int region, int *rank_out, int *slot_out)
{
    SML_REQUIRE(nranks > 0 && number_of_regions > 0 && region >= 0 && region < number_of_regions && rank_out && slot_out,
                "sml_domain_region_owner: bad arguments");
    const int per = number_of_regions / nranks, left = number_of_regions % nranks;
    if (region < per * nranks) { *rank_out = region / per; *slot_out = region % per; }
    else { *rank_out = region - (number_of_regions - left) + 1; *slot_out = per; }     // the tail: one each for ranks 1..left_over
    return SML_OK;
}

int sml_domain_region(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                      int vert_overlap, sml_region *out)
{
    Patch p; Vert v;
    SML_REQUIRE(out, "sml_domain_region: null out");
    SML_REQUIRE(patch(number_of_regions, region_num, overlap, p), "sml_domain_region: bad region %d of %d", region_num, number_of_regions);
    SML_REQUIRE(vert(num_vert_levels, vert_level, vert_overlap, v), "sml_domain_region: bad vertical level");
    out->res_xstart = p.rxs; out->res_xend = p.rxe; out->res_ystart = p.rys; out->res_yend = p.rye;
    out->resxchunk = p.rxc; out->resychunk = p.ryc;
    out->res_zstart = v.rzs; out->res_zend = v.rze; out->reszchunk = v.rzc;
    out->input_xstart = p.ixs; out->input_xend = p.ixe; out->input_ystart = p.iys; out->input_yend = p.iye;
    out->inputxchunk = p.ixc; out->inputychunk = p.iyc;
    out->input_zstart = v.izs; out->input_zend = v.ize; out->inputzchunk = v.izc;
    out->pole = p.pole; out->periodicboundary = p.periodic; out->top = v.top; out->bottom = v.bottom;
    out->tdata_xstart = p.tdxs; out->tdata_xend = p.tdxe; out->tdata_ystart = p.tdys; out->tdata_yend = p.tdye;
    out->tdata_zstart = v.tdzs; out->tdata_zend = v.tdze;
    return SML_OK;
}

int sml_domain_sizes(const sml_region *g, int m, int deg, int local_predictvars, int logp_bool, int precip_bool,
                     int sst_bool_input, int tisr_input_bool, int ml_only, sml_res_sizes *s)
{
    SML_REQUIRE(g && s && m > 0, "sml_domain_sizes: bad arguments");
    const int in2d = g->inputxchunk * g->inputychunk, res2d = g->resxchunk * g->resychunk;
    const int logp_in = logp_bool ? in2d : 0, sst_in = sst_bool_input ? in2d : 0, precip_in = precip_bool ? in2d : 0,
              tisr_in = tisr_input_bool ? in2d : 0;
    const int atmo_res = res2d * local_predictvars * g->reszchunk;
    s->chunk_size = atmo_res + (logp_bool ? res2d : 0) + (precip_bool ? res2d : 0);
    s->chunk_size_prediction = s->chunk_size;
    s->chunk_size_speedy = ml_only ? 0 : atmo_res + (logp_bool ? res2d : 0);
    const int atmo_in = in2d * g->inputzchunk * local_predictvars;
    s->locality = atmo_in + logp_in + precip_in + tisr_in + sst_in - s->chunk_size;
    s->reservoir_numinputs = s->chunk_size + s->locality;
    // NINT(m / numinputs): round half away from zero (operands are positive)
    s->nodes_per_input = (int)std::floor((double)m / (double)s->reservoir_numinputs + 0.5);
    s->n = s->nodes_per_input * s->reservoir_numinputs;
    // k = density*n*n with density = deg/real(m), truncated toward zero (quirk Q7, :172)
    s->k = (int)(((double)deg / (double)m) * (double)s->n * (double)s->n);
    s->atmo3d_start = 1; s->atmo3d_end = atmo_in;
    s->logp_start = s->logp_end = s->precip_start = s->precip_end = 0;
    s->sst_start = s->sst_end = s->tisr_start = s->tisr_end = 0;
    int cursor = atmo_in;
    if (logp_bool) { s->logp_start = cursor + 1; s->logp_end = cursor + logp_in; }
    cursor += logp_in;
    if (precip_bool) { s->precip_start = cursor + 1; s->precip_end = cursor + precip_in; }
    cursor += precip_in;
    if (sst_bool_input) { s->sst_start = cursor + 1; s->sst_end = cursor + sst_in; }
    cursor += sst_in;
    if (tisr_input_bool) { s->tisr_start = cursor + 1; s->tisr_end = cursor + tisr_in; }
    return SML_OK;
}

// mean/std slot convention (src/mod_reservoir.f90:1822-1848): (var-1)*inputzchunk + level for the 3-d variables,
// then logp, tisr, precip, sst in the order their flags are enabled.
static void stat_slots(int nz, int logp_bool, int tisr_bool, int precip_bool, int sst_bool, int &logp, int &tisr, int &precip, int &sst)
{
    int len = NV * nz;
    logp = tisr = precip = sst = -1;
    if (logp_bool) logp = len++;
    if (tisr_bool) tisr = len++;
    if (precip_bool) precip = len++;
    if (sst_bool) sst = len++;
}

int sml_domain_out_map(int number_of_regions, int region_num, int num_vert_levels, int vert_level, int vert_overlap,
                       int precip_bool, int32_t *g_index, int32_t *stat_idx, int capacity)
{
    Patch p; Vert v;
    SML_REQUIRE(patch(number_of_regions, region_num, 0, p), "sml_domain_out_map: bad region");
    SML_REQUIRE(vert(num_vert_levels, vert_level, vert_overlap, v), "sml_domain_out_map: bad level");
    const int res2d = p.rxc * p.ryc;
    const int count = NV * res2d * v.rzc + (v.bottom ? res2d * (1 + (precip_bool ? 1 : 0)) : 0);
    SML_REQUIRE(capacity >= count, "sml_domain_out_map: capacity %d < %d", capacity, count);
    int s_logp, s_tisr, s_precip, s_sst;
    // the bottom-level reservoir carries logp, tisr and (optionally) precip statistics; sst follows when present
    stat_slots(v.izc, v.bottom, 1, v.bottom && precip_bool, 0, s_logp, s_tisr, s_precip, s_sst);
    int q = 0;
    for (int z = v.rzs; z <= v.rze; ++z)
        for (int y = p.rys; y <= p.rye; ++y)
            for (int x = p.rxs; x <= p.rxe; ++x)
                for (int var = 1; var <= NV; ++var) {
                    g_index[q] = g4_index(var, x, y, z);
                    if (stat_idx) stat_idx[q] = (var - 1) * v.izc + (v.tdzs - 1) + (z - v.rzs);   // l of unstandardize_state_vec_res
                    ++q;
                }
    if (v.bottom) {
        for (int y = p.rys; y <= p.rye; ++y)
            for (int x = p.rxs; x <= p.rxe; ++x) { g_index[q] = g2_index(SML_G2_OFF, x, y); if (stat_idx) stat_idx[q] = s_logp; ++q; }
        if (precip_bool)
            for (int y = p.rys; y <= p.rye; ++y)
                for (int x = p.rxs; x <= p.rxe; ++x) { g_index[q] = g2_index(SML_GP_OFF, x, y); if (stat_idx) stat_idx[q] = s_precip; ++q; }
    }
    return q;
}

int sml_domain_in_map(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                      int vert_overlap, int precip_bool, int sst_bool_input, int tisr_input_bool,
                      int32_t *g_index, int32_t *stat_idx, int capacity)
{
    Patch p; Vert v;
    SML_REQUIRE(patch(number_of_regions, region_num, overlap, p), "sml_domain_in_map: bad region");
    SML_REQUIRE(vert(num_vert_levels, vert_level, vert_overlap, v), "sml_domain_in_map: bad level");
    const int in2d = p.ixc * p.iyc;
    const int n2d = v.bottom ? (1 + (precip_bool ? 1 : 0) + (sst_bool_input ? 1 : 0)) : 0;
    const int count = NV * in2d * v.izc + in2d * (n2d + (tisr_input_bool ? 1 : 0));
    SML_REQUIRE(capacity >= count, "sml_domain_in_map: capacity %d < %d", capacity, count);
    int s_logp, s_tisr, s_precip, s_sst;
    // slots exist whenever the reservoir was *trained* with the variable (sst_bool = slab_ocean_model_bool), even if
    // this region does not take SST as input; the bench/driver passes the trained layout through mean/std.
    stat_slots(v.izc, v.bottom, tisr_input_bool, v.bottom && precip_bool, v.bottom, s_logp, s_tisr, s_precip, s_sst);
    int q = 0;
    for (int z = v.izs; z <= v.ize; ++z)
        for (int y = p.iys; y <= p.iye; ++y)
            for (int lx = 1; lx <= p.ixc; ++lx) {
                const int x = input_gx(p, lx);
                for (int var = 1; var <= NV; ++var) {
                    g_index[q] = g4_index(var, x, y, z);
                    if (stat_idx) stat_idx[q] = (var - 1) * v.izc + (z - v.izs);
                    ++q;
                }
            }
    auto plane = [&](int base, int slot) {
        for (int y = p.iys; y <= p.iye; ++y)
            for (int lx = 1; lx <= p.ixc; ++lx) { g_index[q] = g2_index(base, input_gx(p, lx), y); if (stat_idx) stat_idx[q] = slot; ++q; }
    };
    if (v.bottom) {
        plane(SML_G2_OFF, s_logp);
        if (precip_bool) plane(SML_GP_OFF, s_precip);
        if (sst_bool_input) plane(SML_GS_OFF, s_sst);
    }
    if (tisr_input_bool) plane(SML_GT_OFF, s_tisr);
    return q;
}

// tile_full_input_to_target_data (src/res_domain.f90:602-689): which entries of the region's INPUT vector u(t) form its TARGET
// vector (the res patch inside the input patch: tdata_x/y/zstart..end), as 0-based positions: (var, x, y, z) var fastest, then
// logp(x, y), then precip(x, y) when predicted.  Training targets are trainingdata(in_pos(:), columns).
int sml_domain_target_map(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                          int vert_overlap, int precip_bool, int32_t *in_pos, int capacity)
{
    Patch p; Vert v;
    SML_REQUIRE(in_pos, "sml_domain_target_map: null output");
    SML_REQUIRE(patch(number_of_regions, region_num, overlap, p), "sml_domain_target_map: bad region");
    SML_REQUIRE(vert(num_vert_levels, vert_level, vert_overlap, v), "sml_domain_target_map: bad level");
    const int in2d = p.ixc * p.iyc, res2d = p.rxc * p.ryc;
    const int count = NV * res2d * v.rzc + (v.bottom ? res2d * (1 + (precip_bool ? 1 : 0)) : 0);
    SML_REQUIRE(capacity >= count, "sml_domain_target_map: capacity %d < %d", capacity, count);
    int q = 0;
    for (int z = v.tdzs; z <= v.tdze; ++z)
        for (int y = p.tdys; y <= p.tdye; ++y)
            for (int x = p.tdxs; x <= p.tdxe; ++x)
                for (int var = 1; var <= NV; ++var) in_pos[q++] = (var - 1) + NV * ((x - 1) + p.ixc * ((y - 1) + p.iyc * (z - 1)));
    if (v.bottom) {
        const int logp0 = NV * in2d * v.izc;                       // logp_start - 1 (src/mod_reservoir.f90:1859-1885)
        for (int plane = 0; plane < 1 + (precip_bool ? 1 : 0); ++plane)
            for (int y = p.tdys; y <= p.tdye; ++y)
                for (int x = p.tdxs; x <= p.tdxe; ++x) in_pos[q++] = logp0 + plane * in2d + (x - 1) + p.ixc * (y - 1);
    }
    return q;
}

// getsend_receive_size_{res,speedy,input,res_slab,input_slab} (src/mpires.f90:806-925): lengths of the per-region vectors
// the reference ships between ranks; here they size the outvec / local_model / feedback slabs.
int sml_domain_message_sizes(int number_of_regions, int region_num, int overlap, int num_vert_levels, int vert_level,
                             int vert_overlap, int precip_bool, int ohtc_bool_input, int32_t *sizes5)
{
    Patch p; Vert v;
    SML_REQUIRE(sizes5, "sml_domain_message_sizes: null output");
    SML_REQUIRE(patch(number_of_regions, region_num, overlap, p), "sml_domain_message_sizes: bad region");
    SML_REQUIRE(vert(num_vert_levels, vert_level, vert_overlap, v), "sml_domain_message_sizes: bad level");
    const int res2d = p.rxc * p.ryc, in2d = p.ixc * p.iyc;
    sizes5[0] = res2d * v.rzc * NV + (v.bottom ? res2d * (1 + (precip_bool ? 1 : 0)) : 0);     // _res    (:806-830)
    sizes5[1] = res2d * v.rzc * NV + (v.bottom ? res2d : 0);                                   // _speedy (:832-853)
    sizes5[2] = in2d * v.izc * NV + (v.bottom ? in2d * (1 + (precip_bool ? 1 : 0)) : 0);       // _input  (:877-902)
    sizes5[3] = res2d * (1 + (ohtc_bool_input ? 1 : 0));                                       // _res_slab (:855-875)
    sizes5[4] = in2d * (1 + (ohtc_bool_input ? 1 : 0));                                        // _input_slab (:904-925)
    return SML_OK;
}

// find_closest_divisor (src/mod_utilities.f90:1598-1636): batch size for the chunked training
// (initialize_chunk_training, src/mod_reservoir.f90:1561-1592: approx = (traininglength-discardlength)/(20*timestep))
int sml_find_closest_divisor(int target, int number)
{
    SML_REQUIRE(target > 0 && number > 0, "sml_find_closest_divisor: arguments must be positive");
    if (number % target == 0) return target;
    for (int radius = 2;; ++radius)
        for (int i = target - radius; i <= target + radius; ++i)
            if (i > 0 && number % i == 0) return i;
}

// The hybrid's own calendar (src/mod_calendar.f90), integer bookkeeping reproduced statement by statement, quirks included
// (8760-hour years when counting years elapsed, leap days subtracted from the day of the year, day-of-year 0 mapping to
// 31 December of the previous year).  date_out = (year, month, day, hour).
static bool leap(int y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }          // leap_year_check :93-105

int sml_calendar_date(int startyear, int hours_elapsed, int32_t *date_out)
{   // get_current_time_delta_hour (src/mod_calendar.f90:24-91)
    SML_REQUIRE(date_out && hours_elapsed >= 0, "sml_calendar_date: bad arguments");
    int ncal[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const int years = hours_elapsed / 8760;
    int year = years + startyear, leap_days = 0;
    for (int i = 0; i < years; ++i) leap_days += leap(startyear + i);
    const int day_of_year = (hours_elapsed % 8760) / 24 - leap_days;
    if (leap(year)) ncal[1] = 29;
    int counter = day_of_year, month = 1;
    while (counter > 0) {
        SML_REQUIRE(month <= 12, "sml_calendar_date: the reference's month loop runs off its table for hours_elapsed=%d", hours_elapsed);
        counter -= ncal[month - 1];
        ++month;
    }
    --month;
    if (month <= 0) { month = 12; --year; }
    date_out[0] = year; date_out[1] = month; date_out[2] = ncal[month - 1] + counter; date_out[3] = hours_elapsed % 24;
    return SML_OK;
}

int sml_hours_into_year(int year, int month, int day, int hour)
{   // numof_hours_into_year (src/mod_calendar.f90:133-175)
    static const int c365[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}, cleap[12] = {31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    SML_REQUIRE(month >= 1 && month <= 12, "sml_hours_into_year: month %d", month);
    int h = 0;
    for (int i = 0; i < month - 1; ++i) h += 24 * (leap(year) ? cleap[i] : c365[i]);
    if (day > 1) h += 24 * (day - 1);
    h += hour;
    return h == 0 ? 1 : h;
}

int sml_tisr_index(int startyear, int hours_elapsed)
{   // get_tisr_by_date (src/mpires.f90:1676-1708): 1-based slice of full_tisr(:,:,8760)
    int32_t d[4];
    int rc = sml_calendar_date(startyear, hours_elapsed, d);
    if (rc) return rc;
    int idx = sml_hours_into_year(d[0], d[1], d[2], d[3]);
    if (idx < 0) return idx;
    if (idx > 24 * 365) idx -= 24 * 365;
    return idx;
}

}  // extern "C"
