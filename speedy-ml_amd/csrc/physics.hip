// SPEEDY's column physics on gfx950: the parametrisations phypar drives for every grid column each time step.
//
// Replaces src/phy_phypar.f90:80-230 (the grid-point part, sections 1.2-4.2) and the routines it calls:
//   shtorh  src/phy_shtorh.f90        saturation humidity / relative humidity
//   convmf  src/phy_convmf.f90        simplified mass-flux convection
//   lscond  src/phy_lscond.f90        large-scale condensation
//   cloud, radsw, radlw, radset, sol_oz, solar   src/phy_radiat.f90   clouds, short- and long-wave radiation
//   suflux, sflset  src/phy_suflux.f90  surface fluxes and skin temperature
//   vdifsc  src/phy_vdifsc.f90        vertical diffusion and shallow convection
//   inphys  src/ini_inphys.f90        sigma-level constants
// Not here (and not in the reference's hybrid step either: nstrdf = 0, sppt_on = .false., src/mod_tsteps.f90:68,72): the random
// diabatic forcing and SPPT.  The daily coupler (land / sea-ice models, dmflux) stays with the host: the surface state is an
// input (sml_phys_set_surface).
//
// The work is independent per column: one thread per column (4608 threads) runs the whole sequence with the column's 8 levels
// in registers (run-time level indices are predicates in fixed-bound loops) and the long-lived accumulators parked in LDS; per-level
// arrays are indexed 1..8 as in the Fortran so that the statements can be read side by side.  The device code is in
// physics_dev.h: this file holds the stand-alone kernel and the C-ABI, dynamics.hip fuses the same column function behind
// grtend's grid-point tendencies (k_gridtend_physics).  What the reference keeps in module variables between calls -- the long-wave transmissivities, the
// stratospheric correction, the short-wave heating and surface flux of the last short-wave step (every nstrad-th) -- lives in
// the handle's device arrays.  Operation order is the reference's (no FMA contraction); exp/sqrt come from the device libm
// (<= 1-2 ulp from the host's).
#include <cmath>
#include <vector>

#include "common.h"
#include "physics_dev.h"

using namespace smlphys;

namespace {

__global__ __launch_bounds__(64) void k_physics(PhysLev L, PhysDev D, PhysIn in, double *__restrict__ tend, int lradsw,
                                                 int off_u, int off_v, int off_t, int off_q, int accumulate, int want_diag)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= GR) return;
    __shared__ double park[PARK_DOUBLES];
    LA tt{park + threadIdx.x}, qt{park + NLP * 64 + threadIdx.x};
    for (int k = 1; k <= KX; ++k) {
        tt[k] = accumulate ? tend[(size_t)(off_t + k - 1) * GR + p] : 0.0;
        qt[k] = accumulate ? tend[(size_t)(off_q + k - 1) * GR + p] : 0.0;
    }
    const double u_dyn = accumulate ? tend[(size_t)(off_u + KX - 1) * GR + p] : 0.0;
    const double v_dyn = accumulate ? tend[(size_t)(off_v + KX - 1) * GR + p] : 0.0;
    if (!accumulate)
        for (int k = 1; k < KX; ++k) { tend[(size_t)(off_u + k - 1) * GR + p] = 0.0; tend[(size_t)(off_v + k - 1) * GR + p] = 0.0; }
    physics_column(L, D, in, tend, lradsw, off_u, off_v, off_t, off_q, want_diag, p, park, u_dyn, v_dyn);
}

// fordate(0)'s grid-point work (src/ini_fordate.f90:54-61,72-109), which the hybrid repeats at the start of every window through
// agcm_init: the surface albedos from snow depth and sea-ice fraction, and the two fields whose spectra correct the horizontal
// diffusion of temperature and humidity over orography -- corh_t = gamlat phis0 and corh_q = refrh1 (q_sat(tref, p = 1) -
// q_sat(tsfc, psfc)) with tsfc the land / sea mix of stl_am and sst_am (in the hybrid: the ML-predicted SST).  One thread per
// grid point, statements in the reference's order; psfc_dummy = 1.0 is the reference's own stand-in for the reference pressure.
__global__ __launch_bounds__(256) void k_fordate(const double *__restrict__ fmask_l, const double *__restrict__ fmask_s, const double *__restrict__ phis0,
                                                  const double *__restrict__ stl_am, const double *__restrict__ sst_am, const double *__restrict__ alb0,
                                                  const double *__restrict__ snowd_am, const double *__restrict__ sice_am, double *__restrict__ alb_l,
                                                  double *__restrict__ alb_s, double *__restrict__ albsfc, double *__restrict__ snowc,
                                                  double *__restrict__ corh /* [2][GR]: temperature, humidity */)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= GR) return;
    const FordatePoint a{fmask_l, fmask_s, phis0, stl_am, alb0, snowd_am, sice_am, alb_l, alb_s, albsfc, snowc};
    double ct, cq;
    fordate_point(a, p, sst_am[p], ct, cq);
    corh[p] = ct;
    corh[GR + p] = cq;
}

struct ZonalArgs { double z[6 * IL]; };
__global__ void k_set_zonal(ZonalArgs a, double *__restrict__ zonal) { zonal[threadIdx.x] = a.z[threadIdx.x]; }

}  // namespace

extern "C" {

int sml_phys_destroy(sml_phys *ph)
{
    if (!ph) return SML_OK;
    for (void *p : ph->allocs) (void)hipFree(p);
    delete ph;
    return SML_OK;
}

int sml_phys_create(const double *hsg9, const double *rlat48, sml_phys **out)
{
    SML_REQUIRE(hsg9 && rlat48 && out, "sml_phys_create: bad arguments");
    sml_phys *ph = new sml_phys;
    build_levels(ph->lev, hsg9);
    for (int j = 0; j < IL; ++j) { ph->slat[j] = std::sin(rlat48[j]); ph->clat[j] = std::cos(rlat48[j]); }
    std::vector<double> fband((size_t)301 * 4);
    build_fband(fband.data());
    int rc = SML_OK;
    auto zeros = [&](double **p, size_t n) { int r = sml::dev_zeros(p, n); if (!r) ph->allocs.push_back(*p); return r; };
    double *fb = nullptr;
    rc = sml::dev_upload(&fb, fband.data(), fband.size());
    if (!rc) { ph->allocs.push_back(fb); ph->dev.fband = fb; }
    if (!rc) rc = zeros(&ph->surf, (size_t)10 * GR);
    if (!rc) rc = zeros(&ph->zonal, (size_t)6 * IL);
    if (!rc) rc = zeros(&ph->dev.tau2, (size_t)4 * KX * GR);
    if (!rc) rc = zeros(&ph->dev.stratc, (size_t)2 * GR);
    if (!rc) rc = zeros(&ph->dev.tt_rsw, (size_t)KX * GR);
    if (!rc) rc = zeros(&ph->dev.ssrd, (size_t)GR);
    if (!rc) rc = zeros(&ph->dev.diag, (size_t)NDIAG * GR);
    if (rc) { sml_phys_destroy(ph); return rc; }
    PhysDev &d = ph->dev;
    d.fmask = ph->surf; d.phis0 = ph->surf + GR; d.tland = ph->surf + 2 * GR; d.tsea = ph->surf + 3 * GR; d.swav = ph->surf + 4 * GR;
    d.alb_l = ph->surf + 5 * GR; d.alb_s = ph->surf + 6 * GR; d.albsfc = ph->surf + 7 * GR; d.snowc = ph->surf + 8 * GR; d.forog = ph->surf + 9 * GR;
    d.fsol = ph->zonal; d.ozone = ph->zonal + IL; d.ozupp = ph->zonal + 2 * IL; d.zenit = ph->zonal + 3 * IL; d.stratz = ph->zonal + 4 * IL;
    d.sqclat = ph->zonal + 5 * IL;
    *out = ph;
    return SML_OK;
}

int sml_phys_set_surface(sml_phys *ph, const double *fmask, const double *phis0, const double *tland, const double *tsea, const double *swav,
                         const double *alb_l, const double *alb_s, const double *albsfc, const double *snowc)
{   // host arrays [48][96]; forog = sflset(phis0) (src/phy_suflux.f90:355-382)
    SML_REQUIRE(ph && fmask && phis0 && tland && tsea && swav && alb_l && alb_s && albsfc && snowc, "sml_phys_set_surface: bad arguments");
    std::vector<double> h((size_t)10 * GR);
    const double *src[9] = {fmask, phis0, tland, tsea, swav, alb_l, alb_s, albsfc, snowc};
    for (int i = 0; i < 9; ++i) memcpy(&h[(size_t)i * GR], src[i], GR * sizeof(double));
    const double rhdrag = 1. / (GG * HDRAG);
    for (int p = 0; p < GR; ++p) h[(size_t)9 * GR + p] = 1. + FHDRAG * (1. - std::exp(-std::max(phis0[p], 0.) * rhdrag));
    SML_HIP(hipMemcpy(ph->surf, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_phys_update_surface(sml_phys *ph, const double *tland, const double *swav, const double *snowd_am, const double *sice_am)
{   // the coupler's daily output (src/cpl_land.f90 / cpl_sea.f90 -> stl_am, soilw_am, snowd_am, sice_am), host arrays [48][96], each optional
    SML_REQUIRE(ph, "sml_phys_update_surface: null handle");
    if (tland) SML_HIP(hipMemcpy(ph->surf + 2 * (size_t)GR, tland, GR * sizeof(double), hipMemcpyHostToDevice));
    if (swav) SML_HIP(hipMemcpy(ph->surf + 4 * (size_t)GR, swav, GR * sizeof(double), hipMemcpyHostToDevice));
    if (snowd_am || sice_am) {
        SML_REQUIRE(ph->fordate && ph->fordate_albedo, "sml_phys_update_surface: snow depth / sea ice only enter through fordate's albedos (sml_phys_set_fordate_fields with alb0)");
        if (snowd_am) SML_HIP(hipMemcpy(ph->fordate + 2 * (size_t)GR, snowd_am, GR * sizeof(double), hipMemcpyHostToDevice));
        if (sice_am) SML_HIP(hipMemcpy(ph->fordate + 3 * (size_t)GR, sice_am, GR * sizeof(double), hipMemcpyHostToDevice));
    }
    return SML_OK;
}

int sml_phys_set_sst_dev(sml_phys *ph, const double *tsea_dev, void *stream)
{   // the hybrid model supplies the sea-surface temperature every step (G's SST segment)
    SML_REQUIRE(ph && tsea_dev, "sml_phys_set_sst_dev: bad arguments");
    SML_HIP(hipMemcpyAsync(ph->surf + 3 * GR, tsea_dev, GR * sizeof(double), hipMemcpyDeviceToDevice, sml::as_stream(stream)));
    return SML_OK;
}

int sml_phys_bind_sst_dev(sml_phys *ph, const double *tsea_dev)
{   // sst_am read in place: the kernel takes the sea temperature from the caller's device array (NULL: back to the handle's copy)
    SML_REQUIRE(ph, "sml_phys_bind_sst_dev: null handle");
    ph->dev.tsea = tsea_dev ? tsea_dev : ph->surf + 3 * (size_t)smlphys::GR;
    return SML_OK;
}

int sml_phys_set_fordate_fields(sml_phys *ph, const double *fmask_s, const double *alb0, const double *snowd_am, const double *sice_am)
{   // host arrays [48][96]; alb0 / snowd_am / sice_am together or not at all
    SML_REQUIRE(ph && fmask_s, "sml_phys_set_fordate_fields: bad arguments");
    const bool alb = alb0 || snowd_am || sice_am;
    SML_REQUIRE(!alb || (alb0 && snowd_am && sice_am), "sml_phys_set_fordate_fields: alb0, snowd_am and sice_am come together");
    if (!ph->fordate) {
        int rc = sml::dev_zeros(&ph->fordate, (size_t)6 * GR);          // fmask_s alb0 snowd_am sice_am | corh(2)
        if (rc) return rc;
        ph->allocs.push_back(ph->fordate);
    }
    SML_HIP(hipMemcpy(ph->fordate, fmask_s, GR * sizeof(double), hipMemcpyHostToDevice));
    if (alb) {
        SML_HIP(hipMemcpy(ph->fordate + GR, alb0, GR * sizeof(double), hipMemcpyHostToDevice));
        SML_HIP(hipMemcpy(ph->fordate + 2 * GR, snowd_am, GR * sizeof(double), hipMemcpyHostToDevice));
        SML_HIP(hipMemcpy(ph->fordate + 3 * GR, sice_am, GR * sizeof(double), hipMemcpyHostToDevice));
    }
    ph->fordate_albedo = alb;
    return SML_OK;
}

int sml_phys_fordate(sml_phys *ph, sml_spectral *sp, double *corh_spec_dev, void *stream)
{
    SML_REQUIRE(ph && sp && corh_spec_dev, "sml_phys_fordate: bad arguments");
    SML_REQUIRE(ph->fordate, "sml_phys_fordate: sml_phys_set_fordate_fields has not been called");
    double *f = ph->fordate, *surf = ph->surf;
    hipLaunchKernelGGL(k_fordate, dim3((GR + 255) / 256), dim3(256), 0, sml::as_stream(stream), ph->dev.fmask, (const double *)f, ph->dev.phis0,
                       ph->dev.tland, ph->dev.tsea, ph->fordate_albedo ? (const double *)(f + GR) : (const double *)nullptr, (const double *)(f + 2 * GR),
                       (const double *)(f + 3 * GR), surf + 5 * GR, surf + 6 * GR, surf + 7 * GR, surf + 8 * GR, f + 4 * GR);
    SML_HIP(hipGetLastError());
    return sml_spectral_spec(sp, f + 4 * GR, corh_spec_dev, 2, stream);      // spec(corh, tcorh), spec(corh, qcorh): no trunct (:86,113)
}

int sml_phys_get_surface(sml_phys *ph, int which, double *out_host)
{   // 0 fmask 1 phis0 2 tland 3 tsea (the handle's copy) 4 swav 5 alb_l 6 alb_s 7 albsfc 8 snowc 9 forog | 10 corh_t 11 corh_q (fordate's grids)
    SML_REQUIRE(ph && out_host && which >= 0 && which < 12, "sml_phys_get_surface: bad arguments");
    SML_REQUIRE(which < 10 || ph->fordate, "sml_phys_get_surface: fordate has not been set up");
    const double *src = which < 10 ? ph->surf + (size_t)which * GR : ph->fordate + (size_t)(which - 6) * GR;
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpy(out_host, src, sizeof(double) * GR, hipMemcpyDeviceToHost));
    return SML_OK;
}

int sml_phys_sol_oz(sml_phys *ph, double tyear)
{
    SML_REQUIRE(ph, "sml_phys_sol_oz: null handle");
    double z[6 * IL];
    sol_oz(tyear, ph->clat, ph->slat, z, z + IL, z + 2 * IL, z + 3 * IL, z + 4 * IL);
    for (int j = 0; j < IL; ++j) z[5 * IL + j] = std::sqrt(ph->clat[j]);
    SML_HIP(hipMemcpy(ph->zonal, z, sizeof z, hipMemcpyHostToDevice));
    return SML_OK;
}

/* the same in stream order: the 288 values travel as kernel arguments (no host synchronisation: a blocking copy once per model day
 * drained the hybrid engine's queue every fourth step) */
int sml_phys_sol_oz_async(sml_phys *ph, double tyear, void *stream)
{
    SML_REQUIRE(ph, "sml_phys_sol_oz_async: null handle");
    ZonalArgs a;
    sol_oz(tyear, ph->clat, ph->slat, a.z, a.z + IL, a.z + 2 * IL, a.z + 3 * IL, a.z + 4 * IL);
    for (int j = 0; j < IL; ++j) a.z[5 * IL + j] = std::sqrt(ph->clat[j]);
    hipLaunchKernelGGL(k_set_zonal, dim3(1), dim3(6 * IL), 0, sml::as_stream(stream), a, ph->zonal);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_phys_get_tables(sml_phys *ph, double *zonal_host /* [6][48] */, double *fband_host /* [301][4] */, double *levels_host /* [9][9] */)
{
    SML_REQUIRE(ph, "sml_phys_get_tables: null handle");
    if (zonal_host) SML_HIP(hipMemcpy(zonal_host, ph->zonal, sizeof(double) * 6 * IL, hipMemcpyDeviceToHost));
    if (fband_host) SML_HIP(hipMemcpy(fband_host, ph->dev.fband, sizeof(double) * 301 * 4, hipMemcpyDeviceToHost));
    if (levels_host) memcpy(levels_host, &ph->lev, sizeof(PhysLev));
    return SML_OK;
}

static int launch_physics(sml_phys *ph, const PhysIn &in, int lradsw, double *tend_dev, int off_u, int off_v, int off_t, int off_q,
                          int accumulate, int want_diag, void *stream)
{
    SML_REQUIRE(ph && in.t && tend_dev && off_u >= 0 && off_v >= 0 && off_t >= 0 && off_q >= 0, "sml_phys_tendencies: bad arguments");
    hipLaunchKernelGGL(k_physics, dim3(GR / 64), dim3(64), 0, sml::as_stream(stream), ph->lev, ph->dev, in, tend_dev, lradsw ? 1 : 0,
                       off_u, off_v, off_t, off_q, accumulate ? 1 : 0, want_diag ? 1 : 0);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_phys_tendencies(sml_phys *ph, const double *g, int lradsw, double *tend_dev, int off_u, int off_v, int off_t, int off_q,
                        int accumulate, void *stream)
{   // phypar's full set [41][GR]: ug1(8) vg1(8) tg1(8) qg1(8) phig1(8) pslg1
    SML_REQUIRE(g, "sml_phys_tendencies: bad arguments");
    PhysIn in{g + (size_t)7 * GR, g + (size_t)15 * GR, g + (size_t)16 * GR, g + (size_t)24 * GR, g + (size_t)32 * GR, g + (size_t)40 * GR};
    return launch_physics(ph, in, lradsw, tend_dev, off_u, off_v, off_t, off_q, accumulate, 1, stream);
}

int sml_phys_tendencies_sfcwind(sml_phys *ph, const double *g, int lradsw, double *tend_dev, int off_u, int off_v, int off_t, int off_q,
                                int accumulate, int want_diag, void *stream)
{   // the set the parametrisations actually read, [27][GR]: ug1(:,kx) vg1(:,kx) tg1(8) qg1(8) phig1(8) pslg1
    SML_REQUIRE(g, "sml_phys_tendencies_sfcwind: bad arguments");
    PhysIn in{g, g + (size_t)GR, g + (size_t)2 * GR, g + (size_t)10 * GR, g + (size_t)18 * GR, g + (size_t)26 * GR};
    return launch_physics(ph, in, lradsw, tend_dev, off_u, off_v, off_t, off_q, accumulate, want_diag, stream);
}

int sml_phys_debug_stamps(unsigned long long *out)      // not part of the C-ABI (no declaration in include/): phase profiling aid
{
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phys_dbg), sizeof(unsigned long long) * 16));
    return SML_OK;
}

int sml_phys_diag(sml_phys *ph, int which, double *out_host)
{
    SML_REQUIRE(ph && out_host && which >= 0 && which < NDIAG, "sml_phys_diag: bad arguments");
    SML_HIP(hipMemcpy(out_host, ph->dev.diag + (size_t)which * GR, sizeof(double) * GR, hipMemcpyDeviceToHost));
    return SML_OK;
}

}  // extern "C"
