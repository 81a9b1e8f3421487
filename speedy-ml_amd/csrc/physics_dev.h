// Device side of SPEEDY's column physics (see physics.hip for the overview): constants, level tables, the parametrisations as
// device functions and physics_column, the whole grid-point sequence of phypar for one column.  Included by physics.hip (the
// stand-alone kernel and the C-ABI) and by dynamics.hip (the kernel that fuses grtend's grid-point tendencies with the physics).
#pragma once
#include <cmath>
#include <vector>

#include "common.h"

namespace smlphys {
namespace {


constexpr int IX = 96, IL = 48, KX = 8, GR = IX * IL, NLP = KX + 1;

// src/mod_physcon.f90:14-34
constexpr double P0 = 1.e+5, GG = 9.81, RD = 287., CP = 1004., ALHC = 2501.0, SBC = 5.67e-8;
// src/mod_dyncon0.f90:10,19 (reference lapse rate [K/km], reference relative humidity of the diffusion correction), src/mod_surfcon.f90:34,
// src/mod_radcon.f90:59-61 -- fordate's constants
constexpr double GAMMA = 6.0, REFRH1 = 0.7, SD2SC = 60.0, ALBSEA = 0.07, ALBICE = 0.60, ALBSN = 0.60;
// src/mod_cnvcon.f90
constexpr double PSMIN = 0.8, TRCNV = 6.0, RHBL = 0.9, RHIL = 0.7, ENTMAX = 0.5, SMF = 0.8;
// src/mod_lsccon.f90
constexpr double TRLSC = 4.0, RHLSC = 0.9, DRHLSC = 0.1, RHBLSC = 0.95;
// src/mod_vdicon.f90
constexpr double TRSHC = 6.0, TRVDI = 24.0, TRVDS = 6.0, REDSHC = 0.5, RHGRAD = 0.5, SEGRAD = 0.1;
// src/mod_sflcon.f90
constexpr double FWIND0 = 0.95, FTEMP0 = 1.0, FHUM0 = 0.0, CDL = 2.4e-3, CDS = 1.0e-3, CHL = 1.2e-3, CHS = 0.9e-3, VGUST = 5.0,
                 CTDAY = 1.0e-2, DTHETA = 3.0, FSTAB = 0.67, HDRAG = 2000.0, FHDRAG = 0.5, CLAMBDA = 7.0, CLAMBSN = 7.0;
// src/mod_radcon.f90:62-100
constexpr double SOLC = 342.0, RHCL1 = 0.30, RHCL2 = 1.00, QACL = 0.20, WPCL = 0.2, PMAXCL = 10.0, CLSMAX = 0.60, CLSMINL = 0.15,
                 GSE_S0 = 0.25, GSE_S1 = 0.40, ALBCL = 0.43, ALBCLS = 0.50, EPSSW = 0.020, EPSLW = 0.05, EMISFC = 0.98,
                 ABSDRY = 0.033, ABSAER = 0.033, ABSWV1 = 0.022, ABSWV2 = 15.000, ABSCL1 = 0.015, ABSCL2 = 0.15, ABLWIN = 0.3,
                 ABLCO2 = 6.0, ABLWV1 = 0.7, ABLWV2 = 50.0, ABLCL1 = 12.0, ABLCL2 = 0.6;

struct PhysLev {                 // inphys (src/ini_inphys.f90), indices 1..8 (0 unused except sigh)
    double sig[KX + 1], sigl[KX + 1], dsig[KX + 1], sigh[KX + 1], grdsig[KX + 1], grdscp[KX + 1], wvi2[KX + 1], wvi1[KX + 1];
    double entr[KX + 1];         // convmf's entrainment profile (:95-103), a function of sigma only
};

struct PhysDev {                 // device arrays, [GR] unless noted
    const double *fmask, *phis0, *tland, *tsea, *swav, *alb_l, *alb_s, *albsfc, *snowc, *forog;
    const double *fsol, *ozone, *ozupp, *zenit, *stratz, *sqclat;     // [48] per latitude (sol_oz; sqrt(clat) for suflux 2.1)
    const double *fband;         // [301][4] energy fractions of the LW bands for T = 100..400 K (radset)
    double *tau2;                // [4][8][GR] long-wave transmissivities of the last short-wave step
    double *stratc;              // [2][GR]
    double *tt_rsw;              // [8][GR] short-wave heating of the last short-wave step (already a temperature tendency)
    double *ssrd;                // [GR] downward short-wave flux at the surface of the last short-wave step
    double *diag;                // [NDIAG][GR]
};
enum { D_PRECNV = 0, D_PRECLS, D_CBMF, D_TS, D_TSKIN, D_SSRD, D_SLRD, D_OLR, D_SHF, D_EVAP, D_USTR, D_VSTR, D_CLOUDC, D_CLSTR, D_TSR,
       D_SSR, D_SLR, D_HFLUXN1, D_HFLUXN2, D_T0, D_Q0, D_IPTOP, D_ICLTOP, NDIAG };

// ---------------------------------------------------------------- host: tables
void build_levels(PhysLev &L, const double *hsg)
{   // inphys :26-47
    L.sigh[0] = hsg[0];
    for (int k = 1; k <= KX; ++k) {
        L.sig[k] = 0.5 * (hsg[k] + hsg[k - 1]);
        L.sigl[k] = std::log(L.sig[k]);
        L.sigh[k] = hsg[k];
        L.dsig[k] = hsg[k] - hsg[k - 1];
        L.grdsig[k] = GG / (L.dsig[k] * P0);
        L.grdscp[k] = L.grdsig[k] / CP;
    }
    for (int k = 1; k <= KX - 1; ++k) {
        L.wvi1[k] = 1. / (L.sigl[k + 1] - L.sigl[k]);
        L.wvi2[k] = (std::log(L.sigh[k]) - L.sigl[k]) * L.wvi1[k];
    }
    L.wvi1[KX] = 0.;
    L.wvi2[KX] = (std::log(0.99) - L.sigl[KX]) * L.wvi1[KX - 1];
    // convmf :95-103
    double sentr = 0.;
    for (int k = 1; k <= KX; ++k) L.entr[k] = 0.;
    for (int k = 2; k <= KX - 1; ++k) {
        const double e = std::max(0., L.sig[k] - 0.5);
        L.entr[k] = e * e;
        sentr = sentr + L.entr[k];
    }
    sentr = ENTMAX / sentr;
    for (int k = 2; k <= KX - 1; ++k) L.entr[k] = L.entr[k] * sentr;
    L.sig[0] = L.sigl[0] = L.dsig[0] = L.grdsig[0] = L.grdscp[0] = L.wvi1[0] = L.wvi2[0] = 0.;
}

void build_fband(double *fband /* [301][4], T = 100..400 */)
{   // radset (src/phy_radiat.f90:660-692), epslw = 0.05
    const double eps1 = 1. - EPSLW;
    auto F = [&](int t, int jb) -> double & { return fband[(size_t)(t - 100) * 4 + (jb - 1)]; };
    for (int t = 200; t <= 320; ++t) {
        F(t, 2) = (0.148 - 3.0e-6 * (double)((t - 247) * (t - 247))) * eps1;
        F(t, 3) = (0.356 - 5.2e-6 * (double)((t - 282) * (t - 282))) * eps1;
        F(t, 4) = (0.314 + 1.0e-5 * (double)((t - 315) * (t - 315))) * eps1;
        F(t, 1) = eps1 - (F(t, 2) + F(t, 3) + F(t, 4));
    }
    for (int jb = 1; jb <= 4; ++jb) {
        for (int t = 100; t <= 199; ++t) F(t, jb) = F(200, jb);
        for (int t = 321; t <= 400; ++t) F(t, jb) = F(320, jb);
    }
}

void solar(double tyear, double csol, const double *clat, const double *slat, double *topsr)
{   // src/phy_radiat.f90:85-127 (Hartmann 1994)
    const double pigr = 2. * std::asin(1.), alpha = 2. * pigr * tyear;
    const double ca1 = std::cos(alpha), sa1 = std::sin(alpha);
    const double ca2 = ca1 * ca1 - sa1 * sa1, sa2 = 2. * sa1 * ca1;
    const double ca3 = ca1 * ca2 - sa1 * sa2, sa3 = sa1 * ca2 + sa2 * ca1;
    const double decl = 0.006918 - 0.399912 * ca1 + 0.070257 * sa1 - 0.006758 * ca2 + 0.000907 * sa2 - 0.002697 * ca3 + 0.001480 * sa3;
    const double fdis = 1.000110 + 0.034221 * ca1 + 0.001280 * sa1 + 0.000719 * ca2 + 0.000077 * sa2;
    const double cdecl = std::cos(decl), sdecl = std::sin(decl), tdecl = sdecl / cdecl;
    const double csolp = csol / pigr;
    for (int j = 0; j < IL; ++j) {
        const double ch0 = std::min(1., std::max(-1., -tdecl * slat[j] / clat[j]));
        const double h0 = std::acos(ch0), sh0 = std::sin(h0);
        topsr[j] = csolp * fdis * (h0 * slat[j] * sdecl + sh0 * clat[j] * cdecl);
    }
}

void sol_oz(double tyear, const double *clat, const double *slat, double *fsol, double *ozone, double *ozupp, double *zenit, double *stratz)
{   // src/phy_radiat.f90:1-83
    const double alpha = 4. * std::asin(1.) * (tyear + 10. / 365.), dalpha = 0.;
    const double coz1 = 1.0 * std::max(0., std::cos(alpha - dalpha)), coz2 = 1.8;
    const double azen = 1.0;
    const double rzen = -std::cos(alpha) * 23.45 * std::asin(1.) / 90.;
    const double czen = std::cos(rzen), szen = std::sin(rzen);
    const double fs0 = 6.;
    double topsr[IL];
    solar(tyear, 4. * SOLC, clat, slat, topsr);
    for (int j = 0; j < IL; ++j) {
        const double flat2 = 1.5 * slat[j] * slat[j] - 0.5;
        fsol[j] = topsr[j];
        ozupp[j] = 0.5 * EPSSW;
        ozone[j] = 0.4 * EPSSW * (1.0 + coz1 * slat[j] + coz2 * flat2);
        const double z = 1. - (clat[j] * czen + slat[j] * szen);
        zenit[j] = 1. + azen * (z * z);                       // **nzen with nzen = 2
        ozupp[j] = fsol[j] * ozupp[j] * zenit[j];
        ozone[j] = fsol[j] * ozone[j] * zenit[j];
        stratz[j] = std::max(fs0 - fsol[j], 0.);
    }
}

// ---------------------------------------------------------------- device: the parametrisations, one column each
__device__ __forceinline__ double qsat_of(double ta, double p)
{   // shtorh section 1 with P = p (already sigma * ps)
    const double e0 = 6.108e-3, c1 = 17.269, c2 = 21.875, t0 = 273.16, t1 = 35.86, t2 = 7.66;
    double q;
    if (ta >= t0) q = e0 * exp(c1 * (ta - t0) / (ta - t1));
    else q = e0 * exp(c2 * (ta - t0) / (ta - t2));
    return 622. * q / (p - 0.378 * q);
}

// fordate(0)'s grid-point work for one point (src/ini_fordate.f90:54-61,72-109), statements in the reference's order; sst = sst_am
// of the point (in the hybrid: the SST just assembled from the reservoirs' outputs).  Shared by the stand-alone kernel (physics.hip)
// and the hybrid engine's ingest kernel (hybrid.hip), which has the SST in a register when it gets here.
struct FordatePoint {
    const double *fmask_l, *fmask_s, *phis0, *stl_am, *alb0, *snowd_am, *sice_am;      // alb0 == NULL: albedos are the host's
    double *alb_l, *alb_s, *albsfc, *snowc;
};
__device__ __forceinline__ void fordate_point(const FordatePoint &a, int p, double sst, double &corh_t, double &corh_q)
{
    if (a.alb0) {                                                      // :54-61
        const double sc = fmin(1., a.snowd_am[p] / SD2SC);
        const double al = a.alb0[p] + sc * (ALBSN - a.alb0[p]);
        const double as = ALBSEA + a.sice_am[p] * (ALBICE - ALBSEA);
        a.snowc[p] = sc; a.alb_l[p] = al; a.alb_s[p] = as;
        a.albsfc[p] = as + a.fmask_l[p] * (al - as);
    }
    const double gamlat = GAMMA / (1000. * GG);                        // setgam :116-136
    const double ct = gamlat * a.phis0[p];                             // :77
    const double pexp = 1. / (RD * gamlat);                            // :91
    const double tsfc = a.fmask_l[p] * a.stl_am[p] + a.fmask_s[p] * sst;
    const double tref = tsfc + ct;
    const double psfc = pow(tsfc / tref, pexp);
    const double qref = qsat_of(tref, 1.0);                            // shtorh(0, ngp, tref, psfc_dummy = 1, -1., ...): P = ps(1)
    const double qsfc = qsat_of(tsfc, 1. * psfc);                      // shtorh(0, ngp, tsfc, psfc, 1., ...): P = sig ps(j), sig = 1
    corh_t = ct;
    corh_q = REFRH1 * (qref - qsfc);                                   // :109
}

// Level arrays are 1-based like the Fortran.  Every level index below is a compile-time constant once the loops are unrolled
// (run-time tops -- convection top, cloud top -- are predicates inside fixed-bound loops), so the arrays live in registers: with
// run-time indices the compiler put them in scratch (1264 B per lane) and the kernel spent most of its 26 us waiting on scratch
// round trips.  The winds enter the parametrisations at the lowest level only (suflux), so only ug1(kx), vg1(kx) are kept.
// Long-lived, rarely touched per-column arrays (the tendency accumulators, the short-wave heating, the long-wave table rows)
// are parked in LDS, lane-interleaved (element k of a thread's array at base[k * 64]): left to the register allocator they were
// spilled to scratch and re-read one waited-for load at a time (6 us of the kernel's tail).
struct LA {
    double *b;
    __device__ __forceinline__ double &operator[](int k) const { return b[k << 6]; }
};
template <int NB> struct LA2 {       // [level][band]
    double *b;
    __device__ __forceinline__ LA operator[](int k) const { return LA{b + ((k * NB) << 6)}; }
};
struct Column {
    double tg[NLP], qg[NLP], phig[NLP], se[NLP], rh[NLP], qsat[NLP];
    double usfc, vsfc, psg, rps;
};

__device__ void convmf(const PhysLev &L, const Column &c, int &itop, double &cbmf, double &precnv, double *dfse, double *dfqa)
{
    const int nl1 = KX - 1;
    const double fqmax = 5.;
    const double fm0 = P0 * L.dsig[KX] / (GG * TRCNV * 3600);
    const double rdps = 2. / (1. - PSMIN);
    for (int k = 1; k <= KX; ++k) { dfse[k] = 0.0; dfqa[k] = 0.0; }
    cbmf = 0.0; precnv = 0.0;
    double mss[NLP];
    for (int k = 2; k <= KX; ++k) mss[k] = c.se[k] + ALHC * c.qsat[k];
    const double rlhc = 1. / ALHC;
    itop = NLP;
    double qdif = 0.;
    const double psa = c.psg;
    if (psa > PSMIN) {
        const double mse0 = c.se[KX] + ALHC * c.qg[KX];
        double mse1 = c.se[nl1] + ALHC * c.qg[nl1];
        mse1 = fmin(mse0, mse1);
        const double mss0 = fmax(mse0, mss[KX]);
        int ktop1 = KX, ktop2 = KX;
        double msthr = 0.;
        for (int k = KX - 3; k >= 3; --k) {
            const double mss2 = mss[k] + L.wvi2[k] * (mss[k + 1] - mss[k]);
            if (mss0 > mss2) ktop1 = k;
            if (mse1 > mss2) { ktop2 = k; msthr = mss2; }
        }
        if (ktop1 < KX) {
            const double qthr0 = RHBL * c.qsat[KX], qthr1 = RHBL * c.qsat[nl1];
            const bool lqthr = c.qg[KX] > qthr0 && c.qg[nl1] > qthr1;
            if (ktop2 < KX) { itop = ktop1; qdif = fmax(c.qg[KX] - qthr0, (mse0 - msthr) * rlhc); }
            else if (lqthr) { itop = ktop1; qdif = c.qg[KX] - qthr0; }
        }
    }
    if (itop == NLP) return;
    int k = KX, k1 = k - 1;
    const double qmax = fmax(1.01 * c.qg[k], c.qsat[k]);
    double sb = c.se[k1] + L.wvi2[k1] * (c.se[k] - c.se[k1]);
    double qb = c.qg[k1] + L.wvi2[k1] * (c.qg[k] - c.qg[k1]);
    qb = fmin(qb, c.qg[k]);
    const double fpsa = psa * fmin(1., (psa - PSMIN) * rdps);
    double fmass = fm0 * fpsa * fmin(fqmax, qdif / (qmax - qb));
    cbmf = fmass;
    double fus = fmass * c.se[k], fuq = fmass * qmax;
    double fds = fmass * sb, fdq = fmass * qb;
    dfse[k] = fds - fus;
    dfqa[k] = fdq - fuq;
#pragma unroll
    for (k = KX - 1; k >= 4; --k) {          // do k = kx-1, itop+1, -1 with itop >= 3: the top is a predicate (no run-time index)
        if (k < itop + 1) continue;
        k1 = k - 1;
        dfse[k] = fus - fds;
        dfqa[k] = fuq - fdq;
        const double enmass = L.entr[k] * psa * cbmf;
        fmass = fmass + enmass;
        fus = fus + enmass * c.se[k];
        fuq = fuq + enmass * c.qg[k];
        sb = c.se[k1] + L.wvi2[k1] * (c.se[k] - c.se[k1]);
        qb = c.qg[k1] + L.wvi2[k1] * (c.qg[k] - c.qg[k1]);
        fds = fmass * sb;
        fdq = fmass * qb;
        dfse[k] = dfse[k] + fds - fus;
        dfqa[k] = dfqa[k] + fdq - fuq;
        const double delq = RHIL * c.qsat[k] - c.qg[k];
        if (delq > 0.0) {
            const double fsq = SMF * cbmf * delq;
            dfqa[k] = dfqa[k] + fsq;
            dfqa[KX] = dfqa[KX] - fsq;
        }
    }
#pragma unroll
    for (k = 3; k <= KX - 1; ++k) {          // k = itop
        if (k != itop) continue;
        const double qsatb = c.qsat[k] + L.wvi2[k] * (c.qsat[k + 1] - c.qsat[k]);
        precnv = fmax(fuq - fmass * qsatb, 0.0);
        dfse[k] = fus - fds + ALHC * precnv;
        dfqa[k] = fuq - fdq - precnv;
    }
}

__device__ void lscond(const PhysLev &L, const Column &c, int &itop, double &precls, double *dtlsc, double *dqlsc)
{
    const double qsmax = 10.;
    const double rtlsc = 1. / (TRLSC * 3600.), tfact = ALHC / CP, prg = P0 / GG;
    dtlsc[1] = 0.; dqlsc[1] = 0.;
    precls = 0.;
    const double psa = c.psg, psa2 = psa * psa;
    for (int k = 2; k <= KX; ++k) {
        const double sig2 = L.sig[k] * L.sig[k];
        double rhref = RHLSC + DRHLSC * (sig2 - 1.);
        if (k == KX) rhref = fmax(rhref, RHBLSC);
        const double dqmax = qsmax * sig2 * rtlsc;
        const double dqa = rhref * c.qsat[k] - c.qg[k];
        if (dqa < 0.0) {
            itop = min(k, itop);
            dqlsc[k] = dqa * rtlsc;
            dtlsc[k] = tfact * fmin(-dqlsc[k], dqmax * psa2);
        } else {
            dqlsc[k] = 0.;
            dtlsc[k] = 0.;
        }
    }
    for (int k = 2; k <= KX; ++k) {
        const double pfact = L.dsig[k] * prg;
        precls = precls - pfact * dqlsc[k];
    }
    precls = precls * psa;
}

__device__ void cloud(const Column &c, double precnv, double precls, int iptop, double gse, double fmask, int &icltop, double &cloudc,
                      double &clstr, double &qcloud)
{
    const int nl1 = KX - 1;
    const double rrcl = 1. / (RHCL2 - RHCL1);
    if (c.rh[nl1] > RHCL1) { cloudc = c.rh[nl1] - RHCL1; icltop = nl1; }
    else { cloudc = 0.; icltop = NLP; }
    for (int k = 3; k <= KX - 2; ++k) {
        const double drh = c.rh[k] - RHCL1;
        if (drh > cloudc && c.qg[k] > QACL) { cloudc = drh; icltop = k; }
    }
    const double cl1 = fmin(1., cloudc * rrcl);
    const double pr1 = fmin(PMAXCL, 86.4 * (precnv + precls));
    cloudc = fmin(1., WPCL * sqrt(pr1) + cl1 * cl1);
    icltop = min(iptop, icltop);
    qcloud = c.qg[nl1];
    const double clfact = 1.2, rgse = 1. / (GSE_S1 - GSE_S0);
    const double fstab = fmax(0., fmin(1., rgse * (gse - GSE_S0)));
    clstr = fstab * fmax(CLSMAX - clfact * cloudc, 0.);
    const double clstrl = fmax(clstr, CLSMINL) * c.rh[KX];
    clstr = clstr + fmask * (clstrl - clstr);
}

// radsw: tau2 is the column's transmissivity work array [level 1..8][band 1..4]; on return it holds the LONG-wave values
__device__ void radsw(const PhysLev &L, const Column &c, int icltop, double cloudc, double clstr, double qcloud, double fsol, double ozone,
                      double ozupp, double zenit, double stratz, double albsfc, double (*tau2)[5], double *stratc, double &fsfcd,
                      double &fsfc, double &ftop, double *dfabs)
{
    const int nl1 = KX - 1;
    const double fband2 = 0.05, fband1 = 1. - fband2;
    const double psa = c.psg;
    for (int k = 1; k <= KX; ++k) for (int b = 1; b <= 4; ++b) tau2[k][b] = 0.0;
#pragma unroll
    for (int k = 1; k <= KX; ++k)
        if (k == icltop) tau2[k][3] = ALBCL * cloudc;
    tau2[KX][3] = ALBCLS * clstr;
    const double psaz = psa * zenit;
    const double acloud = cloudc * fmin(ABSCL1 * qcloud, ABSCL2);
    double deltap = psaz * L.dsig[1];
    tau2[1][1] = exp(-deltap * ABSDRY);
    for (int k = 2; k <= nl1; ++k) {
        const double abs1 = ABSDRY + ABSAER * (L.sig[k] * L.sig[k]);
        deltap = psaz * L.dsig[k];
        if (k >= icltop) tau2[k][1] = exp(-deltap * (abs1 + ABSWV1 * c.qg[k] + acloud));
        else tau2[k][1] = exp(-deltap * (abs1 + ABSWV1 * c.qg[k]));
    }
    {
        const double abs1 = ABSDRY + ABSAER * (L.sig[KX] * L.sig[KX]);
        deltap = psaz * L.dsig[KX];
        tau2[KX][1] = exp(-deltap * (abs1 + ABSWV1 * c.qg[KX]));
    }
    for (int k = 2; k <= KX; ++k) {
        deltap = psaz * L.dsig[k];
        tau2[k][2] = exp(-deltap * ABSWV2 * c.qg[k]);
    }
    ftop = fsol;
    double flux1 = fsol * fband1, flux2 = fsol * fband2;
    int k = 1;
    dfabs[k] = flux1;
    flux1 = tau2[k][1] * (flux1 - ozupp * psa);
    dfabs[k] = dfabs[k] - flux1;
    k = 2;
    dfabs[k] = flux1;
    flux1 = tau2[k][1] * (flux1 - ozone * psa);
    dfabs[k] = dfabs[k] - flux1;
    for (k = 3; k <= KX; ++k) {
        tau2[k][3] = flux1 * tau2[k][3];
        flux1 = flux1 - tau2[k][3];
        dfabs[k] = flux1;
        flux1 = tau2[k][1] * flux1;
        dfabs[k] = dfabs[k] - flux1;
    }
    for (k = 2; k <= KX; ++k) {
        dfabs[k] = dfabs[k] + flux2;
        flux2 = tau2[k][2] * flux2;
        dfabs[k] = dfabs[k] - flux2;
    }
    fsfcd = flux1 + flux2;
    flux1 = flux1 * albsfc;
    fsfc = fsfcd - flux1;
    for (k = KX; k >= 1; --k) {
        dfabs[k] = dfabs[k] + flux1;
        flux1 = tau2[k][1] * flux1;
        dfabs[k] = dfabs[k] - flux1;
        flux1 = flux1 + tau2[k][3];
    }
    ftop = ftop - flux1;
    // 5. long-wave transmissivities
    k = 1;
    deltap = psa * L.dsig[k];
    tau2[k][1] = exp(-deltap * ABLWIN);
    tau2[k][2] = exp(-deltap * ABLCO2);
    tau2[k][3] = 1.;
    tau2[k][4] = 1.;
    for (k = 2; k <= KX; k += KX - 2) {
        deltap = psa * L.dsig[k];
        tau2[k][1] = exp(-deltap * ABLWIN);
        tau2[k][2] = exp(-deltap * ABLCO2);
        tau2[k][3] = exp(-deltap * ABLWV1 * c.qg[k]);
        tau2[k][4] = exp(-deltap * ABLWV2 * c.qg[k]);
    }
    const double acl = cloudc * ABLCL2;
    for (k = 3; k <= nl1; ++k) {
        deltap = psa * L.dsig[k];
        const double acloud1 = k < icltop ? acl : ABLCL1 * cloudc;
        tau2[k][1] = exp(-deltap * (ABLWIN + acloud1));
        tau2[k][2] = exp(-deltap * ABLCO2);
        tau2[k][3] = exp(-deltap * fmax(ABLWV1 * c.qg[k], acl));
        tau2[k][4] = exp(-deltap * fmax(ABLWV2 * c.qg[k], acl));
    }
    const double eps1 = EPSLW / (L.dsig[1] + L.dsig[2]);
    stratc[1] = stratz * psa;
    stratc[2] = eps1 * psa;
}

__device__ __forceinline__ double fband_of(const double *__restrict__ fband, double t, int jb)
{
    int it = (int)floor(t + 0.5);                       // nint for the positive temperatures met here
    it = min(400, max(100, it));                        // the Fortran table is fband(100:400,4); stay inside it
    return fband[(size_t)(it - 100) * 4 + (jb - 1)];
}

struct LwState { double st4a[NLP][3], flux[5]; LA2<5> fb; };

// radlw(-1, ...): downward pass.  radlw(+1, ...): upward pass; both use the column's tau2 / st4a / flux
__device__ void radlw_down(const PhysLev &L, const Column &c, const double *__restrict__ fband, const double (*tau2)[5], LwState &s, double &fsfcd,
                           double *dfabs)
{
    const int nl1 = KX - 1;
    for (int k = 1; k <= nl1; ++k) s.st4a[k][1] = c.tg[k] + L.wvi2[k] * (c.tg[k + 1] - c.tg[k]);
    s.st4a[1][2] = 0.75 * c.tg[1] + 0.25 * s.st4a[1][1];
    s.st4a[2][2] = 0.50 * c.tg[2] + 0.25 * (s.st4a[1][1] + s.st4a[2][1]);
    const double anis = 1.0, anish = 0.5 * anis;
    for (int k = 3; k <= nl1; ++k) s.st4a[k][2] = anish * fmax(s.st4a[k][1] - s.st4a[k - 1][1], 0.);
    s.st4a[KX][2] = anis * fmax(c.tg[KX] - s.st4a[nl1][1], 0.);
    for (int k = 1; k <= 2; ++k) {
        const double t2 = s.st4a[k][2] * s.st4a[k][2];
        s.st4a[k][1] = SBC * (t2 * t2);                 // **4
        s.st4a[k][2] = 0.;
    }
    for (int k = 3; k <= KX; ++k) {
        const double st3a = SBC * (c.tg[k] * c.tg[k] * c.tg[k]);
        s.st4a[k][1] = st3a * c.tg[k];
        s.st4a[k][2] = 4. * st3a * s.st4a[k][2];
    }
    fsfcd = 0.0;
    for (int k = 1; k <= KX; ++k) dfabs[k] = 0.0;
    // the table rows of the 8 level temperatures, fetched in one batch (one exposed latency) and kept for the upward pass
    for (int k = 1; k <= KX; ++k)
        for (int jb = 1; jb <= 4; ++jb) s.fb[k][jb] = fband_of(fband, c.tg[k], jb);
    int k = 1;
    for (int jb = 1; jb <= 2; ++jb) {
        const double emis = 1. - tau2[k][jb];
        const double brad = s.fb[k][jb] * (s.st4a[k][1] + emis * s.st4a[k][2]);
        s.flux[jb] = emis * brad;
        dfabs[k] = dfabs[k] - s.flux[jb];
    }
    s.flux[3] = 0.0; s.flux[4] = 0.0;
    for (int jb = 1; jb <= 4; ++jb)
        for (k = 2; k <= KX; ++k) {
            const double emis = 1. - tau2[k][jb];
            const double brad = s.fb[k][jb] * (s.st4a[k][1] + emis * s.st4a[k][2]);
            dfabs[k] = dfabs[k] + s.flux[jb];
            s.flux[jb] = tau2[k][jb] * s.flux[jb] + emis * brad;
            dfabs[k] = dfabs[k] - s.flux[jb];
        }
    for (int jb = 1; jb <= 4; ++jb) fsfcd = fsfcd + EMISFC * s.flux[jb];
    const double eps1 = EPSLW * EMISFC;
    const double corlw = eps1 * s.st4a[KX][1];
    dfabs[KX] = dfabs[KX] - corlw;
    fsfcd = fsfcd + corlw;
}

__device__ void radlw_up(const PhysLev &L, const Column &c, const double *__restrict__ fband, const double (*tau2)[5], const double *stratc,
                         LwState &s, double ts, double fsfcd, double fsfcu, double &fsfc, double &ftop, double *dfabs)
{
    const double refsfc = 1. - EMISFC;
    fsfc = fsfcu - fsfcd;
    for (int jb = 1; jb <= 4; ++jb) s.flux[jb] = fband_of(fband, ts, jb) * fsfcu + refsfc * s.flux[jb];
    dfabs[KX] = dfabs[KX] + EPSLW * fsfcu;
    for (int jb = 1; jb <= 4; ++jb)
        for (int k = KX; k >= 2; --k) {
            const double emis = 1. - tau2[k][jb];
            const double brad = s.fb[k][jb] * (s.st4a[k][1] - emis * s.st4a[k][2]);
            dfabs[k] = dfabs[k] + s.flux[jb];
            s.flux[jb] = tau2[k][jb] * s.flux[jb] + emis * brad;
            dfabs[k] = dfabs[k] - s.flux[jb];
        }
    const int k = 1;
    for (int jb = 1; jb <= 2; ++jb) {
        const double emis = 1. - tau2[k][jb];
        const double brad = s.fb[k][jb] * (s.st4a[k][1] - emis * s.st4a[k][2]);
        dfabs[k] = dfabs[k] + s.flux[jb];
        s.flux[jb] = tau2[k][jb] * s.flux[jb] + emis * brad;
        dfabs[k] = dfabs[k] - s.flux[jb];
    }
    const double corlw1 = L.dsig[1] * stratc[2] * s.st4a[1][1] + stratc[1];
    const double corlw2 = L.dsig[2] * stratc[2] * s.st4a[2][1];
    dfabs[1] = dfabs[1] - corlw1;
    dfabs[2] = dfabs[2] - corlw2;
    ftop = corlw1 + corlw2;
    for (int jb = 1; jb <= 4; ++jb) ftop = ftop + s.flux[jb];
}

struct Surface { double ustr[4], vstr[4], shf[4], evap[4], slru[4], hfluxn[3], tsfc, tskin, u0, v0, t0, q0; };

// suflux with lfluxland = .true. (the only call in the hybrid configuration: icsea = 0, src/mod_cpl_flags.f90:13)
__device__ void suflux(const PhysLev &L, const Column &c, double phi0, double fmask, double tland, double tsea, double swav, double ssrd,
                       double slrd, double alb_l, double alb_s, double snowc, double forog, double sqclat, Surface &o)
{
    const double psa = c.psg;
    const double esbc = EMISFC * SBC, esbc4 = 4. * esbc;
    const double ghum0 = 1. - FHUM0;
    const double dlambda = CLAMBSN - CLAMBDA;
    const int nl1 = KX - 1;
    o.u0 = FWIND0 * c.usfc;
    o.v0 = FWIND0 * c.vsfc;
    const double gtemp0 = 1. - FTEMP0, rcp = 1. / CP, rdphi0 = -1. / (RD * 288. * L.sigl[KX]);
    const double dt1 = L.wvi2[KX] * (c.tg[KX] - c.tg[nl1]);
    double t1[3], t2[3], q1[3], qsat0[3], denvvs[3];
    t1[1] = c.tg[KX] + dt1;
    t1[2] = t1[1] + phi0 * dt1 * rdphi0;
    t2[2] = c.tg[KX] + rcp * c.phig[KX];
    t2[1] = t2[2] - rcp * phi0;
    if (c.tg[KX] > c.tg[nl1]) {
        t1[1] = FTEMP0 * t1[1] + gtemp0 * t2[1];
        t1[2] = FTEMP0 * t1[2] + gtemp0 * t2[2];
    } else {
        t1[1] = c.tg[KX];
        t1[2] = c.tg[KX];
    }
    o.t0 = t1[2] + fmask * (t1[1] - t1[2]);
    const double prd = P0 / RD, vg2 = VGUST * VGUST;
    denvvs[0] = (prd * psa / o.t0) * sqrt(o.u0 * o.u0 + o.v0 * o.v0 + vg2);
    double tskin = tland + CTDAY * sqclat * ssrd * (1. - alb_l) * psa;
    const double rdth = FSTAB / DTHETA, astab = 0.5;
    double dthl;
    if (tskin > t2[1]) dthl = fmin(DTHETA, tskin - t2[1]);
    else dthl = fmax(-DTHETA, astab * (tskin - t2[1]));
    denvvs[1] = denvvs[0] * (1. + dthl * rdth);
    const double cdldv = CDL * denvvs[0] * forog;
    o.ustr[1] = -cdldv * c.usfc;
    o.vstr[1] = -cdldv * c.vsfc;
    const double chlcp = CHL * CP;
    o.shf[1] = chlcp * denvvs[1] * (tskin - t1[1]);
    if (FHUM0 > 0.) {
        qsat0[1] = qsat_of(t1[1], psa);
        q1[1] = c.rh[KX] * qsat0[1];
        q1[1] = FHUM0 * q1[1] + ghum0 * c.qg[KX];
    } else {
        q1[1] = c.qg[KX];
    }
    qsat0[1] = qsat_of(tskin, psa);
    o.evap[1] = CHL * denvvs[1] * fmax(0., swav * qsat0[1] - q1[1]);
    const double tsk3 = tskin * tskin * tskin;
    const double dslr = esbc4 * tsk3;
    o.slru[1] = esbc * tsk3 * tskin;
    o.hfluxn[1] = ssrd * (1. - alb_l) + slrd - (o.slru[1] + o.shf[1] + ALHC * o.evap[1]);
    {   // 3.2 skin temperature from the energy balance (lskineb)
        const double clamb = CLAMBDA + snowc * dlambda;
        o.hfluxn[1] = o.hfluxn[1] - clamb * (tskin - tland);
        double dtskin = tskin + 1.;
        qsat0[2] = qsat_of(dtskin, psa);
        if (o.evap[1] > 0) qsat0[2] = swav * (qsat0[2] - qsat0[1]);
        else qsat0[2] = 0.;
        const double dhfdt = clamb + dslr + CHL * denvvs[1] * (CP + ALHC * qsat0[2]);
        dtskin = o.hfluxn[1] / dhfdt;
        tskin = tskin + dtskin;
        o.shf[1] = o.shf[1] + chlcp * denvvs[1] * dtskin;
        o.evap[1] = o.evap[1] + CHL * denvvs[1] * qsat0[2] * dtskin;
        o.slru[1] = o.slru[1] + dslr * dtskin;
        o.hfluxn[1] = clamb * (tskin - tland);
    }
    double dths;
    if (tsea > t2[2]) dths = fmin(DTHETA, tsea - t2[2]);
    else dths = fmax(-DTHETA, astab * (tsea - t2[2]));
    denvvs[2] = denvvs[0] * (1. + dths * rdth);
    if (FHUM0 > 0.) {
        qsat0[2] = qsat_of(t1[2], psa);
        q1[2] = c.rh[KX] * qsat0[2];
        q1[2] = FHUM0 * q1[2] + ghum0 * c.qg[KX];
    } else {
        q1[2] = c.qg[KX];
    }
    const int ks = 2;
    const double cdsdv = CDS * denvvs[ks];
    o.ustr[2] = -cdsdv * c.usfc;
    o.vstr[2] = -cdsdv * c.vsfc;
    const double chscp = CHS * CP;
    o.shf[2] = chscp * denvvs[ks] * (tsea - t1[2]);
    qsat0[2] = qsat_of(tsea, psa);
    o.evap[2] = CHS * denvvs[ks] * (qsat0[2] - q1[2]);
    {
        const double ts2 = tsea * tsea;
        o.slru[2] = esbc * (ts2 * ts2);                 // tsea**4
    }
    o.hfluxn[2] = ssrd * (1. - alb_s) + slrd - (o.slru[2] + o.shf[2] + ALHC * o.evap[2]);
    o.ustr[3] = o.ustr[2] + fmask * (o.ustr[1] - o.ustr[2]);
    o.vstr[3] = o.vstr[2] + fmask * (o.vstr[1] - o.vstr[2]);
    o.shf[3] = o.shf[2] + fmask * (o.shf[1] - o.shf[2]);
    o.evap[3] = o.evap[2] + fmask * (o.evap[1] - o.evap[2]);
    o.slru[3] = o.slru[2] + fmask * (o.slru[1] - o.slru[2]);
    o.tsfc = tsea + fmask * (tland - tsea);
    o.tskin = tsea + fmask * (tskin - tsea);
    o.t0 = t1[2] + fmask * (t1[1] - t1[2]);
    o.q0 = q1[2] + fmask * (q1[1] - q1[2]);
}

__device__ void vdifsc(const PhysLev &L, const Column &c, int icnv, double *tt, double *qt)
{
    const int nl1 = KX - 1;
    const double cshc = L.dsig[KX] / 3600.;
    const double cvdi = (L.sigh[nl1] - L.sigh[1]) / ((nl1 - 1) * 3600.);
    const double fshcq = cshc / TRSHC, fshcse = cshc / (TRSHC * CP);
    const double fvdiq = cvdi / TRVDI, fvdise = cvdi / (TRVDS * CP);
    double rsig[NLP], rsig1[NLP];
    for (int k = 1; k <= nl1; ++k) { rsig[k] = 1. / L.dsig[k]; rsig1[k] = 1. / (1. - L.sigh[k]); }
    rsig[KX] = 1. / L.dsig[KX];
    for (int k = 1; k <= KX; ++k) { tt[k] = 0.0; qt[k] = 0.0; }     // utenvd = vtenvd = 0 (:44-49): the winds only feel the surface stress
    double drh0 = RHGRAD * (L.sig[KX] - L.sig[nl1]);
    double fvdiq2 = fvdiq * L.sigh[nl1];
    {
        const double dmse = (c.se[KX] - c.se[nl1]) + ALHC * (c.qg[KX] - c.qsat[nl1]);
        const double drh = c.rh[KX] - c.rh[nl1];
        double fcnv = 1.;
        if (dmse >= 0.0) {
            if (icnv > 0) fcnv = REDSHC;
            const double fluxse = fcnv * fshcse * dmse;
            tt[nl1] = fluxse * rsig[nl1];
            tt[KX] = -fluxse * rsig[KX];
            if (drh >= 0.0) {
                const double fluxq = fcnv * fshcq * c.qsat[KX] * drh;
                qt[nl1] = fluxq * rsig[nl1];
                qt[KX] = -fluxq * rsig[KX];
            }
        } else if (drh >= drh0) {
            const double fluxq = fvdiq2 * c.qsat[nl1] * drh;
            qt[nl1] = fluxq * rsig[nl1];
            qt[KX] = -fluxq * rsig[KX];
        }
    }
    for (int k = 3; k <= KX - 2; ++k) {
        if (L.sigh[k] > 0.5) {
            drh0 = RHGRAD * (L.sig[k + 1] - L.sig[k]);
            fvdiq2 = fvdiq * L.sigh[k];
            const double drh = c.rh[k + 1] - c.rh[k];
            if (drh >= drh0) {
                const double fluxq = fvdiq2 * c.qsat[k] * drh;
                qt[k] = qt[k] + fluxq * rsig[k];
                qt[k + 1] = qt[k + 1] - fluxq * rsig[k + 1];
            }
        }
    }
    for (int k = 1; k <= nl1; ++k) {
        const double se0 = c.se[k + 1] + SEGRAD * (c.phig[k] - c.phig[k + 1]);
        if (c.se[k] < se0) {
            const double fluxse = fvdise * (se0 - c.se[k]);
            tt[k] = tt[k] + fluxse * rsig[k];
            for (int k1 = k + 1; k1 <= KX; ++k1) tt[k1] = tt[k1] - fluxse * rsig1[k];
        }
    }
}

// phase time stamps of one tropical workgroup (profiles/micro/physics_phase_stamps.py); compiled in with -DSML_PHYS_STAMPS only
__device__ unsigned long long g_phys_dbg[48];
#ifdef SML_PHYS_STAMPS
#define PSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); \
        if (blockIdx.x == 36 && threadIdx.x == 0) g_phys_dbg[slot] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
// the two-wavefront kernel of dynamics.hip: lane 0 of either wavefront stamps into its own half, [wave][16]
#ifdef SML_PHYS_STAMPS_LIGHT      // without the wait for outstanding memory operations: where the wavefront IS at that moment, not when its loads have landed
#define CSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 36 && (threadIdx.x & 63) == 0) smlphys::g_phys_dbg[(threadIdx.x >> 6) * 16 + (slot)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); \
        if (blockIdx.x == 36 && (threadIdx.x & 63) == 0) smlphys::g_phys_dbg[(threadIdx.x >> 6) * 16 + (slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
// (Neither kind pins the ARITHMETIC: a scheduling barrier stops the machine scheduler, not the optimiser's sinking of a computation to its
// first use -- in the three-wavefront kernel ~1000 double-precision instructions of the long-wave scheme sit behind the stamp that follows
// radlw_up in the source.  What a wavefront's chain costs is therefore read from its arrival at the workgroup barrier, span.h.)
#elif defined(SML_WAVE_SPAN)
// the per-wavefront diagnostic build (span.h): lane 0 of a wavefront notes the clock at the chain's stamps in LDS (no wait added); the kernel
// packs some of them into its wavefront record (SML_SPAN_PACK_MARKS)
__shared__ unsigned long long g_wave_marks[4][8];
#define PSTAMP(slot) do { } while (0)
#define CSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0) smlphys::g_wave_marks[(threadIdx.x >> 6) & 3][(slot) & 7] = (unsigned long long)wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#define CSTAMP(slot) do { } while (0)
#endif

// where phypar's grid-point inputs of time level 1 are: 8 consecutive level fields each for t, q, phi; ONE field each for the
// lowest-level winds ug1(:,kx), vg1(:,kx) (the only wind levels the parametrisations read) and for log(ps)
struct PhysIn { const double *usfc, *vsfc, *t, *q, *phi, *ps; };

// the grid-point part of phypar (src/phy_phypar.f90:80-230) for one column.
// tend: the grid-point tendencies utend, vtend, ttend, qtend as 8 consecutive fields each, starting at fields off_u, off_v,
// off_t, off_q of `tend`.  accumulate != 0: they hold the dynamical tendencies and the physics is ADDED in the reference's order
// (:117-118,172-174,193-196: each `x = x + a + b` is two roundings, left to right); accumulate == 0: they are overwritten with
// the physics tendencies alone.  The wind tendencies of the physics are zero above the lowest level (vdifsc :44-49 leaves
// utenvd = vtenvd = 0; only the surface stress acts), so with accumulate the upper 7 levels of utend, vtend are not touched.
// want_diag == 0 skips the 2-D diagnostics.
//
// The sequence is cut into pieces that two wavefronts can run side by side for the same 64 columns (dynamics.hip,
// k_gridtend_physics: wave 0 = grid-point dynamics, convection, condensation, vertical diffusion; wave 1 = radiation and surface
// fluxes) or one wavefront runs back to back (physics_column / k_physics).  The pieces talk through the workgroup's LDS park,
// lane-interleaved slots of 64 doubles (element k of a column's array at park[(slot + k) * 64 + lane]):
//   P_TT, P_QT  ttend, qtend accumulators, filled by the caller with the dynamical tendencies (or zeros)
//   P_RSW       short-wave heating (tendency) of this or the last short-wave step        P_RLW  long-wave heating (tendency)
//   P_FB        long-wave band fractions of the level temperatures [level][band]
//   P_XB        radiation chain -> finish: ustr, vstr, shf, evap of the surface (fmask-weighted)
//   P_UV        (three-wavefront form) dynamics wave -> finish: utend(kx), vtend(kx)
// The sums keep the reference's order whatever runs where: ttend = ((dyn + cnv) + lsc), then (+ rsw) + rlw, then + pbl.
constexpr int P_TT = 0, P_QT = NLP, P_RSW = 2 * NLP, P_FB = 3 * NLP, P_RLW = 8 * NLP, P_XB = 9 * NLP, P_UV = 9 * NLP + 4, P_SLOTS = 9 * NLP + 6;
constexpr int PARK_DOUBLES = P_SLOTS * 64;

__device__ __forceinline__ LA park_array(double *park, int slot, int lane) { return LA{park + slot * 64 + lane}; }

// phypar 1.1-1.2 (:54-100): the column's grid-point state of time level 1 (column_load: the loads only, so that a caller can issue
// them together with its other loads) and its thermodynamic variables (column_thermo)
__device__ __forceinline__ void column_load(const PhysIn &in, int p, Column &c)
{
    for (int k = 1; k <= KX; ++k) {
        c.tg[k] = in.t[(size_t)(k - 1) * GR + p];
        c.qg[k] = in.q[(size_t)(k - 1) * GR + p];
        c.phig[k] = in.phi[(size_t)(k - 1) * GR + p];
    }
    c.usfc = in.usfc[p];
    c.vsfc = in.vsfc[p];
    c.psg = in.ps[p];               // log(ps) until column_thermo
}

__device__ __forceinline__ void column_thermo(const PhysLev &L, Column &c)
{
    const double pslg = c.psg;
    c.psg = exp(pslg);
    c.rps = 1. / c.psg;
    for (int k = 1; k <= KX; ++k) {
        c.qg[k] = fmax(c.qg[k], 0.);
        c.se[k] = CP * c.tg[k] + c.phig[k];
        c.qsat[k] = qsat_of(c.tg[k], L.sig[k] * c.psg);
        c.rh[k] = c.qg[k] / c.qsat[k];
    }
}

__device__ __forceinline__ void column_state(const PhysLev &L, const PhysIn &in, int p, Column &c)
{
    column_load(in, p, c);
    column_thermo(L, c);
}

// what the radiation / surface chain reads from memory: boundary fields of the column, zonal fields of its latitude and, when this
// is not a short-wave step, what the last one left (the short-wave heating goes straight into the park)
struct RadIn {
    double fmask, phis0, tland, tsea, swav, alb_l, alb_s, snowc, forog, albsfc, fsol, ozone, ozupp, zenit, stratz, sqclat;
    double tau2[NLP][5], stratc[3], ssrd;
};

__device__ __forceinline__ void radiation_load(const PhysDev &D, int p, int lradsw, double *park, int lane, RadIn &r)
{
    const int jlat = p / IX;
    r.fmask = D.fmask[p]; r.phis0 = D.phis0[p]; r.tland = D.tland[p]; r.tsea = D.tsea[p]; r.swav = D.swav[p];
    r.alb_l = D.alb_l[p]; r.alb_s = D.alb_s[p]; r.snowc = D.snowc[p]; r.forog = D.forog[p]; r.albsfc = D.albsfc[p];
    r.sqclat = D.sqclat[jlat];
    if (lradsw) {
        r.fsol = D.fsol[jlat]; r.ozone = D.ozone[jlat]; r.ozupp = D.ozupp[jlat]; r.zenit = D.zenit[jlat]; r.stratz = D.stratz[jlat];
    } else {
        LA tt_rsw = park_array(park, P_RSW, lane);
        for (int k = 1; k <= KX; ++k) {
            for (int b = 1; b <= 4; ++b) r.tau2[k][b] = D.tau2[((size_t)(b - 1) * KX + (k - 1)) * GR + p];
            tt_rsw[k] = D.tt_rsw[(size_t)(k - 1) * GR + p];
        }
    }
    // (unconditionally -- a short-wave step overwrites them: stored only in the else branch, the compiler turned the three stores
    // into stores at a selected address and the struct went to scratch, 40 B per lane with five waited-for scratch loads)
    r.stratc[0] = 0.;
    r.stratc[1] = D.stratc[p];
    r.stratc[2] = D.stratc[GR + p];
    r.ssrd = D.ssrd[p];
}

// phypar 2 (:102-118): convection and large-scale condensation as tendencies (tt_cnv, qt_cnv already scaled by rps grdscp / grdsig)
__device__ __forceinline__ void moist_tendencies(const PhysLev &L, const PhysDev &D, const Column &c, int p, int want_diag, int &iptop, int &icnv,
                                                 double &precnv, double &precls, double (&a1)[NLP], double (&a2)[NLP], double (&b1)[NLP], double (&b2)[NLP])
{
    double cbmf;
    convmf(L, c, iptop, cbmf, precnv, a1, a2);
    for (int k = 2; k <= KX; ++k) { a1[k] = a1[k] * c.rps * L.grdscp[k]; a2[k] = a2[k] * c.rps * L.grdsig[k]; }
    icnv = KX - iptop;
    lscond(L, c, iptop, precls, b1, b2);
    if (want_diag) {
        D.diag[(size_t)D_PRECNV * GR + p] = precnv; D.diag[(size_t)D_PRECLS * GR + p] = precls; D.diag[(size_t)D_CBMF * GR + p] = cbmf;
        D.diag[(size_t)D_IPTOP * GR + p] = iptop;
    }
}

// ... and ttend = ttend + tt_cnv + tt_lsc (same for q) on the park's accumulators
__device__ __forceinline__ void chain_moist(const PhysLev &L, const PhysDev &D, const Column &c, int p, int want_diag, double *park, int lane,
                                            int &iptop, int &icnv, double &precnv, double &precls)
{
    LA tt = park_array(park, P_TT, lane), qt = park_array(park, P_QT, lane);
    double a1[NLP], a2[NLP], b1[NLP], b2[NLP];
    moist_tendencies(L, D, c, p, want_diag, iptop, icnv, precnv, precls, a1, a2, b1, b2);
    for (int k = 1; k <= KX; ++k) { tt[k] = tt[k] + a1[k] + b1[k]; qt[k] = qt[k] + a2[k] + b2[k]; }
}

// phypar 3 (:120-176): clouds and short-wave radiation on short-wave steps (else what the last one left), long-wave radiation
// down, surface fluxes, long-wave radiation up.  Leaves the two heating tendencies and the surface fluxes in the park.
// PARK_READY: called once the park holds everything this chain hands on (the heating tendencies and the surface fluxes), before the 2-D
// diagnostics are stored -- the multi-wavefront kernels put their workgroup barrier there, so that the wavefront that finishes the column
// does not wait for this one's stores to drain
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <class PARK_READY = NoHook>
__device__ __forceinline__ void chain_radiation(const PhysLev &L, const PhysDev &D, const Column &c, RadIn &r, int p, int lradsw, int want_diag,
                                                double *park, int lane, double precnv, double precls, int iptop, PARK_READY park_ready = PARK_READY())
{
    LA tt_rsw = park_array(park, P_RSW, lane), tt_rlw = park_array(park, P_RLW, lane), xb = park_array(park, P_XB, lane);
    double (&tau2)[NLP][5] = r.tau2;
    double (&stratc)[3] = r.stratc;
    double &ssrd = r.ssrd;
    if (lradsw) {
        double ssr = 0., tsr = 0., cloudc = 0., clstr = 0., qcloud, a1[NLP];
        int icltop = 0;
        const double gse = (c.se[KX - 1] - c.se[KX]) / (c.phig[KX - 1] - c.phig[KX]);
        cloud(c, precnv, precls, iptop, gse, r.fmask, icltop, cloudc, clstr, qcloud);
        radsw(L, c, icltop, cloudc, clstr, qcloud, r.fsol, r.ozone, r.ozupp, r.zenit, r.stratz, r.albsfc, tau2, stratc, ssrd, ssr, tsr, a1);
        for (int k = 1; k <= KX; ++k) {
            const double h = a1[k] * c.rps * L.grdscp[k];
            tt_rsw[k] = h;
            D.tt_rsw[(size_t)(k - 1) * GR + p] = h;
            for (int b = 1; b <= 4; ++b) D.tau2[((size_t)(b - 1) * KX + (k - 1)) * GR + p] = tau2[k][b];
        }
        D.stratc[p] = stratc[1];
        D.stratc[GR + p] = stratc[2];
        D.ssrd[p] = ssrd;
        if (want_diag) {
            D.diag[(size_t)D_CLOUDC * GR + p] = cloudc; D.diag[(size_t)D_CLSTR * GR + p] = clstr;
            D.diag[(size_t)D_TSR * GR + p] = tsr; D.diag[(size_t)D_SSR * GR + p] = ssr; D.diag[(size_t)D_ICLTOP * GR + p] = icltop;
        }
    }
    CSTAMP(3);
    LwState lw;
    lw.fb = LA2<5>{park + P_FB * 64 + lane};
    double slrd, slr, olr, dfabs[NLP];
    radlw_down(L, c, D.fband, tau2, lw, slrd, dfabs);
    CSTAMP(4);
    Surface sf;
    suflux(L, c, r.phis0, r.fmask, r.tland, r.tsea, r.swav, ssrd, slrd, r.alb_l, r.alb_s, r.snowc, r.forog, r.sqclat, sf);
    CSTAMP(5);
    radlw_up(L, c, D.fband, tau2, stratc, lw, sf.tsfc, slrd, sf.slru[3], slr, olr, dfabs);
    CSTAMP(6);
    for (int k = 1; k <= KX; ++k) tt_rlw[k] = dfabs[k] * c.rps * L.grdscp[k];
    xb[0] = sf.ustr[3]; xb[1] = sf.vstr[3]; xb[2] = sf.shf[3]; xb[3] = sf.evap[3];
    park_ready();
    if (want_diag) {
        double *dg = D.diag;
        dg[(size_t)D_TS * GR + p] = sf.tsfc; dg[(size_t)D_TSKIN * GR + p] = sf.tskin; dg[(size_t)D_SSRD * GR + p] = ssrd;
        dg[(size_t)D_SLRD * GR + p] = slrd; dg[(size_t)D_OLR * GR + p] = olr; dg[(size_t)D_SHF * GR + p] = sf.shf[3];
        dg[(size_t)D_EVAP * GR + p] = sf.evap[3]; dg[(size_t)D_USTR * GR + p] = sf.ustr[3]; dg[(size_t)D_VSTR * GR + p] = sf.vstr[3];
        dg[(size_t)D_SLR * GR + p] = slr; dg[(size_t)D_HFLUXN1 * GR + p] = sf.hfluxn[1]; dg[(size_t)D_HFLUXN2 * GR + p] = sf.hfluxn[2];
        dg[(size_t)D_T0 * GR + p] = sf.t0; dg[(size_t)D_Q0 * GR + p] = sf.q0;
    }
}

// phypar 3.x sums and 4 (:172-196): ttend = ttend + tt_rsw + tt_rlw; vertical diffusion / shallow convection and the surface
// fluxes as tendencies of the lowest level; the final tendencies go to `tend`.
__device__ __forceinline__ void chain_pbl_and_store(const PhysLev &L, const Column &c, int p, int icnv, double *park, int lane,
                                                    double *__restrict__ tend, int off_u, int off_v, int off_t, int off_q, double u_dyn,
                                                    double v_dyn, const double (&pt)[NLP], const double (&pq)[NLP])
{
    LA tt = park_array(park, P_TT, lane), qt = park_array(park, P_QT, lane), tt_rsw = park_array(park, P_RSW, lane);
    LA tt_rlw = park_array(park, P_RLW, lane), xb = park_array(park, P_XB, lane);
    const double ut = 0.0 + xb[0] * c.rps * L.grdsig[KX];        // ut_pbl(:,kx) = utenvd (0) + stress term (:187-190)
    const double vt = 0.0 + xb[1] * c.rps * L.grdsig[KX];
    tend[(size_t)(off_u + KX - 1) * GR + p] = u_dyn + ut;
    tend[(size_t)(off_v + KX - 1) * GR + p] = v_dyn + vt;
    for (int k = 1; k <= KX; ++k) {
        double t_pbl = pt[k], q_pbl = pq[k];
        if (k == KX) { t_pbl = t_pbl + xb[2] * c.rps * L.grdscp[KX]; q_pbl = q_pbl + xb[3] * c.rps * L.grdsig[KX]; }
        const double t_rad = tt[k] + tt_rsw[k] + tt_rlw[k];
        tend[(size_t)(off_t + k - 1) * GR + p] = t_rad + t_pbl;
        tend[(size_t)(off_q + k - 1) * GR + p] = qt[k] + q_pbl;
    }
}

// The same finish for the three-wavefront kernel, whose moist wavefront did NOT add its tendencies into the park (the dynamics wavefront
// was still filling it): ttend = ((dyn + cnv) + lsc), then + rsw + rlw, then + pbl -- the order of chain_moist followed by
// chain_pbl_and_store, hence the same bits.
__device__ __forceinline__ void finish_and_store(const PhysLev &L, const Column &c, int p, double *park, int lane, double *__restrict__ tend,
                                                 int off_u, int off_v, int off_t, int off_q, const double (&a1)[NLP], const double (&a2)[NLP],
                                                 const double (&b1)[NLP], const double (&b2)[NLP], const double (&pt)[NLP], const double (&pq)[NLP])
{
    LA tt = park_array(park, P_TT, lane), qt = park_array(park, P_QT, lane), tt_rsw = park_array(park, P_RSW, lane);
    LA tt_rlw = park_array(park, P_RLW, lane), xb = park_array(park, P_XB, lane), uv = park_array(park, P_UV, lane);
    const double ut = 0.0 + xb[0] * c.rps * L.grdsig[KX];
    const double vt = 0.0 + xb[1] * c.rps * L.grdsig[KX];
    tend[(size_t)(off_u + KX - 1) * GR + p] = uv[0] + ut;
    tend[(size_t)(off_v + KX - 1) * GR + p] = uv[1] + vt;
    for (int k = 1; k <= KX; ++k) {
        double t_pbl = pt[k], q_pbl = pq[k];
        if (k == KX) { t_pbl = t_pbl + xb[2] * c.rps * L.grdscp[KX]; q_pbl = q_pbl + xb[3] * c.rps * L.grdsig[KX]; }
        const double t_moist = tt[k] + a1[k] + b1[k], q_moist = qt[k] + a2[k] + b2[k];
        const double t_rad = t_moist + tt_rsw[k] + tt_rlw[k];
        tend[(size_t)(off_t + k - 1) * GR + p] = t_rad + t_pbl;
        tend[(size_t)(off_q + k - 1) * GR + p] = q_moist + q_pbl;
    }
}

// physics_column: the whole sequence for column p in one wavefront.  The caller has put the dynamical ttend, qtend (or zeros) of
// the column into the park (P_TT, P_QT) and passes the dynamical utend(kx), vtend(kx).
__device__ __forceinline__ void physics_column(const PhysLev &L, const PhysDev &D, const PhysIn &in, double *__restrict__ tend, int lradsw,
                                               int off_u, int off_v, int off_t, int off_q, int want_diag, int p, double *park,
                                               double u_dyn, double v_dyn)
{
    const int lane = threadIdx.x & 63;
    PSTAMP(0);
    Column c;
    RadIn r;
    column_load(in, p, c);                              // one batch of loads for everything the sequence reads
    radiation_load(D, p, lradsw, park, lane, r);
    column_thermo(L, c);
    PSTAMP(1);
    int iptop, icnv;
    double precnv, precls;
    chain_moist(L, D, c, p, want_diag, park, lane, iptop, icnv, precnv, precls);
    PSTAMP(3);
    chain_radiation(L, D, c, r, p, lradsw, want_diag, park, lane, precnv, precls, iptop);
    PSTAMP(7);
    double pt[NLP], pq[NLP];
    vdifsc(L, c, icnv, pt, pq);
    PSTAMP(8);
    chain_pbl_and_store(L, c, p, icnv, park, lane, tend, off_u, off_v, off_t, off_q, u_dyn, v_dyn, pt, pq);
    PSTAMP(9);
}

}  // namespace
}  // namespace smlphys

struct sml_phys {
    smlphys::PhysLev lev;
    smlphys::PhysDev dev{};
    std::vector<void *> allocs;
    double *surf = nullptr;      // 10 x GR: fmask phis0 tland tsea swav alb_l alb_s albsfc snowc forog
    double *zonal = nullptr;     // 6 x 48: fsol ozone ozupp zenit stratz sqclat
    double *fordate = nullptr;   // 6 x GR: fmask_s alb0 snowd_am sice_am (inputs of fordate) | corh_t corh_q (its two grid fields)
    bool fordate_albedo = false; // fordate recomputes snowc / alb_l / alb_s / albsfc from alb0, snowd_am, sice_am
    double clat[48], slat[48];
};

