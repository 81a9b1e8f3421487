// SPEEDY's adiabatic time step on gfx950: the spectral-space step with the model state resident in HBM.
//
// Replaces, for the hybrid window between iogrid(30) and iogrid(31) (src/ppo_iogrid.f90:497-601):
//   src/ini_indyns.f90, src/ini_impint.f90 (+ inv/ludcmp/lubksb of src/spe_matinv.f90)   -> host tables, uploaded once per (dt, alph)
//   src/dyn_grtend.f90 (without the phypar call :222-225; column physics is out of scope)  -> k_gridtend + 2 transform launches
//   src/dyn_sptend.f90, dyn_geop.f90, dyn_implic.f90, dyn_step.f90 (hordif, timint)        -> k_spectral
//   src/ini_stepone.f90 + the step loop of src/dyn_stloop.f90:28-43                        -> sml_dyn_window
//
// The reference runs 164 single-field transforms and ~60 small array loops per step on one core.  Here one step is FOUR
// launches on the caller's stream and nothing returns to the host:
//   k_grid      (spectral.hip) all 50 inverse transforms of grtend in one launch; uvspec of the 8 levels and grad(ps) are
//               evaluated while the spectral fields are staged into LDS (sml_spectral_grid_derived), nothing is materialised
//   k_gridtend  one thread per grid point walks the 8-level column: means, sigma-dot, u/v/T/q tendencies, the flux
//               products -> ONE 73-field grid batch
//   k_spec      (spectral.hip) all 73 forward transforms, per-field 1/cos pre-scaling
//   k_spectral  one thread per spectral coefficient (m,n,re/im) walks its 8-level column: vds of the flux pairs, the
//               Laplacian of kinetic energy, sptend (+geop), the semi-implicit correction (three 8x8 mat-vecs with the
//               xd / xj(l) / xc tables), horizontal diffusion, truncation and the Robert-Asselin-Williams leapfrog update
//               of both time levels in place.
// Everything between the transforms is pointwise in the horizontal, so no intermediate array other than the two
// transform batches is ever written to HBM.  Every expression keeps the reference's operation order (no FMA
// contraction): given identical transform outputs the step is bit-identical to the Fortran.
//
// State layout (device, caller-owned): state[2][33][32][62] doubles -- time level j, then 33 spectral fields
// (vor(8) | div(8) | t(8) | tr(8) | ps), each (mx2=62, nx=32) as Fortran stores complex (mx,nx).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "span.h"
#include "physics_dev.h"

namespace {

constexpr int IX = 96, IY = 24, IL = 48, NX = 32, MX = 31, MX2 = 62, KX = 8, KXP = 9, LMAX = 61;
constexpr int SP = MX2 * NX;       // 1984
constexpr int GR = IX * IL;        // 4608
constexpr int NSTATE = 33;         // fields per time level
constexpr int F_VOR = 0, F_DIV = 8, F_T = 16, F_TR = 24, F_PS = 32;
constexpr int NB_SPEC = 50;        // inverse batch: the 32 3-D state fields, ucos(8), vcos(8), d(ps)/dx, d(ps)/dy
constexpr int NB_GRID = 73;        // forward batch, see k_gridtend
constexpr int NB_PHYS = 27;        // phypar's inverse set of time level 1 (phy_phypar.f90:54-66) without the 14 wind levels that no
                                   // parametrisation reads: u(kx) v(kx) t(8) q(8) phi(8) ps
constexpr int NB_ALL = NB_SPEC + NB_PHYS;

// src/mod_dyncon0.f90:9-45, src/mod_dyncon1.f90:12-27 (all reals are promoted to 8 bytes, src/makefile:6,12)
constexpr double REARTH = 6.371e+6, OMEGA = 7.292e-05, GRAV = 9.81, GAMMA = 6.0, HSCALE = 7.5, HSHUM = 2.5;
constexpr double THD = 2.4, THDD = 2.4, THDS = 12.0, TDRS = 24.0 * 30.0;
constexpr double AKAP = 2. / 7.;
constexpr double RGAS = AKAP * 1004.;

struct LevelTables {               // per-level constants, passed to the kernels by value (0.9 KB)
    double dhs[KX], dhsr[KX], fsgr[KX], xgeop1[KX], xgeop2[KX], tcorv[KX], qcorv[KX];       // indyns
    double tref[KX], tref1[KX], tref2[KX], tref3[KX], dhsx[KX];                              // impint
    double corf[KX];               // the m=0 geopotential correction factor of dyn_geop.f90:30-35 (levels 2..kx-1)
    double akap, sdrag;
};

struct HostDyn {                   // indyns
    double hsg[KXP], dhs[KX], fsg[KX], dhsr[KX], fsgr[KX], coriol[IL], xgeop1[KX], xgeop2[KX], tcorv[KX], qcorv[KX];
    double dmp[NX][MX], dmpd[NX][MX], dmps[NX][MX];
};

struct ImpSlot {                   // impint(dt, alph)
    double dt = 0, alph = 0;
    double dmp1[NX][MX], dmp1d[NX][MX], dmp1s[NX][MX], elz[NX][MX];
    double tref[KX], tref1[KX], tref2[KX], tref3[KX], dhsx[KX];
    double xc[KX][KX], xd[KX][KX], xj[LMAX][KX][KX];       // Fortran x(k,k1) -> x[k1][k]
    double *d_h = nullptr;         // device: dmp1 | dmp1d | dmp1s | elz   (4 x 992)
    double *d_x = nullptr;         // device: xd(64) | xc(64) | xj(61*64) | level tables (13 x 8, rows LV_*)
    LevelTables lv;
};

struct DevHoriz {                  // device tables that do not depend on dt
    const double *coriol;                                   // [48]
    const double *dmp, *dmpd, *dmps;                        // [32][31]
    const double *el2, *trfilt, *gradx, *gradym, *gradyp, *uvdx, *uvdym, *uvdyp, *vddym, *vddyp;
};

// ---------------------------------------------------------------- host tables
void build_indyns(HostDyn &h, const double *sia)
{   // src/ini_indyns.f90
    static const double hsg8[KXP] = {0.000, 0.050, 0.140, 0.260, 0.420, 0.600, 0.770, 0.900, 1.000};
    for (int k = 0; k < KXP; ++k) h.hsg[k] = hsg8[k];
    for (int k = 0; k < KX; ++k) {
        h.dhs[k] = h.hsg[k + 1] - h.hsg[k];
        h.fsg[k] = 0.5 * (h.hsg[k + 1] + h.hsg[k]);
    }
    for (int k = 0; k < KX; ++k) {
        h.dhsr[k] = 0.5 / h.dhs[k];
        h.fsgr[k] = AKAP / (2. * h.fsg[k]);
    }
    for (int j = 0; j < IY; ++j) {          // southern row j has sin(lat) = -sia(j)
        h.coriol[j] = 2. * OMEGA * (-sia[j]);
        h.coriol[IL - 1 - j] = 2. * OMEGA * sia[j];
    }
    h.xgeop2[0] = 0.0;
    for (int k = 0; k < KX; ++k) {
        h.xgeop1[k] = RGAS * std::log(h.hsg[k + 1] / h.fsg[k]);
        if (k + 1 < KX) h.xgeop2[k + 1] = RGAS * std::log(h.fsg[k + 1] / h.hsg[k + 1]);
    }
    const double hdiff = 1. / (THD * 3600.), hdifd = 1. / (THDD * 3600.), hdifs = 1. / (THDS * 3600.);
    const double rlap = 1. / (double)(30 * 31);
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m) {
            const double twn = (double)(m + n);
            const double elap = twn * (twn + 1.) * rlap;
            const double elapn = elap * elap * elap * elap;        // elap**npowhd, npowhd = 4
            h.dmp[n][m] = hdiff * elapn;
            h.dmpd[n][m] = hdifd * elapn;
            h.dmps[n][m] = hdifs * elap;
        }
    const double rgam = RGAS * GAMMA / (1000. * GRAV), qexp = HSCALE / HSHUM;
    h.tcorv[0] = 0.; h.qcorv[0] = 0.; h.qcorv[1] = 0.;
    for (int k = 1; k < KX; ++k) {
        h.tcorv[k] = std::pow(h.fsg[k], rgam);
        if (k > 1) h.qcorv[k] = std::pow(h.fsg[k], qexp);
    }
}

// inverse of an n x n column-major matrix by Crout LU with implicit row scaling and back-substitution of the unit
// vectors -- the algorithm of inv/ludcmp/lubksb (src/spe_matinv.f90), so the xj tables round like the reference's.
void invert(double *a, double *inv, int n)
{
    std::vector<int> piv(n);
    std::vector<double> scale(n);
    auto A = [&](int i, int j) -> double & { return a[(size_t)j * n + i]; };
    for (int i = 0; i < n; ++i) {
        double big = 0.;
        for (int j = 0; j < n; ++j) big = std::max(big, std::fabs(A(i, j)));
        scale[i] = 1. / big;
    }
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < j; ++i) {
            double s = A(i, j);
            for (int k = 0; k < i; ++k) s = s - A(i, k) * A(k, j);
            A(i, j) = s;
        }
        double big = 0.;
        int imax = j;
        for (int i = j; i < n; ++i) {
            double s = A(i, j);
            for (int k = 0; k < j; ++k) s = s - A(i, k) * A(k, j);
            A(i, j) = s;
            const double fom = scale[i] * std::fabs(s);
            if (fom >= big) { big = fom; imax = i; }
        }
        if (imax != j) {
            for (int k = 0; k < n; ++k) std::swap(A(imax, k), A(j, k));
            scale[imax] = scale[j];
        }
        piv[j] = imax;
        if (j != n - 1) {
            const double r = 1. / A(j, j);
            for (int i = j + 1; i < n; ++i) A(i, j) = A(i, j) * r;
        }
    }
    for (int c = 0; c < n; ++c) {
        double *b = inv + (size_t)c * n;
        for (int i = 0; i < n; ++i) b[i] = i == c ? 1. : 0.;
        int first = -1;
        for (int i = 0; i < n; ++i) {
            const int ip = piv[i];
            double s = b[ip];
            b[ip] = b[i];
            if (first >= 0) { for (int j = first; j < i; ++j) s = s - A(i, j) * b[j]; }
            else if (s != 0.) first = i;
            b[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = b[i];
            for (int j = i + 1; j < n; ++j) s = s - A(i, j) * b[j];
            b[i] = s / A(i, i);
        }
    }
}

void build_impint(ImpSlot &s, const HostDyn &h, double dt, double alph)
{   // src/ini_impint.f90
    s.dt = dt; s.alph = alph;
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m) {
            s.dmp1[n][m] = 1. / (1. + h.dmp[n][m] * dt);
            s.dmp1d[n][m] = 1. / (1. + h.dmpd[n][m] * dt);
            s.dmp1s[n][m] = 1. / (1. + h.dmps[n][m] * dt);
        }
    const double rgam = RGAS * GAMMA / (1000. * GRAV);
    for (int k = 0; k < KX; ++k) {
        s.tref[k] = 288. * std::pow(std::max(0.2, h.fsg[k]), rgam);
        s.tref1[k] = RGAS * s.tref[k];
        s.tref2[k] = AKAP * s.tref[k];
        s.tref3[k] = h.fsgr[k] * s.tref[k];
    }
    const double xi = dt * alph, xxi = xi / (REARTH * REARTH);
    for (int k = 0; k < KX; ++k) s.dhsx[k] = xi * h.dhs[k];
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m) { const int l = m + n; s.elz[n][m] = (double)l * (double)(l + 1) * xxi; }
    // index convention below: M[k1][k] = Fortran M(k,k1)
    double ya[KX][KX], xa[KX][KX] = {}, xb[KX][KX] = {}, xcl[KX][KX], xe[KX][KX], dsum[KX];
    for (int k = 0; k < KX; ++k) for (int k1 = 0; k1 < KX; ++k1) ya[k1][k] = -AKAP * s.tref[k] * h.dhs[k1];
    for (int k = 1; k < KX; ++k) xa[k - 1][k] = 0.5 * (AKAP * s.tref[k] / h.fsg[k] - (s.tref[k] - s.tref[k - 1]) / h.dhs[k]);
    for (int k = 0; k < KX - 1; ++k) xa[k][k] = 0.5 * (AKAP * s.tref[k] / h.fsg[k] - (s.tref[k + 1] - s.tref[k]) / h.dhs[k]);
    dsum[0] = h.dhs[0];
    for (int k = 1; k < KX; ++k) dsum[k] = dsum[k - 1] + h.dhs[k];
    for (int k = 0; k < KX - 1; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            xb[k1][k] = h.dhs[k1] * dsum[k];
            if (k1 <= k) xb[k1][k] = xb[k1][k] - h.dhs[k1];
        }
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            double v = ya[k1][k];
            for (int k2 = 0; k2 < KX - 1; ++k2) v = v + xa[k2][k] * xb[k1][k2];
            xcl[k1][k] = v;
        }
    memset(s.xd, 0, sizeof s.xd);
    for (int k = 0; k < KX; ++k) {
        for (int k1 = k + 1; k1 < KX; ++k1) s.xd[k1][k] = RGAS * std::log(h.hsg[k1 + 1] / h.hsg[k1]);
        s.xd[k][k] = RGAS * std::log(h.hsg[k + 1] / h.fsg[k]);
    }
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            double v = 0.;
            for (int k2 = 0; k2 < KX; ++k2) v = v + s.xd[k2][k] * xcl[k1][k2];
            xe[k1][k] = v;
        }
    for (int l = 1; l <= LMAX; ++l) {
        double xf[KX][KX];
        const double xxx = ((double)l * (double)(l + 1)) / (REARTH * REARTH);
        for (int k = 0; k < KX; ++k)
            for (int k1 = 0; k1 < KX; ++k1) xf[k1][k] = xi * xi * xxx * (RGAS * s.tref[k] * h.dhs[k1] - xe[k1][k]);
        for (int k = 0; k < KX; ++k) xf[k][k] = xf[k][k] + 1.;
        invert(&xf[0][0], &s.xj[l - 1][0][0], KX);
    }
    for (int k = 0; k < KX; ++k) for (int k1 = 0; k1 < KX; ++k1) s.xc[k1][k] = xcl[k1][k] * xi;
    LevelTables &lv = s.lv;
    for (int k = 0; k < KX; ++k) {
        lv.dhs[k] = h.dhs[k]; lv.dhsr[k] = h.dhsr[k]; lv.fsgr[k] = h.fsgr[k]; lv.xgeop1[k] = h.xgeop1[k]; lv.xgeop2[k] = h.xgeop2[k];
        lv.tcorv[k] = h.tcorv[k]; lv.qcorv[k] = h.qcorv[k];
        lv.tref[k] = s.tref[k]; lv.tref1[k] = s.tref1[k]; lv.tref2[k] = s.tref2[k]; lv.tref3[k] = s.tref3[k]; lv.dhsx[k] = s.dhsx[k];
        lv.corf[k] = 0.;
    }
    for (int k = 1; k < KX - 1; ++k)        // dyn_geop.f90:30-35, 1-based levels 2..kx-1
        lv.corf[k] = h.xgeop1[k] * 0.5 * std::log(h.hsg[k + 1] / h.fsg[k]) / std::log(h.fsg[k + 1] / h.fsg[k - 1]);
    lv.akap = AKAP;
    lv.sdrag = 1. / (TDRS * 3600.);
}

// ---------------------------------------------------------------- device code
// (i g z): (re,im) -> (-g im, g re)
__device__ __forceinline__ double irot(const double *a, int base, int c, double g)
{
    return (c & 1) ? g * a[base + c - 1] : -g * a[base + c + 1];
}

// the 3-point-in-n stencil shared by uvspec (:351-387) and vds (:307-349) of src/spe_spectral.f90:
//   A = ym*P(n-1) - yp*P(n+1) + i x Q ;  B = -ym*Q(n-1) + yp*Q(n+1) + i x P          (P, Q: one field each, [32][62])
__device__ __forceinline__ void stencil(const double *__restrict__ P, const double *__restrict__ Q, int n, int c, double gx,
                                        double ym, double yp, double &a, double &b)
{
    const int row = n * MX2;
    if (n == 0) {
        a = irot(Q, row, c, gx) - yp * P[row + MX2 + c];
        b = irot(P, row, c, gx) + yp * Q[row + MX2 + c];
    } else if (n == NX - 1) {
        a = ym * P[row - MX2 + c];
        b = -ym * Q[row - MX2 + c];
    } else {
        a = ym * P[row - MX2 + c] - yp * P[row + MX2 + c] + irot(Q, row, c, gx);
        b = -ym * Q[row - MX2 + c] + yp * Q[row + MX2 + c] + irot(P, row, c, gx);
    }
}

// k_gridtend: grid-point tendencies (dyn_grtend.f90:80-216 and the flux products of :237-276), one thread per grid point.
// Input batch G[50][48][96]: vor div t tr (8 levels each) | u(8) | v(8) | d(ps)/dx | d(ps)/dy.  Output batch O[73][48][96]:
//   0..7 utend   8..15 vtend   16..23 -u*T'   24..31 -v*T'   32..39 -u*q   40..47 -v*q      (forward transform pre-scaled
//   by 1/cos: the specx halves of vdspec(.,.,2))      48..55 (u^2+v^2)/2   56..63 ttend   64..71 qtend   72 ps tendency
__global__ __launch_bounds__(64) void k_gridtend(DevHoriz H, LevelTables L, const double *__restrict__ G, double *__restrict__ O)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= GR) return;
    const int j = p / IX;
    double ug[KX], vg[KX], tg[KX], vorg[KX], divg[KX], trg[KX], puv[KX], sigdt[KXP], sigm[KXP];
    const double cor = H.coriol[j];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        vorg[k] = G[(size_t)(F_VOR + k) * GR + p] + cor;
        divg[k] = G[(size_t)(F_DIV + k) * GR + p];
        tg[k] = G[(size_t)(F_T + k) * GR + p];
        trg[k] = G[(size_t)(F_TR + k) * GR + p];
        ug[k] = G[(size_t)(32 + k) * GR + p];
        vg[k] = G[(size_t)(40 + k) * GR + p];
    }
    double px = G[(size_t)48 * GR + p], py = G[(size_t)49 * GR + p];
    double umean = 0.0, vmean = 0.0, dmean = 0.0;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        umean = umean + ug[k] * L.dhs[k];
        vmean = vmean + vg[k] * L.dhs[k];
        dmean = dmean + divg[k] * L.dhs[k];
    }
    O[(size_t)72 * GR + p] = -umean * px - vmean * py;
    sigdt[0] = 0.0; sigm[0] = 0.0;
#pragma unroll
    for (int k = 0; k < KX; ++k) puv[k] = (ug[k] - umean) * px + (vg[k] - vmean) * py;
#pragma unroll
    for (int k = 0; k < KX; ++k) {          // the reference's loop runs to kx and so overwrites the zero it put at kxp
        sigdt[k + 1] = sigdt[k] - L.dhs[k] * (puv[k] + divg[k] - dmean);
        sigm[k + 1] = sigm[k] - L.dhs[k] * puv[k];
    }
    double tgg[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) tgg[k] = tg[k] - L.tref[k];
    px = RGAS * px;
    py = RGAS * py;
    double tmp[KXP];
    tmp[0] = 0.0; tmp[KX] = 0.0;
    // zonal wind
#pragma unroll
    for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (ug[k] - ug[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k) O[(size_t)k * GR + p] = vg[k] * vorg[k] - tgg[k] * px - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
    // meridional wind
#pragma unroll
    for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (vg[k] - vg[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k) O[(size_t)(8 + k) * GR + p] = -ug[k] * vorg[k] - tgg[k] * py - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
    // temperature
#pragma unroll
    for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (tgg[k] - tgg[k - 1]) + sigm[k] * (L.tref[k] - L.tref[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k)
        O[(size_t)(56 + k) * GR + p] = tgg[k] * divg[k] - (tmp[k + 1] + tmp[k]) * L.dhsr[k] + L.fsgr[k] * tgg[k] * (sigdt[k + 1] + sigdt[k])
                                       + L.tref3[k] * (sigm[k + 1] + sigm[k]) + L.akap * (tg[k] * puv[k] - tgg[k] * dmean);
    // tracer (specific humidity): no vertical advection across the two uppermost interfaces (:196-203)
#pragma unroll
    for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (trg[k] - trg[k - 1]);
    tmp[1] = 0.; tmp[2] = 0.;
#pragma unroll
    for (int k = 0; k < KX; ++k) O[(size_t)(64 + k) * GR + p] = trg[k] * divg[k] - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
    // flux products (:241-246, :262-267)
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        O[(size_t)(48 + k) * GR + p] = 0.5 * (ug[k] * ug[k] + vg[k] * vg[k]);
        O[(size_t)(16 + k) * GR + p] = -ug[k] * tgg[k];
        O[(size_t)(24 + k) * GR + p] = -vg[k] * tgg[k];
        O[(size_t)(32 + k) * GR + p] = -ug[k] * trg[k];
        O[(size_t)(40 + k) * GR + p] = -vg[k] * trg[k];
    }
}

// grtend's grid-point part (src/dyn_grtend.f90:101-216) for column p: the dynamical tendencies of vorticity / divergence (as flux
// components) and the flux products go straight to the forward batch O; ttend and qtend go to the workgroup's park (P_TT, P_QT), where
// phypar's accumulators start from them; utend(kx), vtend(kx) are returned (the physics adds the surface stress to them).
__device__ __forceinline__ void gridpoint_dynamics(const DevHoriz &H, const LevelTables &L, const double *__restrict__ G, double *__restrict__ O, int p,
                                                   double *park, int lane, double &u_dyn_out, double &v_dyn_out, int32_t *guard = nullptr)
{
        const int j = p / IX;
        smlphys::LA tt = smlphys::park_array(park, smlphys::P_TT, lane), qt = smlphys::park_array(park, smlphys::P_QT, lane);   // 1-based levels
        double ug[KX], vg[KX], tg[KX], vorg[KX], divg[KX], trg[KX], puv[KX], sigdt[KXP], sigm[KXP];
        const double cor = H.coriol[j];
    #pragma unroll
        for (int k = 0; k < KX; ++k) {
            vorg[k] = G[(size_t)(F_VOR + k) * GR + p] + cor;
            divg[k] = G[(size_t)(F_DIV + k) * GR + p];
            tg[k] = G[(size_t)(F_T + k) * GR + p];
            trg[k] = G[(size_t)(F_TR + k) * GR + p];
            ug[k] = G[(size_t)(32 + k) * GR + p];
            vg[k] = G[(size_t)(40 + k) * GR + p];
        }
        double px = G[(size_t)48 * GR + p], py = G[(size_t)49 * GR + p];
        if (guard) {
            // iogrid(30)'s range guard (see k_range_guard) on the values this wavefront has just loaded: the first time step of a hybrid
            // window checks T, q, u, v of time level 1 here instead of in a launch of its own behind the step
            bool ok = true;
    #pragma unroll
            for (int k = 0; k < KX; ++k)
                ok = ok && tg[k] >= 160.0 && tg[k] <= 330.0 && trg[k] >= -6.0 && trg[k] <= 30.0 && ug[k] >= -150.0 && ug[k] <= 150.0 && vg[k] >= -120.0 &&
                     vg[k] <= 120.0;                                      // (a NaN fails every comparison)
            if (!ok) *guard = 0;
        }
        double umean = 0.0, vmean = 0.0, dmean = 0.0;
    #pragma unroll
        for (int k = 0; k < KX; ++k) {
            umean = umean + ug[k] * L.dhs[k];
            vmean = vmean + vg[k] * L.dhs[k];
            dmean = dmean + divg[k] * L.dhs[k];
        }
        O[(size_t)72 * GR + p] = -umean * px - vmean * py;
        sigdt[0] = 0.0; sigm[0] = 0.0;
    #pragma unroll
        for (int k = 0; k < KX; ++k) puv[k] = (ug[k] - umean) * px + (vg[k] - vmean) * py;
    #pragma unroll
        for (int k = 0; k < KX; ++k) {          // the reference's loop runs to kx and so overwrites the zero it put at kxp
            sigdt[k + 1] = sigdt[k] - L.dhs[k] * (puv[k] + divg[k] - dmean);
            sigm[k + 1] = sigm[k] - L.dhs[k] * puv[k];
        }
        double tgg[KX];
    #pragma unroll
        for (int k = 0; k < KX; ++k) tgg[k] = tg[k] - L.tref[k];
        px = RGAS * px;
        py = RGAS * py;
        double tmp[KXP];
        tmp[0] = 0.0; tmp[KX] = 0.0;
        // zonal wind
    #pragma unroll
        for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (ug[k] - ug[k - 1]);
    #pragma unroll
        for (int k = 0; k < KX - 1; ++k) O[(size_t)k * GR + p] = vg[k] * vorg[k] - tgg[k] * px - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
        const double u_dyn = vg[KX - 1] * vorg[KX - 1] - tgg[KX - 1] * px - (tmp[KX] + tmp[KX - 1]) * L.dhsr[KX - 1];
        // meridional wind
    #pragma unroll
        for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (vg[k] - vg[k - 1]);
    #pragma unroll
        for (int k = 0; k < KX - 1; ++k) O[(size_t)(8 + k) * GR + p] = -ug[k] * vorg[k] - tgg[k] * py - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
        const double v_dyn = -ug[KX - 1] * vorg[KX - 1] - tgg[KX - 1] * py - (tmp[KX] + tmp[KX - 1]) * L.dhsr[KX - 1];
        // temperature
    #pragma unroll
        for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (tgg[k] - tgg[k - 1]) + sigm[k] * (L.tref[k] - L.tref[k - 1]);
    #pragma unroll
        for (int k = 0; k < KX; ++k)
            tt[k + 1] = tgg[k] * divg[k] - (tmp[k + 1] + tmp[k]) * L.dhsr[k] + L.fsgr[k] * tgg[k] * (sigdt[k + 1] + sigdt[k])
                                           + L.tref3[k] * (sigm[k + 1] + sigm[k]) + L.akap * (tg[k] * puv[k] - tgg[k] * dmean);
        // tracer (specific humidity): no vertical advection across the two uppermost interfaces (:196-203)
    #pragma unroll
        for (int k = 1; k < KX; ++k) tmp[k] = sigdt[k] * (trg[k] - trg[k - 1]);
        tmp[1] = 0.; tmp[2] = 0.;
    #pragma unroll
        for (int k = 0; k < KX; ++k) qt[k + 1] = trg[k] * divg[k] - (tmp[k + 1] + tmp[k]) * L.dhsr[k];
        // flux products (:241-246, :262-267)
    #pragma unroll
        for (int k = 0; k < KX; ++k) {
            O[(size_t)(48 + k) * GR + p] = 0.5 * (ug[k] * ug[k] + vg[k] * vg[k]);
            O[(size_t)(16 + k) * GR + p] = -ug[k] * tgg[k];
            O[(size_t)(24 + k) * GR + p] = -vg[k] * tgg[k];
            O[(size_t)(32 + k) * GR + p] = -ug[k] * trg[k];
            O[(size_t)(40 + k) * GR + p] = -vg[k] * trg[k];
        }
    u_dyn_out = u_dyn;
    v_dyn_out = v_dyn;
}

// k_gridtend_physics: grtend's grid-point part (dyn_grtend.f90:80-216) and phypar (:222-225, phy_phypar.f90:80-230) in ONE launch
// of 72 workgroups x 2 wavefronts for 64 columns each:
//   wave 0: the dynamical tendencies (k_gridtend's body; ttend, qtend, utend(kx), vtend(kx) go straight into the physics'
//           accumulators, no trip through memory), convection and condensation, vertical diffusion, the final sums and stores;
//   wave 1: clouds, short- and long-wave radiation, surface fluxes.
// The two chains meet once, through the LDS park (smlphys::P_*): the final sums need the heating rates and surface fluxes of wave 1
// (one barrier).  On short-wave steps cloud() needs the precipitation and the convection top, which wave 1 re-derives.  Every
// sum keeps the reference's order (see physics_dev.h), so the result equals the one-wave sequence bit for bit
// (sml_dyn_select_physics_form(0) runs that for comparison).  One wave for both chains took 19.9 us per launch.
// PG: the grids of time level 1 the parametrisations read (smlphys::PhysIn).
__global__ __launch_bounds__(128) void k_gridtend_physics(DevHoriz H, LevelTables L, const double *__restrict__ G, double *__restrict__ O,
                                                           smlphys::PhysLev PL, smlphys::PhysDev PD, smlphys::PhysIn PG, int lradsw, int want_diag)
{
    SML_SPAN(2);
    __shared__ double park[smlphys::PARK_DOUBLES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // uniform: the two chains are scalar branches
    const int p = blockIdx.x * 64 + lane;                                     // the grid is exactly GR / 64 workgroups
    CSTAMP(0);
    if (wave == 0) {
        smlphys::Column c;
        smlphys::column_load(PG, p, c);                  // issued with the loads of the dynamics below: one exposed round trip
        double u_dyn, v_dyn;
        gridpoint_dynamics(H, L, G, O, p, park, lane, u_dyn, v_dyn);
        CSTAMP(1);
            smlphys::column_thermo(PL, c);
        CSTAMP(2);
        int iptop, icnv;
        double precnv, precls;
        smlphys::chain_moist(PL, PD, c, p, want_diag, park, lane, iptop, icnv, precnv, precls);
        CSTAMP(3);
        double pt[smlphys::NLP], pq[smlphys::NLP];
        smlphys::vdifsc(PL, c, icnv, pt, pq);
        CSTAMP(4);
        __syncthreads();                                                      // the radiation chain has left its results in the park
        CSTAMP(5);
        smlphys::chain_pbl_and_store(PL, c, p, icnv, park, lane, O, 0, 8, 56, 64, u_dyn, v_dyn, pt, pq);
        CSTAMP(6);
    } else {
        smlphys::Column c;
        smlphys::RadIn r;
        smlphys::column_load(PG, p, c);                  // everything this chain reads, in one batch
        smlphys::radiation_load(PD, p, lradsw, park, lane, r);
        smlphys::column_thermo(PL, c);
        CSTAMP(1);
        double precnv = 0., precls = 0.;
        int iptop = 0;
        if (lradsw) {
            // cloud() needs the precipitation and the convection top: this wave evaluates convmf / lscond for itself (same inputs,
            // same code, same bits) instead of waiting 7 us for wave 0 to get there behind the grid-point dynamics
            double cbmf, s1[smlphys::NLP], s2[smlphys::NLP];
            smlphys::convmf(PL, c, iptop, cbmf, precnv, s1, s2);
            smlphys::lscond(PL, c, iptop, precls, s1, s2);
        }
        CSTAMP(2);
        smlphys::chain_radiation(PL, PD, c, r, p, lradsw, want_diag, park, lane, precnv, precls, iptop);
        CSTAMP(7);
        __syncthreads();
    }
}

// THREE wavefronts per 64 columns (round 4, the default since): phase stamps of the two-wavefront kernel (profiles/micro/
// physics_wave_stamps.py) put 4.6-5.5 of wave 0's 10 us into the loads and arithmetic of the grid-point dynamics -- 77 fields per
// column through one wavefront's miss queue -- before its moist chain (3.1 us) could even start, while the radiation wavefront of a step
// without short-wave radiation was done after 4 us.  Here the dynamics have a wavefront of their own:
//   wave 0  grid-point dynamics: 50 grids in, the flux fields out, ttend / qtend / utend(kx) / vtend(kx) into the park
//   wave 1  column state (25 values), convection, condensation, vertical diffusion; after the barrier the column's sums and stores
//   wave 2  radiation and surface fluxes as before
// One workgroup barrier, raw (s_barrier behind an LDS-only wait): every wavefront reaches it when its park entries are written, none
// waits for anybody's global stores.  Each sum keeps the reference's order (finish_and_store), so the result equals the other forms bit
// for bit (tests/test_physics_gpu.py::test_fused_forms_agree_bit_for_bit).
__device__ __forceinline__ void park_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wavefront's LDS writes have landed (vmcnt / expcnt left alone)
    __builtin_amdgcn_s_barrier();
}
struct ParkBarrier { __device__ __forceinline__ void operator()() const { park_barrier(); } };

__global__ __launch_bounds__(192) void k_gridtend_physics3(DevHoriz H, LevelTables L, const double *__restrict__ G, double *__restrict__ O,
                                                            smlphys::PhysLev PL, smlphys::PhysDev PD, smlphys::PhysIn PG, int lradsw, int want_diag,
                                                            int32_t *guard)
{
    SML_SPAN(2);
    __shared__ double park[smlphys::PARK_DOUBLES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // uniform: the three chains are scalar branches
    const int p = blockIdx.x * 64 + lane;                                     // the grid is exactly GR / 64 workgroups
    CSTAMP(0);
    if (wave == 0) {
        double u_dyn, v_dyn;
        gridpoint_dynamics(H, L, G, O, p, park, lane, u_dyn, v_dyn, guard);
        smlphys::LA uv = smlphys::park_array(park, smlphys::P_UV, lane);
        uv[0] = u_dyn; uv[1] = v_dyn;
        CSTAMP(1);
        SML_SPAN_MARK();
        park_barrier();
        CSTAMP(2);
    } else if (wave == 1) {
        smlphys::Column c;
        smlphys::column_load(PG, p, c);
        smlphys::column_thermo(PL, c);
        CSTAMP(1);
        int iptop, icnv;
        double precnv, precls, a1[smlphys::NLP], a2[smlphys::NLP], b1[smlphys::NLP], b2[smlphys::NLP];
        smlphys::moist_tendencies(PL, PD, c, p, want_diag, iptop, icnv, precnv, precls, a1, a2, b1, b2);
        CSTAMP(2);
        double pt[smlphys::NLP], pq[smlphys::NLP];
        smlphys::vdifsc(PL, c, icnv, pt, pq);
        CSTAMP(3);
        SML_SPAN_MARK();
        park_barrier();                                                       // the dynamics and the radiation have left their results
        CSTAMP(4);
        smlphys::finish_and_store(PL, c, p, park, lane, O, 0, 8, 56, 64, a1, a2, b1, b2, pt, pq);
        CSTAMP(5);
    } else {
        smlphys::Column c;
        smlphys::RadIn r;
        smlphys::column_load(PG, p, c);                  // everything this chain reads, in one batch
        smlphys::radiation_load(PD, p, lradsw, park, lane, r);
        smlphys::column_thermo(PL, c);
        CSTAMP(1);
        double precnv = 0., precls = 0.;
        int iptop = 0;
        if (lradsw) {
            // cloud() needs the precipitation and the convection top: this wavefront evaluates convmf / lscond for itself (same inputs, same
            // code, same bits) rather than wait for wave 1 to hand them over through the park (they would arrive 1 us later)
            double cbmf, s1[smlphys::NLP], s2[smlphys::NLP];
            smlphys::convmf(PL, c, iptop, cbmf, precnv, s1, s2);
            smlphys::lscond(PL, c, iptop, precls, s1, s2);
        }
        CSTAMP(2);
        smlphys::chain_radiation(PL, PD, c, r, p, lradsw, want_diag, park, lane, precnv, precls, iptop, [&]() { SML_SPAN_MARK(); park_barrier(); });
        CSTAMP(7);
    }
}

// The ONE-wavefront form of k_gridtend_physics (sml_dyn_select_physics_form(2)): both chains in sequence by one wavefront per 64
// columns, 476 registers (220 of them AGPRs used as spill space) and 56 B of scratch at -O3.  Round 1 saw a kernel of this shape --
// an uncommitted intermediate of the cut into chain functions -- give non-repeatable NaNs at -O3 and dropped it for the two-wave
// form.  Rebuilt here from the committed chain functions it is repeatable and bit-identical to the two-launch and two-wave forms
// (tests/test_physics_gpu.py::test_fused_forms_agree_bit_for_bit; so is a variant without launch bounds, 128 registers + 1600 B of
// scratch, tried and removed), so the round-1 failure was in that intermediate source, not a defect of the compiler that survives
// in the tree.  It stays selectable as the subject of that test and of profiles/micro/fused_determinism.py.
__global__ __launch_bounds__(64) void k_gridtend_physics_onewave(DevHoriz H, LevelTables L, const double *__restrict__ G, double *__restrict__ O,
                                                                  smlphys::PhysLev PL, smlphys::PhysDev PD, smlphys::PhysIn PG, int lradsw, int want_diag)
{
    __shared__ double park[smlphys::PARK_DOUBLES];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 64 + lane;
        smlphys::Column c;
        smlphys::column_load(PG, p, c);                  // issued with the loads of the dynamics below: one exposed round trip
        double u_dyn, v_dyn;
        gridpoint_dynamics(H, L, G, O, p, park, lane, u_dyn, v_dyn);
            smlphys::column_thermo(PL, c);
        int iptop, icnv;
        double precnv, precls;
        smlphys::chain_moist(PL, PD, c, p, want_diag, park, lane, iptop, icnv, precnv, precls);
        double pt[smlphys::NLP], pq[smlphys::NLP];
        smlphys::vdifsc(PL, c, icnv, pt, pq);
        {   // the radiation chain, by the same wavefront (its own copy of the column, as wave 1 has it)
            smlphys::Column c2;
            smlphys::RadIn r;
            smlphys::column_load(PG, p, c2);
            smlphys::radiation_load(PD, p, lradsw, park, lane, r);
            smlphys::column_thermo(PL, c2);
            smlphys::chain_radiation(PL, PD, c2, r, p, lradsw, want_diag, park, lane, lradsw ? precnv : 0., lradsw ? precls : 0., lradsw ? iptop : 0);
        }
        smlphys::chain_pbl_and_store(PL, c, p, icnv, park, lane, O, 0, 8, 56, 64, u_dyn, v_dyn, pt, pq);
}

// iogrid(30)'s physical-range guard (src/ppo_iogrid.f90:563-577: |u| <= 150, |v| <= 120, 160 <= T <= 330, -6 <= q <= 30 on the
// grid fields obtained from the truncated spectral state) evaluated on the inverse set of the window's FIRST time step, which holds
// exactly those fields (T, q, u, v of time level 1): the hand-off then needs no 33-field inverse set of its own.  NaN trips it.
__global__ void k_range_guard(const double *__restrict__ G, int32_t *__restrict__ safe)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * GR) return;
    const int var = (t / GR) >> 3;                            // 0 T, 1 q, 2 u, 3 v: fields 16..47 of the dynamics batch
    const double v = G[(size_t)16 * GR + t];
    bool bad;
    if (var == 0) bad = v < 160.0 || v > 330.0;
    else if (var == 1) bad = v < -6.0 || v > 30.0;
    else if (var == 2) bad = v < -150.0 || v > 150.0;
    else bad = v < -120.0 || v > 120.0;
    if (bad || v != v) *safe = 0;
}

struct StepArgs {
    int j1, j2, j4, implicit, integrate;
    double dt, eps, wil;
};

// k_spectral: the whole spectral-space part of step().  One 64-thread workgroup = 8 spectral coefficients x 8 levels
// (248 workgroups): a thread owns one (coefficient, level) and the vertical couplings -- the column means and running
// sums of sptend, the hydrostatic integration of geop, the three 8x8 mat-vecs of implic -- go through LDS.  Every thread
// re-derives the running sums it needs in the reference's order (a handful of LDS-broadcast operands), so the arithmetic
// is the same sequence of operations as the Fortran column loops; a first version with one thread per coefficient
// walking its column serially took 47 us per launch (31 waves, ~250 dependent global loads each), this one 7 us.
//   FROM_FLUX = true : tendencies start from the forward batch S[73] (dyn_grtend.f90:232-288)
//   FROM_FLUX = false: tendencies are read from tend_in[33] (testing entry point sml_dyn_spectral_step)
//   tend_out (optional): the tendencies after sptend/implic/hordif, [vordt(8) | divdt(8) | tdt(8) | trdt(8) | psdt]
// phase time stamps of workgroup 100 of k_spectral (profiles/micro/spectral_step_stamps.py); compiled in with -DSML_KSPEC_STAMPS only
__device__ unsigned long long g_kspec_dbg[16];
#ifdef SML_KSPEC_STAMPS
#define KSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); \
        if (blockIdx.x == 100 && threadIdx.x == 0) g_kspec_dbg[slot] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KSTAMP(slot) do { } while (0)
#endif
constexpr int LV_DHS = 0, LV_DHSR = 1, LV_XG1 = 2, LV_XG2 = 3, LV_TCORV = 4, LV_QCORV = 5, LV_TREF = 6, LV_TREF1 = 7, LV_TREF2 = 8,
              LV_TREF3 = 9, LV_DHSX = 10, LV_CORF = 11, LV_FSGR = 12, LV_ROWS = 13;

template <bool FROM_FLUX>
__global__ __launch_bounds__(64) void k_spectral(DevHoriz H, LevelTables L, StepArgs a, const double *__restrict__ S,
                                                  const double *tend_in, double *tend_out, int stop_after_grtend,
                                                  double *__restrict__ state, const double *__restrict__ imp_h,
                                                  const double *__restrict__ imp_x, const double *__restrict__ phis,
                                                  const double *__restrict__ tcorh, const double *__restrict__ qcorh,
                                                  double *__restrict__ phi_out)
{
    SML_SPAN(4);
    KSTAMP(0);
    __shared__ double lv[LV_ROWS][KX];          // level tables, indexed by a lane-varying level
    __shared__ double xdc[2][KX * KX];          // xd, xc
    __shared__ double d4[KX][8], t4[KX][8];     // div and t of time level j4, [level][coefficient]
    __shared__ double tds[KX][8], yfs[KX][8], dvs[KX][8];
    const int tid = threadIdx.x, ci = tid & 7, k = tid >> 3;
    const int e = blockIdx.x * 8 + ci;
    const int c = e % MX2, n = e / MX2, m = c >> 1, hm = n * MX + m;
    // ---- EVERY operand of the launch in ONE unconditional batch of loads (round 4).  The version before this one looked batched in the
    // source, but its conditional loads -- the stencils' first / last total wavenumber, `two ? s2[..] : 0`, the partial second round of
    // the level tables, `ll != 0 ? ...` -- compiled into 30 exec-masked branches, each with its own s_waitcnt vmcnt(0): a chain of
    // dependent round trips that was 4.7 of the kernel's 7 us (profiles/micro/spectral_step_stamps.py).  Here every address is clamped
    // into its array, every load is issued, and what a branch used to skip is selected away afterwards (v_cndmask); the arithmetic keeps
    // the reference's expressions (a term that is absent at n = 0 or n = 31 enters as a product with zero).
    const double *lvg = imp_x + 128 + LMAX * 64;
    const int lv_second = tid + 64 < LV_ROWS * KX ? tid + 64 : LV_ROWS * KX - 1;
    const double lv_a = lvg[tid], lv_b = lvg[lv_second], xd_v = imp_x[tid], xc_v = imp_x[64 + tid];
    const int row = n * MX2, rm = (n > 0 ? row - MX2 : row) + c, rp = (n < NX - 1 ? row + MX2 : row) + c, rc = row + (c ^ 1);
    double s_pm[3], s_pp[3], s_pc[3], s_qm[3], s_qp[3], s_qc[3], s_one[4], t_in[5];
    double gx = 0., ym = 0., yp = 0., el2 = H.el2[hm];
    if (FROM_FLUX) {
        gx = H.gradx[m]; ym = H.vddym[hm]; yp = H.vddyp[hm];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double *P = S + (size_t)(16 * q + k) * SP, *Q = S + (size_t)(16 * q + 8 + k) * SP;
            s_pm[q] = P[rm]; s_pp[q] = P[rp]; s_pc[q] = P[rc];
            s_qm[q] = Q[rm]; s_qp[q] = Q[rp]; s_qc[q] = Q[rc];
        }
        s_one[0] = S[(size_t)(48 + k) * SP + e]; s_one[1] = S[(size_t)(56 + k) * SP + e]; s_one[2] = S[(size_t)(64 + k) * SP + e];
        s_one[3] = S[(size_t)72 * SP + e];
    } else {
        t_in[0] = tend_in[(size_t)(F_VOR + k) * SP + e]; t_in[1] = tend_in[(size_t)(F_DIV + k) * SP + e];
        t_in[2] = tend_in[(size_t)(F_T + k) * SP + e]; t_in[3] = tend_in[(size_t)(F_TR + k) * SP + e];
        t_in[4] = tend_in[(size_t)F_PS * SP + e];
    }
    double *s1 = state, *s2 = state + (size_t)NSTATE * SP;
    const size_t ov = (size_t)(F_VOR + k) * SP + e, od = (size_t)(F_DIV + k) * SP + e, ot = (size_t)(F_T + k) * SP + e,
                 oq = (size_t)(F_TR + k) * SP + e, op = (size_t)F_PS * SP + e;
    const bool two = a.j1 != 1;
    const double v1 = s1[ov], d1 = s1[od], t1 = s1[ot], q1 = s1[oq], p1 = s1[op];
    const double v2r = s2[ov], d2r = s2[od], t2r = s2[ot], q2r = s2[oq], p2r = s2[op];          // (always loaded; `two` selects below)
    const double dmp = H.dmp[hm], dmpd = H.dmpd[hm], dmps = H.dmps[hm];
    const double dmp1 = imp_h[hm], dmp1d = imp_h[NX * MX + hm], dmp1s = imp_h[2 * NX * MX + hm];
    const double tch = tcorh[e], qch = qcorh[e], tf = H.trfilt[hm], phis_e = phis[e];
    const int ll = m + n;
    const double elz = imp_h[3 * NX * MX + hm];
    double xl[KX];
    {
        const double *xrow = imp_x + 128 + (size_t)(ll > 0 ? ll - 1 : 0) * 64 + k;
#pragma unroll
        for (int k1 = 0; k1 < KX; ++k1) xl[k1] = xrow[k1 * KX];
    }
    // ---- from here on nothing is loaded from memory ----
    {
        double *lvf = &lv[0][0];
        lvf[tid] = lv_a;
        if (tid + 64 < LV_ROWS * KX) lvf[tid + 64] = lv_b;
        xdc[0][tid] = xd_v;
        xdc[1][tid] = xc_v;
    }
    const double v2 = two ? v2r : 0., d2 = two ? d2r : 0., t2 = two ? t2r : 0., q2 = two ? q2r : 0., p2 = two ? p2r : 0.;
#pragma unroll
    for (int k1 = 0; k1 < KX; ++k1) xl[k1] = (a.implicit && ll != 0) ? xl[k1] : 0.0;
    double vordt, divdt, tdt, trdt, psdt;
    if (FROM_FLUX) {
        // the 3-point-in-n stencil of vds (src/spe_spectral.f90:307-349) with the outer rows' missing terms as zero factors:
        //   A = ym P(n-1) - yp P(n+1) + i x Q ;  B = -ym Q(n-1) + yp Q(n+1) + i x P      (n = 0: no ym term; n = 31: the ym term alone)
        const double ym_e = n == 0 ? 0. : ym, yp_e = n == NX - 1 ? 0. : yp, g_e = n == NX - 1 ? 0. : gx, sgx = (c & 1) ? g_e : -g_e;
        double sa[3], sb[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            sa[q] = ym_e * s_pm[q] - yp_e * s_pp[q] + sgx * s_qc[q];
            sb[q] = -ym_e * s_qm[q] + yp_e * s_qp[q] + sgx * s_pc[q];
        }
        vordt = sa[0]; divdt = sb[0];
        const double lapke = -s_one[0] * el2;
        divdt = divdt - lapke;
        tdt = sb[1] + s_one[1];
        trdt = sb[2] + s_one[2];
        psdt = s_one[3];
        if (e < 2) psdt = 0.;
    } else {
        vordt = t_in[0]; divdt = t_in[1]; tdt = t_in[2]; trdt = t_in[3]; psdt = t_in[4];
    }
    if (!stop_after_grtend) {
        // ---- sptend (src/dyn_sptend.f90) on time level j4
        d4[k][ci] = a.j4 == 1 ? d1 : d2r;
        t4[k][ci] = a.j4 == 1 ? t1 : t2r;
        const double ps4 = a.j4 == 1 ? p1 : p2r;
        KSTAMP(1);
        __syncthreads();
        KSTAMP(2);
        double dmeanc = 0.0;
#pragma unroll
        for (int j = 0; j < KX; ++j) dmeanc = dmeanc + d4[j][ci] * lv[LV_DHS][j];
        psdt = psdt - dmeanc;
        if (e < 2) psdt = 0.;
        // sigdtc(k), sigdtc(k+1) (1-based interfaces k and k+1 of this level): running sum from the top, zero at both ends
        double sg = 0.0, sig_k = 0.0, sig_k1 = 0.0;
#pragma unroll
        for (int j = 0; j < KX - 1; ++j) {
            sg = sg - lv[LV_DHS][j] * (d4[j][ci] - dmeanc);
            if (j + 1 == k) sig_k = sg;
            if (j == k) sig_k1 = sg;
        }
        const double dumk_k = k > 0 ? sig_k * (lv[LV_TREF][k] - lv[LV_TREF][k > 0 ? k - 1 : 0]) : 0.0;
        const double dumk_k1 = k < KX - 1 ? sig_k1 * (lv[LV_TREF][k < KX - 1 ? k + 1 : k] - lv[LV_TREF][k]) : 0.0;
        tdt = tdt - (dumk_k1 + dumk_k) * lv[LV_DHSR][k] + lv[LV_TREF3][k] * (sig_k1 + sig_k) - lv[LV_TREF2][k] * dmeanc;
        // geop (src/dyn_geop.f90): hydrostatic integration from the surface up to this level
        double phi = phis_e + lv[LV_XG1][KX - 1] * t4[KX - 1][ci];
#pragma unroll
        for (int j = KX - 2; j >= 0; --j)
            if (j >= k) phi = phi + lv[LV_XG2][j + 1] * t4[j + 1][ci] + lv[LV_XG1][j] * t4[j][ci];
        if (c < 2 && k >= 1 && k <= KX - 2) phi = phi + lv[LV_CORF][k] * (t4[k + 1][ci] - t4[k - 1][ci]);
        {
            const double g1 = phi + lv[LV_TREF1][k] * ps4;
            const double g2 = -g1 * el2;
            divdt = divdt - g2;
        }
        KSTAMP(3);
        // ---- implic (src/dyn_implic.f90)
        if (a.implicit) {
            tds[k][ci] = tdt;
            __syncthreads();
            double ye = 0.;
#pragma unroll
            for (int k1 = 0; k1 < KX; ++k1) ye = ye + xdc[0][k1 * KX + k] * tds[k1][ci];
            ye = ye + lv[LV_TREF1][k] * psdt;
            yfs[k][ci] = divdt + elz * ye;
            __syncthreads();
            divdt = 0.;
            if (ll != 0) {
#pragma unroll
                for (int k1 = 0; k1 < KX; ++k1) divdt = divdt + xl[k1] * yfs[k1][ci];
            }
            dvs[k][ci] = divdt;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KX; ++j) psdt = psdt - dvs[j][ci] * lv[LV_DHSX][j];
#pragma unroll
            for (int k1 = 0; k1 < KX; ++k1) tdt = tdt + xdc[1][k1 * KX + k] * dvs[k1][ci];
        }
        KSTAMP(4);
        // ---- horizontal diffusion (src/dyn_step.f90:60-104, hordif :130-150), always on time level 1
        const double ct = t1 + tch * lv[LV_TCORV][k];
        const double cq = q1 + qch * lv[LV_QCORV][k];
        vordt = (vordt - dmp * v1) * dmp1;
        divdt = (divdt - dmpd * d1) * dmp1d;
        tdt = (tdt - dmp * ct) * dmp1;
        trdt = (trdt - dmpd * cq) * dmp1d;
        if (k == 0) {
            if (c < 2) { vordt = vordt - L.sdrag * v1; divdt = divdt - L.sdrag * d1; }
            vordt = (vordt - dmps * v1) * dmp1s;
            divdt = (divdt - dmps * d1) * dmp1s;
            tdt = (tdt - dmps * ct) * dmp1s;
        }
    }
    if (tend_out) {
        tend_out[(size_t)(F_VOR + k) * SP + e] = vordt;
        tend_out[(size_t)(F_DIV + k) * SP + e] = divdt;
        tend_out[(size_t)(F_T + k) * SP + e] = tdt;
        tend_out[(size_t)(F_TR + k) * SP + e] = trdt;
        if (k == 0) tend_out[(size_t)F_PS * SP + e] = psdt;
    }
    if (stop_after_grtend || !a.integrate) return;
    // ---- timint (src/dyn_step.f90:152-190): truncation, leapfrog, Robert-Asselin-Williams filter, both levels in place
    auto timint = [&](double f1, double f2, double fdt, double &o1, double &o2) {
        fdt = fdt * tf;
        const double fj = two ? f2 : f1;
        const double fnew = f1 + a.dt * fdt;
        const double n1 = fj + a.wil * a.eps * (f1 - 2 * fj + fnew);
        const double fj_after = two ? fj : n1;                // field(.,1) is overwritten before field(.,2) is formed
        o1 = n1;
        o2 = fnew - (1 - a.wil) * a.eps * (n1 - 2 * fj_after + fnew);
    };
    double nv1, nv2, nd1, nd2, nt1, nt2, nq1, nq2, np1 = 0., np2 = 0.;
    timint(v1, v2, vordt, nv1, nv2);
    timint(d1, d2, divdt, nd1, nd2);
    timint(t1, t2, tdt, nt1, nt2);
    timint(q1, q2, trdt, nq1, nq2);
    if (k == 0) timint(p1, p2, psdt, np1, np2);       // a column's ps is only touched by its k == 0 thread
    s1[ov] = nv1; s2[ov] = nv2;
    s1[od] = nd1; s2[od] = nd2;
    s1[ot] = nt1; s2[ot] = nt2;
    s1[oq] = nq1; s2[oq] = nq2;
    if (k == 0) { s1[op] = np1; s2[op] = np2; }
    KSTAMP(5);
    if (phi_out) {
        // geop(1) of the time level 1 this step leaves behind (src/dyn_geop.f90:19-35): what the NEXT step's phypar reads as phig1
        // (src/dyn_grtend.f90:222-224).  The 8 levels of a coefficient are in this wavefront anyway (lane 8 j + ci holds level j); the
        // next k_grid launch then stages eight plain fields instead of rebuilding the hydrostatic chain from eight temperature levels in
        // each of its workgroups of those rows (its slowest: 14 against 9-10 us, the launch's tail).  Same expression, same order, same
        // bits as derived_coeff 7.  Lane exchange, no barrier: the chain runs while the stores above drain.
        double tl[KX];
#pragma unroll
        for (int j = 0; j < KX; ++j) tl[j] = __shfl(nt1, j * 8 + ci, 64);
        double phi = phis_e + lv[LV_XG1][KX - 1] * tl[KX - 1];
#pragma unroll
        for (int j = KX - 2; j >= 0; --j)
            if (j >= k) phi = phi + lv[LV_XG2][j + 1] * tl[j + 1] + lv[LV_XG1][j] * tl[j];
        if (c < 2 && k >= 1 && k <= KX - 2) {
            double up = 0., dn = 0.;                            // t(k+1), t(k-1) of this thread's own level (predicated picks)
#pragma unroll
            for (int j = 0; j < KX; ++j) { if (j == k + 1) up = tl[j]; if (j == k - 1) dn = tl[j]; }
            phi = phi + lv[LV_CORF][k] * (up - dn);
        }
        phi_out[(size_t)k * SP + e] = phi;
    }
    KSTAMP(6);
}


// =====================================================================================================================
// The 6-hour window as TWO kernels per time step (sml_dyn_window): zonal-wavenumber space <-> latitude space.
//
// A spectral transform is a Legendre transform (couples the total wavenumbers n of ONE zonal wavenumber m across latitudes)
// times a Fourier transform (couples the longitudes / zonal wavenumbers of ONE latitude).  Everything SPEEDY does between two
// transforms is local to one of the two spaces: the spectral step couples n (3-point stencils) and the 8 levels of one m;
// the grid-point tendencies couple the 8 levels of one grid point.  So a time step needs only ONE exchange in each direction:
//
//   k_latspace (one workgroup per latitude row, 48):  Fourier coefficients of the 50 inverse fields of this row ->
//       Fourier synthesis (96 longitudes) -> grid-point tendencies of the row's 96 columns -> forward Fourier transform of the
//       73 tendency fields -> their Fourier coefficients FB[m][field][lat].
//   k_mspace  (one workgroup per zonal wavenumber, 31):  FB[m] -> Gaussian quadrature / Legendre analysis of the 73 fields
//       -> vds, Laplacian, sptend, geop, implic, hordif, truncation, leapfrog of the 32 x 8 coefficients of this m -> uvspec and
//       grad of the NEW state -> Legendre synthesis of the 50 inverse fields of the next step -> FA[lat][field][m].
//
// Against the four-launch form (k_grid, k_gridtend, k_spec, k_spectral) this halves the launches, removes the 8-fold / 6-fold
// re-staging of every field by the batched transform kernels, and never writes a grid or a spectral batch to HBM: the only
// traffic between the two kernels is 1.2 MB + 1.7 MB of Fourier coefficients.  Every sum keeps the order of the kernels it
// replaces (Legendre sums as gridy/specy, DFT as k_grid / k_spec, vertical and semi-implicit algebra as k_gridtend /
// k_spectral), so the results are BIT-IDENTICAL to theirs (tests/test_dynamics_gpu.py).
//
// MEASURED, and why it is not the default (SML_DYN_TWO_KERNEL=1 selects it): 32.7 + 27.7 = 60 us per time step against 43 us
// for the four launches.  A zonal wavenumber or a latitude row is served by ONE compute unit -- 31 and 48 workgroups on a
// 256-CU chip -- and the 23 M multiply-adds of a step then run at the fp64 rate and LDS bandwidth of 79 CUs with 5-12
// wavefronts each: phase stamps (profiles/micro/window_phase_stamps.py) give k_mspace = load 8.5 (strided state gather) +
// analysis 5.3 + spectral step 5.2 + synthesis 9 us and k_latspace = load 1.9 + synthesis 5.8 + tendencies 2.8 + fold 1.5 +
// forward DFT 6.5 us.  Register tiling of the four transform loops (4 x 4 / 2 x 4 outputs per thread) took k_mspace from
// 51 to 33 us; LDS stride padding and deeper unrolling changed nothing -- the loops are bound by the latency of a nearly
// empty CU, not by bank conflicts.  The four-launch form pays 19 us of launch floor per step but spreads the same arithmetic
// over 300-580 workgroups.
constexpr int MS_THREADS = 512, LS_THREADS = 768, NFA = 50, NFB = 73;
// LDS strides are padded off the 256-byte bank period: lanes of a wavefront differ in field / latitude-pair index, and the
// natural strides (64, 96 doubles per field, 32 per Legendre row) put them all on the same banks (measured: the Legendre
// synthesis loop took 5.8 us instead of 1.5)
constexpr int FS = 66;       // doubles per field of one zonal wavenumber: [32][2] + 2
constexpr int FBS = 98;      // doubles per forward field: [48][2] + 2
constexpr int PLS = 33;      // doubles per Legendre row: [32] + 1
constexpr int TS = 98;       // doubles per tendency-field row in k_latspace: [96] + 2
constexpr int MS_FB = NFB * FBS, MS_SP = NFB * FS, MS_PL = 24 * PLS, MS_ST = 2 * NSTATE * FS, MS_VC = 5 * 512;
constexpr size_t MS_LDS = (size_t)(MS_FB + MS_SP + MS_PL + MS_ST + MS_VC + LV_ROWS * KX + 128 + 24) * sizeof(double) + 32 * sizeof(int);
constexpr size_t LS_LDS = (size_t)(NFA * MX2 + NFA * IX + NFB * TS + 2 * IX + LV_ROWS * KX) * sizeof(double);

// phase time stamps of workgroup 5 (profiles/micro/window_phase_stamps.py); compiled in with -DSML_DYN_STAMPS only
__device__ unsigned long long g_dbg[64];
#ifdef SML_DYN_STAMPS
#define STAMP(slot) do { if (blockIdx.x == 5 && threadIdx.x == 0 && g_dbg[slot] == 0) g_dbg[slot] = wall_clock64(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif
struct WinTables {
    const double *polT;      // [31][24][32]: P_m^n at latitude j, one contiguous slab per zonal wavenumber
    const double *wt;        // [24] Gaussian weights
    const double *cosgr;     // [48]
    const double *twc, *tws; // [96] cos / sin(2 pi t / 96), the table of spectral.hip
    const int *nsh2;         // [32]
};

__device__ __forceinline__ double irot_l(const double *a, int n, int ri, double g) { return ri ? g * a[n * 2] : -g * a[n * 2 + 1]; }

// the 3-point-in-n stencil of uvspec / vds on one zonal wavenumber held as [n][re,im] (same expressions as stencil())
__device__ __forceinline__ void stencil_l(const double *P, const double *Q, int n, int ri, double gx, double ym, double yp, double &a, double &b)
{
    if (n == 0) {
        a = irot_l(Q, n, ri, gx) - yp * P[(n + 1) * 2 + ri];
        b = irot_l(P, n, ri, gx) + yp * Q[(n + 1) * 2 + ri];
    } else if (n == NX - 1) {
        a = ym * P[(n - 1) * 2 + ri];
        b = -ym * Q[(n - 1) * 2 + ri];
    } else {
        a = ym * P[(n - 1) * 2 + ri] - yp * P[(n + 1) * 2 + ri] + irot_l(Q, n, ri, gx);
        b = -ym * Q[(n - 1) * 2 + ri] + yp * Q[(n + 1) * 2 + ri] + irot_l(P, n, ri, gx);
    }
}

__global__ __launch_bounds__(MS_THREADS) void k_mspace(DevHoriz H, WinTables W, StepArgs a, double sdrag, int do_step, int next_j2,
                                                        const double *__restrict__ FB, double *__restrict__ FA, double *__restrict__ state,
                                                        const double *__restrict__ imp_h, const double *__restrict__ imp_x,
                                                        const double *__restrict__ phis, const double *__restrict__ tcorh,
                                                        const double *__restrict__ qcorh)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *fb = sm;                        // [73][48][2] Fourier coefficients of the forward fields, then sym / antisym parts
    double *sp = fb + MS_FB;                // [73][32][2] their spectral coefficients; later [18][32][2] derived inverse fields
    double *pl = sp + MS_SP;                // [24][32]    Legendre functions of this zonal wavenumber
    double *st = pl + MS_PL;                // [2][33][32][2] both time levels of the state
    double *vc = st + MS_ST;                // 5 x [8][64] vertical coupling
    double *lv = vc + MS_VC;                // [13][8]
    double *xdc = lv + LV_ROWS * KX;        // xd, xc
    double *wts = xdc + 128;                // [24]
    int *nsh = reinterpret_cast<int *>(wts + 24);
    const int m = blockIdx.x, tid = threadIdx.x;
    STAMP(0);
    for (int i = tid; i < 24 * 32; i += MS_THREADS) pl[(i >> 5) * PLS + (i & 31)] = W.polT[(size_t)m * 768 + i];
    for (int i = tid; i < LV_ROWS * KX; i += MS_THREADS) lv[i] = imp_x[128 + LMAX * 64 + i];
    if (tid < 128) xdc[tid] = imp_x[tid];
    if (tid < 24) wts[tid] = W.wt[tid];
    if (tid < 32) nsh[tid] = W.nsh2[tid];
    for (int i = tid; i < 2 * NSTATE * 64; i += MS_THREADS) {
        const int f = i >> 6, q = i & 63;                 // f counts both levels: 0..65
        st[f * FS + q] = state[(size_t)f * SP + (q >> 1) * MX2 + 2 * m + (q & 1)];
    }
    if (do_step)
        for (int i = tid; i < NFB * 96; i += MS_THREADS) fb[(i / 96) * FBS + i % 96] = FB[(size_t)m * NFB * 96 + i];
    __syncthreads();
    STAMP(1);
    if (do_step) {
        // symmetric / antisymmetric parts times the Gaussian weight, in place (specy :511-517)
        for (int it = tid; it < NFB * 48; it += MS_THREADS) {
            const int f = it / 48, r = it % 48, j = r >> 1, ri = r & 1;
            const double n_ = fb[f * FBS + (47 - j) * 2 + ri], s_ = fb[f * FBS + j * 2 + ri], wj = wts[j];
            fb[f * FBS + j * 2 + ri] = (n_ + s_) * wj;
            fb[f * FBS + (47 - j) * 2 + ri] = (n_ - s_) * wj;
        }
        __syncthreads();
        STAMP(2);
        // Legendre analysis (specy :519-537), latitude sums in the reference's order.  Register tile of 4 fields x 4 total
        // wavenumbers of one parity per thread: 8 LDS operands feed 16 accumulators (the one-output-per-thread form spent
        // 6 us per launch on LDS reads: this whole zonal wavenumber is served by ONE CU's LDS)
        for (int it = tid; it < 19 * 16; it += MS_THREADS) {
            const int nq = it & 3, par = (it >> 2) & 1, ri = (it >> 3) & 1, f0 = (it >> 4) * 4, c = 2 * m + ri;
            double acc[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
            const int n0 = par + 8 * nq;                                  // n = n0 + 2 t, t = 0..3
#pragma unroll 6
            for (int j = 0; j < IY; ++j) {
                const int lat = par ? 47 - j : j;
                double pv[4], fv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) pv[t] = pl[j * PLS + n0 + 2 * t];
#pragma unroll
                for (int x = 0; x < 4; ++x) fv[x] = fb[min(f0 + x, NFB - 1) * FBS + lat * 2 + ri];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[x][t] = acc[x][t] + pv[t] * fv[x];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n = n0 + 2 * t;
                    if (f0 + x < NFB) sp[(f0 + x) * FS + n * 2 + ri] = (n < NX - 1 && c < nsh[n]) ? acc[x][t] : 0.0;
                }
        }
        __syncthreads();
        STAMP(3);
        // ---- the spectral step for the 32 x 2 coefficients of this m and the 8 levels: one thread each (cf. k_spectral)
        const int ci = tid & 63, k = tid >> 6, n = ci >> 1, ri = ci & 1, hm = n * MX + m;
        const size_t e = (size_t)n * MX2 + 2 * m + ri;
        double *d4 = vc, *t4 = vc + 512, *tds = vc + 1024, *yfs = vc + 1536, *dvs = vc + 2048;
        double *s1 = st, *s2 = st + NSTATE * FS;
        double vordt, divdt, tdt, trdt, psdt;
        {
            const double gx = H.gradx[m], ym = H.vddym[hm], yp = H.vddyp[hm], el2 = H.el2[hm];
            double dummy;
            stencil_l(sp + k * FS, sp + (8 + k) * FS, n, ri, gx, ym, yp, vordt, divdt);
            const double lapke = -sp[(48 + k) * FS + ci] * el2;
            divdt = divdt - lapke;
            stencil_l(sp + (16 + k) * FS, sp + (24 + k) * FS, n, ri, gx, ym, yp, dummy, tdt);
            tdt = tdt + sp[(56 + k) * FS + ci];
            stencil_l(sp + (32 + k) * FS, sp + (40 + k) * FS, n, ri, gx, ym, yp, dummy, trdt);
            trdt = trdt + sp[(64 + k) * FS + ci];
            psdt = sp[72 * FS + ci];
            if (m == 0 && n == 0) psdt = 0.;
        }
        {
            const double *s4 = a.j4 == 1 ? s1 : s2;
            d4[k * 64 + ci] = s4[(F_DIV + k) * FS + ci];
            t4[k * 64 + ci] = s4[(F_T + k) * FS + ci];
            const double ps4 = s4[F_PS * FS + ci];
            const double el2 = H.el2[hm];
            __syncthreads();
            double dmeanc = 0.0;
#pragma unroll
            for (int j = 0; j < KX; ++j) dmeanc = dmeanc + d4[j * 64 + ci] * lv[LV_DHS * KX + j];
            psdt = psdt - dmeanc;
            if (m == 0 && n == 0) psdt = 0.;
            double sg = 0.0, sig_k = 0.0, sig_k1 = 0.0;
#pragma unroll
            for (int j = 0; j < KX - 1; ++j) {
                sg = sg - lv[LV_DHS * KX + j] * (d4[j * 64 + ci] - dmeanc);
                if (j + 1 == k) sig_k = sg;
                if (j == k) sig_k1 = sg;
            }
            const double dumk_k = k > 0 ? sig_k * (lv[LV_TREF * KX + k] - lv[LV_TREF * KX + (k > 0 ? k - 1 : 0)]) : 0.0;
            const double dumk_k1 = k < KX - 1 ? sig_k1 * (lv[LV_TREF * KX + (k < KX - 1 ? k + 1 : k)] - lv[LV_TREF * KX + k]) : 0.0;
            tdt = tdt - (dumk_k1 + dumk_k) * lv[LV_DHSR * KX + k] + lv[LV_TREF3 * KX + k] * (sig_k1 + sig_k) - lv[LV_TREF2 * KX + k] * dmeanc;
            double phi = phis[e] + lv[LV_XG1 * KX + KX - 1] * t4[(KX - 1) * 64 + ci];
#pragma unroll
            for (int j = KX - 2; j >= 0; --j)
                if (j >= k) phi = phi + lv[LV_XG2 * KX + j + 1] * t4[(j + 1) * 64 + ci] + lv[LV_XG1 * KX + j] * t4[j * 64 + ci];
            if (m == 0 && k >= 1 && k <= KX - 2) phi = phi + lv[LV_CORF * KX + k] * (t4[(k + 1) * 64 + ci] - t4[(k - 1) * 64 + ci]);
            {
                const double g1 = phi + lv[LV_TREF1 * KX + k] * ps4;
                const double g2 = -g1 * el2;
                divdt = divdt - g2;
            }
            if (a.implicit) {
                const double elz = imp_h[3 * NX * MX + hm];
                const int ll = m + n;
                double xl[KX];
#pragma unroll
                for (int k1 = 0; k1 < KX; ++k1) xl[k1] = ll != 0 ? imp_x[128 + (size_t)(ll - 1) * 64 + k1 * KX + k] : 0.0;
                tds[k * 64 + ci] = tdt;
                __syncthreads();
                double ye = 0.;
#pragma unroll
                for (int k1 = 0; k1 < KX; ++k1) ye = ye + xdc[k1 * KX + k] * tds[k1 * 64 + ci];
                ye = ye + lv[LV_TREF1 * KX + k] * psdt;
                yfs[k * 64 + ci] = divdt + elz * ye;
                __syncthreads();
                divdt = 0.;
                if (ll != 0) {
#pragma unroll
                    for (int k1 = 0; k1 < KX; ++k1) divdt = divdt + xl[k1] * yfs[k1 * 64 + ci];
                }
                dvs[k * 64 + ci] = divdt;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < KX; ++j) psdt = psdt - dvs[j * 64 + ci] * lv[LV_DHSX * KX + j];
#pragma unroll
                for (int k1 = 0; k1 < KX; ++k1) tdt = tdt + xdc[64 + k1 * KX + k] * dvs[k1 * 64 + ci];
            }
            const double dmp = H.dmp[hm], dmpd = H.dmpd[hm], dmps = H.dmps[hm];
            const double dmp1 = imp_h[hm], dmp1d = imp_h[NX * MX + hm], dmp1s = imp_h[2 * NX * MX + hm];
            const double v1 = s1[(F_VOR + k) * FS + ci], d1 = s1[(F_DIV + k) * FS + ci];
            const double ct = s1[(F_T + k) * FS + ci] + tcorh[e] * lv[LV_TCORV * KX + k];
            const double cq = s1[(F_TR + k) * FS + ci] + qcorh[e] * lv[LV_QCORV * KX + k];
            vordt = (vordt - dmp * v1) * dmp1;
            divdt = (divdt - dmpd * d1) * dmp1d;
            tdt = (tdt - dmp * ct) * dmp1;
            trdt = (trdt - dmpd * cq) * dmp1d;
            if (k == 0) {
                if (m == 0) { vordt = vordt - sdrag * v1; divdt = divdt - sdrag * d1; }
                vordt = (vordt - dmps * v1) * dmp1s;
                divdt = (divdt - dmps * d1) * dmp1s;
                tdt = (tdt - dmps * ct) * dmp1s;
            }
        }
        if (a.integrate) {
            const double tf = H.trfilt[hm];
            double *g1 = state, *g2 = state + (size_t)NSTATE * SP;
            auto timint = [&](int f, double fdt) {
                fdt = fdt * tf;
                const int o = f * FS + ci;
                const double f1 = s1[o];
                const double fj = a.j1 == 1 ? f1 : s2[o];
                const double fnew = f1 + a.dt * fdt;
                const double n1 = fj + a.wil * a.eps * (f1 - 2 * fj + fnew);
                const double fj_after = a.j1 == 1 ? n1 : fj;
                const double n2 = fnew - (1 - a.wil) * a.eps * (n1 - 2 * fj_after + fnew);
                s1[o] = n1; s2[o] = n2;
                g1[(size_t)f * SP + e] = n1; g2[(size_t)f * SP + e] = n2;
            };
            timint(F_VOR + k, vordt);
            timint(F_DIV + k, divdt);
            timint(F_T + k, tdt);
            timint(F_TR + k, trdt);
            if (k == 0) timint(F_PS, psdt);
        }
        __syncthreads();
    }
    STAMP(4);
    if (!next_j2) return;
    // ---- inverse side for the next grtend: uvspec of the 8 levels and grad(ps) of time level next_j2, then Legendre synthesis
    const double *sl = st + (next_j2 - 1) * NSTATE * FS;
    double *syn = sp;
    for (int it = tid; it < 18 * 64; it += MS_THREADS) {
        const int fd = it >> 6, q = it & 63, n = q >> 1, ri = q & 1, hm = n * MX + m;
        double out;
        if (fd < 16) {
            const int k = fd & 7;
            double ua, vb;
            stencil_l(sl + (F_VOR + k) * FS, sl + (F_DIV + k) * FS, n, ri, H.uvdx[hm], H.uvdym[hm], H.uvdyp[hm], ua, vb);
            out = fd < 8 ? ua : vb;
        } else {
            const double *ps = sl + F_PS * FS;
            if (fd == 16) out = irot_l(ps, n, ri, H.gradx[m]);
            else if (n == 0) out = H.gradyp[m] * ps[(n + 1) * 2 + ri];
            else if (n == NX - 1) out = -H.gradym[hm] * ps[(n - 1) * 2 + ri];
            else out = -H.gradym[hm] * ps[(n - 1) * 2 + ri] + H.gradyp[hm] * ps[(n + 1) * 2 + ri];
        }
        syn[fd * FS + q] = out;
    }
    __syncthreads();
    STAMP(5);
    // Legendre synthesis (gridy :454-495): register tile of 2 fields x 4 latitude pairs per thread
    for (int it = tid; it < 25 * 12; it += MS_THREADS) {
        const int jq = it % 6, ri = (it / 6) & 1, f0 = (it / 12) * 2, c = 2 * m + ri;
        const double *v0 = (f0 < 32 ? sl + f0 * FS : syn + (f0 - 32) * FS) + ri, *v1 = v0 + FS;
        double ev[2][4], od[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int t = 0; t < 4; ++t) { ev[x][t] = 0.0; od[x][t] = 0.0; }
#pragma unroll 4
        for (int n = 0; n < NX; n += 2) {
            const bool me = c < nsh[n], mo = c < nsh[n + 1];
            const double a0 = v0[n * 2], a1 = v1[n * 2], b0 = v0[(n + 1) * 2], b1 = v1[(n + 1) * 2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double pe = pl[(jq * 4 + t) * PLS + n], po = pl[(jq * 4 + t) * PLS + n + 1];
                if (me) { ev[0][t] = ev[0][t] + a0 * pe; ev[1][t] = ev[1][t] + a1 * pe; }
                if (mo) { od[0][t] = od[0][t] + b0 * po; od[1][t] = od[1][t] + b1 * po; }
            }
        }
        STAMP(7);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int jj = jq * 4 + t, f = f0 + x;
                FA[((size_t)(IL - 1 - jj) * NFA + f) * MX2 + c] = ev[x][t] + od[x][t];       // northern row il+1-j
                FA[((size_t)jj * NFA + f) * MX2 + c] = ev[x][t] - od[x][t];                   // southern row j
            }
    }
    STAMP(6);
}

__global__ __launch_bounds__(LS_THREADS) void k_latspace(DevHoriz H, WinTables W, double akap, const double *__restrict__ lvg,
                                                          const double *__restrict__ FA, double *__restrict__ FB)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *fa = sm;                  // [50][62] Fourier coefficients of this latitude row
    double *g = fa + NFA * MX2;       // [50][96] grid row: vor div t tr (8 levels each) | u(8) | v(8) | dps/dx | dps/dy
    double *T = g + NFA * IX;         // [73][96] tendency fields (layout of k_gridtend), then folded about longitude 48
    double *tc = T + NFB * TS, *ts = tc + IX;
    double *lv = ts + IX;             // [13][8]
    const int row = blockIdx.x, tid = threadIdx.x;
    STAMP(10);
    for (int i = tid; i < NFA * MX2; i += LS_THREADS) fa[i] = FA[(size_t)row * NFA * MX2 + i];
    for (int i = tid; i < IX; i += LS_THREADS) { tc[i] = W.twc[i]; ts[i] = W.tws[i]; }
    for (int i = tid; i < LV_ROWS * KX; i += LS_THREADS) lv[i] = lvg[i];
    __syncthreads();
    const double cg = W.cosgr[row];
    STAMP(11);
    // Fourier synthesis (gridx) of the 31 retained modes, two longitudes per work item (cf. k_grid); 4 fields per thread share
    // the twiddle reads
    for (int w = tid; w < 13 * (IX / 2 + 1); w += LS_THREADS) {
        const int i = w % (IX / 2 + 1), f0 = (w / (IX / 2 + 1)) * 4;
        const double *fc[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) fc[x] = fa + min(f0 + x, NFA - 1) * MX2;
        double A[4] = {0.0, 0.0, 0.0, 0.0}, B[4] = {0.0, 0.0, 0.0, 0.0};
        int ph = 0;
#pragma unroll 5
        for (int kk = 1; kk <= MX - 1; ++kk) {
            ph += i;
            if (ph >= IX) ph -= IX;
            const double cs = tc[ph], sn = ts[ph];
#pragma unroll
            for (int x = 0; x < 4; ++x) { A[x] += fc[x][2 * kk] * cs; B[x] += fc[x][2 * kk + 1] * sn; }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int f = f0 + x;
            if (f >= NFA) break;
            double x0 = fc[x][0] + 2.0 * (A[x] - B[x]), x1 = fc[x][0] + 2.0 * (A[x] + B[x]);
            if (f >= 32) { x0 = x0 * cg; x1 = x1 * cg; }              // grid(.,.,2) for u, v and grad(ps)
            g[f * IX + i] = x0;
            if (i != 0 && i != IX / 2) g[f * IX + IX - i] = x1;
        }
    }
    __syncthreads();
    STAMP(12);
    // grid-point tendencies (dyn_grtend.f90:80-216): one thread per (longitude, level); the column sums are re-derived by every
    // thread from the LDS row in the reference's order, so no exchange between the level-threads is needed
    {
        const int i = tid % IX, k = tid / IX;
        const double cor = H.coriol[row];
        auto G = [&](int f) { return g[f * IX + i]; };
        double px = G(48), py = G(49);
        double umean = 0.0, vmean = 0.0, dmean = 0.0;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            umean = umean + G(32 + j) * lv[LV_DHS * KX + j];
            vmean = vmean + G(40 + j) * lv[LV_DHS * KX + j];
            dmean = dmean + G(F_DIV + j) * lv[LV_DHS * KX + j];
        }
        if (k == 0) T[72 * TS + i] = -umean * px - vmean * py;
        // sigma-dot and its mass-flux part at interfaces k and k+1 of this level (running sums from the top)
        double sd = 0.0, sm_ = 0.0, sd_k = 0.0, sd_k1 = 0.0, sm_k = 0.0, sm_k1 = 0.0, puv_k = 0.0;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const double puv = (G(32 + j) - umean) * px + (G(40 + j) - vmean) * py;
            if (j == k) { sd_k = sd; sm_k = sm_; puv_k = puv; }
            sd = sd - lv[LV_DHS * KX + j] * (puv + G(F_DIV + j) - dmean);
            sm_ = sm_ - lv[LV_DHS * KX + j] * puv;
            if (j == k) { sd_k1 = sd; sm_k1 = sm_; }
        }
        const double ug = G(32 + k), vg = G(40 + k), tg = G(F_T + k), trg = G(F_TR + k), divg = G(F_DIV + k);
        const double vorg = G(F_VOR + k) + cor;
        const double trefk = lv[LV_TREF * KX + k];
        const double tgg = tg - trefk;
        const int km = k > 0 ? k - 1 : 0, kp = k < KX - 1 ? k + 1 : KX - 1;
        const double ug_m = G(32 + km), ug_p = G(32 + kp), vg_m = G(40 + km), vg_p = G(40 + kp);
        const double tgg_m = G(F_T + km) - lv[LV_TREF * KX + km], tgg_p = G(F_T + kp) - lv[LV_TREF * KX + kp];
        const double trg_m = G(F_TR + km), trg_p = G(F_TR + kp);
        px = RGAS * px;
        py = RGAS * py;
        const double dhsr = lv[LV_DHSR * KX + k];
        const bool top = k == 0, bot = k == KX - 1;
        // temp(k) = sigdt(k) * (f(k) - f(k-1)) at the upper interface, temp(k+1) at the lower one; zero at the two ends
        double t0 = top ? 0.0 : sd_k * (ug - ug_m), t1 = bot ? 0.0 : sd_k1 * (ug_p - ug);
        T[k * TS + i] = vg * vorg - tgg * px - (t1 + t0) * dhsr;
        t0 = top ? 0.0 : sd_k * (vg - vg_m); t1 = bot ? 0.0 : sd_k1 * (vg_p - vg);
        T[(8 + k) * TS + i] = -ug * vorg - tgg * py - (t1 + t0) * dhsr;
        t0 = top ? 0.0 : sd_k * (tgg - tgg_m) + sm_k * (trefk - lv[LV_TREF * KX + km]);
        t1 = bot ? 0.0 : sd_k1 * (tgg_p - tgg) + sm_k1 * (lv[LV_TREF * KX + kp] - trefk);
        T[(56 + k) * TS + i] = tgg * divg - (t1 + t0) * dhsr + lv[LV_FSGR * KX + k] * tgg * (sd_k1 + sd_k)
                               + lv[LV_TREF3 * KX + k] * (sm_k1 + sm_k) + akap * (tg * puv_k - tgg * dmean);
        // tracer: no vertical advection across the two uppermost interior interfaces (:196-203)
        t0 = (top || k == 1 || k == 2) ? 0.0 : sd_k * (trg - trg_m);
        t1 = (bot || k == 0 || k == 1) ? 0.0 : sd_k1 * (trg_p - trg);
        T[(64 + k) * TS + i] = trg * divg - (t1 + t0) * dhsr;
        T[(48 + k) * TS + i] = 0.5 * (ug * ug + vg * vg);
        T[(16 + k) * TS + i] = -ug * tgg;
        T[(24 + k) * TS + i] = -vg * tgg;
        T[(32 + k) * TS + i] = -ug * trg;
        T[(40 + k) * TS + i] = -vg * trg;
    }
    __syncthreads();
    STAMP(13);
    // fold about longitude 48 in place, after the vdspec pre-scaling (fields 0..47 by 1/cos): T[f][i] <- x_i + x_{96-i},
    // T[f][96-i] <- x_i - x_{96-i} (i = 1..47); x_0 and x_48 stay
    for (int w = tid; w < NFB * (IX / 2 + 1); w += LS_THREADS) {
        const int i = w % (IX / 2 + 1), f = w / (IX / 2 + 1);
        double x = T[f * TS + i];
        if (i == 0 || i == IX / 2) {
            if (f < 48) T[f * TS + i] = x * cg;
        } else {
            double y = T[f * TS + IX - i];
            if (f < 48) { x = x * cg; y = y * cg; }
            T[f * TS + i] = x + y;
            T[f * TS + IX - i] = x - y;
        }
    }
    __syncthreads();
    STAMP(14);
    // forward DFT (specx) of the 31 retained modes: Re_k = sum ss cos / 96, Im_k = - sum sd sin / 96, Im of k = 0 is 0
    for (int w = tid; w < 19 * MX; w += LS_THREADS) {
        const int kk = w % MX, f0 = (w / MX) * 4;
        const double *x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = T + min(f0 + q, NFB - 1) * TS;
        double re[4], im[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { re[q] = 0.0; im[q] = 0.0; re[q] += x[q][0] * tc[0]; im[q] -= x[q][0] * ts[0]; }
        int ph = 0;
#pragma unroll 4
        for (int i = 1; i <= IX / 2; ++i) {
            ph += kk;
            if (ph >= IX) ph -= IX;
            const double cs = tc[ph], sn = ts[ph];
            const int io = i == IX / 2 ? i : IX - i;
#pragma unroll
            for (int q = 0; q < 4; ++q) { re[q] += x[q][i] * cs; im[q] -= x[q][io] * sn; }
        }
        const double sc = 1. / (double)IX;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (f0 + q >= NFB) break;
            double *o = FB + ((size_t)kk * NFB + f0 + q) * 96 + row * 2;
            o[0] = re[q] * sc;
            o[1] = kk == 0 ? 0.0 : im[q] * sc;
        }
    }
    STAMP(15);
}

}  // namespace

struct sml_dyn {
    sml_spectral *sp = nullptr;
    HostDyn h;
    DevHoriz d{};
    std::vector<void *> allocs;
    std::vector<ImpSlot *> slots;
    ImpSlot *cur = nullptr;
    double *bc = nullptr;              // phis | tcorh | qcorh  (3 x SP)
    double *own_state = nullptr;       // [2][33][SP] for hosts that keep the state in Fortran arrays (sml_dyn_*_host)
    double *batch_grid = nullptr;      // [50 + 41][GR]: grtend's inverse set | phypar's inverse set (time level 1)
    sml_phys *phys = nullptr;          // column physics added to the grid-point tendencies (sml_dyn_attach_physics), not owned
    int phys_diag = 1;                 // keep the physics' 2-D diagnostics up to date (sml_phys_diag) during time steps
    int32_t *guard = nullptr;          // sml_dyn_set_range_guard: flag cleared when the first step's grids leave the physical range
    int nstrad = 3, lradsw = 1;        // short-wave radiation every nstrad-th step; flag of the next sml_dyn_step (mod_lflags.f90:22)
    int32_t *desc_phys = nullptr;      // inverse-batch descriptors with physics: [2 (j2)][91][4], fields relative to the whole state
    int32_t *desc_phys_phi = nullptr;  // the same with the eight geopotential rows taken from aux (type 8) instead of rebuilt (type 7)
    double *aux = nullptr;             // xgeop1 | xgeop2 | corf | phis | phi(8 levels): geopotential operands of the type-7 rows, and the
                                       // geop(1) fields k_spectral leaves for the type-8 rows of the next step
    double *tend_grid = nullptr;       // [73][GR]
    double *tend_spec = nullptr;       // [73][SP]
    int32_t *desc = nullptr, *scale = nullptr;     // inverse-batch descriptors [50][4], forward-batch scaling flags [73]
    WinTables win{};                   // tables of the two-kernel window
    double *fa = nullptr, *fb = nullptr;   // Fourier coefficients exchanged by k_mspace / k_latspace: [48][50][62], [31][73][48][2]
};

namespace {

int upload(sml_dyn *d, const double **dst, const double *src, size_t n)
{
    double *p = nullptr;
    int rc = sml::dev_upload(&p, src, n);
    if (rc) return rc;
    d->allocs.push_back(p);
    *dst = p;
    return SML_OK;
}

int fetch_table(sml_dyn *d, int which, const double **dst, std::vector<double> &tmp, int n)
{
    tmp.assign(n, 0.0);
    int got = sml_spectral_get_table(d->sp, which, tmp.data(), n);
    if (got != n) return sml::fail(SML_ERR_ARG, "sml_dyn_create: spectral table %d has %d entries, expected %d", which, got, n);
    return dst ? upload(d, dst, tmp.data(), n) : SML_OK;
}

int g_physics_fused = 3;     // sml_dyn_select_physics_form

// phi_ready: the previous launch of this window's k_spectral left geop(1) of the current time level 1 in d->aux (its phi_out), so
// the inverse set takes the physics' eight geopotential levels from there (descriptor type 8) instead of rebuilding them (type 7)
int run_step(sml_dyn *d, double *state, const StepArgs &a, int stop_after_grtend, double *tend_out, hipStream_t st, int lradsw = 1, bool phi_ready = false,
             bool leave_phi = false, int32_t *guard = nullptr)
{
    const double *sj2 = state + (size_t)(a.j2 - 1) * NSTATE * SP;
    // the 50 inverse transforms of grtend (:61-99) straight from the state: uvspec and grad are formed while the fields are staged.
    // With physics attached the same launch also produces the 27 grids of time level 1 that phypar's parametrisations read
    // (phy_phypar.f90:54-66 minus the unused wind levels), geop(1) included.
    int rc = d->phys ? sml_spectral_grid_derived_aux(d->sp, state, (phi_ready ? d->desc_phys_phi : d->desc_phys) + (size_t)(a.j2 - 1) * NB_ALL * 4, d->aux,
                                                     d->batch_grid, NB_ALL, st)
                     : sml_spectral_grid_derived(d->sp, sj2, d->desc, d->batch_grid, NB_SPEC, st);
    if (rc) return rc;
    if (d->phys && g_physics_fused) {      // dyn_grtend.f90:80-225 in one launch: grid-point tendencies + phypar
        const double *pg = d->batch_grid + (size_t)NB_SPEC * GR;
        smlphys::PhysIn in{pg, pg + (size_t)GR, pg + (size_t)2 * GR, pg + (size_t)10 * GR, pg + (size_t)18 * GR, pg + (size_t)26 * GR};
        if (g_physics_fused == 3)
            hipLaunchKernelGGL(k_gridtend_physics3, dim3(GR / 64), dim3(192), 0, st, d->d, d->cur->lv, d->batch_grid, d->tend_grid, d->phys->lev,
                               d->phys->dev, in, lradsw ? 1 : 0, d->phys_diag, guard);
        else if (g_physics_fused == 2)
            hipLaunchKernelGGL(k_gridtend_physics_onewave, dim3(GR / 64), dim3(64), 0, st, d->d, d->cur->lv, d->batch_grid, d->tend_grid, d->phys->lev,
                               d->phys->dev, in, lradsw ? 1 : 0, d->phys_diag);
        else
        hipLaunchKernelGGL(k_gridtend_physics, dim3(GR / 64), dim3(128), 0, st, d->d, d->cur->lv, d->batch_grid, d->tend_grid, d->phys->lev,
                           d->phys->dev, in, lradsw ? 1 : 0, d->phys_diag);
    } else {
        hipLaunchKernelGGL(k_gridtend, dim3(GR / 64), dim3(64), 0, st, d->d, d->cur->lv, d->batch_grid, d->tend_grid);
    }
    SML_HIP(hipGetLastError());
    if (d->phys && !g_physics_fused) {     // the same as two launches (sml_dyn_select_physics_form(0)): tests compare the two forms
        rc = sml_phys_tendencies_sfcwind(d->phys, d->batch_grid + (size_t)NB_SPEC * GR, lradsw, d->tend_grid, 0, 8, 56, 64, 1, d->phys_diag, st);
        if (rc) return rc;
    }
    rc = sml_spectral_spec_mixed(d->sp, d->tend_grid, d->tend_spec, NB_GRID, d->scale, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_spectral<true>, dim3(SP / 8), dim3(64), 0, st, d->d, d->cur->lv, a, d->tend_spec, (const double *)nullptr,
                       tend_out, stop_after_grtend, state, d->cur->d_h, d->cur->d_x, d->bc, d->bc + SP, d->bc + 2 * SP,
                       (d->phys && leave_phi) ? d->aux + 24 + SP : (double *)nullptr);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int g_window_form = -1;      // sml_dyn_select_window_form

StepArgs make_args(int j1, int j2, double dt, double alph, double rob, double wil)
{
    StepArgs a;
    a.j1 = j1; a.j2 = j2;
    a.implicit = alph != 0.;
    a.j4 = a.implicit ? 1 : j2;            // dyn_step.f90:47-57
    a.integrate = dt > 0.;
    a.dt = dt;
    a.eps = j1 == 1 ? 0. : rob;            // :112-116
    a.wil = wil;
    return a;
}

}  // namespace

extern "C" {

int sml_dyn_create(sml_spectral *sp, sml_dyn **out)
{
    SML_REQUIRE(sp && out, "sml_dyn_create: bad arguments");
    sml_dyn *d = new sml_dyn;
    d->sp = sp;
    std::vector<double> tmp;
    int rc = fetch_table(d, 1, nullptr, tmp, IY);
    if (!rc) {
        build_indyns(d->h, tmp.data());
        rc = upload(d, &d->d.coriol, d->h.coriol, IL);
    }
    if (!rc) rc = upload(d, &d->d.dmp, &d->h.dmp[0][0], NX * MX);
    if (!rc) rc = upload(d, &d->d.dmpd, &d->h.dmpd[0][0], NX * MX);
    if (!rc) rc = upload(d, &d->d.dmps, &d->h.dmps[0][0], NX * MX);
    if (!rc) rc = fetch_table(d, 8, &d->d.el2, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 11, &d->d.trfilt, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 16, &d->d.gradx, tmp, MX);
    if (!rc) rc = fetch_table(d, 17, &d->d.gradym, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 18, &d->d.gradyp, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 19, &d->d.uvdx, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 20, &d->d.uvdym, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 21, &d->d.uvdyp, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 22, &d->d.vddym, tmp, NX * MX);
    if (!rc) rc = fetch_table(d, 23, &d->d.vddyp, tmp, NX * MX);
    if (!rc) {
        // tables of the two-kernel window: Legendre functions regrouped by zonal wavenumber, weights, 1/cos, twiddles
        std::vector<double> cpol, polT((size_t)MX * IY * NX), w24, cg48, n32;
        rc = fetch_table(d, 24, nullptr, cpol, MX2 * NX * IY);
        for (int m = 0; m < MX && !rc; ++m)
            for (int j = 0; j < IY; ++j)
                for (int n = 0; n < NX; ++n) polT[((size_t)m * IY + j) * NX + n] = cpol[((size_t)j * NX + n) * MX2 + 2 * m];
        if (!rc) rc = upload(d, &d->win.polT, polT.data(), polT.size());
        if (!rc) rc = fetch_table(d, 3, &d->win.wt, w24, IY);
        if (!rc) rc = fetch_table(d, 6, &d->win.cosgr, cg48, IL);
        if (!rc) rc = fetch_table(d, 12, nullptr, n32, NX);
        if (!rc) {
            int32_t nsh[NX];
            for (int n = 0; n < NX; ++n) nsh[n] = (int32_t)n32[n];
            int32_t *dn = nullptr;
            rc = sml::dev_upload(&dn, nsh, NX);
            if (!rc) { d->allocs.push_back(dn); d->win.nsh2 = dn; }
        }
        if (!rc) {
            double twc[IX], tws[IX];                      // the twiddle table of spectral.hip (build_tables), same expressions
            for (int k = 0; k < IX; ++k) {
                const long double ang = 2.0L * 3.14159265358979323846264338327950288L * k / (long double)IX;
                twc[k] = (double)cosl(ang);
                tws[k] = (double)sinl(ang);
            }
            twc[24] = twc[72] = 0.0; tws[0] = tws[48] = 0.0;
            rc = upload(d, &d->win.twc, twc, IX);
            if (!rc) rc = upload(d, &d->win.tws, tws, IX);
        }
    }
    auto zeros = [&](double **p, size_t n) { int r = sml::dev_zeros(p, n); if (!r) d->allocs.push_back(*p); return r; };
    if (!rc) rc = zeros(&d->fa, (size_t)IL * NFA * MX2);
    if (!rc) rc = zeros(&d->fb, (size_t)MX * NFB * 96);
    if (!rc) {
        SML_HIP(hipFuncSetAttribute((const void *)k_mspace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MS_LDS));
        SML_HIP(hipFuncSetAttribute((const void *)k_latspace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LS_LDS));
    }
    if (!rc) rc = zeros(&d->bc, (size_t)3 * SP);
    if (!rc) rc = zeros(&d->own_state, (size_t)2 * NSTATE * SP);
    if (!rc) rc = zeros(&d->batch_grid, (size_t)NB_ALL * GR);
    if (!rc) rc = zeros(&d->aux, (size_t)24 + SP + (size_t)KX * SP);
    if (!rc) {
        double lv[24];
        for (int k = 0; k < KX; ++k) {
            lv[k] = d->h.xgeop1[k]; lv[8 + k] = d->h.xgeop2[k];
            lv[16 + k] = (k >= 1 && k <= KX - 2) ? d->h.xgeop1[k] * 0.5 * std::log(d->h.hsg[k + 1] / d->h.fsg[k]) / std::log(d->h.fsg[k + 1] / d->h.fsg[k - 1]) : 0.;
        }
        SML_HIP(hipMemcpy(d->aux, lv, sizeof lv, hipMemcpyHostToDevice));
    }
    if (!rc) rc = zeros(&d->tend_grid, (size_t)NB_GRID * GR);
    if (!rc) rc = zeros(&d->tend_spec, (size_t)NB_GRID * SP);
    if (!rc) {
        int32_t kc[NB_SPEC][4], sc[NB_GRID];
        for (int f = 0; f < NB_SPEC; ++f) {        // (type, src0, src1, kcos), see sml_spectral_grid_derived
            const int k = f & 7;
            if (f < 32) { kc[f][0] = 0; kc[f][1] = kc[f][2] = f; kc[f][3] = 1; }                              // vor div t tr: grid(.,.,1)
            else if (f < 48) { kc[f][0] = f < 40 ? 1 : 2; kc[f][1] = F_VOR + k; kc[f][2] = F_DIV + k; kc[f][3] = 2; }   // u, v
            else { kc[f][0] = f == 48 ? 3 : 4; kc[f][1] = kc[f][2] = F_PS; kc[f][3] = 2; }                     // grad(ps)
        }
        for (int f = 0; f < NB_GRID; ++f) sc[f] = f < 48 ? 1 : 0;      // vdspec(.,.,2) scales by cosgr (src/spe_spectral.f90:429-443)
        rc = sml::dev_upload(&d->desc, &kc[0][0], NB_SPEC * 4);
        if (!rc) { d->allocs.push_back(d->desc); rc = sml::dev_upload(&d->scale, sc, NB_GRID); }
        if (!rc) d->allocs.push_back(d->scale);
        // the same set addressed from the start of the state for j2 = 1 | 2, followed by phypar's set of time level 1:
        // u(kx) v(kx) (uvspec, kcos 2) | t q (kcos 1) | phi (type 7) | ps
        static int32_t kp[2][NB_ALL][4];
        for (int j2 = 0; j2 < 2; ++j2) {
            for (int f = 0; f < NB_SPEC; ++f) {
                kp[j2][f][0] = kc[f][0]; kp[j2][f][1] = kc[f][1] + j2 * NSTATE; kp[j2][f][2] = kc[f][2] + j2 * NSTATE; kp[j2][f][3] = kc[f][3];
            }
            for (int f = 0; f < NB_PHYS; ++f) {
                int32_t *r = kp[j2][NB_SPEC + f];
                const int k = (f - 2) & 7;
                if (f < 2) { r[0] = f == 0 ? 1 : 2; r[1] = F_VOR + KX - 1; r[2] = F_DIV + KX - 1; r[3] = 2; }
                else if (f < 10) { r[0] = 0; r[1] = r[2] = F_T + k; r[3] = 1; }
                else if (f < 18) { r[0] = 0; r[1] = r[2] = F_TR + k; r[3] = 1; }
                else if (f < 26) { r[0] = 7; r[1] = F_T; r[2] = k; r[3] = 1; }
                else { r[0] = 0; r[1] = r[2] = F_PS; r[3] = 1; }
            }
        }
        if (!rc) rc = sml::dev_upload(&d->desc_phys, &kp[0][0][0], 2 * NB_ALL * 4);
        if (!rc) d->allocs.push_back(d->desc_phys);
        for (int j2 = 0; j2 < 2; ++j2)
            for (int f = 18; f < 26; ++f) kp[j2][NB_SPEC + f][0] = 8;          // phi level (f - 18) as k_spectral left it in aux
        if (!rc) rc = sml::dev_upload(&d->desc_phys_phi, &kp[0][0][0], 2 * NB_ALL * 4);
        if (!rc) d->allocs.push_back(d->desc_phys_phi);
    }
    if (rc) { sml_dyn_destroy(d); return rc; }
    *out = d;
    return SML_OK;
}

int sml_dyn_destroy(sml_dyn *d)
{
    if (!d) return SML_OK;
    for (void *p : d->allocs) (void)hipFree(p);
    for (ImpSlot *s : d->slots) {
        if (s->d_h) (void)hipFree(s->d_h);
        if (s->d_x) (void)hipFree(s->d_x);
        delete s;
    }
    delete d;
    return SML_OK;
}

int sml_dyn_impint(sml_dyn *d, double dt, double alph)
{
    SML_REQUIRE(d, "sml_dyn_impint: null handle");
    for (ImpSlot *s : d->slots)
        if (s->dt == dt && s->alph == alph) { d->cur = s; return SML_OK; }
    ImpSlot *s = new ImpSlot;
    build_impint(*s, d->h, dt, alph);
    std::vector<double> hh((size_t)4 * NX * MX), xx((size_t)128 + LMAX * 64 + LV_ROWS * KX);
    memcpy(&hh[0], s->dmp1, sizeof s->dmp1);
    memcpy(&hh[NX * MX], s->dmp1d, sizeof s->dmp1d);
    memcpy(&hh[2 * NX * MX], s->dmp1s, sizeof s->dmp1s);
    memcpy(&hh[3 * NX * MX], s->elz, sizeof s->elz);
    memcpy(&xx[0], s->xd, sizeof s->xd);
    memcpy(&xx[64], s->xc, sizeof s->xc);
    memcpy(&xx[128], s->xj, sizeof s->xj);
    {
        const LevelTables &lv = s->lv;
        const double *rows[LV_ROWS] = {lv.dhs, lv.dhsr, lv.xgeop1, lv.xgeop2, lv.tcorv, lv.qcorv, lv.tref, lv.tref1, lv.tref2, lv.tref3, lv.dhsx, lv.corf,
                                       lv.fsgr};
        for (int r = 0; r < LV_ROWS; ++r) memcpy(&xx[128 + LMAX * 64 + r * KX], rows[r], KX * sizeof(double));   // order = LV_* in k_spectral
    }
    int rc = sml::dev_upload(&s->d_h, hh.data(), hh.size());
    if (!rc) rc = sml::dev_upload(&s->d_x, xx.data(), xx.size());
    if (rc) { if (s->d_h) (void)hipFree(s->d_h); delete s; return rc; }
    d->slots.push_back(s);
    d->cur = s;
    return SML_OK;
}

int sml_dyn_get_table(sml_dyn *d, int which, double *out, int capacity)
{
    SML_REQUIRE(d && out, "sml_dyn_get_table: bad arguments");
    const HostDyn &h = d->h;
    const ImpSlot *s = d->cur;
    const double *src = nullptr;
    int n = 0;
    const double half = 0.5;
    SML_REQUIRE(which < 12 || which == 15 || which == 16 || which == 26 || s, "sml_dyn_get_table: table %d needs sml_dyn_impint first", which);
    switch (which) {
    case 1: src = h.hsg; n = KXP; break;             case 2: src = h.dhs; n = KX; break;
    case 3: src = h.fsg; n = KX; break;              case 4: src = h.dhsr; n = KX; break;
    case 5: src = h.fsgr; n = KX; break;             case 6: src = h.coriol; n = IL; break;
    case 7: src = h.xgeop1; n = KX; break;           case 8: src = h.xgeop2; n = KX; break;
    case 9: src = &h.dmp[0][0]; n = NX * MX; break;  case 10: src = &h.dmpd[0][0]; n = NX * MX; break;
    case 11: src = &h.dmps[0][0]; n = NX * MX; break;
    case 12: src = &s->dmp1[0][0]; n = NX * MX; break;  case 13: src = &s->dmp1d[0][0]; n = NX * MX; break;
    case 14: src = &s->dmp1s[0][0]; n = NX * MX; break;
    case 15: src = h.tcorv; n = KX; break;           case 16: src = h.qcorv; n = KX; break;
    case 17: src = s->tref; n = KX; break;           case 18: src = s->tref1; n = KX; break;
    case 19: src = s->tref2; n = KX; break;          case 20: src = s->tref3; n = KX; break;
    case 21: src = &s->xc[0][0]; n = KX * KX; break; case 22: src = &s->xd[0][0]; n = KX * KX; break;
    case 23: src = &s->xj[0][0][0]; n = KX * KX * LMAX; break;
    case 24: src = s->dhsx; n = KX; break;           case 25: src = &s->elz[0][0]; n = NX * MX; break;
    case 26: src = &half; n = 1; break;              /* alph as indyns sets it (ini_indyns.f90:34) */
    default: return sml::fail(SML_ERR_ARG, "sml_dyn_get_table: unknown table %d", which);
    }
    SML_REQUIRE(capacity >= n, "sml_dyn_get_table: capacity %d < %d", capacity, n);
    memcpy(out, src, sizeof(double) * n);
    return n;
}

int sml_dyn_set_boundary(sml_dyn *d, const double *phis_dev, const double *tcorh_dev, const double *qcorh_dev, void *stream)
{
    SML_REQUIRE(d && phis_dev && tcorh_dev && qcorh_dev, "sml_dyn_set_boundary: bad arguments");
    hipStream_t st = sml::as_stream(stream);
    SML_HIP(hipMemcpyAsync(d->bc, phis_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
    SML_HIP(hipMemcpyAsync(d->aux + 24, phis_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
    SML_HIP(hipMemcpyAsync(d->bc + SP, tcorh_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
    SML_HIP(hipMemcpyAsync(d->bc + 2 * SP, qcorh_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
    return SML_OK;
}

SML_SPAN_ATTACH(sml_span_attach_dyn)

int sml_dyn_kspec_stamps(unsigned long long *out16)       // not part of the C-ABI: KSTAMP records of k_spectral (-DSML_KSPEC_STAMPS builds)
{
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_kspec_dbg), sizeof(unsigned long long) * 16));
    return SML_OK;
}

int sml_dyn_phys_stamps(unsigned long long *out48)        // not part of the C-ABI: CSTAMP records of k_gridtend_physics (-DSML_PHYS_STAMPS builds)
{
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpyFromSymbol(out48, HIP_SYMBOL(smlphys::g_phys_dbg), sizeof(unsigned long long) * 48));
    return SML_OK;
}

int sml_dyn_debug_stamps(unsigned long long *out)        // not part of the C-ABI (no declaration in include/): phase profiling aid
{
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 64));
    return SML_OK;
}

int sml_dyn_state_dev(sml_dyn *d, double **state_dev)
{
    SML_REQUIRE(d && state_dev, "sml_dyn_state_dev: bad arguments");
    *state_dev = d->own_state;
    return SML_OK;
}

// Host arrays in the reference's own shapes (src/mod_dynvar.f90:14-27): complex vor/div/t(mx,nx,kx,2), ps(mx,nx,2),
// tr(mx,nx,kx,2,ntr) of which the first tracer is used.  One contiguous block of 8 levels per field and time level.
static int copy_state(sml_dyn *d, double *vor, double *div, double *t, double *ps, double *tr, bool to_device)
{
    double *host3[4] = {vor, div, t, tr};
    const int off3[4] = {F_VOR, F_DIV, F_T, F_TR};
    for (int j = 0; j < 2; ++j) {
        double *lev = d->own_state + (size_t)j * NSTATE * SP;
        for (int f = 0; f < 4; ++f) {
            double *h = host3[f] + (size_t)j * KX * SP, *g = lev + (size_t)off3[f] * SP;
            SML_HIP(hipMemcpy(to_device ? g : h, to_device ? h : g, (size_t)KX * SP * sizeof(double), to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
        }
        double *h = ps + (size_t)j * SP, *g = lev + (size_t)F_PS * SP;
        SML_HIP(hipMemcpy(to_device ? g : h, to_device ? h : g, (size_t)SP * sizeof(double), to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    }
    return SML_OK;
}

int sml_dyn_set_state_host(sml_dyn *d, const double *vor, const double *div, const double *t, const double *ps, const double *tr)
{
    SML_REQUIRE(d && vor && div && t && ps && tr, "sml_dyn_set_state_host: bad arguments");
    return copy_state(d, const_cast<double *>(vor), const_cast<double *>(div), const_cast<double *>(t), const_cast<double *>(ps),
                      const_cast<double *>(tr), true);
}

int sml_dyn_get_state_host(sml_dyn *d, double *vor, double *div, double *t, double *ps, double *tr)
{
    SML_REQUIRE(d && vor && div && t && ps && tr, "sml_dyn_get_state_host: bad arguments");
    SML_HIP(hipDeviceSynchronize());
    return copy_state(d, vor, div, t, ps, tr, false);
}

double *sml_dyn_boundary_dev(sml_dyn *d) { return d ? d->bc : nullptr; }

int sml_dyn_set_boundary_host(sml_dyn *d, const double *phis, const double *tcorh, const double *qcorh)
{
    SML_REQUIRE(d && phis && tcorh && qcorh, "sml_dyn_set_boundary_host: bad arguments");
    SML_HIP(hipMemcpy(d->bc, phis, SP * sizeof(double), hipMemcpyHostToDevice));
    SML_HIP(hipMemcpy(d->aux + 24, phis, SP * sizeof(double), hipMemcpyHostToDevice));
    SML_HIP(hipMemcpy(d->bc + SP, tcorh, SP * sizeof(double), hipMemcpyHostToDevice));
    SML_HIP(hipMemcpy(d->bc + 2 * SP, qcorh, SP * sizeof(double), hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_dyn_grtend(sml_dyn *d, const double *state_dev, int j2, double *tend_dev, void *stream)
{
    SML_REQUIRE(d && state_dev && tend_dev && (j2 == 1 || j2 == 2), "sml_dyn_grtend: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_grtend: call sml_dyn_impint first (tref enters the grid-point tendencies)");
    StepArgs a = make_args(1, j2, 0., 0., 0., 0.);
    return run_step(d, const_cast<double *>(state_dev), a, 1, tend_dev, sml::as_stream(stream), d->lradsw);
}

int sml_dyn_spectral_step(sml_dyn *d, double *state_dev, double *tend_dev, int j1, int j2, double dt, double alph, double rob, double wil,
                          void *stream)
{
    SML_REQUIRE(d && state_dev && tend_dev && (j1 == 1 || j1 == 2) && (j2 == 1 || j2 == 2), "sml_dyn_spectral_step: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_spectral_step: call sml_dyn_impint first");
    StepArgs a = make_args(j1, j2, dt, alph, rob, wil);
    hipLaunchKernelGGL(k_spectral<false>, dim3(SP / 8), dim3(64), 0, sml::as_stream(stream), d->d, d->cur->lv, a, (const double *)nullptr,
                       (const double *)tend_dev, tend_dev, 0, state_dev, d->cur->d_h, d->cur->d_x, d->bc, d->bc + SP, d->bc + 2 * SP, (double *)nullptr);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_dyn_step(sml_dyn *d, double *state_dev, int j1, int j2, double dt, double alph, double rob, double wil, void *stream)
{
    SML_REQUIRE(d && state_dev && (j1 == 1 || j1 == 2) && (j2 == 1 || j2 == 2), "sml_dyn_step: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_step: call sml_dyn_impint first");
    return run_step(d, state_dev, make_args(j1, j2, dt, alph, rob, wil), 0, nullptr, sml::as_stream(stream), d->lradsw);
}

int sml_dyn_attach_physics(sml_dyn *d, sml_phys *phys, int nstrad)
{
    SML_REQUIRE(d && nstrad >= 1, "sml_dyn_attach_physics: bad arguments");
    d->phys = phys;
    d->nstrad = nstrad;
    d->lradsw = 1;
    return SML_OK;
}

int sml_dyn_physics_diag(sml_dyn *d, int on)
{
    SML_REQUIRE(d, "sml_dyn_physics_diag: null handle");
    d->phys_diag = on ? 1 : 0;
    return SML_OK;
}

int sml_dyn_set_range_guard(sml_dyn *d, int32_t *safe_dev)
{
    SML_REQUIRE(d, "sml_dyn_set_range_guard: null handle");
    d->guard = safe_dev;
    return SML_OK;
}

int sml_dyn_set_lradsw(sml_dyn *d, int lradsw)
{
    SML_REQUIRE(d, "sml_dyn_set_lradsw: null handle");
    d->lradsw = lradsw ? 1 : 0;
    return SML_OK;
}

int sml_dyn_select_physics_form(int fused)
{
    g_physics_fused = (fused == 2 || fused == 3) ? fused : fused ? 1 : 0;      // 3: three wavefronts (default); 1: two; 2: one (regression subjects)
    return SML_OK;
}

int sml_dyn_select_window_form(int form)
{
    SML_REQUIRE(form >= -1 && form <= 1, "sml_dyn_select_window_form: -1 default, 0 four launches per step, 1 two kernels per step");
    g_window_form = form;
    return SML_OK;
}

int sml_dyn_window(sml_dyn *d, double *state_dev, int start, int nsteps, double delt, double alph, double rob, double wil, void *stream)
{
    SML_REQUIRE(d && state_dev && nsteps >= 0 && delt > 0, "sml_dyn_window: bad arguments");
    hipStream_t st = sml::as_stream(stream);
    static const int env_two = getenv("SML_DYN_TWO_KERNEL") ? atoi(getenv("SML_DYN_TWO_KERNEL")) : 0;
    const int four_launch = !(g_window_form < 0 ? env_two : g_window_form);
    // the schedule: stepone for istart = 0 or 2 (src/ini_stepone.f90:16-31), then impint(2 delt) (:34) and the leapfrog loop
    // (src/dyn_stloop.f90:28-43)
    struct Item { double dt_imp; StepArgs a; int lradsw; };
    std::vector<Item> sched;
    // short-wave flag: stepone runs with whatever the module variable holds -- .true. at start-up (mod_lflags.f90:22), the last
    // leapfrog step's value when a later window restarts -- then mod(istep, nstrad) == 1 with istep = 1, 2, ...
    // (dyn_stloop.f90:39, at_gcm.f90:81)
    if (start) {
        sched.push_back({0.5 * delt, make_args(1, 1, 0.5 * delt, alph, rob, wil), d->lradsw});
        sched.push_back({delt, make_args(1, 2, delt, alph, rob, wil), d->lradsw});
    }
    for (int i = 0; i < nsteps; ++i) sched.push_back({2 * delt, make_args(2, 2, 2 * delt, alph, rob, wil), (i + 1) % d->nstrad == 1});
    int rc = SML_OK;
    SML_REQUIRE(four_launch || !d->phys, "sml_dyn_window: the two-kernel form has no physics hook; select form 0");
    if (four_launch) {
        for (size_t i = 0; i < sched.size() && !rc; ++i) {
            rc = sml_dyn_impint(d, sched[i].dt_imp, alph);
            // (within one window every step integrates, so step i > 0 finds the geopotential of its time level 1 where step i - 1's
            // spectral kernel left it; the first step -- after a hand-off, or a host that touched the state -- rebuilds it)
            // iogrid(30)'s range guard looks at the first step's inverse set: inside that step's grid-point kernel (three-wavefront form),
            // else in a launch of its own behind the step, while the set is still in batch_grid
            const bool guard_now = i == 0 && start && d->guard, guard_inside = guard_now && d->phys && g_physics_fused == 3;
            if (!rc) rc = run_step(d, state_dev, sched[i].a, 0, nullptr, st, sched[i].lradsw, i > 0, i + 1 < sched.size(), guard_inside ? d->guard : nullptr);
            if (!rc && guard_now && !guard_inside) {
                hipLaunchKernelGGL(k_range_guard, dim3((32 * GR + 255) / 256), dim3(256), 0, st, (const double *)d->batch_grid, d->guard);
                SML_HIP(hipGetLastError());
            }
        }
        if (nsteps > 0) d->lradsw = sched.back().lradsw;
        if (!rc) rc = sml_dyn_impint(d, 2 * delt, alph);
        return rc;
    }
    // two kernels per time step (see k_mspace / k_latspace); one extra k_mspace launch starts the pipeline with the inverse
    // side of the first step
    const double *phis = d->bc, *tcorh = d->bc + SP, *qcorh = d->bc + 2 * SP;
    for (size_t i = 0; i < sched.size() && !rc; ++i) {
        rc = sml_dyn_impint(d, sched[i].dt_imp, alph);
        if (rc) break;
        const ImpSlot *imp = d->cur;
        if (i == 0) {
            hipLaunchKernelGGL(k_mspace, dim3(MX), dim3(MS_THREADS), MS_LDS, st, d->d, d->win, sched[0].a, imp->lv.sdrag, 0, sched[0].a.j2,
                               (const double *)d->fb, d->fa, state_dev, (const double *)imp->d_h, (const double *)imp->d_x, phis, tcorh, qcorh);
            SML_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_latspace, dim3(IL), dim3(LS_THREADS), LS_LDS, st, d->d, d->win, imp->lv.akap,
                           (const double *)(imp->d_x + 128 + LMAX * 64), (const double *)d->fa, d->fb);
        SML_HIP(hipGetLastError());
        const int next_j2 = i + 1 < sched.size() ? sched[i + 1].a.j2 : 0;
        hipLaunchKernelGGL(k_mspace, dim3(MX), dim3(MS_THREADS), MS_LDS, st, d->d, d->win, sched[i].a, imp->lv.sdrag, 1, next_j2,
                           (const double *)d->fb, d->fa, state_dev, (const double *)imp->d_h, (const double *)imp->d_x, phis, tcorh, qcorh);
        SML_HIP(hipGetLastError());
    }
    if (!rc) rc = sml_dyn_impint(d, 2 * delt, alph);
    return rc;
}

}  // extern "C"
