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

namespace {

constexpr int IX = 96, IY = 24, IL = 48, NX = 32, MX = 31, MX2 = 62, KX = 8, KXP = 9, LMAX = 61;
constexpr int SP = MX2 * NX;       // 1984
constexpr int GR = IX * IL;        // 4608
constexpr int NSTATE = 33;         // fields per time level
constexpr int F_VOR = 0, F_DIV = 8, F_T = 16, F_TR = 24, F_PS = 32;
constexpr int NB_SPEC = 50;        // inverse batch: the 32 3-D state fields, ucos(8), vcos(8), d(ps)/dx, d(ps)/dy
constexpr int NB_GRID = 73;        // forward batch, see k_gridtend

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
    double *d_x = nullptr;         // device: xd(64) | xc(64) | xj(61*64) | level tables (12 x 8, rows LV_*)
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
constexpr int LV_DHS = 0, LV_DHSR = 1, LV_XG1 = 2, LV_XG2 = 3, LV_TCORV = 4, LV_QCORV = 5, LV_TREF = 6, LV_TREF1 = 7, LV_TREF2 = 8,
              LV_TREF3 = 9, LV_DHSX = 10, LV_CORF = 11, LV_ROWS = 12;

template <bool FROM_FLUX>
__global__ __launch_bounds__(64) void k_spectral(DevHoriz H, LevelTables L, StepArgs a, const double *__restrict__ S,
                                                  const double *tend_in, double *tend_out, int stop_after_grtend,
                                                  double *__restrict__ state, const double *__restrict__ imp_h,
                                                  const double *__restrict__ imp_x, const double *__restrict__ phis,
                                                  const double *__restrict__ tcorh, const double *__restrict__ qcorh)
{
    __shared__ double lv[LV_ROWS][KX];          // level tables, indexed by a lane-varying level
    __shared__ double xdc[2][KX * KX];          // xd, xc
    __shared__ double d4[KX][8], t4[KX][8];     // div and t of time level j4, [level][coefficient]
    __shared__ double tds[KX][8], yfs[KX][8], dvs[KX][8];
    const int tid = threadIdx.x, ci = tid & 7, k = tid >> 3;
    const int e = blockIdx.x * 8 + ci;
    const int c = e % MX2, n = e / MX2, m = c >> 1, hm = n * MX + m;
    {   // level tables (indexed by a lane-varying level below) and the two dense 8x8 matrices -> LDS
        const double *lvg = imp_x + 128 + LMAX * 64;
        double *lvf = &lv[0][0];
        for (int i = tid; i < LV_ROWS * KX; i += 64) lvf[i] = lvg[i];
        xdc[0][tid] = imp_x[tid];
        xdc[1][tid] = imp_x[64 + tid];
    }
    double vordt, divdt, tdt, trdt, psdt;
    if (FROM_FLUX) {
        const double gx = H.gradx[m], ym = H.vddym[hm], yp = H.vddyp[hm], el2 = H.el2[hm];
        double dummy;
        stencil(S + (size_t)k * SP, S + (size_t)(8 + k) * SP, n, c, gx, ym, yp, vordt, divdt);
        const double lapke = -S[(size_t)(48 + k) * SP + e] * el2;
        divdt = divdt - lapke;
        stencil(S + (size_t)(16 + k) * SP, S + (size_t)(24 + k) * SP, n, c, gx, ym, yp, dummy, tdt);
        tdt = tdt + S[(size_t)(56 + k) * SP + e];
        stencil(S + (size_t)(32 + k) * SP, S + (size_t)(40 + k) * SP, n, c, gx, ym, yp, dummy, trdt);
        trdt = trdt + S[(size_t)(64 + k) * SP + e];
        psdt = S[(size_t)72 * SP + e];
        if (e < 2) psdt = 0.;
    } else {
        vordt = tend_in[(size_t)(F_VOR + k) * SP + e];
        divdt = tend_in[(size_t)(F_DIV + k) * SP + e];
        tdt = tend_in[(size_t)(F_T + k) * SP + e];
        trdt = tend_in[(size_t)(F_TR + k) * SP + e];
        psdt = tend_in[(size_t)F_PS * SP + e];
    }
    double *s1 = state, *s2 = state + (size_t)NSTATE * SP;
    if (!stop_after_grtend) {
        // ---- sptend (src/dyn_sptend.f90) on time level j4
        const double *s4 = a.j4 == 1 ? s1 : s2;
        d4[k][ci] = s4[(size_t)(F_DIV + k) * SP + e];
        t4[k][ci] = s4[(size_t)(F_T + k) * SP + e];
        const double ps4 = s4[(size_t)F_PS * SP + e];
        const double el2 = H.el2[hm];
        __syncthreads();
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
        double phi = phis[e] + lv[LV_XG1][KX - 1] * t4[KX - 1][ci];
#pragma unroll
        for (int j = KX - 2; j >= 0; --j)
            if (j >= k) phi = phi + lv[LV_XG2][j + 1] * t4[j + 1][ci] + lv[LV_XG1][j] * t4[j][ci];
        if (c < 2 && k >= 1 && k <= KX - 2) phi = phi + lv[LV_CORF][k] * (t4[k + 1][ci] - t4[k - 1][ci]);
        {
            const double g1 = phi + lv[LV_TREF1][k] * ps4;
            const double g2 = -g1 * el2;
            divdt = divdt - g2;
        }
        // ---- implic (src/dyn_implic.f90)
        if (a.implicit) {
            const double elz = imp_h[3 * NX * MX + hm];
            const int ll = m + n;
            double xl[KX];
#pragma unroll
            for (int k1 = 0; k1 < KX; ++k1) xl[k1] = ll != 0 ? imp_x[128 + (size_t)(ll - 1) * 64 + k1 * KX + k] : 0.0;
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
        // ---- horizontal diffusion (src/dyn_step.f90:60-104, hordif :130-150), always on time level 1
        const double dmp = H.dmp[hm], dmpd = H.dmpd[hm], dmps = H.dmps[hm];
        const double dmp1 = imp_h[hm], dmp1d = imp_h[NX * MX + hm], dmp1s = imp_h[2 * NX * MX + hm];
        const double v1 = s1[(size_t)(F_VOR + k) * SP + e], d1 = s1[(size_t)(F_DIV + k) * SP + e];
        const double ct = s1[(size_t)(F_T + k) * SP + e] + tcorh[e] * lv[LV_TCORV][k];
        const double cq = s1[(size_t)(F_TR + k) * SP + e] + qcorh[e] * lv[LV_QCORV][k];
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
    const double tf = H.trfilt[hm];
    auto timint = [&](int f, double fdt) {
        fdt = fdt * tf;
        const size_t o = (size_t)f * SP + e;
        const double f1 = s1[o];
        const double fj = a.j1 == 1 ? f1 : s2[o];
        const double fnew = f1 + a.dt * fdt;
        const double n1 = fj + a.wil * a.eps * (f1 - 2 * fj + fnew);
        const double fj_after = a.j1 == 1 ? n1 : fj;          // field(.,1) is overwritten before field(.,2) is formed
        s1[o] = n1;
        s2[o] = fnew - (1 - a.wil) * a.eps * (n1 - 2 * fj_after + fnew);
    };
    // every level-thread of a coefficient has read its time-level-1 diffusion operands above; a column's ps is only
    // touched by its k == 0 thread
    timint(F_VOR + k, vordt);
    timint(F_DIV + k, divdt);
    timint(F_T + k, tdt);
    timint(F_TR + k, trdt);
    if (k == 0) timint(F_PS, psdt);
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
    double *batch_grid = nullptr;      // [50][GR]
    double *tend_grid = nullptr;       // [73][GR]
    double *tend_spec = nullptr;       // [73][SP]
    int32_t *desc = nullptr, *scale = nullptr;     // inverse-batch descriptors [50][4], forward-batch scaling flags [73]
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

int run_step(sml_dyn *d, double *state, const StepArgs &a, int stop_after_grtend, double *tend_out, hipStream_t st)
{
    const double *sj2 = state + (size_t)(a.j2 - 1) * NSTATE * SP;
    // the 50 inverse transforms of grtend (:61-99) straight from the state: uvspec and grad are formed while the fields are staged
    int rc = sml_spectral_grid_derived(d->sp, sj2, d->desc, d->batch_grid, NB_SPEC, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gridtend, dim3(GR / 64), dim3(64), 0, st, d->d, d->cur->lv, d->batch_grid, d->tend_grid);
    SML_HIP(hipGetLastError());
    rc = sml_spectral_spec_mixed(d->sp, d->tend_grid, d->tend_spec, NB_GRID, d->scale, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_spectral<true>, dim3(SP / 8), dim3(64), 0, st, d->d, d->cur->lv, a, d->tend_spec, (const double *)nullptr,
                       tend_out, stop_after_grtend, state, d->cur->d_h, d->cur->d_x, d->bc, d->bc + SP, d->bc + 2 * SP);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

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
    auto zeros = [&](double **p, size_t n) { int r = sml::dev_zeros(p, n); if (!r) d->allocs.push_back(*p); return r; };
    if (!rc) rc = zeros(&d->bc, (size_t)3 * SP);
    if (!rc) rc = zeros(&d->own_state, (size_t)2 * NSTATE * SP);
    if (!rc) rc = zeros(&d->batch_grid, (size_t)NB_SPEC * GR);
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
    std::vector<double> hh((size_t)4 * NX * MX), xx((size_t)128 + LMAX * 64 + 12 * KX);
    memcpy(&hh[0], s->dmp1, sizeof s->dmp1);
    memcpy(&hh[NX * MX], s->dmp1d, sizeof s->dmp1d);
    memcpy(&hh[2 * NX * MX], s->dmp1s, sizeof s->dmp1s);
    memcpy(&hh[3 * NX * MX], s->elz, sizeof s->elz);
    memcpy(&xx[0], s->xd, sizeof s->xd);
    memcpy(&xx[64], s->xc, sizeof s->xc);
    memcpy(&xx[128], s->xj, sizeof s->xj);
    {
        const LevelTables &lv = s->lv;
        const double *rows[12] = {lv.dhs, lv.dhsr, lv.xgeop1, lv.xgeop2, lv.tcorv, lv.qcorv, lv.tref, lv.tref1, lv.tref2, lv.tref3, lv.dhsx, lv.corf};
        for (int r = 0; r < 12; ++r) memcpy(&xx[128 + LMAX * 64 + r * KX], rows[r], KX * sizeof(double));   // order = LV_* in k_spectral
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
    SML_HIP(hipMemcpyAsync(d->bc + SP, tcorh_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
    SML_HIP(hipMemcpyAsync(d->bc + 2 * SP, qcorh_dev, SP * sizeof(double), hipMemcpyDeviceToDevice, st));
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

int sml_dyn_set_boundary_host(sml_dyn *d, const double *phis, const double *tcorh, const double *qcorh)
{
    SML_REQUIRE(d && phis && tcorh && qcorh, "sml_dyn_set_boundary_host: bad arguments");
    SML_HIP(hipMemcpy(d->bc, phis, SP * sizeof(double), hipMemcpyHostToDevice));
    SML_HIP(hipMemcpy(d->bc + SP, tcorh, SP * sizeof(double), hipMemcpyHostToDevice));
    SML_HIP(hipMemcpy(d->bc + 2 * SP, qcorh, SP * sizeof(double), hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_dyn_grtend(sml_dyn *d, const double *state_dev, int j2, double *tend_dev, void *stream)
{
    SML_REQUIRE(d && state_dev && tend_dev && (j2 == 1 || j2 == 2), "sml_dyn_grtend: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_grtend: call sml_dyn_impint first (tref enters the grid-point tendencies)");
    StepArgs a = make_args(1, j2, 0., 0., 0., 0.);
    return run_step(d, const_cast<double *>(state_dev), a, 1, tend_dev, sml::as_stream(stream));
}

int sml_dyn_spectral_step(sml_dyn *d, double *state_dev, double *tend_dev, int j1, int j2, double dt, double alph, double rob, double wil,
                          void *stream)
{
    SML_REQUIRE(d && state_dev && tend_dev && (j1 == 1 || j1 == 2) && (j2 == 1 || j2 == 2), "sml_dyn_spectral_step: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_spectral_step: call sml_dyn_impint first");
    StepArgs a = make_args(j1, j2, dt, alph, rob, wil);
    hipLaunchKernelGGL(k_spectral<false>, dim3(SP / 8), dim3(64), 0, sml::as_stream(stream), d->d, d->cur->lv, a, (const double *)nullptr,
                       (const double *)tend_dev, tend_dev, 0, state_dev, d->cur->d_h, d->cur->d_x, d->bc, d->bc + SP, d->bc + 2 * SP);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_dyn_step(sml_dyn *d, double *state_dev, int j1, int j2, double dt, double alph, double rob, double wil, void *stream)
{
    SML_REQUIRE(d && state_dev && (j1 == 1 || j1 == 2) && (j2 == 1 || j2 == 2), "sml_dyn_step: bad arguments");
    SML_REQUIRE(d->cur, "sml_dyn_step: call sml_dyn_impint first");
    return run_step(d, state_dev, make_args(j1, j2, dt, alph, rob, wil), 0, nullptr, sml::as_stream(stream));
}

int sml_dyn_window(sml_dyn *d, double *state_dev, int start, int nsteps, double delt, double alph, double rob, double wil, void *stream)
{
    SML_REQUIRE(d && state_dev && nsteps >= 0 && delt > 0, "sml_dyn_window: bad arguments");
    hipStream_t st = sml::as_stream(stream);
    int rc = SML_OK;
    if (start) {                                   // stepone, istart = 0 or 2 (src/ini_stepone.f90:16-31)
        rc = sml_dyn_impint(d, 0.5 * delt, alph);
        if (!rc) rc = run_step(d, state_dev, make_args(1, 1, 0.5 * delt, alph, rob, wil), 0, nullptr, st);
        if (!rc) rc = sml_dyn_impint(d, delt, alph);
        if (!rc) rc = run_step(d, state_dev, make_args(1, 2, delt, alph, rob, wil), 0, nullptr, st);
    }
    if (!rc) rc = sml_dyn_impint(d, 2 * delt, alph);         // :34
    for (int i = 0; i < nsteps && !rc; ++i)                    // src/dyn_stloop.f90:28-43
        rc = run_step(d, state_dev, make_args(2, 2, 2 * delt, alph, rob, wil), 0, nullptr, st);
    return rc;
}

}  // extern "C"
