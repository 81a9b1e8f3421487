/* TEST INFRASTRUCTURE ONLY -- CPU restatement of SPEEDY's adiabatic dynamical core.
 *
 * PINNED against the compiled reference (oracle/_ref/libref_dyn.so: the reference's own ini_indyns.f90, ini_impint.f90,
 * spe_matinv.f90, dyn_geop.f90, dyn_sptend.f90, dyn_implic.f90 and dyn_step.f90 compiled in place) for: the indyns and
 * impint tables, geop, sptend, implic, hordif and timint (tests/test_oracle_dynamics.py, fixtures
 * tests/golden/dynamics_golden.npz).
 * NOT pinned by a runnable reference: do_grtend_dry -- src/dyn_grtend.f90 calls phypar (the column physics, out of scope and
 * not built here), so the reference routine cannot be executed; its grid-point algebra is restated statement by statement
 * with the physics call omitted (adiabatic core), every transform it uses IS pinned (spectral_oracle.c), and
 * tests check invariants (state of rest, solid-body rotation, mass conservation).
 *
 * Layout: Fortran complex a(mx,nx,kx[,2]) is stored as interleaved doubles, element (m,n,k,j) -> [((j*8+k)*32+n)*62+2m+ri]
 * (each level is a spectral field in the layout of spectral_oracle.c).  Grid fields are [48][96].
 * Promoted-precision constants as in src/mod_dyncon0.f90, mod_dyncon1.f90, mod_tsteps.f90.
 */
#include "sml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { KX = 8, KXP = 9, LMAX = 61, SP = SO_MX2 * SO_NX, GR = SO_IX * SO_IL };

static const double REARTH = 6.371e+6, OMEGA = 7.292e-05, GRAV = 9.81, AKAP = 2. / 7., GAMMA_ = 6.0, HSCALE = 7.5, HSHUM = 2.5,
                    THD = 2.4, THDD = 2.4, THDS = 12.0, TDRS = 24.0 * 30.0;
#define RGAS (AKAP * 1004.)

typedef struct do_tables {
    double hsg[KXP], dhs[KX], fsg[KX], dhsr[KX], fsgr[KX], coriol[SO_IL], xgeop1[KX], xgeop2[KX];
    double dmp[SO_NX][SO_MX], dmpd[SO_NX][SO_MX], dmps[SO_NX][SO_MX], dmp1[SO_NX][SO_MX], dmp1d[SO_NX][SO_MX], dmp1s[SO_NX][SO_MX];
    double tcorv[KX], qcorv[KX], tref[KX], tref1[KX], tref2[KX], tref3[KX];
    double xc[KX][KX], xd[KX][KX];           /* Fortran x(k,k1) -> x[k1][k] */
    double xj[LMAX][KX][KX];                 /* xj(k,k1,l)   -> xj[l][k1][k] */
    double dhsx[KX], elz[SO_NX][SO_MX], alph;
} do_tables;

do_tables *do_tables_new(void) { return (do_tables *)calloc(1, sizeof(do_tables)); }
void do_tables_free(do_tables *t) { free(t); }

/* src/ini_indyns.f90 (parmtr is so_parmtr) */
void do_indyns(do_tables *d, const so_tables *s)
{
    static const double hsg8[9] = {0.000, 0.050, 0.140, 0.260, 0.420, 0.600, 0.770, 0.900, 1.000};
    d->alph = 0.5;
    const int npowhd = 4;
    memcpy(d->hsg, hsg8, sizeof hsg8);
    for (int k = 0; k < KX; ++k) { d->dhs[k] = d->hsg[k + 1] - d->hsg[k]; d->fsg[k] = 0.5 * (d->hsg[k + 1] + d->hsg[k]); }
    for (int k = 0; k < KX; ++k) { d->dhsr[k] = 0.5 / d->dhs[k]; d->fsgr[k] = AKAP / (2. * d->fsg[k]); }
    for (int j = 0; j < SO_IY; ++j) {
        int jj = SO_IL - 1 - j;
        double gs_s = -s->sia[j], gs_n = s->sia[j];
        d->coriol[j] = 2. * OMEGA * gs_s;
        d->coriol[jj] = 2. * OMEGA * gs_n;
    }
    for (int k = 0; k < KX; ++k) {
        d->xgeop1[k] = RGAS * log(d->hsg[k + 1] / d->fsg[k]);
        if (k != KX - 1) d->xgeop2[k + 1] = RGAS * log(d->fsg[k + 1] / d->hsg[k + 1]);
    }
    d->xgeop2[0] = 0.0;
    double hdiff = 1. / (THD * 3600.), hdifd = 1. / (THDD * 3600.), hdifs = 1. / (THDS * 3600.);
    double rlap = 1. / (double)(SO_NTRUN * (SO_NTRUN + 1));
    for (int j = 1; j <= SO_NX; ++j)
        for (int k = 1; k <= SO_MX; ++k) {
            double twn = (double)((k - 1) + j - 1);
            double elap = (twn * (twn + 1.) * rlap);
            double elapn = elap * elap * elap * elap;          /* elap**npowhd, npowhd = 4 (integer power) */
            (void)npowhd;
            d->dmp[j - 1][k - 1] = hdiff * elapn;
            d->dmpd[j - 1][k - 1] = hdifd * elapn;
            d->dmps[j - 1][k - 1] = hdifs * elap;
        }
    double rgam = RGAS * GAMMA_ / (1000. * GRAV), qexp = HSCALE / HSHUM;
    d->tcorv[0] = 0.; d->qcorv[0] = 0.; d->qcorv[1] = 0.;
    for (int k = 2; k <= KX; ++k) {
        d->tcorv[k - 1] = pow(d->fsg[k - 1], rgam);
        if (k > 2) d->qcorv[k - 1] = pow(d->fsg[k - 1], qexp);
    }
}

/* src/spe_matinv.f90: ludcmp (implicit-scaling Crout, Numerical Recipes), lubksb, inv.  a(n,n) column-major. */
static void nr_inv(double *a, double *y, int n)
{
    int indx[16];
    double vv[16];
#define A(i, j) a[((j) - 1) * n + ((i) - 1)]
    for (int i = 1; i <= n; ++i) {
        double aamax = 0.;
        for (int j = 1; j <= n; ++j) if (fabs(A(i, j)) > aamax) aamax = fabs(A(i, j));
        vv[i - 1] = 1. / aamax;
    }
    for (int j = 1; j <= n; ++j) {
        int imax = j;
        for (int i = 1; i <= j - 1; ++i) {
            double sum = A(i, j);
            if (i > 1) { for (int k = 1; k <= i - 1; ++k) sum = sum - A(i, k) * A(k, j); A(i, j) = sum; }
        }
        double aamax = 0.;
        for (int i = j; i <= n; ++i) {
            double sum = A(i, j);
            if (j > 1) { for (int k = 1; k <= j - 1; ++k) sum = sum - A(i, k) * A(k, j); A(i, j) = sum; }
            double dum = vv[i - 1] * fabs(sum);
            if (dum >= aamax) { imax = i; aamax = dum; }
        }
        if (j != imax) {
            for (int k = 1; k <= n; ++k) { double dum = A(imax, k); A(imax, k) = A(j, k); A(j, k) = dum; }
            vv[imax - 1] = vv[j - 1];
        }
        indx[j - 1] = imax;
        if (j != n) { double dum = 1. / A(j, j); for (int i = j + 1; i <= n; ++i) A(i, j) = A(i, j) * dum; }
    }
    for (int c = 0; c < n; ++c) {
        double *b = y + c * n;
        for (int i = 0; i < n; ++i) b[i] = (i == c) ? 1. : 0.;
        int ii = 0;
        for (int i = 1; i <= n; ++i) {
            int ll = indx[i - 1];
            double sum = b[ll - 1];
            b[ll - 1] = b[i - 1];
            if (ii != 0) { for (int j = ii; j <= i - 1; ++j) sum = sum - A(i, j) * b[j - 1]; }
            else if (sum != 0.) ii = i;
            b[i - 1] = sum;
        }
        for (int i = n; i >= 1; --i) {
            double sum = b[i - 1];
            if (i < n) for (int j = i + 1; j <= n; ++j) sum = sum - A(i, j) * b[j - 1];
            b[i - 1] = sum / A(i, i);
        }
    }
#undef A
}

/* src/ini_impint.f90 */
void do_impint(do_tables *d, double dt, double alph)
{
    double xa[KX][KX], xb[KX][KX], ya[KX][KX], xe[KX][KX], xcl[KX][KX], dsum[KX];   /* local, [k1][k] */
    memset(xa, 0, sizeof xa); memset(xb, 0, sizeof xb);
    for (int n = 0; n < SO_NX; ++n)
        for (int m = 0; m < SO_MX; ++m) {
            d->dmp1[n][m] = 1. / (1. + d->dmp[n][m] * dt);
            d->dmp1d[n][m] = 1. / (1. + d->dmpd[n][m] * dt);
            d->dmp1s[n][m] = 1. / (1. + d->dmps[n][m] * dt);
        }
    double rgam = RGAS * GAMMA_ / (1000. * GRAV);
    for (int k = 0; k < KX; ++k) {
        double f = d->fsg[k] > 0.2 ? d->fsg[k] : 0.2;
        d->tref[k] = 288. * pow(f, rgam);
        d->tref1[k] = RGAS * d->tref[k];
        d->tref2[k] = AKAP * d->tref[k];
        d->tref3[k] = d->fsgr[k] * d->tref[k];
    }
    double xi = dt * alph, xxi = xi / (REARTH * REARTH);
    for (int k = 0; k < KX; ++k) d->dhsx[k] = xi * d->dhs[k];
    for (int n = 1; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MX; ++m) {
            int ll = m + n - 2;
            d->elz[n - 1][m - 1] = (double)ll * (double)(ll + 1) * xxi;
        }
#define X(a, k, k1) a[(k1) - 1][(k) - 1]
    for (int k = 1; k <= KX; ++k) for (int k1 = 1; k1 <= KX; ++k1) X(ya, k, k1) = -AKAP * d->tref[k - 1] * d->dhs[k1 - 1];
    for (int k = 2; k <= KX; ++k) X(xa, k, k - 1) = 0.5 * (AKAP * d->tref[k - 1] / d->fsg[k - 1] - (d->tref[k - 1] - d->tref[k - 2]) / d->dhs[k - 1]);
    for (int k = 1; k <= KX - 1; ++k) X(xa, k, k) = 0.5 * (AKAP * d->tref[k - 1] / d->fsg[k - 1] - (d->tref[k] - d->tref[k - 1]) / d->dhs[k - 1]);
    dsum[0] = d->dhs[0];
    for (int k = 2; k <= KX; ++k) dsum[k - 1] = dsum[k - 2] + d->dhs[k - 1];
    for (int k = 1; k <= KX - 1; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            X(xb, k, k1) = d->dhs[k1 - 1] * dsum[k - 1];
            if (k1 <= k) X(xb, k, k1) = X(xb, k, k1) - d->dhs[k1 - 1];
        }
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            X(xcl, k, k1) = X(ya, k, k1);
            for (int k2 = 1; k2 <= KX - 1; ++k2) X(xcl, k, k1) = X(xcl, k, k1) + X(xa, k, k2) * X(xb, k2, k1);
        }
    memset(d->xd, 0, sizeof d->xd);
    for (int k = 1; k <= KX; ++k) for (int k1 = k + 1; k1 <= KX; ++k1) X(d->xd, k, k1) = RGAS * log(d->hsg[k1] / d->hsg[k1 - 1]);
    for (int k = 1; k <= KX; ++k) X(d->xd, k, k) = RGAS * log(d->hsg[k] / d->fsg[k - 1]);
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            X(xe, k, k1) = 0.;
            for (int k2 = 1; k2 <= KX; ++k2) X(xe, k, k1) = X(xe, k, k1) + X(d->xd, k, k2) * X(xcl, k2, k1);
        }
    for (int l = 1; l <= LMAX; ++l) {
        double xf[KX][KX];
        double xxx = ((double)l * (double)(l + 1)) / (REARTH * REARTH);
        for (int k = 1; k <= KX; ++k)
            for (int k1 = 1; k1 <= KX; ++k1) X(xf, k, k1) = xi * xi * xxx * (RGAS * d->tref[k - 1] * d->dhs[k1 - 1] - X(xe, k, k1));
        for (int k = 1; k <= KX; ++k) X(xf, k, k) = X(xf, k, k) + 1.;
        nr_inv(&xf[0][0], &d->xj[l - 1][0][0], KX);
    }
    for (int k = 1; k <= KX; ++k) for (int k1 = 1; k1 <= KX; ++k1) X(d->xc, k, k1) = X(xcl, k, k1) * xi;
#undef X
}

void do_get_table(const do_tables *d, int which, double *out)
{
    const double *src = 0; int n = 0;
    switch (which) {
    case 1: src = d->hsg; n = KXP; break;      case 2: src = d->dhs; n = KX; break;
    case 3: src = d->fsg; n = KX; break;       case 4: src = d->dhsr; n = KX; break;
    case 5: src = d->fsgr; n = KX; break;      case 6: src = d->coriol; n = SO_IL; break;
    case 7: src = d->xgeop1; n = KX; break;    case 8: src = d->xgeop2; n = KX; break;
    case 9: src = &d->dmp[0][0]; n = SO_MX * SO_NX; break;    case 10: src = &d->dmpd[0][0]; n = SO_MX * SO_NX; break;
    case 11: src = &d->dmps[0][0]; n = SO_MX * SO_NX; break;  case 12: src = &d->dmp1[0][0]; n = SO_MX * SO_NX; break;
    case 13: src = &d->dmp1d[0][0]; n = SO_MX * SO_NX; break; case 14: src = &d->dmp1s[0][0]; n = SO_MX * SO_NX; break;
    case 15: src = d->tcorv; n = KX; break;    case 16: src = d->qcorv; n = KX; break;
    case 17: src = d->tref; n = KX; break;     case 18: src = d->tref1; n = KX; break;
    case 19: src = d->tref2; n = KX; break;    case 20: src = d->tref3; n = KX; break;
    case 21: src = &d->xc[0][0]; n = KX * KX; break;          case 22: src = &d->xd[0][0]; n = KX * KX; break;
    case 23: src = &d->xj[0][0][0]; n = KX * KX * LMAX; break;
    case 24: src = d->dhsx; n = KX; break;     case 25: src = &d->elz[0][0]; n = SO_MX * SO_NX; break;
    case 26: out[0] = d->alph; return;
    default: return;
    }
    memcpy(out, src, sizeof(double) * n);
}

/* src/dyn_geop.f90: t = level-jj temperature [8][SP], phis [SP] -> phi [8][SP] */
void do_geop(const do_tables *d, const double *t, const double *phis, double *phi)
{
    for (int e = 0; e < SP; ++e) phi[(KX - 1) * SP + e] = phis[e] + d->xgeop1[KX - 1] * t[(KX - 1) * SP + e];
    for (int k = KX - 2; k >= 0; --k)
        for (int e = 0; e < SP; ++e)
            phi[k * SP + e] = phi[(k + 1) * SP + e] + d->xgeop2[k + 1] * t[(k + 1) * SP + e] + d->xgeop1[k] * t[k * SP + e];
    for (int k = 2; k <= KX - 1; ++k) {
        double corf = d->xgeop1[k - 1] * 0.5 * log(d->hsg[k] / d->fsg[k - 1]) / log(d->fsg[k] / d->fsg[k - 2]);
        for (int n = 0; n < SO_NX; ++n)
            for (int ri = 0; ri < 2; ++ri) {        /* phi(1,:,k): zonal wavenumber 0, all n */
                int e = n * SO_MX2 + ri;
                phi[(k - 1) * SP + e] = phi[(k - 1) * SP + e] + corf * (t[k * SP + e] - t[(k - 2) * SP + e]);
            }
    }
}

/* src/dyn_sptend.f90: div,t [8][SP] and ps [SP] at time level j4; tendencies in/out; phi (work/out) [8][SP] */
void do_sptend(const do_tables *d, const so_tables *s, const double *div, const double *t, const double *ps, const double *phis,
               double *divdt, double *tdt, double *psdt, double *phi)
{
    static double dmeanc[SP], sigdtc[KXP][SP], dumk[KXP][SP], dumc1[SP], dumc2[SP];
    memset(dmeanc, 0, sizeof dmeanc);
    for (int k = 0; k < KX; ++k) for (int e = 0; e < SP; ++e) dmeanc[e] = dmeanc[e] + div[k * SP + e] * d->dhs[k];
    for (int e = 0; e < SP; ++e) psdt[e] = psdt[e] - dmeanc[e];
    psdt[0] = 0.; psdt[1] = 0.;
    memset(sigdtc[0], 0, sizeof sigdtc[0]); memset(sigdtc[KX], 0, sizeof sigdtc[KX]);
    for (int k = 0; k < KX - 1; ++k)
        for (int e = 0; e < SP; ++e) sigdtc[k + 1][e] = sigdtc[k][e] - d->dhs[k] * (div[k * SP + e] - dmeanc[e]);
    memset(dumk[0], 0, sizeof dumk[0]); memset(dumk[KX], 0, sizeof dumk[KX]);
    for (int k = 1; k < KX; ++k) for (int e = 0; e < SP; ++e) dumk[k][e] = sigdtc[k][e] * (d->tref[k] - d->tref[k - 1]);
    for (int k = 0; k < KX; ++k)
        for (int e = 0; e < SP; ++e)
            tdt[k * SP + e] = tdt[k * SP + e] - (dumk[k + 1][e] + dumk[k][e]) * d->dhsr[k] + d->tref3[k] * (sigdtc[k + 1][e] + sigdtc[k][e])
                              - d->tref2[k] * dmeanc[e];
    do_geop(d, t, phis, phi);
    for (int k = 0; k < KX; ++k) {
        for (int e = 0; e < SP; ++e) dumc1[e] = phi[k * SP + e] + RGAS * d->tref[k] * ps[e];
        so_lap(s, dumc1, dumc2);
        for (int e = 0; e < SP; ++e) divdt[k * SP + e] = divdt[k * SP + e] - dumc2[e];
    }
}

/* src/dyn_implic.f90 */
void do_implic(const do_tables *d, double *divdt, double *tdt, double *psdt)
{
    static double ye[KX][SP], yf[KX][SP];
    memset(ye, 0, sizeof ye);
    for (int k1 = 0; k1 < KX; ++k1)
        for (int k = 0; k < KX; ++k)
            for (int e = 0; e < SP; ++e) ye[k][e] = ye[k][e] + d->xd[k1][k] * tdt[k1 * SP + e];
    for (int k = 0; k < KX; ++k) for (int e = 0; e < SP; ++e) ye[k][e] = ye[k][e] + d->tref1[k] * psdt[e];
    for (int k = 0; k < KX; ++k)
        for (int n = 0; n < SO_NX; ++n)
            for (int c = 0; c < SO_MX2; ++c) {
                int e = n * SO_MX2 + c;
                yf[k][e] = divdt[k * SP + e] + d->elz[n][c / 2] * ye[k][e];
            }
    for (int e = 0; e < KX * SP; ++e) divdt[e] = 0.;
    for (int n = 1; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MX; ++m) {
            int ll = m + n - 2;
            if (ll != 0)
                for (int k1 = 0; k1 < KX; ++k1)
                    for (int k = 0; k < KX; ++k)
                        for (int ri = 0; ri < 2; ++ri) {
                            int e = (n - 1) * SO_MX2 + 2 * (m - 1) + ri;
                            divdt[k * SP + e] = divdt[k * SP + e] + d->xj[ll - 1][k1][k] * yf[k1][e];
                        }
        }
    for (int k = 0; k < KX; ++k) for (int e = 0; e < SP; ++e) psdt[e] = psdt[e] - divdt[k * SP + e] * d->dhsx[k];
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1)
            for (int e = 0; e < SP; ++e) tdt[k * SP + e] = tdt[k * SP + e] + d->xc[k1][k] * divdt[k1 * SP + e];
}

/* hordif (src/dyn_step.f90:130-150): which = 1 dmp/dmp1, 2 dmpd/dmp1d, 3 dmps/dmp1s */
void do_hordif(const do_tables *d, int nlev, const double *field, double *fdt, int which)
{
    const double(*a)[SO_MX] = which == 1 ? d->dmp : which == 2 ? d->dmpd : d->dmps;
    const double(*b)[SO_MX] = which == 1 ? d->dmp1 : which == 2 ? d->dmp1d : d->dmp1s;
    for (int k = 0; k < nlev; ++k)
        for (int n = 0; n < SO_NX; ++n)
            for (int c = 0; c < SO_MX2; ++c) {
                int e = k * SP + n * SO_MX2 + c;
                fdt[e] = (fdt[e] - a[n][c / 2] * field[e]) * b[n][c / 2];
            }
}

/* timint (src/dyn_step.f90:152-190): field [2][nlev][SP]; fdt is truncated in place as in the reference */
void do_timint(const so_tables *s, int j1, double dt, double eps, double wil, int nlev, double *field, double *fdt)
{
    for (int k = 0; k < nlev; ++k) so_trunct(s, fdt + k * SP);
    double *f1 = field, *f2 = field + (size_t)nlev * SP;
    double *fj1 = (j1 == 1) ? f1 : f2;
    for (int k = 0; k < nlev; ++k)
        for (int e = 0; e < SP; ++e) {
            int i = k * SP + e;
            double fnew = f1[i] + dt * fdt[i];
            double a = fj1[i];
            f1[i] = a + wil * eps * (f1[i] - 2 * a + fnew);
            /* field(.,1) has just been updated; when j1 == 1 field(.,j1) aliases it, exactly as in the Fortran */
            double aj = (j1 == 1) ? f1[i] : f2[i];
            f2[i] = fnew - (1 - wil) * eps * (f1[i] - 2 * aj + fnew);
        }
}

/* grtend (src/dyn_grtend.f90).  State level j2: vor,div,t,tr [8][SP], ps [SP].  The physics call (:222-225: geop(j1);
 * phypar(vor(j1),div(j1),t(j1),tr(j1),phi,ps(j1),utend,vtend,ttend,trtend)) is a hook: when phys != NULL the oracle forms
 * phypar's grid-point inputs from state level 1 exactly as phy_phypar.f90:54-66 does (uvspec + grid(.,2); grid(.,1) of t, q,
 * phi; grid of ps) and hands them to phys, which adds its tendencies in place -- the tests plug the COMPILED reference
 * parametrisations (oracle/_ref/libref_phy.so) in there.  phys == NULL is the adiabatic core. */
void do_grtend(const do_tables *d, const so_tables *s, const double *vor, const double *div, const double *t, const double *tr,
               const double *ps, const double *vor1, const double *div1, const double *t1, const double *tr1, const double *ps1,
               const double *phis, do_phys_fn phys, void *ctx, double *vordt, double *divdt, double *tdt, double *psdt, double *trdt)
{
    static double ug[KX][GR], vg[KX][GR], tg[KX][GR], vorg[KX][GR], divg[KX][GR], tgg[KX][GR], puv[KX][GR], trg[KX][GR];
    static double utend[KX][GR], vtend[KX][GR], ttend[KX][GR], trtend[KX][GR];
    static double px[GR], py[GR], umean[GR], vmean[GR], dmean[GR], sigdt[KXP][GR], temp[KXP][GR], sigm[KXP][GR], dumr[3][GR];
    static double dumc[3][SP];
    for (int k = 0; k < KX; ++k) {
        so_grid(s, vor + k * SP, vorg[k], 1);
        so_grid(s, div + k * SP, divg[k], 1);
        so_grid(s, t + k * SP, tg[k], 1);
        so_grid(s, tr + k * SP, trg[k], 1);
        so_uvspec(s, vor + k * SP, div + k * SP, dumc[0], dumc[1]);
        so_grid(s, dumc[1], vg[k], 2);
        so_grid(s, dumc[0], ug[k], 2);
        for (int j = 0; j < SO_IL; ++j) for (int i = 0; i < SO_IX; ++i) vorg[k][j * SO_IX + i] = vorg[k][j * SO_IX + i] + d->coriol[j];
    }
    for (int p = 0; p < GR; ++p) umean[p] = vmean[p] = dmean[p] = 0.0;
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) {
            umean[p] = umean[p] + ug[k][p] * d->dhs[k];
            vmean[p] = vmean[p] + vg[k][p] * d->dhs[k];
            dmean[p] = dmean[p] + divg[k][p] * d->dhs[k];
        }
    so_grad(s, ps, dumc[1], dumc[2]);
    so_grid(s, dumc[1], px, 2);
    so_grid(s, dumc[2], py, 2);
    for (int p = 0; p < GR; ++p) dumr[0][p] = -umean[p] * px[p] - vmean[p] * py[p];
    so_spec(s, dumr[0], psdt);
    psdt[0] = 0.; psdt[1] = 0.;
    for (int p = 0; p < GR; ++p) { sigdt[0][p] = sigdt[KX][p] = 0.0; sigm[0][p] = sigm[KX][p] = 0.0; }
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) puv[k][p] = (ug[k][p] - umean[p]) * px[p] + (vg[k][p] - vmean[p]) * py[p];
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) {
            sigdt[k + 1][p] = sigdt[k][p] - d->dhs[k] * (puv[k][p] + divg[k][p] - dmean[p]);
            sigm[k + 1][p] = sigm[k][p] - d->dhs[k] * puv[k][p];
        }
    for (int k = 0; k < KX; ++k) for (int p = 0; p < GR; ++p) tgg[k][p] = tg[k][p] - d->tref[k];
    for (int p = 0; p < GR; ++p) { px[p] = RGAS * px[p]; py[p] = RGAS * py[p]; }
    for (int p = 0; p < GR; ++p) temp[0][p] = temp[KX][p] = 0.0;
    for (int k = 1; k < KX; ++k) for (int p = 0; p < GR; ++p) temp[k][p] = sigdt[k][p] * (ug[k][p] - ug[k - 1][p]);
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) utend[k][p] = vg[k][p] * vorg[k][p] - tgg[k][p] * px[p] - (temp[k + 1][p] + temp[k][p]) * d->dhsr[k];
    for (int k = 1; k < KX; ++k) for (int p = 0; p < GR; ++p) temp[k][p] = sigdt[k][p] * (vg[k][p] - vg[k - 1][p]);
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) vtend[k][p] = -ug[k][p] * vorg[k][p] - tgg[k][p] * py[p] - (temp[k + 1][p] + temp[k][p]) * d->dhsr[k];
    for (int k = 1; k < KX; ++k)
        for (int p = 0; p < GR; ++p) temp[k][p] = sigdt[k][p] * (tgg[k][p] - tgg[k - 1][p]) + sigm[k][p] * (d->tref[k] - d->tref[k - 1]);
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p)
            ttend[k][p] = tgg[k][p] * divg[k][p] - (temp[k + 1][p] + temp[k][p]) * d->dhsr[k]
                          + d->fsgr[k] * tgg[k][p] * (sigdt[k + 1][p] + sigdt[k][p]) + d->tref3[k] * (sigm[k + 1][p] + sigm[k][p])
                          + AKAP * (tg[k][p] * puv[k][p] - tgg[k][p] * dmean[p]);
    for (int k = 1; k < KX; ++k) for (int p = 0; p < GR; ++p) temp[k][p] = sigdt[k][p] * (trg[k][p] - trg[k - 1][p]);
    for (int k = 1; k <= 2; ++k) for (int p = 0; p < GR; ++p) temp[k][p] = 0.;       /* do k=2,3 (1-based) */
    for (int k = 0; k < KX; ++k)
        for (int p = 0; p < GR; ++p) trtend[k][p] = trg[k][p] * divg[k][p] - (temp[k + 1][p] + temp[k][p]) * d->dhsr[k];
    if (phys) {          /* :222-225 */
        static double phi1[KX * SP], ug1[KX][GR], vg1[KX][GR], tg1[KX][GR], qg1[KX][GR], phig1[KX][GR], pslg1[GR];
        do_geop(d, t1, phis, phi1);
        for (int k = 0; k < KX; ++k) {
            so_uvspec(s, vor1 + k * SP, div1 + k * SP, dumc[0], dumc[1]);
            so_grid(s, dumc[0], ug1[k], 2);
            so_grid(s, dumc[1], vg1[k], 2);
        }
        for (int k = 0; k < KX; ++k) {
            so_grid(s, t1 + k * SP, tg1[k], 1);
            so_grid(s, tr1 + k * SP, qg1[k], 1);
            so_grid(s, phi1 + k * SP, phig1[k], 1);
        }
        so_grid(s, ps1, pslg1, 1);
        phys(ctx, &ug1[0][0], &vg1[0][0], &tg1[0][0], &qg1[0][0], &phig1[0][0], pslg1, &utend[0][0], &vtend[0][0], &ttend[0][0], &trtend[0][0]);
    }
    for (int k = 0; k < KX; ++k) {
        so_vdspec(s, utend[k], vtend[k], vordt + k * SP, divdt + k * SP, 2);
        for (int p = 0; p < GR; ++p) {
            dumr[0][p] = 0.5 * (ug[k][p] * ug[k][p] + vg[k][p] * vg[k][p]);
            dumr[1][p] = -ug[k][p] * tgg[k][p];
            dumr[2][p] = -vg[k][p] * tgg[k][p];
        }
        so_spec(s, dumr[0], dumc[0]);
        so_lap(s, dumc[0], dumc[1]);
        for (int e = 0; e < SP; ++e) divdt[k * SP + e] = divdt[k * SP + e] - dumc[1][e];
        so_vdspec(s, dumr[1], dumr[2], dumc[0], tdt + k * SP, 2);
        so_spec(s, ttend[k], dumc[1]);
        for (int e = 0; e < SP; ++e) tdt[k * SP + e] = tdt[k * SP + e] + dumc[1][e];
        for (int p = 0; p < GR; ++p) { dumr[1][p] = -ug[k][p] * trg[k][p]; dumr[2][p] = -vg[k][p] * trg[k][p]; }
        so_spec(s, trtend[k], dumc[1]);
        so_vdspec(s, dumr[1], dumr[2], dumc[0], trdt + k * SP, 2);
        for (int e = 0; e < SP; ++e) trdt[k * SP + e] = trdt[k * SP + e] + dumc[1][e];
    }
}

void do_grtend_dry(const do_tables *d, const so_tables *s, const double *vor, const double *div, const double *t, const double *tr,
                   const double *ps, double *vordt, double *divdt, double *tdt, double *psdt, double *trdt)
{
    do_grtend(d, s, vor, div, t, tr, ps, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, vordt, divdt, tdt, psdt, trdt);
}

/* step (src/dyn_step.f90:1-128).  State arrays hold both time levels: vor,div,t,tr [2][8][SP], ps [2][SP]; phis, tcorh,
 * qcorh [SP].  phys: see do_grtend (NULL = adiabatic). */
void do_step(const do_tables *d, const so_tables *s, int j1, int j2, double dt, double alph, double rob, double wil,
             double *vor, double *div, double *t, double *tr, double *ps, const double *phis, const double *tcorh, const double *qcorh,
             do_phys_fn phys, void *ctx)
{
    static double vordt[KX * SP], divdt[KX * SP], tdt[KX * SP], trdt[KX * SP], psdt[SP], ctmp[KX * SP], phi[KX * SP];
    const size_t L = (size_t)KX * SP;
    const double *vor2 = vor + (j2 - 1) * L, *div2 = div + (j2 - 1) * L, *t2 = t + (j2 - 1) * L, *tr2 = tr + (j2 - 1) * L, *ps2 = ps + (j2 - 1) * SP;
    do_grtend(d, s, vor2, div2, t2, tr2, ps2, vor, div, t, tr, ps, phis, phys, ctx, vordt, divdt, tdt, psdt, trdt);
    if (alph == 0.) {
        do_sptend(d, s, div2, t2, ps2, phis, divdt, tdt, psdt, phi);
    } else {
        do_sptend(d, s, div, t, ps, phis, divdt, tdt, psdt, phi);      /* j4 = 1 */
        do_implic(d, divdt, tdt, psdt);
    }
    do_hordif(d, KX, vor, vordt, 1);
    do_hordif(d, KX, div, divdt, 2);
    for (int k = 0; k < KX; ++k) for (int e = 0; e < SP; ++e) ctmp[k * SP + e] = t[k * SP + e] + tcorh[e] * d->tcorv[k];
    do_hordif(d, KX, ctmp, tdt, 1);
    double sdrag = 1. / (TDRS * 3600.);
    for (int n = 0; n < SO_NX; ++n)
        for (int ri = 0; ri < 2; ++ri) {
            int e = n * SO_MX2 + ri;
            vordt[e] = vordt[e] - sdrag * vor[e];
            divdt[e] = divdt[e] - sdrag * div[e];
        }
    do_hordif(d, 1, vor, vordt, 3);
    do_hordif(d, 1, div, divdt, 3);
    do_hordif(d, 1, ctmp, tdt, 3);
    for (int k = 0; k < KX; ++k) for (int e = 0; e < SP; ++e) ctmp[k * SP + e] = tr[k * SP + e] + qcorh[e] * d->qcorv[k];
    do_hordif(d, KX, ctmp, trdt, 2);
    if (dt <= 0.) return;
    double eps = (j1 == 1) ? 0. : rob;
    do_timint(s, j1, dt, eps, wil, 1, ps, psdt);
    do_timint(s, j1, dt, eps, wil, KX, vor, vordt);
    do_timint(s, j1, dt, eps, wil, KX, div, divdt);
    do_timint(s, j1, dt, eps, wil, KX, t, tdt);
    do_timint(s, j1, dt, eps, wil, KX, tr, trdt);
}

void do_step_dry(const do_tables *d, const so_tables *s, int j1, int j2, double dt, double alph, double rob, double wil,
                 double *vor, double *div, double *t, double *tr, double *ps, const double *phis, const double *tcorh, const double *qcorh)
{
    do_step(d, s, j1, j2, dt, alph, rob, wil, vor, div, t, tr, ps, phis, tcorh, qcorh, NULL, NULL);
}
