/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the SPEEDY spectral transforms.
 * See sml_oracle.h for scope and pinning status.  PINNED against oracle/_ref/libref_spectral.so
 * (the reference's own spe_spectral.f90 / spe_subfft_fftpack*.f90 compiled in place) by
 * tests/test_oracle_spectral.py and the committed fixtures tests/golden/spectral_*.npz.
 *
 * Everything is fp64: the reference is built with -r8/-fdefault-real-8 (src/makefile:6,12) so every
 * `real` and every default-real literal is a double; `3.141592654d0` and `eps=3.d-14` stay as written
 * (src/spe_spectral.f90:16,24).  Index convention: Fortran a(m,n) -> C a[n-1][m-1].
 *
 * The longitudinal FFT is restated from FFTPACK's published *definition* of rfftf/rfftb (half-complex
 * layout, unnormalised backward transform) as a direct 96-point DFT with an exact-argument twiddle table;
 * it agrees with the vendored mixed-radix code (src/spe_subfft_fftpack2.f90, factors 2*4*4*3) to rounding.
 */
#include "sml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

so_tables *so_tables_new(void) { return (so_tables *)calloc(1, sizeof(so_tables)); }
void so_tables_free(so_tables *t) { free(t); }

/* src/spe_spectral.f90:2-43 */
static void gaussl(double *x, double *w, int m)
{
    const double eps = 3.e-14;
    int n = 2 * m;
    double z1 = 2.0;
    for (int i = 1; i <= m; ++i) {
        double z = cos(3.141592654 * (i - .25) / (n + .5));
        double pp = 0.0;
        while (fabs(z - z1) > eps) {
            double p1 = 1.0, p2 = 0.0, p3;
            for (int j = 1; j <= n; ++j) {
                p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        }
        x[i - 1] = z;
        w[i - 1] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

/* src/spe_spectral.f90:194-242 ; alp(mxp,nx) -> alp[n][m], poly(m,n)=alp(m,n) for isc=1 */
static void lgndre(const so_tables *t, int j, double poly[SO_NX][SO_MX])
{
    const double small = 1.e-30;
    static double alp[SO_NX][SO_MXP];
    double y = t->coa[j], x = t->sia[j];
    alp[0][0] = t->sqrhlf;
    for (int m = 2; m <= SO_MXP; ++m) alp[0][m - 1] = t->consq[m - 1] * y * alp[0][m - 2];
    for (int m = 1; m <= SO_MXP; ++m) alp[1][m - 1] = (x * alp[0][m - 1]) * t->repsi[1][m - 1];
    for (int n = 3; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MXP; ++m)
            alp[n - 1][m - 1] = (x * alp[n - 2][m - 1] - t->epsi[n - 2][m - 1] * alp[n - 3][m - 1]) * t->repsi[n - 1][m - 1];
    for (int n = 0; n < SO_NX; ++n)
        for (int m = 0; m < SO_MXP; ++m)
            if (fabs(alp[n][m]) <= small) alp[n][m] = 0.0;
    for (int n = 0; n < SO_NX; ++n)
        for (int m = 0; m < SO_MX; ++m) poly[n][m] = alp[n][m];
}

/* src/spe_spectral.f90:45-192 */
void so_parmtr(so_tables *t, double a)
{
    t->a = a;
    gaussl(t->sia, t->wt, SO_IY);
    double am2 = 1. / (a * a);
    for (int j = 0; j < SO_IY; ++j) {
        double cosqr = 1.0 - t->sia[j] * t->sia[j];
        t->coa[j] = sqrt(cosqr);
        t->wght[j] = t->wt[j] / (a * cosqr);
    }
    for (int j = 0; j < SO_IY; ++j) {
        int jj = SO_IL - 1 - j;
        t->cosg[j] = t->cosg[jj] = t->coa[j];
        t->cosgr[j] = t->cosgr[jj] = 1. / t->coa[j];
        t->cosgr2[j] = t->cosgr2[jj] = 1. / (t->coa[j] * t->coa[j]);
    }
    static int ll[SO_NX][SO_MX];
    for (int n = 1; n <= SO_NX; ++n) {
        t->nsh2[n - 1] = 0;
        for (int m = 1; m <= SO_MX; ++m) {
            int mm = m - 1;
            int l = mm + n - 1;
            ll[n - 1][m - 1] = l;
            int l2 = l * (l + 1);
            t->el2[n - 1][m - 1] = (double)l2 * am2;
            t->el4[n - 1][m - 1] = t->el2[n - 1][m - 1] * t->el2[n - 1][m - 1];
            if (l <= SO_NTRUN1 || SO_IX != 4 * SO_IY) t->nsh2[n - 1] += 2;
            t->trfilt[n - 1][m - 1] = (l <= SO_NTRUN) ? 1. : 0.;
        }
    }
    t->elm2[0][0] = 0.;
    for (int m = 2; m <= SO_MX; ++m)
        for (int n = 1; n <= SO_NX; ++n) t->elm2[n - 1][m - 1] = 1. / t->el2[n - 1][m - 1];
    for (int n = 2; n <= SO_NX; ++n) t->elm2[n - 1][0] = 1. / t->el2[n - 1][0];

    for (int m = 1; m <= SO_MXP; ++m)
        for (int n = 1; n <= SO_NXP; ++n) {
            double emm = (double)(m - 1), ell = (double)(n + m - 2);
            double emm2 = emm * emm, ell2 = ell * ell, e;
            if (n == SO_NXP) e = 0.0;
            else if (n == 1 && m == 1) e = 0.0;
            else e = sqrt((ell2 - emm2) / (4. * ell2 - 1.));
            t->epsi[n - 1][m - 1] = e;
            t->repsi[n - 1][m - 1] = (e > 0.) ? 1. / e : 0.0;
        }
    t->sqrhlf = sqrt(.5);
    t->consq[0] = 0.0;
    for (int m = 2; m <= SO_MXP; ++m) {
        double emm = (double)(m - 1);
        t->consq[m - 1] = sqrt(.5 * (2. * emm + 1.) / emm);
    }
    for (int m = 1; m <= SO_MX; ++m)
        for (int n = 1; n <= SO_NX; ++n) {
            int m1 = m - 1;                  /* mm(m) */
            int m2 = m1 + 1;                 /* 1-based row into epsi */
            double el1 = (double)ll[n - 1][m - 1];
            if (n == 1) {
                t->gradx[m - 1] = (double)m1 / a;
                t->uvdx[0][m - 1] = -a / (double)(m1 + 1);
                t->uvdym[0][m - 1] = 0.0;
                t->vddym[0][m - 1] = 0.0;
                t->gradym[0][m - 1] = 0.0;   /* never assigned by the reference; unused */
            } else {
                t->uvdx[n - 1][m - 1] = -a * (double)m1 / (el1 * (el1 + 1));
                t->gradym[n - 1][m - 1] = (el1 - 1.) * t->epsi[n - 1][m2 - 1] / a;
                t->uvdym[n - 1][m - 1] = -a * t->epsi[n - 1][m2 - 1] / el1;
                t->vddym[n - 1][m - 1] = (el1 + 1) * t->epsi[n - 1][m2 - 1] / a;
            }
            t->gradyp[n - 1][m - 1] = (el1 + 2.) * t->epsi[n][m2 - 1] / a;
            t->uvdyp[n - 1][m - 1] = -a * t->epsi[n][m2 - 1] / (el1 + 1.);
            t->vddyp[n - 1][m - 1] = el1 * t->epsi[n][m2 - 1] / a;
        }
    static double poly[SO_NX][SO_MX];
    for (int j = 0; j < SO_IY; ++j) {
        lgndre(t, j, poly);
        for (int n = 0; n < SO_NX; ++n)
            for (int m = 0; m < SO_MX; ++m) {
                t->cpol[j][n][2 * m] = poly[n][m];
                t->cpol[j][n][2 * m + 1] = poly[n][m];
            }
    }
    /* twiddles for the DFT restatement of FFTPACK (exact quadrant symmetry, long double argument) */
    for (int k = 0; k < SO_IX; ++k) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)SO_IX;
        t->dftc[k] = (double)cosl(ang);
        t->dfts[k] = (double)sinl(ang);
    }
    t->dftc[24] = 0.0; t->dftc[72] = 0.0; t->dfts[0] = 0.0; t->dfts[48] = 0.0;
}

void so_get_table(const so_tables *t, int which, double *out)
{
    const double *src = 0; int n = 0;
    switch (which) {
    case 1: src = t->sia; n = SO_IY; break;
    case 2: src = t->coa; n = SO_IY; break;
    case 3: src = t->wt; n = SO_IY; break;
    case 4: src = t->wght; n = SO_IY; break;
    case 5: src = t->cosg; n = SO_IL; break;
    case 6: src = t->cosgr; n = SO_IL; break;
    case 7: src = t->cosgr2; n = SO_IL; break;
    case 8: src = &t->el2[0][0]; n = SO_MX * SO_NX; break;
    case 9: src = &t->elm2[0][0]; n = SO_MX * SO_NX; break;
    case 10: src = &t->el4[0][0]; n = SO_MX * SO_NX; break;
    case 11: src = &t->trfilt[0][0]; n = SO_MX * SO_NX; break;
    case 12: for (int i = 0; i < SO_NX; ++i) out[i] = (double)t->nsh2[i]; return;
    case 13: src = &t->epsi[0][0]; n = SO_MXP * SO_NXP; break;
    case 14: src = &t->repsi[0][0]; n = SO_MXP * SO_NXP; break;
    case 15: src = t->consq; n = SO_MXP; break;
    case 16: src = t->gradx; n = SO_MX; break;
    case 17: src = &t->gradym[0][0]; n = SO_MX * SO_NX; break;
    case 18: src = &t->gradyp[0][0]; n = SO_MX * SO_NX; break;
    case 19: src = &t->uvdx[0][0]; n = SO_MX * SO_NX; break;
    case 20: src = &t->uvdym[0][0]; n = SO_MX * SO_NX; break;
    case 21: src = &t->uvdyp[0][0]; n = SO_MX * SO_NX; break;
    case 22: src = &t->vddym[0][0]; n = SO_MX * SO_NX; break;
    case 23: src = &t->vddyp[0][0]; n = SO_MX * SO_NX; break;
    case 24: src = &t->cpol[0][0][0]; n = SO_MX2 * SO_NX * SO_IY; break;
    case 26: out[0] = t->sqrhlf; return;
    default: return;
    }
    memcpy(out, src, (size_t)n * sizeof(double));
}

/* FFTPACK rfftf: r(1)=sum x; r(2k)=sum x cos(2 pi k i/n); r(2k+1)=-sum x sin(2 pi k i/n); r(n)=sum (-1)^i x */
void so_rfftf(const so_tables *t, double *r)
{
    double x[SO_IX];
    memcpy(x, r, sizeof x);
    double s0 = 0.0;
    for (int i = 0; i < SO_IX; ++i) s0 += x[i];
    r[0] = s0;
    for (int k = 1; k < SO_IX / 2; ++k) {
        double re = 0.0, im = 0.0;
        for (int i = 0; i < SO_IX; ++i) {
            int ph = (k * i) % SO_IX;
            re += x[i] * t->dftc[ph];
            im -= x[i] * t->dfts[ph];
        }
        r[2 * k - 1] = re;
        r[2 * k] = im;
    }
    double sn = 0.0;
    for (int i = 0; i < SO_IX; ++i) sn += (i & 1) ? -x[i] : x[i];
    r[SO_IX - 1] = sn;
}

/* FFTPACK rfftb (unnormalised): x_i = r1 + sum_k 2[r(2k) cos - r(2k+1) sin] + (-1)^i r(n) */
void so_rfftb(const so_tables *t, double *r)
{
    double c[SO_IX];
    memcpy(c, r, sizeof c);
    for (int i = 0; i < SO_IX; ++i) {
        double acc = c[0];
        for (int k = 1; k < SO_IX / 2; ++k) {
            int ph = (k * i) % SO_IX;
            acc += 2.0 * (c[2 * k - 1] * t->dftc[ph] - c[2 * k] * t->dfts[ph]);
        }
        acc += (i & 1) ? -c[SO_IX - 1] : c[SO_IX - 1];
        r[i] = acc;
    }
}

/* src/spe_spectral.f90:454-495 */
void so_gridy(const so_tables *t, const double *v, double *varm)
{
    double vm1[SO_MX2], vm2[SO_MX2];
    for (int j = 0; j < SO_IY; ++j) {
        int j1 = SO_IL - 1 - j;
        for (int m = 0; m < SO_MX2; ++m) vm1[m] = vm2[m] = 0.;
        for (int n = 1; n <= SO_NX; n += 2)
            for (int m = 0; m < t->nsh2[n - 1]; ++m) vm1[m] = vm1[m] + v[(n - 1) * SO_MX2 + m] * t->cpol[j][n - 1][m];
        for (int n = 2; n <= SO_NX; n += 2)
            for (int m = 0; m < t->nsh2[n - 1]; ++m) vm2[m] = vm2[m] + v[(n - 1) * SO_MX2 + m] * t->cpol[j][n - 1][m];
        for (int m = 0; m < SO_MX2; ++m) {
            varm[j1 * SO_MX2 + m] = vm1[m] + vm2[m];
            varm[j * SO_MX2 + m] = vm1[m] - vm2[m];
        }
    }
}

/* src/spe_subfft_fftpack.f90:15-51 */
void so_gridx(const so_tables *t, const double *varm, double *vorg, int kcos)
{
    double fvar[SO_IX];
    for (int j = 0; j < SO_IL; ++j) {
        fvar[0] = varm[j * SO_MX2 + 0];
        for (int m = 3; m <= SO_MX2; ++m) fvar[m - 2] = varm[j * SO_MX2 + m - 1];
        for (int m = SO_MX2; m <= SO_IX; ++m) fvar[m - 1] = 0.0;
        so_rfftb(t, fvar);
        if (kcos == 1) for (int i = 0; i < SO_IX; ++i) vorg[j * SO_IX + i] = fvar[i];
        else           for (int i = 0; i < SO_IX; ++i) vorg[j * SO_IX + i] = fvar[i] * t->cosgr[j];
    }
}

/* src/spe_subfft_fftpack.f90:55-87 */
void so_specx(const so_tables *t, const double *vorg, double *varm)
{
    double fvar[SO_IX];
    for (int j = 0; j < SO_IL; ++j) {
        memcpy(fvar, vorg + j * SO_IX, sizeof fvar);
        so_rfftf(t, fvar);
        double scale = 1. / (double)SO_IX;
        varm[j * SO_MX2 + 0] = fvar[0] * scale;
        varm[j * SO_MX2 + 1] = 0.0;
        for (int m = 3; m <= SO_MX2; ++m) varm[j * SO_MX2 + m - 1] = fvar[m - 2] * scale;
    }
}

/* src/spe_spectral.f90:497-538 (accumulation j-outer, n-inner as in the reference) */
void so_specy(const so_tables *t, const double *varm, double *vorm)
{
    static double svarm[SO_IY][SO_MX2], dvarm[SO_IY][SO_MX2];
    for (int i = 0; i < SO_MX2 * SO_NX; ++i) vorm[i] = 0.0;
    for (int j = 0; j < SO_IY; ++j) {
        int j1 = SO_IL - 1 - j;
        for (int m = 0; m < SO_MX2; ++m) {
            svarm[j][m] = (varm[j1 * SO_MX2 + m] + varm[j * SO_MX2 + m]) * t->wt[j];
            dvarm[j][m] = (varm[j1 * SO_MX2 + m] - varm[j * SO_MX2 + m]) * t->wt[j];
        }
    }
    for (int j = 0; j < SO_IY; ++j) {
        for (int n = 1; n <= SO_NTRUN1; n += 2)
            for (int m = 0; m < t->nsh2[n - 1]; ++m)
                vorm[(n - 1) * SO_MX2 + m] = vorm[(n - 1) * SO_MX2 + m] + t->cpol[j][n - 1][m] * svarm[j][m];
        for (int n = 2; n <= SO_NTRUN1; n += 2)
            for (int m = 0; m < t->nsh2[n - 1]; ++m)
                vorm[(n - 1) * SO_MX2 + m] = vorm[(n - 1) * SO_MX2 + m] + t->cpol[j][n - 1][m] * dvarm[j][m];
    }
}

void so_grid(const so_tables *t, const double *vorm, double *vorg, int kcos)
{
    double varm[SO_MX2 * SO_IL];
    so_gridy(t, vorm, varm);
    so_gridx(t, varm, vorg, kcos);
}

void so_spec(const so_tables *t, const double *vorg, double *vorm)
{
    double varm[SO_MX2 * SO_IL];
    so_specx(t, vorg, varm);
    so_specy(t, varm, vorm);
}

/* complex helpers on (re,im)-interleaved spectral arrays: element (k,m,n) -> [n*62 + 2*m + k] */
#define SP(a, k, m, n) (a)[((n) - 1) * SO_MX2 + 2 * ((m) - 1) + ((k) - 1)]

/* src/spe_spectral.f90:307-349 */
void so_vds(const so_tables *t, const double *ucosm, const double *vcosm, double *vorm, double *divm)
{
    static double zc[SO_MX2 * SO_NX], zp[SO_MX2 * SO_NX];
    for (int n = 1; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(zp, 2, m, n) = t->gradx[m - 1] * SP(ucosm, 1, m, n);
            SP(zp, 1, m, n) = -t->gradx[m - 1] * SP(ucosm, 2, m, n);
            SP(zc, 2, m, n) = t->gradx[m - 1] * SP(vcosm, 1, m, n);
            SP(zc, 1, m, n) = -t->gradx[m - 1] * SP(vcosm, 2, m, n);
        }
    for (int k = 1; k <= 2; ++k)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(vorm, k, m, 1) = SP(zc, k, m, 1) - t->vddyp[0][m - 1] * SP(ucosm, k, m, 2);
            SP(vorm, k, m, SO_NX) = t->vddym[SO_NX - 1][m - 1] * SP(ucosm, k, m, SO_NTRUN1);
            SP(divm, k, m, 1) = SP(zp, k, m, 1) + t->vddyp[0][m - 1] * SP(vcosm, k, m, 2);
            SP(divm, k, m, SO_NX) = -t->vddym[SO_NX - 1][m - 1] * SP(vcosm, k, m, SO_NTRUN1);
        }
    for (int k = 1; k <= 2; ++k)
        for (int n = 2; n <= SO_NTRUN1; ++n)
            for (int m = 1; m <= SO_MX; ++m) {
                SP(vorm, k, m, n) = t->vddym[n - 1][m - 1] * SP(ucosm, k, m, n - 1) - t->vddyp[n - 1][m - 1] * SP(ucosm, k, m, n + 1) + SP(zc, k, m, n);
                SP(divm, k, m, n) = -t->vddym[n - 1][m - 1] * SP(vcosm, k, m, n - 1) + t->vddyp[n - 1][m - 1] * SP(vcosm, k, m, n + 1) + SP(zp, k, m, n);
            }
}

/* src/spe_spectral.f90:351-387 */
void so_uvspec(const so_tables *t, const double *vorm, const double *divm, double *ucosm, double *vcosm)
{
    static double zc[SO_MX2 * SO_NX], zp[SO_MX2 * SO_NX];
    for (int n = 1; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(zp, 2, m, n) = t->uvdx[n - 1][m - 1] * SP(vorm, 1, m, n);
            SP(zp, 1, m, n) = -t->uvdx[n - 1][m - 1] * SP(vorm, 2, m, n);
            SP(zc, 2, m, n) = t->uvdx[n - 1][m - 1] * SP(divm, 1, m, n);
            SP(zc, 1, m, n) = -t->uvdx[n - 1][m - 1] * SP(divm, 2, m, n);
        }
    for (int k = 1; k <= 2; ++k)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(ucosm, k, m, 1) = SP(zc, k, m, 1) - t->uvdyp[0][m - 1] * SP(vorm, k, m, 2);
            SP(ucosm, k, m, SO_NX) = t->uvdym[SO_NX - 1][m - 1] * SP(vorm, k, m, SO_NTRUN1);
            SP(vcosm, k, m, 1) = SP(zp, k, m, 1) + t->uvdyp[0][m - 1] * SP(divm, k, m, 2);
            SP(vcosm, k, m, SO_NX) = -t->uvdym[SO_NX - 1][m - 1] * SP(divm, k, m, SO_NTRUN1);
        }
    for (int k = 1; k <= 2; ++k)
        for (int n = 2; n <= SO_NTRUN1; ++n)
            for (int m = 1; m <= SO_MX; ++m) {
                SP(vcosm, k, m, n) = -t->uvdym[n - 1][m - 1] * SP(divm, k, m, n - 1) + t->uvdyp[n - 1][m - 1] * SP(divm, k, m, n + 1) + SP(zp, k, m, n);
                SP(ucosm, k, m, n) = t->uvdym[n - 1][m - 1] * SP(vorm, k, m, n - 1) - t->uvdyp[n - 1][m - 1] * SP(vorm, k, m, n + 1) + SP(zc, k, m, n);
            }
}

/* src/spe_spectral.f90:271-305 */
void so_grad(const so_tables *t, const double *psi, double *psdx, double *psdy)
{
    for (int n = 1; n <= SO_NX; ++n)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(psdx, 2, m, n) = t->gradx[m - 1] * SP(psi, 1, m, n);
            SP(psdx, 1, m, n) = -t->gradx[m - 1] * SP(psi, 2, m, n);
        }
    for (int k = 1; k <= 2; ++k)
        for (int m = 1; m <= SO_MX; ++m) {
            SP(psdy, k, m, 1) = t->gradyp[0][m - 1] * SP(psi, k, m, 2);
            SP(psdy, k, m, SO_NX) = -t->gradym[SO_NX - 1][m - 1] * SP(psi, k, m, SO_NTRUN1);
        }
    for (int k = 1; k <= 2; ++k)
        for (int n = 2; n <= SO_NTRUN1; ++n)
            for (int m = 1; m <= SO_MX; ++m)
                SP(psdy, k, m, n) = -t->gradym[n - 1][m - 1] * SP(psi, k, m, n - 1) + t->gradyp[n - 1][m - 1] * SP(psi, k, m, n + 1);
}

/* src/spe_spectral.f90:244-269, 540-551 : complex * real table */
void so_lap(const so_tables *t, const double *strm, double *vorm)
{
    for (int n = 0; n < SO_NX; ++n)
        for (int c = 0; c < SO_MX2; ++c) vorm[n * SO_MX2 + c] = -strm[n * SO_MX2 + c] * t->el2[n][c / 2];
}
void so_invlap(const so_tables *t, const double *vorm, double *strm)
{
    for (int n = 0; n < SO_NX; ++n)
        for (int c = 0; c < SO_MX2; ++c) strm[n * SO_MX2 + c] = -vorm[n * SO_MX2 + c] * t->elm2[n][c / 2];
}
void so_trunct(const so_tables *t, double *vor)
{
    for (int n = 0; n < SO_NX; ++n)
        for (int c = 0; c < SO_MX2; ++c) vor[n * SO_MX2 + c] = vor[n * SO_MX2 + c] * t->trfilt[n][c / 2];
}

/* src/spe_spectral.f90:416-452 */
void so_vdspec(const so_tables *t, const double *ug, const double *vg, double *vorm, double *divm, int kcos)
{
    static double ug1[SO_IX * SO_IL], vg1[SO_IX * SO_IL], um[SO_MX2 * SO_IL], vm[SO_MX2 * SO_IL];
    static double dumc1[SO_MX2 * SO_NX], dumc2[SO_MX2 * SO_NX];
    const double *sc = (kcos == 2) ? t->cosgr : t->cosgr2;
    for (int j = 0; j < SO_IL; ++j)
        for (int i = 0; i < SO_IX; ++i) {
            ug1[j * SO_IX + i] = ug[j * SO_IX + i] * sc[j];
            vg1[j * SO_IX + i] = vg[j * SO_IX + i] * sc[j];
        }
    so_specx(t, ug1, um);
    so_specx(t, vg1, vm);
    so_specy(t, um, dumc1);
    so_specy(t, vm, dumc2);
    so_vds(t, dumc1, dumc2, vorm, divm);
}
