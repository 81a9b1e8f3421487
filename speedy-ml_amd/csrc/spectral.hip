// SPEEDY T30 spectral transforms on gfx950, batched over fields.
//
// Replaces src/spe_spectral.f90 (parmtr/gaussl/lgndre tables :2-242, operators :244-387, grid/spec/vdspec :389-452,
// Legendre synthesis/analysis gridy/specy :454-538, trunct :540-551) and the longitudinal FFT wrappers
// gridx/specx over vendored FFTPACK (src/spe_subfft_fftpack.f90:15-87, src/spe_subfft_fftpack2.f90).
//
// The reference does one 96x48 transform per call (23 us on a CPU core, 164 calls per time step).  A single transform
// is ~0.13 Mflop on 53 KB -- far too small for a GPU launch -- so every entry point here takes nf fields and runs ONE
// workgroup per field with the whole field resident in LDS:
//   inverse (grid): spectral field (15.9 KB) -> LDS; Legendre synthesis with the even/odd-n split that gives both
//       hemispheres from one pass (same summation order as gridy, so bit-identical to it without FMA contraction);
//       Fourier synthesis of the 31 retained zonal modes as a pruned 96-point real DFT from an LDS twiddle table
//       (60 multiply-adds per grid point, no butterfly stages, no barriers inside) instead of FFTPACK's 2*4*4*3
//       butterflies: with 65 of the 96 half-complex inputs structurally zero the pruned form needs fewer LDS
//       round trips than four radix stages; it differs from FFTPACK only in rounding (<= 1e-14 relative).
//   forward (spec): grid field (36.9 KB) -> LDS (optionally scaled by 1/cos or 1/cos^2 per latitude, the vdspec
//       prologue); 31-mode forward DFT; symmetric/antisymmetric combinations times the Gaussian weight in place;
//       Legendre analysis accumulated over latitude in the reference's order (bit-identical to specy).
// The Legendre table is stored once per zonal wavenumber (cpol(2m-1,n,j)=cpol(2m,n,j), :181-191): 24*32*31 doubles
// = 190 KB, L2-resident and shared by all workgroups.
// Spectral arrays: [nf][32][62] doubles (= Fortran (mx2,nx) per field), grids: [nf][48][96] (= (ix,il)).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "span.h"

namespace {

constexpr int IX = 96, IY = 24, IL = 48, NX = 32, MX = 31, MX2 = 62, NTRUN = 30, NTRUN1 = 31, NXP = 33, MXP = 31;
constexpr int SPEC_N = MX2 * NX;   // 1984
constexpr int GRID_N = IX * IL;    // 4608

struct HostTables {
    double sia[IY], coa[IY], wt[IY], wght[IY], cosg[IL], cosgr[IL], cosgr2[IL];
    double el2[NX][MX], elm2[NX][MX], el4[NX][MX], trfilt[NX][MX];
    int nsh2[NX];
    double epsi[NXP][MXP], repsi[NXP][MXP], consq[MXP], sqrhlf;
    double gradx[MX], gradym[NX][MX], gradyp[NX][MX], uvdx[NX][MX], uvdym[NX][MX], uvdyp[NX][MX], vddym[NX][MX], vddyp[NX][MX];
    double pol[IY][NX][MX];          // P_m^n at latitude j (cpol without the re/im duplication)
    double twc[IX], tws[IX];         // cos/sin(2 pi t / 96)
};

struct DevTables {
    const double *pol;               // [24][32][31]
    const double *polt;              // [32][31][24]: the same, latitude fastest (k_spec's Legendre analysis: a coefficient's 24 values in one 192-byte run)
    const double *wt, *cosgr, *cosgr2;
    const int *nsh2;
    const double *twc, *tws;
    const double *el2, *elm2, *trfilt, *gradx, *gradym, *gradyp, *uvdx, *uvdym, *uvdyp, *vddym, *vddyp;   // [32][31]
};

// ---- table construction on the host, fp64 (the reference promotes every real to 8 bytes, src/makefile:6,12) ----
void gauss_latitudes(double *x, double *w, int m)
{   // Newton iteration on P_{2m} from the Numerical-Recipes start value; pi literal and tolerance as src/spe_spectral.f90:16,24
    const int n = 2 * m;
    double zprev = 2.0;
    for (int i = 1; i <= m; ++i) {
        double z = std::cos(3.141592654 * (i - .25) / (n + .5)), pp = 0.0;
        while (std::fabs(z - zprev) > 3.e-14) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j; }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            zprev = z;
            z = zprev - p1 / pp;
        }
        x[i - 1] = z;
        w[i - 1] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

void build_tables(HostTables &t, double a)
{
    gauss_latitudes(t.sia, t.wt, IY);
    for (int j = 0; j < IY; ++j) {
        const double c2 = 1.0 - t.sia[j] * t.sia[j];
        t.coa[j] = std::sqrt(c2);
        t.wght[j] = t.wt[j] / (a * c2);
        const int jj = IL - 1 - j;
        t.cosg[j] = t.cosg[jj] = t.coa[j];
        t.cosgr[j] = t.cosgr[jj] = 1. / t.coa[j];
        t.cosgr2[j] = t.cosgr2[jj] = 1. / (t.coa[j] * t.coa[j]);
    }
    const double am2 = 1. / (a * a);
    for (int n = 0; n < NX; ++n) {
        t.nsh2[n] = 0;
        for (int m = 0; m < MX; ++m) {
            const int l = m + n;                         // total wavenumber
            t.el2[n][m] = (double)(l * (l + 1)) * am2;
            t.el4[n][m] = t.el2[n][m] * t.el2[n][m];
            if (l <= NTRUN1) t.nsh2[n] += 2;
            t.trfilt[n][m] = l <= NTRUN ? 1. : 0.;
            t.elm2[n][m] = (m == 0 && n == 0) ? 0. : 1. / t.el2[n][m];
        }
    }
    for (int m = 0; m < MXP; ++m)
        for (int n = 0; n < NXP; ++n) {
            const double em = m, el = n + m;
            double e;
            if (n == NXP - 1 || (n == 0 && m == 0)) e = 0.0;
            else e = std::sqrt((el * el - em * em) / (4. * (el * el) - 1.));
            t.epsi[n][m] = e;
            t.repsi[n][m] = e > 0. ? 1. / e : 0.0;
        }
    t.sqrhlf = std::sqrt(.5);
    t.consq[0] = 0.0;
    for (int m = 1; m < MXP; ++m) t.consq[m] = std::sqrt(.5 * (2. * m + 1.) / (double)m);
    for (int m = 0; m < MX; ++m)
        for (int n = 0; n < NX; ++n) {
            const double el1 = m + n;
            if (n == 0) {
                t.gradx[m] = (double)m / a;
                t.uvdx[0][m] = -a / (double)(m + 1);
                t.uvdym[0][m] = t.vddym[0][m] = t.gradym[0][m] = 0.0;
            } else {
                t.uvdx[n][m] = -a * (double)m / (el1 * (el1 + 1));
                t.gradym[n][m] = (el1 - 1.) * t.epsi[n][m] / a;
                t.uvdym[n][m] = -a * t.epsi[n][m] / el1;
                t.vddym[n][m] = (el1 + 1) * t.epsi[n][m] / a;
            }
            t.gradyp[n][m] = (el1 + 2.) * t.epsi[n + 1][m] / a;
            t.uvdyp[n][m] = -a * t.epsi[n + 1][m] / (el1 + 1.);
            t.vddyp[n][m] = el1 * t.epsi[n + 1][m] / a;
        }
    // associated Legendre functions by the three-term recurrence in n, seeded on the m = l diagonal
    std::vector<double> alp((size_t)NX * MXP);
    auto A = [&](int m, int n) -> double & { return alp[(size_t)n * MXP + m]; };
    for (int j = 0; j < IY; ++j) {
        const double x = t.sia[j], y = t.coa[j];
        A(0, 0) = t.sqrhlf;
        for (int m = 1; m < MXP; ++m) A(m, 0) = t.consq[m] * y * A(m - 1, 0);
        for (int m = 0; m < MXP; ++m) A(m, 1) = (x * A(m, 0)) * t.repsi[1][m];
        for (int n = 2; n < NX; ++n)
            for (int m = 0; m < MXP; ++m) A(m, n) = (x * A(m, n - 1) - t.epsi[n - 1][m] * A(m, n - 2)) * t.repsi[n][m];
        for (int n = 0; n < NX; ++n)
            for (int m = 0; m < MX; ++m) { double v = A(m, n); if (std::fabs(v) <= 1.e-30) v = 0.0; t.pol[j][n][m] = v; }
    }
    for (int k = 0; k < IX; ++k) {
        const long double ang = 2.0L * 3.14159265358979323846264338327950288L * k / (long double)IX;
        t.twc[k] = (double)cosl(ang);
        t.tws[k] = (double)sinl(ang);
    }
    t.twc[24] = t.twc[72] = 0.0; t.tws[0] = t.tws[48] = 0.0;
}

// ------------------------------------------------ kernels ------------------------------------------------

// A single 96x48 transform is far too small to fill the chip, and even a whole transform set (<= 91 fields) gives
// only 91 workgroups for 256 CUs.  Each field is therefore split over several workgroups along the dimension that
// needs no cross-workgroup reduction: latitude pairs for the inverse transform (Legendre synthesis and Fourier
// synthesis are both independent per latitude), zonal wavenumbers for the forward transform (the DFT is independent
// per wavenumber and the Legendre analysis only sums over latitude within one wavenumber).  Every workgroup re-reads
// its field from L2 (16-37 KB); nothing is exchanged between workgroups.
// How far to split was re-decided in round 4 from per-wavefront records (profiles/micro/window_span.py): a time step's sets are 77
// inverse and 73 forward fields, and what decides a launch is whether its workgroups exceed the 256 CUs -- a CU that holds two
// workgroups runs both slower (they share its LDS pipe), and the launch ends with those.  3 workgroups per field keep both sets at
// ONE workgroup per CU (231 and 219): k_grid 15.1 -> 10.8 us per launch against 6 x 512 threads (462 workgroups, 206 CUs doubly
// occupied, together with the geopotential rows moved to the spectral step), k_spec 12.1 -> 10.4 us against 4 x 512 (292, 36 doubly).
#ifndef SML_LATG
#define SML_LATG 8
#endif
constexpr int LATG = SML_LATG;          // latitude pairs per workgroup (inverse)
constexpr int NLG = IY / LATG;          // 3 workgroups per field
#ifndef SML_MG
#define SML_MG 11
#endif
constexpr int MG = SML_MG;              // zonal wavenumbers per workgroup (forward)
constexpr int NMG = (MX + MG - 1) / MG; // workgroups per field
constexpr int TT = 64 * MG;             // threads: 2 MG coefficient columns x 32 total wavenumbers in the Legendre analysis

// inverse transform: vorm[32][62] -> vorg[48][96]; workgroup = (field, group of LATG latitude pairs)
// (i g z): (re,im) -> (-g im, g re)
__device__ __forceinline__ double irot(const double *a, int base, int c, double g)
{
    return (c & 1) ? g * a[base + c - 1] : -g * a[base + c + 1];
}

// One spectral coefficient of a DERIVED field, so that uvspec / grad need not be materialised before their inverse
// transform (sml_spectral_grid_derived).  type 1 | 2: ucos | vcos of (vor = P, div = Q), uvspec :351-387;
// type 3 | 4: d/dx | d/dy of P, grad :271-305.  Same expressions, in the same order, as k_uvvds / k_grad below.
// type 7: geopotential of level `lev` from the 8 temperature levels that start at field P and the surface geopotential,
// dyn_geop.f90:19-35 (hydrostatic chain from the bottom layer up, then the m = 0 lapse-rate correction);
// aux = xgeop1(8) | xgeop2(8) | corf(8) | phis(62x32).
__device__ __forceinline__ double derived_coeff(const DevTables &T, int type, const double *__restrict__ P, const double *__restrict__ Q, int n, int c,
                                                int lev = 0, const double *__restrict__ aux = nullptr)
{
    const int m = c >> 1, row = n * MX2;
    if (type == 7) {
        constexpr int KX = 8;
        const int e = row + c;
        double phi = aux[24 + e] + aux[KX - 1] * P[(size_t)(KX - 1) * SPEC_N + e];
        for (int k = KX - 2; k >= lev; --k) phi = phi + aux[8 + k + 1] * P[(size_t)(k + 1) * SPEC_N + e] + aux[k] * P[(size_t)k * SPEC_N + e];
        if (m == 0 && lev >= 1 && lev <= KX - 2) phi = phi + aux[16 + lev] * (P[(size_t)(lev + 1) * SPEC_N + e] - P[(size_t)(lev - 1) * SPEC_N + e]);
        return phi;
    }
    if (type <= 2) {
        const double gx = T.uvdx[n * MX + m];
        const double *ym = T.uvdym, *yp = T.uvdyp;
        if (type == 1) {
            if (n == 0) return irot(Q, row, c, gx) - yp[m] * P[row + MX2 + c];
            if (n == NX - 1) return ym[n * MX + m] * P[row - MX2 + c];
            return ym[n * MX + m] * P[row - MX2 + c] - yp[n * MX + m] * P[row + MX2 + c] + irot(Q, row, c, gx);
        }
        if (n == 0) return irot(P, row, c, gx) + yp[m] * Q[row + MX2 + c];
        if (n == NX - 1) return -ym[n * MX + m] * Q[row - MX2 + c];
        return -ym[n * MX + m] * Q[row - MX2 + c] + yp[n * MX + m] * Q[row + MX2 + c] + irot(P, row, c, gx);
    }
    if (type == 3) return irot(P, row, c, T.gradx[m]);
    if (n == 0) return T.gradyp[m] * P[row + MX2 + c];
    if (n == NX - 1) return -T.gradym[n * MX + m] * P[row - MX2 + c];
    return -T.gradym[n * MX + m] * P[row - MX2 + c] + T.gradyp[n * MX + m] * P[row + MX2 + c];
}

// phase time stamps of one workgroup (profiles/micro/grid_phase_stamps.py); compiled in with -DSML_GRID_STAMPS only
__device__ unsigned long long g_grid_dbg[16];
#ifdef SML_GRID_STAMPS
#define GSTAMP(slot) do { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == SML_GRID_STAMPS && threadIdx.x == 0) g_grid_dbg[slot] = wall_clock64(); } while (0)
#else
#define GSTAMP(slot) do { } while (0)
#endif
__device__ __host__ constexpr int nsh2_of(int n) { return 2 * (NX - n) < MX2 ? 2 * (NX - n) : MX2; }
#ifndef SML_TG
#define SML_TG 1024
#endif
constexpr int TG = SML_TG;  // phase ablation: staging+launch floor 6.7 us, Legendre 2.6 us, Fourier 6.5 us at 256 threads (two
                            // rounds of the 392 Fourier items); 512 threads do them in one round
#ifndef SML_FPW
#define SML_FPW 1
#endif
constexpr int FPW = SML_FPW;            // fields per workgroup: each gets TG threads of its own, the Legendre slab is staged once
constexpr size_t GRID_LDS = sizeof(double) * ((size_t)FPW * SPEC_N + (size_t)LATG * NX * MX + (size_t)FPW * 2 * LATG * MX2 + 2 * IX) + sizeof(int) * NX;
// desc (optional): int32 [nf][4] = (type, src0, src1, kcos) per output field; type 0 = field src0 of vorm as it is,
// 1..4 = derived_coeff of fields src0 (P) and src1 (Q); 7 = geopotential of level src1 from the temperature levels at src0 (needs aux);
// 8 = geopotential of level src1 as the previous spectral step left it behind aux's tables (aux + 24 + SPEC_N + src1 * SPEC_N), a plain copy
// A workgroup can serve FPW fields of one latitude group (the Legendre slab is then staged once for both).  Measured on the 77-field
// set of a time step with physics (462 one-field workgroups, two on most CUs): FPW = 2 (231 workgroups of 1024 threads) is 1 us
// SLOWER per launch, and the launch costs 14.2 us even when every field is a plain copy (the geopotential rows add 1.3 us) against
// 10.8 us for 50 fields: the kernel's time follows the total LDS traffic of the Fourier phase, not the workgroup count.  FPW = 1.
__global__ __launch_bounds__(TG * FPW) void k_grid(DevTables T, const double *__restrict__ vorm, double *__restrict__ vorg, int kcos_all,
                                                    const int *__restrict__ kcos_of_field, const int *__restrict__ desc,
                                                    const double *__restrict__ aux, int nf)
{
    SML_SPAN(1);
    GSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) double grid_lds[];
    double *sv_all = grid_lds;                                            // [FPW][SPEC_N] spectral coefficients
    double (*sp)[NX][MX] = reinterpret_cast<double (*)[NX][MX]>(sv_all + FPW * SPEC_N);      // [LATG]: this latitude group's slab of the Legendre table (31 KB)
    double *sf_all = &sp[0][0][0] + LATG * NX * MX;                       // [FPW][2 * LATG][MX2] Fourier coefficients
    double *stc = sf_all + FPW * 2 * LATG * MX2, *sts = stc + IX;
    int *snsh = reinterpret_cast<int *>(sts + IX);
    const int sub = threadIdx.x / TG, tid = threadIdx.x % TG;             // sub is uniform per wavefront
    const int f = (blockIdx.x / NLG) * FPW + sub, lg = blockIdx.x % NLG;
    const bool active = f < nf;
    double *sv = sv_all + sub * SPEC_N;
    double (*sf)[MX2] = reinterpret_cast<double (*)[MX2]>(sf_all + sub * 2 * LATG * MX2);
    int kcos = kcos_all, type = 0, src0 = active ? f : 0, src1 = src0;      // (an idle half of a two-field workgroup stages field 0 and drops it)
    if (active) {
        if (kcos_of_field) kcos = kcos_of_field[f];
        if (desc) { const int4 d4 = *reinterpret_cast<const int4 *>(desc + 4 * f); type = d4.x; src0 = d4.y; src1 = d4.z; kcos = d4.w; }
    }
    const double *v = type == 8 ? aux + 24 + SPEC_N + (size_t)src1 * SPEC_N : vorm + (size_t)src0 * SPEC_N;
    double *g = vorg + (size_t)f * GRID_N;
    // ---- staging: EVERY global load of the workgroup goes out before the first LDS write (round 4).  The loops this replaces were
    // rolled -- load, wait, ds_write, next iteration -- so a workgroup walked through 8 + 2 dependent round trips, and the derived rows
    // (uvspec, grad) added a branch with its own wait per edge case of n.  Now: clamped addresses, unconditional loads into registers,
    // selects instead of branches; what a branch used to skip enters as a zero factor (the same values up to the sign of a zero).  Only
    // the coefficients inside the triangular truncation are staged (c < nsh2(n) = min(62, 2 (32 - n)), src/spe_spectral.f90:99-125).
    constexpr int NFP = (NX + TG / 64 - 1) / (TG / 64);
    const double *pg = T.pol + (size_t)lg * LATG * NX * MX;
    // The slab goes over as ONE flat run of 16-byte loads, entries outside the triangular truncation included (63 KB instead of the 34 KB
    // inside it): 4 load instructions per thread where 8-byte loads of the rows' retained parts took 8 -- what a workgroup waits for at
    // its start is the texture path working through its load instructions, not the bytes (7.8 -> 7.5 us per launch).
    typedef double d2s __attribute__((ext_vector_type(2)));
    constexpr int NSL = (LATG * NX * MX / 2 + TG * FPW - 1) / (TG * FPW);
    static_assert((LATG * NX * MX) % 2 == 0, "the slab is a whole number of 16-byte pairs");
    d2s pv2[NSL];
#pragma unroll
    for (int it = 0; it < NSL; ++it) {
        const int e = threadIdx.x + it * TG * FPW;
        pv2[it] = reinterpret_cast<const d2s *>(pg)[e < LATG * NX * MX / 2 ? e : 0];
    }
    // (a wavefront stages whole rows of the field: lane = coefficient within total wavenumber n, no index division per element)
    const int wv = tid >> 6, ln = tid & 63;
    double fv[NFP];
    bool pair_path = false, pair_ok = false;       // derived rows staged as (re, im) pairs (below)
    int pair_at = 0;
    double pair_o0 = 0., pair_o1 = 0.;
    if ((type == 0 || type == 8) && FPW == 1 && TG >= SPEC_N / 2) {
        // a plain field likewise: one flat 16-byte load per thread (the whole 15.9 KB, not only the part inside the truncation)
        pair_path = true; pair_ok = tid < SPEC_N / 2; pair_at = 2 * (pair_ok ? tid : 0);
        const d2s pv = reinterpret_cast<const d2s *>(v)[pair_ok ? tid : 0];
        pair_o0 = pv[0]; pair_o1 = pv[1];
    } else if (type == 0 || type == 8) {
#pragma unroll
        for (int it = 0; it < NFP; ++it) {
            const int n = wv + it * (TG / 64);
            fv[it] = v[(n < NX && ln < nsh2_of(n)) ? n * MX2 + ln : 0];
        }
    } else if (type == 7) {
#pragma unroll
        for (int it = 0; it < NFP; ++it) {
            const int n = wv + it * (TG / 64);
            fv[it] = (n < NX && ln < nsh2_of(n)) ? derived_coeff(T, 7, v, v, n, ln, src1, aux) : 0.0;       // (first step of a window only)
        }
    } else if (FPW == 1 && 2 * (TG / 64) >= NX) {
        // uvspec / grad rows with one lane per COMPLEX coefficient (end of round 4): a lane fetches the (re, im) pair of each of its three
        // operands with one 16-byte load and the three table values of its (n, m) once, a wavefront covers two rows of total wavenumber --
        // 6 load instructions per lane where the element-per-lane form below issues 12.  The 60 workgroups of a time step's 20 derived
        // fields were the last of every inverse launch by 1.2 us (7.7 against 6.5 us, per-wavefront records): what they wait for is the
        // texture path working through their load instructions.  Same expressions on the same operands: same bits.
        typedef double d2 __attribute__((ext_vector_type(2)));
        const double *q = vorm + (size_t)src1 * SPEC_N;
        const double *tx = type <= 2 ? T.uvdx : T.gradx, *tm = type <= 2 ? T.uvdym : T.gradym, *tp = type <= 2 ? T.uvdyp : T.gradyp;
        const double *M = type == 2 ? q : v, *X = type == 1 ? q : v;
        const int n = 2 * wv + (ln >> 5), mm = ln & 31;
        pair_ok = n < NX && mm < MX && 2 * mm < nsh2_of(n);
        const int nn = pair_ok ? n : 0, m = pair_ok ? mm : 0, row = nn * MX2, c0 = 2 * m;
        const d2 dm2 = *reinterpret_cast<const d2 *>(M + (nn > 0 ? row - MX2 : row) + c0);
        const d2 dp2 = *reinterpret_cast<const d2 *>(M + (nn < NX - 1 ? row + MX2 : row) + c0);
        const d2 x2 = *reinterpret_cast<const d2 *>(X + row + c0);
        const double cm = tm[nn * MX + m], cp = tp[nn * MX + m], cx = type >= 3 ? tx[m] : tx[nn * MX + m];
        const double ym_e = n == 0 ? 0. : cm, yp_e = n == NX - 1 ? 0. : cp;
        const double gx_e = (type <= 2 && n == NX - 1) ? 0. : cx, ngx = -gx_e;        // (re, im) -> (-g im, g re)
        double o0, o1;
        if (type == 1) { o0 = ym_e * dm2[0] - yp_e * dp2[0] + ngx * x2[1]; o1 = ym_e * dm2[1] - yp_e * dp2[1] + gx_e * x2[0]; }
        else if (type == 2) { o0 = -ym_e * dm2[0] + yp_e * dp2[0] + ngx * x2[1]; o1 = -ym_e * dm2[1] + yp_e * dp2[1] + gx_e * x2[0]; }
        else if (type == 3) { o0 = ngx * x2[1]; o1 = gx_e * x2[0]; }
        else { o0 = -ym_e * dm2[0] + yp_e * dp2[0]; o1 = -ym_e * dm2[1] + yp_e * dp2[1]; }
        pair_path = true; pair_at = nn * MX2 + c0; pair_o0 = o0; pair_o1 = o1;
    } else {
        // uvspec (types 1 | 2, :351-387) and grad (3 | 4, :271-305) of the fields P = src0, Q = src1, as derived_coeff writes them, branch-free:
        //   1: ym P(n-1) - yp P(n+1) + i gx Q     2: -ym Q(n-1) + yp Q(n+1) + i gx P     3: i gradx P     4: -gradym P(n-1) + gradyp P(n+1)
        // (n = 0 has no (n-1) term, n = 31 only its (n-1) term)
        const double *q = vorm + (size_t)src1 * SPEC_N;
        const double *tx = type <= 2 ? T.uvdx : T.gradx, *tm = type <= 2 ? T.uvdym : T.gradym, *tp = type <= 2 ? T.uvdyp : T.gradyp;
        const double *M = type == 2 ? q : v, *X = type == 1 ? q : v;        // neighbours in n come from M, the rotated partner from X
        double dm[NFP], dp[NFP], dx[NFP], cm[NFP], cp[NFP], cx[NFP];
#pragma unroll
        for (int it = 0; it < NFP; ++it) {
            const int n = wv + it * (TG / 64);
            const bool ok = n < NX && ln < nsh2_of(n);
            const int nn = ok ? n : 0, c = ok ? ln : 0, m = c >> 1, row = nn * MX2;
            dm[it] = M[(nn > 0 ? row - MX2 : row) + c];
            dp[it] = M[(nn < NX - 1 ? row + MX2 : row) + c];
            dx[it] = X[row + (c ^ 1)];
            cm[it] = tm[nn * MX + m];
            cp[it] = tp[nn * MX + m];
            cx[it] = type >= 3 ? tx[m] : tx[nn * MX + m];                 // (gradx is per zonal wavenumber, uvdx per coefficient; unused by type 4)
        }
#pragma unroll
        for (int it = 0; it < NFP; ++it) {
            const int n = wv + it * (TG / 64);
            const double ym_e = n == 0 ? 0. : cm[it], yp_e = n == NX - 1 ? 0. : cp[it];
            const double gx_e = (type <= 2 && n == NX - 1) ? 0. : cx[it], sgx = (ln & 1) ? gx_e : -gx_e;
            if (type == 1) fv[it] = ym_e * dm[it] - yp_e * dp[it] + sgx * dx[it];
            else if (type == 2) fv[it] = -ym_e * dm[it] + yp_e * dp[it] + sgx * dx[it];
            else if (type == 3) fv[it] = sgx * dx[it];
            else fv[it] = -ym_e * dm[it] + yp_e * dp[it];
        }
    }
    const int ti = threadIdx.x < IX ? threadIdx.x : 0, ni = threadIdx.x < NX ? threadIdx.x : 0;
    const double twc_v = T.twc[ti], tws_v = T.tws[ti];
    const int nsh_v = T.nsh2[ni];
    // ---- nothing is loaded below this line ----
    {
        double *pl = &sp[0][0][0];
#pragma unroll
        for (int it = 0; it < NSL; ++it) {
            const int e = threadIdx.x + it * TG * FPW;
            if (e < LATG * NX * MX / 2) reinterpret_cast<d2s *>(pl)[e] = pv2[it];
        }
    }
    if (active && pair_path) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        if (pair_ok) *reinterpret_cast<d2 *>(sv + pair_at) = d2{pair_o0, pair_o1};
    } else if (active) {
#pragma unroll
        for (int it = 0; it < NFP; ++it) {
            const int n = wv + it * (TG / 64);
            if (n < NX && ln < nsh2_of(n)) sv[n * MX2 + ln] = fv[it];
        }
    }
    if (threadIdx.x < IX) { stc[threadIdx.x] = twc_v; sts[threadIdx.x] = tws_v; }
    if (threadIdx.x < NX) snsh[threadIdx.x] = nsh_v;
    GSTAMP(1);
    __syncthreads();
    GSTAMP(2);
    // Legendre synthesis (gridy): E over odd n (1-based), O over even n; north = E+O, south = E-O.
    // Summation order is the reference's, so without FMA contraction this is bit-identical to gridy.
    // (Two threads per output pair since the end of round 4: lane 2q sums E, lane 2q + 1 sums O -- each sum in the order it always had --
    // and they exchange the results by DPP: half the chain length per thread, twice the threads busy, same bits.)
    if constexpr (TG >= 2 * LATG * MX2) {
        const int q = tid >> 1, par = tid & 1;
        const bool on = active && q < LATG * MX2;
        const int c = on ? q % MX2 : 0, jj = on ? q / MX2 : 0, m = c >> 1;
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < NX; n += 2) {
            const int nn = n + par;
            if (c < nsh2_of(nn)) acc = acc + sv[nn * MX2 + c] * sp[jj][nn][m];
        }
        const double other = __shfl_xor(acc, 1);
        if (on) {
            if (par == 0) sf[2 * jj + 1][c] = acc + other;       // northern row il+1-j: E + O
            else sf[2 * jj][c] = other - acc;                    // southern row j: E - O
        }
    } else if (active && tid < LATG * MX2) {
        const int c = tid % MX2, jj = tid / MX2, m = c >> 1;
        double e = 0.0, o = 0.0;
#pragma unroll
        for (int n = 0; n < NX; n += 2) {
            if (c < nsh2_of(n)) e = e + sv[n * MX2 + c] * sp[jj][n][m];
            if (c < nsh2_of(n + 1)) o = o + sv[(n + 1) * MX2 + c] * sp[jj][n + 1][m];
        }
        sf[2 * jj + 1][c] = e + o;       // northern row il+1-j
        sf[2 * jj][c] = e - o;           // southern row j
    }
    GSTAMP(3);
    __syncthreads();
    GSTAMP(4);
    // Fourier synthesis (gridx) of the 31 retained modes, two longitudes per result: with A = sum_k Re_k cos(k i t)
    // and B = sum_k Im_k sin(k i t),  x_i = a0 + 2(A - B)  and  x_{96-i} = a0 + 2(A + B)   (i = 0..48).
    if constexpr (FPW == 1) {
        // On the matrix cores since round 4 (as the forward DFT of k_spec): A and B are two small dense products, (2 LATG rows x 30 modes)
        // times (30 modes x 49 longitudes), v_mfma_f64_4x4x4_4b with a block = (quad of latitude rows, quad of longitudes) and 8 k-steps
        // over the modes 1..30 (+ two zero columns).  Operand lane = 16 k + 4 blk + (row | longitude), result lane = 16 row + 4 blk +
        // longitude.  The vector loop this replaces read four LDS words per multiply-add pair (3.1 of a workgroup's 8 us at the old
        // geometry, LDS-pipe bound).
        constexpr int NRQ = 2 * LATG / 4, NIQ = (IX / 2 + 1 + 3) / 4, NBLK = NRQ * NIQ, NGRP = (NBLK + 3) / 4;
        static_assert((2 * LATG) % 4 == 0, "latitude rows of a workgroup come in quads");
        const int lane = threadIdx.x & 63, kq = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
        for (int gq = threadIdx.x >> 6; gq < NGRP; gq += TG / 64) {
            const int B = 4 * gq + blk, iq = B / NRQ, rq = B % NRQ;
            const bool bok = B < NBLK && active;
            const double *fa = sf[rq * 4 + x];                             // this lane's row of Fourier coefficients (A operand)
            const int ib = iq * 4 + x;                                     // this lane's longitude (B operand)
            const bool iok = bok && ib <= IX / 2;
            double A = 0.0, Bq = 0.0;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int k = 1 + 4 * ks + kq;                             // zonal wavenumber of this lane's operand column
                const bool kok = k <= MX - 1;
                const int ph = (k * ib) % IX;
                const double are = (bok && kok) ? fa[2 * k] : 0.0, aim = (bok && kok) ? fa[2 * k + 1] : 0.0;
                const double bc = (iok && kok) ? stc[ph] : 0.0, bs = (iok && kok) ? sts[ph] : 0.0;
                A = __builtin_amdgcn_mfma_f64_4x4x4f64(are, bc, A, 0, 0, 0);
                Bq = __builtin_amdgcn_mfma_f64_4x4x4f64(aim, bs, Bq, 0, 0, 0);
            }
            const int r = rq * 4 + kq, i = iq * 4 + x;                     // result lane = 16 row + 4 blk + longitude
            if (bok && i <= IX / 2) {
                const int j = lg * LATG + (r >> 1);
                const int row = (r & 1) ? IL - 1 - j : j;
                const double a0 = sf[r][0];
                double x0 = a0 + 2.0 * (A - Bq), x1 = a0 + 2.0 * (A + Bq);
                if (kcos != 1) { const double cg = T.cosgr[row]; x0 = x0 * cg; x1 = x1 * cg; }
                g[row * IX + i] = x0;
                if (i != 0 && i != IX / 2) g[row * IX + IX - i] = x1;
            }
        }
    } else {
    for (int w = active ? tid : 2 * LATG * (IX / 2 + 1); w < 2 * LATG * (IX / 2 + 1); w += TG) {
        const int i = w % (IX / 2 + 1), r = w / (IX / 2 + 1);
        const int j = lg * LATG + (r >> 1);
        const int row = (r & 1) ? IL - 1 - j : j;
        const double *fc = sf[r];
        double A = 0.0, B = 0.0;
        int ph = 0;
#pragma unroll 6
        for (int k = 1; k <= MX - 1; ++k) {
            ph += i;
            if (ph >= IX) ph -= IX;
            A += fc[2 * k] * stc[ph];
            B += fc[2 * k + 1] * sts[ph];
        }
        double x0 = fc[0] + 2.0 * (A - B), x1 = fc[0] + 2.0 * (A + B);
        if (kcos != 1) { const double cg = T.cosgr[row]; x0 = x0 * cg; x1 = x1 * cg; }
        g[row * IX + i] = x0;
        if (i != 0 && i != IX / 2) g[row * IX + IX - i] = x1;
    }
    }
    GSTAMP(5);
}

// forward transform: vorg[48][96] -> vorm[32][62]; workgroup = (field, group of MG zonal wavenumbers)
// scale: 0 none, 1 *cosgr(j), 2 *cosgr2(j)  (the vdspec prologue, :429-443)
__global__ __launch_bounds__(TT) void k_spec(DevTables T, const double *__restrict__ vorg, double *__restrict__ vorm, int scale_all,
                                              const int *__restrict__ scale_of_field)
{
    SML_SPAN(3);
    // The field is folded about longitude index 48 while it is staged: ss = x_i + x_{96-i}, sd = x_i - x_{96-i}
    // (i = 1..47; ss[0] = x_0, ss[48] = x_48), which halves the DFT: Re_k = sum ss cos, Im_k = - sum sd sin.
    __shared__ __attribute__((aligned(16))) double ss[IL][IX / 2 + 2];      // padded rows: lanes that differ in latitude hit different banks
    __shared__ __attribute__((aligned(16))) double sd[IL][IX / 2 + 2];
    __shared__ double twc[MG][IX / 2 + 3], tws[MG][IX / 2 + 3];
    __shared__ double sf[IL][2 * MG];          // this workgroup's Fourier coefficients [lat][re/im of its wavenumbers]
    __shared__ double swt[IY];                 // Gaussian weights, staged with the field (a load after the DFT costs a round trip)
    GSTAMP(8);
    const int f = blockIdx.x / NMG, mg = blockIdx.x % NMG;
    const int scale = scale_of_field ? scale_of_field[f] : scale_all;
    const int k0 = mg * MG, nk = min(MG, MX - k0);
    const double *g = vorg + (size_t)f * GRID_N;
    double *v = vorm + (size_t)f * SPEC_N;
    // staging: all of these global loads first (registers), the LDS writes after them -- as a rolled loop (load, wait, fold, ds_write,
    // next) this was four dependent round trips per thread (round 4, see k_grid).  The 24 Legendre-table values of the analysis at the
    // end stay where they are: fetched here as well they make the launch's start slower by what they save at its end (measured again
    // in round 4: workgroup 6.6 against 5.5 us).
    // Two longitudes per item and 16-byte loads (end of round 4): item (j, p) takes x_{2p}, x_{2p+1} in one load and their mirror images
    // x_{96-2p}, x_{95-2p} in another (8-byte aligned, which global loads allow) -- 6 load instructions per thread where one longitude per
    // item took 12.  The folds and products are the same expressions on the same values.
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    constexpr int NPAIR = IX / 4 + 1, NSP = (IL * NPAIR + TT - 1) / TT;      // 25 pairs per latitude row: (0,1) ... (46,47), (48,-)
    d2 ga[NSP], gb[NSP];
    double cgv[NSP];
#pragma unroll
    for (int it = 0; it < NSP; ++it) {
        const int w = threadIdx.x + it * TT, wc = w < IL * NPAIR ? w : 0;
        const int p = wc % NPAIR, j = wc / NPAIR;
        ga[it] = *reinterpret_cast<const d2 *>(g + j * IX + 2 * p);
        gb[it] = *reinterpret_cast<const d2u *>(g + j * IX + (p == 0 ? IX - 2 : IX - 1 - 2 * p));      // (x_{95-2p}, x_{96-2p}); p = 0: (x_94, x_95), inside the row
        cgv[it] = scale == 2 ? T.cosgr2[j] : T.cosgr[j];
    }
    constexpr int NTP = (MG * (IX / 2 + 1) + TT - 1) / TT;
    double tcv[NTP], tsv[NTP];
#pragma unroll
    for (int it = 0; it < NTP; ++it) {
        const int w = threadIdx.x + it * TT, wc = w < MG * (IX / 2 + 1) ? w : 0;
        const int i = wc % (IX / 2 + 1), kk = wc / (IX / 2 + 1);
        const int ph = ((k0 + kk) * i) % IX;
        tcv[it] = T.twc[ph];
        tsv[it] = T.tws[ph];
    }
    const double wt_v = T.wt[threadIdx.x < IY ? threadIdx.x : 0];
    // ---- nothing is loaded from here to the Legendre table below ----
#pragma unroll
    for (int it = 0; it < NSP; ++it) {
        const int w = threadIdx.x + it * TT;
        if (w < IL * NPAIR) {
            const int p = w % NPAIR, j = w / NPAIR;
            // longitude 2p (no mirror image at 0 and 48) and longitude 2p + 1 (mirror x_{95-2p}: gb[0], or gb[1] of the p = 0 load); the
            // second half of pair 24 lands in the rows' padding column
            double a0 = ga[it][0], b0 = (p == 0 || p == NPAIR - 1) ? 0.0 : gb[it][1];
            double a1 = ga[it][1], b1 = p == 0 ? gb[it][1] : gb[it][0];
            if (scale == 1 || scale == 2) { a0 = a0 * cgv[it]; b0 = b0 * cgv[it]; a1 = a1 * cgv[it]; b1 = b1 * cgv[it]; }
            *reinterpret_cast<d2 *>(&ss[j][2 * p]) = d2{a0 + b0, a1 + b1};
            *reinterpret_cast<d2 *>(&sd[j][2 * p]) = d2{a0 - b0, a1 - b1};
        }
    }
#pragma unroll
    for (int it = 0; it < NTP; ++it) {
        const int w = threadIdx.x + it * TT;
        if (w < MG * (IX / 2 + 1)) { twc[w / (IX / 2 + 1)][w % (IX / 2 + 1)] = tcv[it]; tws[w / (IX / 2 + 1)][w % (IX / 2 + 1)] = tsv[it]; }
    }
    if (threadIdx.x < IY) swt[threadIdx.x] = wt_v;
    GSTAMP(9);
    __syncthreads();
    GSTAMP(10);
    // forward DFT (specx): a0 = sum x / 96 ; Re_k = sum x cos / 96 ; Im_k = - sum x sin / 96 ; Im of k=0 is set to 0.
    // On the matrix cores since round 4: Re[m][lat] = sum_i twc[m][i] ss[lat][i] and S[m][lat] = sum_i tws[m][i] sd[lat][i] are two small
    // dense products (MG x 49 times 49 x 48), v_mfma_f64_4x4x4_4b: four independent 4 x 4 x 4 blocks per instruction, a block = (quad of
    // zonal wavenumbers, quad of latitudes), 13 k-steps over the 49 folded longitudes.  Operand lane = 16 k + 4 blk + (row | column)
    // and result lane = 16 i + 4 blk + j (measured layout, csrc/train.hip): every lane reads one twiddle and one data value from LDS per
    // instruction.  The vector form this replaces (one wavefront per wavenumber, lanes = latitudes, twiddles by v_readlane) spent 4.2 of a
    // workgroup's 8.5 us here at 18 % of the vector rate: 48 of 64 lanes, 11 wavefronts on 4 SIMDs, unfused multiply + add.  Fused
    // multiply-adds in a fixed order: deterministic, within 1e-15 of the vector form (the DFT never was FFTPACK's operation order).
    {
        constexpr int NQ_L = IL / 4, NBLK = ((MG + 3) / 4) * NQ_L, NGRP = (NBLK + 3) / 4;
        const int lane = threadIdx.x & 63, kq = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
        const double sc = 1. / (double)IX;
        for (int g = threadIdx.x >> 6; g < NGRP; g += TT / 64) {
            const int B = 4 * g + blk, mq = B / NQ_L, lq = B % NQ_L;
            const int mrow = mq * 4 + x, lrow = lq * 4 + x;               // this lane's twiddle row (A operand) and data row (B operand)
            const bool bok = B < NBLK, aok = bok && mrow < nk;
            double re = 0.0, im = 0.0;
#pragma unroll
            for (int i0 = 0; i0 < IX / 2 + 4; i0 += 4) {
                const int i = i0 + kq;
                const bool in = i <= IX / 2;                               // (49 longitudes: the last k-step is one real column and three zeros)
                const double ac = (aok && in) ? twc[mrow][i] : 0.0, as = (aok && in) ? tws[mrow][i] : 0.0;
                const double bs = (bok && in) ? ss[lrow][i] : 0.0, bd = (bok && in) ? sd[lrow][i] : 0.0;
                re = __builtin_amdgcn_mfma_f64_4x4x4f64(ac, bs, re, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bd, im, 0, 0, 0);
            }
            const int kk = mq * 4 + kq, lat = lq * 4 + x;                  // result lane = 16 i + 4 blk + j: i = wavenumber, j = latitude of the quad
            if (bok && kk < nk) {
                sf[lat][2 * kk] = re * sc;
                sf[lat][2 * kk + 1] = (k0 + kk == 0) ? 0.0 : -im * sc;    // sd[.][0] = x_0 and sd[.][48] = x_48 meet sin = 0
            }
        }
    }
    GSTAMP(11);
    __syncthreads();
    // symmetric / antisymmetric parts times the Gaussian weight, in place (specy :511-517)
    if (threadIdx.x < IY * 2 * MG) {
        const int cc = threadIdx.x % (2 * MG), j = threadIdx.x / (2 * MG), j1 = IL - 1 - j;
        const double n_ = sf[j1][cc], s_ = sf[j][cc], wj = swt[j];
        sf[j][cc] = (n_ + s_) * wj;       // svarm
        sf[j1][cc] = (n_ - s_) * wj;      // dvarm
    }
    __syncthreads();
    GSTAMP(12);
    // Legendre analysis (specy :519-537): odd n (1-based) use svarm, even n use dvarm, n <= ntrun1, c < nsh2(n);
    // accumulation over latitude in the reference's order -> bit-identical to specy for identical Fourier input.
    // The 24 table loads of a thread are independent and issued together (full unroll).
    {
        const int cc = threadIdx.x % (2 * MG), n = threadIdx.x / (2 * MG);     // 2 MG x 32 = TT threads
        const int c = 2 * k0 + cc, m = c >> 1;
        if (cc < 2 * nk) {
            double acc = 0.0;
            if (n < NTRUN1 && c < nsh2_of(n)) {
                // (fetching these 24 table values ahead of the DFT was measured twice: the quadrature and Legendre phases shrink by
                // 1.4 us per workgroup, the staging and DFT phases grow by as much -- the launch's loads all compete at its start)
                // Six 16-byte loads per lane instead of 24 8-byte ones (end of round 4): the (re, im) lanes of a coefficient need the
                // same 24 table values, so the even lane fetches latitudes 0..11 and the odd one 12..23 from the latitude-fastest copy
                // of the table and they swap halves by DPP.  Same values in the same sums.
                typedef double d2 __attribute__((ext_vector_type(2)));
                const int par = cc & 1;
                const d2 *p2 = reinterpret_cast<const d2 *>(T.polt + ((size_t)n * MX + m) * IY + (IY / 2) * par);
                d2 mine[IY / 4];
#pragma unroll
                for (int i = 0; i < IY / 4; ++i) mine[i] = p2[i];
                double pj[IY];
#pragma unroll
                for (int i = 0; i < IY / 4; ++i) {
                    const double o0 = __shfl_xor(mine[i][0], 1), o1 = __shfl_xor(mine[i][1], 1);
                    pj[2 * i] = par ? o0 : mine[i][0];
                    pj[2 * i + 1] = par ? o1 : mine[i][1];
                    pj[IY / 2 + 2 * i] = par ? mine[i][0] : o0;
                    pj[IY / 2 + 2 * i + 1] = par ? mine[i][1] : o1;
                }
                if ((n & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < IY; ++j) acc = acc + pj[j] * sf[j][cc];
                } else {
#pragma unroll
                    for (int j = 0; j < IY; ++j) acc = acc + pj[j] * sf[IL - 1 - j][cc];
                }
            }
            v[n * MX2 + c] = acc;
        }
    }
    GSTAMP(13);
}

// pointwise / 3-point-in-n spectral operators; one thread per real element [nf][32][62]
enum { OP_LAP = 0, OP_INVLAP = 1, OP_TRUNCT = 2 };
__global__ void k_scale(DevTables T, const double *__restrict__ in, double *__restrict__ out, int total, int op)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int e = t % SPEC_N, c = e % MX2, n = e / MX2, m = c >> 1;
    const double x = in[t];
    if (op == OP_LAP) out[t] = -x * T.el2[n * MX + m];
    else if (op == OP_INVLAP) out[t] = -x * T.elm2[n * MX + m];
    else out[t] = x * T.trfilt[n * MX + m];
}

// uvspec (:351-387) and vds (:307-349) share one stencil:  A = ym*P(n-1) - yp*P(n+1) + i x Q ; B = -ym*Q(n-1) + yp*Q(n+1) + i x P
// uvspec: P=vor,Q=div, x=uvdx(m,n), ym=uvdym, yp=uvdyp -> (A,B)=(ucos,vcos).  vds: P=ucos,Q=vcos, x=gradx(m), ym=vddym, yp=vddyp -> (vor,div)
__global__ void k_uvvds(DevTables T, const double *__restrict__ P, const double *__restrict__ Q, double *__restrict__ A,
                        double *__restrict__ B, int total, int is_vds)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int f = t / SPEC_N, e = t % SPEC_N, c = e % MX2, n = e / MX2, m = c >> 1;
    const int fb = f * SPEC_N, row = fb + n * MX2;
    const double gx = is_vds ? T.gradx[m] : T.uvdx[n * MX + m];
    const double *ym = is_vds ? T.vddym : T.uvdym, *yp = is_vds ? T.vddyp : T.uvdyp;
    const double zc = irot(Q, row, c, gx);      // i x Q
    const double zp = irot(P, row, c, gx);      // i x P
    double a, b;
    if (n == 0) {
        a = zc - yp[m] * P[row + MX2 + c];
        b = zp + yp[m] * Q[row + MX2 + c];
    } else if (n == NX - 1) {
        a = ym[n * MX + m] * P[row - MX2 + c];
        b = -ym[n * MX + m] * Q[row - MX2 + c];
    } else {
        a = ym[n * MX + m] * P[row - MX2 + c] - yp[n * MX + m] * P[row + MX2 + c] + zc;
        b = -ym[n * MX + m] * Q[row - MX2 + c] + yp[n * MX + m] * Q[row + MX2 + c] + zp;
    }
    A[t] = a;
    B[t] = b;
}

// Spectral post-processing by descriptor (sml_spectral_spec_post): out field f = trunct?( type 0: field src0 | type 5 / 6: vor /
// div of vds(ucos = src0, vcos = src1) ).  Same expressions as k_uvvds followed by k_scale(OP_TRUNCT).
// Output fields past the first total_main / SPEC_N go to out2 (the hybrid engine transforms fordate's two correction fields in the same
// launch as iogrid(30)'s 33 and wants them in the time steps' boundary arrays, not behind the state).
__global__ void k_post(DevTables T, const double *__restrict__ in, const int *__restrict__ desc, double *__restrict__ out, int total,
                       double *__restrict__ out2, int total_main)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int f = t / SPEC_N, e = t % SPEC_N, c = e % MX2, n = e / MX2, m = c >> 1;
    const int type = desc[4 * f], src0 = desc[4 * f + 1], src1 = desc[4 * f + 2], trunc = desc[4 * f + 3];
    double v;
    if (type == 0) {
        v = in[(size_t)src0 * SPEC_N + e];
    } else {
        const double *P = in + (size_t)src0 * SPEC_N, *Q = in + (size_t)src1 * SPEC_N;
        const int row = n * MX2;
        const double gx = T.gradx[m], ym = T.vddym[n * MX + m], yp = T.vddyp[n * MX + m];
        if (type == 5) {
            if (n == 0) v = irot(Q, row, c, gx) - T.vddyp[m] * P[row + MX2 + c];
            else if (n == NX - 1) v = ym * P[row - MX2 + c];
            else v = ym * P[row - MX2 + c] - yp * P[row + MX2 + c] + irot(Q, row, c, gx);
        } else {
            if (n == 0) v = irot(P, row, c, gx) + T.vddyp[m] * Q[row + MX2 + c];
            else if (n == NX - 1) v = -ym * Q[row - MX2 + c];
            else v = -ym * Q[row - MX2 + c] + yp * Q[row + MX2 + c] + irot(P, row, c, gx);
        }
    }
    if (trunc) v = v * T.trfilt[n * MX + m];
    if (t < total_main) out[t] = v;
    else out2[t - total_main] = v;
}

// grad (:271-305)
__global__ void k_grad(DevTables T, const double *__restrict__ psi, double *__restrict__ dx, double *__restrict__ dy, int total)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int f = t / SPEC_N, e = t % SPEC_N, c = e % MX2, n = e / MX2, m = c >> 1;
    const int row = f * SPEC_N + n * MX2;
    dx[t] = irot(psi, row, c, T.gradx[m]);
    double y;
    if (n == 0) y = T.gradyp[m] * psi[row + MX2 + c];
    else if (n == NX - 1) y = -T.gradym[n * MX + m] * psi[row - MX2 + c];
    else y = -T.gradym[n * MX + m] * psi[row - MX2 + c] + T.gradyp[n * MX + m] * psi[row + MX2 + c];
    dy[t] = y;
}

}  // namespace

struct sml_spectral {
    HostTables h;
    DevTables d{};
    std::vector<void *> allocs;
    double *scratch = nullptr;      // 2 * nf_cap * SPEC_N for vdspec
    int scratch_fields = 0;
    double *hs_spec[4] = {nullptr, nullptr, nullptr, nullptr};   // host-call staging (F77 drop-ins)
    double *hs_grid[2] = {nullptr, nullptr};
};

namespace {

template <class T>
int up(sml_spectral *sp, const T **dst, const T *src, size_t n)
{
    T *p = nullptr;
    int rc = sml::dev_upload(&p, src, n);
    if (rc) return rc;
    sp->allocs.push_back(p);
    *dst = p;
    return SML_OK;
}

int ensure_scratch(sml_spectral *sp, int nf)
{
    if (nf <= sp->scratch_fields) return SML_OK;
    if (sp->scratch) (void)hipFree(sp->scratch);
    sp->scratch = nullptr;
    SML_HIP(hipMalloc((void **)&sp->scratch, (size_t)2 * nf * SPEC_N * sizeof(double)));
    sp->scratch_fields = nf;
    return SML_OK;
}

sml_spectral *g_sp = nullptr;   // used by the F77 drop-ins

}  // namespace

extern "C" {

int sml_spectral_create(double a, sml_spectral **out)
{
    SML_REQUIRE(out && a > 0, "sml_spectral_create: bad arguments");
    sml_spectral *sp = new sml_spectral;
    build_tables(sp->h, a);
    HostTables &h = sp->h;
    int rc = 0;
#define UP(field, src, n) if (!rc) rc = up(sp, &sp->d.field, src, n)
    UP(pol, &h.pol[0][0][0], (size_t)IY * NX * MX);
    {
        std::vector<double> pt((size_t)NX * MX * IY);
        for (int j = 0; j < IY; ++j)
            for (int n = 0; n < NX; ++n)
                for (int m = 0; m < MX; ++m) pt[((size_t)n * MX + m) * IY + j] = h.pol[j][n][m];
        UP(polt, pt.data(), pt.size());
    }
    UP(wt, h.wt, IY); UP(cosgr, h.cosgr, IL); UP(cosgr2, h.cosgr2, IL); UP(nsh2, h.nsh2, NX);
    UP(twc, h.twc, IX); UP(tws, h.tws, IX);
    UP(el2, &h.el2[0][0], NX * MX); UP(elm2, &h.elm2[0][0], NX * MX); UP(trfilt, &h.trfilt[0][0], NX * MX);
    UP(gradx, h.gradx, MX); UP(gradym, &h.gradym[0][0], NX * MX); UP(gradyp, &h.gradyp[0][0], NX * MX);
    UP(uvdx, &h.uvdx[0][0], NX * MX); UP(uvdym, &h.uvdym[0][0], NX * MX); UP(uvdyp, &h.uvdyp[0][0], NX * MX);
    UP(vddym, &h.vddym[0][0], NX * MX); UP(vddyp, &h.vddyp[0][0], NX * MX);
#undef UP
    for (int n = 0; n < NX && !rc; ++n)
        if (h.nsh2[n] != nsh2_of(n)) rc = sml::fail(SML_ERR_ARG, "sml_spectral_create: nsh2(%d) = %d, the kernels assume %d", n + 1, h.nsh2[n], nsh2_of(n));
    if (!rc && hipFuncSetAttribute((const void *)k_grid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRID_LDS) != hipSuccess)
        rc = sml::fail(SML_ERR_HIP, "sml_spectral_create: cannot reserve %zu bytes of LDS for k_grid", GRID_LDS);
    if (rc) { sml_spectral_destroy(sp); return rc; }
    *out = sp;
    return SML_OK;
}

int sml_spectral_destroy(sml_spectral *sp)
{
    if (!sp) return SML_OK;
    for (void *p : sp->allocs) (void)hipFree(p);
    if (sp->scratch) (void)hipFree(sp->scratch);
    for (auto p : sp->hs_spec) if (p) (void)hipFree(p);
    for (auto p : sp->hs_grid) if (p) (void)hipFree(p);
    if (g_sp == sp) g_sp = nullptr;
    delete sp;
    return SML_OK;
}

int sml_spectral_get_table(sml_spectral *sp, int which, double *out, int capacity)
{
    SML_REQUIRE(sp && out, "sml_spectral_get_table: bad arguments");
    const HostTables &h = sp->h;
    const double *src = nullptr;
    int n = 0;
    std::vector<double> tmp;
    switch (which) {
    case 1: src = h.sia; n = IY; break;
    case 2: src = h.coa; n = IY; break;
    case 3: src = h.wt; n = IY; break;
    case 4: src = h.wght; n = IY; break;
    case 5: src = h.cosg; n = IL; break;
    case 6: src = h.cosgr; n = IL; break;
    case 7: src = h.cosgr2; n = IL; break;
    case 8: src = &h.el2[0][0]; n = NX * MX; break;
    case 9: src = &h.elm2[0][0]; n = NX * MX; break;
    case 10: src = &h.el4[0][0]; n = NX * MX; break;
    case 11: src = &h.trfilt[0][0]; n = NX * MX; break;
    case 12: tmp.resize(NX); for (int i = 0; i < NX; ++i) tmp[i] = h.nsh2[i]; src = tmp.data(); n = NX; break;
    case 13: src = &h.epsi[0][0]; n = NXP * MXP; break;
    case 14: src = &h.repsi[0][0]; n = NXP * MXP; break;
    case 15: src = h.consq; n = MXP; break;
    case 16: src = h.gradx; n = MX; break;
    case 17: src = &h.gradym[0][0]; n = NX * MX; break;
    case 18: src = &h.gradyp[0][0]; n = NX * MX; break;
    case 19: src = &h.uvdx[0][0]; n = NX * MX; break;
    case 20: src = &h.uvdym[0][0]; n = NX * MX; break;
    case 21: src = &h.uvdyp[0][0]; n = NX * MX; break;
    case 22: src = &h.vddym[0][0]; n = NX * MX; break;
    case 23: src = &h.vddyp[0][0]; n = NX * MX; break;
    case 24:   // cpol(mx2,nx,iy): duplicate each wavenumber into its (re,im) pair
        tmp.resize((size_t)MX2 * NX * IY);
        for (int j = 0; j < IY; ++j) for (int nn = 0; nn < NX; ++nn) for (int c = 0; c < MX2; ++c)
            tmp[((size_t)j * NX + nn) * MX2 + c] = h.pol[j][nn][c / 2];
        src = tmp.data(); n = MX2 * NX * IY; break;
    case 26: src = &h.sqrhlf; n = 1; break;
    default: return sml::fail(SML_ERR_ARG, "sml_spectral_get_table: unknown table %d", which);
    }
    SML_REQUIRE(capacity >= n, "sml_spectral_get_table: capacity %d < %d", capacity, n);
    memcpy(out, src, sizeof(double) * n);
    return n;
}

SML_SPAN_ATTACH(sml_span_attach_spectral)

int sml_spectral_debug_stamps(unsigned long long *out)      // not part of the C-ABI (no declaration in include/): phase profiling aid
{
    SML_HIP(hipDeviceSynchronize());
    SML_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grid_dbg), sizeof(unsigned long long) * 16));
    return SML_OK;
}

int sml_spectral_grid(sml_spectral *sp, const double *vorm, double *vorg, int nf, int kcos, void *stream)
{
    SML_REQUIRE(sp && nf >= 0 && (kcos == 1 || kcos == 2) && (nf == 0 || (vorm && vorg)), "sml_spectral_grid: bad arguments");
    if (!nf) return SML_OK;
    hipLaunchKernelGGL(k_grid, dim3((nf + FPW - 1) / FPW * NLG), dim3(TG * FPW), GRID_LDS, sml::as_stream(stream), sp->d, vorm, vorg, kcos,
                       (const int *)nullptr, (const int *)nullptr, (const double *)nullptr, nf);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

static int spec_scaled(sml_spectral *sp, const double *vorg, double *vorm, int nf, int scale, void *stream)
{
    hipLaunchKernelGGL(k_spec, dim3(nf * NMG), dim3(TT), 0, sml::as_stream(stream), sp->d, vorg, vorm, scale, (const int *)nullptr);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_grid_mixed(sml_spectral *sp, const double *vorm, double *vorg, int nf, const int32_t *kcos_dev, void *stream)
{
    SML_REQUIRE(sp && nf >= 0 && (nf == 0 || (vorm && vorg && kcos_dev)), "sml_spectral_grid_mixed: bad arguments");
    if (!nf) return SML_OK;
    hipLaunchKernelGGL(k_grid, dim3((nf + FPW - 1) / FPW * NLG), dim3(TG * FPW), GRID_LDS, sml::as_stream(stream), sp->d, vorm, vorg, 1,
                       (const int *)kcos_dev, (const int *)nullptr, (const double *)nullptr, nf);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_grid_derived_aux(sml_spectral *sp, const double *spec_base, const int32_t *desc_dev, const double *aux_dev, double *vorg, int nf,
                                  void *stream)
{
    SML_REQUIRE(sp && nf >= 0 && (nf == 0 || (spec_base && desc_dev && vorg)), "sml_spectral_grid_derived: bad arguments");
    if (!nf) return SML_OK;
    hipLaunchKernelGGL(k_grid, dim3((nf + FPW - 1) / FPW * NLG), dim3(TG * FPW), GRID_LDS, sml::as_stream(stream), sp->d, spec_base, vorg, 1,
                       (const int *)nullptr, (const int *)desc_dev, aux_dev, nf);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_grid_derived(sml_spectral *sp, const double *spec_base, const int32_t *desc_dev, double *vorg, int nf, void *stream)
{
    return sml_spectral_grid_derived_aux(sp, spec_base, desc_dev, nullptr, vorg, nf, stream);
}

int sml_spectral_spec_mixed(sml_spectral *sp, const double *vorg, double *vorm, int nf, const int32_t *scale_dev, void *stream)
{
    SML_REQUIRE(sp && nf >= 0 && (nf == 0 || (vorm && vorg && scale_dev)), "sml_spectral_spec_mixed: bad arguments");
    if (!nf) return SML_OK;
    hipLaunchKernelGGL(k_spec, dim3(nf * NMG), dim3(TT), 0, sml::as_stream(stream), sp->d, vorg, vorm, 0, (const int *)scale_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_spec_post(sml_spectral *sp, const double *spec_in, const int32_t *desc_dev, double *spec_out, int nf_out, void *stream)
{
    SML_REQUIRE(sp && nf_out >= 0 && (nf_out == 0 || (spec_in && desc_dev && spec_out)), "sml_spectral_spec_post: bad arguments");
    if (!nf_out) return SML_OK;
    const int total = nf_out * SPEC_N;
    hipLaunchKernelGGL(k_post, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), sp->d, spec_in, (const int *)desc_dev, spec_out, total,
                       (double *)nullptr, total);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_spec_post_split(sml_spectral *sp, const double *spec_in, const int32_t *desc_dev, double *spec_out, int nf_out, double *spec_out2,
                                 int nf_out2, void *stream)
{
    SML_REQUIRE(sp && nf_out >= 0 && nf_out2 >= 0 && spec_in && desc_dev && (nf_out == 0 || spec_out) && (nf_out2 == 0 || spec_out2),
                "sml_spectral_spec_post_split: bad arguments");
    if (!(nf_out + nf_out2)) return SML_OK;
    const int total = (nf_out + nf_out2) * SPEC_N;
    hipLaunchKernelGGL(k_post, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), sp->d, spec_in, (const int *)desc_dev, spec_out, total,
                       spec_out2, nf_out * SPEC_N);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_spec(sml_spectral *sp, const double *vorg, double *vorm, int nf, void *stream)
{
    SML_REQUIRE(sp && nf >= 0 && (nf == 0 || (vorm && vorg)), "sml_spectral_spec: bad arguments");
    if (!nf) return SML_OK;
    return spec_scaled(sp, vorg, vorm, nf, 0, stream);
}

static int pointwise2(sml_spectral *sp, const double *p, const double *q, double *a, double *b, int nf, int is_vds, void *stream)
{
    const int total = nf * SPEC_N;
    hipLaunchKernelGGL(k_uvvds, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), sp->d, p, q, a, b, total, is_vds);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_vdspec(sml_spectral *sp, const double *ug, const double *vg, double *vorm, double *divm, int nf, int kcos, void *stream)
{
    SML_REQUIRE(sp && ug && vg && vorm && divm && nf >= 0, "sml_spectral_vdspec: bad arguments");
    if (!nf) return SML_OK;
    int rc = ensure_scratch(sp, nf);
    if (rc) return rc;
    double *um = sp->scratch, *vm = sp->scratch + (size_t)nf * SPEC_N;
    const int scale = kcos == 2 ? 1 : 2;                // cosgr for kcos=2, cosgr2 otherwise (:429-443)
    if ((rc = spec_scaled(sp, ug, um, nf, scale, stream))) return rc;
    if ((rc = spec_scaled(sp, vg, vm, nf, scale, stream))) return rc;
    return pointwise2(sp, um, vm, vorm, divm, nf, 1, stream);
}

int sml_spectral_uvspec(sml_spectral *sp, const double *vorm, const double *divm, double *ucosm, double *vcosm, int nf, void *stream)
{
    SML_REQUIRE(sp && vorm && divm && ucosm && vcosm && nf >= 0, "sml_spectral_uvspec: bad arguments");
    if (!nf) return SML_OK;
    return pointwise2(sp, vorm, divm, ucosm, vcosm, nf, 0, stream);
}

int sml_spectral_vds(sml_spectral *sp, const double *ucosm, const double *vcosm, double *vorm, double *divm, int nf, void *stream)
{
    SML_REQUIRE(sp && vorm && divm && ucosm && vcosm && nf >= 0, "sml_spectral_vds: bad arguments");
    if (!nf) return SML_OK;
    return pointwise2(sp, ucosm, vcosm, vorm, divm, nf, 1, stream);
}

int sml_spectral_grad(sml_spectral *sp, const double *psi, double *psdx, double *psdy, int nf, void *stream)
{
    SML_REQUIRE(sp && psi && psdx && psdy && nf >= 0, "sml_spectral_grad: bad arguments");
    if (!nf) return SML_OK;
    const int total = nf * SPEC_N;
    hipLaunchKernelGGL(k_grad, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), sp->d, psi, psdx, psdy, total);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

static int scale_op(sml_spectral *sp, const double *in, double *out, int nf, int op, void *stream)
{
    SML_REQUIRE(sp && in && out && nf >= 0, "spectral operator: bad arguments");
    if (!nf) return SML_OK;
    const int total = nf * SPEC_N;
    hipLaunchKernelGGL(k_scale, dim3((total + 255) / 256), dim3(256), 0, sml::as_stream(stream), sp->d, in, out, total, op);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_spectral_lap(sml_spectral *sp, const double *strm, double *vorm, int nf, void *stream) { return scale_op(sp, strm, vorm, nf, OP_LAP, stream); }
int sml_spectral_invlap(sml_spectral *sp, const double *vorm, double *strm, int nf, void *stream) { return scale_op(sp, vorm, strm, nf, OP_INVLAP, stream); }
int sml_spectral_trunct(sml_spectral *sp, double *vor, int nf, void *stream) { return scale_op(sp, vor, vor, nf, OP_TRUNCT, stream); }

// ---------------- F77 link-level drop-ins (host arrays, one field per call) ----------------
static void die(const char *where)
{
    fprintf(stderr, "speedyml_hip: %s failed: %s\n", where, sml_last_error());
    abort();     // the reference prints and `stop`s on library errors (src/mod_linalg.f90:18-22)
}

static sml_spectral *global_sp(const char *who)
{
    if (!g_sp) { sml::fail(SML_ERR_STATE, "%s called before parmtr_()", who); die(who); }
    if (!g_sp->hs_spec[0]) {
        for (auto &p : g_sp->hs_spec) if (hipMalloc((void **)&p, SPEC_N * sizeof(double)) != hipSuccess) { sml::fail(SML_ERR_HIP, "hipMalloc"); die(who); }
        for (auto &p : g_sp->hs_grid) if (hipMalloc((void **)&p, GRID_N * sizeof(double)) != hipSuccess) { sml::fail(SML_ERR_HIP, "hipMalloc"); die(who); }
    }
    return g_sp;
}

#define H2D(dst, src, n) if (hipMemcpy(dst, src, (n) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { sml::fail(SML_ERR_HIP, "hipMemcpy H2D"); die(__func__); }
#define D2H(dst, src, n) if (hipMemcpy(dst, src, (n) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { sml::fail(SML_ERR_HIP, "hipMemcpy D2H"); die(__func__); }
#define CK(call) if ((call) != SML_OK) die(__func__)

void parmtr_(const double *a)
{
    if (g_sp) sml_spectral_destroy(g_sp);
    g_sp = nullptr;
    sml_spectral *sp = nullptr;
    if (sml_spectral_create(*a, &sp) != SML_OK) die("parmtr_");
    g_sp = sp;
}

void inifft_(void) {}   // the twiddle table is part of sml_spectral_create

void grid_(const double *vorm, double *vorg, const int *kcos)
{
    sml_spectral *sp = global_sp("grid_");
    H2D(sp->hs_spec[0], vorm, SPEC_N);
    CK(sml_spectral_grid(sp, sp->hs_spec[0], sp->hs_grid[0], 1, *kcos, nullptr));
    D2H(vorg, sp->hs_grid[0], GRID_N);
}

void spec_(const double *vorg, double *vorm)
{
    sml_spectral *sp = global_sp("spec_");
    H2D(sp->hs_grid[0], vorg, GRID_N);
    CK(sml_spectral_spec(sp, sp->hs_grid[0], sp->hs_spec[0], 1, nullptr));
    D2H(vorm, sp->hs_spec[0], SPEC_N);
}

void vdspec_(const double *ug, const double *vg, double *vorm, double *divm, const int *kcos)
{
    sml_spectral *sp = global_sp("vdspec_");
    H2D(sp->hs_grid[0], ug, GRID_N);
    H2D(sp->hs_grid[1], vg, GRID_N);
    CK(sml_spectral_vdspec(sp, sp->hs_grid[0], sp->hs_grid[1], sp->hs_spec[0], sp->hs_spec[1], 1, *kcos, nullptr));
    D2H(vorm, sp->hs_spec[0], SPEC_N);
    D2H(divm, sp->hs_spec[1], SPEC_N);
}

void uvspec_(const double *vorm, const double *divm, double *ucosm, double *vcosm)
{
    sml_spectral *sp = global_sp("uvspec_");
    H2D(sp->hs_spec[0], vorm, SPEC_N);
    H2D(sp->hs_spec[1], divm, SPEC_N);
    CK(sml_spectral_uvspec(sp, sp->hs_spec[0], sp->hs_spec[1], sp->hs_spec[2], sp->hs_spec[3], 1, nullptr));
    D2H(ucosm, sp->hs_spec[2], SPEC_N);
    D2H(vcosm, sp->hs_spec[3], SPEC_N);
}

void vds_(const double *ucosm, const double *vcosm, double *vorm, double *divm)
{
    sml_spectral *sp = global_sp("vds_");
    H2D(sp->hs_spec[0], ucosm, SPEC_N);
    H2D(sp->hs_spec[1], vcosm, SPEC_N);
    CK(sml_spectral_vds(sp, sp->hs_spec[0], sp->hs_spec[1], sp->hs_spec[2], sp->hs_spec[3], 1, nullptr));
    D2H(vorm, sp->hs_spec[2], SPEC_N);
    D2H(divm, sp->hs_spec[3], SPEC_N);
}

void grad_(const double *psi, double *psdx, double *psdy)
{
    sml_spectral *sp = global_sp("grad_");
    H2D(sp->hs_spec[0], psi, SPEC_N);
    CK(sml_spectral_grad(sp, sp->hs_spec[0], sp->hs_spec[1], sp->hs_spec[2], 1, nullptr));
    D2H(psdx, sp->hs_spec[1], SPEC_N);
    D2H(psdy, sp->hs_spec[2], SPEC_N);
}

void lap_(const double *strm, double *vorm)
{
    sml_spectral *sp = global_sp("lap_");
    H2D(sp->hs_spec[0], strm, SPEC_N);
    CK(sml_spectral_lap(sp, sp->hs_spec[0], sp->hs_spec[1], 1, nullptr));
    D2H(vorm, sp->hs_spec[1], SPEC_N);
}

void invlap_(const double *vorm, double *strm)
{
    sml_spectral *sp = global_sp("invlap_");
    H2D(sp->hs_spec[0], vorm, SPEC_N);
    CK(sml_spectral_invlap(sp, sp->hs_spec[0], sp->hs_spec[1], 1, nullptr));
    D2H(strm, sp->hs_spec[1], SPEC_N);
}

void trunct_(double *vor)
{
    sml_spectral *sp = global_sp("trunct_");
    H2D(sp->hs_spec[0], vor, SPEC_N);
    CK(sml_spectral_trunct(sp, sp->hs_spec[0], 1, nullptr));
    D2H(vor, sp->hs_spec[0], SPEC_N);
}

}  // extern "C"
