// Ridge-regression training on gfx950: fp64-MFMA accumulation of the Gram matrices and the LU solve.
//
// Replaces chunking_matmul (src/mod_reservoir.f90:1645-1701: temp = matmul(targetdata, transpose(aug)); DGEMM('N','N',
// n,n,m, aug, transpose(aug))), fit_chunk_hybrid (:1235-1334) and mldivide = dgesv (src/mod_linalg.f90:109-151).
//
//   k_gemm_acc : C(i,j) += alpha * sum_k A(i,k) B(j,k) with v_mfma_f64_4x4x4_4b_f64 (see the kernel).  128x128 tile per 256-thread
//                workgroup, 64x64 per wavefront (4x16 MFMA tiles of 16x4, 128 accumulator VGPRs), K-tile 16 staged through
//                LDS with a 144-double row stride (rows land 128 B apart modulo the 256-B bank period, so the
//                16-lane x 4-row operand reads are conflict free).  The MFMA is issued as D[j][i] = B.A so that the
//                accumulator's lane index runs along i, the contiguous direction of the column-major C: every
//                16-lane group stores 128 contiguous bytes.  For the symmetric update only tiles on or below the
//                diagonal are computed; sml_train_fit mirrors them once before factorising.
//   LU         : right-looking blocked LU with partial pivoting (dgesv semantics) on [A | B] so that the forward
//                substitution of the right-hand sides rides along: panel (one workgroup, pivot search by
//                workgroup reduction) -> row interchanges outside the panel -> U12 = L11^-1 A12 -> trailing update
//                with k_gemm_acc (alpha = -1) ; then a blocked back substitution.
// All matrices are column-major fp64, as in the reference.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KT = 16, LDS_LD = 144, GT = 256;
constexpr int LU_NB = 32;      // LU panel width

// ---- shared pieces of the two GEMM kernels ----
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction, 16 cycles, measured 75 TFLOP/s on this part
// versus 36-48 TFLOP/s for v_mfma_f64_16x16x4_f64 (profiles/micro/mfma_f64_peak.hip).  Measured lane layout
// (profiles/micro/mfma_f64_4x4x4_layout.hip): A operand lane = 16 k + 4 blk + i, B operand lane = 16 k + 4 blk + j,
// D lane = 16 i + 4 blk + j.  C's column index J goes on the MFMA's i and C's row index I on (blk, j), so one instruction
// produces a 16 (I) x 4 (J) tile of C whose 16 consecutive lanes are 16 consecutive rows of one column: contiguous
// 128-byte stores into the column-major C.  A wavefront owns 64 x 64 of C = 4 x 16 such tiles (128 accumulator VGPRs).
struct WavePos { int wi, wj, l15, l4, l3; };

__device__ __forceinline__ WavePos wave_pos()
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return {(wave & 1) * 64, (wave >> 1) * 64, lane & 15, lane >> 4, lane & 3};
}

template <int KTILE>
__device__ __forceinline__ void mma_ktile(const double (*As)[LDS_LD], const double (*Bs)[LDS_LD], const WavePos &w, double (&acc)[4][16])
{
    // All 20 operand reads of a k-step are issued before its 64 MFMAs (profiles/micro/mfma_f64_pattern.hip: this shape
    // sustains 71-74 TFLOP/s with the operands re-read from LDS every k-step; splitting the B fragment in halves to save
    // registers put a wait in front of every 32 MFMAs and ran at 29 TFLOP/s).
#pragma unroll
    for (int kk = 0; kk < KTILE; kk += 4) {
        double af[4], bf[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = As[kk + w.l4][w.wi + 16 * t + w.l15];       // A(I0 + 4 blk + j, k): MFMA "B" operand
#pragma unroll
        for (int u = 0; u < 16; ++u) bf[u] = Bs[kk + w.l4][w.wj + 4 * u + w.l3];        // B(J0 + i, k): MFMA "A" operand
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(bf[u], af[t], acc[t][u], 0, 0, 0);
    }
}

// lane l holds C(I0 + 16 t + (l & 15), J0 + 4 u + (l >> 4)).  C += alpha * acc as a read-modify-write in chunks of 16
// elements per lane: the 16 loads of a chunk are issued together (one exposed round trip per chunk, not per element).
__device__ __forceinline__ void store_tile(double *__restrict__ C, long ldc, int M, int N, int i0, int j0, const WavePos &w, double alpha,
                                           const double (&acc)[4][16])
{
#pragma unroll
    for (int u0 = 0; u0 < 16; u0 += 4) {
        double cv[4][4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int j = j0 + w.wj + 4 * (u0 + uu) + w.l4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + w.wi + 16 * t + w.l15;
                cv[uu][t] = (i < M && j < N) ? C[(long)j * ldc + i] : 0.0;
            }
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int j = j0 + w.wj + 4 * (u0 + uu) + w.l4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + w.wi + 16 * t + w.l15;
                if (i < M && j < N) C[(long)j * ldc + i] = cv[uu][t] + alpha * acc[t][u0 + uu];
            }
        }
    }
}

// Staging of one 128 x 16 operand tile through registers: element (r, k) = p[r * sr + k * sk], zero outside rows < nrows,
// k < K.  Split in two halves (T14-style): stage_load issues the 8 global loads early, stage_write puts them into LDS
// after the barrier, so the loads of K-tile t+1 fly under the MFMAs of K-tile t.
template <bool K_CONTIG>
__device__ __forceinline__ void stage_load(const double *__restrict__ p, long sr, long sk, int r0, int nrows, int k0, int K, double (&v)[8])
{
    if (!K_CONTIG) {
        const int r = threadIdx.x & 127, kb = threadIdx.x >> 7;
        const bool rok = r0 + r < nrows;
        const double *q = p + (long)(r0 + r) * sr + (long)(k0 + kb) * sk;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (rok && k0 + kb + 2 * i < K) ? q[(long)(2 * i) * sk] : 0.0;
    } else {
        const int k = threadIdx.x & 15, rb = threadIdx.x >> 4;
        const bool kok = k0 + k < K;
        const double *q = p + (long)(r0 + rb) * sr + (long)(k0 + k) * sk;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (kok && r0 + rb + 16 * i < nrows) ? q[(long)(16 * i) * sr] : 0.0;
    }
}

template <bool K_CONTIG>
__device__ __forceinline__ void stage_write(const double (&v)[8], double (*dst)[LDS_LD])
{
    if (!K_CONTIG) {
        const int r = threadIdx.x & 127, kb = threadIdx.x >> 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[kb + 2 * i][r] = v[i];
    } else {
        const int k = threadIdx.x & 15, rb = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[k][rb + 16 * i] = v[i];
    }
}

// General kernel (any strides, any K, edge tiles): C[i + j*ldc] += alpha * sum_k A(i,k) * B(j,k).
// lower_only: skip tiles strictly above the diagonal.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(GT, 2) void k_gemm_acc(const double *__restrict__ A, long sai, long sak, const double *__restrict__ B,
                                                     long sbj, long sbk, double *__restrict__ C, long ldc, int M, int N, int K,
                                                     double alpha, int lower_only)
{
    __shared__ double As[KT][LDS_LD];
    __shared__ double Bs[KT][LDS_LD];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (lower_only && tj > ti) return;
    const int i0 = ti * BM, j0 = tj * BN;
    const WavePos w = wave_pos();
    double acc[4][16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[a][b] = 0.0;
    double ra[8], rb[8];
    stage_load<A_KC>(A, sai, sak, i0, M, 0, K, ra);
    stage_load<B_KC>(B, sbj, sbk, j0, N, 0, K, rb);
    for (int k0 = 0; k0 < K; k0 += KT) {
        __syncthreads();                       // everyone is done reading the previous tile
        stage_write<A_KC>(ra, As);
        stage_write<B_KC>(rb, Bs);
        __syncthreads();
        if (k0 + KT < K) {                     // next tile's loads stay in flight during the MFMAs below
            stage_load<A_KC>(A, sai, sak, i0, M, k0 + KT, K, ra);
            stage_load<B_KC>(B, sbj, sbk, j0, N, k0 + KT, K, rb);
        }
        mma_ktile<KT>(As, Bs, w, acc);
    }
    store_tile(C, ldc, M, N, i0, j0, w, alpha, acc);
}

// Fast path for the Gram updates: both operands row-contiguous (A(i,k) = A[i + k*lda], B(j,k) = B[j + k*ldb]), 16-byte
// aligned columns, K a multiple of 8.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): one
// wavefront instruction moves 64 lanes x 16 B = one 128-row k-column of the tile straight into its (padded) LDS row --
// no staging registers, no ds_write.  Four LDS buffers of 8 k-columns form a ring; the DMA runs THREE K-tiles ahead of
// the MFMAs (a counted s_waitcnt vmcnt(N) + a raw s_barrier per K-tile; __syncthreads would drain the ring), because
// one tile of look-ahead (2048 MFMA cycles) does not cover a loaded HBM/Infinity-Cache round trip.
// Rows past M/N are clamped to valid memory (their products only feed rows that are never stored).
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
constexpr int DKT = 8, NBUF = 4, AHEAD = 3;

// Work decomposition (mode): 0 = 2-D grid over all tiles of an M x N product; 1 = 1-D list of the tiles on or below the
// diagonal (tile L = blockIdx.x, row-major in the triangle), so no workgroups are spent on the upper triangle.
//
// Where the time goes (m = 2920, 1035 + 45 tiles, MI355X, compile-time ablations of this kernel, executed-flop TF/s):
//   full kernel 42.2 | MFMAs only (no DMA, no barrier, operands from registers) 57.8 | + LDS operand reads 57.0 |
//   DMA + barrier but operands from registers 49.9.  So the MFMA stream of this tiling tops out at 58 TF/s under load
//   (the 69-75 TF/s of profiles/micro/mfma_f64_peak.hip is an idle-chip figure; zeros instead of random data give +7 %,
//   i.e. clocks), the operand reads are free on their own and cost 15 % once the DMA writes share the LDS, and DMA +
//   barrier cost 14 %.  Measured and found irrelevant (all within 2 %): K-tile 8 vs 16, ring depth 3 vs 4, look-ahead 1-3
//   tiles, an XCD-aware super-tile map (L2 hit rate is 52 % either way), splitting the ragged last round along K (the
//   two resident workgroups per CU already keep the chip busy until the last tile).
__device__ __forceinline__ void tri_tile(int L, int &ti, int &tj)
{
    ti = (int)((sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
    while ((long)(ti + 1) * (ti + 2) / 2 <= L) ++ti;
    while ((long)ti * (ti + 1) / 2 > L) --ti;
    tj = L - (int)((long)ti * (ti + 1) / 2);
}

__global__ __launch_bounds__(GT, 2) void k_gemm_nt_dma(const double *__restrict__ A, long lda, const double *__restrict__ B, long ldb,
                                                        double *__restrict__ C, long ldc, int M, int N, int K, double alpha, int mode)
{
    __shared__ __attribute__((aligned(16))) double S[NBUF][2][DKT][LDS_LD];       // [ring slot][operand][k][row]  73.7 KB
    int ti, tj;
    if (mode == 0) { ti = blockIdx.x; tj = blockIdx.y; }
    else tri_tile((int)blockIdx.x, ti, tj);
    const int i0 = ti * BM, j0 = tj * BN;
    const WavePos w = wave_pos();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ra = min(i0 + 2 * lane, (M - 2) & ~1), rb = min(j0 + 2 * lane, (N - 2) & ~1);
    const double *pa = A + ra, *pb = B + rb;
    double acc[4][16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[a][b] = 0.0;

    // each wave moves 2 of the 8 k-columns of each operand: 4 DMAs per wave per K-tile
    auto issue = [&](int slot, int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = wave * 2 + q;
            __builtin_amdgcn_global_load_lds((gptr_t)(pa + (long)(k0 + k) * lda), (lptr_t)&S[slot][0][k][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(pb + (long)(k0 + k) * ldb), (lptr_t)&S[slot][1][k][0], 16, 0, 0);
        }
    };
    const int ntiles = K / DKT;
    for (int t = 0; t < AHEAD && t < ntiles; ++t) issue(t, t * DKT);
    for (int t = 0; t < ntiles; ++t) {
        // tile t must have landed; the younger tiles (4 DMAs each, at most two of them) may stay in flight
        const int younger = min(AHEAD - 1, ntiles - 1 - t);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // every wave's share of tile t is in LDS; everyone is done with slot (t-1)%4
        if (t + AHEAD < ntiles) issue((t + AHEAD) % NBUF, (t + AHEAD) * DKT);
        mma_ktile<DKT>(S[t % NBUF][0], S[t % NBUF][1], w, acc);
    }
    store_tile(C, ldc, M, N, i0, j0, w, alpha, acc);
}

__global__ void k_symmetrize(double *__restrict__ c, int n)
{   // upper <- lower^T, tile-wise through LDS so both the read and the write are contiguous
    __shared__ double t[32][33];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj >= bi) return;                       // strictly-lower tiles feed the strictly-upper ones
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) {
        const int i = bi * 32 + tx, j = bj * 32 + y;
        t[y][tx] = (i < n && j < n) ? c[(long)i + (long)j * n] : 0.0;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int i = bj * 32 + tx, j = bi * 32 + y;      // transposed position
        if (i < n && j < n) c[(long)i + (long)j * n] = t[tx][y];
    }
}

__global__ void k_symmetrize_diag(double *__restrict__ c, int n)
{   // inside the diagonal 32x32 tiles
    const int b = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) {
        const int i = b * 32 + tx, j = b * 32 + y;
        if (i < n && j < n && j > i) c[(long)i + (long)j * n] = c[(long)j + (long)i * n];
    }
}

// W = [ (C + reg)^T | (B + prior)^T ]   (fit_chunk_hybrid :1261-1309); C is symmetric at this point
__global__ void k_build_system(const double *__restrict__ c, const double *__restrict__ b, double *__restrict__ w, int n_aug, int n_model,
                               int n_out, double reg_model, double reg_res, double prior_diag)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_aug * (n_aug + n_out);
    if (t >= total) return;
    const int i = (int)(t % n_aug);
    const int j = (int)(t / n_aug);
    double v;
    if (j < n_aug) {
        v = c[(long)i + (long)j * n_aug];                 // a_trans = transpose(C); C is symmetric here, read it contiguously
        if (i == j) v = v + (i < n_model ? reg_model : reg_res);
    } else {
        const int o = j - n_aug;                          // b_trans(i, o) = B(o, i) + prior(o, i)
        v = b[(long)o + (long)i * n_out];
        if (o == i && o < n_model) v = v + prior_diag;
    }
    w[t] = v;
}

__global__ void k_extract_wout(const double *__restrict__ w, double *__restrict__ wout, int n_aug, int n_out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_aug * n_out) return;
    const int o = (int)(t % n_out), i = (int)(t / n_out);
    wout[t] = w[(long)i + (long)(n_aug + o) * n_aug];     // wout(o,i) = Z(i,o)
}

// ---- LU panel: columns [k0, k0+nb) of W, rows [k0, n); one workgroup ----
constexpr int PT = 1024;
__global__ __launch_bounds__(PT) void k_panel(double *__restrict__ w, int n, int k0, int nb, int *__restrict__ ipiv, int *__restrict__ info)
{
    __shared__ double sval[PT / 64];
    __shared__ int sidx[PT / 64];
    __shared__ int spiv;
    __shared__ double srow[LU_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = 0; c < nb; ++c) {
        const int col = k0 + c;
        double *wc = w + (long)col * n;
        // pivot search: first maximum of |a| (dgetf2 / idamax)
        double best = -1.0; int bi = n;
        for (int r = col + tid; r < n; r += PT) {
            const double a = fabs(wc[r]);
            if (a > best) { best = a; bi = r; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double b2 = sval[0]; int i2 = sidx[0];
            for (int q = 1; q < PT / 64; ++q)
                if (sval[q] > b2 || (sval[q] == b2 && sidx[q] < i2)) { b2 = sval[q]; i2 = sidx[q]; }
            spiv = i2;
            ipiv[col] = i2;
            if (b2 == 0.0 && *info == 0) *info = col + 1;
        }
        __syncthreads();
        const int p = spiv;
        // interchange rows col <-> p inside the panel and stage the pivot row
        if (tid < nb) {
            double *q = w + (long)(k0 + tid) * n;
            const double a = q[col], b = q[p];
            if (p != col) { q[col] = b; q[p] = a; }
            srow[tid] = b;
        }
        __syncthreads();
        const double piv = srow[c];
        if (piv != 0.0) {
            const double inv = 1.0 / piv;
            for (int r = col + 1 + tid; r < n; r += PT) {
                const double l = wc[r] * inv;
                wc[r] = l;
                for (int cc = c + 1; cc < nb; ++cc) {
                    double *q = w + (long)(k0 + cc) * n;
                    q[r] = q[r] - l * srow[cc];
                }
            }
        }
        __syncthreads();
    }
}

// The same factorisation of a panel, blocked: the panel's columns are taken SW at a time; a thread keeps its rows of those SW columns
// in registers while their pivots are found and eliminated, and the panel's remaining columns are updated ONCE per sub-panel
// (u = L11^-1 a12, then a22 -= l21 u, every subtraction in the order the column-by-column form does them, so the bits are the same)
// instead of once per column.  k_panel walks all trailing panel columns through memory for every one of its 32 pivots: 0.5 ms per
// panel, 93 of the 114 ms of a 5892 x 5892 factorisation.  Rows below the panel's first row: at most PBT * RPT.
constexpr int SW = 6, PBT = 512, RPT = 12;       // 96 doubles of the sub-panel per thread: needs the 256-VGPR budget of 8 waves per CU
__global__ __launch_bounds__(PBT) void k_panel_blocked(double *__restrict__ w, int n, int k0, int nb, int *__restrict__ ipiv, int *__restrict__ info)
{
    __shared__ double sval[2 * (PBT / 64)];     // per-wave maxima, double-buffered by column parity (one barrier per reduction)
    __shared__ int sidx[2 * (PBT / 64)];
    __shared__ double rowPC[2][2][SW];          // [column parity][pivot row | current row][sub-panel column]: the rows being interchanged
    __shared__ double l11[SW][SW];              // unit-lower factor of the sub-panel's diagonal block
    __shared__ double u12[SW][LU_NB];           // the sub-panel's rows of the panel's remaining columns, eliminated
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < nb; c0 += SW) {
        const int sw = min(SW, nb - c0), top = k0 + c0;
        double a[RPT][SW];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = k0 + tid + i * PBT;
#pragma unroll
            for (int j = 0; j < SW; ++j) a[i][j] = (r < n && j < sw) ? w[(long)(top + j) * n + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            if (j < sw) {                                   // uniform
                const int col = top + j;
                // pivot search: first maximum of |a| (dgetf2 / idamax)
                double best = -1.0; int bi = n;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = k0 + tid + i * PBT;
                    if (r >= col && r < n) {
                        const double v = fabs(a[i][j]);
                        if (v > best) { best = v; bi = r; }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (lane == 0) { sval[(j & 1) * (PBT / 64) + wave] = best; sidx[(j & 1) * (PBT / 64) + wave] = bi; }
                __syncthreads();
                // every thread finishes the reduction for itself (8 LDS broadcasts): no second barrier for a shared result
                double b2 = sval[(j & 1) * (PBT / 64)]; int p = sidx[(j & 1) * (PBT / 64)];
#pragma unroll
                for (int q = 1; q < PBT / 64; ++q) {
                    const double sv = sval[(j & 1) * (PBT / 64) + q];
                    const int si = sidx[(j & 1) * (PBT / 64) + q];
                    if (sv > b2 || (sv == b2 && si < p)) { b2 = sv; p = si; }
                }
                if (tid == 0) {
                    ipiv[col] = p;
                    if (b2 == 0.0 && *info == 0) *info = col + 1;
                }
                double *rowP = rowPC[j & 1][0], *rowC = rowPC[j & 1][1];
                // interchange rows col <-> p: inside the sub-panel through LDS, in the panel's other columns in memory
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = k0 + tid + i * PBT;
                    if (r == p) {
#pragma unroll
                        for (int jj = 0; jj < SW; ++jj) rowP[jj] = a[i][jj];
                    }
                    if (r == col) {
#pragma unroll
                        for (int jj = 0; jj < SW; ++jj) rowC[jj] = a[i][jj];
                    }
                }
                __syncthreads();
                // (the panel's other columns: in memory, by one thread each; issued after the barrier so that its round trip
                // runs under the elimination below instead of holding everybody at the barrier)
                if (tid < nb && (tid < c0 || tid >= c0 + sw) && p != col) {
                    double *q = w + (long)(k0 + tid) * n;
                    const double x = q[col], y = q[p];
                    q[col] = y; q[p] = x;
                }
                if (p != col) {
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = k0 + tid + i * PBT;
                        if (r == col) {
#pragma unroll
                            for (int jj = 0; jj < SW; ++jj) a[i][jj] = rowP[jj];
                        } else if (r == p) {
#pragma unroll
                            for (int jj = 0; jj < SW; ++jj) a[i][jj] = rowC[jj];
                        }
                    }
                }
                const double piv = rowP[j];
                if (piv != 0.0) {
                    const double inv = 1.0 / piv;
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = k0 + tid + i * PBT;
                        if (r > col && r < n) {
                            const double l = a[i][j] * inv;
                            a[i][j] = l;
#pragma unroll
                            for (int jj = 0; jj < SW; ++jj)
                                if (jj > j) a[i][jj] = a[i][jj] - l * rowP[jj];
                        }
                    }
                }
                // no barrier here: the next column uses the other halves of sval / sidx / rowPC, and its first barrier orders
                // everything else
            }
        }
        // the sub-panel goes back to memory; its diagonal block's L part is shared for the elimination of the remaining columns
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = k0 + tid + i * PBT;
            if (r >= top && r < n) {
#pragma unroll
                for (int j = 0; j < SW; ++j)
                    if (j < sw) w[(long)(top + j) * n + r] = a[i][j];
                if (r < top + sw) {
#pragma unroll
                    for (int jj = 0; jj < SW; ++jj) l11[r - top][jj] = a[i][jj];
                }
            }
        }
        const int rem = nb - c0 - sw;                       // panel columns to the right of the sub-panel
        if (rem > 0) {
            __syncthreads();
            if (tid < rem) {                                // u = L11^-1 a12, one thread per column, subtractions in elimination order
                double *q = w + (long)(top + sw + tid) * n;
                double u[SW];
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    if (j < sw) {
                        double v = q[top + j];
#pragma unroll
                        for (int jj = 0; jj < SW; ++jj)
                            if (jj < j) v = v - l11[j][jj] * u[jj];
                        u[j] = v;
                        q[top + j] = v;
                        u12[j][tid] = v;
                    }
                }
            }
            __syncthreads();
            for (int cc = 0; cc < rem; ++cc) {              // a22 -= l21 u, row by row in registers' L
                double *q = w + (long)(top + sw + cc) * n;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = k0 + tid + i * PBT;
                    if (r >= top + sw && r < n) {
                        double v = q[r];
#pragma unroll
                        for (int j = 0; j < SW; ++j)
                            if (j < sw) v = v - a[i][j] * u12[j][cc];
                        q[r] = v;
                    }
                }
            }
        }
        __syncthreads();                                    // the next sub-panel reads what this one wrote
    }
}

// apply the panel's interchanges to every column outside the panel (columns [0,k0) and [k0+nb, ncols))
__global__ void k_swap(double *__restrict__ w, int n, int ncols, int k0, int nb, const int *__restrict__ ipiv)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols - nb) return;
    if (col >= k0) col += nb;
    double *q = w + (long)col * n;
    for (int c = 0; c < nb; ++c) {
        const int r = k0 + c, p = ipiv[r];
        if (p != r) { const double a = q[r]; q[r] = q[p]; q[p] = a; }
    }
}

// U12 = L11^-1 A12: one thread per column of A12 (columns [k0+nb, ncols)); unit-lower L11 and the thread's column
// segment live in LDS (xs[r][thread] is conflict free), so nothing is indexed dynamically in registers.
constexpr int TRSM_T = 64;
__global__ __launch_bounds__(TRSM_T) void k_trsm_lower(double *__restrict__ w, int n, int ncols, int k0, int nb)
{
    __shared__ double L[LU_NB][LU_NB + 1];
    __shared__ double xs[LU_NB][TRSM_T];
    for (int e = threadIdx.x; e < nb * nb; e += TRSM_T) {
        const int r = e % nb, c = e / nb;
        L[r][c] = w[(long)(k0 + r) + (long)(k0 + c) * n];
    }
    __syncthreads();
    const int col = k0 + nb + blockIdx.x * TRSM_T + threadIdx.x;
    if (col >= ncols) return;
    double *q = w + (long)col * n + k0;
    for (int r = 0; r < nb; ++r) xs[r][threadIdx.x] = q[r];
    for (int r = 1; r < nb; ++r) {
        double s = xs[r][threadIdx.x];
        for (int c = 0; c < r; ++c) s = s - L[r][c] * xs[c][threadIdx.x];
        xs[r][threadIdx.x] = s;
    }
    for (int r = 0; r < nb; ++r) q[r] = xs[r][threadIdx.x];
}

// back substitution block: X_kb = U_kk^-1 Y_kb for every right-hand side (columns [n, ncols))
__global__ __launch_bounds__(TRSM_T) void k_trsm_upper(double *__restrict__ w, int n, int ncols, int k0, int nb)
{
    __shared__ double U[LU_NB][LU_NB + 1];
    __shared__ double xs[LU_NB][TRSM_T];
    for (int e = threadIdx.x; e < nb * nb; e += TRSM_T) {
        const int r = e % nb, c = e / nb;
        U[r][c] = w[(long)(k0 + r) + (long)(k0 + c) * n];
    }
    __syncthreads();
    const int col = n + blockIdx.x * TRSM_T + threadIdx.x;
    if (col >= ncols) return;
    double *q = w + (long)col * n + k0;
    for (int r = 0; r < nb; ++r) xs[r][threadIdx.x] = q[r];
    for (int r = nb - 1; r >= 0; --r) {
        double s = xs[r][threadIdx.x];
        for (int c = r + 1; c < nb; ++c) s = s - U[r][c] * xs[c][threadIdx.x];
        xs[r][threadIdx.x] = s / U[r][r];
    }
    for (int r = 0; r < nb; ++r) q[r] = xs[r][threadIdx.x];
}

template <bool A_KC, bool B_KC>
int gemm(const double *A, long sai, long sak, const double *B, long sbj, long sbk, double *C, long ldc, int M, int N, int K,
         double alpha, int lower_only, hipStream_t st)
{
    if (M <= 0 || N <= 0 || K <= 0) return SML_OK;
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    hipLaunchKernelGGL((k_gemm_acc<A_KC, B_KC>), grid, dim3(GT), 0, st, A, sai, sak, B, sbj, sbk, C, ldc, M, N, K, alpha, lower_only);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

// C += alpha * A * B^T for row-contiguous operands: LDS-DMA kernel on the part of K that is a multiple of 16 (when the
// alignment conditions hold), general kernel on the tail.
int gemm_nt(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int M, int N, int K, double alpha, int lower_only,
            hipStream_t st)
{
    if (M <= 0 || N <= 0 || K <= 0) return SML_OK;
    auto ok = [](const double *p, long ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 1) == 0; };
    int kmain = 0;
    // short K (the shipped batch size m = 98) is bound by the read-modify-write of C: one pass with the general kernel
    // beats main + tail passes.  Long K (m = 2920, the 40-year configuration) takes the LDS-DMA kernel.
    if (ok(A, lda) && ok(B, ldb) && M >= 2 && N >= 2 && K >= 512) {
        kmain = (K / DKT) * DKT;
        const int nbi = (M + BM - 1) / BM, nbj = (N + BN - 1) / BN;
        if (!lower_only || nbi != nbj)
            hipLaunchKernelGGL(k_gemm_nt_dma, dim3(nbi, nbj), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain, alpha, 0);
        else
            hipLaunchKernelGGL(k_gemm_nt_dma, dim3(nbi * (nbi + 1) / 2), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain, alpha, 1);
        SML_HIP(hipGetLastError());
    }
    if (kmain < K)
        return gemm<false, false>(A + (long)kmain * lda, 1, lda, B + (long)kmain * ldb, 1, ldb, C, ldc, M, N, K - kmain, alpha, lower_only, st);
    return SML_OK;
}

}  // namespace

extern "C" {

int sml_train_accumulate(const double *states, const double *model, const double *y, int n, int n_model, int n_out, int m,
                         double *c, double *b, void *stream)
{
    SML_REQUIRE(states && y && c && b && n > 0 && n_model >= 0 && n_out > 0 && m > 0 && (n_model == 0 || model),
                "sml_train_accumulate: bad arguments");
    hipStream_t st = sml::as_stream(stream);
    const int n_aug = n + n_model;
    int rc;
    // aug = [model ; states] is never materialised: the four (model|states) x (model|states) blocks of C and the two
    // blocks of B are separate launches on the original arrays.  Only tiles on/below the diagonal of C are updated.
    // The skinny products (132- and 136-row operands: 90 tiles each) would each cost a full tile latency (the K loop of
    // a tile is sequential: 0.36 ms at m = 2920) if they queued behind the big one; they are forked onto two side
    // streams so they fill the SYRK's ragged last round instead.
    static hipStream_t side[2] = {nullptr, nullptr};
    static hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    if (!side[0]) {
        for (int i = 0; i < 2; ++i) {
            SML_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
            SML_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
        SML_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    }
    SML_HIP(hipEventRecord(ev_fork, st));
    for (int i = 0; i < 2; ++i) SML_HIP(hipStreamWaitEvent(side[i], ev_fork, 0));
    double *c_ss = c + (long)n_model + (long)n_model * n_aug;
    if ((rc = gemm_nt(states, n, states, n, c_ss, n_aug, n, n, m, 1.0, 1, st))) return rc;
    if (n_model) {
        if ((rc = gemm_nt(model, n_model, model, n_model, c, n_aug, n_model, n_model, m, 1.0, 0, side[0]))) return rc;
        // lower-left block: rows = states, cols = model
        if ((rc = gemm_nt(states, n, model, n_model, c + n_model, n_aug, n, n_model, m, 1.0, 0, side[0]))) return rc;
        if ((rc = gemm_nt(y, n_out, model, n_model, b, n_out, n_out, n_model, m, 1.0, 0, side[1]))) return rc;
    }
    if ((rc = gemm_nt(y, n_out, states, n, b + (long)n_model * n_out, n_out, n_out, n, m, 1.0, 0, side[1]))) return rc;
    for (int i = 0; i < 2; ++i) {
        SML_HIP(hipEventRecord(ev_join[i], side[i]));
        SML_HIP(hipStreamWaitEvent(st, ev_join[i], 0));
    }
    return SML_OK;
}

int sml_train_symmetrize(double *c, int n_aug, void *stream)
{
    SML_REQUIRE(c && n_aug > 0, "sml_train_symmetrize: bad arguments");
    const int nb = (n_aug + 31) / 32;
    hipLaunchKernelGGL(k_symmetrize, dim3(nb, nb), dim3(256), 0, sml::as_stream(stream), c, n_aug);
    SML_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_symmetrize_diag, dim3(nb), dim3(256), 0, sml::as_stream(stream), c, n_aug);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

// enqueue one ridge solve on `st` (no synchronisation): W, ipiv are scratch of size n_aug*(n_aug+n_out) / n_aug, info_dev one int
static int fit_enqueue(double *c, const double *b, int n, int n_model, int n_out, double beta_res, double beta_model, double prior_val,
                       int using_prior, double *wout, double *w, int *ipiv, int *info, hipStream_t st)
{
    const int n_aug = n + n_model, ncols = n_aug + n_out;
    constexpr int NB = LU_NB;
    int rc;
    if ((rc = sml_train_symmetrize(c, n_aug, (void *)st))) return rc;
    SML_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
    // with a prior the betas enter squared (quirk Q8, src/mod_reservoir.f90:1271-1290)
    const double reg_model = using_prior ? beta_model * beta_model : beta_model;
    const double reg_res = using_prior ? beta_res * beta_res : beta_res;
    const double prior_diag = using_prior ? prior_val * (beta_model * beta_model) : 0.0;
    const long total = (long)n_aug * ncols;
    hipLaunchKernelGGL(k_build_system, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, c, b, w, n_aug, n_model, n_out,
                       reg_model, reg_res, prior_diag);
    rc = SML_OK;
    for (int k0 = 0; k0 < n_aug && rc == SML_OK; k0 += NB) {
        const int nb = std::min(NB, n_aug - k0);
        static const int old_panel = getenv("SML_LU_OLD_PANEL") ? atoi(getenv("SML_LU_OLD_PANEL")) : 0;
        if (old_panel || n_aug - k0 > PBT * RPT)
            hipLaunchKernelGGL(k_panel, dim3(1), dim3(PT), 0, st, w, n_aug, k0, nb, ipiv, info);
        else
            hipLaunchKernelGGL(k_panel_blocked, dim3(1), dim3(PBT), 0, st, w, n_aug, k0, nb, ipiv, info);
        hipLaunchKernelGGL(k_swap, dim3((ncols - nb + 255) / 256), dim3(256), 0, st, w, n_aug, ncols, k0, nb, ipiv);
        const int rest_cols = ncols - (k0 + nb), rest_rows = n_aug - (k0 + nb);
        if (rest_cols > 0) {
            hipLaunchKernelGGL(k_trsm_lower, dim3((rest_cols + TRSM_T - 1) / TRSM_T), dim3(TRSM_T), 0, st, w, n_aug, ncols, k0, nb);
            if (rest_rows > 0)
                rc = gemm<false, true>(w + (k0 + nb) + (long)k0 * n_aug, 1, n_aug,                  // L21(i,k)
                                       w + k0 + (long)(k0 + nb) * n_aug, n_aug, 1,                 // U12(k,j) as B(j,k)
                                       w + (k0 + nb) + (long)(k0 + nb) * n_aug, n_aug, rest_rows, rest_cols, nb, -1.0, 0, st);
        }
    }
    // back substitution on the right-hand sides
    for (int k0 = ((n_aug - 1) / NB) * NB; k0 >= 0 && rc == SML_OK; k0 -= NB) {
        const int nb = std::min(NB, n_aug - k0);
        hipLaunchKernelGGL(k_trsm_upper, dim3((n_out + TRSM_T - 1) / TRSM_T), dim3(TRSM_T), 0, st, w, n_aug, ncols, k0, nb);
        if (k0 > 0)
            rc = gemm<false, true>(w + (long)k0 * n_aug, 1, n_aug,                                  // U(i, k0+k), i < k0
                                   w + k0 + (long)n_aug * n_aug, n_aug, 1,                          // X(k0+k, j) as B(j,k)
                                   w + (long)n_aug * n_aug, n_aug, k0, n_out, nb, -1.0, 0, st);
    }
    if (rc == SML_OK) {
        const long tw = (long)n_aug * n_out;
        hipLaunchKernelGGL(k_extract_wout, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, st, w, wout, n_aug, n_out);
        if (hipGetLastError() != hipSuccess) rc = sml::fail(SML_ERR_HIP, "k_extract_wout launch failed");
    }
    return rc;
}

// Several independent ridge solves at once.  One LU is latency-bound on its panel kernel (a single workgroup finds each
// pivot), so up to FIT_STREAMS systems are kept in flight on separate streams: the panels of one system run beside the
// trailing updates of the others.  All systems share the sizes (n, n_model, n_out); c/b/wout are host arrays of device
// pointers.  Synchronises; returns SML_ERR_NUMERIC if any system is singular (sml_last_error names the first).
constexpr int FIT_STREAMS = 8;
int sml_train_fit_batched(int count, double *const *c, const double *const *b, int n, int n_model, int n_out, double beta_res,
                          double beta_model, double prior_val, int using_prior, double *const *wout, void *stream)
{
    SML_REQUIRE(count > 0 && c && b && wout && n > 0 && n_model >= 0 && n_out > 0, "sml_train_fit_batched: bad arguments");
    for (int i = 0; i < count; ++i) SML_REQUIRE(c[i] && b[i] && wout[i], "sml_train_fit_batched: null system %d", i);
    hipStream_t st = sml::as_stream(stream);
    const int n_aug = n + n_model, ncols = n_aug + n_out;
    const int ns = std::min(count, FIT_STREAMS);
    std::vector<hipStream_t> streams(ns, nullptr);
    std::vector<double *> w(ns, nullptr);
    std::vector<int *> ipiv(ns, nullptr);
    int *info = nullptr;
    hipEvent_t fork = nullptr;
    int rc = SML_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < ns; ++i) {
            if (w[i]) (void)hipFree(w[i]);
            if (ipiv[i]) (void)hipFree(ipiv[i]);
            if (streams[i] && ns > 1) (void)hipStreamDestroy(streams[i]);
        }
        if (info) (void)hipFree(info);
        if (fork) (void)hipEventDestroy(fork);
    };
#define FB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return sml::fail(SML_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    FB_HIP(hipMalloc((void **)&info, sizeof(int) * count));
    FB_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    FB_HIP(hipEventRecord(fork, st));
    for (int i = 0; i < ns; ++i) {
        if (ns == 1) streams[i] = st;
        else { FB_HIP(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking)); FB_HIP(hipStreamWaitEvent(streams[i], fork, 0)); }
        FB_HIP(hipMalloc((void **)&w[i], (size_t)n_aug * ncols * sizeof(double)));
        FB_HIP(hipMalloc((void **)&ipiv[i], (size_t)n_aug * sizeof(int)));
    }
    for (int i = 0; i < count && rc == SML_OK; ++i) {
        const int s_ = i % ns;
        rc = fit_enqueue(c[i], b[i], n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout[i], w[s_], ipiv[s_], info + i, streams[s_]);
    }
    std::vector<int> hinfo(count, 0);
    for (int i = 0; i < ns; ++i) {
        hipError_t e = hipStreamSynchronize(streams[i]);
        if (e != hipSuccess && rc == SML_OK) rc = sml::fail(SML_ERR_HIP, "sml_train_fit_batched: %s", hipGetErrorString(e));
    }
    if (rc == SML_OK && hipMemcpy(hinfo.data(), info, sizeof(int) * count, hipMemcpyDeviceToHost) != hipSuccess)
        rc = sml::fail(SML_ERR_HIP, "sml_train_fit_batched: reading info failed");
#undef FB_HIP
    cleanup();
    if (rc) return rc;
    for (int i = 0; i < count; ++i)
        if (hinfo[i]) return sml::fail(SML_ERR_NUMERIC, "sml_train_fit: system %d: U(%d,%d) is exactly zero; the factorisation is singular (dgesv info=%d)", i, hinfo[i], hinfo[i], hinfo[i]);
    return SML_OK;
}

int sml_train_fit(double *c, const double *b, int n, int n_model, int n_out, double beta_res, double beta_model,
                  double prior_val, int using_prior, double *wout, void *stream)
{
    SML_REQUIRE(c && b && wout && n > 0 && n_model >= 0 && n_out > 0, "sml_train_fit: bad arguments");
    double *cc[1] = {c};
    const double *bb[1] = {b};
    double *ww[1] = {wout};
    return sml_train_fit_batched(1, cc, bb, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, ww, stream);
}

}  // extern "C"
