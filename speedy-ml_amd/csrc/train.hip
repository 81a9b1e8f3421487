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
//   k_gemm_nt_dma : the same product with operands row-contiguous in k (the Gram updates, the LU's trailing updates): operand
//                tiles go global -> LDS by LDS-DMA, four ring slots, three K-tiles ahead.
//   k_gemm_nt_big : the long Gram updates (m >= 256): 256x128 tile, one workgroup per CU, all products of an update in one
//                balanced work list (see the kernel: 76 % of the fp64 MFMA spec).
//   LU         : right-looking blocked LU with partial pivoting (dgesv semantics) on the row-major system [A^T+reg | B^T+prior], so
//                that the forward substitution of the right-hand sides rides along.  128-column panels, factored recursively:
//                8-column register-resident leaves (k_lu_leaf: one barrier per pivot) and k_lu_panel_update for the rest of
//                the panel; one composite row permutation per panel (k_lu_perm_src), applied as U12 is formed by k_lu_trsm_mfma<1>;
//                trailing update with k_gemm_nt_dma (alpha = -1), the next panel's strip first (look-ahead on two streams, the
//                trailing stream CU-masked so that the leaf always finds a free CU); blocked back substitution
//                (k_lu_trsm_mfma<0>, k_lu_backsub_near, k_lu_backsub_step).  Up to FIT_BATCH systems advance in lockstep through one chain of
//                launches (grid dimension z).
// All matrices are column-major fp64, as in the reference.
#include <cstdlib>
#include <vector>

#include "bank.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KT = 16, LDS_LD = 144, GT = 256;

// ---- shared pieces of the two GEMM kernels ----
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction, 16 cycles, measured 75 TFLOP/s on this part
// versus 36-48 TFLOP/s for v_mfma_f64_16x16x4_f64 (profiles/micro/mfma_f64_peak.hip).  Measured lane layout
// (profiles/micro/mfma_f64_4x4x4_layout.hip): A operand lane = 16 k + 4 blk + i, B operand lane = 16 k + 4 blk + j,
// D lane = 16 i + 4 blk + j.  C's column index J goes on the MFMA's i and C's row index I on (blk, j), so one instruction
// produces a 16 (I) x 4 (J) tile of C whose 16 consecutive lanes are 16 consecutive rows of one column: contiguous
// 128-byte stores into the column-major C.  A wavefront owns 64 x 64 of C = 4 x 16 such tiles (128 accumulator VGPRs).
struct WavePos { int wi, wj, l15, l4, l3; };

// (UN = J tiles of 4 per wavefront: 16 = the 64 x 64 wavefront tile; 8 = 64 (I) x 32 (J), a 128 x 64 workgroup tile)
template <int UN = 16>
__device__ __forceinline__ WavePos wave_pos()
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return {(wave & 1) * 64, (wave >> 1) * 4 * UN, lane & 15, lane >> 4, lane & 3};
}

template <int KTILE, int UN = 16>
__device__ __forceinline__ void mma_ktile(const double (*As)[LDS_LD], const double (*Bs)[LDS_LD], const WavePos &w, double (&acc)[4][UN])
{
    // All 20 operand reads of a k-step are issued before its 64 MFMAs (profiles/micro/mfma_f64_pattern.hip: this shape
    // sustains 71-74 TFLOP/s with the operands re-read from LDS every k-step; splitting the B fragment in halves to save
    // registers put a wait in front of every 32 MFMAs and ran at 29 TFLOP/s).
#pragma unroll
    for (int kk = 0; kk < KTILE; kk += 4) {
        double af[4], bf[UN];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = As[kk + w.l4][w.wi + 16 * t + w.l15];       // A(I0 + 4 blk + j, k): MFMA "B" operand
#pragma unroll
        for (int u = 0; u < UN; ++u) bf[u] = Bs[kk + w.l4][w.wj + 4 * u + w.l3];        // B(J0 + i, k): MFMA "A" operand
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(bf[u], af[t], acc[t][u], 0, 0, 0);
    }
}

// lane l holds C(I0 + 16 t + (l & 15), J0 + 4 u + (l >> 4)).  C += alpha * acc as a read-modify-write in chunks of 16
// elements per lane: the 16 loads of a chunk are issued together (one exposed round trip per chunk, not per element).
template <int UN = 16>
__device__ __forceinline__ void store_tile(double *__restrict__ C, long ldc, int M, int N, int i0, int j0, const WavePos &w, double alpha,
                                           const double (&acc)[4][UN])
{
#pragma unroll
    for (int u0 = 0; u0 < UN; u0 += 4) {
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
struct GemmBatch { long sa, sb, sc; };          // element strides between the systems of a batch (blockIdx.z)

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(GT, 2) void k_gemm_acc(const double *__restrict__ A, long sai, long sak, const double *__restrict__ B,
                                                     long sbj, long sbk, double *__restrict__ C, long ldc, int M, int N, int K,
                                                     double alpha, int lower_only, GemmBatch gb)
{
    A += gb.sa * blockIdx.z; B += gb.sb * blockIdx.z; C += gb.sc * blockIdx.z;
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

template <int UN>
__global__ __launch_bounds__(GT, 2) void k_gemm_nt_dma(const double *__restrict__ A, long lda, const double *__restrict__ B, long ldb,
                                                        double *__restrict__ C, long ldc, int M, int N, int K, double alpha, int mode, int Mr, int Nr,
                                                        GemmBatch gb)
{
    A += gb.sa * blockIdx.z; B += gb.sb * blockIdx.z; C += gb.sc * blockIdx.z;
    // Mr, Nr: even numbers of READABLE operand rows (>= M, N: a padded operand may be read one row past an odd M or N; those
    // products only reach rows of C that are never stored)
    __shared__ __attribute__((aligned(16))) double S[NBUF][2][DKT][LDS_LD];       // [ring slot][operand][k][row]  73.7 KB
    int ti, tj;
    if (mode == 0) { ti = blockIdx.x; tj = blockIdx.y; }
    else {
        // mode 1: tiles on or below the diagonal of a square product; mode 2: the same for M > N (the Cholesky's trailing update with
        // the right-hand sides riding along): the triangle of the nbj x nbj square, then the full tile rows below it
        const int nbj = (N + BN - 1) / BN, ntri = nbj * (nbj + 1) / 2, L = (int)blockIdx.x;
        if (mode == 1 || L < ntri) tri_tile(L, ti, tj);
        else { ti = nbj + (L - ntri) / nbj; tj = (L - ntri) % nbj; }
    }
    // (UN = 8: 64 rows of B per workgroup; the DMA still moves 128 -- the second half is the next tile's, clamped like any row past N)
    const int i0 = ti * BM, j0 = tj * (8 * UN);
    const WavePos w = wave_pos<UN>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ra = min(i0 + 2 * lane, Mr - 2), rb = min(j0 + 2 * lane, Nr - 2);
    const double *pa = A + ra, *pb = B + rb;
    double acc[4][UN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < UN; ++b) acc[a][b] = 0.0;

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
        mma_ktile<DKT, UN>(S[t % NBUF][0], S[t % NBUF][1], w, acc);
    }
    store_tile<UN>(C, ldc, M, N, i0, j0, w, alpha, acc);
}

// ---- Gram update with 256 x 128 tiles: four wavefronts of 128 (I) x 64 (J) each, ONE workgroup per CU ----
// Why: in k_gemm_nt_dma (128 x 128 tile, four 64 x 64 wavefronts, two workgroups per CU) the eight wavefronts of a CU issue
// 20 operand reads per 64 MFMAs (31 % of the LDS cycles, in bursts right after every barrier) beside 16 B/clk of LDS-DMA writes;
// its own ablation prices that contention at 15 % and DMA + barrier at 14 %.  A 128 x 64 wavefront tile needs 24 reads per 128
// MFMAs (9 % of the LDS cycles with four wavefronts) and the 256 x 128 workgroup tile 6 B/clk of DMA.  With ONE wavefront per
// SIMD nothing covers an instruction that is not an MFMA, so the kernel is one MFMA stream with everything else threaded
// through it (profiles/micro/mfma_f64_issue.hip: one wavefront per SIMD issues independent f64 MFMAs at 93-96 % of the
// two-wavefront rate when nothing else is in the stream):
//   * the fragments of the NEXT k-step are read from LDS while the MFMAs of the current one issue, one two-double read per group
//     of eight MFMAs -- across K-tile boundaries too: the barrier of a K-tile sits at the top of its iteration, where that tile's
//     first fragments are already in registers (read, after the previous barrier, from a slot that had landed by then);
//   * the six LDS-DMA instructions a wavefront owes per K-tile go one each into six of those groups; what they need besides
//     (counted wait, barrier, M0, base advance) sits at one scalar point per K-tile (see "What a non-MFMA instruction costs").
// A workgroup of this kernel owns its CU (512 registers per lane, 108 KB of LDS): anything else in flight -- the skinny products
// of the same update on side streams, as k_gemm_nt_dma had them -- takes whole CUs away from it (measured: 1.77 -> 2.39 ms).  So
// ALL products of one Gram update are tiles of one work list (BigItem), balanced over the chip by the host (gemm_big_plan).
// Measured, one Gram update of the shipped shape (n = 5760, 132 model rows, 136 targets, m = 2920; profiles/gram_only.py):
// 1.81 ms = 59.9 TF/s of executed flops (lower-triangle 128-blocks + skinny blocks) = 76 % of the 78.6 TF/s fp64 MFMA spec, against
// 2.50 ms = 43.3 TF/s for k_gemm_nt_dma + side streams; 16.6 cycles per MFMA inside the K loop (16.0 = the instruction's length;
// SML_GEMM_STAMPS=1), matrix pipe busy 88 % of the launch (SQ_VALU_MFMA_BUSY_CYCLES, profiles/r2_gram_pmc.json; the rest is the
// K loop's 4 %, prologue / epilogue of 624 tiles and the last round).  History of this kernel, same update: DMA issue and LDS
// reads as blocks in front of the MFMAs 2.7 ms; reads and DMAs threaded through the MFMAs 2.22; all products in one list 1.98; no
// vector ALU work in the loop and one scalar point per K-tile 1.84; the six DMAs in separate groups 1.80; tail dealt out by K-tiles
// (stream-K) 1.79.  Ablations (SML_GEMM_ABL, profiles/micro/gemm_phase_ablation.py; SYRK + 8 target rows): full 1.82 ms, no DMA
// 1.74, no LDS reads 1.82, no barrier 1.80, none of the three 1.75.  An XCD-aware deal of the list (runs of 32 tiles sharing operand rows per L2) is
// worth < 1 %; a look-ahead of 2 K-tiles instead of 3 costs 230 cycles of exposed wait per K-tile.
// Accumulators: 128 doubles per lane (the 256 AGPRs); two fragment sets of 8 + 16 doubles.
constexpr int GB_I = 256, GB_J = 128, GB_T = 256, GB_LDA = 272, GB_LDB = 160, GB_NBUF = 4, GB_AHEAD = 3;
constexpr int GB_ROW = GB_LDA + GB_LDB;                          // doubles per k: the A rows then the B rows
constexpr int GB_SLOT = DKT * GB_ROW;                            // doubles per ring slot

struct BigFrag { double a[8], b[16]; };

// (the 12 two-double reads of a fragment set: units 0-3 = a[2j], a[2j+1]; units 4-11 = b[2j'], b[2j'+1] -- gb_read_unit)
__device__ __forceinline__ void gb_mma_u(const BigFrag &f, int u, double (&acc)[8][16])
{
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(f.b[u], f.a[t], acc[t][u], 0, 0, 0);
}

// The operands of one Gram update: sources 0 = states, 1 = imperfect-model states, 2 = targets (row-contiguous: X(i,k) = X[i + k*ld]),
// outputs 0 = C (n_aug x n_aug), 1 = B (n_out x n_aug), both column-major.
struct BigOperands { const double *src[3]; long ld[3]; double *out[2]; long ldo[2]; unsigned long long *stamps; };

// One work item: D = A_tile * B_tile^T over the K range [k_begin, k_end) (multiples of DKT), A_tile = rows a_row0.. of source a_src
// (a_rows of them readable: rows past that are clamped to valid memory, their products only reach elements that are never stored),
// likewise B.  Rows < si and columns < sj of D are added to out[c_out] at element offset c_off (dst = 0), or the whole tile is
// stored into partial[dst - 1] (a K-split piece; k_gemm_big_reduce adds the pieces of a tile into the output in a fixed order).
struct BigItem { int a_src, a_row0, a_rows, b_src, b_row0, b_rows, c_out, si, sj, k_begin, k_end, dst, np, pad; long c_off; };

// What a non-MFMA instruction costs a wavefront that owns its SIMD (profiles/micro/mfma_f64_issue.hip, cycles lost per occurrence
// in a stream of independent f64 MFMAs, 16 cycles each): ds_read2_b64 2.8, global_load 8, ONE scalar instruction 16 (eight in a row
// 36), ONE vector-ALU instruction 28 (four in a row 40), a not-taken branch 26.  So the K loop holds no vector ALU work at all
// (every LDS read address and DMA lane offset is a register computed before the loop; the ring is unrolled so the slot is a
// compile-time choice of register), and its scalar work -- the counted wait, the barrier, the M0 write, the advance of the two
// operand bases, the loop test -- sits at ONE point per K-tile.  The six DMA instructions share that one M0: the instruction
// offset of global_load_lds applies to the global address AND the LDS address (profiles/micro/lds_dma_offset.hip), so the lane
// offsets carry its opposite; each sits alone in an MFMA group, because a DMA whose lines miss the vector L1 holds the issue port
// for ~36 cycles and six in a row cost 240 cycles per K-tile (measured) against ~60 spread out.  First version of this loop
// (address arithmetic and an M0 write per DMA, a branch around each): 18.6 cycles per MFMA, 1.6 of them the DMA issue.
typedef __attribute__((address_space(3))) const double *lds_cptr_t;
constexpr int GB_MID = 2816;        // M0 points this many bytes into the 6 DMA targets of a wavefront; instruction offsets are +-GB_MID

__device__ __forceinline__ void gb_read_unit(unsigned ra, unsigned rb, int j, BigFrag &f)
{
    lds_cptr_t ar = (lds_cptr_t)(uintptr_t)ra, br = (lds_cptr_t)(uintptr_t)rb;
    if (j < 4) { f.a[2 * j] = ar[32 * j]; f.a[2 * j + 1] = ar[32 * j + 16]; }
    else { const int u = 2 * (j - 4); f.b[u] = br[4 * u]; f.b[u + 1] = br[4 * u + 4]; }
}

// the six DMAs of one K-tile at once (the prologue only: in the loop they go one per MFMA group).  M0 is a register the compiler
// reserves and sets itself before each of its own uses (none in this kernel's loop), so writing it here needs no clobber.
#define GB_DMA6                                                                                                              \
    "s_mov_b32 m0, %[m0]\n s_nop 0\n"                                                                                         \
    "global_load_lds_dwordx4 %[v0], %[sa] offset:%[i0]\n global_load_lds_dwordx4 %[v1], %[sa] offset:%[i1]\n"               \
    "global_load_lds_dwordx4 %[v2], %[sb] offset:%[i2]\n global_load_lds_dwordx4 %[v3], %[sa] offset:%[i3]\n"               \
    "global_load_lds_dwordx4 %[v4], %[sa] offset:%[i4]\n global_load_lds_dwordx4 %[v5], %[sb] offset:%[i5]\n"
#define GB_DMA6_OPS(M0V)                                                                                                     \
    [m0] "s"(M0V), [v0] "v"(vq[0]), [v1] "v"(vq[1]), [v2] "v"(vq[2]), [v3] "v"(vq[3]), [v4] "v"(vq[4]), [v5] "v"(vq[5]), [sa] "s"(sa),  \
        [sb] "s"(sb), [i0] "n"(0 - GB_MID), [i1] "n"(1024 - GB_MID), [i2] "n"(GB_LDA * 8 - GB_MID), [i3] "n"(GB_ROW * 8 - GB_MID),       \
        [i4] "n"(GB_ROW * 8 + 1024 - GB_MID), [i5] "n"(GB_ROW * 8 + GB_LDA * 8 - GB_MID)

template <int ABL>
__global__ __launch_bounds__(GB_T, 1) void k_gemm_nt_big(BigOperands ops, double alpha, const BigItem *__restrict__ items,
                                                          const int *__restrict__ wg_first, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double gb_lds[];       // [GB_NBUF][DKT][GB_ROW]; the only LDS of the kernel: byte address 0
    // a workgroup of the bulk owns one item; a workgroup of the tail owns a run of K-tiles that may end one tile and begin the next
    for (int item = wg_first[blockIdx.x], item_end = wg_first[blockIdx.x + 1]; item < item_end; ++item) {
    if (item != wg_first[blockIdx.x]) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");   // (the ring is reused)
    const BigItem it = items[item];
    const unsigned long long stamp_c0 = (ABL & 8) ? __builtin_readcyclecounter() : 0, stamp_r0 = (ABL & 8) ? __builtin_amdgcn_s_memrealtime() : 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = (wave & 1) * 128, wj = (wave >> 1) * 64, l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3;
    const int ntiles = (it.k_end - it.k_begin) / DKT;
    // DMA: wavefront w moves k-columns 2w and 2w+1 of a K-tile: A rows 0-127, A rows 128-255, B rows 0-127 of each (6 instructions).
    // Global address = scalar base (advanced by one K-tile per iteration, stopping at the last: the tail of the loop re-fetches that
    // tile into slots nobody reads) + lane offset + instruction offset; the bases sit 4096 bytes low so no lane offset is negative.
    const long lda = ops.ld[it.a_src], ldb = ops.ld[it.b_src];
    const char *sa = (const char *)(ops.src[it.a_src] + it.a_row0 + (long)it.k_begin * lda) - 4096;
    const char *sb = (const char *)(ops.src[it.b_src] + it.b_row0 + (long)it.k_begin * ldb) - 4096;
    unsigned vq[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int k = 2 * wave + q / 3, part = q % 3;
        const int row = part == 0 ? min(2 * lane, it.a_rows - 2) : part == 1 ? min(128 + 2 * lane, it.a_rows - 2) : min(2 * lane, it.b_rows - 2);
        const int imm = (q / 3) * GB_ROW * 8 + (part == 0 ? 0 : part == 1 ? 1024 : GB_LDA * 8) - GB_MID;
        vq[q] = (unsigned)(((long)k * (part < 2 ? lda : ldb) + row) * 8 + 4096 - imm);
        asm volatile("" : "+v"(vq[q]));
    }
    const long step_a = (long)DKT * lda * 8, step_b = (long)DKT * ldb * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)gb_lds;          // LDS byte address of the ring (0: the kernel's only LDS)
    unsigned m0v[GB_NBUF];
#pragma unroll
    for (int sl = 0; sl < GB_NBUF; ++sl) m0v[sl] = lds0 + (unsigned)((sl * GB_SLOT + 2 * wave * GB_ROW) * 8 + GB_MID);
    // LDS read addresses (bytes) of the two k-steps of every ring slot
    unsigned ra[GB_NBUF][2], rb[GB_NBUF][2];
#pragma unroll
    for (int sl = 0; sl < GB_NBUF; ++sl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ra[sl][h] = lds0 + (unsigned)((sl * GB_SLOT + (4 * h + l4) * GB_ROW + wi + l15) * 8);
            rb[sl][h] = lds0 + (unsigned)((sl * GB_SLOT + (4 * h + l4) * GB_ROW + GB_LDA + wj + l3) * 8);
            asm volatile("" : "+v"(ra[sl][h]), "+v"(rb[sl][h]));
        }
    double acc[8][16];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[a][b] = 0.0;
    // prologue: tiles 0, 1, 2 on their way (a K range shorter than that repeats its last tile)
    int ahead = 0;                                                         // K-tiles the bases are ahead of k_begin
#pragma unroll
    for (int T = 0; T < GB_AHEAD; ++T) {
        if (!(ABL & 1)) asm volatile(GB_DMA6 ::GB_DMA6_OPS(m0v[T]) : "memory");
        if (ahead + 1 < ntiles) { sa += step_a; sb += step_b; ++ahead; }
    }
    asm volatile("s_waitcnt vmcnt(12)\n s_barrier" ::: "memory");       // tile 0 has landed, for every wavefront
    BigFrag f0, f1;
#pragma unroll
    for (int j = 0; j < 12; ++j) gb_read_unit(ra[0][0], rb[0][0], j, f0);
    unsigned long long wait_top = 0;
    // one K-tile; SL = its ring slot.  Returns false after the last.
    auto ktile = [&](auto slot_tag, int t) {
        constexpr int SL = decltype(slot_tag)::value, NX = (SL + 1) % GB_NBUF, FILL = (SL + GB_AHEAD) % GB_NBUF;
        const unsigned long long w0 = (ABL & 8) ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0), where the compiler's own bookkeeping sees it
        // tile t+1 has landed (tile t+2 may stay in flight); the reads of f0 are done; every wavefront is done with slot (t-1) % 4,
        // which the DMAs of tile t+3 refill
        if (ABL & 4) asm volatile("s_waitcnt vmcnt(6)\n s_mov_b32 m0, %0" ::"s"(m0v[FILL]) : "memory");
        else if (ABL & 1) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)\n s_barrier\n s_mov_b32 m0, %0" ::"s"(m0v[FILL]) : "memory");
        if (ABL & 8) wait_top += __builtin_readcyclecounter() - w0;
        __builtin_amdgcn_sched_barrier(0);
        // k-step 0 of tile t, the reads of k-step 1 under it
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < 12 && (!(ABL & 2) || t == 0)) gb_read_unit(ra[SL][1], rb[SL][1], u, f1);
            if (!(ABL & 1) && (u & 1) && u < 12) {                        // the six DMAs of tile t+3, one per odd group (M0 from the scalar point)
                constexpr int imm[6] = {0 - GB_MID, 1024 - GB_MID, GB_LDA * 8 - GB_MID, GB_ROW * 8 - GB_MID, GB_ROW * 8 + 1024 - GB_MID,
                                        GB_ROW * 8 + GB_LDA * 8 - GB_MID};
                const int q = u >> 1;
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(vq[q]), "s"(q % 3 == 2 ? sb : sa), "n"(imm[q]) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            gb_mma_u(f0, u, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): f1 (read >= 32 MFMAs ago)
        // k-step 1 of tile t, the reads of k-step 0 of tile t+1 under it (after the last tile: dead reads)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < 12 && !(ABL & 2)) gb_read_unit(ra[NX][0], rb[NX][0], u, f0);
            __builtin_amdgcn_sched_barrier(0);
            gb_mma_u(f1, u, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ahead + 1 < ntiles) { sa += step_a; sb += step_b; ++ahead; }     // (scalar: lands at the next tile's scalar point)
    };
    for (int t = 0; t < ntiles; t += GB_NBUF) {
        ktile(std::integral_constant<int, 0>{}, t);
        if (t + 1 >= ntiles) break;
        ktile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 >= ntiles) break;
        ktile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 >= ntiles) break;
        ktile(std::integral_constant<int, 3>{}, t + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the re-fetched last tiles)
    if ((ABL & 8) && ops.stamps && threadIdx.x == 0) {          // (SML_GEMM_STAMPS: shader cycles and 100 MHz ticks of the K loop, per workgroup)
        ops.stamps[4 * blockIdx.x] = __builtin_readcyclecounter() - stamp_c0;
        ops.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        ops.stamps[4 * blockIdx.x + 2] = wait_top;
        ops.stamps[4 * blockIdx.x + 3] = 0;
    }
    // lane l holds D(wi + 16 t + (l & 15), wj + 4 u + (l >> 4)): 16 consecutive lanes = 128 contiguous bytes of a column
    if (it.dst) {
        double *dst = partial + (size_t)(it.dst - 1) * GB_I * GB_J;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 8; ++t) dst[(size_t)(wj + 4 * u + l4) * GB_I + wi + 16 * t + l15] = acc[t][u];
        continue;
    }
    // Every element of the output belongs to exactly one item of a launch, so a fire-and-forget atomic add gives the same sum as
    // load + add + store, without a round trip to memory per column.
    double *C = ops.out[it.c_out] + it.c_off;
    const long ldc = ops.ldo[it.c_out];
    const int room = it.si - (wi + l15);                                   // rows of this lane's column segments that exist: 16 t < room
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = wj + 4 * u + l4;
        double *col = C + (long)j * ldc + wi + l15;
        if (j < it.sj) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (16 * t < room) unsafeAtomicAdd(col + 16 * t, alpha * acc[t][u]);
        }
    }
    }
}

// out tile += alpha * (piece_0 + piece_1 + ...) in that order: the K-split pieces of the tail tiles
__global__ __launch_bounds__(256) void k_gemm_big_reduce(BigOperands ops, double alpha, const BigItem *__restrict__ tails,
                                                          const double *__restrict__ partial)
{
    const BigItem it = tails[blockIdx.y];                                  // .dst = first piece's slot (1-based), .np = number of pieces
    const int pieces = it.np;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int il = e % GB_I, jl = e / GB_I;
    if (il >= it.si || jl >= it.sj) return;
    double s = 0.0;
    for (int p = 0; p < pieces; ++p) s = s + partial[(size_t)(it.dst - 1 + p) * GB_I * GB_J + e];
    double *c = ops.out[it.c_out] + it.c_off + (long)jl * ops.ldo[it.c_out] + il;
    *c = *c + alpha * s;
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

// ---------------------------------------------------------------------------------------------------------------------------
// LU with partial pivoting (dgesv semantics: first maximum of |a| per column, row interchange, scale by the reciprocal pivot)
// on the row-major system  W[n][ld] = [ (C + reg)^T | (B + prior)^T ]  (ld >= n + n_out).
//
// Layout.  A row interchange of a row-major matrix is a contiguous copy, and U12 / the trailing matrix stream along rows; the
// panel being factorised, however, is searched and eliminated column by column, so the current outer panel (LU_NBO = 128 columns,
// rows K0..n) lives in a column-major panel buffer P (two of them: the trailing update of panel k still reads its L21 from one
// while panel k+1 is factorised in the other).  L is never written back: the right-hand sides ride along as extra columns of W, so
// after the factorisation only U and the transformed right-hand sides are needed (back substitution).
//
// Schedule per outer panel (two streams; `P` = panel chain, `G` = everything else):
//   P: 16 x [ k_lu_leaf   : ONE workgroup factorises 8 columns held entirely in registers (<= 7 rows x 8 columns per thread):
//                           one global read and one global write of the leaf, one barrier per pivot.  The same workgroup
//                           applies each interchange to the panel's other columns and finishes U (8 x 120) of the leaf's rows.
//             k_lu_panel_update : rank-8 update of the panel's remaining columns, all CUs ]
//   G: k_lu_perm_src (the composite permutation of the panel's 128 interchanges on the <= 256 rows they touch: no dependent chain of
//      128 swaps, and no separate pass over W -- the triangular solve reads its rows through it),
//      k_lu_trsm_mfma<1> (U12 = L11^-1 A12 + the displaced rows written back), then the trailing update with the MFMA GEMM at
//      K = 128: FIRST the next panel's 128 columns (+ k_lu_strip_to_panel, which hands them to P), THEN the rest, which runs
//      beside the next panel's leaf chain (look-ahead).
// Back substitution, per 128-row block from the bottom: k_lu_backsub_near (the block above gets the solved block's update), then
// k_lu_backsub_step (that block solved beside the update of all rows above it).
// The previous form (32-wide panel in one workgroup that walked the panel through memory for every sub-panel, one dependent
// chain of 32 row swaps per column, trailing updates at K = 32) spent 64 % of an 84 ms factorisation in the panel kernel.
constexpr int LU_NBO = 128;    // outer panel width = K of the trailing update
constexpr int LU_LEAF = 8;     // columns factorised in registers by one workgroup

// W[i][j], i < n_aug: j < n_aug: a_trans(i,j) = C(j,i) (+ reg on the diagonal); n_aug <= j < ncols: b_trans(i,o) = B(o,i) + prior(o,i)
// (fit_chunk_hybrid :1261-1309).  C is symmetric at this point, so C(j,i) = c[j + i*n_aug] is read contiguously along j.
// The LU kernels factorise a BATCH of equally sized systems in lockstep: one grid dimension indexes the system, every scratch array
// is a stack of per-system slabs (LuStride).  One chain of launches then serves all of them; eight separate chains on eight
// stream pairs were multiplexed onto the runtime's four hardware queues and ran no faster than one after another (measured).
struct LuStride { long w, p; int ipiv; };

// upper_only: W's strict lower triangle is left alone (the Cholesky reads j >= i only: C(j,i) with j >= i is the LOWER triangle of the
// column-major C, the part sml_train_accumulate always fills -- no symmetrisation pass)
__global__ void k_build_system(const double *const *__restrict__ c_list, const double *const *__restrict__ b_list, double *__restrict__ w, long ld, int n_aug,
                               int n_model, int n_out, double reg_model, double reg_res, double prior_diag, LuStride ls, int upper_only)
{
    const double *__restrict__ c = c_list[blockIdx.z], *__restrict__ b = b_list[blockIdx.z];
    w += ls.w * blockIdx.z;
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld || (upper_only && j < i)) return;
    double v = 0.0;
    if (j < n_aug) {
        v = c[(long)j + (long)i * n_aug];
        if (i == j) v = v + (i < n_model ? reg_model : reg_res);
    } else if (j < n_aug + n_out) {
        const int o = j - n_aug;
        v = b[(long)o + (long)i * n_out];
        if (o == i && o < n_model) v = v + prior_diag;
    }
    w[(long)i * ld + j] = v;
}

// wout(o,i) = Z(i,o) = W[i][n_aug + o]; wout is the column-major (n_out, n_aug) array of the reference
__global__ void k_extract_wout(const double *__restrict__ w, long ld, double *const *__restrict__ wout_list, int n_aug, int n_out, LuStride ls)
{
    double *__restrict__ wout = wout_list[blockIdx.y];
    w += ls.w * blockIdx.y;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_aug * n_out) return;
    const int o = (int)(t % n_out), i = (int)(t / n_out);
    wout[t] = w[(long)i * ld + n_aug + o];
}

// P[q][r] = W[r][K0 + q] for K0 <= r < n, q < nbp: the next panel's columns, transposed through LDS (both sides contiguous)
__global__ __launch_bounds__(256) void k_lu_strip_to_panel(const double *__restrict__ w, long ld, double *__restrict__ P, long np, int n, int K0, int nbp,
                                                            LuStride ls)
{
    w += ls.w * blockIdx.z; P += ls.p * blockIdx.z;
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = K0 + blockIdx.x * 32, q0 = blockIdx.y * 32;
    for (int y = ty; y < 32; y += 8) {
        const int r = r0 + y, q = q0 + tx;
        t[y][tx] = (r < n && q < nbp) ? w[(long)r * ld + K0 + q] : 0.0;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int q = q0 + y, r = r0 + tx;
        if (r < n && q < nbp) P[(long)q * np + r] = t[tx][y];
    }
}

// ---- leaf: columns [c, c+lw) of the panel, rows [c, n), in registers; one workgroup of LEAF_T threads ----
// Thread t owns rows c + t + LEAF_T i (i < R).  A pivot costs ONE barrier: before it every wavefront reduces its first-maximum
// (DPP butterflies inside the 16-lane rows, v_readlane across them; the value first, then the smallest row among the lanes that
// hold it) and parks value, row and -- from the one lane that owns it -- the candidate row's LW entries in LDS; the owner of the
// diagonal row parks that row too.  After the barrier every thread finishes the reduction over the wavefront results itself and
// reads the winning candidate row: no second barrier for the interchange.  The elimination is in the column-by-column order of
// dgetf2 with fused multiply-adds.  The workgroup is VALU-issue bound (every wavefront repeats the ~290 instructions of a pivot's
// reduction and bookkeeping), so fewer wavefronts with more rows each are faster -- as long as a thread stays within 256
// registers (beyond v255 the vector ALU pays a copy each way).  Hence 4 wavefronts x <= 14 rows up to 3584 rows and 8 wavefronts
// x <= 12 rows above (measured per leaf, in-kernel stamps: 256 threads 7.4 us at 256 rows .. 15.9 us at 3584 rows against
// 9.7 .. 19.5 us with 512 threads; 3584 < rows <= 6144: 21 .. 27 us with 512 threads against 18 .. 47 us with 256; the first
// version with 16 wavefronts x 7 rows took 3.3 us per pivot).
// Threads LEAF_T-128 .. LEAF_T-1 also take one of the panel's OTHER columns each and apply all lw interchanges to it in one batch:
// the leaf's rows of that column are fetched at the start, each pivot row as soon as it is known (consumed only at the end), the
// composite permutation is applied through LDS and, to the right of the leaf, the leaf's rows become U (u = L11^-1 a).
__device__ __forceinline__ double dpp_max_row16(double v)
{   // all-reduce max inside each row of 16 lanes: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror
    union { double d; int i[2]; } a, b;
    a.d = v;
#define DPP_STEP(CTRL)                                                   \
    b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], CTRL, 0xf, 0xf, true);     \
    b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], CTRL, 0xf, 0xf, true);     \
    a.d = fmax(a.d, b.d);
    DPP_STEP(0xB1) DPP_STEP(0x4E) DPP_STEP(0x141) DPP_STEP(0x140)
#undef DPP_STEP
    return a.d;
}

__device__ __forceinline__ int dpp_min_row16(int v)
{
#define DPP_STEP(CTRL) v = min(v, __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true));
    DPP_STEP(0xB1) DPP_STEP(0x4E) DPP_STEP(0x141) DPP_STEP(0x140)
#undef DPP_STEP
    return v;
}

__device__ __forceinline__ double wave_max_f64(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = dpp_max_row16(v);
    double m = -1.0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        r.i[0] = __builtin_amdgcn_readlane(a.i[0], 16 * row);
        r.i[1] = __builtin_amdgcn_readlane(a.i[1], 16 * row);
        m = fmax(m, r.d);
    }
    return m;
}

__device__ __forceinline__ int wave_min_i32(int v)
{
    v = dpp_min_row16(v);
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) m = min(m, __builtin_amdgcn_readlane(v, 16 * row));
    return m;
}

template <int LEAF_T, int R, int LW, bool FULL>
__global__ __launch_bounds__(LEAF_T) void k_lu_leaf(double *__restrict__ P, long np, int n, int K0, int c, int lw_rt, int nbp, int *__restrict__ ipiv,
                                                     int *__restrict__ info, long long *__restrict__ stamps, LuStride ls)
{
    P += ls.p * blockIdx.x; ipiv += ls.ipiv * blockIdx.x; info += blockIdx.x;
    if (blockIdx.x) stamps = nullptr;
    // stamps (diagnostic builds of a run only: SML_LU_STAMP=1): thread 0's cycle counter after the loads, after every pivot, after
    // the stores were issued and at the end; nullptr otherwise
#define LEAF_STAMP(K) if (stamps && threadIdx.x == 0) { stamps[K] = __builtin_readcyclecounter(); stamps[16 + (K)] = __builtin_amdgcn_s_memrealtime(); }
    LEAF_STAMP(0)
    constexpr int NW = LEAF_T / 64, LEAF_SH = LEAF_T == 512 ? 9 : 8;
    static_assert(LEAF_T == 1 << LEAF_SH, "the leaf has 256 or 512 threads");
    __shared__ __attribute__((aligned(16))) double sval[2][NW];
    __shared__ __attribute__((aligned(16))) int sidx[2][NW];
    __shared__ __attribute__((aligned(16))) double cand[2][NW][LU_LEAF];     // every wavefront's candidate pivot row
    __shared__ __attribute__((aligned(16))) double rowC[2][LU_LEAF];         // the diagonal row
    __shared__ double vt[2 * LU_LEAF][LU_NBO];                               // the affected rows of the panel's other columns
    __shared__ double l11[LU_LEAF][LU_LEAF];
    __shared__ int piv[LU_LEAF], srcidx[2 * LU_LEAF];
    const int lw = FULL ? LW : lw_rt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = c - K0;
    // slot i of this thread is row c + tid + LEAF_T i; slots beyond row n hold zeros throughout (0 * l = 0) and lose every tie
    // in the pivot search (R is the next instantiated size >= ceil((n - c) / LEAF_T))
    const int rows = n - c;
    double *const base = P + (long)q0 * np + c + tid;
    double a[R][LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const double *cp = base + (long)j * np;
        const bool cok = FULL || j < lw;
#pragma unroll
        for (int i = 0; i < R; ++i) a[i][j] = (cok && tid + i * LEAF_T < rows) ? cp[i * LEAF_T] : 0.0;
    }
    const int q = tid - (LEAF_T - LU_NBO);
    const bool other = q >= 0 && q < nbp && (q < q0 || q >= q0 + lw);
    double *colq = P + (long)(other ? q : 0) * np;
    double prow[LW];                                          // this column's entry in each pivot row
    if (other) {
#pragma unroll
        for (int k = 0; k < LW; ++k)
            if (FULL || k < lw) vt[k][q] = colq[c + k];
    }
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        prow[j] = 0.0;
        if (FULL || j < lw) {                                 // uniform
            const int col = c + j, par = j & 1;
            // first maximum of |a|: the value ...
            double best = tid >= j ? fabs(a[0][j]) : -1.0;    // slot 0 is row c + tid: a candidate from the diagonal on
#pragma unroll
            for (int i = 1; i < R; ++i) best = fmax(best, fabs(a[i][j]));
            const double mw = wave_max_f64(best);
            if (j == 0) { LEAF_STAMP(1) }
            // ... then the smallest row among the lanes that hold it
            int rowc = 0x7fffffff;
            if (best == mw) {
#pragma unroll
                for (int i = R - 1; i >= 1; --i)
                    if (fabs(a[i][j]) == mw) rowc = c + tid + i * LEAF_T;
                if (tid >= j && fabs(a[0][j]) == mw) rowc = c + tid;
                if (rowc >= n) rowc = 0x7fffffff;
            }
            const int iw = wave_min_i32(rowc);
            if (lane == 0) { sval[par][wave] = mw; sidx[par][wave] = iw; }
            if (iw != 0x7fffffff && tid == ((iw - c) & (LEAF_T - 1))) {     // the candidate row's owner parks it
                const int is = (iw - c) >> LEAF_SH;          // uniform; a chain of scalar compares (an indexed register array would
                                                              // go to scratch, a switch made the compiler copy the whole array)
#pragma unroll
                for (int i = 0; i < R; ++i)
                    if (i == is) {
#pragma unroll
                        for (int jj = 0; jj < LW; ++jj) cand[par][wave][jj] = a[i][jj];
                    }
            }
            if (tid == j) {                                   // the diagonal row c + j is thread j's slot 0
#pragma unroll
                for (int jj = 0; jj < LW; ++jj) rowC[par][jj] = a[0][jj];
            }
            __syncthreads();
            double sv[NW];
            int si[NW];
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { sv[w2] = sval[par][w2]; si[w2] = sidx[par][w2]; }
            double b2 = sv[0];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) b2 = fmax(b2, sv[w2]);
            int p = 0x7fffffff;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) p = min(p, sv[w2] == b2 ? si[w2] : 0x7fffffff);
            if (p == 0x7fffffff) p = col;                     // (no finite candidate: NaN column)
            const int wsel = ((p - c) & (LEAF_T - 1)) >> 6;
            if (tid == 0) {
                ipiv[col] = p;
                piv[j] = p;
                if (b2 == 0.0 && *info == 0) *info = col + 1;
            }
            if (other && p >= c + lw) prow[j] = colq[p];      // consumed after the pivot loop
            double rp[LW];
#pragma unroll
            for (int jj = 0; jj < LW; ++jj) rp[jj] = cand[par][wsel][jj];
            if (p != col) {
                const int tp = (p - c) & (LEAF_T - 1), ip = (p - c) >> LEAF_SH;      // uniform
                if (tid == tp) {                              // (before slot 0 is overwritten when tp == j)
#pragma unroll
                    for (int i = 0; i < R; ++i)
                        if (i == ip) {
#pragma unroll
                            for (int jj = 0; jj < LW; ++jj) a[i][jj] = rowC[par][jj];
                        }
                }
                if (tid == j) {
#pragma unroll
                    for (int jj = 0; jj < LW; ++jj) a[0][jj] = rp[jj];
                }
            }
            const double pv = rp[j];
            if (pv != 0.0) {
                const double inv = 1.0 / pv;
                {
                    const bool el = tid > j;
                    const double l = el ? a[0][j] * inv : 0.0;
                    a[0][j] = el ? l : a[0][j];
#pragma unroll
                    for (int jj = 0; jj < LW; ++jj)
                        if (jj > j) a[0][jj] = __builtin_fma(-l, rp[jj], a[0][jj]);
                }
#pragma unroll
                for (int i = 1; i < R; ++i) {
                    const double l = a[i][j] * inv;
                    a[i][j] = l;
#pragma unroll
                    for (int jj = 0; jj < LW; ++jj)
                        if (jj > j) a[i][jj] = __builtin_fma(-l, rp[jj], a[i][jj]);
                }
            }
            // no barrier here: the next pivot uses the other halves of sval / sidx / cand / rowC
            LEAF_STAMP(2 + j)
        }
    }
    // ---- the panel's other columns: all lw interchanges in one batch ----
    // positions: idx < 8 -> row c + idx, idx >= 8 -> row piv[idx - 8]; after the interchanges position x holds what was at
    // s_0(s_1(...s_{lw-1}(x))), s_t = transposition (c + t, piv[t]); srcidx = index of that row in the position list
    if (other) {
#pragma unroll
        for (int k = 0; k < LW; ++k)
            if (FULL || k < lw) vt[LU_LEAF + k][q] = prow[k];  // (pivot rows inside the leaf's own rows are read through their idx < 8 slot)
    }
    __syncthreads();                                          // piv[] complete
    if (tid < 2 * LU_LEAF) {
        const int t2 = tid & (LU_LEAF - 1);
        int x = t2 < lw ? (tid < LU_LEAF ? c + t2 : piv[t2]) : -1;
        for (int t = lw - 1; t >= 0; --t) {
            const int d = c + t, pp = piv[t];
            x = (x == d) ? pp : (x == pp ? d : x);
        }
        int m = 0;                                            // x is c + m (m < lw) or one of the pivot rows
        if (x >= c && x < c + lw) m = x - c;
        else
            for (int t = 0; t < lw; ++t)
                if (piv[t] == x) { m = LU_LEAF + t; break; }
        srcidx[tid] = m;
    }
    double *sbase = base;
    asm volatile("" : "+v"(sbase));                           // (a fresh address chain: keeps the load-time addresses from living through the loop)
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        double *cp = sbase + (long)j * np;
        if (FULL || j < lw) {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (tid + i * LEAF_T < rows) cp[i * LEAF_T] = a[i][j];
        }
    }
    LEAF_STAMP(10)
    if (tid < LW) {
#pragma unroll
        for (int jj = 0; jj < LW; ++jj) l11[tid][jj] = a[0][jj];       // rows c .. c+7 belong to threads 0..7 (slot 0)
    }
    __syncthreads();
    if (other) {
        const bool right = q >= q0 + lw;
        double u[LW];
#pragma unroll
        for (int r = 0; r < LW; ++r) {
            if (FULL || r < lw) {
                double v = vt[srcidx[r]][q];
                if (right) {
#pragma unroll
                    for (int k = 0; k < LW; ++k)
                        if (k < r) v = __builtin_fma(-l11[r][k], u[k], v);
                }
                u[r] = v;
                colq[c + r] = v;
            }
        }
#pragma unroll
        for (int t = 0; t < LW; ++t) {
            if (FULL || t < lw) {
                const int pr = piv[t];
                if (pr >= c + lw) colq[pr] = vt[srcidx[LU_LEAF + t]][q];
            }
        }
    }
    LEAF_STAMP(11)
}
#undef LEAF_STAMP

// The leaf for panels taller than the register-resident form holds (more than 14 x 512 rows): the same factorisation -- pivot rule,
// multipliers, fused updates, interchanges and the U rows of the panel's other columns, in the same operation order -- on the panel in
// memory, one 512-thread workgroup striding over the rows.  Slow (every pivot is a pass over the column through L2) and rare: systems
// this size are not positive definite ones of the shipped region layouts; it exists so that the LU has no size limit.
__global__ __launch_bounds__(512) void k_lu_leaf_tall(double *__restrict__ P, long np, int n, int K0, int c, int lw, int nbp, int *__restrict__ ipiv,
                                                      int *__restrict__ info, LuStride ls)
{
    P += ls.p * blockIdx.x; ipiv += ls.ipiv * blockIdx.x; info += blockIdx.x;
    constexpr int T = 512, NWV = T / 64;
    __shared__ double sval[NWV];
    __shared__ int sidx[NWV];
    __shared__ int s_piv;
    __shared__ double rowp[LU_LEAF];                         // the pivot row's entries in the leaf's columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = c - K0;
    for (int j = 0; j < lw; ++j) {
        const int col = c + j;
        double *cj = P + (long)(q0 + j) * np;
        // first maximum of |a| from the diagonal on, lowest row among ties
        double best = -1.0;
        int brow = 0x7fffffff;
        for (int r = col + tid; r < n; r += T) {
            const double v = fabs(cj[r]);
            if (v > best) { best = v; brow = r; }            // (rows ascend per thread: the first maximum is the lowest row)
        }
        const double mw = wave_max_f64(best);
        const int iw = wave_min_i32(best == mw ? brow : 0x7fffffff);
        if (lane == 0) { sval[wave] = mw; sidx[wave] = iw; }
        __syncthreads();
        if (tid == 0) {
            double b2 = sval[0];
            for (int w2 = 1; w2 < NWV; ++w2) b2 = fmax(b2, sval[w2]);
            int p = 0x7fffffff;
            for (int w2 = 0; w2 < NWV; ++w2) p = min(p, sval[w2] == b2 ? sidx[w2] : 0x7fffffff);
            if (p == 0x7fffffff) p = col;                     // (no finite candidate: NaN column)
            ipiv[col] = p;
            s_piv = p;
            if (b2 == 0.0 && *info == 0) *info = col + 1;
        }
        __syncthreads();
        const int p = s_piv;
        if (p != col && tid < nbp) {                          // the interchange, in every column of the panel
            double *cq = P + (long)tid * np;
            const double a = cq[col], b = cq[p];
            cq[col] = b; cq[p] = a;
        }
        __syncthreads();
        if (tid < lw) rowp[tid] = P[(long)(q0 + tid) * np + col];
        __syncthreads();
        const double pv = rowp[j];
        if (pv != 0.0) {
            const double inv = 1.0 / pv;
            for (int r = col + 1 + tid; r < n; r += T) {
                const double l = cj[r] * inv;
                cj[r] = l;
                for (int jj = j + 1; jj < lw; ++jj) {
                    double *cjj = P + (long)(q0 + jj) * np;
                    cjj[r] = __builtin_fma(-l, rowp[jj], cjj[r]);
                }
            }
        }
        __syncthreads();
    }
    // the U rows of the panel's columns to the right of the leaf: forward substitution with the leaf's unit lower triangle
    const int q = q0 + lw + tid;
    if (q < nbp) {
        double *cq = P + (long)q * np;
        double u[LU_LEAF];
        for (int r = 0; r < lw; ++r) {
            double v = cq[c + r];
            for (int k = 0; k < r; ++k) v = __builtin_fma(-P[(long)(q0 + k) * np + c + r], u[k], v);
            u[r] = v;
            cq[c + r] = v;
        }
    }
}

// rank-lw update of the panel's columns to the right of a leaf: P[q][r] -= sum_k L(r,k) U(k,q), r >= c+lw, q >= q0+lw.
// 256 rows x 16 columns per workgroup; the 16 x 8 U block goes through LDS, all loads are issued before the arithmetic.
constexpr int PU_COLS = 16;
__global__ __launch_bounds__(256) void k_lu_panel_update(double *__restrict__ P, long np, int n, int K0, int c, int lw, int nbp, LuStride ls)
{
    P += ls.p * blockIdx.z;
    __shared__ double ub[PU_COLS][LU_LEAF];
    const int q0 = c - K0;
    const int r = c + lw + blockIdx.x * 256 + threadIdx.x;
    const int qb = q0 + lw + blockIdx.y * PU_COLS;
    if (threadIdx.x < PU_COLS * LU_LEAF) {
        const int qq = threadIdx.x / LU_LEAF, k = threadIdx.x % LU_LEAF;
        ub[qq][k] = (qb + qq < nbp && k < lw) ? P[(long)(qb + qq) * np + c + k] : 0.0;
    }
    const bool ok = r < n;
    double l[LU_LEAF], av[PU_COLS];
#pragma unroll
    for (int k = 0; k < LU_LEAF; ++k) l[k] = (ok && k < lw) ? P[(long)(q0 + k) * np + r] : 0.0;
#pragma unroll
    for (int qq = 0; qq < PU_COLS; ++qq) av[qq] = (ok && qb + qq < nbp) ? P[(long)(qb + qq) * np + r] : 0.0;
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < PU_COLS; ++qq) {
        double v = av[qq];
#pragma unroll
        for (int k = 0; k < LU_LEAF; ++k) v = v - l[k] * ub[qq][k];
        if (ok && qb + qq < nbp) P[(long)(qb + qq) * np + r] = v;
    }
}

// U11 (upper triangle of the panel's top block) goes into W (needed by the back substitution only: off the critical path)
__global__ __launch_bounds__(256) void k_lu_u11_to_w(const double *__restrict__ P, long np, double *__restrict__ w, long ld, int K0, int nbp, LuStride ls)
{
    P += ls.p * blockIdx.z; w += ls.w * blockIdx.z;
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, q0 = blockIdx.y * 32;
    if (r0 > q0 + 31) return;                                  // tile strictly below the diagonal
    for (int y = ty; y < 32; y += 8) {
        const int qq = q0 + y, r = r0 + tx;                   // consecutive threads: consecutive rows of one panel column
        t[y][tx] = (qq < nbp && r < nbp) ? P[(long)qq * np + K0 + r] : 0.0;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int r = r0 + y, qq = q0 + tx;
        if (r < nbp && qq < nbp && r <= qq) w[(long)(K0 + r) * ld + K0 + qq] = t[tx][y];
    }
}

// The composite permutation of a panel's nbp interchanges on the <= 2 nbp rows they touch.  Position idx < 128 is row K0 + idx,
// idx >= 128 is row ipiv[K0 + idx - 128]; after the interchanges position x holds what was at
// src(x) = s_0(s_1(... s_{nbp-1}(x))), s_t = transposition (K0 + t, ipiv[K0 + t]).  One thread per position, the pivots read from
// LDS 16 at a time.  (Worked out inside every workgroup of the gather kernel instead, that kernel took 22 - 37 us.)
__global__ __launch_bounds__(256) void k_lu_perm_src(int K0, int nbp, const int *__restrict__ ipiv, int *__restrict__ src, LuStride ls)
{
    ipiv += ls.ipiv * blockIdx.x; src += 2 * LU_NBO * blockIdx.x;
    __shared__ int piv[LU_NBO];
    if (threadIdx.x < LU_NBO) piv[threadIdx.x] = (int)threadIdx.x < nbp ? ipiv[K0 + threadIdx.x] : K0 + (int)threadIdx.x;
    __syncthreads();
    const int idx = threadIdx.x;
    int x = idx < LU_NBO ? K0 + idx : piv[idx - LU_NBO];
    for (int t0 = nbp - 1; t0 >= 0; t0 -= 16) {
        int pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = piv[max(t0 - u, 0)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = t0 - u, d = K0 + t, p = pv[u];
            if (t >= 0) x = (x == d) ? p : (x == p ? d : x);
        }
    }
    src[idx] = x;
}

constexpr int TRL_T = 512;                                    // threads of the back substitution's combined step kernel
extern __shared__ __attribute__((aligned(16))) double lu_dyn_lds[];      // (the one dynamic LDS block of the LU kernels below)
// ---- the two triangular solves of a 128-row block, on the matrix cores ----
//   LOWER: X = L11^-1 A (unit lower; A = the rows of W as the panel's interchanges leave them, read through k_lu_perm_src's composite
//          permutation -- no separate gather pass; X -> rows K0.. of W; the displaced rows go back to W on the way)
//   UPPER: X = U11^-1 Y (Y, X = rows K0.. of the right-hand-side columns of W)
// A wavefront owns 16 columns and ALL 128 rows of them, in the accumulator layout of v_mfma_f64_4x4x4 (lane = 16 i + c: row 4 q + i of
// quad q, column c; 32 quads = 32 registers), so wavefronts never exchange anything: no barrier after the staging of the triangle.
// Per quad: x = (the quad's 4 x 4 triangle)^-1 a as ONE MFMA -- the 32 small inverses are formed once per workgroup by substitution
// on unit vectors (UPPER: with true divisions by the pivots) -- then every remaining quad gets its update as ONE MFMA: the solved
// quad is already in the B-operand layout, and the A operand (-L or -U, 4 x 4, negated at staging so that the MFMA's multiply-ADD
// subtracts) is one 16-address LDS read, not 16 broadcasts: 528 MFMAs + 528 reads per wavefront, a dependent chain of two MFMAs per
// quad.  Order of the subtractions per entry: quads ascending (LOWER) / descending (UPPER); inside a quad the MFMA's own order.
// Row stride 133: the A read (addresses i * 133 + k, i, k < 4) and the staging of the column-major L (lanes along i) are both
// conflict-free.
// History.  First version (left-looking vector kernels, two barriers and an LDS round trip per 8 rows): 37 / 46 us per launch.
// Right-looking over 8-row blocks, eight wavefronts each keeping one row per block and solving every 8 x 8 triangle for itself:
// 31 / 34 us, bound by the LDS (12 700 broadcast reads per workgroup, each returning 512 bytes); coefficients through scalar loads
// instead: 90 us; through v_readlane: 31 us (vector ALU).  This layout with the quad solved across the lane groups by ds_bpermute
// steps: 40 us (four dependent LDS round trips per quad); with a uniform `q < nq` branch around every MFMA: 53 us (each MFMA behind
// its own LDS round trip).  Now 34-37 us cold / 24 us when launched twice in a row: ~10 us of a launch is the instruction fetch
// of 40 KB of straight-line code, the rest mostly the three memory round trips in front of the MFMAs (permutation, rows, triangle).
// Backward error of the 5892-row ridge solve of a driven reservoir: 2.92e-17 with these solves, 2.91e-17 with substitution
// throughout, 1.79e-17 for LAPACK on the host (tests/test_train_gpu.py prints it).
constexpr int TRM_LD = 133;
constexpr size_t TRM_LDS = sizeof(double) * ((size_t)LU_NBO * TRM_LD + 2 * LU_NBO + 16 * (LU_NBO / 4));
// MODE 0: UPPER (U X = Y, U row-major in W); 1: LOWER, unit diagonal, the LU's panel + its interchanges; 2: LOWER with its own diagonal and no
// interchanges -- the Cholesky's U12 = U11^-T A12, the triangle read as the transpose of U11 in W (L(i,k) = U11(k,i): i contiguous)
template <int MODE>
__device__ __forceinline__ void lu_trsm_mfma_body(const double *__restrict__ tri, long tri_ld_k, long tri_ld_i, double *__restrict__ w, long ld, int K0,
                                                  int nbp, int c0, int ncols, const int *__restrict__ ipiv, const int *__restrict__ src, LuStride ls,
                                                  long tri_stride, int bx, int by, const double *__restrict__ minv = nullptr)
{
    const long tri_stride_minv = ls.p;
    double *T = lu_dyn_lds;                                   // T[i * TRM_LD + k] = -L(i,k), k < i / -U(i,k), k > i; zero elsewhere
    double *dg = T + (size_t)LU_NBO * TRM_LD, *rdg = dg + LU_NBO;
    tri += tri_stride * by; w += ls.w * by;
    const int nthr = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = nthr >> 6;
    const int ri = lane >> 4, cc = lane & 15;
    const int j = c0 + (bx * nwv + wave) * 16 + cc;
    const bool ok = j < ncols;
    constexpr bool LOWER = MODE != 0, UNIT = MODE == 1;
    constexpr int NQ = LU_NBO / 4;
    // The triangle goes through registers in batches of 32 loads per thread (one batch with 512 threads, two with 256); the first
    // batch is in flight while the rows are fetched (with 8 loads per batch a 256-thread workgroup paid eight round trips).
    constexpr int SB = 32;
    double v[SB];
    auto stage_load = [&](int e0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + nthr * u;
            int i, k;
            if (LOWER) { i = e & (LU_NBO - 1); k = e >> 7; }  // L is column-major in the panel buffer: i contiguous
            else { k = e & (LU_NBO - 1); i = e >> 7; }        // U is row-major in W: k contiguous
            const bool in = i < nbp && k < nbp && (MODE == 1 ? k < i : MODE == 2 ? k <= i : k >= i);
            v[u] = in ? tri[(long)i * tri_ld_i + (long)k * tri_ld_k] : 0.0;
        }
    };
    auto stage_store = [&](int e0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + nthr * u;
            const int i = LOWER ? (e & (LU_NBO - 1)) : (e >> 7), k = LOWER ? (e >> 7) : (e & (LU_NBO - 1));
            T[i * TRM_LD + k] = i == k ? 0.0 : -v[u];
            if (!UNIT && i == k) {
                const double d = i < nbp ? v[u] : 1.0;
                dg[i] = d;
            }
        }
    };
    const bool second = nthr * SB < LU_NBO * LU_NBO;          // (uniform)
    stage_load(threadIdx.x);
    double a[NQ];
    if (MODE == 1) {
        ipiv += ls.ipiv * by; src += 2 * LU_NBO * by;
        // every read of this wavefront's columns comes before the first store into them (a displaced row may be another position's source)
        double dv[NQ];
        int pv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int t = 4 * q + ri;
            const bool in = t < nbp;
            pv[q] = in ? ipiv[K0 + t] : -1;
            const int sa = in ? src[t] : K0, sd = in ? src[LU_NBO + t] : K0;
            a[q] = (ok && in) ? w[(long)sa * ld + j] : 0.0;
            dv[q] = (ok && in) ? w[(long)sd * ld + j] : 0.0;
        }
        stage_store(threadIdx.x);
        if (second) stage_load(threadIdx.x + nthr * SB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (pv[q] >= K0 + nbp && ok) w[(long)pv[q] * ld + j] = dv[q];
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int t = 4 * q + ri;
            a[q] = (ok && t < nbp) ? w[(long)(K0 + t) * ld + j] : 0.0;
        }
        stage_store(threadIdx.x);
        if (second) stage_load(threadIdx.x + nthr * SB);
    }
    if (second) stage_store(threadIdx.x + nthr * SB);
    __syncthreads();
    // the inverses of the 32 diagonal 4 x 4 triangles, one thread per column of one inverse (substitution on a unit vector) -- or, for
    // the Cholesky's split form, the inverses its diagonal-block kernel left behind (minv: the bits the fused panel kernel solves with)
    double *dinv = rdg + LU_NBO;                              // dinv[16 q + 4 i + k]
    if (MODE == 2 && minv) {
        for (int e = threadIdx.x; e < 16 * NQ; e += nthr) dinv[e] = minv[(long)tri_stride_minv * by + 4 * K0 + e];
    } else if (threadIdx.x < LU_NBO) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3;
        const double *tq = T + (4 * q) * TRM_LD + 4 * q;      // (negated off-diagonal entries)
        double y[4] = {0.0, 0.0, 0.0, 0.0};
        if (LOWER) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double v = i == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < i) v = __builtin_fma(tq[i * TRM_LD + k], y[k], v);
                if (!UNIT) v = v / dg[4 * q + i];
                y[i] = i >= c ? v : 0.0;
            }
        } else {
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                double v = i == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = 3; k >= 0; --k)
                    if (k > i) v = __builtin_fma(tq[i * TRM_LD + k], y[k], v);
                y[i] = i <= c ? v / dg[4 * q + i] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dinv[16 * q + 4 * i + c] = y[i];
    }
    __syncthreads();
    const double *ta = T + (lane & 3) * TRM_LD + (lane >> 4);                // A operand of quad pair (q2, q): ta[4 q2 * TRM_LD + 4 q]
    const double *ti = dinv + 4 * (lane & 3) + (lane >> 4);                  // A operand of quad q's inverse: ti[16 q]
    // The 32 quads are walked as 8 groups of 4 in a real loop (LOWER: quad p at position p; UPPER: quad 31 - p): the group being
    // solved always sits in r[0], the groups behind it rotate forward after every iteration, so every register index is static while
    // the loop body is 4 quads long.  (Fully unrolled -- 528 MFMAs and as many LDS reads, 45 KB -- the kernel spent ~10 us of every
    // launch fetching instructions it executes once: it sits in a chain of small kernels, its code is never warm.)  Rows and
    // coefficients past a short last block are zeros (diagonal 1).  The operands of all later quads are read in ONE batch per solved
    // quad (a read inside a uniform branch put every MFMA behind its own LDS round trip: 53 us per launch); groups past the end of
    // the block are clamped to its last group and their MFMAs skipped.
    double r[8][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) r[g][t] = a[LOWER ? 4 * g + t : NQ - 1 - (4 * g + t)];
#pragma unroll 1
    for (int G = 0; G < 8; ++G) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pq = 4 * G + t, q = LOWER ? pq : NQ - 1 - pq;
            const double xa = __builtin_amdgcn_mfma_f64_4x4x4f64(ti[16 * q], r[0][t], 0.0, 0, 0, 0);
            if (ok && 4 * q + ri < nbp) w[(long)(K0 + 4 * q + ri) * ld + j] = xa;
            double ag[4], av[7][4];
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const int p2 = 4 * G + t2, q2 = LOWER ? p2 : NQ - 1 - p2;
                if (t2 > t) ag[t2] = ta[4 * q2 * TRM_LD + 4 * q];
            }
#pragma unroll
            for (int g = 1; g < 8; ++g) {
                const int gg = min(G + g, 7);
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const int p2 = 4 * gg + t2, q2 = LOWER ? p2 : NQ - 1 - p2;
                    av[g - 1][t2] = ta[4 * q2 * TRM_LD + 4 * q];
                }
            }
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)                    // nearest quad first: the next solve waits for it
                if (t2 > t) r[0][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ag[t2], xa, r[0][t2], 0, 0, 0);
#pragma unroll
            for (int g = 1; g < 8; ++g) {
                if (G + g < 8) {                              // (uniform)
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2) r[g][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g - 1][t2], xa, r[g][t2], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 7; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) r[g][t] = r[g + 1][t];
    }
}

template <int MODE, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void k_lu_trsm_mfma(const double *__restrict__ tri, long tri_ld_k, long tri_ld_i, double *__restrict__ w, long ld, int K0,
                                                       int nbp, int c0, int ncols, const int *__restrict__ ipiv, const int *__restrict__ src, LuStride ls,
                                                       long tri_stride, const double *__restrict__ minv)
{
    lu_trsm_mfma_body<MODE>(tri, tri_ld_k, tri_ld_i, w, ld, K0, nbp, c0, ncols, ipiv, src, ls, tri_stride, (int)blockIdx.x, (int)blockIdx.y, minv);
}

// back substitution update: Y(i, :) -= sum_k U(i, K0 + k) X(K0 + k, :) for the rows i < K0 above a solved block (K = nb <= 128,
// nrhs <= 136 right-hand sides).  A skinny product (136 columns) of one K-tile: on the MFMA kernel it costs a whole 128 x 128 x 128
// tile's latency (41 us) for two column tiles, one of them 94 % empty.  Here the block of X (<= 139 KB) is staged in LDS once per
// workgroup with 24 rows of U beside it (all 160 KB: one workgroup per CU, 240 workgroups cover 5760 rows in one round -- staging X
// is the cost of a workgroup, so fewer and taller workgroups win: 15 rows per workgroup took 39 us per launch), and each thread
// keeps 8 outputs of two rows: 16 fused multiply-adds per 6 LDS reads.
constexpr int BS_ROWS = 24, BS_CG = 17;                       // 12 row pairs x 17 column groups of 8 = 204 threads
// The rows [row0, min(row0 + BS_ROWS, row_end)) are this workgroup's; the first 256 threads of the workgroup do the work (the body
// also runs inside the 512-thread k_lu_backsub_step).
__device__ __forceinline__ void lu_backsub_update_body(double *__restrict__ y, const double *__restrict__ U, long ld, int K0, int nb, int nrhs, LuStride ls,
                                                       int row0, int row_stride, int row_end, int by)
{
    y += ls.w * by; U += ls.w * by;
    double *xs = lu_dyn_lds;                                  // [nb][BS_CG * 8]
    double *us = lu_dyn_lds + LU_NBO * BS_CG * 8;             // [BS_ROWS][LU_NBO]
    const int tid = threadIdx.x;
    constexpr int XW = BS_CG * 8;
    // (loads in batches of 17 / 12 per thread: a one-element-per-iteration loop waited for every load before the next, 54 us)
    if (tid < 256)
    for (int e0 = 0; e0 < nb * XW; e0 += 256 * 17) {
        double v[17];
#pragma unroll
        for (int m = 0; m < 17; ++m) {
            const int e = e0 + tid + 256 * m, k = e / XW, o = e % XW;
            v[m] = (e < nb * XW && o < nrhs) ? y[(long)(K0 + k) * ld + o] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < 17; ++m) {
            const int e = e0 + tid + 256 * m;
            if (e < nb * XW) xs[e] = v[m];
        }
    }
    const int rp = tid / BS_CG, g = tid % BS_CG;              // row pair, column group
    const int r = 2 * rp;
    // (a workgroup takes the row groups row0, row0 + row_stride, ...: the solved block is staged once, and a launch whose groups
    // outnumber the CUs runs in 1.4 rounds' time instead of two)
    for (; row0 < row_end; row0 += row_stride) {
        if (tid < 256) {
            double v[12];
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int e = tid + 256 * m, rr = e >> 7, k = e & (LU_NBO - 1);
                v[m] = (k < nb && row0 + rr < row_end) ? U[(long)(row0 + rr) * ld + K0 + k] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int e = tid + 256 * m, rr = e >> 7, k = e & (LU_NBO - 1);
                us[rr * LU_NBO + k] = v[m];
            }
        }
        __syncthreads();
        if (tid < 256 && rp < BS_ROWS / 2 && row0 + r < row_end) {
            const bool two = row0 + r + 1 < row_end;
            double acc0[8], acc1[8];
            double *yp0 = y + (long)(row0 + r) * ld + g * 8, *yp1 = yp0 + ld;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                acc0[o] = g * 8 + o < nrhs ? yp0[o] : 0.0;
                acc1[o] = (two && g * 8 + o < nrhs) ? yp1[o] : 0.0;
            }
            const double *ur0 = us + r * LU_NBO, *ur1 = ur0 + LU_NBO;
            for (int k = nb - 1; k >= 0; --k) {               // descending k, as the column-oriented dtrsm subtracts
                const double u0 = ur0[k], u1 = ur1[k];
                const double *xk = xs + k * XW + g * 8;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const double xv = xk[o];
                    acc0[o] = __builtin_fma(-u0, xv, acc0[o]);
                    acc1[o] = __builtin_fma(-u1, xv, acc1[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (g * 8 + o < nrhs) {
                    yp0[o] = acc0[o];
                    if (two) yp1[o] = acc1[o];
                }
        }
        __syncthreads();                                      // (us is restaged for the next group)
    }
}

// The same update for the 128 rows right above a solved block only -- the rows the next triangular solve waits for.  There the cost
// of the tall-workgroup update above is a workgroup's latency (all of X staged by each of 6 workgroups: 31 us), so this launch is cut the other
// way: 32 rows x 16 right-hand sides per workgroup (48 KB staged in one round trip), two outputs per thread, same descending-k
// fused multiply-subtracts.
constexpr int BN_ROWS = 32, BN_COLS = 16;
__global__ __launch_bounds__(256) void k_lu_backsub_near(double *__restrict__ y, const double *__restrict__ U, long ld, int K0, int nb, int nrhs, int row_lo,
                                                          LuStride ls)
{
    y += ls.w * blockIdx.z; U += ls.w * blockIdx.z;
    __shared__ double us[BN_ROWS][LU_NBO + 1];
    __shared__ __attribute__((aligned(16))) double xs[LU_NBO][BN_COLS];
    const int tid = threadIdx.x;
    const int r0 = row_lo + blockIdx.x * BN_ROWS, c0 = blockIdx.y * BN_COLS;
    double uv[16], xv[8];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int e = tid + 256 * m, r = e >> 7, k = e & (LU_NBO - 1);
        uv[m] = (k < nb && r0 + r < K0) ? U[(long)(r0 + r) * ld + K0 + k] : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int e = tid + 256 * m, k = e >> 4, o = e & (BN_COLS - 1);
        xv[m] = (k < nb && c0 + o < nrhs) ? y[(long)(K0 + k) * ld + c0 + o] : 0.0;
    }
    const int row = tid & (BN_ROWS - 1), cp = tid >> 5;        // 32 rows x 8 column pairs
    const bool rok = r0 + row < K0;
    const int col = c0 + 2 * cp;
    double *yp = y + (long)(r0 + row) * ld + col;
    double acc0 = (rok && col < nrhs) ? yp[0] : 0.0, acc1 = (rok && col + 1 < nrhs) ? yp[1] : 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int e = tid + 256 * m;
        us[e >> 7][e & (LU_NBO - 1)] = uv[m];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int e = tid + 256 * m;
        xs[e >> 4][e & (BN_COLS - 1)] = xv[m];
    }
    __syncthreads();
#pragma unroll 8
    for (int k = LU_NBO - 1; k >= 0; --k) {                   // (k >= nb: zeros)
        const double u = us[row][k];
        acc0 = __builtin_fma(-u, xs[k][2 * cp], acc0);
        acc1 = __builtin_fma(-u, xs[k][2 * cp + 1], acc1);
    }
    if (rok && col < nrhs) yp[0] = acc0;
    if (rok && col + 1 < nrhs) yp[1] = acc1;
}

// One step of the back substitution as ONE launch: the first ntr workgroups solve the block at Kt (as k_lu_trsm_mfma<0>), the others
// apply the block solved in the step before (at Ku = Kt + 128) to the rows above Kt -- the two are independent once the rows of the
// block at Kt have had that update (k_lu_backsub_near, the launch before this one), and one after the other
// on one stream they took 34 + 38 us per step.
// (rhs0 = first column of W of this group of right-hand sides: n_aug + a multiple of BS_CG * 8)
__global__ __launch_bounds__(TRL_T) void k_lu_backsub_step(double *__restrict__ w, long ld, int rhs0, int nrhs, int Kt, int nbt, int Ku, int nbu, int ntr,
                                                            int nwu, LuStride ls)
{
    if ((int)blockIdx.x < ntr)          // (512 threads: eight wavefronts x 16 right-hand sides per solving workgroup)
        lu_trsm_mfma_body<0>(w + (long)Kt * ld + Kt, 1L, ld, w, ld, Kt, nbt, rhs0, rhs0 + nrhs, nullptr, nullptr, ls, ls.w, (int)blockIdx.x,
                                 (int)blockIdx.y);
    else
        lu_backsub_update_body(w + rhs0, w, ld, Ku, nbu, nrhs, ls, ((int)blockIdx.x - ntr) * BS_ROWS, nwu * BS_ROWS, Kt, (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Cholesky for the ridge systems: C + diag(beta) is symmetric positive definite by construction (C = sum aug aug^T, a positive diagonal
// added: src/mod_reservoir.f90:1275-1282), so the general pivoted LU above does work the problem does not need: no pivot search, no
// interchanges, no column-by-column leaf chain, and only the triangle of the trailing matrix.  Same storage as the LU: the row-major
// system W = [A | B^T] (right-hand sides as extra columns); A = U^T U with U upper triangular IN the upper triangle of W (row-major:
// the trailing update's operands U12(k, :) are rows of W, read contiguously by the LDS-DMA GEMM -- no panel buffer at all); the
// forward substitution rides along (the U12 solve covers the right-hand-side columns), and the back substitution U X = Y is the LU's own.
//   per 128-row block k:  k_chol_potrf (ONE workgroup: the diagonal block)  ->  k_lu_trsm_mfma<2> (U12 = U11^-T A12, all columns to the
//   right, MFMA)  ->  trailing update A22 -= U12^T U12 with k_gemm_nt_dma at K = 128, lower-trapezoid tile list (i >= j only):
//   block row k+1 first, on the panel stream (the next potrf waits for nothing else), block row k+2 next and the rest after it on the
//   trailing stream.
// A non-positive (or NaN) pivot sets info; the host then solves the system with the LU (indefinite or numerically singular systems).
//
// k_chol_potrf: the 128 x 128 diagonal block in the registers of eight wavefronts, in the accumulator layout of v_mfma_f64_4x4x4 exactly
// as lu_trsm_mfma_body holds its rows: wavefront w owns columns 16 w .. 16 w + 15 and all 128 rows (lane = 16 i + c: row 4 q + i of
// quad q in register q).  Right-looking over the 32 row quads; per quad: (1) the diagonal 4 x 4 goes to LDS and EVERY lane factorises it
// (R^T R, then M = R^-T by substitution: 4 square roots and 4 divisions in a dependent chain -- the chain of the whole kernel), (2) each
// wavefront solves its 16 columns of the row quad with ONE MFMA (M as the A operand) and stores them, (3) the solved row quad goes to
// LDS and every later quad gets its rank-4 update as ONE MFMA (A operand = -U(quad rows, the later quad's columns)^T, a 16-address LDS
// read).  Two barriers per quad (the LDS buffers alternate); wavefront w stops at quad 4 w + 3 (rows below its columns are never read).
// 1 / sqrt(d): v_rsq_f64 and two Newton steps (the instruction alone is good to about 2^-26).  r = d * y is the square root; no
// division anywhere in the chain (sqrt + 1 / r as the compiler expands them are ~35 dependent instructions per pivot).
__device__ __forceinline__ double chol_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-d * y, y, 1.0);
    return __builtin_fma(y * 0.5, e, y);
}

// k_chol_panel: potrf of the diagonal block AND the U12 solve of 64 more columns per workgroup, in one launch.
// Wavefronts 0-3 hold the diagonal block (every workgroup its own copy: the factorisation is repeated, not communicated -- 50 us of
// one workgroup's time either way, and the solve that used to be a second launch of 25-45 us now rides on the same 32 steps);
// wavefront w owns column groups w and 7 - w of it (a triangle cut this way gives every SIMD the same number of MFMAs).
// Wavefronts 4-7 hold one group of 16 columns to the right of the block each (all 128 rows): per step they take the quad's
// inverse M and the solved row quad's A operands from LDS, solve their own row quad (one MFMA) and update all later quads.
// Workgroup 0 stores the factorised diagonal block -- into a side buffer (row K0 + i of u11[.][128]), because the other workgroups read
// the unfactorised block from W during the same launch; k_chol_diag_to_w puts all of them into W before the back substitution.
// Two barriers per row quad.
constexpr int CP_T = 512, CP_COLS = 64;
__global__ __launch_bounds__(256) void k_chol_diag_to_w(const double *__restrict__ u11, double *__restrict__ w, long ld, int n_aug, LuStride ls)
{
    u11 += ls.p * blockIdx.y; w += ls.w * blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(e >> 7), c = (int)(e & (LU_NBO - 1));
    if (row >= n_aug) return;
    const int K0 = row & ~(LU_NBO - 1), col = K0 + c;
    if (col >= row && col < n_aug) w[(long)row * ld + col] = u11[e];
}

__global__ __launch_bounds__(CP_T) void k_chol_panel(double *__restrict__ w, long ld, int K0, int nb, int ncols, double *__restrict__ u11, int *__restrict__ info,
                                                     LuStride ls, int direct, double *__restrict__ minv)
{
    w += ls.w * blockIdx.y; info += blockIdx.y; u11 += ls.p * blockIdx.y; minv += ls.p * blockIdx.y;
    __shared__ __attribute__((aligned(16))) double dq[2][4][4];
    __shared__ __attribute__((aligned(16))) double mq[2][64];
    __shared__ __attribute__((aligned(16))) double xrow[2][4][LU_NBO + 4];
    constexpr int NQ = LU_NBO / 4, NG = NQ / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ri = lane >> 4, cc = lane & 15;
    const bool diag = wave < 4;
    // column groups: diagonal wavefronts g0 = wave, g1 = 7 - wave (columns of the block); the others one group of W's columns
    const int g0 = diag ? wave : NG, g1 = diag ? NG - 1 - wave : -1;         // "row group gi is needed by group g" <=> gi <= g
    const int col0 = diag ? 16 * wave + cc : nb + CP_COLS * (int)blockIdx.x + 16 * (wave - 4) + cc;      // relative to K0
    const int col1 = 16 * (NG - 1 - wave) + cc;
    const bool ok0 = diag ? col0 < nb : K0 + col0 < ncols, ok1 = diag && col1 < nb;
    const bool store_diag = blockIdx.x == 0;
    double a0[NG][4], a1[NG][4];
    {
        const double *bp = w + (long)K0 * ld + K0;
        // every load is issued unconditionally from a clamped (valid) address and the selection follows: 64 independent loads in
        // flight per lane instead of loads behind per-lane conditions
        const int cmax = ncols - 1 - K0, c0c = min(col0, cmax), c1c = min(max(col1, 0), cmax);
        double v0[NQ], v1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long ro = (long)min(4 * q + ri, nb - 1) * ld;
            v0[q] = bp[ro + c0c];
            v1[q] = diag ? bp[ro + c1c] : 0.0;                                 // (wave-uniform)
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = 4 * q + ri;
            // (rows and columns past a short last block: the identity; below the diagonal: never used, kept finite)
            a0[q >> 2][q & 3] = (row < nb && ok0 && (!diag || row <= col0)) ? v0[q] : ((diag && row == col0) ? 1.0 : 0.0);
            a1[q >> 2][q & 3] = (row < nb && ok1 && row <= col1) ? v1[q] : ((diag && row == col1) ? 1.0 : 0.0);
        }
    }
    int bad = 0;
    const double *xr0 = &xrow[0][lane >> 4][lane & 3], *xr1 = &xrow[1][lane >> 4][lane & 3];
#pragma unroll 1
    for (int G = 0; G < NG; ++G) {
        const int ow = G < 4 ? G : NG - 1 - G;                                  // the wavefront that owns the diagonal quads of this row group
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int par = t & 1, q = 4 * G + t, c0q = 4 * t;
            if (wave == ow && cc >= c0q && cc < c0q + 4) dq[par][ri][cc - c0q] = G < 4 ? a0[0][t] : a1[0][t];
            __syncthreads();
            double mv;
            const int row = 4 * q + ri;
            if (diag) {
                // R^T R = D (upper R), then M = R^-T (lower): m(i,k), k <= i
                const double d00 = dq[par][0][0], d01 = dq[par][0][1], d02 = dq[par][0][2], d03 = dq[par][0][3];
                const double d11 = dq[par][1][1], d12 = dq[par][1][2], d13 = dq[par][1][3];
                const double d22 = dq[par][2][2], d23 = dq[par][2][3], d33 = dq[par][3][3];
                const double i0 = chol_rsqrt(d00);
                const double r01 = d01 * i0, r02 = d02 * i0, r03 = d03 * i0;
                const double e11 = __builtin_fma(-r01, r01, d11);
                const double i1 = chol_rsqrt(e11);
                const double r12 = __builtin_fma(-r01, r02, d12) * i1, r13 = __builtin_fma(-r01, r03, d13) * i1;
                const double e22 = __builtin_fma(-r12, r12, __builtin_fma(-r02, r02, d22));
                const double i2 = chol_rsqrt(e22);
                const double r23 = __builtin_fma(-r12, r13, __builtin_fma(-r02, r03, d23)) * i2;
                const double e33 = __builtin_fma(-r23, r23, __builtin_fma(-r13, r13, __builtin_fma(-r03, r03, d33)));
                const double i3 = chol_rsqrt(e33);
                if (!(d00 > 0.0 && e11 > 0.0 && e22 > 0.0 && e33 > 0.0) && !bad) bad = 4 * q + (!(d00 > 0.0) ? 1 : !(e11 > 0.0) ? 2 : !(e22 > 0.0) ? 3 : 4);
                // M = (R^T)^-1, the lower triangle column by column
                const double m10 = -(r01 * i0) * i1;
                const double m21 = -(r12 * i1) * i2, m20 = -__builtin_fma(r12, m10, r02 * i0) * i2;
                const double m32 = -(r23 * i2) * i3, m31 = -__builtin_fma(r23, m21, r13 * i1) * i3;
                const double m30 = -__builtin_fma(r23, m20, __builtin_fma(r13, m10, r03 * i0)) * i3;
                // this lane's element of the MFMA's A operand: (i, k) = (lane & 3, lane >> 4)
                const int mi = lane & 3, mk = lane >> 4;
                const double row0 = mk == 0 ? i0 : 0.0;
                const double row1 = mk == 0 ? m10 : mk == 1 ? i1 : 0.0;
                const double row2 = mk == 0 ? m20 : mk == 1 ? m21 : mk == 2 ? i2 : 0.0;
                const double row3 = mk == 0 ? m30 : mk == 1 ? m31 : mk == 2 ? m32 : i3;
                mv = mi == 0 ? row0 : mi == 1 ? row1 : mi == 2 ? row2 : row3;
                if (wave == 0) {
                    mq[par][lane] = mv;
                    // (the split form's U12 solve takes the quad inverses from here: the same bits as the fused form's)
                    if (direct && store_diag && lane < 64 && ((lane >> 2) & 3) == 0) minv[4 * K0 + 16 * q + 4 * mi + mk] = mv;
                }
                const double xa0 = __builtin_amdgcn_mfma_f64_4x4x4f64(mv, a0[0][t], 0.0, 0, 0, 0);       // U(4 q + i, col)
                const double xa1 = __builtin_amdgcn_mfma_f64_4x4x4f64(mv, a1[0][t], 0.0, 0, 0, 0);
                if (store_diag) {        // (not into W: the other workgroups of this launch may still be loading the block from there --
                    // unless this launch is the diagonal block alone, `direct`)
                    double *dst = direct ? w + (long)(K0 + row) * ld + K0 : u11 + (long)(K0 + row) * LU_NBO;
                    if (row < nb && ok0 && row <= col0) dst[col0] = xa0;
                    if (row < nb && ok1 && row <= col1) dst[col1] = xa1;
                }
                (par ? xrow[1] : xrow[0])[ri][col0] = xa0;
                (par ? xrow[1] : xrow[0])[ri][col1] = xa1;
                a0[0][t] = -xa0; a1[0][t] = -xa1;                             // (the solved quad's registers now hold the update's B operand)
            }
            __syncthreads();
            if (!diag) {
                mv = mq[par][lane];
                const double xa0 = __builtin_amdgcn_mfma_f64_4x4x4f64(mv, a0[0][t], 0.0, 0, 0, 0);
                if (row < nb && ok0) w[(long)(K0 + row) * ld + K0 + col0] = xa0;
                a0[0][t] = -xa0;
            }
            // rank-4 update of the later quads a column group still needs: A operand (i', k) = U(4 q + k, 4 q2 + i') from LDS, B operand
            // -U(quad rows, this group's columns); all operands of the step in one batch of reads, row groups past the end clamped
            const double *xr = par ? xr1 : xr0;
            const double nx0 = a0[0][t], nx1 = a1[0][t];
            double ag[4], av[NG - 1][4];
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
                if (t2 > t) ag[t2] = xr[4 * (4 * G + t2)];
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                const int gg = min(G + g, NG - 1);
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) av[g - 1][t2] = xr[4 * (4 * gg + t2)];
            }
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                if (t2 > t) {
                    if (G <= g0) a0[0][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ag[t2], nx0, a0[0][t2], 0, 0, 0);
                    if (G <= g1) a1[0][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ag[t2], nx1, a1[0][t2], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                if (G + g < NG && (G + g <= g0 || G + g <= g1)) {                 // (uniform)
                    const bool u0 = G + g <= g0, u1 = G + g <= g1;
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2) {
                        if (u0) a0[g][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g - 1][t2], nx0, a0[g][t2], 0, 0, 0);
                        if (u1) a1[g][t2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g - 1][t2], nx1, a1[g][t2], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG - 1; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) { a0[g][t] = a0[g + 1][t]; a1[g][t] = a1[g + 1][t]; }
    }
    if (bad && store_diag && threadIdx.x == 0 && *info == 0) *info = K0 + bad;
}

template <bool A_KC, bool B_KC>
int gemm(const double *A, long sai, long sak, const double *B, long sbj, long sbk, double *C, long ldc, int M, int N, int K,
         double alpha, int lower_only, hipStream_t st, int nbatch = 1, GemmBatch gb = GemmBatch{0, 0, 0})
{
    if (M <= 0 || N <= 0 || K <= 0) return SML_OK;
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, nbatch);
    hipLaunchKernelGGL((k_gemm_acc<A_KC, B_KC>), grid, dim3(GT), 0, st, A, sai, sak, B, sbj, sbk, C, ldc, M, N, K, alpha, lower_only, gb);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

// C += alpha * A * B^T for row-contiguous operands: LDS-DMA kernel on the part of K that is a multiple of 16 (when the
// alignment conditions hold), general kernel on the tail.
// padded: both operands may be read one row past M / N (the LU's buffers are padded to 16 rows / columns).
int gemm_nt(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int M, int N, int K, double alpha, int lower_only,
            hipStream_t st, bool force_dma = false, bool padded = false, int nbatch = 1, GemmBatch gb = GemmBatch{0, 0, 0}, bool half_j = false)
{
    if (M <= 0 || N <= 0 || K <= 0) return SML_OK;
    auto ok = [](const double *p, long ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 1) == 0; };
    int kmain = 0;
    // short K (the shipped batch size m = 98) is bound by the read-modify-write of C: one pass with the general kernel
    // beats main + tail passes.  Long K (m = 2920, the 40-year configuration) takes the LDS-DMA kernel.
    const int Mr = padded ? (M + 1) & ~1 : M, Nr = padded ? (N + 1) & ~1 : N;
    if (ok(A, lda) && ok(B, ldb) && M >= 2 && N >= 2 && !(Mr & 1) && !(Nr & 1) && (K >= 512 || (force_dma && K >= DKT))) {
        kmain = (K / DKT) * DKT;
        const int nbi = (M + BM - 1) / BM, nbj = (N + BN - 1) / BN;
        if (half_j && !lower_only)      // 128 x 64 tiles: twice the workgroups, half the MFMA stream each (a product of few tiles is one tile's latency)
            hipLaunchKernelGGL(k_gemm_nt_dma<8>, dim3(nbi, (N + 63) / 64, nbatch), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain, alpha, 0, Mr, Nr, gb);
        else if (lower_only && nbi > nbj)      // trapezoid: tiles with ti >= tj
            hipLaunchKernelGGL(k_gemm_nt_dma<16>, dim3(nbj * (nbj + 1) / 2 + (nbi - nbj) * nbj, 1, nbatch), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain,
                               alpha, 2, Mr, Nr, gb);
        else if (!lower_only || nbi != nbj)
            hipLaunchKernelGGL(k_gemm_nt_dma<16>, dim3(nbi, nbj, nbatch), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain, alpha, 0, Mr, Nr, gb);
        else
            hipLaunchKernelGGL(k_gemm_nt_dma<16>, dim3(nbi * (nbi + 1) / 2, 1, nbatch), dim3(GT), 0, st, A, lda, B, ldb, C, ldc, M, N, kmain, alpha, 1, Mr, Nr,
                               gb);
        SML_HIP(hipGetLastError());
    }
    if (kmain < K)
        return gemm<false, false>(A + (long)kmain * lda, 1, lda, B + (long)kmain * ldb, 1, ldb, C, ldc, M, N, K - kmain, alpha, lower_only, st, nbatch, gb);
    return SML_OK;
}

}  // namespace

extern "C" {

// The work list of one Gram update for the 256 x 128 kernel: every product of sml_train_accumulate as tiles of one launch.
//   states x states^T (tiles that touch the lower triangle; row tiles aligned to the BOTTOM of the block: a short first row tile
//   of n mod 256 rows meets one or two column tiles, a short last one would meet all of them), states x model^T, model x model^T
//   -> C;  targets x model^T, targets x states^T -> B.
// The bulk (a whole number of rounds of the chip's CUs) runs as full-K items that update the output directly; the ragged rest is
// cut along K into as many pieces as fill one more round, whose partial products land in a scratch and are added to the output
// in a fixed order (deterministic, unlike atomics).  Shipped shape (n = 5760, 132 model rows, 136 targets): 529 + 46 + 2 + 2 + 45
// = 624 tiles = 2 rounds + 112 tiles in 2 pieces each.
constexpr int GB_SUPER_I = 4, GB_SUPER_J = 8;
struct BigPlan {
    int n = 0, n_model = 0, n_out = 0, k = 0, ncu = 0, dev = -1;
    int nitems = 0, ntails = 0, nwg = 0;
    BigItem *d_items = nullptr, *d_tails = nullptr;
    int *d_wg_first = nullptr;
    double *d_partial = nullptr;
    size_t cap_partial = 0;
};

static void big_product(std::vector<BigItem> &tiles, int a_src, int a_n, int b_src, int b_n, int c_out, long c_base, long ldc, bool lower, int K)
{
    // Row tiles from the bottom (the longest tile rows of a triangular product first), in super-tiles of GB_SUPER_I row tiles x
    // GB_SUPER_J column tiles: 32 consecutive tiles of the list then read 4 x 256 + 8 x 128 operand rows between them instead of
    // 32 x 384, and gemm_big_plan hands such runs to one XCD (one L2) each.
    const int nrt = (a_n + GB_I - 1) / GB_I, nct = (b_n + GB_J - 1) / GB_J;
    for (int R = 0; R < nrt; R += GB_SUPER_I)
        for (int Cs = 0; Cs < nct; Cs += GB_SUPER_J)
            for (int r = R; r < std::min(nrt, R + GB_SUPER_I); ++r) {
                const int i_end = a_n - r * GB_I, i0 = std::max(0, i_end - GB_I);
                for (int ct = Cs; ct < std::min(nct, Cs + GB_SUPER_J); ++ct) {
                    const int j0 = ct * GB_J;
                    if (lower && j0 >= i_end) continue;
                    BigItem t{};
                    t.a_src = a_src; t.a_row0 = i0; t.a_rows = a_n - i0;
                    t.b_src = b_src; t.b_row0 = j0; t.b_rows = b_n - j0;
                    t.c_out = c_out; t.c_off = c_base + i0 + (long)j0 * ldc;
                    t.si = i_end - i0; t.sj = std::min(GB_J, b_n - j0);
                    t.k_begin = 0; t.k_end = K; t.dst = 0;
                    tiles.push_back(t);
                }
            }
}

// Workgroup w of a launch runs on XCD w % 8 (round-robin dispatch), 32 at a time per XCD.  Deal a list out so that XCD x works
// through runs x, x + 8, x + 16 ... of 32 consecutive entries: entry e of the list goes to workgroup 8 * (32 * (run / 8) + e % 32) + run % 8.
static std::vector<BigItem> xcd_deal(const std::vector<BigItem> &list)
{
    static const int on = getenv("SML_GEMM_XCD") ? atoi(getenv("SML_GEMM_XCD")) : 1;
    const size_t n = list.size();
    if (!on || n % 256) return list;              // (whole rounds only: a ragged list keeps its order)
    std::vector<BigItem> out(n);
    for (size_t e = 0; e < n; ++e) {
        const size_t run = e / 32;
        out[8 * (32 * (run / 8) + e % 32) + run % 8] = list[e];
    }
    return out;
}

static int gemm_big_plan(BigPlan &P, int n, int n_model, int n_out, int K)
{
    if (P.n == n && P.n_model == n_model && P.n_out == n_out && P.k == K && P.d_items) return SML_OK;
    if (!P.ncu) {
        int dev = 0;
        SML_HIP(hipGetDevice(&dev));
        SML_HIP(hipDeviceGetAttribute(&P.ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int n_aug = n + n_model;
    std::vector<BigItem> tiles;
    big_product(tiles, 0, n, 0, n, 0, (long)n_model + (long)n_model * n_aug, n_aug, true, K);
    if (n_model) {
        big_product(tiles, 0, n, 1, n_model, 0, n_model, n_aug, false, K);
        big_product(tiles, 1, n_model, 1, n_model, 0, 0, n_aug, false, K);
        big_product(tiles, 2, n_out, 1, n_model, 1, 0, n_out, false, K);
    }
    big_product(tiles, 2, n_out, 0, n, 1, (long)n_model * n_out, n_out, false, K);
    const int total = (int)tiles.size();
    static const int split_on = getenv("SML_GEMM_SPLIT") ? atoi(getenv("SML_GEMM_SPLIT")) : 1;
    const int bulk = split_on ? (total / P.ncu) * P.ncu : total, rest = total - bulk;
    std::vector<BigItem> items = xcd_deal(std::vector<BigItem>(tiles.begin(), tiles.begin() + bulk)), tails;
    std::vector<int> wg_first;
    for (int i = 0; i <= bulk; ++i) wg_first.push_back(i);
    int nslots = 0;
    if (rest > 0) {
        // the tail: rest * ktiles K-tiles of work dealt out evenly, in order, to one more round of workgroups
        const long ktiles = K / DKT, units = (long)rest * ktiles;
        for (int r = 0; r < rest; ++r) { tails.push_back(tiles[bulk + r]); tails.back().np = 0; }
        for (int c = 0; c < P.ncu; ++c) {
            const long u0 = units * c / P.ncu, u1 = units * (c + 1) / P.ncu;
            if (u1 <= u0) continue;
            for (long r = u0 / ktiles; r <= (u1 - 1) / ktiles; ++r) {
                BigItem q = tiles[bulk + r];
                q.k_begin = (int)(std::max(u0, r * ktiles) - r * ktiles) * DKT;
                q.k_end = (int)(std::min(u1, (r + 1) * ktiles) - r * ktiles) * DKT;
                q.dst = ++nslots;
                if (!tails[r].np++) tails[r].dst = q.dst;
                items.push_back(q);
            }
            wg_first.push_back((int)items.size());
        }
    }
    // (a plan in use by launches still in flight must not be freed under them)
    SML_HIP(hipDeviceSynchronize());
    if (P.d_items) (void)hipFree(P.d_items);
    if (P.d_tails) (void)hipFree(P.d_tails);
    if (P.d_wg_first) (void)hipFree(P.d_wg_first);
    P.d_items = P.d_tails = nullptr;
    P.d_wg_first = nullptr;
    SML_HIP(hipMalloc((void **)&P.d_wg_first, wg_first.size() * sizeof(int)));
    SML_HIP(hipMemcpy(P.d_wg_first, wg_first.data(), wg_first.size() * sizeof(int), hipMemcpyHostToDevice));
    SML_HIP(hipMalloc((void **)&P.d_items, items.size() * sizeof(BigItem)));
    SML_HIP(hipMalloc((void **)&P.d_tails, std::max<size_t>(tails.size(), 1) * sizeof(BigItem)));
    const size_t need = std::max<size_t>((size_t)nslots, 1) * GB_I * GB_J * sizeof(double);
    if (need > P.cap_partial) {
        if (P.d_partial) (void)hipFree(P.d_partial);
        P.d_partial = nullptr;
        P.cap_partial = 0;
        SML_HIP(hipMalloc((void **)&P.d_partial, need));
        P.cap_partial = need;
    }
    SML_HIP(hipMemcpy(P.d_items, items.data(), items.size() * sizeof(BigItem), hipMemcpyHostToDevice));
    if (rest) SML_HIP(hipMemcpy(P.d_tails, tails.data(), tails.size() * sizeof(BigItem), hipMemcpyHostToDevice));
    P.n = n; P.n_model = n_model; P.n_out = n_out; P.k = K;
    P.nitems = (int)items.size(); P.ntails = rest; P.nwg = (int)wg_first.size() - 1;
    return SML_OK;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// every product of one Gram update over the first K (a multiple of DKT) time columns, in one launch (+ the reduction of the tail)
static int gemm_big_accumulate(const double *states, const double *model, const double *y, int n, int n_model, int n_out, int K, double *c, double *b,
                               hipStream_t st)
{
    // A few plans are kept (shape + K + device): a training pass alternates between its full flushes and a ragged last one, and a
    // plan change costs a device synchronisation.  The least recently used one is rebuilt when a fifth shape turns up.
    static BigPlan plans[4];
    static unsigned long stamp[4] = {0, 0, 0, 0}, tick = 0;
    int dev = 0;
    SML_HIP(hipGetDevice(&dev));
    int slot = -1, lru = 0;
    for (int i = 0; i < 4; ++i) {
        if (plans[i].d_items && plans[i].dev == dev && plans[i].n == n && plans[i].n_model == n_model && plans[i].n_out == n_out && plans[i].k == K) slot = i;
        if (stamp[i] < stamp[lru]) lru = i;
    }
    if (slot < 0) slot = lru;
    stamp[slot] = ++tick;
    BigPlan &plan = plans[slot];
    int rc;
    if (plan.dev != dev) { plan.ncu = 0; plan.n = 0; plan.dev = dev; }
    if ((rc = gemm_big_plan(plan, n, n_model, n_out, K))) return rc;
    BigOperands ops{};
    ops.src[0] = states; ops.ld[0] = n;
    ops.src[1] = model; ops.ld[1] = n_model;
    ops.src[2] = y; ops.ld[2] = n_out;
    ops.out[0] = c; ops.ldo[0] = n + n_model;
    ops.out[1] = b; ops.ldo[1] = n_out;
    static unsigned long long *d_stamps = nullptr;
    if (getenv("SML_GEMM_STAMPS") && !d_stamps) SML_HIP(hipMalloc((void **)&d_stamps, 4 * 4096 * sizeof(unsigned long long)));
    ops.stamps = d_stamps;
    const size_t lds = (size_t)GB_NBUF * GB_SLOT * sizeof(double);
    static const int abl = getenv("SML_GEMM_ABL") ? atoi(getenv("SML_GEMM_ABL")) : 0;   // (profiles/micro/gemm_phase_ablation.py)
#define GB_LAUNCH(X)                                                                                                                  \
    case X:                                                                                                                           \
        SML_HIP(hipFuncSetAttribute((const void *)k_gemm_nt_big<X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(k_gemm_nt_big<X>, dim3((unsigned)plan.nwg), dim3(GB_T), lds, st, ops, 1.0, (const BigItem *)plan.d_items, (const int *)plan.d_wg_first, \
                           plan.d_partial);                                                                                           \
        break;
    switch (abl | (d_stamps ? 8 : 0)) { GB_LAUNCH(0) GB_LAUNCH(1) GB_LAUNCH(2) GB_LAUNCH(3) GB_LAUNCH(4) GB_LAUNCH(7) GB_LAUNCH(8) default: return SML_ERR_ARG; }
#undef GB_LAUNCH
    SML_HIP(hipGetLastError());
    if (ops.stamps) {
        std::vector<unsigned long long> h(4 * (size_t)plan.nwg);
        SML_HIP(hipMemcpy(h.data(), ops.stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0, wv = 0, wb = 0;
        const int cnt = std::min(plan.nwg, 512);
        for (int i = 0; i < cnt; ++i) { cyc += (double)h[4 * i]; ticks += (double)h[4 * i + 1]; wv += (double)h[4 * i + 2]; wb += (double)h[4 * i + 3]; }
        fprintf(stderr, "[gemm_big] full-K items: %.0f shader cycles, %.1f us each (%.3f GHz); %.2f cycles per MFMA; per K-tile: %.0f cycles at the scalar point (incl. ~40 for the stamp itself)\n",
                cyc / cnt, ticks / cnt / 100.0, cyc / ticks * 0.1, cyc / cnt / ((double)(K / DKT) * 256), wv / cnt / (K / DKT));
        (void)wb;
    }
    if (plan.ntails) {
        hipLaunchKernelGGL(k_gemm_big_reduce, dim3(GB_I * GB_J / 256, plan.ntails), dim3(256), 0, st, ops, 1.0, (const BigItem *)plan.d_tails,
                           (const double *)plan.d_partial);
        SML_HIP(hipGetLastError());
    }
    return SML_OK;
}

int sml_train_accumulate(const double *states, const double *model, const double *y, int n, int n_model, int n_out, int m,
                         double *c, double *b, void *stream)
{
    SML_REQUIRE(states && y && c && b && n > 0 && n_model >= 0 && n_out > 0 && m > 0 && (n_model == 0 || model),
                "sml_train_accumulate: bad arguments");
    hipStream_t st = sml::as_stream(stream);
    const int n_aug = n + n_model;
    int rc;
    // Long products: all of them as one launch of the 256 x 128 kernel over the part of m that is a multiple of 8 (the rest of m,
    // at most 7 columns, through the general path below).  Needs even row counts and 16-byte aligned columns for the LDS-DMA.
    // From m = 256 on: at m = 392 (a four-batch flush of the training pass) 0.41 against 0.43 ms for the general path, at 784 the
    // same 0.64; shorter products are bound by the traffic of C either way and keep the reference's one-batch granularity.
    static const int big_on = getenv("SML_GEMM_BIG") ? atoi(getenv("SML_GEMM_BIG")) : 1;
    static const int big_min_m = getenv("SML_GEMM_BIG_MIN_M") ? atoi(getenv("SML_GEMM_BIG_MIN_M")) : 256;
    if (big_on && m >= big_min_m && !(n & 1) && !(n_model & 1) && !(n_out & 1) && n >= GB_I && aligned16(states) && aligned16(y) && (!n_model || aligned16(model))) {
        const int kb = (m / DKT) * DKT;
        if ((rc = gemm_big_accumulate(states, model, y, n, n_model, n_out, kb, c, b, st))) return rc;
        if (kb == m) return SML_OK;
        states += (long)kb * n; y += (long)kb * n_out;
        if (n_model) model += (long)kb * n_model;
        m -= kb;
    }
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

// ---- host side of the LU ----
struct LuSys {                 // scratch and streams of one batch of factorisations
    int nbatch = 0;
    double *w = nullptr, *p[2] = {nullptr, nullptr};
    int *ipiv = nullptr, *info = nullptr, *src = nullptr;
    const double **c_list = nullptr, **b_list = nullptr;      // device arrays of the callers' pointers
    double **wout_list = nullptr;
    hipStream_t sp = nullptr, sg = nullptr;
    hipEvent_t ev_panel = nullptr, ev_strip = nullptr;
    int sg_cus = 0;                // CUs S.sg may use
    bool confined = false;         // the panel stream has the reserved CUs to itself (single solves: see lu_sys_alloc)
    bool sp_masked = false, sg_masked = false;     // created through sml::masked_stream_create
    bool latency_form = false;     // this call came through sml_train_fit (one system, latency first): see backsub_enqueue
    LuStride ls{};
};

static void lu_sys_free(LuSys &s)
{
    if (s.w) (void)hipFree(s.w);
    for (int i = 0; i < 2; ++i) if (s.p[i]) (void)hipFree(s.p[i]);
    if (s.ipiv) (void)hipFree(s.ipiv);
    if (s.info) (void)hipFree(s.info);
    if (s.src) (void)hipFree(s.src);
    if (s.c_list) (void)hipFree((void *)s.c_list);
    if (s.b_list) (void)hipFree((void *)s.b_list);
    if (s.wout_list) (void)hipFree((void *)s.wout_list);
    if (s.sp) (void)(s.sp_masked ? sml::masked_stream_destroy(s.sp) : (int)hipStreamDestroy(s.sp));
    if (s.sg) (void)(s.sg_masked ? sml::masked_stream_destroy(s.sg) : (int)hipStreamDestroy(s.sg));
    if (s.ev_panel) (void)hipEventDestroy(s.ev_panel);
    if (s.ev_strip) (void)hipEventDestroy(s.ev_strip);
    s = LuSys{};
}

static inline long lu_pad16(long v) { return (v + 15) & ~15L; }
static inline bool lu_dma() { static const bool v = !(getenv("SML_LU_DMA") && atoi(getenv("SML_LU_DMA")) == 0); return v; }

static int lu_sys_alloc(LuSys &s, int n_aug, int ncols, int nbatch, bool chol = false)
{
    const long ld = lu_pad16(ncols), np = lu_pad16(n_aug);
    s.nbatch = nbatch;
    s.ls = LuStride{(long)n_aug * ld, (long)LU_NBO * np, n_aug};
    SML_HIP(hipMalloc((void **)&s.w, (size_t)nbatch * s.ls.w * sizeof(double)));
    for (int i = 0; i < 2; ++i) SML_HIP(hipMalloc((void **)&s.p[i], (size_t)nbatch * s.ls.p * sizeof(double)));
    SML_HIP(hipMalloc((void **)&s.ipiv, (size_t)nbatch * n_aug * sizeof(int)));
    SML_HIP(hipMalloc((void **)&s.info, (size_t)nbatch * sizeof(int)));
    SML_HIP(hipMalloc((void **)&s.src, (size_t)nbatch * 2 * LU_NBO * sizeof(int)));
    SML_HIP(hipMalloc((void **)&s.c_list, (size_t)nbatch * sizeof(double *)));
    SML_HIP(hipMalloc((void **)&s.b_list, (size_t)nbatch * sizeof(double *)));
    SML_HIP(hipMalloc((void **)&s.wout_list, (size_t)nbatch * sizeof(double *)));
    // Both streams are created with an explicit CU mask even where the mask is "every CU": a masked stream owns its hardware queue,
    // while plain streams share the runtime's small pool of queues with every other stream of the process, and two streams on one
    // queue run in order -- the panel chain then no longer overlaps the trailing update (measured: the Cholesky single solve took
    // 10.3 ms instead of 7.2 once two more plain streams existed in the process, profiles/micro/fit_solvers.py with FIT_PRE98=1).
    // The trailing updates run on a stream whose CU mask leaves a few compute units free: a leaf workgroup needs a whole CU
    // (all of its registers), and behind an unmasked GEMM grid it waited for one to drain (measured: 250 us instead of 30).
    {
        static const int reserve_env = getenv("SML_LU_RESERVE") ? atoi(getenv("SML_LU_RESERVE")) : -1;
        static const int confine_env = getenv("SML_LU_CONFINE") ? atoi(getenv("SML_LU_CONFINE")) : 1;
        static const int layout_env = getenv("SML_LU_MASK_LAYOUT") ? atoi(getenv("SML_LU_MASK_LAYOUT")) : 0;
        // A single solve: the panel stream is confined to the reserved CUs and does only what fits there (leaves, panel updates and
        // the NEXT panel's 128 columns); what it writes it reads back from the same L2s instead of through a memory system the
        // trailing update keeps busy.  Batches keep the unconfined panel stream: their strip work is nbatch times larger.
        // (the Cholesky's single solve: no CU masks at all -- 7.1 ms against 7.7 with the LU's confined layout, whose 64-96 reserved CUs
        //  its 93-workgroup panel kernel over- or under-fills)
        s.confined = confine_env == 2 || (confine_env && nbatch == 1 && !chol);
        // (LU, confined: 32 / 48 / 64 / 96 CUs -> 27.2 / 27.5 / 26.9 / 26.9 ms.  The Cholesky's panel kernel is 93 one-per-CU workgroups at
        //  the first panel: 96 reserved CUs hold them in one round; with 64 it ran in two: 133 instead of ~70 us)
        const int reserve = reserve_env >= 0 ? reserve_env : s.confined ? 64 : (chol && nbatch == 1) ? 0 : std::max(16, 2 * nbatch);
        int dev = 0, ncu = 0;
        SML_HIP(hipGetDevice(&dev));
        SML_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        s.sg_cus = (reserve > 0 && reserve < ncu) ? ncu - reserve : ncu;
        if (reserve > 0 && reserve < ncu) {
            // (mask bit i is CU i / 8 of XCD i % 8.  layout 0: the first `reserve` bits = reserve / 8 CUs of every XCD; 1: CUs of XCD 0 only;
            //  2: whole XCDs -- all CUs of the first reserve / 32 XCDs)
            auto reserved = [&](int cu) {
                return layout_env == 0 ? cu < reserve : layout_env == 1 ? (cu % 8 == 0 && cu / 8 < reserve) : (cu % 8) < reserve / 32;
            };
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u), pm((ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu; ++i) (reserved(i) ? pm : mask)[i / 32] |= 1u << (i % 32);
            int rcm = sml::masked_stream_create(&s.sg, mask.data(), (int)mask.size());
            if (rcm) return rcm;
            s.sg_masked = true;
            if (s.confined) {
                if ((rcm = sml::masked_stream_create(&s.sp, pm.data(), (int)pm.size()))) return rcm;
                s.sp_masked = true;
            }
        }
        std::vector<uint32_t> every((ncu + 31) / 32, 0u);
        for (int i = 0; i < ncu; ++i) every[i / 32] |= 1u << (i % 32);
        if (!s.sg) {
            int rcm = sml::masked_stream_create(&s.sg, every.data(), (int)every.size());
            if (rcm) return rcm;
            s.sg_masked = true;
        }
        if (!s.sp) {
            int rcm = sml::masked_stream_create(&s.sp, every.data(), (int)every.size());
            if (rcm) return rcm;
            s.sp_masked = true;
        }
    }
    SML_HIP(hipEventCreateWithFlags(&s.ev_panel, hipEventDisableTiming));
    SML_HIP(hipEventCreateWithFlags(&s.ev_strip, hipEventDisableTiming));
    return SML_OK;
}

// leaf shape by height: (threads, rows per thread, columns); R x LW doubles + ~56 registers must stay within 256
struct LeafShape { int threads, slots, width; };
static LeafShape leaf_shape(int rows_left)
{
    if (rows_left <= 14 * 256) return {256, (rows_left + 255) / 256, 8};
    const int r = (rows_left + 511) / 512;
    if (r > 14) return {512, 0, 8};                  // taller than the registers hold: k_lu_leaf_tall (slots = 0)
    return {512, r, r <= 12 ? 8 : 6};
}

static int leaf_width(int rows_left) { return leaf_shape(rows_left).width; }

static long long *g_lu_stamps = nullptr;                     // SML_LU_STAMP=1: 32 words per leaf (slot = first column / 2) of system 0
constexpr int LU_STAMP_LEAVES = 4096;

static int launch_leaf(LuSys &S, int nb, long np, int n, int K0, int c, int lw, int nbp)
{
    const LeafShape sh = leaf_shape(n - c);
    long long *stamps = (g_lu_stamps && c / 2 < LU_STAMP_LEAVES) ? g_lu_stamps + 32 * (c / 2) : nullptr;
    hipStream_t st = S.sp;
    double *Pk = S.p[(K0 / LU_NBO) & 1];
#define LEAF_CASE(T, R, LW)                                                                                                                      \
    case R:                                                                                                                                      \
        if (lw == LW) hipLaunchKernelGGL((k_lu_leaf<T, R, LW, true>), dim3(nb), dim3(T), 0, st, Pk, np, n, K0, c, lw, nbp, S.ipiv, S.info, stamps, S.ls);  \
        else hipLaunchKernelGGL((k_lu_leaf<T, R, LW, false>), dim3(nb), dim3(T), 0, st, Pk, np, n, K0, c, lw, nbp, S.ipiv, S.info, stamps, S.ls);          \
        break
    if (sh.slots == 0)
        hipLaunchKernelGGL(k_lu_leaf_tall, dim3(nb), dim3(512), 0, st, Pk, np, n, K0, c, lw, nbp, S.ipiv, S.info, S.ls);
    else if (sh.threads == 256) {
        switch (sh.slots) {
            LEAF_CASE(256, 1, 8); LEAF_CASE(256, 2, 8); LEAF_CASE(256, 3, 8); LEAF_CASE(256, 4, 8); LEAF_CASE(256, 5, 8); LEAF_CASE(256, 6, 8);
            LEAF_CASE(256, 7, 8); LEAF_CASE(256, 8, 8); LEAF_CASE(256, 9, 8); LEAF_CASE(256, 10, 8); LEAF_CASE(256, 11, 8); LEAF_CASE(256, 12, 8);
            LEAF_CASE(256, 13, 8); LEAF_CASE(256, 14, 8);
        default: return sml::fail(SML_ERR_STATE, "sml_train_fit: no leaf kernel for %d rows", n - c);
        }
    } else {
        switch (sh.slots) {
            LEAF_CASE(512, 8, 8); LEAF_CASE(512, 9, 8); LEAF_CASE(512, 10, 8); LEAF_CASE(512, 11, 8); LEAF_CASE(512, 12, 8);
            LEAF_CASE(512, 13, 6); LEAF_CASE(512, 14, 6);
        default: return sml::fail(SML_ERR_ARG, "sml_train_fit: n_aug = %d exceeds the %d rows the register-resident LU leaf holds", n, 14 * 512);
        }
    }
#undef LEAF_CASE
    SML_HIP(hipGetLastError());
    return SML_OK;
}

// trailing update W[c0.., j0..j1) -= L21 U12 (row-major C: the contiguous column index j is the MFMA kernel's i)
static int lu_trailing(LuSys &S, int nb, long ld, long np, int n_aug, int K0, int nbp, int j0, int j1, const double *Pk, hipStream_t st)
{
    const int c0 = K0 + nbp;
    static const bool half_env = getenv("SML_LU_STRIP_HALF") ? atoi(getenv("SML_LU_STRIP_HALF")) != 0 : true;
    const bool half_j = half_env && nb == 1 && j1 - j0 <= LU_NBO;          // the strip's product of a single solve: in the leaf chain
    return gemm_nt(S.w + (long)K0 * ld + j0, ld, Pk + c0, np, S.w + (long)c0 * ld + j0, ld, j1 - j0, n_aug - c0, nbp, -1.0, 0, st, lu_dma(), /*padded=*/true,
                   nb, GemmBatch{S.ls.w, S.ls.p, S.ls.w}, half_j);
}

// Back substitution U X = Y on the right-hand-side columns of W (U = the upper triangle of W's rows, from the LU or the Cholesky), on
// S.sg behind everything S.sp did; right-hand sides in groups of BS_CG * 8 = 136 columns (one group for the shipped 132 / 136 outputs).
// L = U^T into the strictly lower triangle of W (free after the Cholesky factorisation: the trailing updates only ever touched
// col >= row): W[c][r] = W[r][c] for r < c < n_aug.  The back substitution of a BATCH reads U(rows above, block) through it as the
// row-contiguous operand of the LDS-DMA GEMM.  32 x 32 tiles through LDS.
namespace {
__global__ __launch_bounds__(256) void k_chol_mirror_u(double *__restrict__ w, long ld, int n_aug, LuStride ls)
{
    if (blockIdx.y > blockIdx.x) return;                     // tile (row block by, column block bx) of the upper triangle
    w += ls.w * blockIdx.z;
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < n_aug && c < n_aug) ? w[(long)r * ld + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;           // element (row c, column r) of the lower triangle
        if (c < n_aug && r < c) w[(long)c * ld + r] = t[tx][ty + 8 * i];
    }
}
}  // namespace

static int backsub_enqueue(LuSys &S, int nb, int n_aug, int n_out, long ld, bool chol = false)
{
    const LuStride ls = S.ls;
    SML_HIP(hipEventRecord(S.ev_panel, S.sp));
    SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
    const size_t bs_lds = (size_t)(LU_NBO * BS_CG * 8 + BS_ROWS * LU_NBO) * sizeof(double);
    static bool bs_attr = false;
    if (!bs_attr) {
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_backsub_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(bs_lds, TRM_LDS)));
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_trsm_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRM_LDS));
        bs_attr = true;
    }
    static const int far_env = getenv("SML_CHOL_BACKSUB_GEMM") ? atoi(getenv("SML_CHOL_BACKSUB_GEMM")) : -1;
    static const int far_half_env = getenv("SML_CHOL_BACKSUB_HALF") ? atoi(getenv("SML_CHOL_BACKSUB_HALF")) : 0;
    static const int two_env = getenv("SML_CHOL_BACKSUB_2S") ? atoi(getenv("SML_CHOL_BACKSUB_2S")) : 1;
    // The far rows' update after a Cholesky: a product on the matrix cores for everything that came through sml_train_fit_batched (any
    // count: one arithmetic, so a queue that trains in groups of 1 or 64 gets the same bits); sml_train_fit, the one-system latency
    // form, keeps the launch that solves a block beside the vector-FMA update of the far rows (7.1 against 7.7 ms; its W_out differs
    // from the batched form's in the last bits, both at a backward error of 3-4e-17).
    const bool far_gemm = chol && (far_env >= 0 ? far_env != 0 : !S.latency_form), far_half = far_half_env != 0;
    if (far_gemm) {
        const int nt = (n_aug + 31) / 32;
        hipLaunchKernelGGL(k_chol_mirror_u, dim3(nt, nt, nb), dim3(256), 0, S.sg, S.w, ld, n_aug, ls);
        SML_HIP(hipGetLastError());
    }
    for (int r0 = 0; r0 < n_out; r0 += BS_CG * 8) {
        const int nr = std::min(BS_CG * 8, n_out - r0), rhs0 = n_aug + r0;
        // U(i,k) = W[(K0 + i) * ld + K0 + k]; the right-hand sides are columns rhs0 .. rhs0 + nr of W
        auto solve_block = [&](int K0) {
            hipLaunchKernelGGL(k_lu_trsm_mfma<0>, dim3((nr + 63) / 64, nb), dim3(256), TRM_LDS, S.sg, S.w + (long)K0 * ld + K0, 1L, ld, S.w, ld, K0,
                               std::min(LU_NBO, n_aug - K0), rhs0, rhs0 + nr, (const int *)nullptr, (const int *)nullptr, ls, ls.w, (const double *)nullptr);
        };
        // per step: the 128 rows of the next block get the update first (they are what the chain waits for), then ONE launch solves that
        // block and updates the rows above it (one workgroup per CU: at most as many far-update workgroups as S.sg has CUs left)
        const int K_last = ((n_aug - 1) / LU_NBO) * LU_NBO, ntm = (nr + 127) / 128;
        solve_block(K_last);
        for (int K0 = K_last; K0 > 0; K0 -= LU_NBO) {
            const int Kn = K0 - LU_NBO;
            hipLaunchKernelGGL(k_lu_backsub_near, dim3(LU_NBO / BN_ROWS, (nr + BN_COLS - 1) / BN_COLS, nb), dim3(256), 0, S.sg, S.w + rhs0, S.w, ld, K0,
                               std::min(LU_NBO, n_aug - K0), nr, Kn, ls);
            if (Kn > 0 && far_gemm) {
                // the block at Kn is solved by its own launch and the rows above it get the block at K0's update as a product on the
                // matrix cores (U through its mirror image below the diagonal) instead of vector FMAs in the same launch (16 systems:
                // 253 us per step) -- on S.sp, beside the solve: they touch different rows
                const int nbu = std::min(LU_NBO, n_aug - K0);
                hipStream_t fst = two_env ? S.sp : S.sg;
                if (two_env) {
                    SML_HIP(hipEventRecord(S.ev_strip, S.sg));
                    SML_HIP(hipStreamWaitEvent(S.sp, S.ev_strip, 0));
                }
                int rcg = gemm_nt(S.w + (long)K0 * ld + rhs0, ld, S.w + (long)K0 * ld, ld, S.w + rhs0, ld, nr, Kn, nbu, -1.0, 0, fst, lu_dma(), /*padded=*/true, nb,
                                  GemmBatch{S.ls.w, S.ls.w, S.ls.w}, far_half);
                if (rcg) return rcg;
                solve_block(Kn);
                if (two_env) {
                    SML_HIP(hipEventRecord(S.ev_panel, S.sp));
                    SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
                }
            } else if (Kn > 0) {
                const int groups = (Kn + BS_ROWS - 1) / BS_ROWS;
                const int nwu = nb == 1 ? std::min(groups, std::max(1, S.sg_cus - ntm)) : groups;
                hipLaunchKernelGGL(k_lu_backsub_step, dim3(ntm + nwu, nb), dim3(TRL_T), std::max(bs_lds, TRM_LDS), S.sg, S.w, ld, rhs0, nr, Kn, LU_NBO, K0,
                                   std::min(LU_NBO, n_aug - K0), ntm, nwu, ls);
            } else
                solve_block(0);
            SML_HIP(hipGetLastError());
        }
    }
    return SML_OK;
}

// enqueue the ridge solves of systems [first, first + nb) (no host synchronisation): everything is ordered on S.sg / S.sp
static int fit_enqueue(double *const *c, const double *const *b, int first, int nb, int n, int n_model, int n_out, double beta_res, double beta_model,
                       double prior_val, int using_prior, double *const *wout, LuSys &S)
{
    const int n_aug = n + n_model, ncols = n_aug + n_out;
    const long ld = lu_pad16(ncols), np = lu_pad16(n_aug);
    const LuStride ls = S.ls;
    int rc;
    static bool trl_attr = false;
    if (!trl_attr) {
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_trsm_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRM_LDS));
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_trsm_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRM_LDS));
        trl_attr = true;
    }
    for (int i = 0; i < nb; ++i)
        if ((rc = sml_train_symmetrize(c[first + i], n_aug, (void *)S.sg))) return rc;
    SML_HIP(hipMemsetAsync(S.info, 0, sizeof(int) * nb, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.c_list, c + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.b_list, b + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.wout_list, wout + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    // with a prior the betas enter squared (quirk Q8, src/mod_reservoir.f90:1271-1290)
    const double reg_model = using_prior ? beta_model * beta_model : beta_model;
    const double reg_res = using_prior ? beta_res * beta_res : beta_res;
    const double prior_diag = using_prior ? prior_val * (beta_model * beta_model) : 0.0;
    hipLaunchKernelGGL(k_build_system, dim3((unsigned)((ld + 255) / 256), n_aug, nb), dim3(256), 0, S.sg, S.c_list, S.b_list, S.w, ld, n_aug, n_model, n_out,
                       reg_model, reg_res, prior_diag, ls, 0);
    SML_HIP(hipGetLastError());
    auto strip_to_panel = [&](int K0, int nbp, double *P, hipStream_t st) {
        hipLaunchKernelGGL(k_lu_strip_to_panel, dim3((n_aug - K0 + 31) / 32, (nbp + 31) / 32, nb), dim3(256), 0, st, S.w, ld, P, np, n_aug, K0, nbp, ls);
    };
    // Streams: everything the NEXT leaf waits for -- the panel's own chain, then the interchanges, U12 and the next panel's strip --
    // is one in-order sequence on S.sp; only the rest of the trailing update goes to the CU-masked S.sg, behind an event.  (With
    // the interchanges / U12 / strip on S.sg, as before, the chain crossed streams twice per panel, and a cross-stream dependency that
    // is not yet satisfied when it is reached costs 13-15 us: 1.3 ms per solve.)  S.sp waits for S.sg only where the wait is long over:
    // panel k+1's interchanges touch the columns panel k's trailing update wrote a leaf chain earlier.
    strip_to_panel(0, std::min(LU_NBO, n_aug), S.p[0], S.sg);
    SML_HIP(hipEventRecord(S.ev_strip, S.sg));
    SML_HIP(hipStreamWaitEvent(S.sp, S.ev_strip, 0));           // (once: build_system and the first strip are on S.sg)
    for (int K0 = 0, k = 0; K0 < n_aug; K0 += LU_NBO, ++k) {
        const int nbp = std::min(LU_NBO, n_aug - K0);
        double *Pk = S.p[k & 1];
        for (int cc = K0, lw = 0; cc < K0 + nbp; cc += lw) {
            lw = std::min(leaf_width(n_aug - cc), K0 + nbp - cc);
            if ((rc = launch_leaf(S, nb, np, n_aug, K0, cc, lw, nbp))) return rc;
            const int rows = n_aug - (cc + lw), cols = K0 + nbp - (cc + lw);
            if (rows > 0 && cols > 0)
                hipLaunchKernelGGL(k_lu_panel_update, dim3((rows + 255) / 256, (cols + PU_COLS - 1) / PU_COLS, nb), dim3(256), 0, S.sp, Pk, np, n_aug, K0, cc, lw,
                                   nbp, ls);
        }
        SML_HIP(hipGetLastError());
        if (k > 0) SML_HIP(hipStreamWaitEvent(S.sp, S.ev_strip, 0));       // panel k-1's trailing update (and U11 copy) has finished
        const int c0 = K0 + nbp;
        hipLaunchKernelGGL(k_lu_perm_src, dim3(nb), dim3(256), 0, S.sp, K0, nbp, S.ipiv, S.src, ls);
        // interchanges and U12 (L(i,k) = Pk[K0 + i + k * np]): in the chain for the columns of the next panel only when the panel stream
        // is confined (cs .. ce), for all columns otherwise; the other columns follow on S.sg
        const int ce = (S.confined && c0 < n_aug) ? c0 + std::min(LU_NBO, n_aug - c0) : ncols;    // (exactly the columns lu_trailing updates on S.sp)
        auto gather_and_u12 = [&](int j0, int j1, hipStream_t st) {
            if (j1 <= j0) return;
            hipLaunchKernelGGL(k_lu_trsm_mfma<1>, dim3((j1 - j0 + 63) / 64, nb), dim3(256), TRM_LDS, st, Pk + K0, np, 1L, S.w, ld, K0, nbp, j0, j1, S.ipiv,
                               S.src, ls, ls.p, (const double *)nullptr);
        };
        if (S.confined) {                                                    // S.sg needs the permutation; it is long done with panel k-1
            SML_HIP(hipEventRecord(S.ev_panel, S.sp));
            SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
            gather_and_u12(ce, ncols, S.sg);
        }
        gather_and_u12(c0, ce, S.sp);
        SML_HIP(hipGetLastError());
        // U12 of this panel is in W: the rest of the trailing update may start (beside the strip's product, which it slows from 33 to
        // 50 us; started after the strip instead it overlaps that much more of the next leaf chain: 28.4 against 28.2 ms per solve)
        if (!S.confined) {
            SML_HIP(hipEventRecord(S.ev_panel, S.sp));
            SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
        }
        if (c0 < n_aug) {
            const int nbn = std::min(LU_NBO, n_aug - c0);
            if ((rc = lu_trailing(S, nb, ld, np, n_aug, K0, nbp, c0, c0 + nbn, Pk, S.sp))) return rc;   // the next panel's columns, in the chain
            strip_to_panel(c0, nbn, S.p[(k + 1) & 1], S.sp);
            SML_HIP(hipGetLastError());
            if ((rc = lu_trailing(S, nb, ld, np, n_aug, K0, nbp, c0 + nbn, ncols, Pk, S.sg))) return rc;  // the rest beside the next leaf chain
        }
        hipLaunchKernelGGL(k_lu_u11_to_w, dim3((nbp + 31) / 32, (nbp + 31) / 32, nb), dim3(256), 0, S.sg, Pk, np, S.w, ld, K0, nbp, ls);
        SML_HIP(hipGetLastError());
        SML_HIP(hipEventRecord(S.ev_strip, S.sg));                          // (reused: "S.sg is done with panel k")
    }
    if ((rc = backsub_enqueue(S, nb, n_aug, n_out, ld))) return rc;
    const long tw = (long)n_aug * n_out;
    hipLaunchKernelGGL(k_extract_wout, dim3((unsigned)((tw + 255) / 256), nb), dim3(256), 0, S.sg, S.w, ld, S.wout_list, n_aug, n_out, ls);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

// ---- the Cholesky chain of systems [first, first + nb) (see k_chol_potrf) ----
static int chol_update(LuSys &S, int nb, long ld, int ncols, int K0, int nbp, int r0, int nrows, hipStream_t st, bool whole_rows)
{
    // W[r0 + j][r0 + i] -= sum_k U(K0 + k, r0 + j) U(K0 + k, r0 + i) for j < nrows, i >= j (and every i when the block is one tile row)
    double *u = S.w + (long)K0 * ld + r0;
    const bool half_j = whole_rows && nb == 1;
    return gemm_nt(u, ld, u, ld, S.w + (long)r0 * ld + r0, ld, ncols - r0, nrows, nbp, -1.0, whole_rows ? 0 : 1, st, lu_dma(), /*padded=*/true, nb,
                   GemmBatch{S.ls.w, S.ls.w, S.ls.w}, half_j);
}

static int chol_enqueue(double *const *c, const double *const *b, int first, int nb, int n, int n_model, int n_out, double beta_res, double beta_model,
                        double prior_val, int using_prior, double *const *wout, LuSys &S)
{
    const int n_aug = n + n_model, ncols = n_aug + n_out;
    const long ld = lu_pad16(ncols);
    const LuStride ls = S.ls;
    int rc;
    static bool attr = false;
    if (!attr) {
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_trsm_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRM_LDS));
        SML_HIP(hipFuncSetAttribute((const void *)k_lu_trsm_mfma<2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRM_LDS));
        attr = true;
    }
    SML_HIP(hipMemsetAsync(S.info, 0, sizeof(int) * nb, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.c_list, c + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.b_list, b + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    SML_HIP(hipMemcpyAsync((void *)S.wout_list, wout + first, sizeof(double *) * nb, hipMemcpyHostToDevice, S.sg));
    const double reg_model = using_prior ? beta_model * beta_model : beta_model;
    const double reg_res = using_prior ? beta_res * beta_res : beta_res;
    const double prior_diag = using_prior ? prior_val * (beta_model * beta_model) : 0.0;
    hipLaunchKernelGGL(k_build_system, dim3((unsigned)((ld + 255) / 256), n_aug, nb), dim3(256), 0, S.sg, S.c_list, S.b_list, S.w, ld, n_aug, n_model, n_out,
                       reg_model, reg_res, prior_diag, ls, 1);
    SML_HIP(hipGetLastError());
    SML_HIP(hipEventRecord(S.ev_strip, S.sg));
    SML_HIP(hipStreamWaitEvent(S.sp, S.ev_strip, 0));
    // Streams and grouping.  S.sp carries what the next diagonal block waits for: the panel kernel (potrf + U12) and the update of
    // the NEXT block row by every panel of the current group (K = 128, 256, ...).  The rest of the trailing matrix is updated once
    // per GROUP of `grp` panels, at K = 128 grp, on S.sg: a 128-deep update reads and writes all of C for 128 columns' worth of
    // flops (half of either roof), two panels per pass halve that traffic.  The group's update starts with the `grp` block rows the
    // next group's S.sp updates touch; S.sp waits for that part only (ev_strip), the rows below were last written by S.sg, in order.
    // Measured: one system (sml_train_fit) 7.05 / 7.33 / 7.69 / 7.94 ms at grp = 1 / 2 / 3 / 4 -- its chain is what it waits for, and the
    // chain's part of a grouped update still runs per panel; 16 in lockstep 2.83 / 2.54 / 2.45 / 2.49 / 2.54 / 2.61 ms per system at
    // 1 / 2 / 3 / 4 / 5 / 8, 8 in lockstep 3.27 / 3.03 at 1 / 3 (at K = 128 a trailing update sits on the ridge between the two roofs: 11
    // flop per byte of C).  Everything that comes through sml_train_fit_batched uses 3 whatever its count (one arithmetic).
    static const int grp_env = getenv("SML_CHOL_GROUP") ? std::max(1, atoi(getenv("SML_CHOL_GROUP"))) : 0;
    const int grp = grp_env ? grp_env : S.latency_form ? 1 : 3;
    // One system: the fused panel kernel (every workgroup repeats the diagonal block's factorisation: one launch less in the chain).
    // A batch: the diagonal blocks alone (one workgroup per system) and the MFMA solve as a second launch -- 93 x nb workgroups that
    // each occupy a CU for the whole factorisation would take the chip from the trailing updates the batch is there to fill it with
    // (16 in lockstep: 3.5 against 3.16 ms per system).
    static const int fused_env = getenv("SML_CHOL_FUSED") ? atoi(getenv("SML_CHOL_FUSED")) : -1;
    const bool fused = fused_env >= 0 ? fused_env != 0 : nb == 1;
    int g_K0 = 0, gi = 0, ngroups = 0;
    for (int K0 = 0; K0 < n_aug; K0 += LU_NBO) {
        const int nbp = std::min(LU_NBO, n_aug - K0), c0 = K0 + nbp;
        if (fused)
            hipLaunchKernelGGL(k_chol_panel, dim3(std::max(1, (ncols - c0 + CP_COLS - 1) / CP_COLS), nb), dim3(CP_T), 0, S.sp, S.w, ld, K0, nbp, ncols, S.p[0], S.info, ls, 0, S.p[1]);
        else {
            // (a launch of the diagonal block alone: ncols = c0 leaves the other wavefronts without columns)
            hipLaunchKernelGGL(k_chol_panel, dim3(1, nb), dim3(CP_T), 0, S.sp, S.w, ld, K0, nbp, c0, S.p[0], S.info, ls, 1, S.p[1]);
            // (512 threads: the triangle's 136 KB of LDS allow one workgroup per CU, so eight wavefronts give every SIMD two dependent MFMA
            //  chains to interleave; 16 / 8 in lockstep 2.43 -> 2.40 / 2.92 -> 2.86 ms per system.  SML_CHOL_TRSM_T=256: four wavefronts)
            static const int trsm_t = getenv("SML_CHOL_TRSM_T") ? atoi(getenv("SML_CHOL_TRSM_T")) : 512;
            if (trsm_t == 512)
                hipLaunchKernelGGL((k_lu_trsm_mfma<2, 512>), dim3((ncols - c0 + 127) / 128, nb), dim3(512), TRM_LDS, S.sp, S.w + (long)K0 * ld + K0, ld, 1L, S.w, ld, K0, nbp,
                                   c0, ncols, (const int *)nullptr, (const int *)nullptr, ls, ls.w, (const double *)S.p[1]);
            else
            hipLaunchKernelGGL(k_lu_trsm_mfma<2>, dim3((ncols - c0 + 63) / 64, nb), dim3(256), TRM_LDS, S.sp, S.w + (long)K0 * ld + K0, ld, 1L, S.w, ld, K0, nbp, c0, ncols,
                               (const int *)nullptr, (const int *)nullptr, ls, ls.w, (const double *)S.p[1]);
        }
        SML_HIP(hipGetLastError());
        if (c0 >= n_aug) break;
        const int kacc = c0 - g_K0;                                            // rows of U the group has produced so far
        if (gi == 0 && ngroups > 0) SML_HIP(hipStreamWaitEvent(S.sp, S.ev_strip, 0));   // S.sg is done with the block rows this group's S.sp updates
        const int nb1 = std::min(LU_NBO, n_aug - c0);
        const int r1 = c0 + nb1;
        const bool close = gi == grp - 1 || r1 >= n_aug;
        if (close) {                                                           // (recorded before the strip: S.sg needs the panels only)
            SML_HIP(hipEventRecord(S.ev_panel, S.sp));
            SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
        }
        if ((rc = chol_update(S, nb, ld, ncols, g_K0, kacc, c0, nb1, S.sp, true))) return rc;
        if (close) {
            if (r1 < n_aug) {
                const int na = std::min(grp * LU_NBO, n_aug - r1);
                if ((rc = chol_update(S, nb, ld, ncols, g_K0, kacc, r1, na, S.sg, na <= LU_NBO))) return rc;
                SML_HIP(hipEventRecord(S.ev_strip, S.sg));
                const int r2 = r1 + na;
                if (r2 < n_aug && (rc = chol_update(S, nb, ld, ncols, g_K0, kacc, r2, n_aug - r2, S.sg, false))) return rc;
            }
            g_K0 = c0; gi = 0; ++ngroups;
        } else
            ++gi;
    }
    // the factorised diagonal blocks go into W (they were kept aside while their launch's other workgroups read the originals)
    SML_HIP(hipEventRecord(S.ev_panel, S.sp));
    SML_HIP(hipStreamWaitEvent(S.sg, S.ev_panel, 0));
    if (fused) hipLaunchKernelGGL(k_chol_diag_to_w, dim3((unsigned)(((long)n_aug * LU_NBO + 255) / 256), nb), dim3(256), 0, S.sg, S.p[0], S.w, ld, n_aug, ls);
    if ((rc = backsub_enqueue(S, nb, n_aug, n_out, ld, true))) return rc;
    const long tw = (long)n_aug * n_out;
    hipLaunchKernelGGL(k_extract_wout, dim3((unsigned)((tw + 255) / 256), nb), dim3(256), 0, S.sg, S.w, ld, S.wout_list, n_aug, n_out, ls);
    SML_HIP(hipGetLastError());
    // the caller's C leaves as the full symmetric matrix the reference's states_x_states_aug is (the factorisation read its lower
    // triangle only): mirrored last, off the chain
    for (int i = 0; i < nb; ++i)
        if ((rc = sml_train_symmetrize(c[first + i], n_aug, (void *)S.sg))) return rc;
    return SML_OK;
}

// Several independent ridge solves at once: up to FIT_BATCH equally sized systems are factorised in lockstep by ONE chain of
// launches (the grid's batch dimension), so the latency-bound panel chain is paid once per batch and the trailing updates of all of
// them fill the chip together.  c/b/wout are host arrays of device pointers.  Synchronises; returns SML_ERR_NUMERIC if any system is
// singular (sml_last_error names the first).
// The scratch of a batch (per system: the row-major system, two panel buffers: ~300 MB at n_aug = 5892) and its streams are kept
// between calls (allocating them took 12 ms of a 43 ms call), one workspace for single solves and one for batches -- their stream
// layouts differ (lu_sys_alloc) -- each tied to the device it was made on.  sml_train_release_workspace frees them.
constexpr int FIT_BATCH = 16;            // (8 / 16 / 32 systems in lockstep: 7.3 / 6.3 / 6.5 ms per 5892-row system, profiles/micro/fit_batch_sizes.py)
struct FitWorkspace { LuSys sys; int n_aug = 0, ncols = 0, dev = -1; };
static FitWorkspace g_ws[4];             // [0] single solves, [1] batches of the LU; [2], [3] the same for the Cholesky (other stream layouts)
static int g_solver = -1;                // 0 auto (Cholesky, LU where it breaks down), 1 LU, 2 Cholesky only; -1: not read from the environment yet

int sml_train_release_workspace(void)
{
    for (FitWorkspace &ws : g_ws) {
        lu_sys_free(ws.sys);
        ws.n_aug = ws.ncols = 0; ws.dev = -1;
    }
    return SML_OK;
}

/* 0 = Cholesky with the LU where a pivot is not positive (default; SML_FIT_SOLVER=lu|chol|auto presets it), 1 = always the pivoted LU
 * (dgesv's algorithm: what mldivide does for a general matrix), 2 = Cholesky only (SML_ERR_NUMERIC when the system is not positive
 * definite).  Returns the previous setting. */
int sml_train_select_solver(int solver)
{
    if (g_solver < 0) {
        const char *e = getenv("SML_FIT_SOLVER");
        g_solver = (e && !strcmp(e, "lu")) ? 1 : (e && !strcmp(e, "chol")) ? 2 : 0;
    }
    const int prev = g_solver;
    if (solver >= 0 && solver <= 2) g_solver = solver;
    return prev;
}

static int fit_run(bool chol, int count, double *const *c, const double *const *b, int n, int n_model, int n_out, double beta_res, double beta_model,
                   double prior_val, int using_prior, double *const *wout, hipStream_t st, std::vector<int> &hinfo, bool latency_form)
{
    const int n_aug = n + n_model, ncols = n_aug + n_out;
    static const int fit_batch = getenv("SML_FIT_BATCH") ? std::max(1, atoi(getenv("SML_FIT_BATCH"))) : FIT_BATCH;
    const int nbmax = std::min(count, fit_batch);
    int rc = SML_OK, dev = 0;
    SML_HIP(hipGetDevice(&dev));
    FitWorkspace &ws = g_ws[(chol ? 2 : 0) + (nbmax == 1 ? 0 : 1)];
    if (ws.n_aug != n_aug || ws.ncols != ncols || ws.sys.nbatch < nbmax || ws.dev != dev) {
        lu_sys_free(ws.sys);
        ws.n_aug = ws.ncols = 0; ws.dev = -1;
        if ((rc = lu_sys_alloc(ws.sys, n_aug, ncols, nbmax, chol))) { lu_sys_free(ws.sys); return rc; }
        ws.n_aug = n_aug; ws.ncols = ncols; ws.dev = dev;
    }
    LuSys &S = ws.sys;
    S.latency_form = latency_form;
    static const bool want_stamps = getenv("SML_LU_STAMP") && atoi(getenv("SML_LU_STAMP"));
    if (want_stamps && !g_lu_stamps) {
        SML_HIP(hipMalloc((void **)&g_lu_stamps, sizeof(long long) * 32 * LU_STAMP_LEAVES));
        SML_HIP(hipMemset(g_lu_stamps, 0, sizeof(long long) * 32 * LU_STAMP_LEAVES));
    }
    // the caller's stream must have produced C and B before the factorisation's streams read them
    hipEvent_t fork = nullptr;
    SML_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    hipError_t fe = hipEventRecord(fork, st);
    if (fe == hipSuccess) fe = hipStreamWaitEvent(S.sg, fork, 0);
    (void)hipEventDestroy(fork);
    if (fe != hipSuccess) return sml::fail(SML_ERR_HIP, "sml_train_fit_batched: %s", hipGetErrorString(fe));
    hinfo.assign(count, 0);
    for (int first = 0; first < count && rc == SML_OK; first += S.nbatch) {
        const int nb = std::min(S.nbatch, count - first);
        rc = chol ? chol_enqueue(c, b, first, nb, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout, S)
                  : fit_enqueue(c, b, first, nb, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout, S);
        // (the pointer lists and info are reused by the next batch: wait for this one)
        hipError_t e = hipStreamSynchronize(S.sg);
        if (e == hipSuccess) e = hipStreamSynchronize(S.sp);
        if (e != hipSuccess && rc == SML_OK) rc = sml::fail(SML_ERR_HIP, "sml_train_fit_batched: %s", hipGetErrorString(e));
        if (rc == SML_OK && hipMemcpy(hinfo.data() + first, S.info, sizeof(int) * nb, hipMemcpyDeviceToHost) != hipSuccess)
            rc = sml::fail(SML_ERR_HIP, "sml_train_fit_batched: reading info failed");
    }
    if (rc) return rc;
    if (g_lu_stamps && !chol) {
        std::vector<long long> h(32 * LU_STAMP_LEAVES);
        if (hipMemcpy(h.data(), g_lu_stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(getenv("SML_LU_STAMP_FILE") ? getenv("SML_LU_STAMP_FILE") : "lu_stamps.bin", "wb")) {
                fwrite(h.data(), sizeof(long long), h.size(), f);
                fclose(f);
            }
        }
    }
    return SML_OK;
}

static int fit_systems(int count, double *const *c, const double *const *b, int n, int n_model, int n_out, double beta_res,
                       double beta_model, double prior_val, int using_prior, double *const *wout, void *stream, bool latency_form)
{
    SML_REQUIRE(count > 0 && c && b && wout && n > 0 && n_model >= 0 && n_out > 0, "sml_train_fit_batched: bad arguments");
    for (int i = 0; i < count; ++i) SML_REQUIRE(c[i] && b[i] && wout[i], "sml_train_fit_batched: null system %d", i);
    hipStream_t st = sml::as_stream(stream);
    const int solver = sml_train_select_solver(-1);
    // (no size limit: panels taller than 14 x 512 rows go through k_lu_leaf_tall instead of the register-resident leaf)
    std::vector<int> hinfo;
    int rc;
    if (solver != 1) {
        if ((rc = fit_run(true, count, c, b, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout, st, hinfo, latency_form))) return rc;
        std::vector<int> redo;
        for (int i = 0; i < count; ++i)
            if (hinfo[i]) redo.push_back(i);
        if (redo.empty()) return SML_OK;
        if (solver == 2)
            return sml::fail(SML_ERR_NUMERIC, "sml_train_fit: system %d is not positive definite (Cholesky pivot %d is not positive)", redo[0], hinfo[redo[0]]);
        // the systems the Cholesky could not factorise (indefinite, or singular to working precision): dgesv's algorithm
        std::vector<double *> c2, w2;
        std::vector<const double *> b2;
        for (int i : redo) { c2.push_back(c[i]); b2.push_back(b[i]); w2.push_back(wout[i]); }
        std::vector<int> info2;
        if ((rc = fit_run(false, (int)redo.size(), c2.data(), b2.data(), n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, w2.data(), st, info2, latency_form))) return rc;
        for (size_t t = 0; t < redo.size(); ++t)
            if (info2[t]) return sml::fail(SML_ERR_NUMERIC, "sml_train_fit: system %d: U(%d,%d) is exactly zero; the factorisation is singular (dgesv info=%d)", redo[t], info2[t], info2[t], info2[t]);
        return SML_OK;
    }
    if ((rc = fit_run(false, count, c, b, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout, st, hinfo, latency_form))) return rc;
    for (int i = 0; i < count; ++i)
        if (hinfo[i]) return sml::fail(SML_ERR_NUMERIC, "sml_train_fit: system %d: U(%d,%d) is exactly zero; the factorisation is singular (dgesv info=%d)", i, hinfo[i], hinfo[i], hinfo[i]);
    return SML_OK;
}

int sml_train_fit_batched(int count, double *const *c, const double *const *b, int n, int n_model, int n_out, double beta_res,
                          double beta_model, double prior_val, int using_prior, double *const *wout, void *stream)
{
    return fit_systems(count, c, b, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, wout, stream, /*latency_form=*/false);
}

int sml_train_fit(double *c, const double *b, int n, int n_model, int n_out, double beta_res, double beta_model,
                  double prior_val, int using_prior, double *wout, void *stream)
{
    SML_REQUIRE(c && b && wout && n > 0 && n_model >= 0 && n_out > 0, "sml_train_fit: bad arguments");
    double *cc[1] = {c};
    const double *bb[1] = {b};
    double *ww[1] = {wout};
    return fit_systems(1, cc, bb, n, n_model, n_out, beta_res, beta_model, prior_val, using_prior, ww, stream, /*latency_form=*/true);
}

}  // extern "C"
