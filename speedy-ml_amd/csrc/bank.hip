// Reservoir bank: every local reservoir of one rank resident in HBM, stepped by two batched kernels.
//
// Replaces predict / predict_ml / synchronize (src/mod_reservoir.f90:1354-1535) and mklsparse /
// smatrix_vector (src/mod_linalg.f90:10-25, 516-531).  What one reference `predict` call does per reservoir
//     y = A x (MKL COO SpMV) ; temp = matmul(win, feedback) ; x = (1-leak) x + leak tanh(y+temp)
//     outvec = matmul(wout, [local_model ; x with even entries squared]) ; unstandardize
// becomes, for ALL resident reservoirs at once:
//   k_update  : one sparse product with the fused operator [A | W_in] on [x ; u] in SELL-64 layout (rows sorted
//               by length, 64-row slices stored column-major so a wavefront's loads are contiguous), x and u staged
//               in LDS, tanh + leak fused, ping-pong state buffers.  HBM-bound: 12 B per nonzero.
//   k_readout : W_out is re-laid out row-major; a one-wavefront workgroup streams R=17 rows (17 x 47 KB) with
//               16-byte loads, every lane keeping 17 partial sums so that 17 independent loads are in flight per
//               lane; the augmented state is formed on the fly (local_model | x, odd 0-based entries squared);
//               wavefront shuffle + LDS reduction; un-standardisation (multiply, then add -- two roundings, as
//               src/mod_utilities.f90:667-831) fused into the epilogue.  HBM-bound: 8 B per W_out element.
// W_in is kept as whatever nonzeros the dense (n,d) array holds (one per row as shipped,
// src/mod_reservoir.f90:272-280); dropping exact zeros leaves every sum bit-identical.
//
// Workgroup -> (reservoir, part) mapping is XCD-aware: consecutive block ids are dealt round-robin over the 8 XCDs,
// so block id b serves reservoir 8*(b/(8*parts)) + b%8; the `parts` workgroups that share one reservoir's x / x~
// vector therefore share an XCD's L2.
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <mutex>
#include <vector>

#include "bank.h"

using sml::HostRes;
using sml::ResDesc;

namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int RO_ROWS = 17;

__device__ __forceinline__ void decode_block(int id, int parts, int res_begin, int &res, int &part)
{
    const int xcd = id & 7;
    const int t = id >> 3;
    part = t % parts;
    res = res_begin + (t / parts) * 8 + xcd;
}

struct TrainSlot { double *states; int n; };      // a slot's states(n, .) buffer of the training pass (NULL: slot not trained)

// x_new = (1-leak) x + leak tanh([A|Win] [x;u])        (src/mod_reservoir.f90:1444-1448)
// train_slots (optional): the training pass also wants the new state as column train_col of the slot's states buffer, in the
// REFERENCE's row order with the even (1-based) rows squared (src/mod_reservoir.f90:1133) -- stored from here, one launch less
// per time column than a separate copy kernel.
// Pointers that come out of a reservoir descriptor in memory are generic to the compiler, and loads through them are FLAT loads: they
// count on BOTH wait counters, so every wait for an LDS gather (lgkmcnt) also waits for whatever global loads are in flight -- which is
// what kept a prefetch of the next slice's entries from ever overlapping the current slice's gathers (rounds 2-3: "prefetch: slower").
// The descriptor's arrays live in device memory by construction (sml_bank_load_*): say so, and the loads are global loads (vmcnt only).
typedef const double __attribute__((address_space(1))) *gcd_t;
typedef double __attribute__((address_space(1))) *gd_t;
typedef const unsigned short __attribute__((address_space(1))) *gcus_t;
typedef const int __attribute__((address_space(1))) *gci_t;
__device__ __forceinline__ gcd_t as_global(const double *p) { return (gcd_t)p; }
__device__ __forceinline__ gd_t as_global(double *p) { return (gd_t)p; }
__device__ __forceinline__ gcus_t as_global(const unsigned short *p) { return (gcus_t)p; }
__device__ __forceinline__ gci_t as_global(const int *p) { return (gci_t)p; }

typedef const float __attribute__((address_space(1))) *gcf_t;
__device__ __forceinline__ gcf_t as_global(const float *p) { return (gcf_t)p; }

// VAL32: the operator's values come from the compact float copy (ResDesc::sell_val32: every value is exactly a float, so the conversion
// back is exact and the sums are the same sums) -- 6 bytes per stored nonzero instead of 10.
template <int THREADS, bool VAL32 = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_update(const ResDesc *__restrict__ descs, int res_begin, int res_end,
                                                     int parts, const double *__restrict__ u_all, int u_stride, int cur, int square_input,
                                                     int whole_blocks, const TrainSlot *__restrict__ train_slots, int train_col)
{
    // Work split: the first whole_blocks workgroups take one reservoir each (x staged once per reservoir); the remaining
    // reservoirs -- the ragged last round of the chip's resident workgroups -- are cut into `parts` slice ranges so that the
    // tail is a round of short workgroups instead of a half-empty round of long ones.
    // square_input: the operand of A x is the state with its even (1-based) entries squared while the leak term keeps the
    // state itself -- the step after a batch flush in the ML-only training loop (quirk Q6, src/mod_reservoir.f90:1031-1044)
    extern __shared__ __attribute__((aligned(16))) double xu[];
    int res, part;
    if ((int)blockIdx.x < whole_blocks) { decode_block(blockIdx.x, 1, res_begin, res, part); parts = 1; }
    else decode_block((int)blockIdx.x - whole_blocks, parts, res_begin + whole_blocks, res, part);
    if (res >= res_end) return;
    const ResDesc &D = descs[res];   // (by reference: a copy with x[cur] indexed at run time lives in scratch, 136 B per lane)
    if (!D.loaded) return;
    const gcd_t x = as_global(cur ? D.x[1] : D.x[0]);
    const gd_t xn = as_global(cur ? D.x[0] : D.x[1]);
    const gcd_t u = as_global(u_all) + (size_t)res * u_stride;
    const gci_t slice_off = as_global(D.slice_off);
    const gcus_t sell_col = as_global(D.sell_col);
    const gcd_t sell_val = as_global(D.sell_val);
    const gcf_t sell_val32 = as_global(D.sell_val32);
    const int nslices = D.nslices, n = D.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64;
    const int stride = parts * NW;
    int s = part * NW + wave;
    // the first slice's metadata goes out before [x ; u] is staged
    int off_n = 0, end_n = 0;
    if (s < nslices) { off_n = slice_off[s]; end_n = slice_off[s + 1]; }
    // stage [x ; u] with 16-byte loads (x is 256-byte aligned)
    typedef const f64x2 __attribute__((address_space(1))) *gcd2_t;
    const int n2 = n >> 1;
    if (!square_input) {
        for (int i = threadIdx.x; i < n2; i += THREADS) reinterpret_cast<f64x2 *>(xu)[i] = ((gcd2_t)x)[i];
    } else {
        for (int i = threadIdx.x; i < n2; i += THREADS) {
            f64x2 v = ((gcd2_t)x)[i];
            v[1] = v[1] * v[1];                      // device position parity == reference row parity (see load_common)
            reinterpret_cast<f64x2 *>(xu)[i] = v;
        }
    }
    if ((n & 1) && threadIdx.x == 0) xu[n - 1] = x[n - 1];
    for (int i = threadIdx.x; i < D.d; i += THREADS) xu[n + i] = u[i];
    __syncthreads();
    // slice metadata is fetched one slice ahead (and, for the first slice, before [x ; u] is staged), so each slice costs
    // one exposed memory round trip (its entries) instead of two
    for (; s < nslices; s += stride) {
        const int off = off_n, width = (end_n - off_n) >> 6;
        if (s + stride < nslices) { off_n = slice_off[s + stride]; end_n = slice_off[s + stride + 1]; }
        const int r = s * 64 + lane;                      // device position == sorted position: contiguous stores
        const gcus_t cp = sell_col + off + lane;
        const gcd_t vp = sell_val + off + lane;
        const gcf_t vp32 = sell_val32 + off + lane;
        double acc = 0.0;
        // The whole row (makesparse gives 6-8 stored entries per row incl. W_in) is fetched in ONE batch of loads before
        // the first LDS gather: PMC showed 78 % of the wave cycles parked in s_waitcnt, and a 4-wide loop plus a scalar tail
        // exposed the memory latency four times per slice.  Accumulation stays in storage order (A in COO order, then W_in).
        // (Round 4: the descriptor's arrays are addressed as GLOBAL memory -- through the generic pointers of the descriptor these were
        // flat loads, which count on the LDS wait counter as well.  With that out of the way the next slice's batch was prefetched under
        // this slice's gathers once more: 0.178 ms at the 111 registers it needs (4 wavefronts per SIMD), 0.32 ms spilling at 80 --
        // against 0.112: the kernel lives on 24 wavefronts per CU, not on one wavefront's look-ahead.)
        constexpr int WB = 8;
        int cc[WB];
        double vv[WB];
        float vf[WB];           // (VAL32: converted after the batch -- a conversion inside the guarded load made the compiler wait for every pair)
#pragma unroll
        for (int q = 0; q < WB; ++q) {
            cc[q] = 0; vv[q] = 0.0; vf[q] = 0.f;
            if (VAL32) {
                // (branch-free: an entry past the slice's width re-reads entry 0 and is replaced by the padding; guarded loads of the
                //  float copy compiled into one waited-for pair per entry)
                const int qq = q < width ? q : 0;                          // width is wave-uniform
                const int c = cp[qq * 64];
                const float f = vp32[qq * 64];
                cc[q] = q < width ? c : 0;
                vf[q] = q < width ? f : 0.f;
            } else if (q < width) {
                cc[q] = cp[q * 64];
                vv[q] = vp[q * 64];
            }
        }
        if (VAL32) {
#pragma unroll
            for (int q = 0; q < WB; ++q) vv[q] = (double)vf[q];
        }
        // The gathers are unconditional and issued together (one LDS round trip per slice, not one per entry behind a divergent
        // branch each): an entry past the row's length is the layout's padding (value +0.0, column 0), whose product is a zero
        // of either sign, and acc -- which starts at +0.0 and therefore is never -0.0 -- plus a zero is acc, bit for bit
        // (as long as x[0] is finite; a state that is not has tripped the range guard anyway).
        // (two batches of four: eight gathered values at once cost 8 more registers than the 80 that 6 wavefronts per SIMD allow)
#pragma unroll
        for (int h = 0; h < WB; h += 4) {
            double xg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xg[q] = xu[cc[h + q]];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += vv[h + q] * xg[q];
        }
        for (int j = WB; j < width; j += 4) {                                // long rows (rare): four more per trip
            int c4[4];
            double v4[4], x4[4];
            float f4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c4[q] = 0; v4[q] = 0.0; f4[q] = 0.f;
                if (j + q < width) { c4[q] = cp[(j + q) * 64]; if (VAL32) f4[q] = vp32[(j + q) * 64]; else v4[q] = vp[(j + q) * 64]; }
            }
            if (VAL32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v4[q] = (double)f4[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) x4[q] = xu[c4[q]];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v4[q] * x4[q];
        }
        if (r < n) {
#ifdef SML_EXPERIMENT_NO_TANH                            // (profiles/micro/update_floor.sh: what the update costs without its tanh)
            const double xt = acc;
#else
            const double xt = tanh(acc);
#endif
            const double v = (1.0 - D.leak) * (square_input ? (double)x[r] : xu[r]) + D.leak * xt;
            xn[r] = v;
            if (train_slots) {
                const TrainSlot t = train_slots[res];
                if (t.states) {
                    const int rr = D.perm[r];              // device position -> reference row (same parity)
                    t.states[(size_t)train_col * n + rr] = (rr & 1) ? v * v : v;
                }
            }
        }
    }
}

// states(:, col) <- x with the even (1-based) rows squared, in the REFERENCE's row order (src/mod_reservoir.f90:1133):
// the Gram matrices and W_out are defined over the reference's state indices.  One workgroup row per slot.
__global__ void k_store_state(const ResDesc *__restrict__ descs, const TrainSlot *__restrict__ ts, int nslots, int col, int cur)
{
    const int slot = blockIdx.y;
    if (slot >= nslots) return;
    const ResDesc &D = descs[slot];
    const TrainSlot t = ts[slot];
    if (!D.loaded || !t.states) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.n) return;
    const double v = (cur ? D.x[1] : D.x[0])[p];
    const int r = D.perm[p];                       // device position -> reference row (same parity as p)
    t.states[(size_t)col * D.n + r] = (r & 1) ? v * v : v;
}

__global__ void k_zero_state(const ResDesc *__restrict__ descs, int nslots, int cur)
{
    const int slot = blockIdx.y;
    if (slot >= nslots) return;
    const ResDesc &D = descs[slot];
    if (!D.loaded) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < D.n) (cur ? D.x[1] : D.x[0])[p] = 0.0;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// outvec = unstandardize(Wout [local_model ; x~])      (src/mod_reservoir.f90:1450-1471)
template <int R, int RO_THREADS, bool NT>
__global__ __launch_bounds__(RO_THREADS) void k_readout(const ResDesc *__restrict__ descs, int res_begin, int res_end, int parts,
                                                          const double *__restrict__ lm_all, int lm_stride,
                                                          double *__restrict__ out_all, int out_stride, int cur, int flags,
                                                          double *__restrict__ partial_all)
{
    // flags: bit0 raw (no un-standardisation); bits 1-2 = column split of the product (sml_bank_readout_part):
    //   0 all columns; 2 state columns only, raw sums -> partial_all; 4 physics-model columns only, added to partial_all
    __shared__ double red[RO_THREADS / 64][R];
    int res, grp;
    decode_block(blockIdx.x, parts, res_begin, res, grp);
    if (res >= res_end) return;
    const ResDesc &D = descs[res];   // (by reference: a copy with x[cur] indexed at run time lives in scratch, 136 B per lane)
    if (!D.loaded) return;
    const int r0 = grp * R;
    if (r0 >= D.n_out) return;
    const double *__restrict__ x = cur ? D.x[1] : D.x[0];
    const double *__restrict__ lm = lm_all + (size_t)res * lm_stride;
    const size_t ld = (size_t)D.n_aug_pad;
    const double *wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = D.wout + (size_t)min(r0 + r, D.n_out - 1) * ld;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;

    const bool aligned_model = (D.n_model & 1) == 0;
    const int csplit = (D.n_model + 1) & ~1;      // even (16-byte column pairs); every physics-model column is below it
    const int kk_begin = (flags & 2) ? csplit : 0, kk_end = (flags & 4) ? csplit : D.n_aug_pad;
    for (int kk = kk_begin + threadIdx.x * 2; kk < kk_end; kk += RO_THREADS * 2) {
        double a0, a1;
        if (kk + 1 < D.n_model) { a0 = lm[kk]; a1 = lm[kk + 1]; }
        else if (aligned_model && kk >= D.n_model && kk + 1 < D.n_aug) {
            const double2 xv = *reinterpret_cast<const double2 *>(x + (kk - D.n_model));
            a0 = xv.x; a1 = xv.y * xv.y;                    // 0-based odd entry == the reference's even (1-based) entry
        } else {
            auto aug = [&](int i) -> double {
                if (i < D.n_model) return lm[i];
                if (i >= D.n_aug) return 0.0;
                const int j = i - D.n_model;
                const double v = x[j];
                return (j & 1) ? v * v : v;
            };
            a0 = aug(kk); a1 = aug(kk + 1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // streamed once per step, 7.4 GB per sweep >> Infinity Cache: non-temporal 16-byte loads
            const f64x2 w = NT ? __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(wrow[r] + kk))
                               : *reinterpret_cast<const f64x2 *>(wrow[r] + kk);
            acc[r] += w[0] * a0;
            acc[r] += w[1] * a1;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) red[wave][r] = s;
    }
    __syncthreads();
    if (threadIdx.x < R) {
        const int row = r0 + threadIdx.x;
        if (row < D.n_out) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < RO_THREADS / 64; ++w) v += red[w][threadIdx.x];
            if (flags & 2) { partial_all[(size_t)res * out_stride + row] = v; return; }
            if (flags & 4) v = partial_all[(size_t)res * out_stride + row] + v;
            if (!(flags & 1)) {
                const int si = D.out_stat[row];
                if (si >= 0) {
                    // unstandardize_data_*: data*std, then +mean, two roundings (src/mod_utilities.f90:667-831)
                    v = __dadd_rn(__dmul_rn(v, D.stdv[si]), D.mean[si]);
                }
            }
            out_all[(size_t)res * out_stride + row] = v;
        }
    }
}

// The readout from the COMPACT copy of W_out (ResDesc::wout32: every weight is exactly a float -- what a reservoir read from the reference's
// NetCDF weight files holds, NF90_REAL -- so the conversion back to double is exact): half the bytes of the dominant kernel.  Same
// structure as k_readout; a 16-byte load is four columns here, so a lane's partial sums associate differently from the 8-byte-value
// kernel (each is a fixed order; the two agree to the last bits' accumulation, 1e-13).  The whole product only (no column split).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int R, int RO_THREADS>
__global__ __launch_bounds__(RO_THREADS) void k_readout32(const ResDesc *__restrict__ descs, int res_begin, int res_end, int parts,
                                                            const double *__restrict__ lm_all, int lm_stride,
                                                            double *__restrict__ out_all, int out_stride, int cur, int flags)
{
    __shared__ double red[RO_THREADS / 64][R];
    int res, grp;
    decode_block(blockIdx.x, parts, res_begin, res, grp);
    if (res >= res_end) return;
    const ResDesc &D = descs[res];
    if (!D.loaded) return;
    const int r0 = grp * R;
    if (r0 >= D.n_out) return;
    const double *__restrict__ x = cur ? D.x[1] : D.x[0];
    const double *__restrict__ lm = lm_all + (size_t)res * lm_stride;
    const size_t ld = (size_t)D.n_aug_pad32;
    const float *wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = D.wout32 + (size_t)min(r0 + r, D.n_out - 1) * ld;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    const bool aligned_model = (D.n_model & 3) == 0;
    for (int kk = threadIdx.x * 4; kk < D.n_aug_pad32; kk += RO_THREADS * 4) {
        double a[4];
        if (kk + 3 < D.n_model) { a[0] = lm[kk]; a[1] = lm[kk + 1]; a[2] = lm[kk + 2]; a[3] = lm[kk + 3]; }
        else if (aligned_model && kk >= D.n_model && kk + 3 < D.n_aug) {
            const double2 x0 = *reinterpret_cast<const double2 *>(x + (kk - D.n_model)), x1 = *reinterpret_cast<const double2 *>(x + (kk - D.n_model) + 2);
            a[0] = x0.x; a[1] = x0.y * x0.y; a[2] = x1.x; a[3] = x1.y * x1.y;      // 0-based odd entry == the reference's even (1-based) entry
        } else {
            auto aug = [&](int i) -> double {
                if (i < D.n_model) return lm[i];
                if (i >= D.n_aug) return 0.0;
                const int j = i - D.n_model;
                const double v = x[j];
                return (j & 1) ? v * v : v;
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = aug(kk + q);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(wrow[r] + kk));
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r] += (double)w[q] * a[q];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) red[wave][r] = s;
    }
    __syncthreads();
    if (threadIdx.x < R) {
        const int row = r0 + threadIdx.x;
        if (row < D.n_out) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < RO_THREADS / 64; ++w) v += red[w][threadIdx.x];
            if (!(flags & 1)) {
                const int si = D.out_stat[row];
                if (si >= 0) v = __dadd_rn(__dmul_rn(v, D.stdv[si]), D.mean[si]);
            }
            out_all[(size_t)res * out_stride + row] = v;
        }
    }
}

// The same product as k_readout<R, 64, .>, as a PERSISTENT kernel with a bounded footprint, for the pipelined hybrid step
// (hybrid.py): the readout then runs on a side stream underneath the SPEEDY window, whose small latency-bound kernels
// must keep finding free wave slots and LDS.  A plain launch floods every CU with single-wave workgroups and a 512-thread
// transform workgroup is starved until the whole readout has drained (measured: one k_grid launch waited 1.26 ms).  Here
// one workgroup of WAVES waves is launched per CU with a capped register budget, every wave is an independent worker
// that pulls (reservoir, row group) items from a global counter until none are left (so every wave terminates), and
// the rest of the CU -- wave slots, half of the VGPR file, all of the LDS -- stays free for the other stream.  The
// bounded kernel alone cannot keep HBM busy (few loads in flight), so when the SPEEDY window is over a second launch at
// full occupancy (flags bit 4, "drain") pulls from the SAME counter until the queue is empty.
template <int R, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_readout_persist(const ResDesc *__restrict__ descs, int res_begin, int res_end, int parts,
                                                                 const double *__restrict__ lm_all, int lm_stride,
                                                                 double *__restrict__ out_all, int out_stride, int cur, int flags,
                                                                 double *__restrict__ partial_all, unsigned *__restrict__ counter,
                                                                 unsigned total_items)
{
    // MINW (minimum waves per SIMD the compiler must allow) caps the VGPRs: MINW = 4 -> 128 of the SIMD's 512 per wave
    const int lane = threadIdx.x & 63;
    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);      // wave-uniform (SGPR): descriptor and row bases stay scalar
        if (item >= total_items) return;
        int res, grp;
        decode_block((int)item, parts, res_begin, res, grp);
        if (res >= res_end) continue;
        const ResDesc &D = descs[res];   // (by reference: a copy with x[cur] indexed at run time lives in scratch, 136 B per lane)
        if (!D.loaded) continue;
        const int r0 = grp * R;
        if (r0 >= D.n_out) continue;
        const double *__restrict__ x = cur ? D.x[1] : D.x[0];
        const double *__restrict__ lm = lm_all + (size_t)res * lm_stride;
        const size_t ld = (size_t)D.n_aug_pad;
        const double *wrow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wrow[r] = D.wout + (size_t)min(r0 + r, D.n_out - 1) * ld;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        const bool aligned_model = (D.n_model & 1) == 0;
        const int csplit = (D.n_model + 1) & ~1;
        const int kk_begin = (flags & 2) ? csplit : 0, kk_end = (flags & 4) ? csplit : D.n_aug_pad;
        for (int kk = kk_begin + lane * 2; kk < kk_end; kk += 128) {         // the loop of k_readout<R, 64, true>
            double a0, a1;
            if (kk + 1 < D.n_model) { a0 = lm[kk]; a1 = lm[kk + 1]; }
            else if (aligned_model && kk >= D.n_model && kk + 1 < D.n_aug) {
                const double2 xv = *reinterpret_cast<const double2 *>(x + (kk - D.n_model));
                a0 = xv.x; a1 = xv.y * xv.y;
            } else {
                auto aug = [&](int i) -> double {
                    if (i < D.n_model) return lm[i];
                    if (i >= D.n_aug) return 0.0;
                    const int j = i - D.n_model;
                    const double v = x[j];
                    return (j & 1) ? v * v : v;
                };
                a0 = aug(kk); a1 = aug(kk + 1);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const f64x2 w = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(wrow[r] + kk));
                acc[r] += w[0] * a0;
                acc[r] += w[1] * a1;
            }
        }
        double mine = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double sum = wave_sum(acc[r]);
            if (lane == r) mine = sum;
        }
        if (lane < R) {
            const int row = r0 + lane;
            if (row < D.n_out) {
                double v = mine;
                if (flags & 2) { partial_all[(size_t)res * out_stride + row] = v; }
                else {
                    if (flags & 4) v = partial_all[(size_t)res * out_stride + row] + v;
                    if (!(flags & 1)) {
                        const int si = D.out_stat[row];
                        if (si >= 0) v = __dadd_rn(__dmul_rn(v, D.stdv[si]), D.mean[si]);
                    }
                    out_all[(size_t)res * out_stride + row] = v;
                }
            }
        }
    }
}

}  // namespace


namespace {

int sync_descs(sml_bank *b)
{
    return sml::bank_sync_descs(b);
}

}  // namespace

int sml::bank_sync_descs(sml_bank *b)
{
    if (!b->descs_dirty) return SML_OK;
    std::vector<ResDesc> h(b->capacity);
    for (int i = 0; i < b->capacity; ++i) h[i] = b->res[i].desc;
    // the compact copies are used when EVERY loaded reservoir of the bank has them (one kernel shape per launch)
    int loaded = 0, with32 = 0;
    for (int i = 0; i < b->capacity; ++i)
        if (h[i].loaded) { ++loaded; with32 += (h[i].wout32 && h[i].sell_val32) ? 1 : 0; }
    b->compact = (b->allow_compact && loaded > 0 && with32 == loaded) ? 1 : 0;
    SML_HIP(hipMemcpy(b->d_descs, h.data(), sizeof(ResDesc) * b->capacity, hipMemcpyHostToDevice));
    b->descs_dirty = false;
    return SML_OK;
}

namespace {

void free_slot(HostRes &r)
{
    for (void *p : r.allocs) (void)hipFree(p);
    r.allocs.clear();
    r.desc = ResDesc{};
    r.wout32_alloc = nullptr;
}

template <class T>
int upload(HostRes &r, const T **dst, const std::vector<T> &src)
{
    T *p = nullptr;
    int rc = sml::dev_upload(&p, src.data(), src.size());
    if (rc) return rc;
    r.allocs.push_back(p);
    *dst = p;
    return SML_OK;
}

int load_common(sml_bank *bank, int slot, int n, int d, int k, int n_model, int n_out,
                const int32_t *rows, const int32_t *cols, const double *vals,
                const std::vector<int> &wr, const std::vector<int> &wc, const std::vector<double> &wv,
                const double *wout, double leakage, const double *mean, const double *stdv, int nstat,
                const int32_t *out_stat_idx)
{
    SML_REQUIRE(bank, "sml_bank_load: null bank");
    SML_REQUIRE(slot >= 0 && slot < bank->capacity, "sml_bank_load: slot %d out of range [0,%d)", slot, bank->capacity);
    SML_REQUIRE(n > 0 && d > 0 && k >= 0 && n_model >= 0 && n_out > 0, "sml_bank_load: bad sizes n=%d d=%d k=%d n_model=%d n_out=%d", n, d, k, n_model, n_out);
    SML_REQUIRE(d <= bank->max_d && n_model <= bank->max_n_model && n_out <= bank->max_n_out,
                "sml_bank_load: d/n_model/n_out (%d,%d,%d) exceed the bank strides (%d,%d,%d)", d, n_model, n_out, bank->max_d, bank->max_n_model, bank->max_n_out);
    SML_REQUIRE(rows && cols && vals && wout && mean && stdv && nstat > 0, "sml_bank_load: null array");
    for (int e = 0; e < k; ++e)
        SML_REQUIRE(rows[e] >= 1 && rows[e] <= n && cols[e] >= 1 && cols[e] <= n, "sml_bank_load: COO entry %d = (%d,%d) outside 1..%d", e, rows[e], cols[e], n);
    for (int i = 0; i < n_out && out_stat_idx; ++i)
        SML_REQUIRE(out_stat_idx[i] < nstat, "sml_bank_load: out_stat_idx[%d]=%d >= nstat=%d", i, out_stat_idx[i], nstat);

    HostRes &R = bank->res[slot];
    free_slot(R);

    // ---- fused operator [A | Win] as CSR (A entries in COO storage order, then Win), then SELL-64 ----
    std::vector<int> cnt(n, 0);
    for (int e = 0; e < k; ++e) cnt[rows[e] - 1]++;
    for (size_t e = 0; e < wr.size(); ++e) cnt[wr[e]]++;
    std::vector<int> ptr(n + 1, 0);
    for (int i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + cnt[i];
    const int nnz = ptr[n];
    std::vector<int> ccol(nnz);
    std::vector<double> cval(nnz);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int e = 0; e < k; ++e) { int p = fill[rows[e] - 1]++; ccol[p] = cols[e] - 1; cval[p] = vals[e]; }
    for (size_t e = 0; e < wr.size(); ++e) { int p = fill[wr[e]]++; ccol[p] = n + wc[e]; cval[p] = wv[e]; }

    // Device order of the state vector: rows sorted by length (SELL), but with the even-indexed and odd-indexed rows
    // sorted separately and interleaved, so that position p holds a row of the same parity as p.  The state lives in HBM
    // in THIS order: k_update stores x_new contiguously (scattered 8-byte stores cost a 64-byte write each: measured
    // WRITE_SIZE 0.35 GB per sweep instead of 0.05 GB), the readout's "square the odd entries" rule stays positional, and
    // W_out's state columns are permuted once at load.  inv[row] = position.
    std::vector<int> order(n), ev, od;
    for (int i = 0; i < n; ++i) (i & 1 ? od : ev).push_back(i);
    auto by_len = [&](int a, int b) { return cnt[a] > cnt[b]; };
    std::stable_sort(ev.begin(), ev.end(), by_len);
    std::stable_sort(od.begin(), od.end(), by_len);
    for (int p = 0; p < n; ++p) order[p] = (p & 1) ? od[p >> 1] : ev[p >> 1];
    std::vector<int> inv(n);
    for (int p = 0; p < n; ++p) inv[order[p]] = p;
    const int nslices = (n + 63) / 64;
    SML_REQUIRE(n + d <= 65535, "sml_bank_load: n + d = %d exceeds the 16-bit column index of the device layout", n + d);
    for (int i = 0; i < n; ++i) SML_REQUIRE(cnt[i] <= 255, "sml_bank_load: row %d has %d nonzeros (> 255)", i, cnt[i]);
    std::vector<int> slice_off(nslices + 1, 0), perm(nslices * 64, -1);
    std::vector<unsigned char> row_len(nslices * 64, 0);
    for (int s = 0; s < nslices; ++s) {
        int width = 0;
        for (int l = 0; l < 64; ++l) {
            const int pos = s * 64 + l;
            if (pos < n) { perm[pos] = order[pos]; row_len[pos] = (unsigned char)cnt[order[pos]]; width = std::max(width, cnt[order[pos]]); }
        }
        slice_off[s + 1] = slice_off[s] + width * 64;
    }
    std::vector<unsigned short> scol(slice_off[nslices], 0);
    std::vector<double> sval(slice_off[nslices], 0.0);
    for (int s = 0; s < nslices; ++s)
        for (int l = 0; l < 64; ++l) {
            const int pos = s * 64 + l;
            if (pos >= n) continue;
            const int r = order[pos];
            for (int j = 0; j < cnt[r]; ++j) {
                const int c = ccol[ptr[r] + j];
                scol[slice_off[s] + j * 64 + l] = (unsigned short)(c < n ? inv[c] : c);           // state columns refer to device positions
                sval[slice_off[s] + j * 64 + l] = cval[ptr[r] + j];
            }
        }

    // ---- W_out: (n_out, n_aug) column-major -> [n_out][n_aug_pad] row-major ----
    // Row stride padded to 16 doubles = one 128-byte line: with the even padding of before (5892 -> 47136 B = 368.25 lines) every
    // 1-KB chunk a wavefront reads from a row straddled 9 lines instead of 8 -- PMC: 69.7 M L2 requests and 65.4 M misses per sweep
    // against 58.8 M lines of W_out, the 12.5 % that separated the measured 8.39 GB of HBM traffic from the 7.53 GB algorithmic.
    const int n_aug = n + n_model, n_aug_pad = (n_aug + 15) & ~15;
    std::vector<double> wrm((size_t)n_out * n_aug_pad, 0.0);
    for (int j = 0; j < n_aug; ++j) {
        const int jd = j < n_model ? j : n_model + inv[j - n_model];             // state columns follow the device order
        for (int i = 0; i < n_out; ++i) wrm[(size_t)i * n_aug_pad + jd] = wout[(size_t)j * n_out + i];
    }
    R.order = order;

    std::vector<double> hmean(mean, mean + nstat), hstd(stdv, stdv + nstat), zeros(n, 0.0);
    std::vector<int> hstat(n_out, -1);
    if (out_stat_idx) hstat.assign(out_stat_idx, out_stat_idx + n_out);

    ResDesc D{};
    D.n = n; D.d = d; D.n_model = n_model; D.n_out = n_out; D.n_aug = n_aug; D.n_aug_pad = n_aug_pad; D.nslices = nslices;
    D.leak = leakage;
    int rc;
    if ((rc = upload(R, &D.slice_off, slice_off))) return rc;
    if ((rc = upload(R, &D.sell_col, scol))) return rc;
    if ((rc = upload(R, &D.sell_val, sval))) return rc;
    if ((rc = upload(R, &D.perm, perm))) return rc;
    if ((rc = upload(R, &D.row_len, row_len))) return rc;
    if ((rc = upload(R, &D.wout, wrm))) return rc;
    {
        // compact copies when nothing is lost (SML_BANK_COMPACT=0: never): W_out with rows padded to 32 floats, the operator's values as they are
        static const bool want = !(getenv("SML_BANK_COMPACT") && atoi(getenv("SML_BANK_COMPACT")) == 0);
        bool exact = want;
        for (size_t i = 0; i < wrm.size() && exact; ++i) exact = (double)(float)wrm[i] == wrm[i];
        for (size_t i = 0; i < sval.size() && exact; ++i) exact = (double)(float)sval[i] == sval[i];
        if (exact) {
            const int pad32 = (n_aug + 31) & ~31;
            std::vector<float> w32((size_t)n_out * pad32, 0.f), v32(sval.size());
            for (int i = 0; i < n_out; ++i)
                for (int j = 0; j < n_aug; ++j) w32[(size_t)i * pad32 + j] = (float)wrm[(size_t)i * n_aug_pad + j];
            for (size_t i = 0; i < sval.size(); ++i) v32[i] = (float)sval[i];
            D.n_aug_pad32 = pad32;
            if ((rc = upload(R, &D.wout32, w32))) return rc;
            R.wout32_alloc = const_cast<float *>(D.wout32);
            if ((rc = upload(R, &D.sell_val32, v32))) return rc;
        }
    }
    if ((rc = upload(R, &D.mean, hmean))) return rc;
    if ((rc = upload(R, &D.stdv, hstd))) return rc;
    if ((rc = upload(R, &D.out_stat, hstat))) return rc;
    const double *x0 = nullptr, *x1 = nullptr;
    if ((rc = upload(R, &x0, zeros))) return rc;
    if ((rc = upload(R, &x1, zeros))) return rc;
    D.x[0] = const_cast<double *>(x0); D.x[1] = const_cast<double *>(x1);
    D.loaded = 1;
    R.desc = D;
    // algorithmic bytes: nonzeros*(8+4) + slice table + x read + x write + u ; W_out + x~ + outvec
    // A as CSR k*(8+4)+(n+1)*4 ; x read+write 2*n*8 ; W_in nonzeros 8 B each ; u d*8
    R.update_bytes = (uint64_t)k * 12 + (uint64_t)(n + 1) * 4 + (uint64_t)n * 16 + (uint64_t)wr.size() * 8 + (uint64_t)d * 8;
    // W_out n_out*n_aug*8 ; local_model + outvec + mean/std
    R.readout_bytes = (uint64_t)n_out * n_aug * 8 + (uint64_t)(n_model + n_out + 2 * nstat) * 8;
    R.update_bytes32 = R.update_bytes - ((uint64_t)k + (uint64_t)wr.size()) * 4;
    R.readout_bytes32 = R.readout_bytes - (uint64_t)n_out * n_aug * 4;
    bank->max_nd = std::max(bank->max_nd, n + d);
    bank->max_n_out_loaded = std::max(bank->max_n_out_loaded, n_out);
    bank->descs_dirty = true;
    return SML_OK;
}

// host <-> device state: the host sees the reference's row order, HBM holds the device order (HostRes::order)
int upload_state(const HostRes &R, double *dst, const double *x_host)
{
    std::vector<double> tmp(R.order.size());
    for (size_t p = 0; p < tmp.size(); ++p) tmp[p] = x_host[R.order[p]];
    SML_HIP(hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    return SML_OK;
}

int download_state(const HostRes &R, const double *src, double *x_host)
{
    std::vector<double> tmp(R.order.size());
    SML_HIP(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t p = 0; p < tmp.size(); ++p) x_host[R.order[p]] = tmp[p];
    return SML_OK;
}

int launch_update(sml_bank *b, int res_begin, int res_end, const double *u_all, hipStream_t st, int square_input = 0,
                  const TrainSlot *train_slots = nullptr, int train_col = 0, int parts_cap = 4)
{
    // 512-thread workgroups: three per CU (3 x 50.7 KB LDS, 24 waves).  A workgroup's life is latency-bound (29 us for a whole
    // reservoir on an idle chip: staging + ~11 dependent slice batches per wave) and staging [x ; u] costs 18 us per copy
    // over the whole bank, so: one workgroup per reservoir for every full round of the chip's resident slots, and the
    // ragged remainder cut into slice ranges.  Measured (1152 reservoirs, ms): 512 x 2 parts everywhere 0.142, 512 x 1
    // everywhere 0.132 (its second round is half empty), 3..8 parts 0.164..0.265, 1024 x 1 0.164, 256 x 4 0.18.
    static const int cfg = getenv("SML_UPD_CFG") ? atoi(getenv("SML_UPD_CFG")) : -1;
    const int nres8 = ((res_end - res_begin + 7) / 8) * 8;
    if (!b->ncu) {
        int dev = 0;
        SML_HIP(hipGetDevice(&dev));
        SML_HIP(hipDeviceGetAttribute(&b->ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int slots = 3 * b->ncu;
    int threads = 512, parts = 2, whole = 0;
    if (cfg == 1) { threads = 1024; parts = 1; }
    else if (cfg == 2) { threads = 256; parts = 4; }
    else if (cfg == 9) { parts = 1; whole = nres8; }
    else if (cfg >= 3) parts = cfg;
    else if (cfg < 0) {
        whole = (nres8 / slots) * slots;                     // multiples of 8 as long as the CU count is
        whole -= whole % 8;
        const int rem = nres8 - whole;
        parts = rem ? std::max(1, std::min(parts_cap, slots / rem)) : 1;
    }
    const int nblocks = whole + (nres8 - whole) * parts;
    const size_t lds = (size_t)b->max_nd * sizeof(double);
    SML_REQUIRE(lds <= 160 * 1024, "reservoir too large for the LDS-staged update (n+d=%d)", b->max_nd);
    static bool attr_set = false;                            // (one process drives one GPU: the attribute is set once per process)
    if (!attr_set) {
        SML_HIP(hipFuncSetAttribute((const void *)k_update<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SML_HIP(hipFuncSetAttribute((const void *)k_update<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SML_HIP(hipFuncSetAttribute((const void *)k_update<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SML_HIP(hipFuncSetAttribute((const void *)k_update<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed_u = b->timing && b->timing_update;
    if (timed_u) { SML_HIP(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence)); SML_HIP(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence)); SML_HIP(hipEventRecord(e0, st)); }
    if (threads == 1024)
        hipLaunchKernelGGL(k_update<1024>, dim3(nblocks), dim3(1024), lds, st, b->d_descs, res_begin, res_end, parts, u_all, b->max_d, b->cur, square_input, whole, train_slots, train_col);
    else if (threads == 256)
        hipLaunchKernelGGL(k_update<256>, dim3(nblocks), dim3(256), lds, st, b->d_descs, res_begin, res_end, parts, u_all, b->max_d, b->cur, square_input, whole, train_slots, train_col);
    else if (b->compact == 1)          // every loaded reservoir's values are floats: the compact copies (6 B per nonzero)
        hipLaunchKernelGGL((k_update<512, true>), dim3(nblocks), dim3(512), lds, st, b->d_descs, res_begin, res_end, parts, u_all, b->max_d, b->cur, square_input, whole, train_slots, train_col);
    else
        hipLaunchKernelGGL(k_update<512>, dim3(nblocks), dim3(512), lds, st, b->d_descs, res_begin, res_end, parts, u_all, b->max_d, b->cur, square_input, whole, train_slots, train_col);
    SML_HIP(hipGetLastError());
    if (timed_u) { SML_HIP(hipEventRecord(e1, st)); b->ev_update.emplace_back(e0, e1); }
    b->cur ^= 1;
    return SML_OK;
}

// v_p = matmul(wout(:, 1:chunk_size_speedy), local_model) (src/mod_reservoir.f90:1459): the physics-model columns of the readout alone,
// standardised like v_ml.  132 columns x 136 rows per reservoir: one wavefront per output row, diagnostics only.
__global__ __launch_bounds__(64) void k_vp(const ResDesc *__restrict__ descs, const double *__restrict__ lm_all, int lm_stride, double *__restrict__ vp, int out_stride)
{
    const ResDesc &D = descs[blockIdx.y];
    const int o = blockIdx.x;
    if (!D.loaded || o >= D.n_out) return;
    const double *w = D.wout + (size_t)o * D.n_aug_pad, *lm = lm_all + (size_t)blockIdx.y * lm_stride;
    double acc = 0.0;
    for (int c = threadIdx.x; c < D.n_model; c += 64) acc += w[c] * lm[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) vp[(size_t)blockIdx.y * out_stride + o] = acc;
}

int launch_readout(sml_bank *b, int res_begin, int res_end, int flags, hipStream_t st)
{
    // <rows per workgroup, threads>, non-temporal 16-byte loads.  Round 3 (profiles/micro/sweep_readout_variants.sh, event-timed, ms at
    // 1152 / 576 / 288 / 144 resident reservoirs): <4,512> 1.071 / 0.550 / 0.283 / 0.148 -- the default at every size, so a rank of
    // an N-GPU run sums its rows exactly as the single-GPU run does -- against <17,64> 1.141-1.157 / 0.614 / 0.307 / 0.163 (the default
    // of rounds 1-2), <8,64> 1.143-1.152 / 0.600 / 0.308 / 0.157, <4,1024> 1.077 / 0.560 / 0.284 / 0.149, <8,512> 1.110, <17,512> 1.104,
    // <8,256> 1.117, <4,256> 1.107, <4,384> 1.079, <4,768> 1.083, <6,512> 1.082, <2,512> 1.142, <2,1024> 1.278, <17,1024> 1.167,
    // <34,256> 1.276; plain instead of non-temporal loads <4,512> 1.186.  A read-only stream of the same 7.5 GB reaches 6.85-7.15 TB/s
    // on this part (profiles/micro/read_bw_ceiling.hip); <4,512> is at 7.03.
    // (Round 1 found the opposite order -- one wavefront per workgroup best, <17,256> 1.58 ms -- while every workgroup still copied
    //  the reservoir descriptor through scratch and the rows started 32 bytes off a cache line; both were fixed in round 2 without
    //  the sweep being repeated.)
    static const int forced = getenv("SML_RO_VARIANT") ? atoi(getenv("SML_RO_VARIANT")) : -1;
    const int nres8 = ((res_end - res_begin + 7) / 8) * 8;
    const int variant = forced >= 0 ? forced : 18;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = b->timing && !(flags & 4);       // the small physics-model block (part 2) is not the roofline kernel
    if (flags & (8 | 16)) {
        // persistent work-queue variants (sml_bank_readout_part): bit 3 = bounded footprint, resets the queue;
        // bit 4 = full-occupancy drain of the same queue (no reset)
        if (!b->ncu) {
            int dev = 0;
            SML_HIP(hipGetDevice(&dev));
            SML_HIP(hipDeviceGetAttribute(&b->ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const int ncu = b->ncu;
        if (!b->d_counter) { SML_HIP(hipMalloc((void **)&b->d_counter, 256)); SML_HIP(hipMemset(b->d_counter, 0xff, 256)); }
        const int parts = (b->max_n_out_loaded + 17 - 1) / 17;
        const unsigned total = (unsigned)(nres8 * parts);
        if (flags & 8) SML_HIP(hipMemsetAsync(b->d_counter, 0, sizeof(unsigned), st));
        if (timed) { SML_HIP(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence)); SML_HIP(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence)); SML_HIP(hipEventRecord(e0, st)); }
#define ROP_LAUNCH(W, MINW, NWG)                                                                                                     \
        hipLaunchKernelGGL((k_readout_persist<17, W, MINW>), dim3(NWG), dim3((W) * 64), 0, st, b->d_descs, res_begin, res_end, parts, \
                           b->d_local_model, b->max_n_model, b->d_outvec, b->max_n_out, b->cur, flags & 7, b->d_partial, b->d_counter, total)
        // measured in the pipelined hybrid step (ms per step; sequential schedule 2.88): bounded kernel of 4 / 8 / 12 / 16
        // waves per CU + drain 2.91 / 2.72 / 3.21 / 3.29; 8 waves without the drain 2.67
        if (flags & 16) { ROP_LAUNCH(4, 5, ncu * 5); }               // 20 waves per CU, <= 102 VGPRs: the occupancy of k_readout<17,64>
        else { ROP_LAUNCH(8, 4, ncu); }
#undef ROP_LAUNCH
        SML_HIP(hipGetLastError());
        if (timed) { SML_HIP(hipEventRecord(e1, st)); b->ev_readout.emplace_back(e0, e1); }
        return SML_OK;
    }
    if (timed) { SML_HIP(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence)); SML_HIP(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence)); SML_HIP(hipEventRecord(e0, st)); }
    if (b->compact == 1 && !(flags & 6) && forced < 0) {
        // every loaded reservoir's W_out is floats: the compact copy, the whole product (the column split keeps the 8-byte kernel)
        static const int rows32 = getenv("SML_RO32_ROWS") ? atoi(getenv("SML_RO32_ROWS")) : 8;
        static const int thr32 = getenv("SML_RO32_THREADS") ? atoi(getenv("SML_RO32_THREADS")) : 256;
#define RO32_LAUNCH(R, T)                                                                                                  \
        {                                                                                                                  \
            const int parts = (b->max_n_out_loaded + (R) - 1) / (R);                                                       \
            hipLaunchKernelGGL((k_readout32<R, T>), dim3(nres8 * parts), dim3(T), 0, st, b->d_descs, res_begin, res_end, parts, \
                               b->d_local_model, b->max_n_model, b->d_outvec, b->max_n_out, b->cur, flags);                \
        }
        if (rows32 == 8 && thr32 == 512) RO32_LAUNCH(8, 512)
        else if (rows32 == 8 && thr32 == 256) RO32_LAUNCH(8, 256)
        else if (rows32 == 4 && thr32 == 256) RO32_LAUNCH(4, 256)
        else if (rows32 == 4 && thr32 == 1024) RO32_LAUNCH(4, 1024)
        else if (rows32 == 2 && thr32 == 512) RO32_LAUNCH(2, 512)
        else if (rows32 == 17 && thr32 == 256) RO32_LAUNCH(17, 256)
        else if (rows32 == 16 && thr32 == 256) RO32_LAUNCH(16, 256)
        else if (rows32 == 12 && thr32 == 256) RO32_LAUNCH(12, 256)
        else if (rows32 == 6 && thr32 == 256) RO32_LAUNCH(6, 256)
        else if (rows32 == 8 && thr32 == 128) RO32_LAUNCH(8, 128)
        else if (rows32 == 17 && thr32 == 128) RO32_LAUNCH(17, 128)
        else if (rows32 == 8 && thr32 == 384) RO32_LAUNCH(8, 384)
        else if (rows32 == 4 && thr32 == 512) RO32_LAUNCH(4, 512)
        else RO32_LAUNCH(8, 256)
#undef RO32_LAUNCH
        SML_HIP(hipGetLastError());
        if (timed) { SML_HIP(hipEventRecord(e1, st)); b->ev_readout.emplace_back(e0, e1); }
        return SML_OK;
    }
#define RO_LAUNCH(R, T, NT)                                                                                             \
    {                                                                                                                   \
        const int parts = (b->max_n_out_loaded + (R) - 1) / (R);                                                        \
        hipLaunchKernelGGL((k_readout<R, T, NT>), dim3(nres8 * parts), dim3(T), 0, st, b->d_descs, res_begin, res_end, parts, \
                           b->d_local_model, b->max_n_model, b->d_outvec, b->max_n_out, b->cur, flags, b->d_partial);   \
    }
    switch (variant) {
    case 1: RO_LAUNCH(17, 512, true) break;
    case 2: RO_LAUNCH(34, 256, true) break;
    case 3: RO_LAUNCH(8, 256, true) break;
    case 4: RO_LAUNCH(17, 128, true) break;
    case 5: RO_LAUNCH(17, 256, false) break;
    case 6: RO_LAUNCH(34, 512, true) break;
    case 7: RO_LAUNCH(17, 64, true) break;
    case 8: RO_LAUNCH(8, 128, true) break;
    case 9: RO_LAUNCH(34, 128, true) break;
    case 10: RO_LAUNCH(17, 128, false) break;
    case 11: RO_LAUNCH(8, 64, true) break;
    case 12: RO_LAUNCH(34, 64, true) break;
    case 13: RO_LAUNCH(17, 256, true) break;
    case 14: RO_LAUNCH(4, 64, true) break;
    case 15: RO_LAUNCH(8, 64, false) break;
    case 16: RO_LAUNCH(6, 64, true) break;
    case 17: RO_LAUNCH(8, 512, true) break;
    case 18: RO_LAUNCH(4, 512, true) break;
    case 19: RO_LAUNCH(4, 1024, true) break;
    case 20: RO_LAUNCH(8, 1024, true) break;
    case 21: RO_LAUNCH(17, 1024, true) break;
    case 22: RO_LAUNCH(4, 256, true) break;
    case 23: RO_LAUNCH(2, 1024, true) break;
    case 24: RO_LAUNCH(4, 768, true) break;
    case 25: RO_LAUNCH(6, 512, true) break;
    case 26: RO_LAUNCH(2, 512, true) break;
    case 27: RO_LAUNCH(4, 384, true) break;
    case 28: RO_LAUNCH(8, 384, true) break;
    case 29: RO_LAUNCH(4, 512, false) break;
    case 0: RO_LAUNCH(RO_ROWS, 256, true) break;
    default: RO_LAUNCH(RO_ROWS, 64, true) break;
    }
#undef RO_LAUNCH
    SML_HIP(hipGetLastError());
    if (timed) { SML_HIP(hipEventRecord(e1, st)); b->ev_readout.emplace_back(e0, e1); }
    return SML_OK;
}

}  // namespace

// ---- CU-masked streams and process exit ----
// A process that ends while a CU-masked stream (hipExtStreamCreateWithCUMask) is still alive crashes inside the runtime's own
// finalisation when rocprofv3 is attached (SIGSEGV in __cxa_finalize after the tool's finalisation, ROCm 7.2; seen with the ridge
// solver's trailing stream).  Every masked stream of the library is therefore created through this registry, and the first one
// registers ONE exit handler -- inside the library, so that every host is covered, a Fortran program as much as Python -- that
// releases the ridge solver's workspace and destroys whatever masked stream its owner left behind.  The handler is registered
// after the HIP runtime came up (a stream was just created), so at exit it runs before the runtime's own handlers.
namespace sml {
namespace {
std::vector<hipStream_t> &masked_streams()
{
    static std::vector<hipStream_t> *v = new std::vector<hipStream_t>;      // (never destroyed: the exit handler may run late)
    return *v;
}
std::mutex &registry_mutex()        // the C-ABI may be called from several host threads: the registry and sml_set_device's choice are shared
{
    static std::mutex *m = new std::mutex;
    return *m;
}

void exit_cleanup()
{
    (void)sml_train_release_workspace();
    std::vector<hipStream_t> left;
    { std::lock_guard<std::mutex> lk(registry_mutex()); left.swap(masked_streams()); }
    for (hipStream_t st : left) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
}
}  // namespace

int masked_stream_create(hipStream_t *out, const uint32_t *mask, int nwords)
{
    static std::once_flag registered;
    SML_HIP(hipExtStreamCreateWithCUMask(out, (uint32_t)nwords, mask));
    { std::lock_guard<std::mutex> lk(registry_mutex()); masked_streams().push_back(*out); }
    std::call_once(registered, [] { atexit(exit_cleanup); });
    return SML_OK;
}

int masked_stream_destroy(hipStream_t st)
{
    if (!st) return SML_OK;
    {
        std::lock_guard<std::mutex> lk(registry_mutex());
        auto &v = masked_streams();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == st) { v.erase(v.begin() + i); break; }
    }
    SML_HIP(hipStreamDestroy(st));
    return SML_OK;
}
}  // namespace sml

extern "C" {

const char *sml_last_error(void) { return sml::last_error_ref().c_str(); }
int sml_version(void) { return 100; }

int sml_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sml_set_device(int ordinal)
{
    // One process drives ONE GPU (one rank per GPU): the library's cached streams, launch attributes and solver workspaces belong to
    // the device that was current when they were made.  A second device in the same process is refused rather than served wrongly.
    static int chosen = -1;
    std::lock_guard<std::mutex> lk(sml::registry_mutex());
    if (chosen >= 0 && chosen != ordinal)
        return sml::fail(SML_ERR_STATE, "sml_set_device(%d): this process already drives device %d (one process per GPU)", ordinal, chosen);
    SML_HIP(hipSetDevice(ordinal));
    chosen = ordinal;
    return SML_OK;
}

int sml_device_synchronize(void)
{
    SML_HIP(hipDeviceSynchronize());
    return SML_OK;
}

int sml_stream_create_cu_mask(const uint32_t *mask, int nwords, void **stream_out)
{
    SML_REQUIRE(mask && nwords > 0 && stream_out, "sml_stream_create_cu_mask: bad arguments");
    hipStream_t st = nullptr;
    int rc = sml::masked_stream_create(&st, mask, nwords);
    if (rc) return rc;
    *stream_out = (void *)st;
    return SML_OK;
}

int sml_stream_destroy(void *stream)
{
    return sml::masked_stream_destroy((hipStream_t)stream);
}

// device memory for hosts that have no HIP binding of their own (the Fortran drop-ins): plain hipMalloc / hipMemcpy

int sml_dev_alloc(uint64_t bytes, void **out_dev)
{
    SML_REQUIRE(out_dev, "sml_dev_alloc: null pointer");
    SML_HIP(hipMalloc(out_dev, bytes ? bytes : 16));
    return SML_OK;
}

int sml_dev_free(void *dev)
{
    if (dev) SML_HIP(hipFree(dev));
    return SML_OK;
}

int sml_dev_zero(void *dev, uint64_t bytes)
{
    SML_REQUIRE(dev || !bytes, "sml_dev_zero: null pointer");
    if (bytes) SML_HIP(hipMemset(dev, 0, bytes));
    return SML_OK;
}

int sml_dev_upload(void *dst_dev, const void *src_host, uint64_t bytes)
{
    SML_REQUIRE((dst_dev && src_host) || !bytes, "sml_dev_upload: null pointer");
    if (bytes) SML_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_dev_download(void *dst_host, const void *src_dev, uint64_t bytes)
{
    SML_REQUIRE((dst_host && src_dev) || !bytes, "sml_dev_download: null pointer");
    if (bytes) SML_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return SML_OK;
}

int sml_bank_create(int capacity, int max_d, int max_n_model, int max_n_out, sml_bank **out)
{
    SML_REQUIRE(out && capacity > 0 && max_d > 0 && max_n_model >= 0 && max_n_out > 0, "sml_bank_create: bad arguments");
    sml_bank *b = new sml_bank;
    b->capacity = capacity; b->max_d = max_d; b->max_n_model = std::max(max_n_model, 1); b->max_n_out = max_n_out;
    b->res.resize(capacity);
    int rc;
    if ((rc = sml::dev_zeros(&b->d_descs, (size_t)capacity)) || (rc = sml::dev_zeros(&b->d_feedback, (size_t)capacity * b->max_d)) ||
        (rc = sml::dev_zeros(&b->d_local_model, (size_t)capacity * b->max_n_model)) ||
        (rc = sml::dev_zeros(&b->d_outvec, (size_t)capacity * b->max_n_out)) ||
        (rc = sml::dev_zeros(&b->d_partial, (size_t)capacity * b->max_n_out))) {
        delete b;
        return rc;
    }
    *out = b;
    return SML_OK;
}

int sml_bank_destroy(sml_bank *b)
{
    if (!b) return SML_OK;
    for (auto &r : b->res) free_slot(r);
    (void)hipFree(b->d_descs); (void)hipFree(b->d_feedback); (void)hipFree(b->d_local_model); (void)hipFree(b->d_outvec); (void)hipFree(b->d_partial); if (b->d_vp) (void)hipFree(b->d_vp);
    if (b->d_counter) (void)hipFree(b->d_counter);
    for (auto &t : b->train_states)
        if (t.first) (void)hipFree(t.first);
    delete b;
    return SML_OK;
}

int sml_bank_load(sml_bank *bank, int slot, int n, int d, int k, int n_model, int n_out,
                  const int32_t *rows, const int32_t *cols, const double *vals,
                  const double *win, const double *wout, double leakage,
                  const double *mean, const double *stdv, int nstat, const int32_t *out_stat_idx)
{
    SML_REQUIRE(win, "sml_bank_load: null win");
    SML_REQUIRE(n > 0 && d > 0, "sml_bank_load: bad sizes");
    // keep the nonzeros of the dense (n,d) column-major W_in, column by column (src/mod_reservoir.f90:272-280)
    std::vector<int> wr, wc;
    std::vector<double> wv;
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < n; ++i) {
            const double v = win[(size_t)j * n + i];
            if (v != 0.0) { wr.push_back(i); wc.push_back(j); wv.push_back(v); }
        }
    return load_common(bank, slot, n, d, k, n_model, n_out, rows, cols, vals, wr, wc, wv, wout, leakage, mean, stdv, nstat, out_stat_idx);
}

int sml_bank_load_sparse_win(sml_bank *bank, int slot, int n, int d, int k, int n_model, int n_out,
                             const int32_t *rows, const int32_t *cols, const double *vals,
                             int win_nnz, const int32_t *win_rows, const int32_t *win_cols, const double *win_vals,
                             const double *wout, double leakage,
                             const double *mean, const double *stdv, int nstat, const int32_t *out_stat_idx)
{
    SML_REQUIRE(win_nnz >= 0 && (win_nnz == 0 || (win_rows && win_cols && win_vals)), "sml_bank_load_sparse_win: null W_in triplets");
    std::vector<int> wr(win_nnz), wc(win_nnz);
    std::vector<double> wv(win_nnz);
    for (int e = 0; e < win_nnz; ++e) {
        SML_REQUIRE(win_rows[e] >= 1 && win_rows[e] <= n && win_cols[e] >= 1 && win_cols[e] <= d, "sml_bank_load_sparse_win: W_in entry %d out of range", e);
        wr[e] = win_rows[e] - 1; wc[e] = win_cols[e] - 1; wv[e] = win_vals[e];
    }
    return load_common(bank, slot, n, d, k, n_model, n_out, rows, cols, vals, wr, wc, wv, wout, leakage, mean, stdv, nstat, out_stat_idx);
}

#define BANK_SLOT(bank, slot)                                                                                  \
    SML_REQUIRE(bank, "null bank");                                                                            \
    SML_REQUIRE(slot >= 0 && slot < bank->capacity, "slot %d out of range [0,%d)", slot, bank->capacity);      \
    if (!bank->res[slot].desc.loaded) return sml::fail(SML_ERR_STATE, "slot %d has no reservoir loaded", slot); \
    const ResDesc &D = bank->res[slot].desc

int sml_bank_set_wout(sml_bank *bank, int slot, const double *wout)
{
    BANK_SLOT(bank, slot);
    SML_REQUIRE(wout, "sml_bank_set_wout: null wout");
    // same re-layout as load_common: row-major, state columns in the device (SELL) order of the state vector
    const std::vector<int> &order = bank->res[slot].order;
    SML_REQUIRE((int)order.size() == D.n, "sml_bank_set_wout: slot %d has no state order", slot);
    std::vector<double> wrm((size_t)D.n_out * D.n_aug_pad, 0.0);
    for (int jd = 0; jd < D.n_aug; ++jd) {
        const int j = jd < D.n_model ? jd : D.n_model + order[jd - D.n_model];       // device column jd holds reference column j
        for (int i = 0; i < D.n_out; ++i) wrm[(size_t)i * D.n_aug_pad + jd] = wout[(size_t)j * D.n_out + i];
    }
    SML_HIP(hipMemcpy(const_cast<double *>(D.wout), wrm.data(), wrm.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        // the compact copy follows the new weights: rewritten when they are all floats (and the operator's values have their copy),
        // withdrawn otherwise -- a trained W_out is arbitrary doubles until it has been through a weights file
        HostRes &R = bank->res[slot];
        ResDesc &W = R.desc;
        static const bool want = !(getenv("SML_BANK_COMPACT") && atoi(getenv("SML_BANK_COMPACT")) == 0);
        bool exact = want && W.sell_val32 != nullptr;
        for (size_t i = 0; i < wrm.size() && exact; ++i) exact = (double)(float)wrm[i] == wrm[i];
        if (exact) {
            const int pad32 = (W.n_aug + 31) & ~31;
            std::vector<float> w32((size_t)W.n_out * pad32, 0.f);
            for (int i = 0; i < W.n_out; ++i)
                for (int j = 0; j < W.n_aug; ++j) w32[(size_t)i * pad32 + j] = (float)wrm[(size_t)i * W.n_aug_pad + j];
            if (!R.wout32_alloc) {
                float *p32 = nullptr;
                SML_HIP(hipMalloc((void **)&p32, w32.size() * sizeof(float)));
                R.allocs.push_back(p32);
                R.wout32_alloc = p32;
            }
            SML_HIP(hipMemcpy(R.wout32_alloc, w32.data(), w32.size() * sizeof(float), hipMemcpyHostToDevice));
            W.wout32 = R.wout32_alloc; W.n_aug_pad32 = pad32;
        } else {
            W.wout32 = nullptr;
        }
        bank->descs_dirty = true;
    }
    return SML_OK;
}

int sml_bank_set_state(sml_bank *bank, int slot, const double *x)
{
    BANK_SLOT(bank, slot);
    return upload_state(bank->res[slot], D.x[bank->cur], x);
}

int sml_bank_get_state(sml_bank *bank, int slot, double *x)
{
    BANK_SLOT(bank, slot);
    return download_state(bank->res[slot], D.x[bank->cur], x);
}

int sml_bank_set_feedback(sml_bank *bank, int slot, const double *u)
{
    BANK_SLOT(bank, slot);
    SML_HIP(hipMemcpy(bank->d_feedback + (size_t)slot * bank->max_d, u, sizeof(double) * D.d, hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_bank_set_local_model(sml_bank *bank, int slot, const double *lm)
{
    BANK_SLOT(bank, slot);
    if (D.n_model) SML_HIP(hipMemcpy(bank->d_local_model + (size_t)slot * bank->max_n_model, lm, sizeof(double) * D.n_model, hipMemcpyHostToDevice));
    return SML_OK;
}

int sml_bank_get_outvec(sml_bank *bank, int slot, double *out)
{
    BANK_SLOT(bank, slot);
    SML_HIP(hipMemcpy(out, bank->d_outvec + (size_t)slot * bank->max_n_out, sizeof(double) * D.n_out, hipMemcpyDeviceToHost));
    return SML_OK;
}

/* predict's split readout (src/mod_reservoir.f90:1458-1461, outvec_component_contribs): after a predict of every slot, v_ml =
 * W_out[:, n_model:] x~ (the state block of the readout, its own launch over the state columns) and v_p = W_out[:, :n_model]
 * local_model, both standardised as in the reference (which un-standardises outvec only); v_p + v_ml = the readout before
 * unstandardize_state_vec_res up to the association of the column sum.  Call while local_model still holds the step's values. */
int sml_bank_outvec_contribs(sml_bank *b, void *stream)
{
    SML_REQUIRE(b, "sml_bank_outvec_contribs: null bank");
    int rc = sync_descs(b);
    if (rc) return rc;
    hipStream_t st = sml::as_stream(stream);
    for (auto &h : b->res)       // (the state block of the readout starts at an even column: 132 and 0 as shipped)
        SML_REQUIRE(!h.desc.loaded || h.desc.n_model % 2 == 0, "sml_bank_outvec_contribs: a slot has an odd number of model columns (%d)", h.desc.n_model);
    if (!b->d_vp && (rc = sml::dev_zeros(&b->d_vp, (size_t)b->capacity * b->max_n_out))) return rc;
    if ((rc = launch_readout(b, 0, b->capacity, 2, st))) return rc;                   // part 1: the state block -> d_partial
    hipLaunchKernelGGL(k_vp, dim3(b->max_n_out_loaded, b->capacity), dim3(64), 0, st, (const ResDesc *)b->d_descs, (const double *)b->d_local_model,
                       b->max_n_model, b->d_vp, b->max_n_out);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_bank_get_contribs(sml_bank *bank, int slot, double *v_p, double *v_ml)
{
    BANK_SLOT(bank, slot);
    SML_REQUIRE(bank->d_vp, "sml_bank_get_contribs: sml_bank_outvec_contribs has not run");
    if (v_p) SML_HIP(hipMemcpy(v_p, bank->d_vp + (size_t)slot * bank->max_n_out, sizeof(double) * D.n_out, hipMemcpyDeviceToHost));
    if (v_ml) SML_HIP(hipMemcpy(v_ml, bank->d_partial + (size_t)slot * bank->max_n_out, sizeof(double) * D.n_out, hipMemcpyDeviceToHost));
    return SML_OK;
}

double *sml_bank_feedback_dev(sml_bank *b) { return b ? b->d_feedback : nullptr; }
double *sml_bank_local_model_dev(sml_bank *b) { return b ? b->d_local_model : nullptr; }
double *sml_bank_outvec_dev(sml_bank *b) { return b ? b->d_outvec : nullptr; }

int sml_bank_advance_all(sml_bank *b, void *stream)
{
    SML_REQUIRE(b, "null bank");
    int rc = sync_descs(b);
    if (rc) return rc;
    return launch_update(b, 0, b->capacity, b->d_feedback, sml::as_stream(stream));
}

int sml_bank_predict_all(sml_bank *b, int flags, void *stream)
{
    SML_REQUIRE(b, "null bank");
    int rc = sync_descs(b);
    if (rc) return rc;
    if ((rc = launch_update(b, 0, b->capacity, b->d_feedback, sml::as_stream(stream)))) return rc;
    return launch_readout(b, 0, b->capacity, flags, sml::as_stream(stream));
}

int sml_bank_readout_part(sml_bank *b, int part, int flags, void *stream)
{
    SML_REQUIRE(b && (part == 1 || part == 2), "sml_bank_readout_part: part must be 1 (state columns) or 2 (model columns)");
    int rc = sync_descs(b);
    if (rc) return rc;
    // flags bit 3 (8): persistent bounded-footprint kernel (part 1 on a side stream); bit 4 (16): full-occupancy drain of its queue
    SML_REQUIRE(!(flags & 24) || part == 1, "sml_bank_readout_part: the persistent variants exist for part 1 only");
    return launch_readout(b, 0, b->capacity, (flags & 25) | (part == 1 ? 2 : 4), sml::as_stream(stream));
}

int sml_bank_predict_one(sml_bank *bank, int slot, double *x_inout, const double *lm, double *outvec)
{
    BANK_SLOT(bank, slot);
    SML_REQUIRE(x_inout && outvec, "sml_bank_predict_one: null array");
    int rc = sync_descs(bank);
    if (rc) return rc;
    // every slot shares the ping-pong parity: stage x into the *current* buffer of this slot, step only this slot,
    // then copy the new state into BOTH buffers so that the slot is consistent whatever the bank parity is.
    if ((rc = upload_state(bank->res[slot], D.x[bank->cur], x_inout))) return rc;
    if (D.n_model && lm) SML_HIP(hipMemcpy(bank->d_local_model + (size_t)slot * bank->max_n_model, lm, sizeof(double) * D.n_model, hipMemcpyHostToDevice));
    const int before = bank->cur;
    if ((rc = launch_update(bank, slot, slot + 1, bank->d_feedback, nullptr))) return rc;
    if ((rc = launch_readout(bank, slot, slot + 1, 0, nullptr))) return rc;
    if ((rc = download_state(bank->res[slot], D.x[bank->cur], x_inout))) return rc;
    SML_HIP(hipMemcpy(D.x[before], D.x[bank->cur], sizeof(double) * D.n, hipMemcpyDeviceToDevice));
    bank->cur = before;
    SML_HIP(hipMemcpy(outvec, bank->d_outvec + (size_t)slot * bank->max_n_out, sizeof(double) * D.n_out, hipMemcpyDeviceToHost));
    return SML_OK;
}

int sml_bank_synchronize_one(sml_bank *bank, int slot, const double *inputs, int length, double *x_inout)
{
    // synchronize (src/mod_reservoir.f90:1354-1381) for ONE reservoir of a shared bank, host arrays as the reference passes them:
    // inputs(d, length) column-major, x in/out.  Only this slot is stepped; the bank's ping-pong parity is left as it was.
    BANK_SLOT(bank, slot);
    SML_REQUIRE(inputs && x_inout && length >= 0, "sml_bank_synchronize_one: bad arguments");
    int rc = sync_descs(bank);
    if (rc) return rc;
    if ((rc = upload_state(bank->res[slot], D.x[bank->cur], x_inout))) return rc;
    const int before = bank->cur;
    double *u = nullptr;
    const size_t stride = (size_t)bank->max_d;                 // launch_update reads u_all + slot * max_d
    SML_HIP(hipMalloc((void **)&u, sizeof(double) * stride * (size_t)std::max(length, 1)));
    if (hipMemcpy2D(u, stride * sizeof(double), inputs, (size_t)D.d * sizeof(double), (size_t)D.d * sizeof(double), (size_t)length, hipMemcpyHostToDevice) != hipSuccess &&
        length > 0) {
        (void)hipFree(u);
        return sml::fail(SML_ERR_HIP, "sml_bank_synchronize_one: input upload failed");
    }
    for (int t = 0; t < length && rc == SML_OK; ++t)
        rc = launch_update(bank, slot, slot + 1, u + (size_t)t * stride - (size_t)slot * stride, nullptr);
    if (rc == SML_OK) rc = download_state(bank->res[slot], D.x[bank->cur], x_inout);
    if (rc == SML_OK && bank->cur != before) SML_HIP(hipMemcpy(D.x[before], D.x[bank->cur], sizeof(double) * D.n, hipMemcpyDeviceToDevice));
    if (rc == SML_OK && bank->cur == before) SML_HIP(hipMemcpy(D.x[before ^ 1], D.x[before], sizeof(double) * D.n, hipMemcpyDeviceToDevice));
    bank->cur = before;
    (void)hipFree(u);
    return rc;
}

int sml_bank_synchronize_all(sml_bank *b, const double *inputs_dev, int length, void *stream)
{
    SML_REQUIRE(b && inputs_dev && length >= 0, "sml_bank_synchronize_all: bad arguments");
    int rc = sync_descs(b);
    if (rc) return rc;
    const size_t step = (size_t)b->capacity * b->max_d;
    for (int t = 0; t < length; ++t)
        if ((rc = launch_update(b, 0, b->capacity, inputs_dev + step * t, sml::as_stream(stream)))) return rc;
    return SML_OK;
}

int sml_bank_timing(sml_bank *b, int enable)
{
    SML_REQUIRE(b, "null bank");
    b->timing = enable != 0;
    b->timing_update = enable != 2;
    return SML_OK;
}

static int drain(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double *ms, int *count)
{
    double total = 0.0;
    for (auto &p : v) {
        SML_HIP(hipEventSynchronize(p.second));
        float t = 0.f;
        SML_HIP(hipEventElapsedTime(&t, p.first, p.second));
        total += t;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    if (ms) *ms = total;
    if (count) *count = (int)v.size();
    v.clear();
    return SML_OK;
}

int sml_bank_timing_collect(sml_bank *b, double *update_ms, int *update_launches, double *readout_ms, int *readout_launches)
{
    SML_REQUIRE(b, "null bank");
    int rc;
    if ((rc = drain(b->ev_update, update_ms, update_launches))) return rc;
    return drain(b->ev_readout, readout_ms, readout_launches);
}

int sml_bank_train_pass(sml_bank *b, const double *noisy_inputs_dev, int T, int discard, int batch,
                        const double *const *model_dev, const double *const *targets_dev,
                        double *const *c_dev, double *const *b_dev, int ml_variant, void *stream)
{
    SML_REQUIRE(b && noisy_inputs_dev && model_dev && targets_dev && c_dev && b_dev, "sml_bank_train_pass: null argument");
    SML_REQUIRE(T > discard && discard >= 0 && batch > 0, "sml_bank_train_pass: need T > discard >= 0 and batch > 0 (T=%d discard=%d batch=%d)", T, discard, batch);
    hipStream_t st = sml::as_stream(stream);
    int rc = sync_descs(b);
    if (rc) return rc;
    // Per-slot states buffers.  The reference flushes C += aug aug^T after every batch (m = 98 columns as shipped); a product that
    // short is bound by the read-modify-write of C (0.18 ms per batch, 20 TF/s).  The state columns of up to `group` consecutive
    // batches are therefore kept and flushed as ONE product (K = group * batch, up to ~2048 columns and 16 GB of buffers): the same
    // columns enter the same sums, only the association of the partial sums changes (1e-15 relative, as inside any GEMM).
    size_t per_batch_bytes = 0;
    for (int s = 0; s < b->capacity; ++s) {
        const ResDesc &D = b->res[s].desc;
        if (D.loaded && c_dev[s] && b_dev[s] && targets_dev[s]) per_batch_bytes += (size_t)D.n * batch * sizeof(double);
    }
    static const int forced_group = getenv("SML_TRAIN_FLUSH_GROUP") ? atoi(getenv("SML_TRAIN_FLUSH_GROUP")) : 0;
    int group = forced_group > 0 ? forced_group : std::max(1, std::min(16, 2048 / batch));
    while (group > 1 && per_batch_bytes * group > ((size_t)16 << 30)) --group;
    std::vector<TrainSlot> ts(b->capacity, TrainSlot{nullptr, 0});
    int nmax = 0;
    // (the states buffers stay with the bank between passes: a pass per 20 batches would otherwise allocate and free
    // 72 MB per resident reservoir every time)
    b->train_states.resize(b->capacity, std::make_pair((double *)nullptr, (size_t)0));
    for (int s = 0; s < b->capacity; ++s) {
        const ResDesc &D = b->res[s].desc;
        if (!D.loaded || !c_dev[s] || !b_dev[s] || !targets_dev[s]) continue;
        const size_t need = (size_t)D.n * batch * group * sizeof(double);
        auto &buf = b->train_states[s];
        if (buf.second < need) {
            if (buf.first) (void)hipFree(buf.first);
            buf = std::make_pair((double *)nullptr, (size_t)0);
            if (hipMalloc((void **)&buf.first, need) != hipSuccess)
                return sml::fail(SML_ERR_HIP, "sml_bank_train_pass: out of device memory for the states buffers");
            buf.second = need;
        }
        ts[s] = TrainSlot{buf.first, D.n};
        nmax = std::max(nmax, D.n);
    }
    TrainSlot *d_ts = nullptr;
    auto cleanup = [&]() { if (d_ts) (void)hipFree(d_ts); };
    if ((rc = sml::dev_upload(&d_ts, ts.data(), ts.size()))) { cleanup(); return rc; }
    const dim3 sgrid((nmax + 255) / 256, b->capacity);
    const size_t step = (size_t)b->capacity * b->max_d;
    hipLaunchKernelGGL(k_zero_state, sgrid, dim3(256), 0, st, b->d_descs, b->capacity, b->cur);          // x = 0 (:1089)
    for (int c = 0; c < discard && rc == SML_OK; ++c) rc = launch_update(b, 0, b->capacity, noisy_inputs_dev + step * c, st);
    if (rc == SML_OK) hipLaunchKernelGGL(k_store_state, sgrid, dim3(256), 0, st, b->d_descs, d_ts, b->capacity, 0, b->cur);   // states(:,1) = x
    int flushed = 0, pending = 0;            // batches completed; of those, not yet multiplied into C and B
    const int training_length = T - discard;
    auto flush = [&]() {
        const size_t c0 = (size_t)discard + (size_t)(flushed - pending) * batch;      // first data column of the pending batches
        for (int s = 0; s < b->capacity && rc == SML_OK; ++s) {
            if (!ts[s].states) continue;
            const ResDesc &D = b->res[s].desc;
            rc = sml_train_accumulate(ts[s].states, D.n_model ? model_dev[s] + c0 * D.n_model : nullptr, targets_dev[s] + c0 * D.n_out,
                                      D.n, D.n_model, D.n_out, pending * batch, c_dev[s], b_dev[s], stream);
        }
        pending = 0;
    };
    for (int i = 1; i <= training_length - 1 && rc == SML_OK; ++i) {
        // the running (unsquared) state lives in the bank, so "restart from saved_state after a flush" (quirk Q6) is implicit;
        // the ML-only loop instead feeds the squared column into A x on the step after a flush (:1031-1044)
        // (up to 12 workgroups per reservoir: a pass has few residents and one launch per time column, so the launch is as long as its
        //  slowest workgroup -- 32 / 64 residents: 63.2 -> 59.2 / 112.2 -> 109.3 ms per 20-batch pass, profiles/micro/sweep_train_parts.sh)
        rc = launch_update(b, 0, b->capacity, noisy_inputs_dev + step * (discard + i - 1), st, ml_variant && i % batch == 0, d_ts,
                           pending * batch + i % batch, 12);              // the new state goes into its states column from the same launch
        if (rc) break;
        if ((i + 1) % batch == 0) {
            ++flushed; ++pending;
            if (pending == group) flush();
        }
    }
    if (rc == SML_OK && pending) flush();       // (columns of an unfinished batch are dropped, as the reference drops them)
    if (rc == SML_OK && hipStreamSynchronize(st) != hipSuccess) rc = sml::fail(SML_ERR_HIP, "sml_bank_train_pass: stream synchronise failed");
    cleanup();
    return rc == SML_OK ? flushed : rc;
}

int sml_bank_readout_part_bytes(sml_bank *b, int part, uint64_t *bytes)
{
    SML_REQUIRE(b && bytes && (part == 1 || part == 2), "sml_bank_readout_part_bytes: bad arguments");
    uint64_t r = 0;
    for (auto &h : b->res) {
        if (!h.desc.loaded) continue;
        const ResDesc &D = h.desc;
        const uint64_t csplit = (uint64_t)((D.n_model + 1) & ~1);
        const uint64_t model_block = (uint64_t)D.n_out * std::min<uint64_t>(csplit, (uint64_t)D.n_aug) * 8;
        // part 1: the state columns of W_out + the partial sums written; part 2: the rest of the full readout's bytes
        // + the partial sums read back
        const uint64_t p1 = (uint64_t)D.n_out * D.n_aug * 8 - model_block + (uint64_t)D.n_out * 8;
        r += part == 1 ? p1 : h.readout_bytes - ((uint64_t)D.n_out * D.n_aug * 8 - model_block) + (uint64_t)D.n_out * 8;
    }
    *bytes = r;
    return SML_OK;
}

/* 1 when the predict kernels read the compact (float) copies of W_out and of the operator's values -- every loaded reservoir's weights
 * are exactly floats, as after sml_bank_load of a reference weights file -- else 0 */
int sml_bank_storage(sml_bank *b, int *compact)
{
    SML_REQUIRE(b && compact, "sml_bank_storage: bad arguments");
    int rc = sml::bank_sync_descs(b);
    if (rc) return rc;
    *compact = b->compact == 1 ? 1 : 0;
    return SML_OK;
}

/* allow = 0: the predict kernels of this bank read the 8-byte copies whatever the weights are (the compact copies stay in memory);
 * 1 (default): automatic, as described at sml_bank_storage */
int sml_bank_use_compact(sml_bank *b, int allow)
{
    SML_REQUIRE(b, "sml_bank_use_compact: null bank");
    b->allow_compact = allow != 0;
    b->descs_dirty = true;
    return SML_OK;
}

int sml_bank_algorithmic_bytes(sml_bank *b, uint64_t *update_bytes, uint64_t *readout_bytes)
{
    SML_REQUIRE(b, "null bank");
    uint64_t u = 0, r = 0;
    (void)sml::bank_sync_descs(b);           // (settles which copies the kernels read)
    for (auto &h : b->res)
        if (h.desc.loaded) { u += b->compact == 1 ? h.update_bytes32 : h.update_bytes; r += b->compact == 1 ? h.readout_bytes32 : h.readout_bytes; }
    if (update_bytes) *update_bytes = u;
    if (readout_bytes) *readout_bytes = r;
    return SML_OK;
}

}  // extern "C"
