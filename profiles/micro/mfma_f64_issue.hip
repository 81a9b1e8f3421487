// How fast can ONE wavefront per SIMD issue independent v_mfma_f64_4x4x4_4b_f64, and does it depend on which registers the
// operands sit in?  (Question behind the 256 x 128 Gram kernel: its one-wave-per-SIMD MFMA stream ran at 45 TF/s where two
// waves per SIMD sustain 58.)  Each variant issues 64 independent MFMAs per loop trip into a[0:127]; cycles per MFMA come from
// s_memtime around the loop, so clocks / throttling do not matter.
// build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64_issue mfma_f64_issue.hip ; run: ./mfma_f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define STR2(x) #x
#define STR(x) STR2(x)

// variant 0: srcA v[0:1], srcB v[2:3] for all        (banks 0,1 | 2,3)
// variant 1: srcA v[0:1], srcB v[4:5]                (same banks 0,1 | 0,1)
// variant 2: like the kernel: srcA one of 16 pairs (changes every 8), srcB one of 8 pairs (rotates)
template <int VAR>
__global__ __launch_bounds__(512) void k(unsigned long long *cyc, double *sink, int trips, const double *gsrc)
{
    extern __shared__ double pad[];
    double x = threadIdx.x * 1e-3;
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %1" ::"v"((float)x), "v"(1.0f) : CLOB);
    const double *gp = gsrc + (size_t)blockIdx.x * (VAR >= 8 ? 131072 : 4096) + (threadIdx.x & 63) * 2 + (VAR >= 8 ? 0 : (threadIdx.x >> 6) * 128);
    const unsigned lp = (unsigned)(threadIdx.x * 16), lbase = __builtin_amdgcn_readfirstlane((unsigned)((threadIdx.x >> 6) * 1024));
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; ++i) {
        if (VAR == 0) {
            asm volatile(
#define M(i) "v_mfma_f64_4x4x4_4b_f64 a[" STR(i) ":" STR(i) "+1], v[0:1], v[2:3], a[" STR(i) ":" STR(i) "+1]\n"
#define M8(b) M(b) M(b + 2) M(b + 4) M(b + 6) M(b + 8) M(b + 10) M(b + 12) M(b + 14)
#define M64(b) M8(b) M8(b + 16) M8(b + 32) M8(b + 48) M8(b + 64) M8(b + 80) M8(b + 96) M8(b + 112)
                M64(0)::: "memory", CLOB);
#undef M
        } else if (VAR == 1) {
            asm volatile(
#define M(i) "v_mfma_f64_4x4x4_4b_f64 a[" STR(i) ":" STR(i) "+1], v[0:1], v[4:5], a[" STR(i) ":" STR(i) "+1]\n"
                M64(0)::: "memory", CLOB);
#undef M
        } else if (VAR == 2) {
            asm volatile(
#define M(i, A, B) "v_mfma_f64_4x4x4_4b_f64 a[" STR(i) ":" STR(i) "+1], v[" STR(A) ":" STR(A) "+1], v[" STR(B) ":" STR(B) "+1], a[" STR(i) ":" STR(i) "+1]\n"
#define R8(b, A) M(b, A, 40) M(b + 2, A, 42) M(b + 4, A, 44) M(b + 6, A, 46) M(b + 8, A, 48) M(b + 10, A, 50) M(b + 12, A, 52) M(b + 14, A, 54)
                R8(0, 8) R8(16, 10) R8(32, 12) R8(48, 14) R8(64, 16) R8(80, 18) R8(96, 20) R8(112, 22)
                ::: "memory", CLOB);
#undef M
        } else if (VAR == 3) {
            // same as 2 but srcB pairs on the other two banks than srcA (srcA even pair index*2 -> banks alternate anyway); B at 41.. odd start
            asm volatile(
#define M(i, A, B) "v_mfma_f64_4x4x4_4b_f64 a[" STR(i) ":" STR(i) "+1], v[" STR(A) ":" STR(A) "+1], v[" STR(B) ":" STR(B) "+1], a[" STR(i) ":" STR(i) "+1]\n"
#define Q8(b, A) M(b, A, 42) M(b + 2, A, 46) M(b + 4, A, 50) M(b + 6, A, 54) M(b + 8, A, 58) M(b + 10, A, 62) M(b + 12, A, 66) M(b + 14, A, 70)
                Q8(0, 8) Q8(16, 12) Q8(32, 16) Q8(48, 20) Q8(64, 24) Q8(80, 28) Q8(96, 32) Q8(112, 36)
                ::: "memory", CLOB);
#undef M
        } else if (VAR >= 4) {
            // variant 2's MFMA stream with TWO memory instructions per 64 MFMAs threaded through it:
            // 4 = global_load_lds_dwordx4 (LDS-DMA), 5 = global_load_dwordx4 into registers, 6 = ds_write_b128, 7 = ds_read2_b64 x 6
#define M(i, A, B) "v_mfma_f64_4x4x4_4b_f64 a[" STR(i) ":" STR(i) "+1], v[" STR(A) ":" STR(A) "+1], v[" STR(B) ":" STR(B) "+1], a[" STR(i) ":" STR(i) "+1]\n"
#define X4 "s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, off\n"
#define X5 "global_load_dwordx4 v[60:63], %0, off\n"
#define X6 "ds_write_b128 %1, v[64:67]\n"
#define X7 "ds_read2_b64 v[56:59], %1 offset1:16\n ds_read2_b64 v[60:63], %1 offset0:32 offset1:48\n ds_read2_b64 v[64:67], %1 offset0:64 offset1:80\n"
#define BODY(X) R8(0, 8) R8(16, 10) X R8(32, 12) R8(48, 14) R8(64, 16) R8(80, 18) X R8(96, 20) R8(112, 22)
            if (VAR == 4) asm volatile(BODY(X4) :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 5) asm volatile(BODY(X5) :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 6) asm volatile(BODY(X6) :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 7) asm volatile(BODY(X7) :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 10) asm volatile(BODY("s_cmp_eq_u32 %2, 77\n s_cbranch_scc1 1f\n v_mov_b32 v70, v71\n 1:\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 11) asm volatile(BODY("v_lshl_add_u64 v[68:69], v[70:71], 0, v[68:69]\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 13) asm volatile(BODY("s_add_u32 m0, %2, 64\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 14) asm volatile(BODY("s_nop 0\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 15) asm volatile(BODY("s_add_u32 s40, %2, 64\n s_addc_u32 s41, s41, 0\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", "s40", "s41", CLOB);
            if (VAR == 16) asm volatile(BODY("s_cmp_eq_u32 %2, 77\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR == 12) asm volatile(BODY("s_add_u32 m0, %2, 64\n s_nop 0\n") :: "v"(gp), "v"(lp), "s"(lbase) : "memory", CLOB);
            if (VAR >= 20) {
                // bundles: how the cost depends on WHERE the non-MFMA instructions of half a k-step (64 MFMAs) sit
#define RD(o) "ds_read2_b64 v[56:59], %1 offset0:" STR(o) " offset1:" STR(o) "+16\n"
#define DM "s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %3, %4 offset:256\n global_load_lds_dwordx4 %3, %4 offset:1280\n s_add_u32 s40, s40, 64\n s_addc_u32 s41, s41, 0\n s_add_u32 s42, s42, 64\n s_addc_u32 s43, s43, 0\n"
#define ONE(X) R8(0, 8) R8(16, 10) X R8(32, 12) R8(48, 14) R8(64, 16) R8(80, 18) R8(96, 20) R8(112, 22)
#define SIX(A, B, C, D, E, F) R8(0, 8) A R8(16, 10) B R8(32, 12) C R8(48, 14) D R8(64, 16) E R8(80, 18) F R8(96, 20) R8(112, 22)
#define OPS :: "v"(gp), "v"(lp), "s"(lbase), "v"(voff), "s"(gsrc) : "memory", "s40", "s41", "s42", "s43", CLOB
                const unsigned voff = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 2048 + blockIdx.x * 8192;
                if (VAR == 20) asm volatile(ONE(RD(0) RD(2) RD(4) RD(6) RD(8) RD(10)) OPS);
                if (VAR == 21) asm volatile(SIX(RD(0), RD(2), RD(4), RD(6), RD(8), RD(10)) OPS);
                if (VAR == 22) asm volatile(ONE(DM) OPS);
                if (VAR == 23) asm volatile(ONE(DM RD(0) RD(2) RD(4) RD(6) RD(8) RD(10)) OPS);
                if (VAR == 24) asm volatile(SIX(DM, , , RD(0) RD(2) RD(4) RD(6) RD(8) RD(10), , ) OPS);
                if (VAR == 25) asm volatile(ONE("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n") OPS);
                if (VAR == 26) asm volatile(ONE("v_lshl_add_u64 v[68:69], v[70:71], 0, v[68:69]\n v_lshl_add_u64 v[66:67], v[70:71], 0, v[66:67]\n v_lshl_add_u64 v[64:65], v[70:71], 0, v[64:65]\n v_lshl_add_u64 v[62:63], v[70:71], 0, v[62:63]\n") OPS);
            }
            if (VAR == 8 || VAR == 9) {
                const double *sp = gp + (size_t)((i * 2) & 127) * 1024 + (threadIdx.x >> 6) * 128;     // a fresh KB per wave per instruction
                if (VAR == 8) asm volatile(BODY(X4) :: "v"(sp), "v"(lp), "s"(lbase) : "memory", CLOB);
                else asm volatile(BODY(X5) :: "v"(sp), "v"(lp), "s"(lbase) : "memory", CLOB);
            }
            if ((i & 15) == 15) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef M
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (trips < 0) sink[threadIdx.x] = pad[threadIdx.x];
}

template <int VAR>
static void run(int threads, const char *what)
{
    unsigned long long *cyc;
    double *sink;
    const int wgs = 256, trips = 2000;
    hipMalloc(&cyc, wgs * 8 * sizeof(*cyc));
    hipMalloc(&sink, 4096);
    double *gsrc;
    hipMalloc(&gsrc, (size_t)256 * 131072 * 8 + 65536);
    hipFuncSetAttribute((const void *)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(threads), 100 * 1024, 0, cyc, sink, trips, gsrc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(threads), 100 * 1024, 0, cyc, sink, trips, gsrc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = 64.0 * trips;
    const double flops = 512.0 * mf * (threads / 64) * wgs;
    printf("%-44s waves/SIMD %d: %.2f counter ticks per MFMA per wave, %.1f TF/s\n", what, threads / 256, (double)h[0] / mf, flops / ms / 1e9);
    hipFree(cyc);
    hipFree(sink);
    hipFree(gsrc);
}

int main()
{
    for (int th : {256, 512}) {
        run<0>(th, "srcA v[0:1] srcB v[2:3]");
        run<1>(th, "srcA v[0:1] srcB v[4:5] (same banks)");
        run<2>(th, "kernel pattern, pairs 8.. and 40..");
        run<3>(th, "kernel pattern, srcA 8,12,.. srcB 42,46,..");
        run<4>(th, "+ 2 global_load_lds_dwordx4 per 64 MFMAs");
        run<5>(th, "+ 2 global_load_dwordx4 per 64 MFMAs");
        run<6>(th, "+ 2 ds_write_b128 per 64 MFMAs");
        run<7>(th, "+ 6 ds_read2_b64 per 64 MFMAs");
        run<10>(th, "+ 2 (s_cmp, not-taken s_cbranch, v_mov)");
        run<11>(th, "+ 2 v_lshl_add_u64");
        run<12>(th, "+ 2 (s_add m0, s_nop)");
        run<13>(th, "+ 2 s_add m0");
        run<14>(th, "+ 2 s_nop 0");
        run<15>(th, "+ 2 (s_add_u32, s_addc_u32)");
        run<16>(th, "+ 2 (s_cmp, not-taken s_cbranch, s_nop)");
        run<20>(th, "6 ds_read2_b64 at ONE point per 64 MFMAs");
        run<21>(th, "6 ds_read2_b64 at SIX points");
        run<22>(th, "ONE point: m0, 2 DMA (saddr+imm), 4 SALU");
        run<23>(th, "ONE point: that + 6 ds_read2_b64");
        run<24>(th, "TWO points: DMA bundle | 6 reads");
        run<25>(th, "ONE point: 8 s_nop 0");
        run<26>(th, "ONE point: 4 v_lshl_add_u64");
        run<8>(th, "+ 2 global_load_lds_dwordx4, streaming");
        run<9>(th, "+ 2 global_load_dwordx4, streaming");
    }
    return 0;
}
