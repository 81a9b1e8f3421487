// How fast does ONE wavefront issue double-precision arithmetic on gfx950, and what changes it?  (The column physics is one column per lane,
// ~1000 dependent-ish fp64 instructions per chain: its wavefronts sit alone on their SIMDs.)
//   hipcc --offload-arch=gfx950 -O2 -o exec_mask_rate exec_mask_rate.hip && ./exec_mask_rate
// 72 workgroups; per lane CH independent fma chains of 4096 steps; cycles per fma INSTRUCTION of one wavefront at the measured time (2.4 GHz
// assumed), for: active lanes 64 / 32 / 16, chains 1 / 2 / 4 / 8 / 16, wavefronts per workgroup 1 / 4 / 8 / 16 (= 1, 1, 2, 4 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int CH>
__global__ void k(double *out, int active, int iters)
{
    if ((int)(threadIdx.x & 63) >= active) return;
    double v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = threadIdx.x * 1e-3 + c;
    const double m = 0.999999, p = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = __builtin_fma(v[c], m, p);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static int run(double *out, int threads, int active, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<CH>, dim3(72), dim3(threads), 0, 0, out, active, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.2f us   %.2f cycles per fma instruction of a wavefront\n", what, best * 1e3, best * 1e-3 * 2.4e9 / (iters * (double)CH));
    return 0;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, 72 * 1024 * 8));
    run<4>(out, 64, 64, "1 wavefront, 4 chains, 64 lanes");
    run<4>(out, 64, 32, "1 wavefront, 4 chains, 32 lanes");
    run<4>(out, 64, 16, "1 wavefront, 4 chains, 16 lanes");
    run<1>(out, 64, 64, "1 wavefront, 1 chain");
    run<2>(out, 64, 64, "1 wavefront, 2 chains");
    run<8>(out, 64, 64, "1 wavefront, 8 chains");
    run<16>(out, 64, 64, "1 wavefront, 16 chains");
    run<4>(out, 256, 64, "4 wavefronts (1 per SIMD), 4 chains");
    run<4>(out, 512, 64, "8 wavefronts (2 per SIMD), 4 chains");
    run<4>(out, 1024, 64, "16 wavefronts (4 per SIMD), 4 chains");
    run<1>(out, 1024, 64, "16 wavefronts (4 per SIMD), 1 chain");
    return 0;
}
