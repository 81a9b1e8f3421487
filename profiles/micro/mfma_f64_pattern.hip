// How fast does v_mfma_f64_4x4x4_4b_f64 run in the GEMM's register pattern (64 accumulators per wave, 4 "af" x 16 "bf"
// operand registers per k-step) -- (a) operands resident in registers, (b) operands re-read from LDS every k-step as
// the GEMM does.  build: hipcc --offload-arch=gfx950 -O3 -w -o pat mfma_f64_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double *out, int iters)
{
    __shared__ double As[16][144], Bs[16][144];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64, l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3;
    for (int i = threadIdx.x; i < 16 * 144; i += 256) { (&As[0][0])[i] = 1.0 + i * 1e-6; (&Bs[0][0])[i] = 0.5 - i * 1e-6; }
    __syncthreads();
    double acc[4][16];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 16; ++b) acc[a][b] = 0.0;
    double af[4], bf[16];
    for (int t = 0; t < 4; ++t) af[t] = As[l4][wi + 16 * t + l15];
    for (int u = 0; u < 16; ++u) bf[u] = Bs[l4][wj + 4 * u + l3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) af[t] = As[kk + l4][wi + 16 * t + l15];
#pragma unroll
                for (int u = 0; u < 16; ++u) bf[u] = Bs[kk + l4][wj + 4 * u + l3];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(bf[u], af[t], acc[t][u], 0, 0, 0);
        }
        if (MODE == 1) { asm volatile("" ::: "memory"); }
    }
    double s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 16; ++b) s += acc[a][b];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    const int blocks = 512, iters = 2000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 512.0 * 64 * 4 /*k-steps*/ * iters * 4.0 * blocks;
    printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main()
{
    run<0>("operands in registers");
    run<1>("operands re-read from LDS");
    return 0;
}
