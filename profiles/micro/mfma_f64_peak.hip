// Register-resident fp64 peak probes for MI355X ("measure peak with a register-resident MFMA loop on the
// box and report against both"): v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and plain v_fma_f64.
// build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64_peak mfma_f64_peak.hip ; run: ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(double *out, int iters, double a0, double b0)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k4(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void kfma(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void time_it(const char *name, K launch, double flops_per_wave_iter, int nacc, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(blocks, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(blocks, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = flops_per_wave_iter * nacc * iters * 4.0 * blocks;
    printf("%-22s nacc=%2d waves/SIMD=%d : %8.3f ms  %7.2f TFLOP/s\n", name, nacc, blocks_per_cu, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

#define RUN16(N, B) time_it("mfma_f64_16x16x4", [](int bl, double *o, int it) { k16<N><<<bl, 256>>>(o, it, 1.0, 0.5); }, 2.0 * 16 * 16 * 4, N, B)
#define RUN4(N, B) time_it("mfma_f64_4x4x4_4b", [](int bl, double *o, int it) { k4<N><<<bl, 256>>>(o, it, 1.0, 0.5); }, 2.0 * 4 * 4 * 4 * 4, N, B)
#define RUNF(N, B) time_it("v_fma_f64", [](int bl, double *o, int it) { kfma<N><<<bl, 256>>>(o, it, 1.0000001, 0.5); }, 2.0 * 64, N, B)

int main()
{
    RUN16(4, 1); RUN16(8, 1); RUN16(16, 1); RUN16(4, 2); RUN16(8, 2); RUN16(16, 2); RUN16(4, 4); RUN16(8, 4);
    RUN4(8, 1); RUN4(16, 1); RUN4(16, 2); RUN4(16, 4);
    RUNF(8, 1); RUNF(16, 1); RUNF(16, 2); RUNF(16, 4); RUNF(32, 2);
    return 0;
}
