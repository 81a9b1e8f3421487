// What a READ-ONLY stream reaches on this part: the calibration for k_readout's roofline fraction (the guide's 6.29 TB/s is a
// float4 COPY, half of its traffic writes).  7.5 GB (the size of one sweep's W_out) read once per launch by
//   r1  one-wavefront workgroups, 17 independent 16-byte non-temporal loads per lane per trip (k_readout's access shape: 17 rows)
//   r2  256-thread workgroups, 8 independent 16-byte loads per lane per trip, grid-stride
//   r3  as r2 with plain (temporal) loads
// each lane sums what it loads and one lane per wavefront stores the sum (so nothing is optimised away).
// build: hipcc --offload-arch=gfx950 -O3 -w -o read_bw_ceiling read_bw_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr size_t BYTES = (size_t)7526662144ull / 1024 * 1024, NV = BYTES / 16;     // 16-byte vectors

__global__ __launch_bounds__(64) void r1(const f64x2 *__restrict__ p, double *__restrict__ out, size_t rows_per_wg, size_t row_vecs)
{
    // workgroup b streams 17 "rows" of row_vecs vectors each, lane-strided, as k_readout does
    const f64x2 *base = p + (size_t)blockIdx.x * 17 * row_vecs;
    double acc[17];
#pragma unroll
    for (int r = 0; r < 17; ++r) acc[r] = 0.0;
    for (size_t c = threadIdx.x; c < row_vecs; c += 64) {
        f64x2 v[17];
#pragma unroll
        for (int r = 0; r < 17; ++r) v[r] = __builtin_nontemporal_load(base + (size_t)r * row_vecs + c);
#pragma unroll
        for (int r = 0; r < 17; ++r) acc[r] += v[r][0] + v[r][1];
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 17; ++r) s += acc[r];
    if (s == 12345.678) out[blockIdx.x] = s;
}

template <bool NT>
__global__ __launch_bounds__(256) void r2(const f64x2 *__restrict__ p, double *__restrict__ out, size_t nv)
{
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0.0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < nv; i += 8 * stride) {
        f64x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k][0] + v[k][1];
    }
    for (; i < nv; i += stride) acc += p[i][0] + p[i][1];
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

// r4: the readout's row structure with r2's shape: a T-thread workgroup owns R consecutive rows (row stride 5904 doubles) and walks
// their columns T x 16 bytes at a time, R independent loads per lane per trip
template <int T, int R>
__global__ __launch_bounds__(T) void r4(const f64x2 *__restrict__ p, double *__restrict__ out, size_t row_vecs)
{
    const f64x2 *base = p + (size_t)blockIdx.x * R * row_vecs;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (size_t c = threadIdx.x; c < row_vecs; c += T) {
        f64x2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(base + (size_t)r * row_vecs + c);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += v[r][0] + v[r][1];
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
    if (s == 12345.678) out[blockIdx.x] = s;
}

int main()
{
    f64x2 *p; double *out;
    hipMalloc(&p, BYTES); hipMalloc(&out, 1 << 24);
    hipMemset(p, 0, BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t row_vecs = 5892 / 2;                                     // a W_out row: 5892 doubles
    const size_t wgs1 = NV / (17 * row_vecs);
    auto time = [&](const char *name, auto launch, double bytes) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    time("r1 one-wave WGs, 17 rows, non-temporal", [&] { hipLaunchKernelGGL(r1, dim3(wgs1), dim3(64), 0, 0, p, out, 17, row_vecs); }, (double)wgs1 * 17 * row_vecs * 16);
    for (int g : {2048, 4096, 8192, 16384})
        { char n[64]; snprintf(n, 64, "r2 256-thread WGs x %d, non-temporal", g); time(n, [&] { hipLaunchKernelGGL(r2<true>, dim3(g), dim3(256), 0, 0, p, out, NV); }, (double)NV * 16); }
    const size_t rv = 5904 / 2;
#define R4(T, R) { char n[64]; snprintf(n, 64, "r4 %d-thread WGs, %d rows each", T, R); const size_t wgs = NV / ((size_t)R * rv); \
        time(n, [&] { hipLaunchKernelGGL((r4<T, R>), dim3(wgs), dim3(T), 0, 0, p, out, rv); }, (double)wgs * R * rv * 16); }
    R4(64, 8) R4(64, 17) R4(128, 8) R4(256, 8) R4(256, 4) R4(256, 17) R4(512, 8) R4(1024, 4)
    time("r3 256-thread WGs x 8192, plain loads", [&] { hipLaunchKernelGGL(r2<false>, dim3(8192), dim3(256), 0, 0, p, out, NV); }, (double)NV * 16);
    return 0;
}
