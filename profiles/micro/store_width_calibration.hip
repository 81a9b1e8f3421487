// Calibration of rocprofv3's WRITE_SIZE / TCC_EA0_WRREQ on gfx950 for the store widths the reservoir update uses (the guide
// calibrates 16 B/lane only).  Each kernel writes the same 53 MB (1152 reservoirs x 5760 doubles, the state write of one sweep):
//   w16   16 B per lane (global_store_dwordx4), contiguous
//   w8     8 B per lane (global_store_dwordx2), contiguous: 64 lanes = 512 B            <- k_update's x_new store
//   w8s    8 B per lane, lanes scattered over 64 different 128-byte lines                <- the states column of the training pass
//   w8h    8 B per lane, contiguous, but each wavefront stores only every other 512-B piece in a first pass and the rest in a
//          second pass of the same kernel (half-written lines sit in L2 in between)
// run under: rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- ./store_width_calibration
// build: hipcc --offload-arch=gfx950 -O3 -w -o store_width_calibration store_width_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr size_t N = (size_t)1152 * 5760;      // doubles

__global__ void w16(double *p)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < N) { double2 v{1.0, 2.0}; *reinterpret_cast<double2 *>(p + i) = v; }
}
__global__ void w8(double *p)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) p[i] = 1.0;
}
__global__ void w8s(double *p)
{
    // lane l of a wavefront writes element l * 16 + (wavefront % 16) of its group of 16 wavefronts' 16 KB: every line is
    // completed by 16 different wavefronts
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, w = t >> 6, l = t & 63;
    const size_t i = (w >> 4) * 1024 + l * 16 + (w & 15);
    if (i < N) p[i] = 1.0;
}
__global__ void w8h(double *p)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, w = t >> 6, l = t & 63;
    const size_t nw = (N + 63) / 64;
    for (int pass = 0; pass < 2; ++pass) {
        const size_t piece = 2 * w + pass;                     // this wavefront's two 512-B pieces are adjacent: one 128-B line never straddles them
        if (piece < nw && piece * 64 + l < N) p[piece * 64 + l] = 1.0;
        __syncthreads();
    }
}

int main()
{
    double *p;
    hipMalloc(&p, N * 8 + 4096);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(w16, dim3((N / 2 + 255) / 256), dim3(256), 0, 0, p);
        hipLaunchKernelGGL(w8, dim3((N + 255) / 256), dim3(256), 0, 0, p);
        hipLaunchKernelGGL(w8s, dim3((N + 255) / 256), dim3(256), 0, 0, p);
        hipLaunchKernelGGL(w8h, dim3((N / 2 + 255) / 256), dim3(256), 0, 0, p);
    }
    hipDeviceSynchronize();
    printf("bytes written per kernel: %zu\n", N * 8);
    return 0;
}
