// What does the instruction offset of an LDS-DMA load (global_load_lds_dwordx4 vOff, s[base:base+1] offset:IMM) apply to on gfx950:
// the global address, the LDS address, or both?  (The 256 x 128 Gram kernel wants ONE M0 write per K-tile and immediate offsets
// for its six DMA instructions.)  Source element e holds the value e; the kernel loads with M0 = 4096, offset = 256 and a lane
// offset of 16 bytes per lane, then dumps LDS.
// build: hipcc --offload-arch=gfx950 -O3 -w -o lds_dma_offset lds_dma_offset.hip ; run: ./lds_dma_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k(const float *src, float *dump, int imm_case)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = -1.0f;
    __syncthreads();
    const unsigned voff = threadIdx.x * 16;
    if (imm_case == 0)
        asm volatile("s_mov_b32 m0, 4096\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n s_waitcnt vmcnt(0)" ::"v"(voff), "s"(src) : "memory");
    else if (imm_case == 1)
        asm volatile("s_mov_b32 m0, 4096\n s_nop 0\n global_load_lds_dwordx4 %0, %1 offset:256\n s_waitcnt vmcnt(0)" ::"v"(voff), "s"(src) : "memory");
    else
        asm volatile("s_mov_b32 m0, 4096\n s_nop 0\n global_load_lds_dwordx4 %0, %1 offset:-256\n s_waitcnt vmcnt(0)" ::"v"(voff), "s"(src + 1024) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 64) dump[i] = lds[i];
}

int main()
{
    std::vector<float> h(16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *src, *dump;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dump, 8192 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 32768, 0, src, dump, c);
        std::vector<float> d(8192);
        hipMemcpy(d.data(), dump, 8192 * 4, hipMemcpyDeviceToHost);
        int first = -1, last = -1;
        for (int i = 0; i < 8192; ++i)
            if (d[i] >= 0) { if (first < 0) first = i; last = i; }
        const char *what[3] = {"no offset", "offset:256", "offset:-256 (base + 4096 B)"};
        if (first < 0) { printf("%-28s: nothing written\n", what[c]); continue; }
        printf("%-28s: LDS bytes [%d, %d) written; first value = source byte %d\n", what[c], first * 4, (last + 1) * 4, (int)d[first] * 4);
        bool contiguous = true;
        for (int i = first; i <= last; ++i) contiguous &= d[i] == d[first] + (i - first);
        printf("%-28s  contiguous copy: %s\n", "", contiguous ? "yes" : "no");
    }
    return 0;
}
