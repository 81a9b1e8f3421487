// Can a chain of dependent 5-10 us launches run faster when consecutive launches sit in TWO hardware queues and order themselves
// through counters in memory instead of the queue's barrier?  (Round 4: the SPEEDY window is 104 such launches, and a launch boundary
// costs 1.5-2.3 us plus whatever of the next kernel's prologue -- table loads that do not depend on the predecessor -- cannot start
// before it.)
//
//   hipcc --offload-arch=gfx950 -O2 -o chain_two_queues chain_two_queues.hip && ./chain_two_queues
//
// Kernel body (the skeleton of a transform kernel): every workgroup stages `table_kb` of a STATIC table in LDS (the part that does not
// depend on the previous launch), then reads 8 doubles per thread that a different workgroup of the previous launch wrote, and writes
// 8 doubles per thread for the next launch.
//   one queue   : plain dependent launches on one stream
//   two queues  : launch i goes to stream i % 2 (CU-masked streams: a hardware queue each); after its static prologue a workgroup
//                 waits until the counter of launch i - 1 has reached that launch's workgroup count (acquire, agent scope); a
//                 workgroup that has written its output does a release fence and bumps its launch's counter.  Stream order keeps
//                 launch i + 2 behind launch i, so at most two launches are in flight; a wait that lasts longer than 2 ms gives up
//                 and raises a flag (no hang).
// Output: microseconds per launch for both, and whether the data arrived intact (each launch adds 1.0 to what it read).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef FENCES
#define FENCES 0      // 1: plain loads / stores + agent-scope release / acquire fences in every wavefront (L2 write-back and invalidate: 57 us per launch)
#endif
extern __shared__ double dyn_lds[];

template <bool CHAINED>
__global__ void k_body(const double *__restrict__ table, int table_doubles, const double *__restrict__ in, double *__restrict__ out,
                       const unsigned *wait_counter, unsigned wait_for, unsigned *my_counter, int *flag)
{
    // static prologue
    double acc = 0.0;
    for (int i = threadIdx.x; i < table_doubles; i += blockDim.x) dyn_lds[i] = table[((size_t)blockIdx.x * 64 + i) % (1 << 17)];
    if (CHAINED) {
        if (threadIdx.x == 0 && wait_counter) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(wait_counter, FENCES ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_for) {
                if (wall_clock64() - t0 > 200000) { *flag = 1; break; }          // 2 ms at 100 MHz
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
#if FENCES
        if (wait_counter) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every wavefront: nothing cached from before the wait
#endif
    } else {
        __syncthreads();
    }
    const unsigned nb = gridDim.x, src = (blockIdx.x * 37u + 11u) % nb, per = blockDim.x * 8;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double *p = in + (size_t)src * per + i * blockDim.x + threadIdx.x;
        if (CHAINED && !FENCES) v[i] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // device-coherent load (sc1): not served from a stale L2 line
        else v[i] = *p;
    }
    acc = dyn_lds[threadIdx.x % table_doubles] * 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double *p = out + (size_t)blockIdx.x * per + i * blockDim.x + threadIdx.x;
        if (CHAINED && !FENCES) __hip_atomic_store(p, v[i] + 1.0 + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through store (sc1)
        else *p = v[i] + 1.0 + acc;
    }
    if (CHAINED) {
#if FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                       // this wavefront's stores are visible device-wide (L2 write-back)
#else
        __builtin_amdgcn_s_waitcnt(0x0f70);                                      // vmcnt(0): the write-through stores have completed
#endif
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(my_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Geo { const char *name; int wgs, threads, table_kb; };

int main()
{
    const int LAUNCHES = 104, REPS = 8;
    std::vector<Geo> cycle = {{"k_grid", 231, 1024, 64}, {"k_physics", 72, 192, 8}, {"k_spec", 219, 704, 24}, {"k_spectral", 248, 64, 2}};
    double *table, *a, *b;
    unsigned *counters;
    int *flag;
    const size_t buf = (size_t)256 * 1024 * 8;
    CK(hipMalloc(&table, sizeof(double) << 17)); CK(hipMemset(table, 0, sizeof(double) << 17));
    CK(hipMalloc(&a, buf * sizeof(double))); CK(hipMalloc(&b, buf * sizeof(double)));
    CK(hipMalloc(&counters, sizeof(unsigned) * (LAUNCHES + 1) * 16)); CK(hipMalloc(&flag, sizeof(int)));
    CK(hipFuncSetAttribute((const void *)k_body<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_body<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t main_st, q[2];
    CK(hipStreamCreate(&main_st));
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
    for (int i = 0; i < 2; ++i) CK(hipExtStreamCreateWithCUMask(&q[i], (uint32_t)mask.size(), mask.data()));
    hipEvent_t e0, e1, fork, join[2];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));

    auto run = [&](bool chained, const std::vector<Geo> &geos) {
        double best = 1e30;
        for (int rep = 0; rep < REPS; ++rep) {
            CK(hipMemsetAsync(a, 0, buf * sizeof(double), main_st));
            CK(hipMemsetAsync(counters, 0, sizeof(unsigned) * (LAUNCHES + 1) * 16, main_st));
            CK(hipMemsetAsync(flag, 0, sizeof(int), main_st));
            CK(hipStreamSynchronize(main_st));
            CK(hipEventRecord(e0, main_st));
            if (chained) {
                CK(hipEventRecord(fork, main_st));
                CK(hipStreamWaitEvent(q[0], fork, 0)); CK(hipStreamWaitEvent(q[1], fork, 0));
            }
            for (int i = 0; i < LAUNCHES; ++i) {
                const Geo &g = geos[i % geos.size()];
                const Geo &gp = geos[(i + geos.size() - 1) % geos.size()];
                const double *in = (i & 1) ? b : a;
                double *out = (i & 1) ? a : b;
                if (chained)
                    hipLaunchKernelGGL(k_body<true>, dim3(g.wgs), dim3(g.threads), (size_t)g.table_kb * 1024, q[i & 1], (const double *)table, g.table_kb * 128, in, out,
                                       i ? (const unsigned *)(counters + 16 * (i - 1)) : (const unsigned *)nullptr, (unsigned)gp.wgs, counters + 16 * i, flag);
                else
                    hipLaunchKernelGGL(k_body<false>, dim3(g.wgs), dim3(g.threads), (size_t)g.table_kb * 1024, main_st, (const double *)table, g.table_kb * 128, in, out,
                                       (const unsigned *)nullptr, 0u, (unsigned *)nullptr, flag);
            }
            if (chained) {
                for (int i = 0; i < 2; ++i) { CK(hipEventRecord(join[i], q[i])); CK(hipStreamWaitEvent(main_st, join[i], 0)); }
            }
            CK(hipEventRecord(e1, main_st));
            CK(hipStreamSynchronize(main_st));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1e3 / LAUNCHES < best) best = ms * 1e3 / LAUNCHES;
        }
        // check: thread 0 of the last launch's workgroup 0 holds LAUNCHES (each launch added 1.0 along some path of workgroups)
        double v = 0.0;
        int f = 0;
        CK(hipMemcpy(&v, (LAUNCHES & 1) ? b : a, sizeof v, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&f, flag, sizeof f, hipMemcpyDeviceToHost));
        printf("    %-10s %7.2f us per launch   (value %.0f, expected %d%s)\n", chained ? "two queues" : "one queue", best, v, LAUNCHES, f ? ", A WAIT TIMED OUT" : "");
        return best;
    };
    printf("window cycle (k_grid, k_physics, k_spec, k_spectral geometries), %d launches:\n", LAUNCHES);
    run(false, cycle); run(true, cycle);
    for (const Geo &g : cycle) {
        printf("%s geometry alone (%d x %d threads, %d KB static table):\n", g.name, g.wgs, g.threads, g.table_kb);
        std::vector<Geo> one = {g};
        run(false, one); run(true, one);
    }
    return 0;
}
