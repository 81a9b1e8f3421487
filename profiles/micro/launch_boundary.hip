// What does a kernel boundary cost on this part, at the launch geometries of the SPEEDY window?  ("next round" 2a.)
//
//   hipcc --offload-arch=gfx950 -O2 -o launch_boundary launch_boundary.hip && ./launch_boundary
//
// Chains of DEPENDENT launches on one stream (and the same chain replayed from a hipGraph), timed with HIP events over the whole
// chain, best of 5; the figure printed is microseconds per launch.  Kernel bodies:
//   empty  : s_endpgm straight away -- launch + dispatch ramp + drain of that geometry, nothing else
//   touch  : every workgroup reads one 64-byte line that a DIFFERENT workgroup of the previous launch wrote and writes one for the
//            next launch -- adds what a real dependent kernel cannot avoid: the L2 write-back / invalidate at the boundary and one
//            first-touch round trip to memory
//   stage  : every thread of a workgroup loads 8 doubles the previous launch wrote (coalesced, 32 KB per 512-thread workgroup),
//            parks them in LDS, barrier, writes 8 doubles for the next launch -- the skeleton of a transform kernel's staging
//            phase without any arithmetic
// Geometries: the window's four kernels as shipped (k_grid 462 x 512 threads, 53 KB dynamic LDS; k_gridtend_physics 72 x 128,
// 36 KB; k_spec 292 x 512, 51 KB; k_spectral 248 x 64, 2 KB), their four-kernel cycle (the window's dependence skeleton: 26 x 4
// launches), and sweeps of one parameter at a time around k_grid's geometry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

extern __shared__ double dyn_lds[];

__global__ void k_empty(const double *, double *) { if (threadIdx.x == 4096) dyn_lds[0] = 0; }

__global__ void k_touch(const double *__restrict__ in, double *__restrict__ out)
{
    if (threadIdx.x < 8) {
        const unsigned nb = gridDim.x, src = (blockIdx.x * 37u + 11u) % nb;      // a line some other workgroup (usually another XCD) wrote
        out[blockIdx.x * 8 + threadIdx.x] = in[src * 8 + threadIdx.x] + 1.0;
    }
    if (threadIdx.x == 4096) dyn_lds[0] = 0;
}

__global__ void k_stage(const double *__restrict__ in, double *__restrict__ out)
{
    const unsigned nb = gridDim.x, src = (blockIdx.x * 37u + 11u) % nb, per = blockDim.x * 8;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in[(size_t)src * per + i * blockDim.x + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) dyn_lds[i * blockDim.x + threadIdx.x] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) out[(size_t)blockIdx.x * per + i * blockDim.x + threadIdx.x] = dyn_lds[i * blockDim.x + (threadIdx.x ^ 1)] + 1.0;
}

struct Geo { const char *name; int wgs, threads, lds; };
typedef void (*kern_t)(const double *, double *);

static double *g_a, *g_b;

static void enqueue(kern_t k, const Geo &g, int i, hipStream_t st)
{
    hipLaunchKernelGGL(k, dim3(g.wgs), dim3(g.threads), (size_t)g.lds, st, (const double *)((i & 1) ? g_b : g_a), (i & 1) ? g_a : g_b);
}

// microseconds per launch of a chain cycling through `geos`
static double time_chain(kern_t k, const std::vector<Geo> &geos, int launches, bool graph)
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t gr;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < launches; ++i) enqueue(k, geos[i % geos.size()], i, st);
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0));
        CK(hipGraphDestroy(gr));
    }
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, st));
        if (graph) CK(hipGraphLaunch(exec, st));
        else for (int i = 0; i < launches; ++i) enqueue(k, geos[i % geos.size()], i, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, (double)ms * 1e3 / launches);       // first repetition = warm-up
    }
    if (exec) CK(hipGraphExecDestroy(exec));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    CK(hipStreamDestroy(st));
    return best;
}

static void row(const char *label, const std::vector<Geo> &geos)
{
    const int n = 1040;      // ten windows' worth
    printf("%-58s", label);
    for (kern_t k : {(kern_t)k_empty, (kern_t)k_touch, (kern_t)k_stage}) {
        bool ok = true;
        for (const Geo &g : geos) if (k == (kern_t)k_stage && g.lds < g.threads * 64) ok = false;      // the staging body needs 64 B of LDS per thread
        if (!ok) { printf("        -        -"); continue; }
        printf(" %8.2f %8.2f", time_chain(k, geos, n, false), time_chain(k, geos, n, true));
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)2048 * 1024 * 8 * 8;
    CK(hipMalloc(&g_a, bytes)); CK(hipMalloc(&g_b, bytes));
    CK(hipMemset(g_a, 0, bytes)); CK(hipMemset(g_b, 0, bytes));
    for (kern_t k : {(kern_t)k_empty, (kern_t)k_touch, (kern_t)k_stage})
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const Geo grid{"k_grid", 462, 512, 53248}, phys{"k_gridtend_physics", 72, 128, 36864}, spec{"k_spec", 292, 512, 52224}, spectral{"k_spectral", 248, 64, 2048};
    printf("microseconds per dependent launch (best of 5 chains of 1040)   empty:stream  graph   touch:stream  graph   stage:stream  graph\n");
    row("k_grid              462 x 512 thr, 53 KB LDS", {grid});
    row("k_gridtend_physics   72 x 128 thr, 36 KB LDS", {phys});
    row("k_spec              292 x 512 thr, 51 KB LDS", {spec});
    row("k_spectral          248 x  64 thr,  2 KB LDS", {spectral});
    row("the window's cycle: grid -> physics -> spec -> spectral", {grid, phys, spec, spectral});
    row("guide's trivial kernel: 256 x 256 thr, no LDS", {{"", 256, 256, 0}});
    printf("-- workgroup count at 512 threads, 53 KB LDS\n");
    for (int w : {64, 128, 256, 462, 768, 1024, 2048}) { char l[64]; snprintf(l, sizeof l, "  %4d workgroups", w); row(l, {{"", w, 512, 53248}}); }
    printf("-- threads per workgroup at 462 workgroups, 53 KB LDS (1024 threads: 231 workgroups, 106 KB)\n");
    for (int t : {64, 128, 256, 512}) { char l[64]; snprintf(l, sizeof l, "  %4d threads", t); row(l, {{"", 462, t, 53248}}); }
    row("  1024 threads x 231", {{"", 231, 1024, 106496}});
    printf("-- dynamic LDS at 462 x 512 threads\n");
    for (int kb : {0, 8, 16, 32, 53, 64, 80, 128}) { char l[64]; snprintf(l, sizeof l, "  %3d KB", kb); row(l, {{"", 462, 512, kb * 1024}}); }
    printf("-- same total threads (236544), different shapes, 32 KB per 512 threads\n");
    row("  231 x 1024 thr, 64 KB", {{"", 231, 1024, 65536}});
    row("  462 x  512 thr, 32 KB", {{"", 462, 512, 32768}});
    row("  924 x  256 thr, 16 KB", {{"", 924, 256, 16384}});
    row(" 1848 x  128 thr,  8 KB", {{"", 1848, 128, 8192}});
    row(" 3696 x   64 thr,  4 KB", {{"", 3696, 64, 4096}});
    return 0;
}
