// Cost of a device-wide barrier between dependent phases on MI355X, against a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier profiles/micro/grid_barrier.hip && /tmp/grid_barrier
// Every workgroup writes a slice, all meet at the barrier (agent-scope release/acquire), every workgroup reads its neighbour's
// slice of the previous phase (on another XCD) -- the dependence pattern of the SPEEDY time step's phases.  The spin is bounded:
// a barrier that is not reached by everybody sets an abort flag and every wave leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef VARIANT
#define VARIANT 1
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *count, unsigned *abort_flag, unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
#if VARIANT == 0
        __atomic_fetch_add(count, 1u, __ATOMIC_RELEASE);          // system scope release / acquire on every poll
        long spins = 0;
        while (__atomic_load_n(count, __ATOMIC_ACQUIRE) < target) {
            if (++spins > 4000000 || __atomic_load_n(abort_flag, __ATOMIC_RELAXED)) { __atomic_store_n(abort_flag, 1u, __ATOMIC_RELAXED); ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#else
        // agent scope: one release before the arrival, relaxed polls, one acquire after the last arrival
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void k_persistent(double *buf, int n_per_wg, int phases, unsigned *count, unsigned *abort_flag)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    for (int ph = 0; ph < phases; ++ph) {
        double *cur = buf + (size_t)(ph & 1) * nwg * n_per_wg, *prev = buf + (size_t)((ph + 1) & 1) * nwg * n_per_wg;
        const int nb = (wg + nwg / 2 + 1) % nwg;                   // a workgroup on another XCD
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) cur[(size_t)wg * n_per_wg + i] = prev[(size_t)nb * n_per_wg + i] + 1.0;
        if (!grid_barrier(count, abort_flag, (unsigned)(ph + 1) * nwg)) return;
    }
}

__global__ __launch_bounds__(512) void k_phase(double *buf, int n_per_wg, int ph)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    double *cur = buf + (size_t)(ph & 1) * nwg * n_per_wg, *prev = buf + (size_t)((ph + 1) & 1) * nwg * n_per_wg;
    const int nb = (wg + nwg / 2 + 1) % nwg;
    for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) cur[(size_t)wg * n_per_wg + i] = prev[(size_t)nb * n_per_wg + i] + 1.0;
}

int main()
{
    const int nwg = 256, n_per_wg = 2048, phases = 104;
    double *buf; unsigned *flags;
    CK(hipMalloc(&buf, sizeof(double) * 2 * nwg * n_per_wg));
    CK(hipMalloc(&flags, 2 * sizeof(unsigned)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(buf, 0, sizeof(double) * 2 * nwg * n_per_wg));
        CK(hipMemset(flags, 0, 2 * sizeof(unsigned)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(512), 0, 0, buf, n_per_wg, phases, flags, flags + 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2]; CK(hipMemcpy(h, flags, sizeof h, hipMemcpyDeviceToHost));
        std::vector<double> out(nwg * n_per_wg);
        CK(hipMemcpy(out.data(), buf + (size_t)((phases - 1) & 1) * nwg * n_per_wg, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
        bool good = true; for (double v : out) good = good && v == (double)phases;
        printf("persistent: %d phases %.1f us = %.2f us per phase  abort=%u  values %s\n", phases, ms * 1e3, ms * 1e3 / phases, h[1], good ? "ok" : "WRONG");
        CK(hipMemset(buf, 0, sizeof(double) * 2 * nwg * n_per_wg));
        CK(hipEventRecord(e0));
        for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(512), 0, 0, buf, n_per_wg, ph);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(out.data(), buf + (size_t)((phases - 1) & 1) * nwg * n_per_wg, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
        good = true; for (double v : out) good = good && v == (double)phases;
        printf("launches  : %d phases %.1f us = %.2f us per phase  values %s\n", phases, ms * 1e3, ms * 1e3 / phases, good ? "ok" : "WRONG");
        // the same launches captured once into a hipGraph and replayed
        static hipGraphExec_t exec = nullptr;
        static hipStream_t cs = nullptr;
        if (!exec) {
            hipGraph_t graph;
            CK(hipStreamCreate(&cs));
            CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
            for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(512), 0, cs, buf, n_per_wg, ph);
            CK(hipStreamEndCapture(cs, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        }
        CK(hipMemset(buf, 0, sizeof(double) * 2 * nwg * n_per_wg));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, cs));
        CK(hipGraphLaunch(exec, cs));
        CK(hipEventRecord(e1, cs)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(out.data(), buf + (size_t)((phases - 1) & 1) * nwg * n_per_wg, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
        good = true; for (double v : out) good = good && v == (double)phases;
        printf("hipGraph  : %d phases %.1f us = %.2f us per phase  values %s\n", phases, ms * 1e3, ms * 1e3 / phases, good ? "ok" : "WRONG");
    }
    return 0;
}
