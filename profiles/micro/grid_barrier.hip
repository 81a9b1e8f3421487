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

// ---- XCD-hierarchical barrier (MI355X_MICROARCH.md, row barrier-xcd): the workgroups of one XCD meet on that XCD's counter (their
// stores sit in the one L2 they share); the last of them -- the XCD's leader for this phase -- writes that L2 back (agent-scope
// release), arrives on the top counter, waits for the other XCDs' leaders, invalidates (agent-scope acquire) and bumps the XCD's
// generation word; every other workgroup polls its XCD's generation word and ends with its own agent-scope acquire (its CU's L1).
// Which XCD a workgroup runs on is read from the hardware (HW_REG_XCC_ID), never derived from blockIdx; the populations are
// counted in-kernel behind one flat barrier.  Every spin is bounded.
struct XBar {
    unsigned pop[8][32];          // workgroups on each XCD (one 128-byte line per word)
    unsigned cnt[8][32];          // arrivals on each XCD, monotonic
    unsigned gen[8][32];          // last phase released on each XCD
    unsigned top[32];             // arrivals of XCD leaders, monotonic
    unsigned flat[32];            // the one flat barrier behind the census
    unsigned abort_flag[32];
};
#define RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RLX_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ bool spin_until(unsigned *word, unsigned target, unsigned *abort_flag)
{
    for (long spins = 0; RLX_LOAD(word) < target; ++spins) {
        if (spins > 4000000 || ((spins & 1023) == 1023 && RLX_LOAD(abort_flag))) { RLX_STORE(abort_flag, 1u); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__global__ __launch_bounds__(512) void k_persistent_xcd(double *buf, int n_per_wg, int phases, XBar *xb)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    __shared__ unsigned s_xcc, s_pop, s_nx, s_ok;
    if (threadIdx.x == 0) {
        // census: which XCD am I on, how many of us are there, how many XCDs are in use
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;        // hwreg(HW_REG_XCC_ID, 0, 4)
        RLX_ADD(&xb->pop[xcc][0], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        RLX_ADD(&xb->flat[0], 1u);
        bool ok = spin_until(&xb->flat[0], (unsigned)nwg, &xb->abort_flag[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += RLX_LOAD(&xb->pop[x][0]) > 0;
        s_xcc = xcc; s_pop = RLX_LOAD(&xb->pop[xcc][0]); s_nx = nx; s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    const unsigned xcc = s_xcc, pop = s_pop, nx = s_nx;
    for (int ph = 0; ph < phases; ++ph) {
        double *cur = buf + (size_t)(ph & 1) * nwg * n_per_wg, *prev = buf + (size_t)((ph + 1) & 1) * nwg * n_per_wg;
        const int nb = (wg + nwg / 2 + 1) % nwg;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) cur[(size_t)wg * n_per_wg + i] = prev[(size_t)nb * n_per_wg + i] + 1.0;
        __syncthreads();                                         // (s_waitcnt vmcnt(0): this workgroup's stores are in its L2)
        if (threadIdx.x == 0) {
            const unsigned p = (unsigned)ph + 1u;
            bool ok = true;
            if (RLX_ADD(&xb->cnt[xcc][0], 1u) + 1u == p * pop) {              // the XCD's last arrival leads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                RLX_ADD(&xb->top[0], 1u);
                ok = spin_until(&xb->top[0], p * nx, &xb->abort_flag[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                RLX_STORE(&xb->gen[xcc][0], p);
            } else {
                ok = spin_until(&xb->gen[xcc][0], p, &xb->abort_flag[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
    }
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
        {   // the same phases behind the XCD-hierarchical barrier
            static XBar *xb = nullptr;
            if (!xb) CK(hipMalloc(&xb, sizeof(XBar)));
            CK(hipMemset(buf, 0, sizeof(double) * 2 * nwg * n_per_wg));
            CK(hipMemset(xb, 0, sizeof(XBar)));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent_xcd, dim3(nwg), dim3(512), 0, 0, buf, n_per_wg, phases, xb);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            XBar hx; CK(hipMemcpy(&hx, xb, sizeof hx, hipMemcpyDeviceToHost));
            CK(hipMemcpy(out.data(), buf + (size_t)((phases - 1) & 1) * nwg * n_per_wg, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
            good = true; for (double v : out) good = good && v == (double)phases;
            printf("xcd barrier: %d phases %.1f us = %.2f us per phase  abort=%u  values %s  workgroups per XCD:", phases, ms * 1e3, ms * 1e3 / phases,
                   hx.abort_flag[0], good ? "ok" : "WRONG");
            for (int x = 0; x < 8; ++x) printf(" %u", hx.pop[x][0]);
            printf("\n");
        }
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
