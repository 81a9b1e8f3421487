// Diagnostic build only (make span -> libspeedyml_hip_span.so; never the shipped library): every wavefront of the SPEEDY window's
// four kernels records when it started and when it ended (wall_clock64: the 100 MHz constant clock, the same on every CU), so that a
// window can be accounted for from the wavefronts' own point of view -- dispatch ramp (first to last wave start of a launch), bodies,
// drain (first to last wave end) and the gap between the last wave of one launch and the first wave of the next: the kernel boundary
// as the hardware delivers it to dependent work.  profiles/micro/window_span.py drives it.
//
// No atomics (a first version drew a ticket per wave from one counter: 3696 returning atomics on one address stretched k_grid's
// waves from 10 to 48 us).  A wave's record slot is [kernel][launch of that kernel][block * waves per block + wave]; the launch
// number is a plain load of launch[kernel], which nobody writes while that kernel runs: kernel k's block 0 bumps the counter of the
// kernel BEFORE it in the window's cycle (grid -> physics -> spec -> spectral -> grid), whose waves have all retired by then.
#pragma once
#ifdef SML_WAVE_SPAN
struct SpanRec { unsigned long long start, end; unsigned hw, xcc; unsigned long long pad; };      // hw = HW_REG_HW_ID (cu, se, simd, wave slot)
struct SpanBuf { unsigned launch[8]; unsigned waves_cap, launches_cap, pad[6]; SpanRec rec[1]; };
static __device__ SpanBuf *g_span;      // one per translation unit, all pointing at the same buffer (sml_span_attach_*)
struct SpanScope {
    unsigned long long t0, mark = 0;       // mark: one more clock reading of the wave's choice (SML_SPAN_MARK: e.g. its arrival at a barrier), stored in the record's pad
    unsigned idx, kid;
    __device__ __forceinline__ SpanScope(unsigned kernel_id) : t0((unsigned long long)wall_clock64()), idx(0), kid(kernel_id)
    {
        SpanBuf *b = g_span;
        if (b) {
            idx = __builtin_nontemporal_load(&b->launch[kid]);
            if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned prev = kid == 1 ? 4 : kid - 1; b->launch[prev] = b->launch[prev] + 1; }
        }
    }
    __device__ __forceinline__ ~SpanScope()
    {
        SpanBuf *b = g_span;
        if ((threadIdx.x & 63) == 0 && b) {
            const unsigned w = blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6);
            if (idx < b->launches_cap && w < b->waves_cap) {
                __builtin_amdgcn_s_waitcnt(0);                   // the wave's stores have been issued and its loads have landed
                SpanRec r{t0, (unsigned long long)wall_clock64(), (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)),
                          (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u, mark};
                b->rec[((size_t)(kid - 1) * b->launches_cap + idx) * b->waves_cap + w] = r;
            }
        }
    }
};
#define SML_SPAN(kernel_id) SpanScope span_scope_(kernel_id)
#define SML_SPAN_MARK() (span_scope_.mark = (unsigned long long)wall_clock64())
// four 16-bit offsets (10 ns ticks from the wave's start) of the clocks m[s0], m[s1], m[s2] and `now` -- for kernels that note several stamps
#define SML_SPAN_PACK_MARKS(m, s0, s1, s2) (span_scope_.mark = (((m)[s0] - span_scope_.t0) & 0xffffull) | ((((m)[s1] - span_scope_.t0) & 0xffffull) << 16) | \
        ((((m)[s2] - span_scope_.t0) & 0xffffull) << 32) | ((((unsigned long long)wall_clock64() - span_scope_.t0) & 0xffffull) << 48))
#define SML_SPAN_ATTACH(fn)                                                                                  \
    extern "C" int fn(void *buf)                                                                             \
    {                                                                                                        \
        SML_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_span), &buf, sizeof buf));                                    \
        return SML_OK;                                                                                       \
    }
#else
#define SML_SPAN(kernel_id) do { } while (0)
#define SML_SPAN_MARK() do { } while (0)
#define SML_SPAN_PACK_MARKS(m, s0, s1, s2) do { } while (0)
#define SML_SPAN_ATTACH(fn)
#endif
