// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A lane x one-hot B lane -> which D lanes fire.
// build: hipcc --offload-arch=gfx950 -O3 -w -o layout mfma_f64_4x4x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(double *out)   // grid: 64 (a lane) x 64 (b lane), block 64
{
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[((size_t)la * 64 + lb) * 64 + l] = d;
}

int main()
{
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    probe<<<dim3(64, 64), 64>>>(d);
    std::vector<double> h(64 * 64 * 64);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // for each A lane: list of (b lane -> d lanes)
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        }
        printf("\n");
    }
    return 0;
}
