// Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by one-hot operands:
// out[la][lb][lane] = D when A = (lane == la), B = (lane == lb).   (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = d;
        }
}
int main() {
    double *d;
    hipMalloc(&d, 64 * 64 * 64 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // for every A lane: which B lanes pair with it, and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (B %2d -> D %2d)", lb, l);
        printf("\n");
    }
    return 0;
}
