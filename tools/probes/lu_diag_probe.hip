// Phase timing of lu_diag256_kernel, alone and beside a large NN trailing update (development probe).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLUK_TIMING -I include -I superscreen_amd/csrc \
//        -o tools/probes/lu_diag_probe tools/probes/lu_diag_probe.hip -L superscreen_amd/lib -lsuperscreen_hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "lu_diag.hpp"
#include "superscreen_hip.h"
using namespace ssa;
int main() {
    const int64_t n = 16384, lda = n;
    double *A, *WL, *WU, *scr, *D; int32_t *info; long long *ts;
    hipMalloc(&A, n * lda * 8); hipMalloc(&WL, 256 * 256 * 8); hipMalloc(&WU, 256 * 256 * 8);
    hipMalloc(&scr, 6 * 64 * 64 * 8); hipMalloc(&D, 256 * 256 * 8);
    hipMalloc(&info, 4); hipMalloc(&ts, 32 * 8);
    std::vector<double> h(256 * 256), l(256 * 256), wl(256 * 256), wu(256 * 256);
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j)
        h[i * 256 + j] = (i == j) ? 300.0 : (1.0 + 0.3 * ((i * 7 + j * 3) % 5)) / (1 + abs(i - j));
    hipStream_t s1, s2; hipStreamCreate(&s1);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(A, 0, n * lda * 8);
    const char *names[] = {"ge0","row0","upd0","ge1","row1","upd1","ge2","row2","upd2","ge3","row3","upd3","S1","W1","S2","W2","S3","W3"};
    hipFuncSetAttribute(reinterpret_cast<const void *>(&luk::lu_diag256_kernel<double>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(luk::LuSmem<double>));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(D, h.data(), 256 * 256 * 8, hipMemcpyHostToDevice);
            hipMemset(WL, 0, 256 * 256 * 8); hipMemset(WU, 0, 256 * 256 * 8); hipMemset(info, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            if (mode == 1)  // load generator: NN update n x n x 512
                ssa_gemm_ex(0, 0, 0, n, n, 512, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, s1);
            hipEventRecord(e1, s1);
            hipLaunchKernelGGL((luk::lu_diag256_kernel<double>), dim3(1), dim3(256), sizeof(luk::LuSmem<double>), s2, D, 256,
                               WL, WU, 256, scr, info, ts);
            hipDeviceSynchronize();
            float lms = 0; hipEventElapsedTime(&lms, e0, e1);
            long long t[32]; hipMemcpy(t, ts, sizeof(t), hipMemcpyDeviceToHost);
            int inf; hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
            if (rep == 2) {
                printf("%s: total %.1f us (info %d), update %.1f us\n", mode == 0 ? "alone" : "beside NN update", (t[18] - t[0]) * 0.01, inf, lms * 1e3);
                for (int i = 0; i < 18; ++i) printf("  %-6s %7.1f us\n", names[i], (t[i + 1] - t[i]) * 0.01);
                // check: L U = D, WL L = I, U WU = I
                hipMemcpy(l.data(), D, 256 * 256 * 8, hipMemcpyDeviceToHost);
                hipMemcpy(wl.data(), WL, 256 * 256 * 8, hipMemcpyDeviceToHost);
                hipMemcpy(wu.data(), WU, 256 * 256 * 8, hipMemcpyDeviceToHost);
                double e_lu = 0, e_wl = 0, e_wu = 0;
                for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j) {
                    double s = 0, a = 0, b = 0;
                    for (int k = 0; k < 256; ++k) {
                        const double lik = (k < i) ? l[i * 256 + k] : (k == i ? 1.0 : 0.0);
                        const double ukj = (k <= j) ? l[k * 256 + j] : 0.0;
                        s += lik * ukj;
                        const double lkj = (j < k) ? l[k * 256 + j] : (k == j ? 1.0 : 0.0);
                        a += wl[i * 256 + k] * lkj;
                        const double uik = (i <= k) ? l[i * 256 + k] : 0.0;
                        b += uik * wu[k * 256 + j];
                    }
                    e_lu = fmax(e_lu, fabs(s - h[i * 256 + j]));
                    e_wl = fmax(e_wl, fabs(a - (i == j)));
                    e_wu = fmax(e_wu, fabs(b - (i == j)));
                }
                printf("  max |LU - D| %.2e, |WL L - I| %.2e, |U WU - I| %.2e\n", e_lu, e_wl, e_wu);
            }
        }
    }
    return 0;
}
