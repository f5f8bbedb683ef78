// Phase timing of chol_diag256_kernel, alone and beside a large SYRK (development probe).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOLK_TIMING -I include -I superscreen_amd/csrc \
//        -o tools/probes/diag_probe tools/probes/diag_probe.hip -L superscreen_amd/lib -lsuperscreen_hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chol_diag.hpp"
#include "superscreen_hip.h"
using namespace ssa;
__device__ unsigned g_ids[4096];
__global__ void mfma_burn(double *out, int iters) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) g_ids[blockIdx.x] = ((xcc & 0xf) << 8) | ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4);
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.0) out[0] = 1;
}
__global__ void bw_burn(double *p, size_t n, int reps) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r)
        for (size_t k = i; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = p[k] + 1.0;
}
int main() {
    const int64_t n = 16384, lda = n;
    double *A, *W, *scr; int32_t *info; long long *ts;
    hipMalloc(&A, n * lda * 8); hipMalloc(&W, 256 * 256 * 8); hipMalloc(&scr, 4 * 64 * 64 * 8);
    hipMalloc(&info, 4); hipMalloc(&ts, 32 * 8);
    std::vector<double> h(256 * 256);
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j) h[i * 256 + j] = (i == j) ? 300.0 : 1.0 / (1 + abs(i - j));
    double *D; hipMalloc(&D, 256 * 256 * 8);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipStream_t m1, m2;
    uint32_t mk1[8], mk2[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) mk1[i] = 0xffffffffu;
    mk1[0] &= ~1u;
    printf("masked streams: %d %d\n", (int)hipExtStreamCreateWithCUMask(&m1, 8, mk1), (int)hipExtStreamCreateWithCUMask(&m2, 8, mk2));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(A, 0, n * lda * 8);
    const char *names[] = {"ge0","trsm0","upd0","ge1","trsm1","upd1","ge2","trsm2","upd2","ge3","trsm3","upd3","S1","W1","S2","W2","S3","W3"};
    hipFuncSetAttribute(reinterpret_cast<const void *>(&cholk::chol_diag256_kernel<double>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(cholk::Ge64Smem<double>));
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(D, h.data(), 256 * 256 * 8, hipMemcpyHostToDevice);
            hipMemset(W, 0, 256 * 256 * 8); hipMemset(info, 0, 4);
            hipDeviceSynchronize();
            hipStream_t ls = (mode >= 2) ? m1 : s1, ds = (mode >= 2) ? m2 : s2;
            hipEventRecord(e0, ls);
            if (mode == 1 || mode == 2 || mode == 3)  // load generator: SYRK n x n x 256 on the lower tiles
                ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, ls);
            if (mode == 4) hipLaunchKernelGGL(mfma_burn, dim3(2040), dim3(256), 0, ls, scr, 4000);  // MFMA only
            if (mode == 5) hipLaunchKernelGGL(bw_burn, dim3(4080), dim3(256), 0, ls, A, (size_t)n * lda, 3);  // HBM only
            hipEventRecord(e1, ls);
            if (mode != 3)
            hipLaunchKernelGGL((cholk::chol_diag256_kernel<double>), dim3(1), dim3(256), sizeof(cholk::Ge64Smem<double>), ds, D, 256, W,
                               256, scr, info, 1, ts);
            hipDeviceSynchronize();
            float lms = 0; hipEventElapsedTime(&lms, e0, e1);
            if (rep == 2) printf("mode %d: SYRK %.1f us\n", mode, lms * 1e3);
            if (rep == 2 && mode == 4) {
                unsigned hid[4096]; hipMemcpyFromSymbol(hid, HIP_SYMBOL(g_ids), sizeof(hid));
                int on0 = 0; for (int i = 0; i < 2040; ++i) on0 += (hid[i] == 0);
                printf("   burn workgroups that ran on (xcc 0, se 0, cu 0): %d of 2040\n", on0);
            }
            long long t[32]; hipMemcpy(t, ts, sizeof(t), hipMemcpyDeviceToHost);
            int inf; hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost);
            if (rep == 2) {
                printf("%s: total %.1f us (info %d)\n", mode == 0 ? "alone" : mode == 1 ? "beside SYRK" : mode == 4 ? "masked beside MFMA-only burn" : mode == 5 ? "masked beside HBM-only burn" : "masked beside masked SYRK", (t[18] - t[0]) * 0.01, inf);
                if (mode <= 5) for (int i = 0; i < 18; ++i) printf("  %-6s %7.1f us\n", names[i], (t[i + 1] - t[i]) * 0.01);
            }
        }
    }
    return 0;
}
