// Does a sleeping placeholder workgroup that holds most of a CU's LDS keep the trailing update's workgroups off that
// CU, so that the one-workgroup diagonal-block kernel runs there without sharing its SIMDs with MFMA waves?
// (a) streams beside a long-running kernel: does any of them queue behind it?  (b) lu_diag256_kernel beside an NN
// update, without and with the placeholder.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLUK_TIMING -I include -I superscreen_amd/csrc \
//        -o tools/probes/hold_probe tools/probes/hold_probe.hip -L superscreen_amd/lib -lsuperscreen_hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>
#include "lu_diag.hpp"
#include "superscreen_hip.h"
using namespace ssa;
__device__ unsigned g_where[8];
__device__ __forceinline__ unsigned where_am_i() {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 0xf) << 8) | ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4);
}
__global__ __launch_bounds__(64) void hold(const int *quit, int token, long long max_ticks, int slot) {
    extern __shared__ char hold_lds[];
    if (threadIdx.x == 0) g_where[slot] = where_am_i();
    const long long t0 = wall_clock64();
    while (true) {
        __builtin_amdgcn_s_sleep(127);
        if (__hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= token) break;
        if (wall_clock64() - t0 > max_ticks) break;
    }
}
__global__ void tiny(int *p, int slot) { if (threadIdx.x == 0) { p[0] = 1; if (slot >= 0) g_where[slot] = where_am_i(); } }
int main() {
    const int64_t n = 16384, lda = n;
    double *A, *WL, *WU, *scr, *D; int32_t *info; long long *ts; int *flag, *junk;
    hipMalloc(&A, n * lda * 8); hipMalloc(&WL, 256 * 256 * 8); hipMalloc(&WU, 256 * 256 * 8);
    hipMalloc(&scr, 6 * 64 * 64 * 8); hipMalloc(&D, 256 * 256 * 8);
    hipMalloc(&info, 4); hipMalloc(&ts, 32 * 8); hipMalloc(&flag, 64); hipMalloc(&junk, 64);
    hipMemset(flag, 0, 64);
    std::vector<double> h(256 * 256);
    for (int i = 0; i < 256; ++i) for (int j = 0; j < 256; ++j)
        h[i * 256 + j] = (i == j) ? 300.0 : (1.0 + 0.3 * ((i * 7 + j * 3) % 5)) / (1 + abs(i - j));
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t s1, s2, sh[2], sq;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&sh[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&sh[1], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sq, hipStreamNonBlocking);
    const int hold_lds = 100 * 1024;
    printf("raise hold LDS: %d\n", (int)hipFuncSetAttribute(reinterpret_cast<const void *>(&hold), hipFuncAttributeMaxDynamicSharedMemorySize, hold_lds));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&luk::lu_diag256_kernel<double>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(luk::LuSmem<double>));
    printf("LuSmem<double> %d bytes\n", (int)sizeof(luk::LuSmem<double>));
    hipMemset(A, 0, n * lda * 8);
    hipDeviceSynchronize();
    // (a) 12 streams beside a 30 ms placeholder
    {
        hipStream_t st[12];
        for (int i = 0; i < 12; ++i) {
            if (i < 3) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
            else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], junk, -1);   // creates the queue
        }
        hipDeviceSynchronize();
        hipLaunchKernelGGL(hold, dim3(1), dim3(64), hold_lds, sh[0], flag, 1, 3000000LL, 0);
        auto t0 = std::chrono::steady_clock::now();
        double worst = 0;
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], junk, -1);
        for (int i = 0; i < 12; ++i) {
            hipStreamSynchronize(st[i]);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > worst) worst = ms;
        }
        printf("(a) 12 streams beside a 30 ms placeholder: all done after %.3f ms (a stream queued behind it would take 30)\n", worst);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sq, junk, -1);
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), 1, 1, sq);
        t0 = std::chrono::steady_clock::now();
        hipStreamSynchronize(sh[0]);
        printf("    placeholder released by a memset on another stream after %.3f ms\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    const char *names[] = {"ge0","row0","upd0","ge1","row1","upd1","ge2","row2","upd2","ge3","row3","upd3","S1","W1","S2","W2","S3","W3"};
    int token = 1;
    for (int mode = 0; mode < 4; ++mode) {   // 0 alone, 1 beside NN update, 2 beside NN update + 1 placeholder, 3 + 2 placeholders
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(D, h.data(), 256 * 256 * 8, hipMemcpyHostToDevice);
            hipMemset(WL, 0, 256 * 256 * 8); hipMemset(WU, 0, 256 * 256 * 8); hipMemset(info, 0, 4);
            hipDeviceSynchronize();
            ++token;
            const int nhold = mode >= 2 ? mode - 1 : 0;
            for (int k = 0; k < nhold; ++k)
                hipLaunchKernelGGL(hold, dim3(1), dim3(64), hold_lds, sh[k], flag, token, 5000000LL, k);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s1);
            if (mode >= 1) ssa_gemm_ex(0, 0, 0, n, n, 512, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, s1);
            hipEventRecord(e1, s1);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, junk, 2);
            hipLaunchKernelGGL((luk::lu_diag256_kernel<double>), dim3(1), dim3(256), sizeof(luk::LuSmem<double>), s2, D, 256,
                               WL, WU, 256, scr, info, ts);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, junk, 3);
            hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), token, 1, s2);
            hipDeviceSynchronize();
            float lms = 0; hipEventElapsedTime(&lms, e0, e1);
            long long t[32]; hipMemcpy(t, ts, sizeof(t), hipMemcpyDeviceToHost);
            unsigned w[8]; hipMemcpyFromSymbol(w, HIP_SYMBOL(g_where), sizeof(w));
            if (rep == 2) {
                printf("mode %d (%s, %d placeholders): diag total %.1f us, update %.1f us; placeholders on %03x %03x, tiny before/after diag on %03x %03x\n",
                       mode, mode == 0 ? "alone" : "beside NN update", nhold, (t[18] - t[0]) * 0.01, lms * 1e3, w[0], w[1], w[2], w[3]);
                for (int i = 0; i < 18; ++i) printf(" %s %.0f", names[i], (t[i + 1] - t[i]) * 0.01);
                printf("\n");
                fflush(stdout);
            }
        }
    }
    return 0;
}
