// Which primitive of the elimination loop slows down beside a trailing update?  One workgroup (256 threads, high-priority
// stream) runs a dependent chain of one primitive; timed with the constant 100 MHz counter, alone and beside a SYRK.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/probes/prim_probe tools/probes/prim_probe.hip \
//        -L superscreen_amd/lib -lsuperscreen_hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "superscreen_hip.h"
__global__ __launch_bounds__(256, 2) void prim(int mode, int iters, double *out, long long *ts) {
    __shared__ double lds[1024];
    __shared__ int chase[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) { lds[i] = 1.0 + 1e-9 * i; chase[i] = (i * 37 + 11) & 1023; }
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    double x = 1.0 + 1e-6 * t, a = 1.0 - 1e-9, b = 1e-9;
    float xf = 1.0f + 1e-3f * t;
    int idx = t;
    double y[16];
    for (int k = 0; k < 16; ++k) y[k] = x + k;
    long long t0 = wall_clock64();
    switch (mode) {
        case 0: for (int i = 0; i < iters; ++i) x = __builtin_fma(x, a, b); break;
        case 1: for (int i = 0; i < iters; ++i) xf = __builtin_fmaf(xf, 0.999f, 1e-3f); break;
        case 2: for (int i = 0; i < iters; ++i) idx = chase[idx]; break;
        case 3: for (int i = 0; i < iters; ++i) __syncthreads(); break;
        case 4: for (int i = 0; i < iters; ++i) {
                    if ((t >> 2) == (i & 63)) lds[(i & 1) * 512 + (t & 3) * 18] = x;
                    __syncthreads();
                    x = __builtin_fma(x, a, lds[(i & 1) * 512 + (t & 3) * 18] * 1e-12);
                } break;
        case 5: { int s = iters; for (int i = 0; i < iters; ++i) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc"); } idx += s; } break;
        case 6: for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) y[k] = __builtin_fma(y[k], a, b);
                } break;
        case 7: for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_rcp(x) + 0.5; break;
        case 8: for (int i = 0; i < iters; ++i) {
                    int lo = __builtin_amdgcn_update_dpp(0, (int)__double2loint(x), 0x55, 0xf, 0xf, true);
                    x = __hiloint2double(__double2hiint(x), lo);
                } break;
        case 9: for (int i = 0; i < iters; ++i) {      // 16 independent LDS reads + 16 FMAs (the independent work of a column)
#pragma unroll
                    for (int k = 0; k < 16; ++k) y[k] = __builtin_fma(y[k], a, lds[(t & 3) * 18 + k + (i & 1) * 512]);
                } break;
    }
    long long t1 = wall_clock64();
    for (int k = 0; k < 16; ++k) x += y[k];
    if (t == 0) { ts[0] = t0; ts[1] = t1; }
    out[t] = x + xf + idx;
}
int main() {
    const int64_t n = 16384, lda = n;
    double *A, *out; long long *ts;
    hipMalloc(&A, n * lda * 8); hipMalloc(&out, 256 * 8); hipMalloc(&ts, 16);
    hipMemset(A, 0, n * lda * 8);
    hipStream_t s1, s2; hipStreamCreate(&s1);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    const char *names[] = {"f64 fma chain", "f32 fma chain", "lds chase", "barrier", "lds store+barrier+load+fma", "salu add",
                           "16 indep f64 fma", "rcp f64 chain", "dpp chain", "16 lds reads + 16 fma"};
    const int iters = 20000;
    for (int mode = 0; mode < 10; ++mode) {
        double ns[3] = {0, 0, 0};
        for (int load = 0; load < 3; ++load) {
            for (int rep = 0; rep < 2; ++rep) {
                hipDeviceSynchronize();
                if (load == 1) ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, s1);
                if (load == 2) ssa_gemm_ex(0, 0, 0, n, n, 512, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, s1);
                hipLaunchKernelGGL(prim, dim3(1), dim3(256), 0, s2, mode, iters, out, ts);
                hipDeviceSynchronize();
                long long t[2]; hipMemcpy(t, ts, 16, hipMemcpyDeviceToHost);
                ns[load] = (t[1] - t[0]) * 10.0 / iters;
            }
        }
        printf("%-30s alone %7.1f ns/iter   beside SYRK %7.1f   beside NN update %7.1f\n", names[mode], ns[0], ns[1], ns[2]);
        fflush(stdout);
    }
    return 0;
}
