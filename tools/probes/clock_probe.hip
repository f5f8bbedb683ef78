// Probe: effective shader clock of an otherwise idle CU while an FP64 MFMA SYRK runs on the rest of
// the chip (dependent integer chain timed with the constant-rate 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "superscreen_hip.h"
__global__ void chain_kernel(long long *out, int iters) {
    unsigned x = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = (x ^ (x >> 3)) * 1664525u + k;  // dependent chain (3 ops)
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void fma_kernel(long long *out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
        }
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0) out[1] = 1;
}
__global__ void lds_kernel(long long *out, int iters) {
    __shared__ double buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = i;
    __syncthreads();
    double s = 0;
    long long t0 = wall_clock64();
    int idx = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s += buf[(idx + 64 * k) & 1023];
        idx += 7;
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (s == 12345.0) out[1] = 1;
}
__global__ void barrier_kernel(long long *out, int iters) {
    __shared__ double buf[2][256];
    double v = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        buf[i & 1][threadIdx.x] = v;
        __syncthreads();
        v += buf[i & 1][(threadIdx.x * 7 + i) & 255];
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (v == 12345.0) out[1] = 1;
}
__global__ void rsq_kernel(long long *out, int iters) {
    double v = 1.0 + threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        double y = __builtin_amdgcn_rsq(v);
        double e = __builtin_fma(-(v * y), y, 1.0);
        y = __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
        v = v + y;
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (v == 12345.0) out[1] = 1;
}
int main() {
    const int64_t n = 16384, lda = n;
    double *A; hipMalloc(&A, n * lda * 8); hipMemset(A, 0, n * lda * 8);
    long long *out; hipMalloc(&out, 16);
    hipStream_t m1, m2;
    uint32_t mk1[8], mk2[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) mk1[i] = 0xffffffffu;
    mk1[0] &= ~1u;
    hipExtStreamCreateWithCUMask(&m1, 8, mk1); hipExtStreamCreateWithCUMask(&m2, 8, mk2);
    const int iters = 2000;  // 256k dependent ops
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            if (mode >= 1) for (int q = 0; q < (mode == 2 ? 4 : 1); ++q)
                ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, m1);
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, m2, out, iters);
            hipDeviceSynchronize();
            long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            if (mode >= 1) ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, m1);
            hipLaunchKernelGGL(fma_kernel, dim3(1), dim3(256), 0, m2, out, 4000);
            hipDeviceSynchronize();
            long long h2[2]; hipMemcpy(h2, out, 16, hipMemcpyDeviceToHost);
            if (mode >= 1) ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, m1);
            hipLaunchKernelGGL(lds_kernel, dim3(1), dim3(256), 0, m2, out, 4000);
            hipDeviceSynchronize();
            long long h3[2]; hipMemcpy(h3, out, 16, hipMemcpyDeviceToHost);
            if (mode >= 1) ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, m1);
            hipLaunchKernelGGL(barrier_kernel, dim3(1), dim3(256), 0, m2, out, 4000);
            hipDeviceSynchronize();
            long long h4[2]; hipMemcpy(h4, out, 16, hipMemcpyDeviceToHost);
            if (mode >= 1) ssa_gemm_ex(0, 1, 1, n, n, 256, -1.0, A, lda, A, lda, 1.0, A, lda, SSA_F64, m1);
            hipLaunchKernelGGL(rsq_kernel, dim3(1), dim3(256), 0, m2, out, 4000);
            hipDeviceSynchronize();
            long long h5[2]; hipMemcpy(h5, out, 16, hipMemcpyDeviceToHost);
            if (rep == 2) printf("   barrier loop: %.1f us (%.0f ns/iter);  rsq chain: %.1f us (%.0f ns/iter)\n", h4[0] * 0.01, h4[0] * 10.0 / 4000, h5[0] * 0.01, h5[0] * 10.0 / 4000);
            if (rep == 2) printf("   f64 FMA throughput loop: %.1f us;  LDS read loop: %.1f us\n", h2[0] * 0.01, h3[0] * 0.01);
            if (rep == 2) printf("mode %d (%s): %.1f us for %d dependent ops -> %.2f ns/op\n", mode,
                                 mode == 0 ? "idle chip" : mode == 1 ? "beside 1 SYRK" : "beside 4 SYRKs", h[0] * 0.01, iters * 64,
                                 h[0] * 10.0 / (iters * 64));
        }
    }
    return 0;
}
