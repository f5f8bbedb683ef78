// Probe (development aid): dependent-accumulator latency of v_mfma_f64_16x16x4_f64 -- ns per instruction of one
// wave with 1, 2, 4, 8 independent accumulator chains (one wave per SIMD, one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void chain_kernel(double *out, int iters) {
    f64x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f64x4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[0] = s;
}
template <int CH>
static void run(int waves) {
    double *out; hipMalloc(&out, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((chain_kernel<CH>), dim3(1), dim3(64 * waves), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((chain_kernel<CH>), dim3(1), dim3(64 * waves), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("chains %d, waves %d: %.1f ns per MFMA per wave\n", CH, waves, ms * 1e6 / (double(iters) * CH));
}
int main() {
    run<1>(4); run<2>(4); run<4>(4); run<8>(4);
    run<1>(8); run<2>(8);
    return 0;
}
