// Which blgp bit of v_mfma_f64_16x16x4_f64 negates which operand (development probe).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    const double a = 1.0 + (lane & 15), b = 2.0;   // A[i][k] = 1 + i, B[k][j] = 2  ->  (A B)[i][j] = 8 (1 + i)
    f64x4 c = {100.0, 100.0, 100.0, 100.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP);
    if (lane == 0) out[BLGP] = c[0];               // row 0, col 0: 8 with C = 100
}
int main() {
    double *d, h[8];
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("blgp 0: %g (expect 108)\nblgp 1: %g\nblgp 2: %g\nblgp 4: %g\n", h[0], h[1], h[2], h[4]);
    return 0;
}
