// Probe (development aid): issue cost of the FP64 instructions the all-pairs kernels are made of, per wave64
// instruction on one SIMD with W waves resident (cycles = s_memtime ticks of the slowest wave x W / instructions).
// build: hipcc --offload-arch=gfx950 -O2 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void op_kernel(double *out, long long *cyc, int iters) {
    double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = 1.5f + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double c = 0.9999999, d = 1e-7;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // v_fma_f64, 8 independent
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (OP == 1) {  // v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                         "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == 2) {  // v_add_f64
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else if (OP == 3) {  // v_rsq_f64
            asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                         "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 4) {  // v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
        } else if (OP == 5) {  // v_cvt_f32_f64
            asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                         "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (OP == 6) {  // v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                         "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        } else if (OP == 7) {  // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(0.99999f), "v"(1e-6f));
        } else if (OP == 8) {  // v_rcp_f64
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 9) {  // ds_read_b128 broadcast (same address for all lanes)
            __shared__ double sh[512];
            if (it == 0) sh[threadIdx.x] = a0;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a0), "=v"(*(__attribute__((ext_vector_type(2))) double *)&a2) : "v"((it & 15) * 32));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
    if (s == 123.456) out[0] = s;
}

template <int OP>
static void run(const char *name, int wgs_per_cu, int per_iter) {
    double *out; long long *cyc;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, 8); hipMalloc(&cyc, grid * 4 * 8);
    const int iters = 20000;
    hipLaunchKernelGGL((op_kernel<OP>), dim3(grid), dim3(256), 0, 0, out, cyc, 1000);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((op_kernel<OP>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = h[h.size() / 2];
    // s_memtime ticks at 100 MHz; wall time gives ns per instruction per SIMD
    const double inst = double(iters) * per_iter * wgs_per_cu;   // wave-instructions per SIMD
    printf("%-16s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz); in-kernel %.2f ns\n", name,
           wgs_per_cu, ms * 1e6 / inst, ms * 1e6 / inst * 2.4, med * 10.0 / (double(iters) * per_iter));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, 8);
        run<1>("v_mul_f64", w, 8);
        run<2>("v_add_f64", w, 8);
        run<3>("v_rsq_f64", w, 8);
        run<8>("v_rcp_f64", w, 8);
        run<4>("v_rsq_f32", w, 8);
        run<5>("v_cvt_f32_f64", w, 8);
        run<6>("v_cvt_f64_f32", w, 8);
        run<7>("v_fma_f32", w, 8);
        run<9>("ds_read_b128 x2", w, 2);
    }
    return 0;
}
