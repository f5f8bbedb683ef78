// Probe (development aid): (1) which physical CU (xcc, se, cu) a bit of hipExtStreamCreateWithCUMask selects;
// (2) does a stream masked to "everything but R" really stay off R; (3) a dependent FP64 FMA chain (what the
// diagonal-block kernels of the factorization are made of) alone / beside an MFMA burn on all CUs / on a stream
// masked to R beside the burn masked to the complement of R.
// build: hipcc --offload-arch=gfx950 -O2 -o cumask2_probe cumask2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned *ids, long long ticks) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) ids[blockIdx.x] = (hwid & 0xffff) | ((xcc & 0xf) << 16);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void burn_kernel(double *out, int iters) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

__global__ void chain_kernel(double *out, int links, long long *cycles) {
    double x = 1.0 + threadIdx.x * 1e-12;
    const double c = 0.999999999, d = 1e-9;
    long long t0 = wall_clock64();
    for (int i = 0; i < links; ++i) x = __builtin_fma(x, c, d);
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = x; cycles[blockIdx.x] = t1 - t0; }
}

static std::set<unsigned> decode(const std::vector<unsigned> &h) {
    std::set<unsigned> v;   // xcc << 8 | se << 4 | cu
    for (unsigned x : h) v.insert(((x >> 16) << 8) | (((x >> 13) & 0x7) << 4) | ((x >> 8) & 0xf));
    return v;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    const int words = (ncu + 31) / 32;
    unsigned *ids; hipMalloc(&ids, 4096 * 4);
    std::vector<unsigned> h(4096);
    // (1) single bits
    for (int bit : {0, 1, 2, 7, 8, 9, 16, 31, 32, 63, 64, 128, 255}) {
        if (bit >= ncu) continue;
        std::vector<uint32_t> mask(words, 0);
        mask[bit / 32] = 1u << (bit % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("bit %d: create failed\n", bit); continue; }
        hipLaunchKernelGGL(where_kernel, dim3(4), dim3(256), 0, s, ids, 200LL);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), ids, 4 * 4, hipMemcpyDeviceToHost);
        auto v = decode(std::vector<unsigned>(h.begin(), h.begin() + 4));
        printf("bit %3d ->", bit);
        for (unsigned x : v) printf(" (xcc %u se %u cu %u)", x >> 8, (x >> 4) & 0xf, x & 0xf);
        printf("\n");
        hipStreamDestroy(s);
    }
    // (2) R = bits 0..15, complement stream
    std::vector<uint32_t> mR(words, 0), mC(words, 0xffffffffu);
    mR[0] = 0xffff;
    mC[0] = 0xffff0000u;
    if (ncu % 32) mC[words - 1] &= (1u << (ncu % 32)) - 1;
    hipStream_t sR, sC, sPlain;
    printf("create R: %s\n", hipGetErrorString(hipExtStreamCreateWithCUMask(&sR, words, mR.data())));
    printf("create C: %s\n", hipGetErrorString(hipExtStreamCreateWithCUMask(&sC, words, mC.data())));
    hipStreamCreateWithFlags(&sPlain, hipStreamNonBlocking);
    for (auto pr : {std::make_pair(sR, "R"), std::make_pair(sC, "complement"), std::make_pair(sPlain, "plain")}) {
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(256), 0, pr.first, ids, 500LL);
        hipStreamSynchronize(pr.first);
        hipMemcpy(h.data(), ids, 4096 * 4, hipMemcpyDeviceToHost);
        auto v = decode(h);
        printf("%s: %zu distinct CUs", pr.second, v.size());
        if (v.size() <= 32) for (unsigned x : v) printf(" (%u,%u,%u)", x >> 8, (x >> 4) & 0xf, x & 0xf);
        printf("\n");
    }
    // (3) dependent chain alone / beside burn / isolated
    double *out; long long *cyc; hipMalloc(&out, 4096 * 8); hipMalloc(&cyc, 64 * 8);
    const int links = 200000;
    auto time_chain = [&](hipStream_t s) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(256), 0, s, out, links, cyc);
        hipEventRecord(b, s);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms * 1e6f / links;   // ns per link
    };
    auto time_burn = [&](hipStream_t s, int grid, int iters) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(burn_kernel, dim3(grid), dim3(256), 0, s, out, iters);
        hipEventRecord(b, s);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms;
    };
    printf("chain alone (plain stream): %.1f ns/link\n", time_chain(sPlain));
    printf("chain alone (R stream):     %.1f ns/link\n", time_chain(sR));
    const int iters = 20000;   // 8 MFMA x 64 cycles x iters = 10 M cycles ~ 5 ms at two waves per SIMD
    printf("burn 512 WGs plain: %.2f ms; complement: %.2f ms\n", time_burn(sPlain, 512, iters / 4), time_burn(sC, 512, iters / 4));
    hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(256), 0, sPlain, out, iters * 2);
    printf("chain (plain) beside burn (plain, all CUs): %.1f ns/link\n", time_chain(sR == nullptr ? sPlain : sPlain));
    hipDeviceSynchronize();
    hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(256), 0, sPlain, out, iters * 2);
    printf("chain (R) beside burn (plain, all CUs):     %.1f ns/link\n", time_chain(sR));
    hipDeviceSynchronize();
    hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(256), 0, sC, out, iters * 2);
    printf("chain (R) beside burn (complement of R):    %.1f ns/link\n", time_chain(sR));
    hipDeviceSynchronize();
    hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(256), 0, sC, out, iters * 2);
    printf("chain (plain) beside burn (complement):     %.1f ns/link\n", time_chain(sPlain));
    hipDeviceSynchronize();
    return 0;
}
