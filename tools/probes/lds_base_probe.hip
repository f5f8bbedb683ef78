// Which workgroups share a CU, and does HW_REG_LDS_ALLOC tell them apart?  (development probe)
// 1024 workgroups of 256 threads with 74 KB of dynamic LDS (two fit on a CU), each spinning ~20 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ void probe(unsigned *out, int spin_ticks) {
    extern __shared__ char smem[];
    unsigned hwid, xcc, alloc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(alloc));
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = hwid;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = alloc;
        out[4 * blockIdx.x + 3] = (unsigned)t0;
    }
    smem[threadIdx.x] = 1;
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
}
int main() {
    const int nwg = 1024;
    unsigned *d;
    hipMalloc(&d, nwg * 16);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 74 * 1024, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 4);
    hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    unsigned tmin = ~0u;
    for (int b = 0; b < nwg; ++b) tmin = h[4 * b + 3] < tmin ? h[4 * b + 3] : tmin;
    for (int b = 0; b < nwg; ++b) {
        const unsigned hw = h[4 * b], cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[4 * b + 1] & 0xf;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
    int shown = 0, first_gen_pairs_ok = 0, cus_with_two_first = 0;
    for (auto &kv : by_cu) {
        int zero = 0, nonzero = 0, first = 0;
        for (int b : kv.second)
            if (b < 512) { ++first; ((h[4 * b + 2] & 0xfff) == 0 ? zero : nonzero)++; }
        if (first == 2) { ++cus_with_two_first; if (zero == 1 && nonzero == 1) ++first_gen_pairs_ok; }
        if (shown++ < 6) {
            printf("cu %05x:", kv.first);
            for (int b : kv.second) printf("  wg %4d alloc %08x wave %u simd %u t+%u", b, h[4 * b + 2], h[4 * b] & 0xf, (h[4 * b] >> 4) & 3, h[4 * b + 3] - tmin);
            printf("\n");
        }
    }
    printf("CUs that got exactly two of the first 512 workgroups: %d; of those with one LDS base == 0 and one != 0: %d\n", cus_with_two_first, first_gen_pairs_ok);
    return 0;
}
