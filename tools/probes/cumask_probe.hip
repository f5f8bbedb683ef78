// Probe: does hipExtStreamCreateWithCUMask confine a stream's workgroups to the masked CUs?
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void spin_kernel(long long cycles, unsigned *ids) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) ids[blockIdx.x] = (hwid & 0xffff) | ((xcc & 0xf) << 16);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}
static float run(hipStream_t s, int wgs, unsigned *ids) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, 2000LL, ids);  // 2000 ticks @100MHz = 20 us
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    const int wgs = 2048;
    unsigned *ids; hipMalloc(&ids, wgs * 4);
    std::vector<unsigned> h(wgs);
    hipStream_t plain, masked;
    hipStreamCreate(&plain);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mask[0] = 0xff;  // CUs 0..7
    hipError_t e = hipExtStreamCreateWithCUMask(&masked, 8, mask);
    printf("create masked: %s\n", hipGetErrorString(e));
    for (int rep = 0; rep < 2; ++rep) {
        float t = run(plain, wgs, ids);
        hipMemcpy(h.data(), ids, wgs * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> u(h.begin(), h.end());
        printf("plain : %.3f ms, distinct (xcc,hw_id low16) = %zu\n", t, u.size());
        t = run(masked, wgs, ids);
        hipMemcpy(h.data(), ids, wgs * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> v;
        for (unsigned x : h) v.insert(((x >> 16) << 8) | ((x >> 8) & 0xf) | (((x >> 13) & 0x7) << 4));  // xcc, cu_id, se_id
        printf("masked: %.3f ms, distinct (xcc,se,cu) = %zu :", t, v.size());
        for (unsigned x : v) printf(" %x", x);
        printf("\n");
    }
    return 0;
}
