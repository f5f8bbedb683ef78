// Probe (development aid): does the cost of a dependent launch on a high-priority stream beside a stream that keeps
// the chip full depend on WHICH stream it is (i.e. on the hardware queue / command-processor pipe the runtime mapped
// it to)?  16 high-priority streams are created in a row; each runs, ALONE, a chain of 100 dependent 10-us
// one-workgroup kernels beside back-to-back big launches (2080 workgroups x 50 us, two per CU) on (a) the null
// stream, (b) a normal-priority stream created before them, (c) one created after them.
// build: hipcc --offload-arch=gfx950 -O2 -o pipe_probe pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

extern __shared__ char smem[];
__global__ __launch_bounds__(256) void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

int main(int argc, char **argv) {
    const int p_links = argc > 1 ? atoi(argv[1]) : 100;
    const long long p_link_ticks = argc > 2 ? atoll(argv[2]) * 100 : 1000;
    const long long p_load_ticks = argc > 3 ? atoll(argv[3]) * 100 : 5000;
    const int p_pairs = argc > 4 ? atoi(argv[4]) : 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    const int kStreams = 16;
    hipStream_t before, after, ch[kStreams], gate_s;
    hipStreamCreateWithFlags(&before, hipStreamNonBlocking);
    for (int i = 0; i < kStreams; ++i) hipStreamCreateWithPriority(&ch[i], hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&after, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&gate_s, hipStreamNonBlocking);
    hipEvent_t gate_ev, t0, t1;
    hipEventCreateWithFlags(&gate_ev, hipEventDisableTiming);
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    const int links = p_links;
    hipStream_t bigs[3] = {nullptr, before, after};
    const char *names[3] = {"null stream", "normal stream created before", "normal stream created after"};
    for (int b = 0; b < 3; ++b) {
        printf("big launches on the %s: dependent launch on high-priority stream j, us:", names[b]);
        for (int j = 0; j < kStreams; ++j) {
            hipDeviceSynchronize();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 0, gate_s, 2000000LL);
            hipEventRecord(gate_ev, gate_s);
            hipStreamWaitEvent(bigs[b], gate_ev, 0);
            hipStreamWaitEvent(ch[j], gate_ev, 0);
            for (int i = 0; i < 26; ++i)
                hipLaunchKernelGGL(spin_kernel, dim3(2080), dim3(256), 68 << 10, bigs[b], p_load_ticks);
            hipEventRecord(t0, ch[j]);
            for (int l = 0; l < links; ++l) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 36 << 10, ch[j], p_link_ticks);
            hipEventRecord(t1, ch[j]);
            hipEventSynchronize(t1);
            float ms = 0;
            hipEventElapsedTime(&ms, t0, t1);
            printf(" %5.1f", ms * 1000.0 / links - p_link_ticks / 100.0);
        }
        printf("\n");
        hipDeviceSynchronize();
    }
    // pairs: chains on streams i and j at the same time beside big launches on the null stream (worse of the two)
    printf("pairs (rows i, columns j), us per dependent launch, worse of the two chains:\n");
    hipEvent_t u0, u1;
    hipEventCreate(&u0);
    hipEventCreate(&u1);
    for (int i = 0; i < p_pairs; ++i) {
        printf("  i=%2d:", i);
        for (int j = 0; j < p_pairs; ++j) {
            if (j <= i) { printf("     -"); continue; }
            hipDeviceSynchronize();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 0, gate_s, 2000000LL);
            hipEventRecord(gate_ev, gate_s);
            hipStreamWaitEvent(nullptr, gate_ev, 0);
            hipStreamWaitEvent(ch[i], gate_ev, 0);
            hipStreamWaitEvent(ch[j], gate_ev, 0);
            for (int k = 0; k < 26; ++k) hipLaunchKernelGGL(spin_kernel, dim3(2080), dim3(256), 68 << 10, nullptr, p_load_ticks);
            hipEventRecord(t0, ch[i]);
            hipEventRecord(u0, ch[j]);
            for (int l = 0; l < links; ++l) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 36 << 10, ch[i], p_link_ticks);
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 36 << 10, ch[j], p_link_ticks);
            }
            hipEventRecord(t1, ch[i]);
            hipEventRecord(u1, ch[j]);
            hipEventSynchronize(t1);
            hipEventSynchronize(u1);
            float a = 0, b = 0;
            hipEventElapsedTime(&a, t0, t1);
            hipEventElapsedTime(&b, u0, u1);
            printf(" %5.1f", (a > b ? a : b) * 1000.0 / links - p_link_ticks / 100.0);
        }
        printf("\n");
    }
    return 0;
}
