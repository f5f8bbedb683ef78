// Probe (development aid): what a DEPENDENT launch costs on a high-priority stream while another stream keeps the
// chip full of long workgroups - the situation of the panel chains of a factorization beside its trailing updates.
//   big kernel : G workgroups of 256 threads, `lds` bytes of dynamic LDS (68 KB: two per CU, 100 KB: one per CU),
//                each lasting `t_big` us (a timed spin, no memory traffic)
//   tiny kernel: W workgroups of 256 threads, `lds_tiny` bytes, 10 us each
// k chain streams (high priority) each run `links` tiny kernels back to back; optionally every link of chain j waits
// for an event of the previous link of chain j^1 (the cross-stream hand-offs of the two chains of a film).
// Reported: (chain time / links) - 10 us = what one dependent launch costs.
// build: hipcc --offload-arch=gfx950 -O2 -o gap_probe gap_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

extern __shared__ char smem[];

__global__ __launch_bounds__(256) void spin_kernel(long long ticks, int touch) {
    if (touch && threadIdx.x == 0) smem[0] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

// 256 registers per lane (as the 128 x 128 update tile and the diagonal-block kernel have): one wave per SIMD
// of this kernel fills half of the SIMD's register file
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void spin_fat_kernel(long long ticks, double *out) {
    v4d acc[28];
    for (int i = 0; i < 28; ++i) acc[i] = v4d{1.0 * i, 0, 0, 0};
    const long long t0 = wall_clock64();
    double a = 1.0 + threadIdx.x * 1e-9;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 28; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

struct Case {
    const char *name;
    int chains;      // number of high-priority chain streams
    int big_grid;    // 0: no big kernel
    int big_lds;     // bytes
    double t_big;    // us per big workgroup
    bool fat;        // big kernel = MFMA burn with 256 registers
    bool cross;      // chains hand off to each other pairwise
    int tiny_wgs;
    int tiny_lds;
    bool chain_hi;   // chain streams high priority
};

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&spin_fat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    double *dout;
    hipMalloc(&dout, 64);
    const int kMaxChains = 6;
    hipStream_t big_s, ch_hi[kMaxChains], ch_lo[kMaxChains];
    hipStreamCreateWithPriority(&big_s, hipStreamNonBlocking, lo);
    for (int i = 0; i < kMaxChains; ++i) {
        hipStreamCreateWithPriority(&ch_hi[i], hipStreamNonBlocking, hi);
        hipStreamCreateWithPriority(&ch_lo[i], hipStreamNonBlocking, lo);
    }
    hipStream_t gate_s;
    hipStreamCreateWithFlags(&gate_s, hipStreamNonBlocking);
    hipEvent_t gate_ev;
    hipEventCreateWithFlags(&gate_ev, hipEventDisableTiming);
    const int links = 100;
    std::vector<hipEvent_t> evs(kMaxChains * links);
    for (auto &e : evs) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0[kMaxChains], t1[kMaxChains];
    for (int i = 0; i < kMaxChains; ++i) { hipEventCreate(&t0[i]); hipEventCreate(&t1[i]); }
    const long long tiny_ticks = 1000;   // 10 us at 100 MHz

    const Case cases[] = {
        {"chains alone", 1, 0, 0, 0, false, false, 1, 36 << 10, true},
        {"chains alone", 2, 0, 0, 0, false, false, 1, 36 << 10, true},
        {"chains alone", 4, 0, 0, 0, false, false, 1, 36 << 10, true},
        {"chains alone, cross", 4, 0, 0, 0, false, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB (2/CU) spin", 1, 2080, 68 << 10, 50, false, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB (2/CU) spin", 2, 2080, 68 << 10, 50, false, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB (2/CU) spin", 4, 2080, 68 << 10, 50, false, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB spin, cross", 4, 2080, 68 << 10, 50, false, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB spin, chains normal priority", 4, 2080, 68 << 10, 50, false, false, 1, 36 << 10, false},
        {"big 2080 x 50us, 100 KB (1/CU) spin", 4, 2080, 100 << 10, 50, false, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 100 KB spin, cross", 4, 2080, 100 << 10, 50, false, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB spin, tiny 66 KB", 4, 2080, 68 << 10, 50, false, false, 1, 66 << 10, true},
        {"big 2080 x 50us, 68 KB spin, tiny 16 wgs", 4, 2080, 68 << 10, 50, false, false, 16, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT (256 regs, mfma)", 1, 2080, 68 << 10, 50, true, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT (256 regs, mfma)", 4, 2080, 68 << 10, 50, true, false, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT, cross", 4, 2080, 68 << 10, 50, true, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT, tiny 16 wgs, cross", 4, 2080, 68 << 10, 50, true, true, 16, 36 << 10, true},
        {"big 2080 x 100us, 68 KB FAT, cross", 4, 2080, 68 << 10, 100, true, true, 1, 36 << 10, true},
        {"big 2080 x 25us, 68 KB FAT, cross", 4, 2080, 68 << 10, 25, true, true, 1, 36 << 10, true},
        {"big 400 x 50us (less than a round), FAT, cross", 4, 400, 68 << 10, 50, true, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT, cross, 2 chains", 2, 2080, 68 << 10, 50, true, true, 1, 36 << 10, true},
        {"big 2080 x 50us, 68 KB FAT, cross, 6 chains", 6, 2080, 68 << 10, 50, true, true, 1, 36 << 10, true},
    };
    for (const Case &c : cases) {
        hipDeviceSynchronize();
        hipStream_t *ch = c.chain_hi ? ch_hi : ch_lo;
        // everything is enqueued behind a gate (a 30 ms one-workgroup spin), so that the host's launch rate is not
        // what is measured
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), 0, gate_s, 3000000LL, 0);
        hipEventRecord(gate_ev, gate_s);
        hipStreamWaitEvent(big_s, gate_ev, 0);
        for (int j = 0; j < c.chains; ++j) hipStreamWaitEvent(ch[j], gate_ev, 0);
        // big kernels: enough launches to outlast the chains (estimated generously), enqueued first
        int big_launches = 0;
        if (c.big_grid) {
            const double per_launch_us = c.t_big * ((c.big_grid + 511) / 512);
            big_launches = static_cast<int>(links * 120.0 / per_launch_us) + 2;
            for (int i = 0; i < big_launches; ++i) {
                if (c.fat)
                    hipLaunchKernelGGL(spin_fat_kernel, dim3(c.big_grid), dim3(256), c.big_lds, big_s,
                                       static_cast<long long>(c.t_big * 100), dout);
                else
                    hipLaunchKernelGGL(spin_kernel, dim3(c.big_grid), dim3(256), c.big_lds, big_s,
                                       static_cast<long long>(c.t_big * 100), 0);
            }
        }
        for (int j = 0; j < c.chains; ++j) hipEventRecord(t0[j], ch[j]);
        for (int l = 0; l < links; ++l) {
            for (int j = 0; j < c.chains; ++j) {
                if (c.cross && l > 0) hipStreamWaitEvent(ch[j], evs[(j ^ 1) * links + l - 1], 0);
                hipLaunchKernelGGL(spin_kernel, dim3(c.tiny_wgs), dim3(256), c.tiny_lds, ch[j], tiny_ticks, 0);
                if (c.cross) hipEventRecord(evs[j * links + l], ch[j]);
            }
        }
        for (int j = 0; j < c.chains; ++j) hipEventRecord(t1[j], ch[j]);
        for (int j = 0; j < c.chains; ++j) hipEventSynchronize(t1[j]);
        double worst = 0, best = 1e9;
        for (int j = 0; j < c.chains; ++j) {
            float ms = 0;
            hipEventElapsedTime(&ms, t0[j], t1[j]);
            const double per = ms * 1000.0 / links - 10.0;
            worst = per > worst ? per : worst;
            best = per < best ? per : best;
        }
        const bool big_still_running = c.big_grid && hipStreamQuery(big_s) == hipErrorNotReady;
        printf("%-56s chains %d : dependent launch %6.1f .. %6.1f us%s\n", c.name, c.chains, best, worst,
               c.big_grid ? (big_still_running ? "" : "   (big kernels ended first!)") : "");
        hipDeviceSynchronize();
    }
    return 0;
}
