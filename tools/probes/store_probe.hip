// HBM write-stream probe (development aid): which store form / work distribution fills a multi-GB buffer
// fastest on this box?  Backs the choice of store instructions in q_assemble_kernel / system_assemble_kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_probe.hip -o tools/probes/store_probe && tools/probes/store_probe [GB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum Form { PLAIN16, NT16, PLAIN4, NT4, SC1_16, SC01_16, PLAIN8, NT8 };

template <int FORM>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v) {
    if (FORM == PLAIN16) *p = v;
    else if (FORM == NT16) __builtin_nontemporal_store(v, p);
    else if (FORM == SC1_16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (FORM == SC01_16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// grid-stride: consecutive workgroups write adjacent 4 KiB pieces; the whole chip sweeps one window
template <int FORM>
__global__ void fill_gridstride16(u32x4 *__restrict__ dst, size_t count16) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count16; i += stride)
        store16<FORM>(dst + i, v);
}

// 4 bytes per lane: 256 B per wave-instruction
template <bool NT>
__global__ void fill_gridstride4(unsigned *__restrict__ dst, size_t count4) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count4; i += stride) {
        if (NT) __builtin_nontemporal_store(0x3f800000u, dst + i);
        else dst[i] = 0x3f800000u;
    }
}

template <bool NT>
__global__ void fill_gridstride8(unsigned long long *__restrict__ dst, size_t count8) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count8; i += stride) {
        if (NT) __builtin_nontemporal_store(0x3f8000003f800000ull, dst + i);
        else dst[i] = 0x3f8000003f800000ull;
    }
}

// chunked: every workgroup owns one contiguous chunk and walks it front to back
template <int FORM>
__global__ void fill_chunked16(u32x4 *__restrict__ dst, size_t count16) {
    const size_t per = (count16 + gridDim.x - 1) / gridDim.x;
    const size_t b = per * blockIdx.x, e = (b + per < count16) ? b + per : count16;
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) store16<FORM>(dst + i, v);
}

// the shape of q_assemble_kernel: a workgroup owns a strip of 16 matrix rows and sweeps the columns;
// a wave-instruction stores 1 KiB of one row, then the same column range of the next row
template <int FORM, int TR>
__global__ void fill_strips16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16) {
    const size_t r0 = static_cast<size_t>(blockIdx.x) * TR;
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t c = threadIdx.x; c < row16; c += blockDim.x)
#pragma unroll
        for (int r = 0; r < TR; ++r)
            if (r0 + r < n_rows) store16<FORM>(dst + (r0 + r) * row16 + c, v);
}

// row-major order inside the strip: finish 4 KiB x k of one row before the next row (wave <-> row quarter)
template <int FORM, int TR, int CH>
__global__ void fill_strips_rowchunks16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16) {
    const size_t r0 = static_cast<size_t>(blockIdx.x) * TR;
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t c0 = 0; c0 < row16; c0 += static_cast<size_t>(blockDim.x) * CH)
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const size_t c = c0 + static_cast<size_t>(k) * blockDim.x + threadIdx.x;
                if (r0 + r < n_rows && c < row16) store16<FORM>(dst + (r0 + r) * row16 + c, v);
            }
}

// one shot: no loop; a workgroup writes U consecutive 4 KiB pieces (U stores per lane, all issued at once)
template <int FORM, int U>
__global__ void fill_oneshot16(u32x4 *__restrict__ dst, size_t count16) {
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    const size_t base = (static_cast<size_t>(blockIdx.x) * U) * blockDim.x + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + static_cast<size_t>(u) * blockDim.x;
        if (i < count16) store16<FORM>(dst + i, v);
    }
}

// one matrix row at a time per workgroup: workgroup b owns rows b, b + G, ...; all its threads sweep the row
template <int FORM>
__global__ void fill_rows_seq16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16) {
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x)
        for (size_t c = threadIdx.x; c < row16; c += blockDim.x) store16<FORM>(dst + r * row16 + c, v);
}
// TR rows at a time, but each WAVE writes one row (4 KiB contiguous per wave and step: 4 stores per lane)
template <int FORM>
__global__ void fill_wave_rows16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16) {
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (size_t r = static_cast<size_t>(blockIdx.x) * nw + wave; r < n_rows; r += static_cast<size_t>(gridDim.x) * nw)
        for (size_t c0 = 0; c0 < row16; c0 += 256)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t c = c0 + 64 * k + lane;
                if (c < row16) store16<FORM>(dst + r * row16 + c, v);
            }
}

// band form: workgroup = (column band of BW 16-byte columns, row group); all workgroups sweep their rows top to bottom
// in step, so the chip fills a window of `groups` consecutive rows at a time.  STORERS waves of the workgroup do the
// storing (the others would only compute): STORERS = waves of the workgroup -> every wave stores its own share.
template <int THREADS, int STORERS>
__global__ __launch_bounds__(THREADS) void fill_bands16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16, int bands, int groups) {
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    const int band = blockIdx.x % bands, grp = blockIdx.x / bands;
    const size_t bw = (row16 + bands - 1) / bands;                 // 16-byte columns per band
    const size_t c0 = band * bw, c1 = (c0 + bw < row16) ? c0 + bw : row16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= STORERS) return;
    const size_t rows_per = (n_rows + groups - 1) / groups;
    // row-interleaved groups: group g takes rows g, g + groups, ... -> at any time the chip is on `groups` ADJACENT rows
    for (size_t k = 0; k < rows_per; ++k) {
        const size_t r = k * groups + grp;
        if (r >= n_rows) break;
        for (size_t c = c0 + wave * 64 + lane; c < c1; c += STORERS * 64) dst[r * row16 + c] = v;
    }
}

// address-ordered tasks: task t = (row group of R rows, chunk of CH x 4 KiB of those rows), tasks in address order
// (row group major), dealt round-robin to a persistent grid: workgroup b runs tasks b, b + G, ...; a task = R rows x
// (CH x 256 lanes x 16 B) written row by row.  R = 1: the grid-stride fill in row-aligned pieces.
template <int R, int CH>
__global__ __launch_bounds__(256) void fill_tasks16(u32x4 *__restrict__ dst, size_t n_rows, size_t row16) {
    const u32x4 v = {0x3f800000u, 0u, 0x3f800000u, 0u};
    const size_t chunk = static_cast<size_t>(CH) * 256;                       // 16-byte columns per task
    const size_t chunks = (row16 + chunk - 1) / chunk, groups = (n_rows + R - 1) / R;
    for (size_t t = blockIdx.x; t < groups * chunks; t += gridDim.x) {
        const size_t g = t / chunks, c0 = (t % chunks) * chunk;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t row = g * R + r;
            if (row >= n_rows) break;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const size_t c = c0 + static_cast<size_t>(k) * 256 + threadIdx.x;
                if (c < row16) dst[row * row16 + c] = v;
            }
        }
    }
}

template <typename F>
double time_ms(F launch, int reps = 7) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 5.05;
    const size_t n_rows = 25117, row16 = (static_cast<size_t>(gb * 1e9) / n_rows / 16 + 63) / 64 * 64;
    const size_t bytes = n_rows * row16 * 16, count16 = bytes / 16;
    void *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("buffer %.3f GB (%zu rows x %zu B)\n", bytes / 1e9, n_rows, row16 * 16);
    auto report = [&](const char *name, double ms) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    u32x4 *d = static_cast<u32x4 *>(buf);
    for (int thr : {256, 512, 1024})
        for (int wgs : {128, 256, 512, 768}) {
            char nm[128];
            snprintf(nm, sizeof nm, "grid-stride 16B plain, %d WG x %d", wgs, thr);
            report(nm, time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<PLAIN16>, dim3(wgs), dim3(thr), 0, 0, d, count16); }));
        }
    report("hipMemsetAsync", time_ms([&] { hipMemsetAsync(buf, 0, bytes, 0); }));
    for (int wgs : {256, 512}) {
        char nm[128];
#define TASKS(R, CH)                                                                                                    \
    snprintf(nm, sizeof nm, "tasks in address order: %d rows x %d KiB, %d WG x 256", R, 4 * CH, wgs);                 \
    report(nm, time_ms([&] { hipLaunchKernelGGL((fill_tasks16<R, CH>), dim3(wgs), dim3(256), 0, 0, d, n_rows, row16); }));
        TASKS(1, 1) TASKS(1, 4) TASKS(2, 1) TASKS(2, 4) TASKS(4, 1) TASKS(4, 4) TASKS(8, 1) TASKS(8, 4) TASKS(16, 1) TASKS(16, 4)
        TASKS(2, 8) TASKS(4, 8) TASKS(4, 16)
#undef TASKS
    }
    for (int bands : {13}) {
        const int groups = 256 / bands;
        char nm[128];
        snprintf(nm, sizeof nm, "bands: %d x %d groups, 256 thr, 4 storing waves", bands, groups);
        report(nm, time_ms([&] { hipLaunchKernelGGL((fill_bands16<256, 4>), dim3(bands * groups), dim3(256), 0, 0, d, n_rows, row16, bands, groups); }));
        snprintf(nm, sizeof nm, "bands: %d x %d groups, 1024 thr, 4 storing waves", bands, groups);
        report(nm, time_ms([&] { hipLaunchKernelGGL((fill_bands16<1024, 4>), dim3(bands * groups), dim3(1024), 0, 0, d, n_rows, row16, bands, groups); }));
        snprintf(nm, sizeof nm, "bands: %d x %d groups, 1024 thr, 16 storing waves", bands, groups);
        report(nm, time_ms([&] { hipLaunchKernelGGL((fill_bands16<1024, 16>), dim3(bands * groups), dim3(1024), 0, 0, d, n_rows, row16, bands, groups); }));
        snprintf(nm, sizeof nm, "bands: %d x %d groups, 512 thr, 8 storing waves", bands, groups);
        report(nm, time_ms([&] { hipLaunchKernelGGL((fill_bands16<512, 8>), dim3(bands * groups), dim3(512), 0, 0, d, n_rows, row16, bands, groups); }));
        snprintf(nm, sizeof nm, "bands: %d x %d groups (2 WG/CU), 256 thr, 4 storing waves", bands, 2 * groups);
        report(nm, time_ms([&] { hipLaunchKernelGGL((fill_bands16<256, 4>), dim3(bands * 2 * groups), dim3(256), 0, 0, d, n_rows, row16, bands, 2 * groups); }));
    }
    for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride 16B plain, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<PLAIN16>, dim3(wgs), dim3(256), 0, 0, d, count16); }));
        snprintf(nm, sizeof nm, "grid-stride 16B nt, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<NT16>, dim3(wgs), dim3(256), 0, 0, d, count16); }));
    }
    report("grid-stride 16B plain, 2048 WG x 1024", time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<PLAIN16>, dim3(2048), dim3(1024), 0, 0, d, count16); }));
    report("grid-stride 16B nt, 512 WG x 1024", time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<NT16>, dim3(512), dim3(1024), 0, 0, d, count16); }));
    report("grid-stride 16B sc1, 2048 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<SC1_16>, dim3(2048), dim3(256), 0, 0, d, count16); }));
    report("grid-stride 16B sc0 sc1, 2048 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<SC01_16>, dim3(2048), dim3(256), 0, 0, d, count16); }));
    report("grid-stride 4B plain, 4096 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride4<false>, dim3(4096), dim3(256), 0, 0, (unsigned *)buf, bytes / 4); }));
    report("grid-stride 4B nt, 4096 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride4<true>, dim3(4096), dim3(256), 0, 0, (unsigned *)buf, bytes / 4); }));
    report("grid-stride 8B plain, 4096 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride8<false>, dim3(4096), dim3(256), 0, 0, (unsigned long long *)buf, bytes / 8); }));
    report("grid-stride 8B nt, 4096 WG x 256", time_ms([&] { hipLaunchKernelGGL(fill_gridstride8<true>, dim3(4096), dim3(256), 0, 0, (unsigned long long *)buf, bytes / 8); }));
    for (int wgs : {256, 512, 2048, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "chunked 16B plain, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_chunked16<PLAIN16>, dim3(wgs), dim3(256), 0, 0, d, count16); }));
        snprintf(nm, sizeof nm, "chunked 16B nt, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_chunked16<NT16>, dim3(wgs), dim3(256), 0, 0, d, count16); }));
    }
    const unsigned strips16 = (n_rows + 15) / 16, strips8 = (n_rows + 7) / 8, strips4 = (n_rows + 3) / 4, strips32 = (n_rows + 31) / 32;
    report("strips of 16 rows (q_assemble shape) 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 16>), dim3(strips16), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows (q_assemble shape) 16B nt", time_ms([&] { hipLaunchKernelGGL((fill_strips16<NT16, 16>), dim3(strips16), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows 16B sc0 sc1", time_ms([&] { hipLaunchKernelGGL((fill_strips16<SC01_16, 16>), dim3(strips16), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 8 rows 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 8>), dim3(strips8), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 8 rows 16B nt", time_ms([&] { hipLaunchKernelGGL((fill_strips16<NT16, 8>), dim3(strips8), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 4 rows 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 4>), dim3(strips4), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 4 rows 16B nt", time_ms([&] { hipLaunchKernelGGL((fill_strips16<NT16, 4>), dim3(strips4), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 32 rows 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 32>), dim3(strips32), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows, 512 threads, 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 16>), dim3(strips16), dim3(512), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows, 1024 threads, 16B plain", time_ms([&] { hipLaunchKernelGGL((fill_strips16<PLAIN16, 16>), dim3(strips16), dim3(1024), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows, 16 KiB of a row at a time, plain", time_ms([&] { hipLaunchKernelGGL((fill_strips_rowchunks16<PLAIN16, 16, 4>), dim3(strips16), dim3(256), 0, 0, d, n_rows, row16); }));
    report("strips of 16 rows, 16 KiB of a row at a time, nt", time_ms([&] { hipLaunchKernelGGL((fill_strips_rowchunks16<NT16, 16, 4>), dim3(strips16), dim3(256), 0, 0, d, n_rows, row16); }));
    {
        auto one = [&](auto kern, int U, int threads, const char *nm) {
            const unsigned grid = static_cast<unsigned>((count16 + static_cast<size_t>(U) * threads - 1) / (static_cast<size_t>(U) * threads));
            char name[128];
            snprintf(name, sizeof name, "one shot 16B %s, %d stores/lane, %d threads (%u WG)", nm, U, threads, grid);
            report(name, time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, count16); }));
        };
        one(fill_oneshot16<PLAIN16, 1>, 1, 256, "plain");
        one(fill_oneshot16<PLAIN16, 2>, 2, 256, "plain");
        one(fill_oneshot16<PLAIN16, 4>, 4, 256, "plain");
        one(fill_oneshot16<PLAIN16, 8>, 8, 256, "plain");
        one(fill_oneshot16<PLAIN16, 16>, 16, 256, "plain");
        one(fill_oneshot16<PLAIN16, 4>, 4, 512, "plain");
        one(fill_oneshot16<PLAIN16, 4>, 4, 1024, "plain");
        one(fill_oneshot16<PLAIN16, 1>, 1, 1024, "plain");
        one(fill_oneshot16<NT16, 4>, 4, 256, "nt");
        one(fill_oneshot16<SC1_16, 4>, 4, 256, "sc1");
    }
    for (int wgs : {32768, 65536, 131072}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride 16B plain, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<PLAIN16>, dim3(wgs), dim3(256), 0, 0, d, count16); }));
    }
    for (int wgs : {256, 512}) {
        for (int thr : {256, 512, 1024}) {
            char nm[128];
            snprintf(nm, sizeof nm, "grid-stride 16B plain, %d WG x %d", wgs, thr);
            report(nm, time_ms([&] { hipLaunchKernelGGL(fill_gridstride16<PLAIN16>, dim3(wgs), dim3(thr), 0, 0, d, count16); }));
        }
    }
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        for (int thr : {256, 1024}) {
            char nm[128];
            snprintf(nm, sizeof nm, "one row at a time per WG, %d WG x %d", wgs, thr);
            report(nm, time_ms([&] { hipLaunchKernelGGL(fill_rows_seq16<PLAIN16>, dim3(wgs), dim3(thr), 0, 0, d, n_rows, row16); }));
        }
    }
    for (int wgs : {256, 512, 1024, 2048}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one row per wave, %d WG x 256", wgs);
        report(nm, time_ms([&] { hipLaunchKernelGGL(fill_wave_rows16<PLAIN16>, dim3(wgs), dim3(256), 0, 0, d, n_rows, row16); }));
    }
    // hipMemset as the runtime's own fill
    report("hipMemsetAsync", time_ms([&] { hipMemsetAsync(buf, 0, bytes, 0); }));
    hipFree(buf);
    return 0;
}
