// Q-assembly probe (development aid): where does the time of q_assemble_kernel go -- its store stream, its FP64
// work, or their overlap?  The production strip kernel's loop with the arithmetic swapped out (store only, seed
// only), with other strip heights / columns per lane, and with the number of resident workgroups capped.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/q_probe.hip -o tools/probes/q_probe && tools/probes/q_probe [K]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../superscreen_amd/csrc/common.hpp"

using namespace ssa;

enum Mode { FULL = 0, STORE_ONLY = 1, SEED_ONLY = 2, COMPUTE_ONLY = 3 };

__host__ __device__ inline void rows_of(int64_t n, int64_t groups, int64_t b, int64_t *i0, int *h) {
    const int64_t base = n / groups, extra = n % groups;
    *i0 = b * base + (b < extra ? b : extra);
    *h = static_cast<int>(base + (b < extra ? 1 : 0));
}

// CPL columns per lane (2: one 16-byte store per row and lane; 4: two adjacent ones, 2 KiB per wave and row)
template <int MODE, int TR, int CPL, bool LDSROWS = false, int STAGGER = 0>
__global__ __launch_bounds__(256) void q_kernel(const double *__restrict__ xy, const double *__restrict__ w, int64_t n,
                                                double *__restrict__ Q, int64_t ldq, double *__restrict__ rowsum) {
    extern __shared__ char pad_lds[];   // occupancy cap only
    __shared__ double s_part[4][TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t i0;
    int h;
    rows_of(n, gridDim.x, blockIdx.x, &i0, &h);
    __shared__ double2 s_xy[TR];
    double xi[LDSROWS ? 1 : TR], yi[LDSROWS ? 1 : TR], acc[TR];
    if (LDSROWS) {
        if (tid < TR) {
            const int64_t i = (tid < h) ? i0 + tid : i0;
            s_xy[tid] = *reinterpret_cast<const double2 *>(xy + 2 * i);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        if (!LDSROWS) {
            const int64_t i = (r < h) ? i0 + r : i0;
            xi[r] = xy[2 * i];
            yi[r] = xy[2 * i + 1];
        }
        acc[r] = 0.0;
    }
    // STAGGER: workgroup b starts its sweep STAGGER x 512 columns further right than workgroup b - 1 (and wraps), so
    // that the workgroups resident at a time do not all write the same column range of their rows
    const int64_t span = (n + CPL * 256 - 1) / (CPL * 256) * (CPL * 256);
    const int64_t shift = STAGGER ? (static_cast<int64_t>(blockIdx.x) * STAGGER * CPL * 256) % span : 0;
    for (int64_t jj = CPL * tid; jj < span; jj += CPL * 256) {
        int64_t j = jj + shift;
        if (j >= span) j -= span;
        if (j >= n) continue;
        double xj[CPL], yj[CPL], wj[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool has = j + c < n;
            xj[c] = has ? xy[2 * (j + c)] : 0.0;
            yj[c] = has ? xy[2 * (j + c) + 1] : 0.0;
            wj[c] = has ? w[j + c] : 0.0;
        }
        // row-invariant scalars kept out of SGPRs: the store address advances by one row per step, the diagonal is
        // found by the lane's own distance to the strip's first row
        double *qp = Q + i0 * ldq + j;
        const int dj = static_cast<int>(j - i0);   // row r is on the diagonal of column j + c when dj + c == r
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (r < h) {
                double q[CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (MODE == STORE_ONLY) {
                        q[c] = xj[c];
                    } else {
                        double2 pr;
                        if (LDSROWS) {
                            int off = r * 16;   // an LDS read per use: hidden from loop-invariant code motion
                            asm volatile("" : "+v"(off));
                            pr = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_xy) + off);
                        } else {
                            pr = double2{xi[r], yi[r]};
                        }
                        const double dx = pr.x - xj[c], dy = pr.y - yj[c];
                        const double r2 = __builtin_fma(dx, dx, dy * dy);
                        q[c] = (MODE == SEED_ONLY) ? __builtin_amdgcn_rsq(r2) : inv_r3_over_4pi(r2);
                        q[c] = (dj + c == r || j + c >= n) ? 0.0 : q[c];
                        acc[r] = __builtin_fma(q[c], wj[c], acc[r]);
                    }
                }
                if (MODE != COMPUTE_ONLY) {
#pragma unroll
                    for (int c = 0; c < CPL; c += 2) {
                        double2 v;
                        v.x = -q[c];
                        v.y = -q[c + 1];
                        *reinterpret_cast<double2 *>(qp + c) = v;
                    }
                }
                qp += ldq;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) s_part[wave][r] = s;
    }
    __syncthreads();
    if (tid < h) rowsum[i0 + tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
}

static std::vector<double> ring_points(int K);
template <typename F>
static double time_ms(F f, int reps = 7);

// FLAT order: the whole device writes ONE contiguous window per iteration (gridDim.x x 4 KiB), workgroup b the b-th
// 4 KiB of it -- the shape of a grid-stride fill, the only store shape that reaches the memset rate
// (store_probe.hip).  A lane owns two adjacent elements per iteration; row and column follow the flat index
// incrementally; ld % 128 == 0, so that a wave's 128 elements lie in one row (row data by scalar loads, row-sum
// partial one value per wave and iteration, partial[row][column / 128]).
// ROWSUM: 0 none, 1 butterfly (ds_bpermute), 2 on the matrix pipe (ones-matrix products)
template <int MODE, int ROWSUM, int PREFETCH>
__global__ __launch_bounds__(256) void q_flat_kernel(const double *__restrict__ xy, const double *__restrict__ w, int64_t n,
                                                     double *__restrict__ Q, int64_t ld, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t total = n * ld, step = static_cast<int64_t>(gridDim.x) * 512;
    int64_t e = static_cast<int64_t>(blockIdx.x) * 512 + 2 * tid;
    int i = static_cast<int>(e / ld), j = static_cast<int>(e - static_cast<int64_t>(i) * ld);
    const int qs = static_cast<int>(step / ld), rs = static_cast<int>(step - static_cast<int64_t>(qs) * ld);
    const int nn = static_cast<int>(n), cpr = static_cast<int>(ld >> 7);
    auto fetch = [&](int jj, double2 &p0, double2 &p1, double2 &ww) {
        const int jc = jj + 1 < nn ? jj : (nn - 2);   // pad columns read a valid pair (values unused)
        p0 = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(jc));
        p1 = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(jc) + 2);
        ww = *reinterpret_cast<const double2 *>(w + jc);
    };
    double2 p0, p1, ww;
    if (e < total) fetch(j, p0, p1, ww);
    while (e < total) {
        int i2 = i + qs, j2 = j + rs;
        if (j2 >= static_cast<int>(ld)) {
            j2 -= static_cast<int>(ld);
            ++i2;
        }
        double2 n0 = p0, n1 = p1, nw = ww;
        if (PREFETCH && e + step < total) fetch(j2, n0, n1, nw);
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const double2 pi = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(iu));
        double q[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double2 pj = c ? p1 : p0;
            if (MODE == STORE_ONLY) {
                q[c] = pj.x;
            } else {
                const double dx = pi.x - pj.x, dy = pi.y - pj.y;
                const double r2 = __builtin_fma(dx, dx, dy * dy);
                q[c] = inv_r3_over_4pi(r2);
                q[c] = (j + c == iu || j + c >= nn) ? 0.0 : q[c];
            }
        }
        double2 v;
        v.x = -q[0];
        v.y = -q[1];
        *reinterpret_cast<double2 *>(Q + e) = v;
        if (ROWSUM == 1) {
            const double s = wave_sum(__builtin_fma(q[0], ww.x, q[1] * ww.y));
            if (lane == 0) partial[static_cast<int64_t>(iu) * cpr + (j >> 7)] = s;
        } else if (ROWSUM == 2) {
            // total over the wave on the (idle) matrix pipe: D1[a][b] = sum_k A[a][k], A[a][k] = lane (a + 16 k);
            // then D2 += ones x B_r with B_r[k][b] = register r of lane (b + 16 k) of D1: every lane holds the total
            typedef double d4 __attribute__((ext_vector_type(4)));
            const double sl = __builtin_fma(q[0], ww.x, q[1] * ww.y);
            d4 z = {0.0, 0.0, 0.0, 0.0};
            const d4 d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sl, 1.0, z, 0, 0, 0);
            d4 d2 = z;
#pragma unroll
            for (int r = 0; r < 4; ++r) d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, d1[r], d2, 0, 0, 0);
            if (lane == 0) partial[static_cast<int64_t>(iu) * cpr + (j >> 7)] = d2[0];
        }
        if (!PREFETCH && e + step < total) fetch(j2, n0, n1, nw);
        p0 = n0;
        p1 = n1;
        ww = nw;
        e += step;
        i = i2;
        j = j2;
    }
}

template <int MODE, int ROWSUM, int PREFETCH>
static void run_flat(const char *name, const double *xy, const double *w, int64_t n, double *Q, int64_t ld, double *partial,
                     int wgs) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((q_flat_kernel<MODE, ROWSUM, PREFETCH>), dim3(wgs), dim3(256), 0, 0, xy, w, n, Q, ld, partial); });
    const double gb = static_cast<double>(n) * n * 8 / 1e9;
    printf("flat %-12s rowsum=%d prefetch=%d wgs=%4d ld=%lld: %7.3f ms  %6.0f GB/s (n^2 x 8 bytes)\n", name, ROWSUM, PREFETCH, wgs,
           static_cast<long long>(ld), ms, gb / ms * 1e3);
}

// FLAT order, second form: a ring of D prefetched column sets per lane, so that D - 1 stores of a wave stay in
// flight while it waits for a load (loads and stores retire through ONE in-order counter: with the next iteration's
// loads issued right before they are used, every iteration waits for its own previous store to reach the L2).
// Row sums: the wave's 64 per-lane partials are summed on the matrix pipe, two ones-matrix products
// (16x16x4: D[a][b] = sum_k A[a][k]; then the four registers of a lane added, and the same product again).
template <int MODE, int ROWSUM, int D>
__global__ __launch_bounds__(256) void q_flat2_kernel(const double *__restrict__ xy, const double *__restrict__ w, int64_t n,
                                                      double *__restrict__ Q, int64_t ld, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t total = n * ld, step = static_cast<int64_t>(gridDim.x) * 512;
    int64_t e = static_cast<int64_t>(blockIdx.x) * 512 + 2 * tid;
    const int ldi = static_cast<int>(ld), nn = static_cast<int>(n), cpr = static_cast<int>(ld >> 7);
    int i = static_cast<int>(e / ld), j = static_cast<int>(e - static_cast<int64_t>(i) * ld);
    const int qs = static_cast<int>(step / ld), rs = static_cast<int>(step - static_cast<int64_t>(qs) * ld);
    // head of the ring: iteration `ahead` of this lane
    int64_t eh = e;
    int jh = j;
    double2 P0[D], P1[D], WW[D];
    auto fetch = [&](int d) {
        if (eh < total) {
            const int jc = jh + 1 < nn ? jh : (nn - 2);
            P0[d] = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(jc));
            P1[d] = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(jc) + 2);
            WW[d] = *reinterpret_cast<const double2 *>(w + jc);
        }
        eh += step;
        jh += rs;
        if (jh >= ldi) jh -= ldi;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    typedef double d4 __attribute__((ext_vector_type(4)));
    while (true) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (e >= total) return;
            const int iu = __builtin_amdgcn_readfirstlane(i);
            const double2 pi = *reinterpret_cast<const double2 *>(xy + 2 * static_cast<int64_t>(iu));
            double q[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double2 pj = c ? P1[d] : P0[d];
                if (MODE == STORE_ONLY) {
                    q[c] = pj.x;
                } else {
                    const double dx = pi.x - pj.x, dy = pi.y - pj.y;
                    const double r2 = __builtin_fma(dx, dx, dy * dy);
                    q[c] = inv_r3_over_4pi(r2);
                    q[c] = (j + c == iu || j + c >= nn) ? 0.0 : q[c];
                }
            }
            double2 v;
            v.x = -q[0];
            v.y = -q[1];
            *reinterpret_cast<double2 *>(Q + e) = v;
            if (ROWSUM) {
                const double sl = __builtin_fma(q[0], WW[d].x, q[1] * WW[d].y);
                const d4 z = {0.0, 0.0, 0.0, 0.0};
                const d4 d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sl, 1.0, z, 0, 0, 0);
                const double t = (d1[0] + d1[1]) + (d1[2] + d1[3]);
                const d4 d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, z, 0, 0, 0);
                if (lane == 0) partial[static_cast<int64_t>(iu) * cpr + (j >> 7)] = d2[0];
            }
            fetch(d);
            e += step;
            i += qs;
            j += rs;
            if (j >= ldi) {
                j -= ldi;
                ++i;
            }
        }
    }
}

template <int MODE, int ROWSUM, int D>
static void run_flat2(const char *name, const double *xy, const double *w, int64_t n, double *Q, int64_t ld, double *partial,
                      int wgs) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((q_flat2_kernel<MODE, ROWSUM, D>), dim3(wgs), dim3(256), 0, 0, xy, w, n, Q, ld, partial); });
    const double gb = static_cast<double>(n) * n * 8 / 1e9;
    printf("flat2 %-11s rowsum=%d ring=%2d wgs=%4d ld=%lld: %7.3f ms  %6.0f GB/s (n^2 x 8 bytes)\n", name, ROWSUM, D, wgs,
           static_cast<long long>(ld), ms, gb / ms * 1e3);
}

static std::vector<double> ring_points(int K) {
    std::vector<double> p;
    p.push_back(0.0);
    p.push_back(0.0);
    for (int k = 1; k <= K; ++k)
        for (int m = 0; m < 6 * k; ++m) {
            const double rad = 5.5 * k / K, phi = 2.0 * M_PI * m / (6.0 * k) + 0.1 * k;
            p.push_back(rad * cos(phi));
            p.push_back(rad * sin(phi));
        }
    return p;
}

template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int MODE, int TR, int CPL, bool LDSROWS = false, int STAGGER = 0>
static void run(const char *name, const double *xy, const double *w, int64_t n, double *Q, int64_t ld, double *rs,
                int groups_per_cu, size_t lds_pad) {
    if (lds_pad > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void *>(&q_kernel<MODE, TR, CPL, LDSROWS, STAGGER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds_pad));
    const int64_t slots = 256 * groups_per_cu;
    const int64_t groups = slots * ((n + static_cast<int64_t>(TR) * slots - 1) / (static_cast<int64_t>(TR) * slots));
    const double ms = time_ms([&] { hipLaunchKernelGGL((q_kernel<MODE, TR, CPL, LDSROWS, STAGGER>), dim3(groups), dim3(256), lds_pad, 0, xy, w, n, Q, ld, rs); });
    const double gb = static_cast<double>(n) * n * 8 / 1e9;
    printf("%-30s stagger=%d lds=%d cap=%3zuK TR=%2d cpl=%d wg/cu=%d groups=%5lld: %7.3f ms  %6.0f GB/s\n", name, STAGGER, int(LDSROWS), lds_pad >> 10, TR, CPL, groups_per_cu,
           static_cast<long long>(groups), ms, gb / ms * 1e3);
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 91;
    std::vector<double> p = ring_points(K);
    const int64_t n = static_cast<int64_t>(p.size() / 2), ld = (n + 15) / 16 * 16, ldf = (n + 127) / 128 * 128;
    std::vector<double> wh(n, 1e-3);
    double *xy, *w, *Q, *rs;
    hipMalloc(&xy, p.size() * 8);
    hipMalloc(&w, n * 8);
    hipMalloc(&rs, n * 8);
    hipMalloc(&Q, static_cast<size_t>(n) * ldf * 8);
    double *partial;
    hipMalloc(&partial, static_cast<size_t>(n) * (ldf / 128) * 8);
    hipMemcpy(xy, p.data(), p.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(w, wh.data(), n * 8, hipMemcpyHostToDevice);
    printf("n = %lld (K = %d), %.2f GB\n", static_cast<long long>(n), K, static_cast<double>(n) * n * 8 / 1e9);
    const double fill = time_ms([&] { hipMemsetAsync(Q, 0, static_cast<size_t>(n) * ld * 8, 0); });
    printf("hipMemsetAsync: %.3f ms %.0f GB/s\n", fill, static_cast<double>(n) * ld * 8 / 1e9 / fill * 1e3);
    // the library's own kernel through the C ABI, on the same buffers
    typedef int (*q_fn)(const double *, const double *, const double *, int64_t, void *, int64_t, int, double *, void *);
    void *lib = dlopen("superscreen_amd/lib/libsuperscreen_hip.so", RTLD_NOW);
    q_fn q_assemble = lib ? reinterpret_cast<q_fn>(dlsym(lib, "ssa_q_assemble")) : nullptr;
    if (q_assemble) {
        for (int rep = 0; rep < 3; ++rep) {
            const double ms = time_ms([&] { q_assemble(xy, w, w, n, Q, ld, 1, rs, nullptr); });
            printf("library ssa_q_assemble (float64, ld=%lld): %7.3f ms  %6.0f GB/s\n", static_cast<long long>(ld), ms,
                   static_cast<double>(n) * n * 8 / 1e9 / ms * 1e3);
            run<FULL, 24, 2, true, 1>("probe copy of it", xy, w, n, Q, ld, rs, 5, 0);
        }
    } else {
        printf("library not loaded: %s\n", dlerror());
    }
    for (int rep = 0; rep < 1; ++rep) {
        for (int wgs : {256, 512}) {
            run_flat2<STORE_ONLY, 0, 4>("store only", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<STORE_ONLY, 0, 8>("store only", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<STORE_ONLY, 0, 12>("store only", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<FULL, 0, 8>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<FULL, 0, 12>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<FULL, 1, 8>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat2<FULL, 1, 12>("full", xy, w, n, Q, ldf, partial, wgs);
        }
        for (int wgs : {256}) {
            run_flat<STORE_ONLY, 0, 1>("store only", xy, w, n, Q, ldf, partial, wgs);
            run_flat<FULL, 0, 1>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat<FULL, 0, 0>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat<FULL, 1, 1>("full", xy, w, n, Q, ldf, partial, wgs);
            run_flat<FULL, 2, 1>("full", xy, w, n, Q, ldf, partial, wgs);
        }
        run<FULL, 24, 2, true, 1>("production", xy, w, n, Q, ld, rs, 5, 0);
        run<FULL, 16, 2>("round 3", xy, w, n, Q, ld, rs, 7, 0);
        run<STORE_ONLY, 16, 2>("store only", xy, w, n, Q, ld, rs, 7, 0);
        run<STORE_ONLY, 16, 2, false, 1>("store only", xy, w, n, Q, ld, rs, 7, 0);
        run<STORE_ONLY, 16, 2, false, 3>("store only", xy, w, n, Q, ld, rs, 7, 0);
        run<STORE_ONLY, 16, 2, false, 7>("store only", xy, w, n, Q, ld, rs, 7, 0);
        run<STORE_ONLY, 32, 2, true, 1>("store only", xy, w, n, Q, ld, rs, 4, 0);
        run<STORE_ONLY, 32, 2, true, 5>("store only", xy, w, n, Q, ld, rs, 4, 0);
        run<STORE_ONLY, 8, 2, false, 1>("store only", xy, w, n, Q, ld, rs, 8, 0);
        run<FULL, 16, 2, false, 1>("full", xy, w, n, Q, ld, rs, 7, 0);
        run<FULL, 16, 2, false, 3>("full", xy, w, n, Q, ld, rs, 7, 0);
        run<FULL, 24, 2, false, 1>("full", xy, w, n, Q, ld, rs, 5, 0);
        run<FULL, 24, 2, false, 3>("full", xy, w, n, Q, ld, rs, 5, 0);
        run<FULL, 32, 2, true, 1>("full", xy, w, n, Q, ld, rs, 4, 0);
        run<FULL, 32, 2, true, 3>("full", xy, w, n, Q, ld, rs, 4, 0);
        run<FULL, 24, 2>("full", xy, w, n, Q, ld, rs, 5, 0);
        run<FULL, 32, 2, true>("full", xy, w, n, Q, ld, rs, 4, 0);
    }
    return 0;
}
