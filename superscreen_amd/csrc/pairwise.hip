// All-pairs reductions with no n^2 output: inter-film Biot-Savart coupling
// (solver/solve.py:28-73) and the matrix-free self-field Q @ (w*g) (solve_film.py:565).
// FP64-VALU bound: ~20 flops per (target, source) pair, 40 bytes per vertex of input.
//
// Decomposition: grid = (target blocks of 256) x (source slices); a workgroup stages 256
// sources at a time in LDS (x, y and the two pre-multiplied "charges"), every lane owns one
// target and reads the staged sources as LDS broadcasts.  Slice partial sums go to a
// workspace [slices, nt] and are combined by a second tiny kernel in a fixed order, so the
// result is bitwise reproducible (no float atomics).
#include "common.hpp"

namespace ssa {

constexpr int kPairThreads = 256;
constexpr int kMaxSlices = 64;

struct alignas(16) Source {
    double x, y, a, b;
};

// Register tile: every lane owns kTPL targets (256 apart, so that loads and stores stay coalesced); one LDS
// broadcast of a staged source (32 bytes) feeds kTPL pair evaluations instead of one, and the kTPL independent
// dependency chains per lane keep the FP64 pipe issuing while a v_rsq_f64 is in flight.  A workgroup covers
// 256 * kTPL targets x one source slice; padding entries of a stage are inert sources (far away, zero charge).
constexpr int kTPL = 2;   // measured at 25 117 x 25 117 (same box): self field 1.52 / 1.75 / 1.71 Tpair/s for 4 / 2 / 1
constexpr double kFarAway = 1.0e15;

template <typename T>
__global__ __launch_bounds__(kPairThreads) void biot_savart_partial_kernel(
    const double *__restrict__ src_xy, const T *__restrict__ src_areas,
    const double *__restrict__ src_J, int64_t src_begin, int64_t src_end, int64_t slice_len,
    const double *__restrict__ tgt_xy, int64_t nt, double dz2, double *__restrict__ partial) {
    __shared__ Source s_src[kPairThreads];
    const int tid = threadIdx.x;
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * (kPairThreads * kTPL) + tid;
    const int64_t j_begin = src_begin + static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < src_end) ? j_begin + slice_len : src_end;
    double xi[kTPL], yi[kTPL], acc[kTPL];
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t i = i0 + r * kPairThreads;
        const int64_t ic = (i < nt) ? i : nt - 1;
        xi[r] = tgt_xy[2 * ic];
        yi[r] = tgt_xy[2 * ic + 1];
        acc[r] = 0.0;
    }
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kPairThreads) {
        const int64_t j = t0 + tid;
        Source s;
        if (j < j_end) {
            const double a = kOneOver4Pi * static_cast<double>(src_areas[j]);
            s.x = src_xy[2 * j];
            s.y = src_xy[2 * j + 1];
            s.a = a * src_J[2 * j];      // (1/4pi) a_j Jx_j
            s.b = a * src_J[2 * j + 1];  // (1/4pi) a_j Jy_j
        } else {
            s.x = kFarAway; s.y = kFarAway; s.a = 0.0; s.b = 0.0;
        }
        __syncthreads();
        s_src[tid] = s;
        __syncthreads();
        const int cnt = (j_end - t0 < kPairThreads) ? static_cast<int>(j_end - t0) : kPairThreads;
        const int cnt2 = (cnt + 1) & ~1;   // entries [cnt, 256) of the stage are inert
#pragma unroll 2
        for (int k = 0; k < cnt2; ++k) {
            const Source q = s_src[k];
#pragma unroll
            for (int r = 0; r < kTPL; ++r) {
                const double dx = xi[r] - q.x, dy = yi[r] - q.y;
                const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2));
                const double cross = __builtin_fma(q.a, dy, -(q.b * dx));  // Jx dy - Jy dx
                acc[r] = __builtin_fma(cross, inv_r3(r2), acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t i = i0 + r * kPairThreads;
        if (i < nt) partial[static_cast<int64_t>(blockIdx.y) * nt + i] = acc[r];
    }
}

// Field of a current sheet at arbitrary points in space (sources/current.py:13-110, the numba
// kernels _biot_savart_2d_z / _biot_savart_2d_vector behind Solution.field_at_position): same
// skeleton, per-point dz, NC = 1 (z component) or 3 (vector) accumulators per evaluation point.
//   Bx += pref Jy dz,   By -= pref Jx dz,   Bz += pref (Jx dy - Jy dx),   pref = a_k r^-3
// (the unit prefactor mu_0 / 4 pi * [A/m per current unit / length unit] is applied by the combine).
template <int NC>
__global__ __launch_bounds__(kPairThreads) void sheet_field_partial_kernel(
    const double *__restrict__ src_xy, const double *__restrict__ src_areas, const double *__restrict__ src_J,
    int64_t ns, int64_t slice_len, double z0, const double *__restrict__ eval_xyz, int64_t np,
    double *__restrict__ partial) {
    __shared__ Source s_src[kPairThreads];
    const int tid = threadIdx.x;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kPairThreads + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < ns) ? j_begin + slice_len : ns;
    const int64_t ic = (i < np) ? i : np - 1;
    const double xi = eval_xyz[3 * ic], yi = eval_xyz[3 * ic + 1], dz = eval_xyz[3 * ic + 2] - z0;
    const double dz2 = dz * dz;
    double az = 0.0, sa = 0.0, sb = 0.0;  // Bz, sum pref a_k Jx, sum pref a_k Jy
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kPairThreads) {
        const int64_t j = t0 + tid;
        Source s;
        if (j < j_end) {
            const double a = src_areas[j];
            s.x = src_xy[2 * j];
            s.y = src_xy[2 * j + 1];
            s.a = a * src_J[2 * j];
            s.b = a * src_J[2 * j + 1];
        } else {
            s.x = 0.0; s.y = 0.0; s.a = 0.0; s.b = 0.0;
        }
        __syncthreads();
        s_src[tid] = s;
        __syncthreads();
        const int cnt = (j_end - t0 < kPairThreads) ? static_cast<int>(j_end - t0) : kPairThreads;
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const Source q = s_src[k];
            const double dx = xi - q.x, dy = yi - q.y;
            const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2));
            const double y3 = inv_r3(r2);
            az = __builtin_fma(__builtin_fma(q.a, dy, -(q.b * dx)), y3, az);
            if (NC == 3) {
                sa = __builtin_fma(q.a, y3, sa);
                sb = __builtin_fma(q.b, y3, sb);
            }
        }
    }
    if (i < np) {
        double *dst = partial + (static_cast<int64_t>(blockIdx.y) * np + i) * NC;
        if (NC == 3) {
            dst[0] = sb * dz;   // Jy dz
            dst[1] = -sa * dz;  // -Jx dz
            dst[2] = az;
        } else {
            dst[0] = az;
        }
    }
}

// Vector potential of a current sheet (solution.py:833-934):  A_xy(r) = sum_k a_k J_k / |r - r_k|
// (prefactor mu_0 / 4 pi and units applied by the combine), two accumulators per point.
__global__ __launch_bounds__(kPairThreads) void sheet_potential_partial_kernel(
    const double *__restrict__ src_xy, const double *__restrict__ src_areas, const double *__restrict__ src_J,
    int64_t ns, int64_t slice_len, double z0, const double *__restrict__ eval_xyz, int64_t np,
    double *__restrict__ partial) {
    __shared__ Source s_src[kPairThreads];
    const int tid = threadIdx.x;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kPairThreads + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < ns) ? j_begin + slice_len : ns;
    const int64_t ic = (i < np) ? i : np - 1;
    const double xi = eval_xyz[3 * ic], yi = eval_xyz[3 * ic + 1], dz = eval_xyz[3 * ic + 2] - z0;
    const double dz2 = dz * dz;
    double ax = 0.0, ay = 0.0;
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kPairThreads) {
        const int64_t j = t0 + tid;
        Source s;
        if (j < j_end) {
            const double a = src_areas[j];
            s.x = src_xy[2 * j];
            s.y = src_xy[2 * j + 1];
            s.a = a * src_J[2 * j];
            s.b = a * src_J[2 * j + 1];
        } else {
            s.x = 0.0; s.y = 0.0; s.a = 0.0; s.b = 0.0;
        }
        __syncthreads();
        s_src[tid] = s;
        __syncthreads();
        const int cnt = (j_end - t0 < kPairThreads) ? static_cast<int>(j_end - t0) : kPairThreads;
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const Source q = s_src[k];
            const double dx = xi - q.x, dy = yi - q.y;
            const double y = rsqrt_f64(__builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2)));
            ax = __builtin_fma(q.a, y, ax);
            ay = __builtin_fma(q.b, y, ay);
        }
    }
    if (i < np) {
        double *dst = partial + (static_cast<int64_t>(blockIdx.y) * np + i) * 2;
        dst[0] = ax;
        dst[1] = ay;
    }
}

__global__ void sheet_field_combine_kernel(const double *__restrict__ partial, int slices, int64_t count,
                                           double prefactor, double *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = sum_strided(partial + i, slices, count);
    out[i] = prefactor * s;
}

// out[i] = alpha * ( qdiag_i w_i g_i - sum_{j != i} q_ij w_j g_j )   (partials: the sum only)
// Shared body of the two self-field kernels: the staged tile [t0, t0 + cnt) of sources against the kTPL targets
// of a lane (vertex indices it[r]).  Only a wave one of whose targets lies inside the tile pays for the
// j == i test.
struct alignas(8) Charge {
    double x, y, c;
};

__device__ __forceinline__ void self_tile(const double *s_x, const double *s_y, const double *s_c, int64_t t0,
                                          int cnt, const int64_t (&it)[kTPL], const double (&xi)[kTPL],
                                          const double (&yi)[kTPL], double (&acc)[kTPL]) {
    bool hit = false;
#pragma unroll
    for (int r = 0; r < kTPL; ++r) hit |= (it[r] >= t0 && it[r] < t0 + cnt);
    const int cnt2 = (cnt + 1) & ~1;
    if (__any(hit)) {
#pragma unroll 2
        for (int k = 0; k < cnt2; ++k) {
            const double qx = s_x[k], qy = s_y[k], qc = s_c[k];
#pragma unroll
            for (int r = 0; r < kTPL; ++r) {
                const double dx = xi[r] - qx, dy = yi[r] - qy;
                const double t = qc * inv_r3(__builtin_fma(dx, dx, dy * dy));
                acc[r] += (t0 + k == it[r]) ? 0.0 : t;
            }
        }
    } else {
#pragma unroll 2
        for (int k = 0; k < cnt2; ++k) {
            const double qx = s_x[k], qy = s_y[k], qc = s_c[k];
#pragma unroll
            for (int r = 0; r < kTPL; ++r) {
                const double dx = xi[r] - qx, dy = yi[r] - qy;
                acc[r] = __builtin_fma(qc, inv_r3(__builtin_fma(dx, dx, dy * dy)), acc[r]);
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ void stage_charges(const double *__restrict__ xy, const double *__restrict__ w,
                                              const T *__restrict__ g, int64_t j, int64_t j_end, int tid, double *s_x,
                                              double *s_y, double *s_c) {
    double sx = kFarAway, sy = kFarAway, sc = 0.0;   // inert padding
    if (j < j_end) {
        sx = xy[2 * j];
        sy = xy[2 * j + 1];
        sc = kOneOver4Pi * (w[j] * static_cast<double>(g[j]));
    }
    __syncthreads();
    s_x[tid] = sx; s_y[tid] = sy; s_c[tid] = sc;
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kPairThreads) void self_field_partial_kernel(
    const double *__restrict__ xy, const double *__restrict__ w, const T *__restrict__ g,
    int64_t n, int64_t slice_len, double *__restrict__ partial) {
    __shared__ double s_x[kPairThreads], s_y[kPairThreads], s_c[kPairThreads];
    const int tid = threadIdx.x;
    const int64_t i0 = static_cast<int64_t>(blockIdx.x) * (kPairThreads * kTPL) + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < n) ? j_begin + slice_len : n;
    int64_t it[kTPL];
    double xi[kTPL], yi[kTPL], acc[kTPL];
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t i = i0 + r * kPairThreads;
        it[r] = (i < n) ? i : n - 1;
        xi[r] = xy[2 * it[r]];
        yi[r] = xy[2 * it[r] + 1];
        acc[r] = 0.0;
    }
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kPairThreads) {
        stage_charges(xy, w, g, t0 + tid, j_end, tid, s_x, s_y, s_c);
        const int cnt = (j_end - t0 < kPairThreads) ? static_cast<int>(j_end - t0) : kPairThreads;
        self_tile(s_x, s_y, s_c, t0, cnt, it, xi, yi, acc);
    }
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t i = i0 + r * kPairThreads;
        if (i < n) partial[static_cast<int64_t>(blockIdx.y) * n + i] = acc[r];
    }
}

template <typename T>
__global__ void combine_partials_kernel(const double *__restrict__ partial, int slices,
                                        int64_t nt, T *__restrict__ out, int accumulate) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    double s = sum_strided(partial + i, slices, nt);
    if (accumulate) s += static_cast<double>(out[i]);  // solver/solve.py:508 (f64 add, one cast)
    out[i] = static_cast<T>(s);
}

template <typename T>
__global__ void self_field_combine_kernel(const double *__restrict__ partial, int slices,
                                          int64_t n, const double *__restrict__ w,
                                          const double *__restrict__ qdiag,
                                          const T *__restrict__ g, double alpha,
                                          T *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = sum_strided(partial + i, slices, n);
    const double d = qdiag[i] * (w[i] * static_cast<double>(g[i]));
    out[i] = static_cast<T>(alpha * (d - s));
}

// The same sum for a LIST of target rows (sorted or not): out[rows[k]], k < nr.  Used for the mesh
// vertices outside the film interior; inside it the London equation gives the self field from O(n)
// data (ssa_london_field_rows).
template <typename T>
__global__ __launch_bounds__(kPairThreads) void self_field_rows_partial_kernel(
    const double *__restrict__ xy, const double *__restrict__ w, const T *__restrict__ g, int64_t n,
    const int64_t *__restrict__ rows, int64_t nr, int64_t slice_len, double *__restrict__ partial) {
    __shared__ double s_x[kPairThreads], s_y[kPairThreads], s_c[kPairThreads];
    const int tid = threadIdx.x;
    const int64_t k0 = static_cast<int64_t>(blockIdx.x) * (kPairThreads * kTPL) + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < n) ? j_begin + slice_len : n;
    int64_t it[kTPL];
    double xi[kTPL], yi[kTPL], acc[kTPL];
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t k = k0 + r * kPairThreads;
        it[r] = rows[(k < nr) ? k : nr - 1];
        xi[r] = xy[2 * it[r]];
        yi[r] = xy[2 * it[r] + 1];
        acc[r] = 0.0;
    }
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kPairThreads) {
        stage_charges(xy, w, g, t0 + tid, j_end, tid, s_x, s_y, s_c);
        const int cnt = (j_end - t0 < kPairThreads) ? static_cast<int>(j_end - t0) : kPairThreads;
        self_tile(s_x, s_y, s_c, t0, cnt, it, xi, yi, acc);
    }
#pragma unroll
    for (int r = 0; r < kTPL; ++r) {
        const int64_t k = k0 + r * kPairThreads;
        if (k < nr) partial[static_cast<int64_t>(blockIdx.y) * nr + k] = acc[r];
    }
}

template <typename T>
__global__ void self_field_rows_combine_kernel(const double *__restrict__ partial, int slices, int64_t nr,
                                               const int64_t *__restrict__ rows, const double *__restrict__ w,
                                               const double *__restrict__ qdiag, const T *__restrict__ g,
                                               double alpha, T *__restrict__ out) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= nr) return;
    double s = 0.0;
    for (int q = 0; q < slices; ++q) s += partial[static_cast<int64_t>(q) * nr + k];
    const int64_t i = rows[k];
    const double d = qdiag[i] * (w[i] * static_cast<double>(g[i]));
    out[i] = static_cast<T>(alpha * (d - s));
}

// The mesh-Laplacian SpMV of the London equation, one vector: an 8-lane segment of a wavefront per CSR row
// (about 7 entries per row: one entry per lane), reduced with xor-shuffles inside the segment.
template <typename T>
__global__ void london_field_rows_seg8_kernel(const int64_t *__restrict__ indptr, const int64_t *__restrict__ indices,
                                              const double *__restrict__ data, const double *__restrict__ Lambda,
                                              const T *__restrict__ g, const T *__restrict__ applied,
                                              const T *__restrict__ other, const int64_t *__restrict__ rows,
                                              int64_t nr, T *__restrict__ out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t k = t >> 3;
    const int sub = static_cast<int>(t & 7);
    const int64_t r = rows[(k < nr) ? k : nr - 1];
    double acc = 0.0;
    if (k < nr) {
        for (int64_t p = indptr[r] + sub; p < indptr[r + 1]; p += 8) {
            const int64_t j = indices[p];
            acc = __builtin_fma(data[p], Lambda[j] * static_cast<double>(g[j]), acc);
        }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 8);  // every lane of the wave takes part
    if (k < nr && sub == 0) {
        double hz = static_cast<double>(applied[r]);
        if (other) hz += static_cast<double>(other[r]);
        out[r] = static_cast<T>(acc - hz);
    }
}

// out[r, v] = sum_j lap[r, j] Lambda_j g[j, v] - applied[r, v] - other[r, v],  r = rows[k]: the London
// equation H_applied + H_other + H_self = Laplacian(Lambda g) read as a formula for the self field.
// One thread per (row, vector); the mesh Laplacian has about 7 entries per row.
template <typename T>
__global__ void london_field_rows_kernel(const int64_t *__restrict__ indptr, const int64_t *__restrict__ indices,
                                         const double *__restrict__ data, const double *__restrict__ Lambda,
                                         const T *__restrict__ g, const T *__restrict__ applied,
                                         const T *__restrict__ other, const int64_t *__restrict__ rows, int64_t nr,
                                         int64_t nvec, T *__restrict__ out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nr * nvec) return;
    const int64_t k = t / nvec, v = t - k * nvec;
    const int64_t r = rows[k];
    double acc = 0.0;
    for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
        const int64_t j = indices[p];
        acc = __builtin_fma(data[p], Lambda[j] * static_cast<double>(g[j * nvec + v]), acc);
    }
    double hz = static_cast<double>(applied[r * nvec + v]);
    if (other) hz += static_cast<double>(other[r * nvec + v]);
    out[r * nvec + v] = static_cast<T>(acc - hz);
}

inline int pick_slices(int64_t nt, int64_t ns) {
    const int64_t tb = ceil_div(nt, kPairThreads);
    int64_t s = ceil_div(2048, tb);
    const int64_t max_by_len = ceil_div(ns, kPairThreads);  // at least one tile per slice
    if (s > max_by_len) s = max_by_len;
    if (s > kMaxSlices) s = kMaxSlices;
    if (s < 1) s = 1;
    return static_cast<int>(s);
}

// Grid of the register-tiled kernels: target blocks of 256 * kTPL x source slices, with the slice count chosen so
// that ALL workgroups are resident at once and the CUs carry the same number of them -- 4 workgroups (16 waves)
// per CU: one round, no tail (the one-target-per-lane form ran 2 079 workgroups on 2 048 slots at 25 117 x
// 25 117).  Slices are at least 64 sources long and at most kMaxSlices in number.
inline void pick_tile_grid(int64_t nt, int64_t ns, int *slices, int64_t *slice_len) {
    const int64_t tb = ceil_div(nt, kPairThreads * kTPL);
    int64_t s = ((kTPL >= 4 ? 4 : 8) * static_cast<int64_t>(device_cu_count())) / tb;
    const int64_t max_by_len = ceil_div(ns, 64);
    if (s > max_by_len) s = max_by_len;
    if (s > kMaxSlices) s = kMaxSlices;
    if (s < 1) s = 1;
    *slice_len = ceil_div(ns, s);
    *slices = static_cast<int>(ceil_div(ns, *slice_len));
}

}  // namespace ssa

using namespace ssa;

extern "C" size_t ssa_biot_savart_workspace_bytes(int64_t nt) {
    return static_cast<size_t>(kMaxSlices) * static_cast<size_t>(nt) * sizeof(double) + 256;
}

extern "C" int ssa_biot_savart(const double *src_xy, const void *src_areas, const double *src_J,
                               int64_t ns, int64_t src_begin, int64_t src_end,
                               const double *tgt_xy, int64_t nt, double dz, void *out,
                               int accumulate, int dtype, void *workspace,
                               size_t workspace_bytes, void *stream) {
    if (!src_xy || !src_areas || !src_J || !tgt_xy || !out || ns <= 0 || nt <= 0)
        return SSA_ERR_INVALID_ARGUMENT;
    if (src_begin < 0 || src_end > ns || src_begin > src_end) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_biot_savart_workspace_bytes(nt))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    const int64_t len = src_end - src_begin;
    int slices = 1;
    if (len > 0) {
        int64_t slice_len = 0;
        pick_tile_grid(nt, len, &slices, &slice_len);
        const dim3 grid(static_cast<unsigned>(ceil_div(nt, kPairThreads * kTPL)), slices);
        if (dtype == SSA_F64) {
            hipLaunchKernelGGL((biot_savart_partial_kernel<double>), grid, dim3(kPairThreads), 0,
                               st, src_xy, static_cast<const double *>(src_areas), src_J,
                               src_begin, src_end, slice_len, tgt_xy, nt, dz * dz, partial);
        } else {
            hipLaunchKernelGGL((biot_savart_partial_kernel<float>), grid, dim3(kPairThreads), 0,
                               st, src_xy, static_cast<const float *>(src_areas), src_J,
                               src_begin, src_end, slice_len, tgt_xy, nt, dz * dz, partial);
        }
        SSA_RETURN_IF_LAUNCH_FAILED();
    } else {
        slices = 0;
    }
    const dim3 cgrid(static_cast<unsigned>(ceil_div(nt, 256)));
    if (dtype == SSA_F64) {
        hipLaunchKernelGGL((combine_partials_kernel<double>), cgrid, dim3(256), 0, st, partial,
                           slices, nt, static_cast<double *>(out), accumulate);
    } else {
        hipLaunchKernelGGL((combine_partials_kernel<float>), cgrid, dim3(256), 0, st, partial,
                           slices, nt, static_cast<float *>(out), accumulate);
    }
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

extern "C" size_t ssa_sheet_field_workspace_bytes(int64_t np, int vector) {
    return static_cast<size_t>(kMaxSlices) * static_cast<size_t>(np) * (vector ? 3 : 1) * sizeof(double) + 256;
}

extern "C" int ssa_sheet_field(const double *src_xy, const double *src_areas, const double *src_J, int64_t ns,
                               double z0, const double *eval_xyz, int64_t np, double prefactor, int vector,
                               double *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!src_xy || !src_areas || !src_J || !eval_xyz || !out || ns <= 0 || np <= 0)
        return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_sheet_field_workspace_bytes(np, vector))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices(np, ns);
    const int64_t slice_len = ceil_div(ceil_div(ns, slices), kPairThreads) * kPairThreads;
    slices = static_cast<int>(ceil_div(ns, slice_len));
    const dim3 grid(static_cast<unsigned>(ceil_div(np, kPairThreads)), slices);
    const int nc = vector ? 3 : 1;
    if (vector)
        hipLaunchKernelGGL((sheet_field_partial_kernel<3>), grid, dim3(kPairThreads), 0, st, src_xy, src_areas,
                           src_J, ns, slice_len, z0, eval_xyz, np, partial);
    else
        hipLaunchKernelGGL((sheet_field_partial_kernel<1>), grid, dim3(kPairThreads), 0, st, src_xy, src_areas,
                           src_J, ns, slice_len, z0, eval_xyz, np, partial);
    SSA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(sheet_field_combine_kernel, dim3(static_cast<unsigned>(ceil_div(np * nc, 256))), dim3(256), 0,
                       st, partial, slices, np * nc, prefactor, out);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

extern "C" int ssa_sheet_potential(const double *src_xy, const double *src_areas, const double *src_J, int64_t ns,
                                   double z0, const double *eval_xyz, int64_t np, double prefactor, double *out,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!src_xy || !src_areas || !src_J || !eval_xyz || !out || ns <= 0 || np <= 0)
        return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_sheet_field_workspace_bytes(np, 1)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices(np, ns);
    const int64_t slice_len = ceil_div(ceil_div(ns, slices), kPairThreads) * kPairThreads;
    slices = static_cast<int>(ceil_div(ns, slice_len));
    hipLaunchKernelGGL(sheet_potential_partial_kernel,
                       dim3(static_cast<unsigned>(ceil_div(np, kPairThreads)), slices), dim3(kPairThreads), 0, st,
                       src_xy, src_areas, src_J, ns, slice_len, z0, eval_xyz, np, partial);
    SSA_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(sheet_field_combine_kernel, dim3(static_cast<unsigned>(ceil_div(np * 2, 256))), dim3(256), 0,
                       st, partial, slices, np * 2, prefactor, out);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

extern "C" size_t ssa_self_field_workspace_bytes(int64_t n) {
    return static_cast<size_t>(kMaxSlices) * static_cast<size_t>(n) * sizeof(double) + 256;
}

extern "C" int ssa_self_field(const double *xy, const double *w, const double *qdiag,
                              const void *g, int64_t n, void *out, double alpha, int dtype,
                              void *workspace, size_t workspace_bytes, void *stream) {
    if (!xy || !w || !qdiag || !g || !out || n <= 0) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_self_field_workspace_bytes(n))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = 1;
    int64_t slice_len = 0;
    pick_tile_grid(n, n, &slices, &slice_len);
    const dim3 grid(static_cast<unsigned>(ceil_div(n, kPairThreads * kTPL)), slices);
    const dim3 cgrid(static_cast<unsigned>(ceil_div(n, 256)));
    if (dtype == SSA_F64) {
        hipLaunchKernelGGL((self_field_partial_kernel<double>), grid, dim3(kPairThreads), 0, st,
                           xy, w, static_cast<const double *>(g), n, slice_len, partial);
        hipLaunchKernelGGL((self_field_combine_kernel<double>), cgrid, dim3(256), 0, st, partial,
                           slices, n, w, qdiag, static_cast<const double *>(g), alpha,
                           static_cast<double *>(out));
    } else {
        hipLaunchKernelGGL((self_field_partial_kernel<float>), grid, dim3(kPairThreads), 0, st, xy,
                           w, static_cast<const float *>(g), n, slice_len, partial);
        hipLaunchKernelGGL((self_field_combine_kernel<float>), cgrid, dim3(256), 0, st, partial,
                           slices, n, w, qdiag, static_cast<const float *>(g), alpha,
                           static_cast<float *>(out));
    }
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}


extern "C" int ssa_self_field_rows(const double *xy, const double *w, const double *qdiag, const void *g, int64_t n,
                                   const int64_t *rows, int64_t nr, void *out, double alpha, int dtype,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!xy || !w || !qdiag || !g || !out || n <= 0 || nr < 0 || (nr > 0 && !rows)) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (nr == 0) return SSA_OK;
    if (!workspace || workspace_bytes < ssa_self_field_workspace_bytes(nr)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = 1;
    int64_t slice_len = 0;
    pick_tile_grid(nr, n, &slices, &slice_len);
    const dim3 grid(static_cast<unsigned>(ceil_div(nr, kPairThreads * kTPL)), slices);
    const dim3 cgrid(static_cast<unsigned>(ceil_div(nr, 256)));
    if (dtype == SSA_F64) {
        hipLaunchKernelGGL((self_field_rows_partial_kernel<double>), grid, dim3(kPairThreads), 0, st, xy, w,
                           static_cast<const double *>(g), n, rows, nr, slice_len, partial);
        hipLaunchKernelGGL((self_field_rows_combine_kernel<double>), cgrid, dim3(256), 0, st, partial, slices, nr,
                           rows, w, qdiag, static_cast<const double *>(g), alpha, static_cast<double *>(out));
    } else {
        hipLaunchKernelGGL((self_field_rows_partial_kernel<float>), grid, dim3(kPairThreads), 0, st, xy, w,
                           static_cast<const float *>(g), n, rows, nr, slice_len, partial);
        hipLaunchKernelGGL((self_field_rows_combine_kernel<float>), cgrid, dim3(256), 0, st, partial, slices, nr,
                           rows, w, qdiag, static_cast<const float *>(g), alpha, static_cast<float *>(out));
    }
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

extern "C" int ssa_london_field_rows(const int64_t *lap_indptr, const int64_t *lap_indices, const double *lap_data,
                                     const double *Lambda, const void *g, const void *applied, const void *other,
                                     const int64_t *rows, int64_t nr, int64_t nvec, void *out, int dtype,
                                     void *stream) {
    if (!lap_indptr || !lap_indices || !lap_data || !Lambda || !g || !applied || !out || nr < 0 || nvec <= 0 ||
        (nr > 0 && !rows))
        return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (nr == 0) return SSA_OK;
    hipStream_t st = as_stream(stream);
    if (nvec == 1) {  // one vector: segmented reduction, 8 lanes per row
        const dim3 sgrid(static_cast<unsigned>(ceil_div(nr * 8, 256)));
        if (dtype == SSA_F64)
            hipLaunchKernelGGL((london_field_rows_seg8_kernel<double>), sgrid, dim3(256), 0, st, lap_indptr, lap_indices,
                               lap_data, Lambda, static_cast<const double *>(g), static_cast<const double *>(applied),
                               static_cast<const double *>(other), rows, nr, static_cast<double *>(out));
        else
            hipLaunchKernelGGL((london_field_rows_seg8_kernel<float>), sgrid, dim3(256), 0, st, lap_indptr, lap_indices,
                               lap_data, Lambda, static_cast<const float *>(g), static_cast<const float *>(applied),
                               static_cast<const float *>(other), rows, nr, static_cast<float *>(out));
        SSA_RETURN_IF_LAUNCH_FAILED();
        return SSA_OK;
    }
    const dim3 grid(static_cast<unsigned>(ceil_div(nr * nvec, 256)));
    if (dtype == SSA_F64)
        hipLaunchKernelGGL((london_field_rows_kernel<double>), grid, dim3(256), 0, st, lap_indptr, lap_indices, lap_data,
                           Lambda, static_cast<const double *>(g), static_cast<const double *>(applied),
                           static_cast<const double *>(other), rows, nr, nvec, static_cast<double *>(out));
    else
        hipLaunchKernelGGL((london_field_rows_kernel<float>), grid, dim3(256), 0, st, lap_indptr, lap_indices, lap_data,
                           Lambda, static_cast<const float *>(g), static_cast<const float *>(applied),
                           static_cast<const float *>(other), rows, nr, nvec, static_cast<float *>(out));
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
