// Multi-vector forms of the two all-pairs reductions of the solve loop, for sweeps that carry NV
// right-hand sides through one factorization (applied-field scans, BASELINE config 4):
//   self field      out[i, v] = alpha ( qdiag_i w_i g[i, v] - sum_{j != i} q_ij w_j g[j, v] )
//   film coupling   out[i, v] (+)= sum_j (1/4pi) a_j (Jx[j, v] dy - Jy[j, v] dx) |r_ij|^-3
// (solver/solve_film.py:565 and solver/solve.py:28-73, :508, applied to [n, nvec] operands.)
//
// The expensive part of a pair, r^-3 (rsq + refinement, ~11 FP64 ops), does not depend on the
// vector: it is evaluated once and reused for a chunk of NV = 16 vectors held in registers, i.e.
// (11 + 16) / 16 = 1.7 ops per pair and vector instead of 13 (self field), (12 + 48) / 16 = 3.75
// instead of 14 (coupling).  Same decomposition as pairwise.hip: grid = target blocks of 256 x
// source slices, sources staged in LDS 64 at a time (coordinates + 16 or 32 "charges" each), slice
// partials combined by a second kernel in a fixed order (bitwise reproducible, no float atomics).
// A launch handles one chunk of <= 16 vectors (columns [v0, v0 + nv) of the row-major operands).
#include "common.hpp"

namespace ssa {
namespace {

constexpr int kMT = 256;       // targets per workgroup
constexpr int kMS = 64;        // sources per LDS stage
constexpr int kNV = 16;        // vectors per launch
constexpr int kMaxSlicesM = 32;

inline int pick_slices_m(int64_t nt, int64_t ns) {
    const int64_t tb = ceil_div(nt, kMT);
    int64_t s = ceil_div(1024, tb);
    const int64_t max_by_len = ceil_div(ns, 4 * kMS);
    if (s > max_by_len) s = max_by_len;
    if (s > kMaxSlicesM) s = kMaxSlicesM;
    if (s < 1) s = 1;
    return static_cast<int>(s);
}

template <typename T>
__global__ __launch_bounds__(kMT) void self_field_multi_kernel(const double *__restrict__ xy,
                                                               const double *__restrict__ w,
                                                               const T *__restrict__ g, int64_t n, int64_t nvec,
                                                               int64_t v0, int nv, int64_t slice_len,
                                                               double *__restrict__ partial) {
    __shared__ double s_x[kMS], s_y[kMS];
    __shared__ __attribute__((aligned(16))) double s_c[kMS][kNV];
    const int tid = threadIdx.x;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kMT + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < n) ? j_begin + slice_len : n;
    const int64_t ic = (i < n) ? i : n - 1;
    const double xi = xy[2 * ic], yi = xy[2 * ic + 1];
    double acc[kNV];
#pragma unroll
    for (int v = 0; v < kNV; ++v) acc[v] = 0.0;
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kMS) {
        __syncthreads();
        if (tid < kMS) {
            const int64_t j = t0 + tid;
            const bool ok = j < j_end;
            s_x[tid] = ok ? xy[2 * j] : 0.0;
            s_y[tid] = ok ? xy[2 * j + 1] : 0.0;
        }
        for (int e = tid; e < kMS * kNV; e += kMT) {  // charges (1/4pi) w_j g[j, v0 + v]
            const int k = e / kNV, v = e % kNV;
            const int64_t j = t0 + k;
            s_c[k][v] = (j < j_end && v < nv) ? kOneOver4Pi * (w[j] * static_cast<double>(g[j * nvec + v0 + v])) : 0.0;
        }
        __syncthreads();
        const int cnt = (j_end - t0 < kMS) ? static_cast<int>(j_end - t0) : kMS;
        for (int k = 0; k < cnt; ++k) {
            const double dx = xi - s_x[k], dy = yi - s_y[k];
            const double r2 = __builtin_fma(dx, dx, dy * dy);
            const double y = rsqrt_f64(r2);
            const double q = (t0 + k == i) ? 0.0 : y * (y * y);
#pragma unroll
            for (int v = 0; v < kNV; ++v) acc[v] = __builtin_fma(q, s_c[k][v], acc[v]);
        }
    }
    if (i < n) {
        double *dst = partial + (static_cast<int64_t>(blockIdx.y) * n + i) * kNV;
#pragma unroll
        for (int v = 0; v < kNV; ++v) dst[v] = acc[v];
    }
}

template <typename T>
__global__ void self_field_multi_combine_kernel(const double *__restrict__ partial, int slices, int64_t n,
                                                int64_t nvec, int64_t v0, int nv, const double *__restrict__ w,
                                                const double *__restrict__ qdiag, const T *__restrict__ g,
                                                double alpha, T *__restrict__ out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n * nv) return;
    const int64_t i = t / nv;
    const int v = static_cast<int>(t - i * nv);
    double s = 0.0;
    for (int k = 0; k < slices; ++k) s += partial[(static_cast<int64_t>(k) * n + i) * kNV + v];
    const double d = qdiag[i] * (w[i] * static_cast<double>(g[i * nvec + v0 + v]));
    out[i * nvec + v0 + v] = static_cast<T>(alpha * (d - s));
}

template <typename T>
__global__ __launch_bounds__(kMT) void biot_savart_multi_kernel(
    const double *__restrict__ src_xy, const T *__restrict__ src_areas, const double *__restrict__ src_J,
    int64_t ns, int64_t nvec, int64_t v0, int nv, int64_t slice_len, const double *__restrict__ tgt_xy,
    int64_t nt, double dz2, double *__restrict__ partial) {
    __shared__ double s_x[kMS], s_y[kMS];
    __shared__ __attribute__((aligned(16))) double s_ab[kMS][2 * kNV];  // (a, b) = (1/4pi) area (Jx, Jy) per vector
    const int tid = threadIdx.x;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kMT + tid;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < ns) ? j_begin + slice_len : ns;
    const int64_t ic = (i < nt) ? i : nt - 1;
    const double xi = tgt_xy[2 * ic], yi = tgt_xy[2 * ic + 1];
    double acc[kNV];
#pragma unroll
    for (int v = 0; v < kNV; ++v) acc[v] = 0.0;
    for (int64_t t0 = j_begin; t0 < j_end; t0 += kMS) {
        __syncthreads();
        if (tid < kMS) {
            const int64_t j = t0 + tid;
            const bool ok = j < j_end;
            s_x[tid] = ok ? src_xy[2 * j] : 0.0;
            s_y[tid] = ok ? src_xy[2 * j + 1] : 0.0;
        }
        for (int e = tid; e < kMS * 2 * kNV; e += kMT) {
            const int k = e / (2 * kNV), c = e % (2 * kNV), v = c >> 1, xyc = c & 1;
            const int64_t j = t0 + k;
            double val = 0.0;
            if (j < j_end && v < nv)
                val = kOneOver4Pi * static_cast<double>(src_areas[j]) * src_J[(j * nvec + v0 + v) * 2 + xyc];
            s_ab[k][c] = val;
        }
        __syncthreads();
        const int cnt = (j_end - t0 < kMS) ? static_cast<int>(j_end - t0) : kMS;
        for (int k = 0; k < cnt; ++k) {
            const double dx = xi - s_x[k], dy = yi - s_y[k];
            const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2));
            const double y = rsqrt_f64(r2);
            const double y3 = y * (y * y);
            const double dyq = dy * y3, dxq = dx * y3;
#pragma unroll
            for (int v = 0; v < kNV; ++v)
                acc[v] = __builtin_fma(s_ab[k][2 * v], dyq, __builtin_fma(-s_ab[k][2 * v + 1], dxq, acc[v]));
        }
    }
    if (i < nt) {
        double *dst = partial + (static_cast<int64_t>(blockIdx.y) * nt + i) * kNV;
#pragma unroll
        for (int v = 0; v < kNV; ++v) dst[v] = acc[v];
    }
}

template <typename T>
__global__ void biot_savart_multi_combine_kernel(const double *__restrict__ partial, int slices, int64_t nt,
                                                 int64_t nvec, int64_t v0, int nv, T *__restrict__ out,
                                                 int accumulate) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt * nv) return;
    const int64_t i = t / nv;
    const int v = static_cast<int>(t - i * nv);
    double s = 0.0;
    for (int k = 0; k < slices; ++k) s += partial[(static_cast<int64_t>(k) * nt + i) * kNV + v];
    T *dst = out + i * nvec + v0 + v;
    if (accumulate) s += static_cast<double>(*dst);
    *dst = static_cast<T>(s);
}

}  // namespace
}  // namespace ssa

using namespace ssa;

extern "C" size_t ssa_pairwise_multi_workspace_bytes(int64_t nt) {
    return static_cast<size_t>(kMaxSlicesM) * static_cast<size_t>(nt) * kNV * sizeof(double) + 256;
}

extern "C" int ssa_self_field_multi(const double *xy, const double *w, const double *qdiag, const void *g,
                                    int64_t n, int64_t nvec, void *out, double alpha, int dtype, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    if (!xy || !w || !qdiag || !g || !out || n <= 0 || nvec <= 0) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_pairwise_multi_workspace_bytes(n)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices_m(n, n);
    const int64_t slice_len = ceil_div(ceil_div(n, slices), kMS) * kMS;
    slices = static_cast<int>(ceil_div(n, slice_len));
    const dim3 grid(static_cast<unsigned>(ceil_div(n, kMT)), slices);
    for (int64_t v0 = 0; v0 < nvec; v0 += kNV) {
        const int nv = static_cast<int>((nvec - v0 < kNV) ? nvec - v0 : kNV);
        const dim3 cgrid(static_cast<unsigned>(ceil_div(n * nv, 256)));
        if (dtype == SSA_F64) {
            hipLaunchKernelGGL((self_field_multi_kernel<double>), grid, dim3(kMT), 0, st, xy, w,
                               static_cast<const double *>(g), n, nvec, v0, nv, slice_len, partial);
            hipLaunchKernelGGL((self_field_multi_combine_kernel<double>), cgrid, dim3(256), 0, st, partial, slices, n,
                               nvec, v0, nv, w, qdiag, static_cast<const double *>(g), alpha,
                               static_cast<double *>(out));
        } else {
            hipLaunchKernelGGL((self_field_multi_kernel<float>), grid, dim3(kMT), 0, st, xy, w,
                               static_cast<const float *>(g), n, nvec, v0, nv, slice_len, partial);
            hipLaunchKernelGGL((self_field_multi_combine_kernel<float>), cgrid, dim3(256), 0, st, partial, slices, n,
                               nvec, v0, nv, w, qdiag, static_cast<const float *>(g), alpha,
                               static_cast<float *>(out));
        }
        SSA_RETURN_IF_LAUNCH_FAILED();
    }
    return SSA_OK;
}

extern "C" int ssa_biot_savart_multi(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                                     const double *tgt_xy, int64_t nt, double dz, int64_t nvec, void *out,
                                     int accumulate, int dtype, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    if (!src_xy || !src_areas || !src_J || !tgt_xy || !out || ns <= 0 || nt <= 0 || nvec <= 0)
        return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_pairwise_multi_workspace_bytes(nt)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices_m(nt, ns);
    const int64_t slice_len = ceil_div(ceil_div(ns, slices), kMS) * kMS;
    slices = static_cast<int>(ceil_div(ns, slice_len));
    const dim3 grid(static_cast<unsigned>(ceil_div(nt, kMT)), slices);
    for (int64_t v0 = 0; v0 < nvec; v0 += kNV) {
        const int nv = static_cast<int>((nvec - v0 < kNV) ? nvec - v0 : kNV);
        const dim3 cgrid(static_cast<unsigned>(ceil_div(nt * nv, 256)));
        if (dtype == SSA_F64) {
            hipLaunchKernelGGL((biot_savart_multi_kernel<double>), grid, dim3(kMT), 0, st, src_xy,
                               static_cast<const double *>(src_areas), src_J, ns, nvec, v0, nv, slice_len, tgt_xy,
                               nt, dz * dz, partial);
            hipLaunchKernelGGL((biot_savart_multi_combine_kernel<double>), cgrid, dim3(256), 0, st, partial, slices,
                               nt, nvec, v0, nv, static_cast<double *>(out), accumulate);
        } else {
            hipLaunchKernelGGL((biot_savart_multi_kernel<float>), grid, dim3(kMT), 0, st, src_xy,
                               static_cast<const float *>(src_areas), src_J, ns, nvec, v0, nv, slice_len, tgt_xy, nt,
                               dz * dz, partial);
            hipLaunchKernelGGL((biot_savart_multi_combine_kernel<float>), cgrid, dim3(256), 0, st, partial, slices, nt,
                               nvec, v0, nv, static_cast<float *>(out), accumulate);
        }
        SSA_RETURN_IF_LAUNCH_FAILED();
    }
    return SSA_OK;
}
