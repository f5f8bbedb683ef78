// Tiled all-pairs assembly of the kernel matrix Q and of the film / hole systems
//   A = Q[rows, cols] * w[cols] - Lambda[cols] * Del2[rows, cols]
// for gfx950.  HBM-write bound: every output element is generated from 40 bytes of vertex
// data held on chip and written exactly once (no intermediate n^2 passes; the reference
// makes 4: q_matrix, einsum row-sum, fill_diagonal, unary minus -- device/mesh.py:453-458).
//
// Work decomposition: one workgroup (4 waves) owns a strip of TR consecutive rows and
// sweeps all columns; lane <-> two adjacent columns, so each wave-instruction stores 1 KiB
// (f64) of one row, fully coalesced.  Row coordinates are wave-uniform and are read with
// scalar loads (SGPRs: no VGPR cost for the 16 rows of a strip); the row sums needed for the
// diagonal stay in registers (TR per lane) and are reduced once per strip (wave shuffle + one
// LDS hop), so the diagonal costs no extra pass over memory.  Optional row scaling and
// lower-tiles-only output serve the Cholesky route (S = diag(w) A, chol.hip).
#include "common.hpp"

namespace ssa {

constexpr int kStripRows = 24;    // system_assemble_kernel
constexpr int kQStripRows = 24;   // q_assemble_kernel: at most this many rows per workgroup
constexpr int kQGroupsPerCu = 5;  // ... and a whole number of rounds of this many workgroups per CU
constexpr int kAsmThreads = 256;

template <typename OutT>
struct Pair;
template <>
struct Pair<double> {
    using type = double2;
};
template <>
struct Pair<float> {
    using type = float2;
};

template <typename OutT>
__device__ __forceinline__ void store_pair(OutT *p, OutT a, OutT b) {
    typename Pair<OutT>::type v;
    v.x = a;
    v.y = b;
    *reinterpret_cast<typename Pair<OutT>::type *>(p) = v;
}

// Q_ij = -q_ij (i != j), Q_ii = (C_i + sum_{l != i} q_il w_l) / w_i.
//
// Rows per workgroup: the grid is a whole number of "rounds" of kQGroupsPerCu workgroups per CU and the n rows are
// dealt out evenly (heights differ by at most one row, <= TR).  Measured (tools/probes/q_probe.hip, round 4, same
// box, n = 25 117 / 50 311): 16-row strips on 7-8 resident workgroups per CU 4.97-5.05 / 5.40-5.42 TB/s, 24-row
// strips in rounds of 5 per CU 5.31-5.35 / 5.58-5.66 TB/s against 5.44-5.47 / 5.55-5.69 TB/s for the same store
// stream with nothing to compute -- fewer, taller workgroups keep fewer write streams open at a time.  What the
// loop keeps out of scalar registers matters as much: per-row store addresses and row indices as scalars cost 88
// SGPR spills (a v_readlane per use) in the 16-row version, 4.5 TB/s; here the store address advances by one row
// per step and the diagonal is found from the lane's own distance to the strip, row coordinates come from LDS.
__host__ __device__ inline void strip_rows(int64_t n, int64_t groups, int64_t b, int64_t *i0, int *h) {
    const int64_t base = n / groups, extra = n % groups;
    *i0 = b * base + (b < extra ? b : extra);
    *h = static_cast<int>(base + (b < extra ? 1 : 0));
}

template <typename OutT, int TR, bool WANT_Q>
__global__ __launch_bounds__(kAsmThreads) void q_assemble_kernel(
    const double *__restrict__ xy, const double *__restrict__ w, const double *__restrict__ C,
    int64_t n, OutT *__restrict__ Q, int64_t ldq, double *__restrict__ qdiag) {
    __shared__ double s_part[kAsmThreads / kWave][TR];
    __shared__ double2 s_xy[TR];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int64_t i0;
    int h;  // rows of this workgroup, 1 .. TR (workgroup-uniform)
    strip_rows(n, gridDim.x, blockIdx.x, &i0, &h);
    if (tid < TR) {
        const int64_t i = (tid < h) ? i0 + tid : i0;
        s_xy[tid] = *reinterpret_cast<const double2 *>(xy + 2 * i);
    }
    __syncthreads();
    double acc[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) acc[r] = 0.0;

    // Workgroup b starts its sweep 512 columns further right than workgroup b - 1 and wraps around: the workgroups
    // resident at a time then do not all write the same column range of their rows (store stream alone at
    // n = 50 311: 5.55 -> 5.66-5.69 TB/s).
    const int64_t span = ((n + 2 * kAsmThreads - 1) / (2 * kAsmThreads)) * (2 * kAsmThreads);
    const int64_t shift = (static_cast<int64_t>(blockIdx.x) * 2 * kAsmThreads) % span;
    for (int64_t jj = 2 * tid; jj < span; jj += 2 * kAsmThreads) {
        int64_t j = jj + shift;
        if (j >= span) j -= span;
        if (j >= n) continue;
        const bool has1 = (j + 1 < n);
        const double xj0 = xy[2 * j], yj0 = xy[2 * j + 1];
        const double xj1 = has1 ? xy[2 * j + 2] : 0.0;
        const double yj1 = has1 ? xy[2 * j + 3] : 0.0;
        const double w0 = w[j];
        const double w1 = has1 ? w[j + 1] : 0.0;
        OutT *qp = WANT_Q ? Q + i0 * ldq + j : nullptr;
        const int dj = static_cast<int>(j - i0);   // row r is on the diagonal of column j when dj == r
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (r < h) {  // uniform branch: rows beyond this workgroup's share cost nothing
                int off = r * 16;   // one LDS read per use (hidden from loop-invariant code motion: 4 TR registers)
                asm volatile("" : "+v"(off));
                const double2 pr = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_xy) + off);
                const double dx0 = pr.x - xj0, dy0 = pr.y - yj0;
                const double dx1 = pr.x - xj1, dy1 = pr.y - yj1;
                double q0 = inv_r3_over_4pi(__builtin_fma(dx0, dx0, dy0 * dy0));
                double q1 = inv_r3_over_4pi(__builtin_fma(dx1, dx1, dy1 * dy1));
                q0 = (dj == r) ? 0.0 : q0;                   // distance.py:104-105
                q1 = (dj + 1 == r || !has1) ? 0.0 : q1;
                acc[r] = __builtin_fma(q0, w0, acc[r]);
                acc[r] = __builtin_fma(q1, w1, acc[r]);
                if (WANT_Q) {
                    store_pair<OutT>(qp, static_cast<OutT>(-q0), static_cast<OutT>(-q1));
                    qp += ldq;
                }
            }
        }
    }
    // Row sums: lanes -> waves -> strip.
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const double s = wave_sum(acc[r]);
        if (lane == 0) s_part[wave][r] = s;
    }
    __syncthreads();  // also drains this workgroup's stores (vmcnt(0)) before the diagonal
    if (tid < h) {
        const int64_t i = i0 + tid;
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < kAsmThreads / kWave; ++v) s += s_part[v][tid];
        const double d = (C[i] + s) / w[i];  // device/mesh.py:455-457
        if (qdiag != nullptr) qdiag[i] = d;
        if (WANT_Q) Q[i * ldq + i] = static_cast<OutT>(d);
    }
}

// Number of workgroups for n rows: a whole number of rounds of kQGroupsPerCu workgroups per CU (every CU then holds
// the same number of equally long workgroups), plain ceil(n / TR) strips for small n.
inline int64_t balanced_groups(int64_t n, int tr) {
    const int64_t plain = ceil_div(n, tr);
    const int64_t slots = static_cast<int64_t>(device_cu_count()) * kQGroupsPerCu;
    // below ~8 rows per workgroup the per-row share of the column loads grows and the launch is latency bound
    // anyway: plain strips
    if (slots <= 0 || n < 8 * slots) return plain;
    return slots * ceil_div(n, static_cast<int64_t>(tr) * slots);
}

// ---------------------------------------------------------------------------------------
// System assembly: gather prep + strip kernel + sparse Laplacian fix-up.
// ---------------------------------------------------------------------------------------
__global__ void gather_vertex_kernel(const double *__restrict__ xy, const double *__restrict__ a,
                                     const double *__restrict__ b,
                                     const int64_t *__restrict__ idx, int64_t count,
                                     double *__restrict__ ox, double *__restrict__ oy,
                                     double *__restrict__ oa, double *__restrict__ ob,
                                     int64_t *__restrict__ oid, int32_t *__restrict__ pos) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t i = idx ? idx[k] : k;
    ox[k] = xy[2 * i];
    oy[k] = xy[2 * i + 1];
    if (oa) oa[k] = a[i];
    if (ob) ob[k] = b[i];
    oid[k] = i;
    if (pos) pos[i] = static_cast<int32_t>(k);
}

template <typename OutT, int TR>
__global__ __launch_bounds__(kAsmThreads) void system_assemble_kernel(
    const double *__restrict__ row_x, const double *__restrict__ row_y,
    const double *__restrict__ row_qd, const int64_t *__restrict__ row_id, int64_t nr,
    const double *__restrict__ col_x, const double *__restrict__ col_y,
    const double *__restrict__ col_w, const double *__restrict__ col_lam,
    const int64_t *__restrict__ col_id, int64_t nc, OutT sign, OutT *__restrict__ out,
    int64_t ldo, const int64_t *__restrict__ lap_indptr, const int64_t *__restrict__ lap_indices,
    const double *__restrict__ lap_data, const int32_t *__restrict__ col_pos,
    const double *__restrict__ row_rs, int lower_only) {
    // per row of the strip: site, diagonal entry Q_ii, sign * row scale (already in the output type), vertex id and the
    // COLUMN POSITION of that vertex (-1: not a column) -- the sweep finds the diagonal by comparing positions
    __shared__ double2 s_xy[TR];
    __shared__ double s_rs[TR];
    __shared__ double s_qd[TR];
    __shared__ int64_t s_id[TR];
    __shared__ int s_dpos[TR];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // lower_only: a strip's work grows with its row index; the long strips go first so that the short
    // ones fill the end of the launch.  Full matrices: the rows are dealt out evenly over the grid (a whole number
    // of rounds of kQGroupsPerCu workgroups per CU, see q_assemble_kernel), nr_end = first row of the next strip.
    const int64_t strip = lower_only ? static_cast<int64_t>(gridDim.x) - 1 - blockIdx.x : blockIdx.x;
    int64_t r0 = strip * TR, nr_end = nr;
    if (!lower_only) {
        int hh;
        strip_rows(nr, gridDim.x, blockIdx.x, &r0, &hh);
        nr_end = r0 + hh;
    } else if (nr_end > r0 + TR) {
        nr_end = r0 + TR;
    }
    const int h = static_cast<int>(nr_end - r0);   // rows of this workgroup, 1 .. TR (workgroup-uniform)
    if (tid < TR) {
        const int64_t r = (r0 + tid < nr) ? r0 + tid : nr - 1;
        s_xy[tid] = double2{row_x[r], row_y[r]};
        s_qd[tid] = row_qd[r];
        const int64_t id = row_id[r];
        s_id[tid] = id;
        s_dpos[tid] = col_pos[id];
        s_rs[tid] = static_cast<double>(sign * static_cast<OutT>(row_rs ? row_rs[r] : 1.0));
    }
    __syncthreads();

    // lower_only (rows == cols): nothing right of the strip's last diagonal entry is needed
    const int64_t c_end = (lower_only && r0 + TR < nc) ? r0 + TR : nc;
    // (full matrices: staggered sweeps, see q_assemble_kernel)
    const int64_t span = (c_end + 2 * kAsmThreads - 1) / (2 * kAsmThreads) * (2 * kAsmThreads);
    const int64_t shift = lower_only ? 0 : (static_cast<int64_t>(blockIdx.x) * 2 * kAsmThreads) % span;
    for (int64_t cc = 2 * tid; cc < span; cc += 2 * kAsmThreads) {
        int64_t c = cc + shift;
        if (c >= span) c -= span;
        if (c >= c_end) continue;
        const bool has1 = (c + 1 < nc);
        const double xj0 = col_x[c], yj0 = col_y[c];
        const double xj1 = has1 ? col_x[c + 1] : 0.0;
        const double yj1 = has1 ? col_y[c + 1] : 0.0;
        const OutT w0 = static_cast<OutT>(col_w[c]);
        const OutT w1 = has1 ? static_cast<OutT>(col_w[c + 1]) : OutT(0);
        const int ci = static_cast<int>(c);
        const bool pair = has1 || c + 1 < ldo;
        // (row-invariant quantities stay out of scalar registers, as in q_assemble_kernel: the store address advances
        // by one row per step)
        OutT *dst = out + r0 * ldo + c;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (r < h) {   // uniform branch
                int off = r * 16;   // one LDS read per use, hidden from loop-invariant code motion
                asm volatile("" : "+v"(off));
                const double2 pr = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s_xy) + off);
                const double dx0 = pr.x - xj0, dy0 = pr.y - yj0;
                const double dx1 = pr.x - xj1, dy1 = pr.y - yj1;
                double q0 = -inv_r3_over_4pi(__builtin_fma(dx0, dx0, dy0 * dy0));
                double q1 = -inv_r3_over_4pi(__builtin_fma(dx1, dx1, dy1 * dy1));
                const int dp = s_dpos[r];
                const double qd = s_qd[r];
                q0 = (ci == dp) ? qd : q0;  // Q_ii
                q1 = (ci + 1 == dp) ? qd : q1;
                // Q is cast to the solve dtype before the product (solver/utils.py:291,
                // solve_film.py:305): out = Q[ix,ix] * w[ix]  (column scaling).
                const OutT rs = static_cast<OutT>(s_rs[r]);  // sign * optional row scaling (w_i: S = W A)
                const OutT v0 = rs * (static_cast<OutT>(q0) * w0);
                const OutT v1 = has1 ? rs * (static_cast<OutT>(q1) * w1) : OutT(0);
                if (pair) {
                    store_pair<OutT>(dst, v0, v1);
                } else {
                    *dst = v0;
                }
                dst += ldo;
            }
        }
    }
    __syncthreads();  // stores of this strip are complete (vmcnt(0)) before the fix-up

    // - Lambda[j] * Del2[i, j]: one wave per row walks the CSR row (about 7 entries).
    for (int r = wave; r < TR; r += kAsmThreads / kWave) {
        if (r0 + r >= nr_end) break;
        const int64_t i = s_id[r];
        const int64_t p1 = lap_indptr[i + 1];
        for (int64_t p = lap_indptr[i] + lane; p < p1; p += kWave) {
            const int32_t c = col_pos[lap_indices[p]];
            if (c >= 0) {
                OutT *dst = out + (r0 + r) * ldo + c;
                const OutT t = static_cast<OutT>(col_lam[c]) * static_cast<OutT>(lap_data[p]);
                if (!lower_only || c <= r0 + r) *dst = *dst - static_cast<OutT>(s_rs[r]) * t;   // s_rs: sign * row scale
            }
        }
    }
}

}  // namespace ssa

using namespace ssa;

extern "C" int ssa_q_assemble(const double *xy, const double *w, const double *C, int64_t n,
                              void *Q, int64_t ldq, int dtype, double *qdiag, void *stream) {
    if (n <= 0 || !xy || !w || !C) return SSA_ERR_INVALID_ARGUMENT;
    if (Q && (ldq < n || (ldq & 1))) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    const dim3 grid(static_cast<unsigned>(balanced_groups(n, kQStripRows)));
    hipStream_t st = as_stream(stream);
    // (the row-sum-only form -- Q == nullptr: the factorizations need only the diagonal -- is its own instantiation:
    // no store code in the sweep)
    if (dtype == SSA_F64) {
        if (Q)
            hipLaunchKernelGGL((q_assemble_kernel<double, kQStripRows, true>), grid, dim3(kAsmThreads), 0, st, xy, w, C, n,
                               static_cast<double *>(Q), ldq, qdiag);
        else
            hipLaunchKernelGGL((q_assemble_kernel<double, kQStripRows, false>), grid, dim3(kAsmThreads), 0, st, xy, w, C, n,
                               static_cast<double *>(nullptr), ldq, qdiag);
    } else {
        if (Q)
            hipLaunchKernelGGL((q_assemble_kernel<float, kQStripRows, true>), grid, dim3(kAsmThreads), 0, st, xy, w, C, n,
                               static_cast<float *>(Q), ldq, qdiag);
        else
            hipLaunchKernelGGL((q_assemble_kernel<float, kQStripRows, false>), grid, dim3(kAsmThreads), 0, st, xy, w, C, n,
                               static_cast<float *>(nullptr), ldq, qdiag);
    }
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

extern "C" size_t ssa_system_assemble_workspace_bytes(int64_t n, int64_t nr, int64_t nc) {
    size_t b = 0;
    b += 5 * align_up(static_cast<size_t>(nr) * 8, 256);  // row x, y, qd, id, scale
    b += 5 * align_up(static_cast<size_t>(nc) * 8, 256);  // col x, y, w, lam, id
    b += align_up(static_cast<size_t>(n) * 4, 256);       // col_pos
    return b + 256;
}

extern "C" int ssa_system_assemble(const double *xy, const double *w, const double *qdiag,
                                   const double *Lambda, int64_t n, const int64_t *lap_indptr,
                                   const int64_t *lap_indices, const double *lap_data,
                                   const int64_t *rows, int64_t nr, const int64_t *cols,
                                   int64_t nc, double sign, const double *row_scale,
                                   int lower_only, void *out, int64_t ldo, int dtype,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (n <= 0 || nr <= 0 || nc <= 0 || !xy || !w || !qdiag || !Lambda || !out || !cols ||
        !lap_indptr || !lap_indices || !lap_data)
        return SSA_ERR_INVALID_ARGUMENT;
    if (ldo < nc) return SSA_ERR_INVALID_ARGUMENT;
    if ((ldo & 1) && nc > 1) return SSA_ERR_INVALID_ARGUMENT;
    if (lower_only && nr != nc) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_system_assemble_workspace_bytes(n, nr, nc))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    Carver cv(workspace);
    double *row_x = cv.take<double>(nr), *row_y = cv.take<double>(nr);
    double *row_qd = cv.take<double>(nr);
    int64_t *row_id = cv.take<int64_t>(nr);
    double *row_rs = cv.take<double>(nr);
    double *col_x = cv.take<double>(nc), *col_y = cv.take<double>(nc);
    double *col_w = cv.take<double>(nc), *col_lam = cv.take<double>(nc);
    int64_t *col_id = cv.take<int64_t>(nc);
    int32_t *col_pos = cv.take<int32_t>(n);
    if (hipMemsetAsync(col_pos, 0xFF, static_cast<size_t>(n) * 4, st) != hipSuccess)
        return SSA_ERR_HIP;
    const int tb = 256;
    hipLaunchKernelGGL(gather_vertex_kernel, dim3(ceil_div(nr, tb)), dim3(tb), 0, st, xy, qdiag,
                       row_scale, rows, nr, row_x, row_y, row_qd,
                       row_scale ? row_rs : (double *)nullptr, row_id, (int32_t *)nullptr);
    hipLaunchKernelGGL(gather_vertex_kernel, dim3(ceil_div(nc, tb)), dim3(tb), 0, st, xy, w,
                       Lambda, cols, nc, col_x, col_y, col_w, col_lam, col_id, col_pos);
    const double *rs = row_scale ? row_rs : nullptr;
    const dim3 grid(static_cast<unsigned>(lower_only ? ceil_div(nr, kStripRows) : balanced_groups(nr, kStripRows)));
    if (dtype == SSA_F64) {
        hipLaunchKernelGGL((system_assemble_kernel<double, kStripRows>), grid, dim3(kAsmThreads),
                           0, st, row_x, row_y, row_qd, row_id, nr, col_x, col_y, col_w, col_lam,
                           col_id, nc, sign, static_cast<double *>(out), ldo, lap_indptr,
                           lap_indices, lap_data, col_pos, rs, lower_only);
    } else {
        hipLaunchKernelGGL((system_assemble_kernel<float, kStripRows>), grid, dim3(kAsmThreads),
                           0, st, row_x, row_y, row_qd, row_id, nr, col_x, col_y, col_w, col_lam,
                           col_id, nc, static_cast<float>(sign), static_cast<float *>(out), ldo,
                           lap_indptr, lap_indices, lap_data, col_pos, rs, lower_only);
    }
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
