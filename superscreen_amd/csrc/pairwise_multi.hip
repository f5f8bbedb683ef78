// Multi-vector forms of the two all-pairs reductions of the solve loop, for sweeps that carry NV
// right-hand sides through one factorization (applied-field scans, BASELINE config 4):
//   self field      out[i, v] = alpha ( qdiag_i w_i g[i, v] - sum_{j != i} q_ij w_j g[j, v] )
//   film coupling   out[i, v] (+)= sum_j (1/4pi) a_j (Jx[j, v] dy - Jy[j, v] dx) |r_ij|^-3
// (solver/solve_film.py:565 and solver/solve.py:28-73, :508, applied to [n, nvec] operands.)
//
// With several vectors these are products  out = K(targets, sources) * C(sources, vectors)  whose
// left operand is a function of the coordinates only.  It is never stored: each lane evaluates ONE
// kernel value per 16 x 4 block, which is exactly the A-operand layout of v_mfma_f64_16x16x4_f64
// (lane l holds row l & 15, k = l >> 4), and the matrix cores multiply it by the staged charges.
// The FP64 work per pair drops from (~12 + 2 NV) vector ops to ~12 vector ops + NV / 32 MFMAs, and
// the LDS traffic from (2 + 2 NV) broadcast reads per pair to one operand read per MFMA.
//
// Work split: a workgroup (4 waves) owns 128 targets x one source slice x a chunk of <= 64 vectors;
// a wave owns 32 targets (two MFMA row blocks sharing every B operand).  Sources are staged 32 at a
// time: coordinates plus the charge planes  c a_j Jx | -c a_j Jy  (coupling) or  c w_j g  (self
// field), rows padded so the four k-rows of an operand read fall in distinct LDS banks.  Slice
// partials are combined by a second kernel in a fixed order (bitwise reproducible, no float atomics).
#include "common.hpp"
#include "mfma_traits.hpp"

namespace ssa {
namespace {

constexpr int kTB = 128;        // targets per workgroup (32 per wave)
constexpr int kKS = 32;         // sources per LDS stage
constexpr int kChunk = 64;      // vectors per launch
constexpr int kMaxSlicesM = 16;

// Source slices per target block: the grid should be a whole number of "rounds" of the workgroup
// slots of the chip (3 resident workgroups per CU: 41 KB of LDS, 160 VGPRs) -- 785 workgroups on 768
// slots take as long as 1536.  Picks the slice count with the best slot utilisation.
inline int pick_slices_m(int64_t nt, int64_t ns) {
    static int64_t slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 3 * static_cast<int64_t>(cus > 0 ? cus : 256);
    }
    const int64_t tb = ceil_div(nt, kTB);
    int64_t max_s = ceil_div(ns, 8 * kKS);
    if (max_s > kMaxSlicesM) max_s = kMaxSlicesM;
    int best = 1;
    double best_eff = 0.0;
    for (int64_t s = 1; s <= max_s; ++s) {
        const int64_t wgs = tb * s;
        const double eff = static_cast<double>(wgs) / static_cast<double>(ceil_div(wgs, slots) * slots);
        if (eff > best_eff + 1e-9) {  // ties: fewer slices, less partial traffic
            best_eff = eff;
            best = static_cast<int>(s);
        }
    }
    return best;
}

// NB = 16-column blocks of this chunk (1..4); SELF: one charge plane and the i == j exclusion.
template <typename T, int NB, bool SELF>
__global__ __launch_bounds__(256) void pair_mfma_kernel(const double *__restrict__ src_xy,
                                                        const void *__restrict__ src_scale,
                                                        const void *__restrict__ src_val, int64_t ns,
                                                        int64_t nvec, int64_t v0, int nv, int64_t slice_len,
                                                        const double *__restrict__ tgt_xy, int64_t nt, double dz2,
                                                        const int64_t *__restrict__ tgt_rows, double *__restrict__ partial) {
    using MF = Mfma<double>;
    using acc_t = MF::acc_t;
    constexpr int NV = NB * 16;
    constexpr int NVP = NV | 16;  // row stride = 128 B mod 256 B: k-rows alternate bank halves
    constexpr int NP = SELF ? 1 : 2;
    __shared__ __attribute__((aligned(16))) double s_b[NP][kKS][NVP];
    __shared__ __attribute__((aligned(16))) double s_xy[kKS][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t t_base = static_cast<int64_t>(blockIdx.x) * kTB + wave * 32;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < ns) ? j_begin + slice_len : ns;

    double xi[2], yi[2];
    int64_t ti[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int64_t t = t_base + mb * 16 + li;            // position in the target list
        const int64_t tc = (t < nt) ? t : nt - 1;
        const int64_t gi = tgt_rows ? tgt_rows[tc] : tc;    // vertex index of that target
        ti[mb] = (t < nt) ? gi : -1;
        xi[mb] = tgt_xy[2 * gi];
        yi[mb] = tgt_xy[2 * gi + 1];
    }
    acc_t acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = acc_t{0, 0, 0, 0};

    for (int64_t s0 = j_begin; s0 < j_end; s0 += kKS) {
        __syncthreads();
        if (tid < kKS) {
            const int64_t j = s0 + tid;
            const bool ok = j < j_end;  // padding sources sit far away and carry zero charge
            s_xy[tid][0] = ok ? src_xy[2 * j] : 1e30;
            s_xy[tid][1] = ok ? src_xy[2 * j + 1] : 1e30;
        }
        for (int e = tid; e < kKS * NV; e += 256) {
            const int k = e / NV, v = e % NV;
            const int64_t j = s0 + k;
            const bool ok = j < j_end && v < nv;
            if constexpr (SELF) {
                const double wj = ok ? static_cast<const double *>(src_scale)[j] : 0.0;
                const double gj = ok ? static_cast<double>(static_cast<const T *>(src_val)[j * nvec + v0 + v]) : 0.0;
                s_b[0][k][v] = kOneOver4Pi * (wj * gj);
            } else {
                double2 Jv = make_double2(0.0, 0.0);
                double ca = 0.0;
                if (ok) {
                    ca = kOneOver4Pi * static_cast<double>(static_cast<const T *>(src_scale)[j]);
                    Jv = *reinterpret_cast<const double2 *>(static_cast<const double *>(src_val) +
                                                            (j * nvec + v0 + v) * 2);
                }
                s_b[0][k][v] = ca * Jv.x;
                s_b[NP - 1][k][v] = -(ca * Jv.y);
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < kKS / 4; ++ks) {
            const int k = ks * 4 + lk;
            const double2 sxy = *reinterpret_cast<const double2 *>(&s_xy[k][0]);
            double a0[2], a1[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const double dx = xi[mb] - sxy.x, dy = yi[mb] - sxy.y;
                const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2));
                const double y3 = inv_r3(r2);
                if constexpr (SELF) {
                    a0[mb] = (s0 + k == ti[mb]) ? 0.0 : y3;
                    a1[mb] = 0.0;
                } else {
                    a0[mb] = dy * y3;
                    a1[mb] = dx * y3;
                }
            }
            double b0[NB], b1[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                b0[nb] = s_b[0][k][nb * 16 + li];
                if constexpr (!SELF) b1[nb] = s_b[NP - 1][k][nb * 16 + li];
            }
            // every accumulator once per sweep: 2 NB independent MFMAs between dependent ones
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = MF::run(a0[mb], b0[nb], acc[mb][nb]);
            if constexpr (!SELF) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = MF::run(a1[mb], b1[nb], acc[mb][nb]);
            }
        }
    }
    double *dst = partial + static_cast<int64_t>(blockIdx.y) * nt * NV;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = t_base + mb * 16 + MF::row(lane, r);
            if (i < nt) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) dst[i * NV + nb * 16 + li] = acc[mb][nb][r];
            }
        }
}

// Few vectors (<= 12): v_mfma_f64_4x4x4_4b_f64 = four independent 4 x 4 x 4 products per instruction,
// a quarter of the 16-column instruction's work.  Lane maps (found with tools/probes/mfma4x4_layout.hip):
// A: row block = (l >> 2) & 3, row i = l & 3, k = l >> 4  -- i.e. target l & 15, source l >> 4: the SAME
// pair-per-lane evaluation as above; B: block, k = l >> 4, column j = l & 3 (the charges do not depend on
// the target block: every block reads the same four values); D: row i = l >> 4, block, column j = l & 3,
// one value per lane.  NQ = groups of four vectors.
template <typename T, int NQ, bool SELF>
__global__ __launch_bounds__(256) void pair_mfma4_kernel(const double *__restrict__ src_xy,
                                                         const void *__restrict__ src_scale,
                                                         const void *__restrict__ src_val, int64_t ns,
                                                         int64_t nvec, int64_t v0, int nv, int64_t slice_len,
                                                         const double *__restrict__ tgt_xy, int64_t nt, double dz2,
                                                         const int64_t *__restrict__ tgt_rows, double *__restrict__ partial) {
    constexpr int NV = NQ * 4;
    constexpr int NP = SELF ? 1 : 2;
    __shared__ __attribute__((aligned(16))) double s_b[NP][kKS][NV];
    __shared__ __attribute__((aligned(16))) double s_xy[kKS][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4, lj = lane & 3;
    const int64_t t_base = static_cast<int64_t>(blockIdx.x) * kTB + wave * 32;
    const int64_t j_begin = static_cast<int64_t>(blockIdx.y) * slice_len;
    const int64_t j_end = (j_begin + slice_len < ns) ? j_begin + slice_len : ns;

    double xi[2], yi[2];
    int64_t ti[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int64_t t = t_base + mb * 16 + li;            // position in the target list
        const int64_t tc = (t < nt) ? t : nt - 1;
        const int64_t gi = tgt_rows ? tgt_rows[tc] : tc;    // vertex index of that target
        ti[mb] = (t < nt) ? gi : -1;
        xi[mb] = tgt_xy[2 * gi];
        yi[mb] = tgt_xy[2 * gi + 1];
    }
    double acc[2][NQ];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[mb][q] = 0.0;

    for (int64_t s0 = j_begin; s0 < j_end; s0 += kKS) {
        __syncthreads();
        if (tid < kKS) {
            const int64_t j = s0 + tid;
            const bool ok = j < j_end;
            s_xy[tid][0] = ok ? src_xy[2 * j] : 1e30;
            s_xy[tid][1] = ok ? src_xy[2 * j + 1] : 1e30;
        }
        for (int e = tid; e < kKS * NV; e += 256) {
            const int k = e / NV, v = e % NV;
            const int64_t j = s0 + k;
            const bool ok = j < j_end && v < nv;
            if constexpr (SELF) {
                const double wj = ok ? static_cast<const double *>(src_scale)[j] : 0.0;
                const double gj = ok ? static_cast<double>(static_cast<const T *>(src_val)[j * nvec + v0 + v]) : 0.0;
                s_b[0][k][v] = kOneOver4Pi * (wj * gj);
            } else {
                double2 Jv = make_double2(0.0, 0.0);
                double ca = 0.0;
                if (ok) {
                    ca = kOneOver4Pi * static_cast<double>(static_cast<const T *>(src_scale)[j]);
                    Jv = *reinterpret_cast<const double2 *>(static_cast<const double *>(src_val) +
                                                            (j * nvec + v0 + v) * 2);
                }
                s_b[0][k][v] = ca * Jv.x;
                s_b[NP - 1][k][v] = -(ca * Jv.y);
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < kKS / 4; ++ks) {
            const int k = ks * 4 + lk;
            const double2 sxy = *reinterpret_cast<const double2 *>(&s_xy[k][0]);
            double a0[2], a1[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const double dx = xi[mb] - sxy.x, dy = yi[mb] - sxy.y;
                const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz2));
                const double y3 = inv_r3(r2);
                if constexpr (SELF) {
                    a0[mb] = (s0 + k == ti[mb]) ? 0.0 : y3;
                    a1[mb] = 0.0;
                } else {
                    a0[mb] = dy * y3;
                    a1[mb] = dx * y3;
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double b0 = s_b[0][k][q * 4 + lj];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[mb][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[mb], b0, acc[mb][q], 0, 0, 0);
            }
            if constexpr (!SELF) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double b1 = s_b[NP - 1][k][q * 4 + lj];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc[mb][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[mb], b1, acc[mb][q], 0, 0, 0);
                }
            }
        }
    }
    double *dst = partial + static_cast<int64_t>(blockIdx.y) * nt * NV;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int64_t i = t_base + mb * 16 + ((lane >> 2) & 3) * 4 + lk;  // D: row l >> 4 of block (l >> 2) & 3
        if (i < nt) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) dst[i * NV + q * 4 + lj] = acc[mb][q];
        }
    }
}

template <typename T>
__global__ void self_field_multi_combine_kernel(const double *__restrict__ partial, int slices, int pstride,
                                                int64_t n, int64_t nvec, int64_t v0, int nv,
                                                const double *__restrict__ w, const double *__restrict__ qdiag,
                                                const T *__restrict__ g, double alpha,
                                                const int64_t *__restrict__ rows, T *__restrict__ out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n * nv) return;  // n = number of targets (rows of the list when there is one)
    const int64_t k = t / nv;
    const int v = static_cast<int>(t - k * nv);
    double s = 0.0;
    for (int q = 0; q < slices; ++q) s += partial[(static_cast<int64_t>(q) * n + k) * pstride + v];
    const int64_t i = rows ? rows[k] : k;
    const double d = qdiag[i] * (w[i] * static_cast<double>(g[i * nvec + v0 + v]));
    out[i * nvec + v0 + v] = static_cast<T>(alpha * (d - s));
}

template <typename T>
__global__ void biot_savart_multi_combine_kernel(const double *__restrict__ partial, int slices, int pstride,
                                                 int64_t nt, int64_t nvec, int64_t v0, int nv,
                                                 const int64_t *__restrict__ rows, T *__restrict__ out,
                                                 int accumulate) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt * nv) return;  // nt = number of targets (rows of the list when there is one)
    const int64_t i = t / nv;
    const int v = static_cast<int>(t - i * nv);
    double s = sum_strided(partial + i * pstride + v, slices, nt * pstride);
    T *dst = out + (rows ? rows[i] : i) * nvec + v0 + v;
    if (accumulate) s += static_cast<double>(*dst);
    *dst = static_cast<T>(s);
}

// launches the pair kernel for one chunk of nv <= 64 vectors; returns the row stride of `partial`
template <typename T, bool SELF>
int launch_pair_mfma(int nv, dim3 grid, hipStream_t st, const double *src_xy, const void *src_scale,
                     const void *src_val, int64_t ns, int64_t nvec, int64_t v0, int64_t slice_len,
                     const double *tgt_xy, int64_t nt, double dz2, const int64_t *tgt_rows, double *partial) {
#define SSA_PAIR_ARGS src_xy, src_scale, src_val, ns, nvec, v0, nv, slice_len, tgt_xy, nt, dz2, tgt_rows, partial
#define SSA_PAIR_CASE(NB) hipLaunchKernelGGL((pair_mfma_kernel<T, NB, SELF>), grid, dim3(256), 0, st, SSA_PAIR_ARGS)
#define SSA_PAIR4_CASE(NQ) hipLaunchKernelGGL((pair_mfma4_kernel<T, NQ, SELF>), grid, dim3(256), 0, st, SSA_PAIR_ARGS)
    if (nv <= 12) {
        const int nq = (nv + 3) / 4;
        switch (nq) {
            case 1: SSA_PAIR4_CASE(1); break;
            case 2: SSA_PAIR4_CASE(2); break;
            default: SSA_PAIR4_CASE(3); break;
        }
        return nq * 4;
    }
    const int nb = (nv + 15) / 16;
    switch (nb) {
        case 1: SSA_PAIR_CASE(1); break;
        case 2: SSA_PAIR_CASE(2); break;
        case 3: SSA_PAIR_CASE(3); break;
        default: SSA_PAIR_CASE(4); break;
    }
#undef SSA_PAIR_CASE
#undef SSA_PAIR4_CASE
#undef SSA_PAIR_ARGS
    return nb * 16;
}

}  // namespace
}  // namespace ssa

using namespace ssa;

extern "C" size_t ssa_pairwise_multi_workspace_bytes(int64_t nt) {
    return static_cast<size_t>(kMaxSlicesM) * static_cast<size_t>(nt) * kChunk * sizeof(double) + 256;
}

namespace {
int self_field_multi_impl(const double *xy, const double *w, const double *qdiag, const void *g, int64_t n,
                          int64_t nvec, const int64_t *rows, int64_t nr, void *out, double alpha, int dtype,
                          void *workspace, size_t workspace_bytes, void *stream) {
    if (!xy || !w || !qdiag || !g || !out || n <= 0 || nvec <= 0) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    const int64_t nt = rows ? nr : n;  // targets: the listed rows, or every vertex
    if (nt == 0) return SSA_OK;
    if (!workspace || workspace_bytes < ssa_pairwise_multi_workspace_bytes(nt)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices_m(nt, n);
    const int64_t slice_len = ceil_div(ceil_div(n, slices), kKS) * kKS;
    slices = static_cast<int>(ceil_div(n, slice_len));
    const dim3 grid(static_cast<unsigned>(ceil_div(nt, kTB)), slices);
    for (int64_t v0 = 0; v0 < nvec; v0 += kChunk) {
        const int nv = static_cast<int>((nvec - v0 < kChunk) ? nvec - v0 : kChunk);
        const dim3 cgrid(static_cast<unsigned>(ceil_div(nt * nv, 256)));
        if (dtype == SSA_F64) {
            const int ps = launch_pair_mfma<double, true>(nv, grid, st, xy, w, g, n, nvec, v0, slice_len, xy, nt, 0.0,
                                                          rows, partial);
            hipLaunchKernelGGL((self_field_multi_combine_kernel<double>), cgrid, dim3(256), 0, st, partial, slices,
                               ps, nt, nvec, v0, nv, w, qdiag, static_cast<const double *>(g), alpha, rows,
                               static_cast<double *>(out));
        } else {
            const int ps = launch_pair_mfma<float, true>(nv, grid, st, xy, w, g, n, nvec, v0, slice_len, xy, nt, 0.0,
                                                         rows, partial);
            hipLaunchKernelGGL((self_field_multi_combine_kernel<float>), cgrid, dim3(256), 0, st, partial, slices,
                               ps, nt, nvec, v0, nv, w, qdiag, static_cast<const float *>(g), alpha, rows,
                               static_cast<float *>(out));
        }
        SSA_RETURN_IF_LAUNCH_FAILED();
    }
    return SSA_OK;
}
}  // namespace

extern "C" int ssa_self_field_multi(const double *xy, const double *w, const double *qdiag, const void *g,
                                    int64_t n, int64_t nvec, void *out, double alpha, int dtype, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return self_field_multi_impl(xy, w, qdiag, g, n, nvec, nullptr, 0, out, alpha, dtype, workspace, workspace_bytes,
                                 stream);
}

extern "C" int ssa_self_field_multi_rows(const double *xy, const double *w, const double *qdiag, const void *g,
                                         int64_t n, int64_t nvec, const int64_t *rows, int64_t nr, void *out,
                                         double alpha, int dtype, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    if (nr < 0 || (nr > 0 && !rows)) return SSA_ERR_INVALID_ARGUMENT;
    if (nr == 0) return SSA_OK;
    return self_field_multi_impl(xy, w, qdiag, g, n, nvec, rows, nr, out, alpha, dtype, workspace, workspace_bytes,
                                 stream);
}

namespace {
int biot_savart_multi_impl(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                           const double *tgt_xy, int64_t nt_all, const int64_t *rows, int64_t nr, double dz,
                           int64_t nvec, void *out, int accumulate, int dtype, void *workspace,
                           size_t workspace_bytes, void *stream) {
    if (!src_xy || !src_areas || !src_J || !tgt_xy || !out || ns <= 0 || nt_all <= 0 || nvec <= 0)
        return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    const int64_t nt = rows ? nr : nt_all;  // targets: the listed rows of tgt_xy, or all of them
    if (nt == 0) return SSA_OK;
    if (!workspace || workspace_bytes < ssa_pairwise_multi_workspace_bytes(nt)) return SSA_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    int slices = pick_slices_m(nt, ns);
    const int64_t slice_len = ceil_div(ceil_div(ns, slices), kKS) * kKS;
    slices = static_cast<int>(ceil_div(ns, slice_len));
    const dim3 grid(static_cast<unsigned>(ceil_div(nt, kTB)), slices);
    for (int64_t v0 = 0; v0 < nvec; v0 += kChunk) {
        const int nv = static_cast<int>((nvec - v0 < kChunk) ? nvec - v0 : kChunk);
        const dim3 cgrid(static_cast<unsigned>(ceil_div(nt * nv, 256)));
        if (dtype == SSA_F64) {
            const int ps = launch_pair_mfma<double, false>(nv, grid, st, src_xy, src_areas, src_J, ns, nvec, v0,
                                                           slice_len, tgt_xy, nt, dz * dz, rows, partial);
            hipLaunchKernelGGL((biot_savart_multi_combine_kernel<double>), cgrid, dim3(256), 0, st, partial, slices,
                               ps, nt, nvec, v0, nv, rows, static_cast<double *>(out), accumulate);
        } else {
            const int ps = launch_pair_mfma<float, false>(nv, grid, st, src_xy, src_areas, src_J, ns, nvec, v0,
                                                          slice_len, tgt_xy, nt, dz * dz, rows, partial);
            hipLaunchKernelGGL((biot_savart_multi_combine_kernel<float>), cgrid, dim3(256), 0, st, partial, slices,
                               ps, nt, nvec, v0, nv, rows, static_cast<float *>(out), accumulate);
        }
        SSA_RETURN_IF_LAUNCH_FAILED();
    }
    return SSA_OK;
}
}  // namespace

extern "C" int ssa_biot_savart_multi(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                                     const double *tgt_xy, int64_t nt, double dz, int64_t nvec, void *out,
                                     int accumulate, int dtype, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    return biot_savart_multi_impl(src_xy, src_areas, src_J, ns, tgt_xy, nt, nullptr, 0, dz, nvec, out, accumulate,
                                  dtype, workspace, workspace_bytes, stream);
}

extern "C" int ssa_biot_savart_multi_rows(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                                          const double *tgt_xy, int64_t nt, const int64_t *rows, int64_t nr,
                                          double dz, int64_t nvec, void *out, int accumulate, int dtype,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    if (nr < 0 || (nr > 0 && !rows)) return SSA_ERR_INVALID_ARGUMENT;
    if (nr == 0) return SSA_OK;
    return biot_savart_multi_impl(src_xy, src_areas, src_J, ns, tgt_xy, nt, rows, nr, dz, nvec, out, accumulate, dtype,
                                  workspace, workspace_bytes, stream);
}
