// Dense Cholesky factor / solve for gfx950 (lower triangular, row-major, in place).
//
// Why it exists: for a homogeneous film the system the reference LU-factors,
//     A = Q[ix,ix] * w[ix] - Lambda * Del2[ix,ix]          (solver/solve_film.py:296-305),
// becomes SYMMETRIC after scaling its rows by the vertex areas:  S = diag(w) A, because
// Q is symmetric off the diagonal and Del2 = diag(1/w) L with L symmetric (fem.py:259-296).
// S has a positive diagonal and is strictly diagonally dominant, hence positive definite, so
//     gf = lu_solve(lu_factor(-A), h)   ==   - S^-1 (w .* h)
// can be computed with a Cholesky factorization S = L L^T: (1/3) n^3 flops instead of (2/3) n^3,
// no pivoting, no row interchanges, no triangular solve between panel and trailing update, and
// the trailing update is a SYRK on the lower triangle only (half the tiles, gemm_ops.hip).
// A non-positive pivot is reported through `info` (LAPACK ?potrf convention) and the host
// falls back to the LU path (lu.hip).
//
// Blocking mirrors lu.hip: 256-column outer panels whose trailing update is one MFMA SYRK with
// K = 256, 64-column sub-panels factored by a register-resident, fully unrolled kernel
// (diagonal block per workgroup in registers, the rows below forward-substituted one row per
// thread), in-panel updates by the NT GEMM.
#include <mutex>
#include <utility>

#include "common.hpp"

namespace ssa {

int gemm_op_f64(int opA, int opB, int lower, int64_t M, int64_t N, int64_t K, double alpha,
                const double *A, int64_t lda, const double *B, int64_t ldb, double beta, double *C,
                int64_t ldc, hipStream_t st);
int gemm_op_f32(int opA, int opB, int lower, int64_t M, int64_t N, int64_t K, double alpha,
                const float *A, int64_t lda, const float *B, int64_t ldb, double beta, float *C,
                int64_t ldc, hipStream_t st);
int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
             const double *B, int64_t ldb, double beta, double *C, int64_t ldc, hipStream_t st);
int gemm_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
             const float *B, int64_t ldb, double beta, float *C, int64_t ldc, hipStream_t st);
int gemv_f64(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
             double alpha, double beta, hipStream_t st);
int gemv_f32(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
             double alpha, double beta, hipStream_t st);
int trtri_lower_batched_f64(const double *Ablk, int64_t lda, int64_t a_stride, int kb, double *out,
                            int64_t ldo, int64_t o_stride, int batch, hipStream_t st);
int trtri_lower_batched_f32(const float *Ablk, int64_t lda, int64_t a_stride, int kb, float *out,
                            int64_t ldo, int64_t o_stride, int batch, hipStream_t st);

namespace {

constexpr int CNB = 256;  // outer panel
constexpr int CPW = 64;   // sub-panel
constexpr int kCholRows = 256;

inline int gemm_op_t(int oa, int ob, int lower, int64_t M, int64_t N, int64_t K, double alpha, const double *A,
                     int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                     hipStream_t st) {
    return gemm_op_f64(oa, ob, lower, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
inline int gemm_op_t(int oa, int ob, int lower, int64_t M, int64_t N, int64_t K, double alpha, const float *A,
                     int64_t lda, const float *B, int64_t ldb, double beta, float *C, int64_t ldc,
                     hipStream_t st) {
    return gemm_op_f32(oa, ob, lower, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
inline int gemm_nn_t(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc, hipStream_t st) {
    return gemm_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
inline int gemm_nn_t(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                     const float *B, int64_t ldb, double beta, float *C, int64_t ldc, hipStream_t st) {
    return gemm_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
inline int gemv_n_t(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
                    double alpha, double beta, hipStream_t st) {
    return gemv_f64(M, nr, nc, ldm, x, y, alpha, beta, st);
}
inline int gemv_n_t(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
                    double alpha, double beta, hipStream_t st) {
    return gemv_f32(M, nr, nc, ldm, x, y, alpha, beta, st);
}
inline int trtri_batched_t(const double *A, int64_t lda, int64_t as, int kb, double *out, int64_t ldo,
                           int64_t os, int batch, hipStream_t st) {
    return trtri_lower_batched_f64(A, lda, as, kb, out, ldo, os, batch, st);
}
inline int trtri_batched_t(const float *A, int64_t lda, int64_t as, int kb, float *out, int64_t ldo,
                           int64_t os, int batch, hipStream_t st) {
    return trtri_lower_batched_f32(A, lda, as, kb, out, ldo, os, batch, st);
}

constexpr int SNB = 1024;  // block size of the triangular solves (pre-inverted diagonal blocks)

// inv <- inverse of the lower triangular diagonal block L[r0 : r0 + sz, r0 : r0 + sz], built
// recursively:  inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]];  leaves (<= 256) by the
// LDS substitution kernel, the off-diagonal quadrants by two MFMA GEMMs.  `inv` has leading
// dimension ldi and is zero above the diagonal (the caller memsets it).
template <typename T>
int build_block_inverse(const T *L, int64_t lda, int64_t r0, int64_t sz, T *inv, int64_t ldi, T *tmp,
                        hipStream_t st) {
    if (sz <= 256) return SSA_OK;  // leaves are inverted by the caller, batched
    const int64_t h = (sz > 512) ? 512 : 256;
    int rc = build_block_inverse(L, lda, r0, h, inv, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    rc = build_block_inverse(L, lda, r0 + h, sz - h, inv + h * ldi + h, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    rc = gemm_nn_t(sz - h, h, h, 1.0, L + (r0 + h) * lda + r0, lda, inv, ldi, 0.0, tmp, 512, st);
    if (rc != SSA_OK) return rc;
    return gemm_nn_t(sz - h, h, sz - h, -1.0, inv + h * ldi + h, ldi, tmp, 512, 0.0, inv + h * ldi, ldi, st);
}

__device__ __forceinline__ double rsqrt_t(double x) { return rsqrt_f64(x); }
__device__ __forceinline__ float rsqrt_t(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
}

template <int S>
__device__ __forceinline__ int cquad_i32(int x) {
    constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);
    return __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, true);
}

// One column of the register-tiled 64 x 64 Cholesky: thread (r, q) holds a[r][q + 4 i].
template <typename T, int J>
__device__ __forceinline__ void chol_step(T (&v)[16], int r, int q, T *colbuf, bool &bad) {
    constexpr int I0 = J >> 2, S = J & 3;
    T *cb = colbuf + (J & 1) * CPW;
    if (q == S) cb[r] = v[I0];  // unscaled column J (rows above J publish stale values, never read)
    __syncthreads();
    const T d = cb[J];
    bad = bad || !(d > T(0));
    const T inv = rsqrt_t(d);
    const T lr = (r >= J) ? cb[r] * inv : T(0);  // l_rJ (= sqrt(d) for r == J)
    {   // register column I0: c = q + 4 I0;  c == J <=> q == S,  c > J <=> q > S
        const int c = q + 4 * I0;
        const T upd = v[I0] - lr * (cb[c] * inv);
        v[I0] = (q == S) ? ((r >= J) ? lr : v[I0]) : ((q > S && c <= r) ? upd : v[I0]);
    }
#pragma unroll
    for (int i = I0 + 1; i < 16; ++i) {
        const int c = q + 4 * i;
        const T upd = v[i] - lr * (cb[c] * inv);
        v[i] = (c <= r) ? upd : v[i];
    }
}
template <typename T, int... Js>
__device__ __forceinline__ void chol_all(T (&v)[16], int r, int q, int nsteps, T *colbuf, bool &bad,
                                         std::integer_sequence<int, Js...>) {
    ((Js < nsteps ? chol_step<T, Js>(v, r, q, colbuf, bad) : (void)0), ...);
}

// row x (in registers) <- row x * inv(L11^T):  x_J = (a_J - sum_{k<J} x_k L11[J][k]) / L11[J][J],
// right-looking; Ut[J][c] = L11[c][J] is read as an LDS broadcast.
template <typename T, int J>
__device__ __forceinline__ void chol_fwd_step(T (&a)[CPW], const T *Ut, const T *rdiag) {
    constexpr int TS = CPW + 1;
    const T l = a[J] * rdiag[J];
    a[J] = l;
#pragma unroll
    for (int c = J + 1; c < CPW; ++c) a[c] -= l * Ut[J * TS + c];
}
template <typename T, int... Js>
__device__ __forceinline__ void chol_fwd_all(T (&a)[CPW], const T *Ut, const T *rdiag, int jb,
                                             std::integer_sequence<int, Js...>) {
    ((Js < jb ? chol_fwd_step<T, Js>(a, Ut, rdiag) : (void)0), ...);
}

// MODE 0: fused (every workgroup factors the diagonal block, then substitutes its rows);
// MODE 1: diagonal block only (one workgroup; L11 -> A, L11^T and 1/diag -> `scratch`);
// MODE 2: rows only (L11^T and 1/diag come from `scratch`).  The split form keeps the many row
// workgroups short, which matters when they have to find room beside a running trailing update.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chol_panel_kernel(T *A, int64_t lda, int64_t j0, int m, int jb,
                                                         int32_t *info, T *scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TS = CPW + 1, SS = 16 + 1;
    T *Ut = reinterpret_cast<T *>(smem_raw);       // [64][TS]: first the block itself, then L11^T
    T *colbuf = Ut + CPW * TS;                      // [2][64]
    T *rdiag = colbuf + 2 * CPW;                    // [64]
    T *stage = rdiag + CPW;                         // [4 waves][64][SS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int row_base = g * kCholRows;
    const int nt = min(CPW, m);
    T *Ap = A + j0 * lda + j0;
    // This kernel is a chain of short dependent steps and usually shares its SIMDs with the MFMA
    // waves of the trailing update (look-ahead): ask the instruction arbiter to serve it first.
    __builtin_amdgcn_s_setprio(3);

    if (MODE == 2) {
        for (int e = tid; e < CPW * TS + CPW; e += 256) {
            if (e < CPW * TS) Ut[e] = scratch[e];
            else rdiag[e - CPW * TS] = scratch[e];
        }
    } else {
    for (int rr = wave; rr < CPW; rr += 4)
        Ut[rr * TS + lane] = (rr < nt && lane < jb) ? Ap[static_cast<int64_t>(rr) * lda + lane] : T(0);
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    T v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = Ut[r * TS + q + 4 * i];
    // rows beyond the block (last, partial sub-panel) get a unit diagonal so that the padding
    // stays positive definite
    if (r >= nt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (q + 4 * i == r) ? T(1) : T(0);
    }
    bool bad = false;
    chol_all<T>(v, r, q, min(jb, nt), colbuf, bad, std::make_integer_sequence<int, CPW>{});
    __syncthreads();
    // Ut[J][c] = L11[c][J] (c >= J), zero elsewhere; reciprocals of the diagonal
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = q + 4 * i;
        if (c <= r) Ut[c * TS + r] = v[i];
    }
    __syncthreads();
    if (tid < CPW) {
        const T d = Ut[tid * TS + tid];
        rdiag[tid] = (d != T(0)) ? T(1) / d : T(1);
    }
    if (g == 0) {
        if (bad && tid == 0 && *info == 0) *info = static_cast<int32_t>(j0 + 1);
        // L11 (lower triangle) back to A
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = q + 4 * i;
            if (r < nt && c <= r && c < jb) Ap[static_cast<int64_t>(r) * lda + c] = v[i];
        }
    }
    if (MODE == 1) {
        __syncthreads();
        for (int e = tid; e < CPW * TS + CPW; e += 256)
            scratch[e] = (e < CPW * TS) ? Ut[e] : rdiag[e - CPW * TS];
        return;
    }
    }  // MODE != 2
    __syncthreads();

    // rows below the block: HBM -> (LDS transpose, 16 columns at a time: 35 KB of staging keeps the
    // kernel at 70 KB of LDS so that it can share a CU with a trailing-update workgroup when the
    // factorization runs panel k+1 beside SYRK k) -> registers -> substitute -> back
    T *st = stage + wave * 64 * SS;
    const int wrow0 = row_base + wave * 64;
    T arow[CPW];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rr = 4 * it + (lane >> 4), c = 16 * h + (lane & 15);
            const int pr = wrow0 + rr;
            T val = T(0);
            if (pr < m && pr >= nt && c < jb) val = Ap[static_cast<int64_t>(pr) * lda + c];
            st[rr * SS + (lane & 15)] = val;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) arow[16 * h + i] = st[lane * SS + i];
    }
    chol_fwd_all<T>(arow, Ut, rdiag, jb, std::make_integer_sequence<int, CPW>{});
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[lane * SS + i] = arow[16 * h + i];
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rr = 4 * it + (lane >> 4), c = 16 * h + (lane & 15);
            const int pr = wrow0 + rr;
            if (pr < m && pr >= nt && c < jb) Ap[static_cast<int64_t>(pr) * lda + c] = st[rr * SS + (lane & 15)];
        }
    }
}

template <typename T>
constexpr size_t chol_panel_smem() {
    return sizeof(T) * (CPW * (CPW + 1) + 2 * CPW + CPW + 4 * 64 * 17) + 64;
}

// y[c] = alpha * sum_r M[r][c] x[r] + beta * y[c]  (M is nr x nc row-major): the transposed GEMV of
// the backward substitution with L^T.  Workgroup = 128 columns; each wave sums a quarter of the
// rows with 16-byte loads (lane <-> 2 columns, no cross-lane reduction), the four partial sums
// meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_kernel(const T *__restrict__ M, int64_t nr, int64_t nc,
                                                     int64_t ldm, const T *__restrict__ x,
                                                     T *__restrict__ y, T alpha, T beta) {
    __shared__ T part[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = static_cast<int64_t>(blockIdx.x) * 128 + 2 * lane;
    T s0 = T(0), s1 = T(0);
    const int64_t rows_per = (nr + 3) / 4;
    const int64_t r_begin = wave * rows_per, r_end = (r_begin + rows_per < nr) ? r_begin + rows_per : nr;
    if (c0 + 1 < nc) {
#pragma unroll 8
        for (int64_t r = r_begin; r < r_end; ++r) {
            const T xv = x[r];
            const T *p = M + r * ldm + c0;
            s0 += p[0] * xv;
            s1 += p[1] * xv;
        }
    } else if (c0 < nc) {
        for (int64_t r = r_begin; r < r_end; ++r) s0 += M[r * ldm + c0] * x[r];
    }
    part[wave][2 * lane] = s0;
    part[wave][2 * lane + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int64_t c = static_cast<int64_t>(blockIdx.x) * 128 + threadIdx.x;
        if (c < nc) {
            const T s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
            T out = alpha * s;
            if (beta != T(0)) out += beta * y[c];
            y[c] = out;
        }
    }
}

template <typename T>
int gemv_trans(const T *M, int64_t nr, int64_t nc, int64_t ldm, const T *x, T *y, double alpha, double beta,
               hipStream_t st) {
    if (nr <= 0 || nc <= 0) return SSA_OK;
    hipLaunchKernelGGL((gemv_t_kernel<T>), dim3(static_cast<unsigned>(ceil_div(nc, 128))), dim3(256), 0, st, M,
                       nr, nc, ldm, x, y, static_cast<T>(alpha), static_cast<T>(beta));
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

// One side stream + two events per concurrently factored matrix (look-ahead lanes).
struct CholLane {
    hipStream_t side = nullptr;
    hipEvent_t ev_strip = nullptr, ev_panel = nullptr;
};
constexpr int kMaxLanes = 16;

inline int get_lanes(int count, CholLane **out) {
    static CholLane lanes[kMaxLanes];
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return SSA_ERR_HIP;
    for (int i = 0; i < count; ++i) {
        if (lanes[i].side != nullptr) continue;
        if (hipStreamCreateWithPriority(&lanes[i].side, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_strip, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_panel, hipEventDisableTiming) != hipSuccess)
            return SSA_ERR_HIP;
    }
    *out = lanes;
    return SSA_OK;
}

template <typename T>
struct CholJob {
    T *A;
    int64_t n, lda;
    int32_t *info;
    T *aux;
};

// One outer panel of one matrix: 64-column sub-panels, each followed by the NT update of the
// rest of the panel.  `scratch` = L11^T + 1/diag of the current sub-panel.
template <typename T>
int chol_factor_panel(const CholJob<T> &J, int64_t k0, hipStream_t s) {
    T *A = J.A;
    const int64_t n = J.n, lda = J.lda;
    T *scratch = J.aux + ceil_div(n, SNB) * SNB * SNB + 512 * 512;
    const int64_t kb = (n - k0 < CNB) ? n - k0 : CNB;
    for (int64_t j0 = k0; j0 < k0 + kb; j0 += CPW) {
        const int64_t jb = (k0 + kb - j0 < CPW) ? k0 + kb - j0 : CPW;
        const int64_t m = n - j0;
        // fused form (MODE 0); the split form (MODE 1 + 2) measured the same under look-ahead,
        // where the f64 VALU chain is slowed by the co-resident f64 MFMA waves either way
        hipLaunchKernelGGL((chol_panel_kernel<T, 0>), dim3(static_cast<unsigned>(ceil_div(m, kCholRows))),
                           dim3(256), chol_panel_smem<T>(), s, A, lda, j0, static_cast<int>(m),
                           static_cast<int>(jb), J.info, scratch);
        SSA_RETURN_IF_LAUNCH_FAILED();
        const int64_t rest = (k0 + kb) - (j0 + jb);
        const int64_t mm = n - (j0 + jb);
        if (rest > 0 && mm > 0) {
            const T *P = A + (j0 + jb) * lda + j0;  // rows below the sub-panel, its 64 columns
            const int r2 = gemm_op_t(0, 1, 0, mm, rest, jb, -1.0, P, lda, P, lda, 1.0,
                                     A + (j0 + jb) * lda + (j0 + jb), lda, s);
            if (r2 != SSA_OK) return r2;
        }
    }
    return SSA_OK;
}

// aux = inverses of the SNB x SNB diagonal blocks of L for the solve phase (+ GEMM scratch)
template <typename T>
int chol_build_inverses(const CholJob<T> &J, hipStream_t st) {
    const T *A = J.A;
    T *aux = J.aux;
    const int64_t n = J.n, lda = J.lda;
    int rc;
    const int64_t nblk = ceil_div(n, SNB);
    if (hipMemsetAsync(aux, 0, static_cast<size_t>(nblk) * SNB * SNB * sizeof(T), st) != hipSuccess)
        return SSA_ERR_HIP;
    T *tmp = aux + nblk * SNB * SNB;
    // leaves: the 256 x 256 diagonal blocks, batched by their position s inside the SNB block
    for (int s = 0; s < SNB / 256; ++s) {
        const int64_t first = static_cast<int64_t>(s) * 256;  // first leaf of this class
        if (first >= n) break;
        const int64_t count_full = (n - first >= 256) ? ((n - first - 256) / SNB + 1) : 0;
        if (count_full > 0) {
            rc = trtri_batched_t(A + first * (lda + 1), lda, static_cast<int64_t>(SNB) * (lda + 1), 256,
                                 aux + first * (SNB + 1), static_cast<int64_t>(SNB),
                                 static_cast<int64_t>(SNB) * SNB, static_cast<int>(count_full), st);
            if (rc != SSA_OK) return rc;
        }
    }
    if (n % 256 != 0) {  // the last, partial leaf
        const int64_t r0 = n / 256 * 256, Jb = r0 / SNB, off = r0 - Jb * SNB;
        rc = trtri_batched_t(A + r0 * (lda + 1), lda, 0, static_cast<int>(n - r0),
                             aux + Jb * SNB * SNB + off * (SNB + 1), static_cast<int64_t>(SNB), 0, 1, st);
        if (rc != SSA_OK) return rc;
    }
    for (int64_t k = 0; k < nblk; ++k) {
        const int64_t r0 = k * SNB, sz = (n - r0 < SNB) ? n - r0 : SNB;
        rc = build_block_inverse(A, lda, r0, sz, aux + k * SNB * SNB, static_cast<int64_t>(SNB), tmp, st);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

// Factor `count` independent matrices (the films of a device) in one interleaved schedule.
//
// Every MFMA trailing update (SYRK) of every matrix goes to the caller's stream, round-robin over
// the matrices, so the updates never compete with each other for the chip.  The latency-bound
// panel chain of matrix i runs on its own high-priority side stream: panel k+1 of matrix i is
// factored while the rest of update k of matrix i AND the updates of the other matrices run.
// With one matrix this is plain look-ahead (the chain is then the critical path for n ~ 20k);
// with two or more the chains hide behind the other films' updates.
//
// Per matrix and outer step k:   strip   C[:, 0:256]   -= P P[0:256]^T       (caller's stream)
//                                panel k+1 factored                           (side stream, after strip)
//                                rest    C[256:, 256:] -= P2 P2^T  (lower)    (caller's stream)
template <typename T>
int potrf_batch(const CholJob<T> *jobs, int count, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_panel_kernel<T, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(chol_panel_smem<T>())) != hipSuccess)
            return SSA_ERR_HIP;
        attr_set = true;
    }
    if (count <= 0 || count > kMaxLanes) return SSA_ERR_INVALID_ARGUMENT;
    CholLane *lanes = nullptr;
    int rc = get_lanes(count, &lanes);
    if (rc != SSA_OK) return rc;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        const CholJob<T> &J = jobs[i];
        CholLane &ln = lanes[i];
        if (J.n > nmax) nmax = J.n;
        if (hipMemsetAsync(J.info, 0, sizeof(int32_t), st) != hipSuccess) return SSA_ERR_HIP;
        // fork: the side stream starts after everything enqueued so far on the caller's stream
        if (hipEventRecord(ln.ev_strip, st) != hipSuccess || hipStreamWaitEvent(ln.side, ln.ev_strip, 0) != hipSuccess)
            return SSA_ERR_HIP;
        rc = chol_factor_panel(J, 0, ln.side);
        if (rc != SSA_OK) return rc;
        if (hipEventRecord(ln.ev_panel, ln.side) != hipSuccess) return SSA_ERR_HIP;
    }
    for (int64_t k0 = 0; k0 + CNB < nmax; k0 += CNB) {
        for (int i = 0; i < count; ++i) {
            const CholJob<T> &J = jobs[i];
            CholLane &ln = lanes[i];
            if (k0 + CNB >= J.n) continue;
            const int64_t right = J.n - k0 - CNB;             // order of the trailing matrix
            const int64_t nw = (right < CNB) ? right : CNB;   // width of the next panel
            const T *P = J.A + (k0 + CNB) * J.lda + k0;       // panel k below its diagonal block
            T *C = J.A + (k0 + CNB) * J.lda + (k0 + CNB);
            if (hipStreamWaitEvent(st, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;  // panel k done
            rc = gemm_op_t(0, 1, 0, right, nw, CNB, -1.0, P, J.lda, P, J.lda, 1.0, C, J.lda, st);  // strip
            if (rc != SSA_OK) return rc;
            if (hipEventRecord(ln.ev_strip, st) != hipSuccess ||
                hipStreamWaitEvent(ln.side, ln.ev_strip, 0) != hipSuccess)
                return SSA_ERR_HIP;
            rc = chol_factor_panel(J, k0 + CNB, ln.side);
            if (rc != SSA_OK) return rc;
            if (hipEventRecord(ln.ev_panel, ln.side) != hipSuccess) return SSA_ERR_HIP;
            if (right > nw) {  // rest of the trailing update: lower tiles of the (right - nw) block
                const T *P2 = P + nw * J.lda;
                rc = gemm_op_t(0, 1, 1, right - nw, right - nw, CNB, -1.0, P2, J.lda, P2, J.lda, 1.0,
                               C + nw * J.lda + nw, J.lda, st);
                if (rc != SSA_OK) return rc;
            }
        }
    }
    for (int i = 0; i < count; ++i) {  // join, then the inverses of the diagonal blocks
        if (hipStreamWaitEvent(st, lanes[i].ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;
    }
    for (int i = 0; i < count; ++i) {
        rc = chol_build_inverses(jobs[i], st);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

// L L^T X = B.  Single right-hand side: GEMV / transposed-GEMV chain over SNB-row blocks;
// several: the same recurrence on the MFMA GEMMs (NN forward, TN backward).
template <typename T>
int potrs(const T *L, int64_t n, int64_t lda, const T *aux, T *B, int64_t nrhs, int64_t ldb, T *X,
          hipStream_t st) {
    const int64_t nblk = ceil_div(n, SNB);
    const int64_t ldx = nrhs;
    int rc;
    for (int64_t k = 0; k < nblk; ++k) {  // forward: L y = b
        const int64_t r0 = k * SNB, kb = (n - r0 < SNB) ? n - r0 : SNB;
        const int64_t below = n - r0 - kb;
        const T *inv = aux + k * SNB * SNB;
        if (nrhs == 1 && ldb == 1) {
            rc = gemv_n_t(inv, kb, kb, SNB, B + r0, X + r0, 1.0, 0.0, st);
            if (rc == SSA_OK && below > 0)
                rc = gemv_n_t(L + (r0 + kb) * lda + r0, below, kb, lda, X + r0, B + r0 + kb, -1.0, 1.0, st);
        } else {
            rc = gemm_nn_t(kb, nrhs, kb, 1.0, inv, SNB, B + r0 * ldb, ldb, 0.0, X + r0 * ldx, ldx, st);
            if (rc == SSA_OK && below > 0)
                rc = gemm_nn_t(below, nrhs, kb, -1.0, L + (r0 + kb) * lda + r0, lda, X + r0 * ldx, ldx, 1.0,
                               B + (r0 + kb) * ldb, ldb, st);
        }
        if (rc != SSA_OK) return rc;
    }
    for (int64_t k = nblk - 1; k >= 0; --k) {  // backward: L^T x = y   (y lives in X, x goes to B)
        const int64_t r0 = k * SNB, kb = (n - r0 < SNB) ? n - r0 : SNB;
        const T *inv = aux + k * SNB * SNB;
        if (nrhs == 1 && ldb == 1) {
            rc = gemv_trans(inv, kb, kb, SNB, X + r0, B + r0, 1.0, 0.0, st);
            if (rc == SSA_OK && r0 > 0)
                rc = gemv_trans(L + r0 * lda, kb, r0, lda, B + r0, X, -1.0, 1.0, st);
        } else {
            rc = gemm_op_t(1, 0, 0, kb, nrhs, kb, 1.0, inv, SNB, X + r0 * ldx, ldx, 0.0, B + r0 * ldb, ldb, st);
            if (rc == SSA_OK && r0 > 0)
                rc = gemm_op_t(1, 0, 0, r0, nrhs, kb, -1.0, L + r0 * lda, lda, B + r0 * ldb, ldb, 1.0, X, ldx, st);
        }
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

}  // namespace
}  // namespace ssa

using namespace ssa;

extern "C" size_t ssa_chol_aux_bytes(int64_t n, int dtype) {
    const int64_t np = ceil_div(n, CNB) * CNB;
    return (static_cast<size_t>(ceil_div(np, SNB)) * SNB * SNB + 512 * 512 + 2 * CPW * (CPW + 2)) *
           (dtype == SSA_F64 ? 8 : 4);
}

extern "C" int64_t ssa_chol_padded_n(int64_t n) { return ceil_div(n, CNB) * CNB; }

namespace ssa {
namespace {
// rows n .. np-1 of the padded matrix: zero with a unit diagonal (keeps it positive definite and
// makes every panel / SYRK tile a full one)
template <typename T>
__global__ void pad_identity_kernel(T *A, int64_t lda, int64_t n, int64_t np) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t r = n + blockIdx.y;
    if (c < np) A[r * lda + c] = (c == r) ? T(1) : T(0);
}
template <typename T>
int potrf_padded_batch(int count, void *const *A, const int64_t *n, const int64_t *lda, int32_t *const *info,
                       void *const *aux, hipStream_t st) {
    CholJob<T> jobs[kMaxLanes];
    for (int i = 0; i < count; ++i) {
        const int64_t np = ssa_chol_padded_n(n[i]);
        T *Ai = static_cast<T *>(A[i]);
        if (np > n[i]) {
            hipLaunchKernelGGL((pad_identity_kernel<T>), dim3(static_cast<unsigned>(ceil_div(np, 256)),
                                                              static_cast<unsigned>(np - n[i])),
                               dim3(256), 0, st, Ai, lda[i], n[i], np);
            SSA_RETURN_IF_LAUNCH_FAILED();
        }
        jobs[i] = CholJob<T>{Ai, np, lda[i], info[i], static_cast<T *>(aux[i])};
    }
    return potrf_batch<T>(jobs, count, st);
}
template <typename T>
int potrs_padded(const T *L, int64_t n, int64_t lda, const T *aux, T *B, int64_t nrhs, int64_t ldb,
                 T *ws, hipStream_t st) {
    const int64_t np = ssa_chol_padded_n(n);
    T *X = ws, *Bp = ws + np * nrhs;
    if (hipMemcpy2DAsync(Bp, nrhs * sizeof(T), B, ldb * sizeof(T), nrhs * sizeof(T), n, hipMemcpyDeviceToDevice,
                         st) != hipSuccess)
        return SSA_ERR_HIP;
    if (np > n && hipMemsetAsync(Bp + n * nrhs, 0, (np - n) * nrhs * sizeof(T), st) != hipSuccess)
        return SSA_ERR_HIP;
    const int rc = potrs<T>(L, np, lda, aux, Bp, nrhs, nrhs, X, st);
    if (rc != SSA_OK) return rc;
    if (hipMemcpy2DAsync(B, ldb * sizeof(T), Bp, nrhs * sizeof(T), nrhs * sizeof(T), n, hipMemcpyDeviceToDevice,
                         st) != hipSuccess)
        return SSA_ERR_HIP;
    return SSA_OK;
}
}  // namespace
}  // namespace ssa

extern "C" int ssa_chol_factor_batch(int count, void *const *A, const int64_t *n, const int64_t *lda,
                                     int32_t *const *info, void *const *aux, int dtype, void *stream) {
    if (count <= 0 || !A || !n || !lda || !info || !aux) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < count; ++i)
        if (!A[i] || !info[i] || !aux[i] || n[i] <= 0 || lda[i] < ssa_chol_padded_n(n[i]))
            return SSA_ERR_INVALID_ARGUMENT;
    // more matrices than look-ahead lanes: groups of kMaxLanes, one after the other
    for (int first = 0; first < count; first += kMaxLanes) {
        const int c = (count - first < kMaxLanes) ? count - first : kMaxLanes;
        const int rc = (dtype == SSA_F64)
                           ? potrf_padded_batch<double>(c, A + first, n + first, lda + first, info + first,
                                                        aux + first, as_stream(stream))
                           : potrf_padded_batch<float>(c, A + first, n + first, lda + first, info + first,
                                                       aux + first, as_stream(stream));
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

extern "C" int ssa_chol_factor(void *A, int64_t n, int64_t lda, int32_t *info, void *aux, int dtype,
                               void *stream) {
    return ssa_chol_factor_batch(1, &A, &n, &lda, &info, &aux, dtype, stream);
}

extern "C" size_t ssa_chol_solve_workspace_bytes(int64_t n, int64_t nrhs, int dtype) {
    return 2 * static_cast<size_t>(ssa_chol_padded_n(n)) * static_cast<size_t>(nrhs) * (dtype == SSA_F64 ? 8 : 4) +
           256;
}

extern "C" int ssa_chol_solve(const void *L, int64_t n, int64_t lda, const void *aux, void *B,
                              int64_t nrhs, int64_t ldb, int dtype, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!L || !aux || !B || n <= 0 || nrhs <= 0 || lda < n || ldb < nrhs) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_chol_solve_workspace_bytes(n, nrhs, dtype))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    if (dtype == SSA_F64)
        return potrs_padded<double>(static_cast<const double *>(L), n, lda, static_cast<const double *>(aux),
                                    static_cast<double *>(B), nrhs, ldb, static_cast<double *>(workspace),
                                    as_stream(stream));
    return potrs_padded<float>(static_cast<const float *>(L), n, lda, static_cast<const float *>(aux),
                               static_cast<float *>(B), nrhs, ldb, static_cast<float *>(workspace),
                               as_stream(stream));
}
