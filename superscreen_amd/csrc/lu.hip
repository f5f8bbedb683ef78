// Dense LU factor / solve for gfx950 (replaces scipy.linalg.lu_factor / lu_solve = LAPACK
// ?getrf / ?getrs at solver/solve_film.py:279 and :530).  Row-major storage, partial (row)
// pivoting, right-looking, two-level blocking:
//
//   outer panel NB = 256 columns  -> trailing update C -= L21 * U12 is an MFMA GEMM with
//                                    K = 256 (compute-bound: 2K/16 = 32 flop per C byte)
//   sub-panel   PW = 64 columns   -> factored speculatively without interchanges and verified
//                                    exactly (lu_panel_spec3_kernel, lu_spec3.hpp); only if
//                                    the check fails, redone by ONE cooperative persistent
//                                    kernel (lu_panel_kernel, below); the rest of the outer
//                                    panel is then updated by laswp + trsm + a skinny GEMM
//
// This is the LAPACK-compatible route (method="lu", LinearSystem.lu_piv, fallback when
// diag(w) A is not positive definite); homogeneous films go through chol.hip by default.
//
// lu_panel_kernel: the (m x 64) sub-panel is cut into row slabs of <= 256 rows, one
// workgroup per slab, each slab living in LDS (column-major, lane <-> row, odd row stride =>
// conflict-free).  Per column: local |max| search -> every workgroup publishes its best
// candidate ROW (write-through 8-byte stores) -> one sharded arrival counter -> every
// workgroup picks the same global pivot, reads the winner's row and eliminates its own slab.
// Rows never move inside the kernel: the LAPACK interchange sequence is tracked as a
// permutation (every row knows its "current position") and applied when slabs are written
// back, so the result -- including ipiv and tie-breaking on the lowest current position, as
// i?amax does -- is what ?getf2 would produce.  Inter-workgroup hand-offs follow the CDNA4
// recipe: sc1 payload stores, per-wave vmcnt(0) drain, relaxed agent-scope counter,
// relaxed poll, sc1 loads; every spin is bounded.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>

#include "chain_streams.hpp"
#include "common.hpp"
#include "lu_diag.hpp"

namespace ssa {

int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
             const double *B, int64_t ldb, double beta, double *C, int64_t ldc, hipStream_t st);
int gemm_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
             const float *B, int64_t ldb, double beta, float *C, int64_t ldc, hipStream_t st);

int gemm_splitk_pick(int64_t M, int64_t N, int64_t K);
size_t gemm_rhs_partial_elems(int64_t m_max, int64_t nrhs);
bool gemm_skinny_ok(int64_t N, int64_t K, const void *A, int64_t lda);
int gemm_skinny_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                    int64_t ldb, double beta, double *C, int64_t ldc, int tri, double *partial, hipStream_t st);
int gemm_splitk_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                    int64_t ldb, double beta, double *C, int64_t ldc, int splits, double *partial, hipStream_t st);
int gemm_splitk_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda, const float *B,
                    int64_t ldb, double beta, float *C, int64_t ldc, int splits, float *partial, hipStream_t st);
// product of the multi-right-hand-side solves: <= 64 columns stream the factor once (skinny kernel,
// tri = 1 / 2: A lower / upper triangular), more columns run on the tiled GEMM with split-K
inline int gemm_rhs_t(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                      int64_t ldb, double beta, double *C, int64_t ldc, int tri, double *partial, hipStream_t st) {
    if (gemm_skinny_ok(N, K, A, lda))
        return gemm_skinny_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, partial, st);
    return gemm_splitk_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, gemm_splitk_pick(M, N, K), partial, st);
}
inline int gemm_rhs_t(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda, const float *B,
                      int64_t ldb, double beta, float *C, int64_t ldc, int, float *partial, hipStream_t st) {
    return gemm_splitk_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, gemm_splitk_pick(M, N, K), partial, st);
}
template <typename T>
int gemm_t(int64_t M, int64_t N, int64_t K, double alpha, const T *A, int64_t lda, const T *B,
           int64_t ldb, double beta, T *C, int64_t ldc, hipStream_t st);
template <>
int gemm_t<double>(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                   const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                   hipStream_t st) {
    return gemm_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
template <>
int gemm_t<float>(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                  const float *B, int64_t ldb, double beta, float *C, int64_t ldc,
                  hipStream_t st) {
    return gemm_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}

int gemv_f64(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
             double alpha, double beta, hipStream_t st);
int gemv_f32(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
             double alpha, double beta, hipStream_t st);
int trmv_f64(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y, double alpha,
             double beta, int tri, hipStream_t st);
int trmv_f32(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y, double alpha,
             double beta, int tri, hipStream_t st);
int gemm_batched_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int batch1,
                     int batch2, const int64_t *strides, int tri, hipStream_t st);
int gemm_batched_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                     const float *B, int64_t ldb, double beta, float *C, int64_t ldc, int batch1,
                     int batch2, const int64_t *strides, int tri, hipStream_t st);
inline int trmv_x(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y, double alpha,
                  double beta, int tri, hipStream_t st) {
    return trmv_f64(M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int trmv_x(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y, double alpha,
                  double beta, int tri, hipStream_t st) {
    return trmv_f32(M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int gemm_batched_x(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                          const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int b1, int b2,
                          const int64_t *strides, int tri, hipStream_t st) {
    return gemm_batched_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, st);
}
inline int gemm_batched_x(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                          const float *B, int64_t ldb, double beta, float *C, int64_t ldc, int b1, int b2,
                          const int64_t *strides, int tri, hipStream_t st) {
    return gemm_batched_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, st);
}

template <typename T>
int gemv_t(const T *M, int64_t nr, int64_t nc, int64_t ldm, const T *x, T *y, double alpha,
           double beta, hipStream_t st);
template <>
int gemv_t<double>(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x,
                   double *y, double alpha, double beta, hipStream_t st) {
    return gemv_f64(M, nr, nc, ldm, x, y, alpha, beta, st);
}
template <>
int gemv_t<float>(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
                  double alpha, double beta, hipStream_t st) {
    return gemv_f32(M, nr, nc, ldm, x, y, alpha, beta, st);
}

constexpr int NB = 256;         // outer panel width
constexpr int PW = 64;          // sub-panel width (one lane per column)
constexpr int kPanelThreads = 256;
constexpr int kSlabStride = kPanelThreads + 1;  // odd => conflict-free both ways
constexpr int kShards = 8;      // arrival-counter shards (128 B apart)
constexpr int kMaxPanelGroups = 256;
constexpr unsigned kSpinLimit = 1u << 22;

// ---- agent-scope (sc1) accessors for inter-workgroup traffic --------------------------
#define SSA_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                       static_cast<unsigned long long>(__double_as_longlong(v)), SSA_AGENT);
}
__device__ __forceinline__ void st_agent(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int *>(p), __float_as_uint(v), SSA_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
        reinterpret_cast<const unsigned long long *>(p), SSA_AGENT)));
}
__device__ __forceinline__ float ld_agent(const float *p) {
    return __uint_as_float(
        __hip_atomic_load(reinterpret_cast<const unsigned int *>(p), SSA_AGENT));
}

struct Cand {  // pivot candidate: larger |value| wins, ties -> smaller current position
    double absval;
    int pos;   // current (LAPACK) position of the row, panel-relative
    int aux;   // local row index (stage 1) or workgroup id (stage 2)
    int orig;  // original panel-relative row index
};
__device__ __forceinline__ bool better(const Cand &a, const Cand &b) {  // a beats b
    return (a.absval > b.absval) || (a.absval == b.absval && a.pos < b.pos);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.absval = __shfl_xor(c.absval, off, 64);
        o.pos = __shfl_xor(c.pos, off, 64);
        o.aux = __shfl_xor(c.aux, off, 64);
        o.orig = __shfl_xor(c.orig, off, 64);
        if (better(o, c)) c = o;
    }
    return c;
}

template <typename T>
struct PanelArgs {
    T *A;
    int64_t lda;
    int64_t j0;        // first row == first column of the sub-panel
    int m;             // rows in the sub-panel (n - j0)
    int jb;            // columns (<= PW)
    int rpw;           // rows per workgroup (<= kPanelThreads)
    int32_t *ipiv;     // absolute, LAPACK style
    int32_t *info;     // device scalar
    unsigned int *cnt;            // [kShards * 32]
    unsigned long long *hdr;      // [2][G][2]: {absval bits, pos << 32 | orig}
    T *rows;                      // [2][G][PW]
    unsigned int *timeout;        // set when a bounded spin gives up
    // speculation protocol (lu_panel_spec3_kernel, lu_spec3.hpp, runs first):
    const T *backup;              // [m][PW] original sub-panel, or nullptr: read A
    const int *spec_flag;         // != 0: speculation failed, this kernel must redo the sub-panel
    const int *zero_col;          // 1-based column of a zero pivot seen by the speculative pass
    const T *top;                 // [64][64] factored diagonal block parked by the speculative pass
};

template <typename T>
__global__ __launch_bounds__(kPanelThreads) void lu_panel_kernel(PanelArgs<T> a) {
    if (a.spec_flag != nullptr && *a.spec_flag == 0) {
        // The speculative pass was exact (no row below the diagonal block ever beat its pivot):
        // nothing to redo.  The factored diagonal block moves from its parking place into A (the
        // speculative kernel must not overwrite what its late workgroups still have to read), and
        // the LAPACK info of an exactly-zero pivot column is finalised.
        if (blockIdx.x == 0) {
            const int nt = (a.m < PW) ? a.m : PW;
            T *Ap = a.A + a.j0 * a.lda + a.j0;
            for (int e = threadIdx.x; e < nt * PW; e += kPanelThreads) {
                const int r = e / PW, c = e % PW;
                if (c < a.jb) Ap[static_cast<int64_t>(r) * a.lda + c] = a.top[r * 64 + c];
            }
            if (threadIdx.x == 0 && *a.zero_col != 0 && *a.info == 0)
                *a.info = static_cast<int32_t>(a.j0 + *a.zero_col);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *slab = reinterpret_cast<T *>(smem_raw);                       // [PW][kSlabStride]
    T *piv = slab + PW * kSlabStride;                                // [PW]
    double *red_abs = reinterpret_cast<double *>(piv + PW);          // [2][4]
    int *red_i = reinterpret_cast<int *>(red_abs + 8);               // [2][4][3]
    int *row_at_top = red_i + 24;                                    // [PW]
    int *piv_pos = row_at_top + PW;                                  // [PW]  ipiv, relative
    int *bk = piv_pos + PW;                                          // [4] o, q, rJ, flag
    int *fpos = bk + 4;                                              // [kPanelThreads]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, G = gridDim.x;
    const int row_base = g * a.rpw;
    const int myrows = max(0, min(a.rpw, a.m - row_base));
    const int jb = a.jb;
    T *Ap = a.A + a.j0 * a.lda + a.j0;  // panel origin

    // ---- load the slab: wave <-> row, lane <-> column (coalesced 512 B row segments) ----
    for (int r = wave; r < myrows; r += kPanelThreads / kWave) {
        if (lane < jb)
            slab[lane * kSlabStride + r] =
                a.backup ? a.backup[static_cast<int64_t>(row_base + r) * PW + lane]
                         : Ap[static_cast<int64_t>(row_base + r) * a.lda + lane];
    }
    if (tid < PW) row_at_top[tid] = tid;
    const int orig = row_base + tid;
    int curpos = orig;
    bool done = (tid >= myrows);
    int finalpos = -1;
    __syncthreads();

    for (int J = 0; J < jb; ++J) {
        const int buf = J & 1;
        // (1) best candidate of this workgroup
        Cand c;
        c.absval = done ? -1.0 : fabs(static_cast<double>(slab[J * kSlabStride + tid]));
        c.pos = curpos;
        c.aux = tid;
        c.orig = orig;
        c = wave_best(c);
        if (lane == 0) {
            red_abs[wave] = c.absval;
            red_i[wave * 3 + 0] = c.pos;
            red_i[wave * 3 + 1] = c.aux;
            red_i[wave * 3 + 2] = c.orig;
        }
        __syncthreads();  // #1
        Cand b;
        b.absval = red_abs[0]; b.pos = red_i[0]; b.aux = red_i[1]; b.orig = red_i[2];
#pragma unroll
        for (int v = 1; v < kPanelThreads / kWave; ++v) {
            Cand o;
            o.absval = red_abs[v]; o.pos = red_i[v * 3]; o.aux = red_i[v * 3 + 1]; o.orig = red_i[v * 3 + 2];
            if (better(o, b)) b = o;
        }
        // (2) publish candidate row + header, arrive, wait for everybody (wave 0 only)
        if (wave == 0) {
            T *slot = a.rows + (static_cast<size_t>(buf) * G + g) * PW;
            if (lane < jb) {
                const T v = (b.absval >= 0.0) ? slab[lane * kSlabStride + b.aux] : T(0);
                st_agent(slot + lane, v);
            }
            if (lane == 0) {
                unsigned long long *h = a.hdr + (static_cast<size_t>(buf) * G + g) * 2;
                __hip_atomic_store(h, static_cast<unsigned long long>(__double_as_longlong(b.absval)), SSA_AGENT);
                __hip_atomic_store(h + 1,
                                   (static_cast<unsigned long long>(static_cast<unsigned int>(b.pos)) << 32) |
                                       static_cast<unsigned int>(b.orig),
                                   SSA_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(a.cnt + (g % kShards) * 32, 1u, SSA_AGENT);
            // shard s receives one arrival per step from every workgroup with id % kShards == s
            const unsigned per_step = (lane < kShards && lane < G) ? (G - lane + kShards - 1) / kShards : 0u;
            const unsigned target = per_step * static_cast<unsigned>(J + 1);
            bool ok = (per_step == 0u);
            unsigned spins = 0;
            int gave_up = 0;
            while (true) {
                if (!ok) ok = __hip_atomic_load(a.cnt + lane * 32, SSA_AGENT) >= target;
                if (__all(ok)) break;
                if (++spins > kSpinLimit) { gave_up = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) {
                bk[3] = gave_up;
                if (gave_up) __hip_atomic_store(a.timeout, 1u, SSA_AGENT);
            }
        }
        __syncthreads();  // #2
        if (bk[3]) {      // a peer never arrived: give up loudly (info = -1), never hang
            // EVERY workgroup that gives up reports it (a late peer may let workgroup 0 finish the last
            // column while this one has not written its slab back: the factors are then not the LU)
            if (tid == 0) __hip_atomic_store(a.info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // (3) global winner: thread t inspects workgroup t's header
        Cand w;
        w.absval = -2.0; w.pos = 0x7fffffff; w.aux = 0; w.orig = 0;
        if (tid < G) {
            const unsigned long long *h = a.hdr + (static_cast<size_t>(buf) * G + tid) * 2;
            const unsigned long long v0 = __hip_atomic_load(h, SSA_AGENT);
            const unsigned long long v1 = __hip_atomic_load(h + 1, SSA_AGENT);
            w.absval = __longlong_as_double(static_cast<long long>(v0));
            w.pos = static_cast<int>(v1 >> 32);
            w.orig = static_cast<int>(v1 & 0xffffffffu);
            w.aux = tid;
        }
        w = wave_best(w);
        if (lane == 0) {
            red_abs[4 + wave] = w.absval;
            red_i[12 + wave * 3 + 0] = w.pos;
            red_i[12 + wave * 3 + 1] = w.aux;
            red_i[12 + wave * 3 + 2] = w.orig;
        }
        __syncthreads();  // #3
        Cand p;
        p.absval = red_abs[4]; p.pos = red_i[12]; p.aux = red_i[13]; p.orig = red_i[14];
#pragma unroll
        for (int v = 1; v < kPanelThreads / kWave; ++v) {
            Cand o;
            o.absval = red_abs[4 + v]; o.pos = red_i[12 + v * 3]; o.aux = red_i[12 + v * 3 + 1]; o.orig = red_i[12 + v * 3 + 2];
            if (better(o, p)) p = o;
        }
        // (4) fetch the pivot row; replay the LAPACK interchange  row J <-> row p.pos
        if (wave == 0) {
            if (lane < jb) piv[lane] = ld_agent(a.rows + (static_cast<size_t>(buf) * G + p.aux) * PW + lane);
            if (lane == 0) {
                const int q = p.pos;
                const int rJ = row_at_top[J];  // the row that currently sits at position J
                if (q < PW) row_at_top[q] = rJ;
                row_at_top[J] = p.orig;
                piv_pos[J] = q;
                bk[0] = p.orig; bk[1] = q; bk[2] = rJ;
            }
        }
        __syncthreads();  // #4
        const int o_row = bk[0], q_pos = bk[1], rJ = bk[2];
        const bool was_active = !done;
        if (orig == o_row) {
            done = true;
            finalpos = J;
        } else if (orig == rJ) {
            curpos = q_pos;
        }
        // (5) eliminate: l = a/pivot; row -= l * pivot_row   (?getf2: exact zero pivot =>
        //     record info, skip the scaling; the column below is then all zeros)
        if (p.absval == 0.0) {
            if (g == 0 && tid == 0 && *a.info == 0) *a.info = static_cast<int32_t>(a.j0 + J + 1);
        } else if (was_active && !done) {
            const T pv = piv[J];
            const T l = slab[J * kSlabStride + tid] / pv;
            slab[J * kSlabStride + tid] = l;
            for (int cidx = J + 1; cidx < jb; ++cidx)
                slab[cidx * kSlabStride + tid] -= l * piv[cidx];
        }
    }

    // ---- write back every row at its final position; workgroup 0 emits ipiv ------------
    fpos[tid] = done && finalpos >= 0 ? finalpos : curpos;
    __syncthreads();
    for (int r = wave; r < myrows; r += kPanelThreads / kWave) {
        if (lane < jb) Ap[static_cast<int64_t>(fpos[r]) * a.lda + lane] = slab[lane * kSlabStride + r];
    }
    if (g == 0 && tid < jb) a.ipiv[a.j0 + tid] = static_cast<int32_t>(a.j0 + piv_pos[tid]);
}

template <typename T>
constexpr size_t panel_smem_bytes() {
    return sizeof(T) * (PW * kSlabStride + PW) + 8 * sizeof(double) +
           sizeof(int) * (24 + PW + PW + 4 + kPanelThreads) + 64;
}

}  // namespace ssa
#include "lu_spec.hpp"
#include "lu_spec3.hpp"
namespace ssa {

// ---- row interchanges on column ranges [c0a,c1a) and [c0b,c1b) -------------------------
template <typename T>
__global__ void laswp_kernel(T *__restrict__ A, int64_t lda, int64_t c0a, int64_t c1a,
                             int64_t c0b, int64_t c1b, const int32_t *__restrict__ ipiv,
                             int64_t k0, int64_t k1) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t la = c1a - c0a;
    int64_t c;
    if (t < la) {
        c = c0a + t;
    } else if (t - la < c1b - c0b) {
        c = c0b + (t - la);
    } else {
        return;
    }
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t p = ipiv[k];
        if (p != k) {
            const T x = A[k * lda + c], y = A[p * lda + c];
            A[k * lda + c] = y;
            A[p * lda + c] = x;
        }
    }
}

// ---- triangular solve with a (kb x kb) block, kb <= 256, in LDS -----------------------
// LOWER-unit:  L X = B;   UPPER (non-unit): U X = B.   B is kb x N, solved in place in
// strips of 32 columns (one workgroup each); blockIdx.y batches independent problems.
// IDENT: B is not read; the right-hand side is the identity (=> X = inverse of the block).
constexpr int TS = 32;   // strip width
constexpr int TBLK = 64; // LDS block of the triangular matrix

template <typename T, bool UPPER, bool IDENT, bool UNITDIAG = true>
__global__ __launch_bounds__(256) void trsm_block_kernel(const T *__restrict__ Tri, int64_t ldt,
                                                         int64_t tri_batch_stride, T *__restrict__ B,
                                                         int64_t ldb, int64_t b_batch_stride,
                                                         int kb, int64_t N) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *Bs = reinterpret_cast<T *>(smem_raw);         // [NB][TS + 1]
    T *Ls = Bs + NB * (TS + 1);                      // [TBLK][TBLK + 1]
    constexpr int SBs = TS + 1, SLs = TBLK + 1;
    const int tid = threadIdx.x;
    const int c = tid & (TS - 1), rg = tid / TS;     // 8 row groups
    const int64_t n0 = static_cast<int64_t>(blockIdx.x) * TS;
    Tri += static_cast<int64_t>(blockIdx.y) * tri_batch_stride;
    B += static_cast<int64_t>(blockIdx.y) * b_batch_stride;
    const bool col_ok = (n0 + c < N);

    const int nblk = (kb + TBLK - 1) / TBLK;
    for (int r = rg; r < nblk * TBLK; r += 8) {  // rows >= kb are zero padding
        T v = T(0);
        if (r < kb) {
            if (IDENT) v = (n0 + c == r) ? T(1) : T(0);
            else if (col_ok) v = B[static_cast<int64_t>(r) * ldb + n0 + c];
        }
        Bs[r * SBs + c] = v;
    }
    auto load_tri = [&](int rb, int cb) {  // Ls <- Tri[rb block, cb block], zero padded
        for (int e = tid; e < TBLK * TBLK; e += 256) {
            const int i = e / TBLK, k = e % TBLK;
            const int gi = rb * TBLK + i, gk = cb * TBLK + k;
            Ls[i * SLs + k] = (gi < kb && gk < kb) ? Tri[static_cast<int64_t>(gi) * ldt + gk] : T(0);
        }
    };
    for (int step = 0; step < nblk; ++step) {
        const int rb = UPPER ? nblk - 1 - step : step;
        // off-diagonal blocks already solved: B_rb -= T[rb, cb] * X_cb
        for (int s2 = 0; s2 < step; ++s2) {
            const int cb = UPPER ? nblk - 1 - s2 : s2;
            __syncthreads();
            load_tri(rb, cb);
            __syncthreads();
            T acc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = T(0);
            for (int k = 0; k < TBLK; ++k) {
                const T bv = Bs[min(cb * TBLK + k, NB - 1) * SBs + c];
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] += Ls[(rg + 8 * t) * SLs + k] * bv;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int gi = rb * TBLK + rg + 8 * t;
                if (gi < kb) Bs[gi * SBs + c] -= acc[t];
            }
        }
        __syncthreads();
        load_tri(rb, rb);
        __syncthreads();
        const int rows_here = min(TBLK, kb - rb * TBLK);
        for (int kk = 0; kk < rows_here; ++kk) {
            const int k = UPPER ? rows_here - 1 - kk : kk;
            T xk = Bs[(rb * TBLK + k) * SBs + c];
            if (UPPER || !UNITDIAG) xk = xk / Ls[k * SLs + k];
            __syncthreads();
            if ((UPPER || !UNITDIAG) && rg == 0) Bs[(rb * TBLK + k) * SBs + c] = xk;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = rg + 8 * t;
                const bool below = UPPER ? (i < k) : (i > k && i < rows_here);
                if (below) Bs[(rb * TBLK + i) * SBs + c] -= Ls[i * SLs + k] * xk;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int r = rg; r < kb; r += 8) {
        if (col_ok) B[static_cast<int64_t>(r) * ldb + n0 + c] = Bs[r * SBs + c];
    }
}

template <typename T>
constexpr size_t trsm_smem_bytes() {
    return sizeof(T) * (NB * (TS + 1) + TBLK * (TBLK + 1));
}

template <typename T, bool UPPER, bool IDENT, bool UNITDIAG = true>
int launch_trsm(const T *Tri, int64_t ldt, int64_t tri_bs, T *B, int64_t ldb, int64_t b_bs, int kb,
                int64_t N, int batch, hipStream_t st) {
    if (kb <= 0 || N <= 0 || batch <= 0) return SSA_OK;
    static DeviceFlags lds_flags;
    if (raise_dynamic_lds(lds_flags, {{reinterpret_cast<const void *>(&trsm_block_kernel<T, UPPER, IDENT, UNITDIAG>),
                                       trsm_smem_bytes<T>()}}) != SSA_OK)
        return SSA_ERR_HIP;
    const dim3 grid(static_cast<unsigned>(ceil_div(N, TS)), static_cast<unsigned>(batch));
    hipLaunchKernelGGL((trsm_block_kernel<T, UPPER, IDENT, UNITDIAG>), grid, dim3(256), trsm_smem_bytes<T>(), st,
                       Tri, ldt, tri_bs, B, ldb, b_bs, kb, N);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

// ---- workspace layout ------------------------------------------------------------------
struct PanelScratchBytes {
    static constexpr size_t cnt = kShards * 32 * sizeof(unsigned int) + 128;  // + timeout word
    static constexpr size_t hdr = 2 * kMaxPanelGroups * 2 * sizeof(unsigned long long);
    template <typename T>
    static constexpr size_t rows() { return 2 * static_cast<size_t>(kMaxPanelGroups) * PW * sizeof(T); }
};

// ---- solve-phase data: inverses of the LSB x LSB diagonal blocks of L (unit lower) and of U ----
// Same construction as the Cholesky route (chol.hip): 256 x 256 leaves by the LDS substitution kernel,
// then one recursion level of all full blocks = two batched GEMMs per factor with triangular K ranges
//     L:  tmp = L21 inv11,   inv21 = -inv22 tmp        U:  tmp = U12 inv22,   inv12 = -inv11 tmp
// so that ssa_lu_solve is a chain of 2 row-major GEMVs per block and factor (HBM bound).
constexpr int64_t LSB = 4096;

struct LuAux {
    int64_t nblk, nfull, invL, invU, tmp, total;  // offsets in elements
};
inline LuAux lu_aux_layout(int64_t n) {
    LuAux a;
    a.nblk = ceil_div(n, LSB);
    a.nfull = n / LSB;
    a.invL = 0;
    a.invU = a.nblk * LSB * LSB;
    a.tmp = 2 * a.nblk * LSB * LSB;
    a.total = a.tmp + a.nblk * (LSB * LSB / 4);
    return a;
}

// inverse of the diagonal block [r0, r0 + sz) of L (UPPER = false) or U (true) from its inverted
// 256-leaves, recursively (the last, partial LSB block; any sz)
template <typename T, bool UPPER>
int lu_block_inverse(const T *A, int64_t lda, int64_t r0, int64_t sz, T *inv, int64_t ldi, T *tmp, hipStream_t st) {
    if (sz <= NB) return SSA_OK;
    int64_t h = NB;
    while (2 * h < sz) h *= 2;
    const int64_t r = sz - h;
    int rc = lu_block_inverse<T, UPPER>(A, lda, r0, h, inv, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    rc = lu_block_inverse<T, UPPER>(A, lda, r0 + h, r, inv + h * ldi + h, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    if (!UPPER) {
        rc = gemm_t<T>(r, h, h, 1.0, A + (r0 + h) * lda + r0, lda, inv, ldi, 0.0, tmp, h, st);
        if (rc != SSA_OK) return rc;
        return gemm_t<T>(r, h, r, -1.0, inv + h * ldi + h, ldi, tmp, h, 0.0, inv + h * ldi, ldi, st);
    }
    rc = gemm_t<T>(h, r, r, 1.0, A + r0 * lda + r0 + h, lda, inv + h * ldi + h, ldi, 0.0, tmp, r, st);
    if (rc != SSA_OK) return rc;
    return gemm_t<T>(h, r, h, -1.0, inv, ldi, tmp, r, 0.0, inv + h, ldi, st);
}

// The block inverses of the full LSB blocks [b0, b1) of L and U from their inverted 256-leaves, level by level
// (pairs of h-blocks -> 2h-blocks), all pairs of all blocks of a level in one batched launch.
template <typename T>
int lu_finish_full_blocks(const T *A, int64_t n, int64_t lda, T *aux, int64_t b0, int64_t b1, hipStream_t st) {
    const LuAux al = lu_aux_layout(n);
    if (b0 < 0 || b1 > al.nfull || b0 >= b1) return SSA_OK;
    const int nb = static_cast<int>(b1 - b0);
    const T *A0 = A + b0 * LSB * (lda + 1);
    T *invL = aux + al.invL + b0 * LSB * LSB, *invU = aux + al.invU + b0 * LSB * LSB;
    T *tmp = aux + al.tmp + b0 * (LSB * LSB / 4);
    int rc;
    for (int64_t h = NB; h < LSB; h *= 2) {
        const int ppb = static_cast<int>(LSB / (2 * h));
        const int64_t pair_l = 2 * h * (lda + 1), blk_l = LSB * (lda + 1);
        const int64_t pair_i = 2 * h * (LSB + 1), blk_i = LSB * LSB;
        const int64_t pair_t = h * h, blk_t = LSB * LSB / 4;
        const int64_t s1[6] = {pair_l, blk_l, pair_i, blk_i, pair_t, blk_t};
        const int64_t s2[6] = {pair_i, blk_i, pair_t, blk_t, pair_i, blk_i};
        // L: tmp = L21 inv11 ; inv21 = -inv22 tmp
        rc = gemm_batched_x(h, h, h, 1.0, A0 + h * lda, lda, invL, LSB, 0.0, tmp, h, ppb, nb, s1, 1, st);
        if (rc != SSA_OK) return rc;
        rc = gemm_batched_x(h, h, h, -1.0, invL + h * (LSB + 1), LSB, tmp, h, 0.0, invL + h * LSB, LSB, ppb, nb, s2, 2,
                            st);
        if (rc != SSA_OK) return rc;
        // U: tmp = U12 inv22 ; inv12 = -inv11 tmp
        const int64_t s3[6] = {pair_l, blk_l, pair_i, blk_i, pair_t, blk_t};
        rc = gemm_batched_x(h, h, h, 1.0, A0 + h, lda, invU + h * (LSB + 1), LSB, 0.0, tmp, h, ppb, nb, s3, 3, st);
        if (rc != SSA_OK) return rc;
        rc = gemm_batched_x(h, h, h, -1.0, invU, LSB, tmp, h, 0.0, invU + h, LSB, ppb, nb, s2, 4, st);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

// first_block: the full blocks below it were finished earlier (lu_finish_full_blocks, the no-interchange route
// finishes a block as soon as its last panel is factored)
template <typename T>
int lu_build_solve_blocks(const T *A, int64_t n, int64_t lda, T *aux, hipStream_t st, bool leaves_done = false,
                          int64_t first_block = 0) {
    const LuAux al = lu_aux_layout(n);
    T *invL = aux + al.invL, *invU = aux + al.invU, *tmp = aux + al.tmp;
    int rc;
    // leaves_done: the 256-leaves were written by the diagonal-block kernels of the no-interchange route
    // (into a zeroed aux), as the Cholesky route does
    if (!leaves_done && hipMemsetAsync(aux, 0, static_cast<size_t>(al.tmp) * sizeof(T), st) != hipSuccess)
        return SSA_ERR_HIP;
    // leaves, batched by their position s inside the LSB block
    for (int64_t s = 0; !leaves_done && s < LSB / NB; ++s) {
        const int64_t first = s * NB;
        if (first + NB > n) break;
        const int count = static_cast<int>((n - first - NB) / LSB + 1);
        rc = launch_trsm<T, false, true>(A + first * (lda + 1), lda, LSB * (lda + 1), invL + first * (LSB + 1), LSB,
                                         LSB * LSB, NB, NB, count, st);
        if (rc != SSA_OK) return rc;
        rc = launch_trsm<T, true, true>(A + first * (lda + 1), lda, LSB * (lda + 1), invU + first * (LSB + 1), LSB,
                                        LSB * LSB, NB, NB, count, st);
        if (rc != SSA_OK) return rc;
    }
    if (!leaves_done && n % NB != 0) {  // the last, partial leaf
        const int64_t r0 = n / NB * NB, Jb = r0 / LSB, off = r0 - Jb * LSB;
        const int kb = static_cast<int>(n - r0);
        rc = launch_trsm<T, false, true>(A + r0 * (lda + 1), lda, 0, invL + Jb * LSB * LSB + off * (LSB + 1), LSB, 0,
                                         kb, kb, 1, st);
        if (rc != SSA_OK) return rc;
        rc = launch_trsm<T, true, true>(A + r0 * (lda + 1), lda, 0, invU + Jb * LSB * LSB + off * (LSB + 1), LSB, 0,
                                        kb, kb, 1, st);
        if (rc != SSA_OK) return rc;
    }
    rc = lu_finish_full_blocks<T>(A, n, lda, aux, first_block, al.nfull, st);
    if (rc != SSA_OK) return rc;
    if (n % LSB != 0) {
        const int64_t r0 = al.nfull * LSB;
        T *t2 = tmp + al.nfull * (LSB * LSB / 4);
        rc = lu_block_inverse<T, false>(A, lda, r0, n - r0, invL + al.nfull * LSB * LSB, LSB, t2, st);
        if (rc != SSA_OK) return rc;
        rc = lu_block_inverse<T, true>(A, lda, r0, n - r0, invU + al.nfull * LSB * LSB, LSB, t2, st);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

template <typename T>
int getrf(T *A, int64_t n, int64_t lda, int32_t *ipiv, int32_t *info, T *aux, void *workspace,
          hipStream_t st) {
    static int cus_of_device[kMaxDevices] = {};  // co-residency bound for the cooperative panel kernel
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    if (cus_of_device[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return SSA_ERR_HIP;
        cus_of_device[dev] = v;
    }
    const int num_cus = cus_of_device[dev];
    const int max_groups = min(num_cus, kMaxPanelGroups);
    if (ceil_div(n, kPanelThreads) > max_groups) return SSA_ERR_UNSUPPORTED_SIZE;

    Carver cv(workspace);
    unsigned int *cnt = cv.take<unsigned int>(kShards * 32 + 32);
    unsigned int *timeout = cnt + kShards * 32;
    unsigned long long *hdr = cv.take<unsigned long long>(2 * kMaxPanelGroups * 2);
    T *rows = cv.take<T>(2 * static_cast<size_t>(kMaxPanelGroups) * PW);
    const int64_t nsub = ceil_div(n, PW) + ceil_div(n, NB);  // upper bound on sub-panels
    int *flags = cv.take<int>(2 * nsub);                     // [nsub] spec flags, [nsub] zero cols
    T *backup = cv.take<T>(static_cast<size_t>(n) * PW);
    T *dinv = cv.take<T>(static_cast<size_t>(NB / PW) * 64 * 64);  // inv(L11) of the sub-panels
    T *top = cv.take<T>(64 * 64);                                   // factored diagonal block in transit

    static DeviceFlags lds_flags;
    if (raise_dynamic_lds(lds_flags,
                          {{reinterpret_cast<const void *>(&lu_panel_kernel<T>), panel_smem_bytes<T>()},
                           {reinterpret_cast<const void *>(&lu_panel_spec3_kernel<T>), spec3_smem_bytes<T>()},
                           {reinterpret_cast<const void *>(&trsm_lower_inv_kernel<T>), trsm_inv_smem_bytes<T>()}}) !=
        SSA_OK)
        return SSA_ERR_HIP;
    if (hipMemsetAsync(info, 0, sizeof(int32_t), st) != hipSuccess) return SSA_ERR_HIP;
    if (hipMemsetAsync(flags, 0, 2 * nsub * sizeof(int), st) != hipSuccess) return SSA_ERR_HIP;

    int rc;
    int64_t sub = 0;
    for (int64_t k0 = 0; k0 < n; k0 += NB) {
        const int64_t kb = (n - k0 < NB) ? n - k0 : NB;
        for (int64_t j0 = k0; j0 < k0 + kb; j0 += PW, ++sub) {
            const int64_t jb = (k0 + kb - j0 < PW) ? k0 + kb - j0 : PW;
            const int64_t m = n - j0;
            // (1) speculative pass: pivots assumed inside the diagonal block, verified on the fly
            const int64_t sidx = (j0 - k0) / PW;
            Spec3Args<T> sa;
            sa.A = A; sa.lda = lda; sa.j0 = j0; sa.m = static_cast<int>(m); sa.jb = static_cast<int>(jb);
            sa.ipiv = ipiv; sa.backup = backup; sa.spec_flag = flags + sub; sa.zero_col = flags + nsub + sub;
            sa.cnt = cnt; sa.dinv = dinv + sidx * 64 * 64; sa.top = top;
            hipLaunchKernelGGL((lu_panel_spec3_kernel<T>), dim3(ceil_div(m, kSpec3Rows) + 1), dim3(256),
                               spec3_smem_bytes<T>(), st, sa);
            SSA_RETURN_IF_LAUNCH_FAILED();
            // (2) exact cooperative pass: returns immediately unless the speculation failed
            const int G = static_cast<int>(ceil_div(m, kPanelThreads));
            const int rpw = static_cast<int>(ceil_div(m, G));
            PanelArgs<T> pa;
            pa.A = A; pa.lda = lda; pa.j0 = j0; pa.m = static_cast<int>(m);
            pa.jb = static_cast<int>(jb); pa.rpw = rpw; pa.ipiv = ipiv; pa.info = info;
            pa.cnt = cnt; pa.hdr = hdr; pa.rows = rows; pa.timeout = timeout;
            pa.backup = backup; pa.spec_flag = flags + sub; pa.zero_col = flags + nsub + sub; pa.top = top;
            hipLaunchKernelGGL((lu_panel_kernel<T>), dim3(G), dim3(kPanelThreads),
                               panel_smem_bytes<T>(), st, pa);
            SSA_RETURN_IF_LAUNCH_FAILED();
            // (3) inverse of the unit-lower diagonal block for the block trsm: already written by
            //     the speculative kernel; recomputed only if the cooperative kernel had to run
            hipLaunchKernelGGL((trtri_lower64_kernel<T>), dim3(1), dim3(256), 0, st, A + j0 * lda + j0, lda,
                               static_cast<int64_t>(0), static_cast<int>(jb), 1, dinv + sidx * 64 * 64,
                               static_cast<const int *>(flags + sub));
            SSA_RETURN_IF_LAUNCH_FAILED();
            // interchanges for the other columns of the outer panel
            const int64_t wa = j0 - k0, wb = (k0 + kb) - (j0 + jb);
            if (wa + wb > 0) {
                hipLaunchKernelGGL((laswp_kernel<T>), dim3(ceil_div(wa + wb, 256)), dim3(256), 0, st, A,
                                   lda, k0, j0, j0 + jb, k0 + kb, ipiv, j0, j0 + jb);
                SSA_RETURN_IF_LAUNCH_FAILED();
            }
            if (wb > 0) {
                T *L11 = A + j0 * lda + j0;
                T *U12 = A + j0 * lda + j0 + jb;
                hipLaunchKernelGGL((trsm_lower_inv_kernel<T>), dim3(ceil_div(wb, 32)), dim3(256),
                                   trsm_inv_smem_bytes<T>(), st, L11, lda, dinv + sidx * 64 * 64, U12, lda,
                                   static_cast<int>(jb), wb);
                SSA_RETURN_IF_LAUNCH_FAILED();
                const int64_t mm = n - j0 - jb;
                if (mm > 0) {
                    rc = gemm_t<T>(mm, wb, jb, -1.0, A + (j0 + jb) * lda + j0, lda, U12, lda, 1.0,
                                   A + (j0 + jb) * lda + j0 + jb, lda, st);
                    if (rc != SSA_OK) return rc;
                }
            }
        }
        // interchanges left and right of the outer panel
        const int64_t right = n - k0 - kb;
        if (k0 + right > 0) {
            hipLaunchKernelGGL((laswp_kernel<T>), dim3(ceil_div(k0 + right, 256)), dim3(256), 0, st, A, lda,
                               static_cast<int64_t>(0), k0, k0 + kb, n, ipiv, k0, k0 + kb);
            SSA_RETURN_IF_LAUNCH_FAILED();
        }
        if (right > 0) {
            T *L11 = A + k0 * lda + k0;
            T *U12 = A + k0 * lda + k0 + kb;
            hipLaunchKernelGGL((trsm_lower_inv_kernel<T>), dim3(ceil_div(right, 32)), dim3(256),
                               trsm_inv_smem_bytes<T>(), st, L11, lda, dinv, U12, lda, static_cast<int>(kb),
                               right);
            SSA_RETURN_IF_LAUNCH_FAILED();
            rc = gemm_t<T>(right, right, kb, -1.0, A + (k0 + kb) * lda + k0, lda, U12, lda, 1.0,
                           A + (k0 + kb) * lda + k0 + kb, lda, st);
            if (rc != SSA_OK) return rc;
        }
    }
    return lu_build_solve_blocks<T>(A, n, lda, aux, st);
}

// L U X = B (B already row-permuted).  Block forward / backward substitution over LSB-row blocks
// with pre-inverted diagonal blocks: X_k = inv(L_kk) B_k ; B_{>k} -= L_{>k,k} X_k, then the mirror
// image with U.  Single right-hand side: two row-major GEMVs per block and factor; several: the same
// recurrence on the MFMA GEMM.  X (workspace, n x nrhs) and B ping-pong.
template <typename T>
int getrs(const T *LU, int64_t n, int64_t lda, const T *aux, T *B, int64_t nrhs, int64_t ldb,
          T *X, hipStream_t st) {
    const LuAux al = lu_aux_layout(n);
    const T *invL = aux + al.invL, *invU = aux + al.invU;
    const int64_t ldx = nrhs;
    T *partial = X + (n * nrhs + 1) / 2 * 2;  // split-K / skinny scratch, 16-byte aligned (ssa_lu_solve_workspace_bytes)
    const bool vec = (nrhs == 1 && ldb == 1);
    int rc;
    for (int64_t k = 0; k < al.nblk; ++k) {
        const int64_t r0 = k * LSB, kb = (n - r0 < LSB) ? n - r0 : LSB;
        const int64_t below = n - r0 - kb;
        const T *inv = invL + k * LSB * LSB;
        if (vec) {
            rc = trmv_x(inv, kb, kb, LSB, B + r0, X + r0, 1.0, 0.0, 1, st);
            if (rc == SSA_OK && below > 0)
                rc = gemv_t<T>(LU + (r0 + kb) * lda + r0, below, kb, lda, X + r0, B + r0 + kb, -1.0, 1.0, st);
        } else {
            rc = gemm_rhs_t(kb, nrhs, kb, 1.0, inv, LSB, B + r0 * ldb, ldb, 0.0, X + r0 * ldx, ldx, 1, partial, st);
            if (rc == SSA_OK && below > 0)
                rc = gemm_rhs_t(below, nrhs, kb, -1.0, LU + (r0 + kb) * lda + r0, lda, X + r0 * ldx, ldx, 1.0,
                                B + (r0 + kb) * ldb, ldb, 0, partial, st);
        }
        if (rc != SSA_OK) return rc;
    }
    for (int64_t k = al.nblk - 1; k >= 0; --k) {
        const int64_t r0 = k * LSB, kb = (n - r0 < LSB) ? n - r0 : LSB;
        const T *inv = invU + k * LSB * LSB;
        if (vec) {
            rc = trmv_x(inv, kb, kb, LSB, X + r0, B + r0, 1.0, 0.0, 2, st);
            if (rc == SSA_OK && r0 > 0) rc = gemv_t<T>(LU + r0, r0, kb, lda, B + r0, X, -1.0, 1.0, st);
        } else {
            rc = gemm_rhs_t(kb, nrhs, kb, 1.0, inv, LSB, X + r0 * ldx, ldx, 0.0, B + r0 * ldb, ldb, 2, partial, st);
            if (rc == SSA_OK && r0 > 0)
                rc = gemm_rhs_t(r0, nrhs, kb, -1.0, LU + r0, lda, B + r0 * ldb, ldb, 1.0, X, ldx, 0, partial, st);
        }
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

// Inverses of the NB x NB diagonal blocks of a NON-unit lower triangular factor (Cholesky),
// inv [ceil(n / NB)][NB][NB]; used by chol.hip.
template <typename T>
int trtri_lower_blocks(const T *A, int64_t lda, int64_t n, T *inv, hipStream_t st) {
    const int64_t full = n / NB;
    int rc = SSA_OK;
    if (full > 0)
        rc = launch_trsm<T, false, true, false>(A, lda, NB * (lda + 1), inv, NB, NB * NB, NB, NB,
                                                static_cast<int>(full), st);
    if (rc == SSA_OK && full * NB < n) {
        const int kb = static_cast<int>(n - full * NB);
        rc = launch_trsm<T, false, true, false>(A + full * NB * (lda + 1), lda, 0, inv + full * NB * NB, NB, 0,
                                                kb, kb, 1, st);
    }
    return rc;
}
// =========================================================================================
// No-interchange route: LU with look-ahead for matrices whose partial pivoting never swaps rows
// =========================================================================================
// The London systems are strictly diagonally dominant by rows: LAPACK's ?getrf
// returns ipiv == arange for them.  Without interchanges the factorization has the structure of the
// Cholesky route (chol.hip) and takes the same schedule:
//
//   per matrix and 256-column panel k (one high-priority chain stream, one update stream):
//   chain     block column  C[:, 0:256] -= L21p U12p[:, 0:256]  and block row  C[0:256, 256:] -= L21p[0:256] U12p[:, 256:]
//                           (pending panels p, K = 256 / 512)
//             diagonal block: L11 \ U11 in place, WL = inv(L11), WU = inv(U11)  (lu_diag256_kernel, lu_diag.hpp:
//                             one workgroup, 64-column Gaussian eliminations in registers + MFMA block products)
//             L21 = A21 WU and U12 = WL A12   (two in-place MFMA GEMMs each)
//   update    rest  C[256:, 256:] -= L21p U12p   (every other panel with K = 512 for large trailing matrices)
//
// The chain is ONE stream per matrix, one of the device's measured chain streams (chain_streams.hpp), as in the
// Cholesky schedule (chol.hip potrf_batch); rounds 2 and 3 ran it as two streams with hand-offs.
// i.e. panel k + 1 is factored while the rest of update k runs (look-ahead), the matrices of a batch hide
// each other's chains, and the trailing update is one NN GEMM.  What makes the result LAPACK's: partial
// pivoting keeps the diagonal at column J iff no multiplier below it exceeds 1 in magnitude (|a_rJ| <=
// |u_JJ|, ties go to the lowest index = the diagonal).  The multipliers are the entries of L: one pass over
// the finished factor checks them (np_check_kernel).  Any violation (or a singular pivot block) is reported
// as info = -2 and the caller factors the matrix again with ssa_lu_factor (full partial pivoting).
constexpr int kMaxLuLanes = 16;
struct LuLane {
    hipStream_t side = nullptr, upd = nullptr, fin = nullptr;   // side: a chain stream of the device (not owned)
    hipEvent_t ev_fork = nullptr, ev_panel = nullptr, ev_rest = nullptr, ev_upd = nullptr, ev_fin = nullptr;
};
struct LuLaneSet {
    LuLane lanes[kMaxLuLanes];
    std::mutex enqueue;
};
LuLaneSet g_lu_lane_sets[kMaxDevices];
std::mutex g_lu_lane_mutex;

inline int get_lu_lanes(int count, LuLaneSet **out) {
    std::lock_guard<std::mutex> lock(g_lu_lane_mutex);
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return SSA_ERR_HIP;
    LuLane *lanes = g_lu_lane_sets[dev].lanes;
    for (int i = 0; i < count; ++i) {
        if (lanes[i].upd != nullptr) continue;
        if (hipStreamCreateWithPriority(&lanes[i].upd, hipStreamNonBlocking, 0) != hipSuccess ||
            hipStreamCreateWithPriority(&lanes[i].fin, hipStreamNonBlocking, lo) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_fin, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_panel, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_rest, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_upd, hipEventDisableTiming) != hipSuccess)
            return SSA_ERR_HIP;
    }
    // the chains run on the device's chain streams, in the order measured against the stream of the trailing updates
    // (the first lane's update stream: two matrices share it, more have one each)
    hipStream_t chains[kMaxLuLanes];
    {
        std::lock_guard<std::mutex> enq(g_lu_lane_sets[dev].enqueue);
        const int rc = chain_streams_get(lanes[0].upd, kMaxLuLanes, chains);
        if (rc != SSA_OK) return rc;
    }
    for (int i = 0; i < kMaxLuLanes; ++i) lanes[i].side = chains[i];
    *out = &g_lu_lane_sets[dev];
    return SSA_OK;
}

int lu_shutdown() {
    std::lock_guard<std::mutex> lock(g_lu_lane_mutex);
    int rc = SSA_OK;
    for (int d = 0; d < kMaxDevices; ++d) {
        std::lock_guard<std::mutex> enq(g_lu_lane_sets[d].enqueue);
        for (LuLane &ln : g_lu_lane_sets[d].lanes) {
            if (ln.upd == nullptr) continue;
            if (hipStreamSynchronize(ln.upd) != hipSuccess || hipStreamSynchronize(ln.fin) != hipSuccess) rc = SSA_ERR_HIP;
            hipEvent_t evs[5] = {ln.ev_fork, ln.ev_panel, ln.ev_rest, ln.ev_upd, ln.ev_fin};
            for (hipEvent_t e : evs)
                if (e != nullptr && hipEventDestroy(e) != hipSuccess) rc = SSA_ERR_HIP;
            if (hipStreamDestroy(ln.upd) != hipSuccess || hipStreamDestroy(ln.fin) != hipSuccess) rc = SSA_ERR_HIP;
            ln = LuLane{};
        }
    }
    return rc;
}

template <typename T>
struct NpScratch {
    unsigned int *cnt, *timeout;
    unsigned long long *hdr;
    T *rows;
    int *flags;       // [nsub] speculation flags | [nsub] zero columns
    int64_t nsub;
    T *backup, *dinv, *top, *WL, *WU, *dscratch;
};
inline size_t np_workspace_bytes(int64_t np, size_t es) {
    const size_t nsub = static_cast<size_t>(np / PW);
    return PanelScratchBytes::cnt + PanelScratchBytes::hdr + 2 * static_cast<size_t>(kMaxPanelGroups) * PW * es +
           2 * nsub * sizeof(int) + static_cast<size_t>(NB) * PW * es + (NB / PW + 1) * 64 * 64 * es +
           2 * static_cast<size_t>(NB) * NB * es + 2 * (3 * 256 * 256 + 6 * 64 * 64) * sizeof(float) + 16 * 256;
}
template <typename T>
NpScratch<T> np_carve(void *workspace, int64_t np) {
    Carver cv(workspace);
    NpScratch<T> s;
    s.cnt = cv.take<unsigned int>(kShards * 32 + 32);
    s.timeout = s.cnt + kShards * 32;
    s.hdr = cv.take<unsigned long long>(2 * kMaxPanelGroups * 2);
    s.rows = cv.take<T>(2 * static_cast<size_t>(kMaxPanelGroups) * PW);
    s.nsub = np / PW;
    s.flags = cv.take<int>(2 * s.nsub);
    s.backup = cv.take<T>(static_cast<size_t>(NB) * PW);
    s.dinv = cv.take<T>(static_cast<size_t>(NB / PW) * 64 * 64);
    s.top = cv.take<T>(64 * 64);
    s.WL = cv.take<T>(static_cast<size_t>(NB) * NB);
    s.WU = cv.take<T>(static_cast<size_t>(NB) * NB);
    s.dscratch = cv.take<T>(luk::lu_diag_scratch_elems<T>());
    return s;
}

template <typename T>
struct NpJob {
    T *A;
    int64_t n, np, lda;   // true order, padded order (multiple of NB), leading dimension (>= np)
    int32_t *ipiv, *info;
    T *aux;
    void *workspace;
};

// rows n .. np-1 and columns n .. np-1 of the padded matrix: zero with a unit diagonal
template <typename T>
__global__ void np_pad_identity_kernel(T *A, int64_t lda, int64_t n, int64_t np) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c >= np) return;
    if (r >= n) A[r * lda + c] = (c == r) ? T(1) : T(0);
    else if (c >= n) A[r * lda + c] = T(0);
}

// any |L_ij| > 1 below the diagonal (a row that LAPACK's partial pivoting would have moved up), or a raised
// speculation flag -> info = -2 (unless a singular pivot was already reported)
template <typename T>
__global__ __launch_bounds__(256) void np_check_kernel(const T *__restrict__ A, int64_t lda, int64_t n,
                                                       const int *__restrict__ flags, int64_t nflags,
                                                       int32_t *__restrict__ info, int32_t *__restrict__ ipiv) {
    bool bad = false;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 16;
    if (threadIdx.x < 16 && r0 + threadIdx.x < n) ipiv[r0 + threadIdx.x] = static_cast<int32_t>(r0 + threadIdx.x);
    for (int rr = 0; rr < 16; ++rr) {
        const int64_t i = r0 + rr;
        if (i >= n) break;
        for (int64_t j = threadIdx.x; j < i; j += 256) bad = bad || (fabs(static_cast<double>(A[i * lda + j])) > 1.0);
    }
    if (blockIdx.x == 0)
        for (int64_t k = threadIdx.x; k < nflags; k += 256) bad = bad || (flags[k] != 0);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMin(info, -2);
}

// One outer panel of one matrix in pieces: the diagonal-block kernel (lu_diag.hpp: L11 \ U11 in place and the inverses
// WL, WU of the two factors), and L21 = A21 WU for a range of rows / U12 = WL A12 for a range of columns as in-place
// MFMA GEMMs.  WU is upper and WL lower triangular: the second 128 columns of L21 need all 256 columns of A21 (done
// first), the first 128 only the first 128 -- every workgroup reads and writes its own rows, which makes the in-place
// product safe; U12 likewise by rows of WL.
template <typename T>
struct NpLeaves {
    T *WL, *WU;
};
template <typename T>
NpLeaves<T> np_leaves(const NpJob<T> &J, int64_t k0) {
    // WL, WU land in the solve-phase buffer: they are the inverted 256-leaves of the LSB block inverses
    const LuAux al = lu_aux_layout(J.n);
    const int64_t leaf = (k0 / LSB) * LSB * LSB + (k0 % LSB) * (LSB + 1);
    return {J.aux + al.invL + leaf, J.aux + al.invU + leaf};
}
template <typename T>
int np_diag(const NpJob<T> &J, const NpScratch<T> &S, int64_t k0, hipStream_t s) {
    const NpLeaves<T> w = np_leaves(J, k0);
    hipLaunchKernelGGL((luk::lu_diag256_kernel<T>), dim3(1), dim3(256), sizeof(luk::LuSmem<double>), s,
                       J.A + k0 * (J.lda + 1), static_cast<int>(J.lda), w.WL, w.WU, static_cast<int>(LSB), S.dscratch,
                       J.info);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
// rows [r0, r1) of L21 of the panel at column k0 (r0 >= k0 + NB)
template <typename T>
int np_rows(const NpJob<T> &J, int64_t k0, int64_t r0, int64_t r1, hipStream_t s) {
    const int64_t M = r1 - r0, lda = J.lda;
    if (M <= 0) return SSA_OK;
    const T *WU = np_leaves(J, k0).WU;
    T *A21 = J.A + r0 * lda + k0;
    int rc = gemm_t<T>(M, 128, 256, 1.0, A21, lda, WU + 128, LSB, 0.0, A21 + 128, lda, s);
    if (rc != SSA_OK) return rc;
    return gemm_t<T>(M, 128, 128, 1.0, A21, lda, WU, LSB, 0.0, A21, lda, s);
}
// columns [c0, c1) of U12 of the panel at column k0 (c0 >= k0 + NB)
template <typename T>
int np_cols(const NpJob<T> &J, int64_t k0, int64_t c0, int64_t c1, hipStream_t s) {
    const int64_t M = c1 - c0, lda = J.lda;
    if (M <= 0) return SSA_OK;
    const T *WL = np_leaves(J, k0).WL;
    T *A12 = J.A + k0 * lda + c0;
    int rc = gemm_t<T>(128, M, 256, 1.0, WL + 128 * LSB, LSB, A12, lda, 0.0, A12 + 128 * lda, lda, s);
    if (rc != SSA_OK) return rc;
    return gemm_t<T>(128, M, 128, 1.0, WL, LSB, A12, lda, 0.0, A12, lda, s);
}
// panel k0 on the chain stream of a lane: diagonal-block kernel, then L21 and U12; ev_panel = all of it
template <typename T>
int np_panel(const NpJob<T> &J, const NpScratch<T> &S, int64_t k0, LuLane &ln) {
    const int64_t kend = k0 + NB;
    int rc = np_diag(J, S, k0, ln.side);
    if (rc != SSA_OK) return rc;
    rc = np_rows(J, k0, kend, J.np, ln.side);
    if (rc != SSA_OK) return rc;
    rc = np_cols(J, k0, kend, J.np, ln.side);
    if (rc != SSA_OK) return rc;
    if (hipEventRecord(ln.ev_panel, ln.side) != hipSuccess) return SSA_ERR_HIP;
    return SSA_OK;
}

template <typename T>
int getrf_np_batch(const NpJob<T> *jobs, int count, hipStream_t st) {
    if (count <= 0 || count > kMaxLuLanes) return SSA_ERR_INVALID_ARGUMENT;
    static DeviceFlags lds_flags;
    if (raise_dynamic_lds(lds_flags, {{reinterpret_cast<const void *>(&luk::lu_diag256_kernel<T>),
                                       sizeof(luk::LuSmem<double>)}}) != SSA_OK)
        return SSA_ERR_HIP;
    LuLaneSet *lane_set = nullptr;
    int rc = get_lu_lanes(count, &lane_set);
    if (rc != SSA_OK) return rc;
    std::lock_guard<std::mutex> enqueue_lock(lane_set->enqueue);
    LuLane *lanes = lane_set->lanes;
    // The trailing updates of two matrices alternate on ONE stream, as in the Cholesky schedule (two concurrent
    // full-chip GEMMs halve each other's L2 reach: config H 205.6 -> 200.5 ms); three or more matrices keep one update
    // stream each (a matrix' chain would wait behind the updates of all the others).
    const bool shared_updates = count < 3;
    NpScratch<T> scratch[kMaxLuLanes];
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        const NpJob<T> &J = jobs[i];
        LuLane &ln = lanes[i];
        hipStream_t us = shared_updates ? lanes[0].upd : ln.upd;   // where this matrix' trailing updates run
        if (J.np % NB != 0 || J.lda < J.np || J.lda > (int64_t(1) << 22)) return SSA_ERR_INVALID_ARGUMENT;  // 32-bit offsets in the block kernel
        if (J.np > nmax) nmax = J.np;
        scratch[i] = np_carve<T>(J.workspace, J.np);
        if (J.np > J.n) {
            hipLaunchKernelGGL((np_pad_identity_kernel<T>), dim3(static_cast<unsigned>(ceil_div(J.np, 256)),
                                                                 static_cast<unsigned>(J.np)),
                               dim3(256), 0, st, J.A, J.lda, J.n, J.np);
            SSA_RETURN_IF_LAUNCH_FAILED();
        }
        if (hipMemsetAsync(J.info, 0, sizeof(int32_t), st) != hipSuccess ||
            hipMemsetAsync(scratch[i].flags, 0, 2 * scratch[i].nsub * sizeof(int), st) != hipSuccess ||
            hipMemsetAsync(J.aux, 0, static_cast<size_t>(lu_aux_layout(J.n).tmp) * sizeof(T), st) != hipSuccess)
            return SSA_ERR_HIP;
        if (hipEventRecord(ln.ev_fork, st) != hipSuccess || hipStreamWaitEvent(ln.side, ln.ev_fork, 0) != hipSuccess ||
            hipStreamWaitEvent(us, ln.ev_fork, 0) != hipSuccess)
            return SSA_ERR_HIP;
        rc = np_panel(J, scratch[i], 0, ln);
        if (rc != SSA_OK) return rc;
    }
    constexpr int kDelayDepth = 2;            // as in the Cholesky schedule: two panels per trailing update
    constexpr int64_t kDelayMinCols = 8192;   // while the trailing matrix is large
    int64_t pending_from[kMaxLuLanes] = {};
    int64_t blocks_finished[kMaxLuLanes] = {};
    constexpr bool early_finish = true;   // block inverses of closed 4096-blocks during the chain-bound tail
    bool rest_recorded[kMaxLuLanes] = {};
    for (int64_t k0 = 0; k0 + NB < nmax; k0 += NB) {
        for (int i = 0; i < count; ++i) {
            const NpJob<T> &J = jobs[i];
            LuLane &ln = lanes[i];
        hipStream_t us = shared_updates ? lanes[0].upd : ln.upd;   // where this matrix' trailing updates run
            if (k0 + NB >= J.np) continue;
            const int64_t right = J.np - k0 - NB;   // order of the trailing matrix (a multiple of NB)
            const int64_t nw = NB;
            const int64_t pend0 = pending_from[i], kp = k0 + NB - pend0;
            const T *PL = J.A + (k0 + NB) * J.lda + pend0;   // pending L panels, rows of the trailing matrix
            const T *PU = J.A + pend0 * J.lda + (k0 + NB);   // pending U panels, columns of the trailing matrix
            T *C = J.A + (k0 + NB) * (J.lda + 1);
            if (hipStreamWaitEvent(us, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;   // panel k done
            // the chain: pending panels onto the next block column (diagonal block included) and onto the block row
            // right of the diagonal block, behind this matrix' last rest update (it wrote those entries); then the
            // panel
            if (rest_recorded[i] && hipStreamWaitEvent(ln.side, ln.ev_rest, 0) != hipSuccess) return SSA_ERR_HIP;
            rc = gemm_t<T>(right, nw, kp, -1.0, PL, J.lda, PU, J.lda, 1.0, C, J.lda, ln.side);
            if (rc != SSA_OK) return rc;
            if (right > nw) {
                rc = gemm_t<T>(nw, right - nw, kp, -1.0, PL, J.lda, PU + nw, J.lda, 1.0, C + nw, J.lda, ln.side);
                if (rc != SSA_OK) return rc;
            }
            rc = np_panel(J, scratch[i], k0 + NB, ln);
            if (rc != SSA_OK) return rc;
            const bool delay = kp < kDelayDepth * NB && right > kDelayMinCols &&
                               ((k0 + J.np) / NB) % kDelayDepth != kDelayDepth - 1;
            if (right > nw && !delay) {
                rc = gemm_t<T>(right - nw, right - nw, kp, -1.0, PL + nw * J.lda, J.lda, PU + nw, J.lda, 1.0,
                               C + nw * (J.lda + 1), J.lda, us);
                if (rc != SSA_OK) return rc;
                if (hipEventRecord(ln.ev_rest, us) != hipSuccess) return SSA_ERR_HIP;
                rest_recorded[i] = true;
            }
            if (!delay) pending_from[i] = k0 + NB;
            // The block inverses of the solve phase (lu_finish_full_blocks) of the LSB blocks whose panels are all
            // factored: on a low-priority stream of their own once the factorization is bound by its panel chain
            // (trailing matrix below kDelayMinCols: the chip is mostly idle then), instead of after the last panel,
            // where nothing hides them.  (Behind the trailing updates on us they delay the next panel by their
            // 3 ms per block; earlier, beside the large updates, they slow the chains: both measured)
            const int64_t blocks_closed = std::min((k0 + NB) / LSB, J.n / LSB);
            if (early_finish && right <= kDelayMinCols && blocks_closed > blocks_finished[i]) {
                if (hipStreamWaitEvent(ln.fin, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;
                rc = lu_finish_full_blocks<T>(J.A, J.n, J.lda, J.aux, blocks_finished[i], blocks_closed, ln.fin);
                if (rc != SSA_OK) return rc;
                blocks_finished[i] = blocks_closed;
            }
        }
    }
    // Every matrix finishes on its own update stream (behind its last panel and its last trailing update): the
    // pivot check and the solve-phase blocks of one matrix overlap the tail of the others
    for (int i = 0; i < count; ++i) {
        const NpJob<T> &J = jobs[i];
        LuLane &ln = lanes[i];
        hipStream_t us = shared_updates ? lanes[0].upd : ln.upd;   // where this matrix' trailing updates run
        if (hipStreamWaitEvent(us, ln.ev_panel, 0) != hipSuccess || hipEventRecord(ln.ev_fin, ln.fin) != hipSuccess ||
            hipStreamWaitEvent(us, ln.ev_fin, 0) != hipSuccess)
            return SSA_ERR_HIP;
        hipLaunchKernelGGL((np_check_kernel<T>), dim3(static_cast<unsigned>(ceil_div(J.n, 16))), dim3(256), 0, us,
                           J.A, J.lda, J.n, scratch[i].flags, scratch[i].nsub, J.info, J.ipiv);
        SSA_RETURN_IF_LAUNCH_FAILED();
        rc = lu_build_solve_blocks<T>(J.A, J.n, J.lda, J.aux, us, true, blocks_finished[i]);
        if (rc != SSA_OK) return rc;
        if (hipEventRecord(ln.ev_upd, us) != hipSuccess || hipStreamWaitEvent(st, ln.ev_upd, 0) != hipSuccess)
            return SSA_ERR_HIP;
    }
    return SSA_OK;
}

// Inverses of `batch` non-unit lower triangular blocks (kb <= NB): block b is read at
// Ablk + b * a_stride and its inverse written at out + b * o_stride with leading dimension ldo.
int trtri_lower_batched_f64(const double *Ablk, int64_t lda, int64_t a_stride, int kb, double *out,
                            int64_t ldo, int64_t o_stride, int batch, hipStream_t st) {
    return launch_trsm<double, false, true, false>(Ablk, lda, a_stride, out, ldo, o_stride, kb, kb, batch, st);
}
int trtri_lower_batched_f32(const float *Ablk, int64_t lda, int64_t a_stride, int kb, float *out,
                            int64_t ldo, int64_t o_stride, int batch, hipStream_t st) {
    return launch_trsm<float, false, true, false>(Ablk, lda, a_stride, out, ldo, o_stride, kb, kb, batch, st);
}
int trtri_lower_blocks_f64(const double *A, int64_t lda, int64_t n, double *inv, hipStream_t st) {
    return trtri_lower_blocks<double>(A, lda, n, inv, st);
}
int trtri_lower_blocks_f32(const float *A, int64_t lda, int64_t n, float *inv, hipStream_t st) {
    return trtri_lower_blocks<float>(A, lda, n, inv, st);
}

}  // namespace ssa

using namespace ssa;

extern "C" size_t ssa_lu_factor_workspace_bytes(int64_t n, int dtype) {
    const size_t es = dtype == SSA_F64 ? 8 : 4;
    const size_t rows = dtype == SSA_F64 ? PanelScratchBytes::rows<double>() : PanelScratchBytes::rows<float>();
    const size_t nsub = static_cast<size_t>(ceil_div(n, PW) + ceil_div(n, NB));
    return PanelScratchBytes::cnt + PanelScratchBytes::hdr + rows + 2 * nsub * sizeof(int) +
           static_cast<size_t>(n) * PW * es + (NB / PW + 1) * 64 * 64 * es + 12 * 256;
}

extern "C" size_t ssa_lu_aux_bytes(int64_t n, int dtype) {
    const size_t es = dtype == SSA_F64 ? 8 : 4;
    return static_cast<size_t>(lu_aux_layout(n).total) * es;
}

extern "C" int ssa_lu_factor(void *A, int64_t n, int64_t lda, int32_t *ipiv, int32_t *info,
                             void *aux, int dtype, void *workspace, size_t workspace_bytes,
                             void *stream) {
    if (!A || !ipiv || !info || !aux || n <= 0 || lda < n) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_lu_factor_workspace_bytes(n, dtype))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    if (dtype == SSA_F64)
        return getrf<double>(static_cast<double *>(A), n, lda, ipiv, info, static_cast<double *>(aux),
                             workspace, as_stream(stream));
    return getrf<float>(static_cast<float *>(A), n, lda, ipiv, info, static_cast<float *>(aux),
                        workspace, as_stream(stream));
}

extern "C" int64_t ssa_lu_padded_n(int64_t n) { return ceil_div(n, NB) * NB; }

extern "C" size_t ssa_lu_factor_nopivot_workspace_bytes(int64_t n, int dtype) {
    return np_workspace_bytes(ssa_lu_padded_n(n), dtype == SSA_F64 ? 8 : 4);
}

extern "C" int ssa_lu_factor_nopivot_batch(int count, void *const *A, const int64_t *n, const int64_t *lda,
                                           int32_t *const *ipiv, int32_t *const *info, void *const *aux, int dtype,
                                           void *const *workspace, const size_t *workspace_bytes, void *stream) {
    if (count <= 0 || !A || !n || !lda || !ipiv || !info || !aux || !workspace || !workspace_bytes)
        return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < count; ++i) {
        if (!A[i] || !ipiv[i] || !info[i] || !aux[i] || n[i] <= 0 || lda[i] < ssa_lu_padded_n(n[i]))
            return SSA_ERR_INVALID_ARGUMENT;
        if (!workspace[i] || workspace_bytes[i] < ssa_lu_factor_nopivot_workspace_bytes(n[i], dtype))
            return SSA_ERR_WORKSPACE_TOO_SMALL;
    }
    for (int first = 0; first < count; first += kMaxLuLanes) {
        const int c = (count - first < kMaxLuLanes) ? count - first : kMaxLuLanes;
        int rc;
        if (dtype == SSA_F64) {
            NpJob<double> jobs[kMaxLuLanes];
            for (int i = 0; i < c; ++i)
                jobs[i] = NpJob<double>{static_cast<double *>(A[first + i]), n[first + i], ssa_lu_padded_n(n[first + i]),
                                        lda[first + i], ipiv[first + i], info[first + i],
                                        static_cast<double *>(aux[first + i]), workspace[first + i]};
            rc = getrf_np_batch<double>(jobs, c, as_stream(stream));
        } else {
            NpJob<float> jobs[kMaxLuLanes];
            for (int i = 0; i < c; ++i)
                jobs[i] = NpJob<float>{static_cast<float *>(A[first + i]), n[first + i], ssa_lu_padded_n(n[first + i]),
                                       lda[first + i], ipiv[first + i], info[first + i],
                                       static_cast<float *>(aux[first + i]), workspace[first + i]};
            rc = getrf_np_batch<float>(jobs, c, as_stream(stream));
        }
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

extern "C" int ssa_lu_pivots_to_permutation(const int32_t *ipiv_host, int64_t n,
                                            int64_t *perm_host) {
    if (!ipiv_host || !perm_host || n < 0) return SSA_ERR_INVALID_ARGUMENT;
    for (int64_t i = 0; i < n; ++i) perm_host[i] = i;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = ipiv_host[i];
        if (p < 0 || p >= n) return SSA_ERR_INVALID_ARGUMENT;
        const int64_t t = perm_host[i];
        perm_host[i] = perm_host[p];
        perm_host[p] = t;
    }
    return SSA_OK;
}

extern "C" size_t ssa_lu_solve_workspace_bytes(int64_t n, int64_t nrhs, int dtype) {
    // X and, for several right-hand sides, the partial products of the split-K / skinny GEMMs
    const size_t nn = static_cast<size_t>(n);
    const size_t elems = (nn + 2) * static_cast<size_t>(nrhs) + (nrhs > 1 ? gemm_rhs_partial_elems(n, nrhs) : 0);
    return elems * (dtype == SSA_F64 ? 8 : 4) + 256;
}

extern "C" int ssa_lu_solve(const void *LU, int64_t n, int64_t lda, const void *aux, void *B,
                            int64_t nrhs, int64_t ldb, int dtype, void *workspace,
                            size_t workspace_bytes, void *stream) {
    if (!LU || !aux || !B || n <= 0 || nrhs <= 0 || lda < n || ldb < nrhs)
        return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_lu_solve_workspace_bytes(n, nrhs, dtype))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    if (dtype == SSA_F64)
        return getrs<double>(static_cast<const double *>(LU), n, lda, static_cast<const double *>(aux),
                             static_cast<double *>(B), nrhs, ldb, static_cast<double *>(workspace),
                             as_stream(stream));
    return getrs<float>(static_cast<const float *>(LU), n, lda, static_cast<const float *>(aux),
                        static_cast<float *>(B), nrhs, ldb, static_cast<float *>(workspace),
                        as_stream(stream));
}
