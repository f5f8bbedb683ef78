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
// Factorization: 256-column outer panels.  Panel = one diagonal-block kernel (chol_diag2.hpp: L11
// and W = L11^-1) + L21 = A21 W^T as in-place MFMA GEMMs; trailing update = one MFMA SYRK with
// K = 256 or 512.  The films of a device are factored in one interleaved schedule (potrf_batch) with two parts:
// while the trailing matrices are large, the panels run on one high-priority side stream per matrix beside the
// updates (look-ahead); once every trailing matrix is at most 10 240 columns -- where the panel chains set the
// pace -- the rest runs as ROUNDS on the caller's stream alone: per round three batched launches for all films
// (diagonal-block kernels as the first workgroups of a launch whose other workgroups are update tiles; panels;
// next block columns: chol_tail.hpp), the finishing passes in slices beside them.
//
// Solve: the factor buffer ends up holding L below and L^T above the diagonal, and `aux` the
// inverses (and their transposes) of the SNB x SNB diagonal blocks of L, so that both triangular
// solves are chains of 2 row-major GEMVs per block that stream at HBM speed
// (x_k = inv_k b_k;  b_rest -= L[rest, k] x_k), with no transposed access and no per-row dependency.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <stdio.h>

#include <string>
#include <vector>
#include <utility>

#include "chain_streams.hpp"
#include "chol_diag.hpp"
#include "chol_diag2.hpp"
#include "chol_tail.hpp"
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
int gemm_splitk_pick(int64_t M, int64_t N, int64_t K);
size_t gemm_rhs_partial_elems(int64_t m_max, int64_t nrhs);
bool gemm_skinny_ok(int64_t N, int64_t K, const void *A, int64_t lda);
int gemm_skinny_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                    int64_t ldb, double beta, double *C, int64_t ldc, int tri, double *partial, hipStream_t st);
int gemm_splitk_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                    int64_t ldb, double beta, double *C, int64_t ldc, int splits, double *partial, hipStream_t st);
int gemm_splitk_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda, const float *B,
                    int64_t ldb, double beta, float *C, int64_t ldc, int splits, float *partial, hipStream_t st);
int gemv_f64(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
             double alpha, double beta, hipStream_t st);
int gemv_f32(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
             double alpha, double beta, hipStream_t st);
int trmv_f64(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y, double alpha,
             double beta, int tri, hipStream_t st);
int trmv_f32(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y, double alpha,
             double beta, int tri, hipStream_t st);
int gemv_batch_f64(int count, const double *const *M, const int64_t *nr, const int64_t *nc, const int64_t *ldm,
                   const double *const *x, double *const *y, double alpha, double beta, int tri, hipStream_t st);
int gemv_batch_f32(int count, const float *const *M, const int64_t *nr, const int64_t *nc, const int64_t *ldm,
                   const float *const *x, float *const *y, double alpha, double beta, int tri, hipStream_t st);
int gemm_batched_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int batch1,
                     int batch2, const int64_t *strides, int tri, hipStream_t st);
int gemm_batched_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                     const float *B, int64_t ldb, double beta, float *C, int64_t ldc, int batch1,
                     int batch2, const int64_t *strides, int tri, hipStream_t st);

int gemm_batched_sliced_f64(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                            int64_t ldb, double beta, double *C, int64_t ldc, int batch1, int batch2, const int64_t *strides,
                            int tri, int64_t kchunk, int64_t max_wgs, hipStream_t st);
int gemm_batched_sliced_f32(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda, const float *B,
                            int64_t ldb, double beta, float *C, int64_t ldc, int batch1, int batch2, const int64_t *strides,
                            int tri, int64_t kchunk, int64_t max_wgs, hipStream_t st);

namespace {

constexpr int CNB = 256;  // outer panel

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
inline int gemv_n_t(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
                    double alpha, double beta, hipStream_t st) {
    return gemv_f64(M, nr, nc, ldm, x, y, alpha, beta, st);
}
inline int gemv_n_t(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y,
                    double alpha, double beta, hipStream_t st) {
    return gemv_f32(M, nr, nc, ldm, x, y, alpha, beta, st);
}
inline int gemv_batch_t(int count, const double *const *M, const int64_t *nr, const int64_t *nc, const int64_t *ldm,
                        const double *const *x, double *const *y, double alpha, double beta, int tri, hipStream_t st) {
    return gemv_batch_f64(count, M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int gemv_batch_t(int count, const float *const *M, const int64_t *nr, const int64_t *nc, const int64_t *ldm,
                        const float *const *x, float *const *y, double alpha, double beta, int tri, hipStream_t st) {
    return gemv_batch_f32(count, M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int trmv_t(const double *M, int64_t nr, int64_t nc, int64_t ldm, const double *x, double *y,
                  double alpha, double beta, int tri, hipStream_t st) {
    return trmv_f64(M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int trmv_t(const float *M, int64_t nr, int64_t nc, int64_t ldm, const float *x, float *y, double alpha,
                  double beta, int tri, hipStream_t st) {
    return trmv_f32(M, nr, nc, ldm, x, y, alpha, beta, tri, st);
}
inline int gemm_batched_t(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                          const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int b1, int b2,
                          const int64_t *strides, int tri, hipStream_t st) {
    return gemm_batched_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, st);
}
inline int gemm_batched_t(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda,
                          const float *B, int64_t ldb, double beta, float *C, int64_t ldc, int b1, int b2,
                          const int64_t *strides, int tri, hipStream_t st) {
    return gemm_batched_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, st);
}
// the same product in slices (gemm.hip): small grids, K in chunks -- beside latency-critical launches of another stream
// (grids of 128 / 64 workgroups: 99 / 105.5 ms against 95.7 ms for config H -- the passes then outlast the rounds;
// 512: no better; chunks of 256: + 0.5 ms)
constexpr int64_t kSliceK = 512, kSliceWgs = 256;
inline int gemm_sliced_t(int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
                         int64_t ldb, double beta, double *C, int64_t ldc, int b1, int b2, const int64_t *strides, int tri,
                         hipStream_t st) {
    return gemm_batched_sliced_f64(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, kSliceK, kSliceWgs, st);
}
inline int gemm_sliced_t(int64_t M, int64_t N, int64_t K, double alpha, const float *A, int64_t lda, const float *B,
                         int64_t ldb, double beta, float *C, int64_t ldc, int b1, int b2, const int64_t *strides, int tri,
                         hipStream_t st) {
    return gemm_batched_sliced_f32(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, b1, b2, strides, tri, kSliceK, kSliceWgs, st);
}

inline int small_batch_t(const double *, int n, const SmallNtJob *jobs, hipStream_t st) {
    return gemm_nt_small_batch_f64(n, jobs, st);
}
inline int small_batch_t(const float *, int n, const SmallNtJob *jobs, hipStream_t st) {
    return gemm_nt_small_batch_f32(n, jobs, st);
}
inline int tail_round_t(const double *, int n, const TailRoundJob *jobs, int exclusive, hipStream_t st) {
    return chol_tail_round_f64(n, jobs, exclusive, st);
}
inline int tail_round_t(const float *, int n, const TailRoundJob *jobs, int exclusive, hipStream_t st) {
    return chol_tail_round_f32(n, jobs, exclusive, st);
}

#ifndef SSA_SNB
#define SSA_SNB 4096
#endif
constexpr int64_t SNB = SSA_SNB;  // block size of the triangular solves (pre-inverted diagonal blocks)

// aux layout (elements):  inv [nblk][SNB][SNB] | invT [nblk][SNB][SNB] | tmp | scratch
struct AuxLayout {
    int64_t nblk, nfull, inv, invT, tmp, scratch, total;
};
inline AuxLayout aux_layout(int64_t n) {
    AuxLayout a;
    a.nblk = ceil_div(n, SNB);
    a.nfull = n / SNB;
    a.inv = 0;
    a.invT = a.nblk * SNB * SNB;
    a.tmp = 2 * a.nblk * SNB * SNB;
    a.scratch = a.tmp + a.nblk * (SNB * SNB / 4);
    // scratch of the diagonal-block kernel: one register image per lower 16 x 16 tile of the 256 x 256 block
    // (chol_diag2.hpp: 34 816 elements of the type the block is factored in -- float64 for both routes, so twice as
    // many elements of a float32 matrix)
    a.total = a.scratch + 2 * cholk2::kScratchElems;
    return a;
}

// The triangular solves work on diagonal blocks of `sblk` rows: SNB (4096, the default) or a smaller power-of-two
// multiple of 256 (2048: ssa_chol_factor_batch_blk).  The STORAGE is the same -- an inverse block of sblk rows sits on
// the block diagonal of its SNB x SNB buffer, leading dimension SNB -- but the levels of the inverse recursion from
// sblk upwards are not built: a quarter of the block-inverse flops at sblk = 2048 (the top level is three quarters
// of them), for twice the dependent launches per solve.  Worth it when a factorization serves few solves.
inline bool valid_solve_block(int64_t sblk) { return sblk >= 256 && sblk <= SNB && (sblk & (sblk - 1)) == 0; }
inline int64_t inv_block_offset(int64_t r0) { return (r0 / SNB) * SNB * SNB + (r0 % SNB) * (SNB + 1); }

// inv <- inverse of the lower triangular diagonal block L[r0 : r0 + sz, r0 : r0 + sz] whose 256-leaves
// are already inverted (chol_diag.hpp), recursively:
//     inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]]
// One product pair per split; used for the last, partial SNB block (the full ones are batched).
template <typename T>
int build_block_inverse(const T *L, int64_t lda, int64_t r0, int64_t sz, T *inv, int64_t ldi, T *tmp,
                        hipStream_t st) {
    if (sz <= 256) return SSA_OK;
    int64_t h = 256;
    while (2 * h < sz) h *= 2;
    int rc = build_block_inverse(L, lda, r0, h, inv, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    rc = build_block_inverse(L, lda, r0 + h, sz - h, inv + h * ldi + h, ldi, tmp, st);
    if (rc != SSA_OK) return rc;
    rc = gemm_nn_t(sz - h, h, h, 1.0, L + (r0 + h) * lda + r0, lda, inv, ldi, 0.0, tmp, h, st);
    if (rc != SSA_OK) return rc;
    return gemm_nn_t(sz - h, h, sz - h, -1.0, inv + h * ldi + h, ldi, tmp, h, 0.0, inv + h * ldi, ldi, st);
}

// dst tile (bj, bi) <- transpose of src tile (bi, bj) for the 64 x 64 tiles on/below the diagonal
// of a square matrix: blockIdx.x = bi, blockIdx.y = bj - bj0 (tiles above the diagonal exit at
// once), blockIdx.z = matrix of the batch.
// STRICT: src == dst, only the strictly lower elements move (upper <- lower^T, diagonal kept).
template <typename T, bool STRICT>
__global__ __launch_bounds__(256) void transpose_lower_kernel(const T *src, int64_t lds_, int64_t s_stride,
                                                              T *dst, int64_t ldd, int64_t d_stride, int64_t n,
                                                              int64_t bj0) {
    __shared__ T tile[64][65];
    const int64_t bi = blockIdx.x, bj = bj0 + blockIdx.y;
    if (bi < bj) return;
    src += blockIdx.z * s_stride;
    dst += blockIdx.z * d_stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {   // the 16 row loads of a wave in flight together (a rolled loop waits for each load before the next)
        T tmp[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int64_t r = bi * 64 + w + 4 * it, c = bj * 64 + lane;
            tmp[it] = (r < n && c < n) ? src[r * lds_ + c] : T(0);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) tile[w + 4 * it][lane] = tmp[it];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = w + 4 * it;
        const int64_t r = bj * 64 + rr, c = bi * 64 + lane;  // destination element (r, c) = source (c, r)
        if (r < n && c < n && (!STRICT || c > r)) dst[r * ldd + c] = tile[lane][rr];
    }
}

// Per concurrently factored matrix (look-ahead lane): the high-priority side stream of the panel chain
// (one of the device's chain streams, see below), its events, and a low-priority stream on which a matrix that is
// done before the others (a smaller film) builds its solve-phase blocks while the tail of the others still runs.
struct CholLane {
    hipStream_t side = nullptr, finish = nullptr, upd = nullptr;
    hipEvent_t ev_strip = nullptr, ev_panel = nullptr, ev_fork = nullptr, ev_finish = nullptr, ev_syrk = nullptr,
               ev_syrk2 = nullptr, ev_upd = nullptr;   // ev_syrk / ev_syrk2: the last two trailing updates, alternating
};
constexpr int kMaxLanes = 16;

// The lanes of one device and the mutex that serialises schedules on that device (the lanes are the
// schedule's streams and events; other devices of the process enqueue concurrently).  Created on first
// use, destroyed by ssa_shutdown().
struct LaneSet {
    CholLane lanes[kMaxLanes];
    std::mutex enqueue;
};
LaneSet g_lane_sets[kMaxDevices];
std::mutex g_lane_create_mutex;

// `st`: the stream the schedule's trailing updates will run on; the lanes' chain streams are the device's chain
// streams in the order measured against it (chain_streams.hpp).
inline int get_lanes(int count, hipStream_t st, LaneSet **out) {
    std::lock_guard<std::mutex> lock(g_lane_create_mutex);
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    LaneSet &set = g_lane_sets[dev];
    CholLane *lanes = set.lanes;  // streams belong to the device they were made on
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return SSA_ERR_HIP;
    for (int i = 0; i < count; ++i) {
        if (lanes[i].finish != nullptr) continue;
        if (hipStreamCreateWithPriority(&lanes[i].finish, hipStreamNonBlocking, lo) != hipSuccess ||
            hipStreamCreateWithPriority(&lanes[i].upd, hipStreamNonBlocking, 0) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_upd, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_strip, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_panel, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_syrk, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_syrk2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&lanes[i].ev_finish, hipEventDisableTiming) != hipSuccess)
            return SSA_ERR_HIP;
    }
    // (the chain streams are idle here: every schedule joins them into its caller's stream, and the schedules of
    // one device are serialised by set.enqueue, taken here for a measurement and by the caller after this)
    hipStream_t chains[kMaxLanes];
    {
        std::lock_guard<std::mutex> enq(set.enqueue);
        const int rc = chain_streams_get(st, kMaxLanes, chains);
        if (rc != SSA_OK) return rc;
    }
    for (int i = 0; i < kMaxLanes; ++i) lanes[i].side = chains[i];
    *out = &set;
    return SSA_OK;
}

// Destroys every lane of every device (ssa_shutdown): waits for the side streams first.
inline int destroy_lanes() {
    std::lock_guard<std::mutex> lock(g_lane_create_mutex);
    int rc = SSA_OK;
    for (int d = 0; d < kMaxDevices; ++d) {
        LaneSet &set = g_lane_sets[d];
        std::lock_guard<std::mutex> enq(set.enqueue);
        for (CholLane &ln : set.lanes) {
            if (ln.finish == nullptr) continue;
            if (hipStreamSynchronize(ln.finish) != hipSuccess || hipStreamSynchronize(ln.upd) != hipSuccess) rc = SSA_ERR_HIP;
            hipEvent_t evs[7] = {ln.ev_strip, ln.ev_panel, ln.ev_fork, ln.ev_finish, ln.ev_syrk, ln.ev_syrk2, ln.ev_upd};
            for (hipEvent_t e : evs)
                if (e != nullptr && hipEventDestroy(e) != hipSuccess) rc = SSA_ERR_HIP;
            if (hipStreamDestroy(ln.finish) != hipSuccess || hipStreamDestroy(ln.upd) != hipSuccess) rc = SSA_ERR_HIP;
            ln = CholLane{};
        }
    }
    return rc;
}

template <typename T>
struct CholJob {
    T *A;
    int64_t n, lda;
    int32_t *info;
    T *aux;
};

// One outer panel of one matrix in pieces: the diagonal-block kernel (L11 and W = L11^-1, chol_diag.hpp) and
// L21 = A21 W^T for a range of rows as two in-place MFMA GEMMs.  W is lower triangular, so columns 128..255 of
// L21 need all 256 columns of A21 (done first) and columns 0..127 only the first 128; every workgroup reads and
// writes its own rows, which makes the in-place update safe (beta = 0: C is not read).
// W lands in the solve-phase buffer: it is the inverted 256-leaf of the SNB block inverses.
template <typename T>
T *chol_leaf(const CholJob<T> &J, int64_t k0) {
    return J.aux + (k0 / SNB) * SNB * SNB + (k0 % SNB) * (SNB + 1);
}
template <typename T>
int chol_panel_diag(const CholJob<T> &J, int64_t k0, hipStream_t s) {
    const int64_t n = J.n, lda = J.lda;
    T *scratch = J.aux + aux_layout(n).scratch;
    if (lda > (int64_t(1) << 22)) return SSA_ERR_INVALID_ARGUMENT;  // 32-bit offsets inside the block
    static DeviceFlags lds_flags2;   // the tile-layout form (chol_diag2.hpp), both precisions
    constexpr size_t smem = sizeof(cholk2::Smem<cholk2::factor_t<T>>);
    if (raise_dynamic_lds(lds_flags2, {{reinterpret_cast<const void *>(&cholk2::chol_diag256_v2_kernel<T>), smem}}) != SSA_OK)
        return SSA_ERR_HIP;
    hipLaunchKernelGGL((cholk2::chol_diag256_v2_kernel<T>), dim3(1), dim3(cholk2::kThreads), smem, s,
                       J.A + k0 * (lda + 1), static_cast<int>(lda), chol_leaf(J, k0), static_cast<int>(SNB), scratch,
                       J.info, static_cast<int>(k0 + 1));
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
// rows [r0, r1) of the panel at column k0 (r0 >= k0 + CNB)
template <typename T>
int chol_panel_rows(const CholJob<T> &J, int64_t k0, int64_t r0, int64_t r1, hipStream_t s) {
    const int64_t lda = J.lda, M = r1 - r0;
    if (M <= 0) return SSA_OK;
    const T *W = chol_leaf(J, k0);
    T *A21 = J.A + r0 * lda + k0;
    int rc = gemm_op_t(0, 1, 0, M, 128, 256, 1.0, A21, lda, W + 128 * SNB, SNB, 0.0, A21 + 128, lda, s);
    if (rc != SSA_OK) return rc;
    return gemm_op_t(0, 1, 0, M, 128, 128, 1.0, A21, lda, W, SNB, 0.0, A21, lda, s);
}

// Finishing passes (solve-phase data), each over a range so that most of them can be slipped into
// the idle slots of the caller's stream during the chain-bound tail of the factorization:
//   mirror_columns   L^T into the upper triangle for the 64-wide tile columns [c0, c1) of L
//   inverse_level    one recursion level h of the inverses of the full SNB blocks [j0, j1):
//                    tmp_p = L21(p) inv11(p)  (first)  /  inv21(p) = -inv22(p) tmp_p  (second)
//                    for every pair p of size 2 h, as ONE batched GEMM with triangular K ranges
//   inverse_transposes  invT blocks [j0, j1)
// (the 256-leaves of the inverses were written by the diagonal-block kernels).
template <typename T>
int mirror_columns(const CholJob<T> &J, int64_t c0, int64_t c1, hipStream_t st) {
    if (c1 <= c0) return SSA_OK;
    const int64_t nt = ceil_div(J.n, 64);
    hipLaunchKernelGGL((transpose_lower_kernel<T, true>),
                       dim3(static_cast<unsigned>(nt), static_cast<unsigned>((c1 - c0) / 64)), dim3(256), 0, st, J.A,
                       J.lda, int64_t(0), J.A, J.lda, int64_t(0), J.n, c0 / 64);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

template <typename T>
int inverse_level(const CholJob<T> &J, int64_t h, bool second, int64_t j0, int64_t j1, hipStream_t st, bool sliced = false) {
    if (j1 <= j0) return SSA_OK;
    const int64_t lda = J.lda;
    const AuxLayout al = aux_layout(J.n);
    const int ppb = static_cast<int>(SNB / (2 * h));  // pairs per SNB block
    const int64_t pair_l = 2 * h * (lda + 1), blk_l = SNB * (lda + 1);
    const int64_t pair_i = 2 * h * (SNB + 1), blk_i = SNB * SNB;
    const int64_t pair_t = h * h, blk_t = SNB * SNB / 4;
    const T *L = J.A + j0 * blk_l;
    T *inv = J.aux + al.inv + j0 * blk_i, *tmp = J.aux + al.tmp + j0 * blk_t;
    const int nb = static_cast<int>(j1 - j0);
    if (!second) {
        const int64_t s1[6] = {pair_l, blk_l, pair_i, blk_i, pair_t, blk_t};
        if (sliced) return gemm_sliced_t(h, h, h, 1.0, L + h * lda, lda, inv, SNB, 0.0, tmp, h, ppb, nb, s1, 1, st);
        return gemm_batched_t(h, h, h, 1.0, L + h * lda, lda, inv, SNB, 0.0, tmp, h, ppb, nb, s1, 1, st);
    }
    const int64_t s2[6] = {pair_i, blk_i, pair_t, blk_t, pair_i, blk_i};
    if (sliced)
        return gemm_sliced_t(h, h, h, -1.0, inv + h * (SNB + 1), SNB, tmp, h, 0.0, inv + h * SNB, SNB, ppb, nb, s2, 2, st);
    return gemm_batched_t(h, h, h, -1.0, inv + h * (SNB + 1), SNB, tmp, h, 0.0, inv + h * SNB, SNB, ppb, nb, s2, 2,
                          st);
}

template <typename T>
int inverse_transposes(const CholJob<T> &J, int64_t j0, int64_t j1, hipStream_t st) {
    if (j1 <= j0) return SSA_OK;
    const AuxLayout al = aux_layout(J.n);
    const int64_t nt = SNB / 64;
    hipLaunchKernelGGL((transpose_lower_kernel<T, false>),
                       dim3(static_cast<unsigned>(nt), static_cast<unsigned>(nt), static_cast<unsigned>(j1 - j0)),
                       dim3(256), 0, st, J.aux + al.inv + j0 * SNB * SNB, SNB, SNB * SNB,
                       J.aux + al.invT + j0 * SNB * SNB, SNB, SNB * SNB, SNB, int64_t(0));
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

// The finishing passes of one matrix.  Full SNB blocks are finished in order as their columns become final
// (run_blocks: inverse levels, inverse transposes, mirror of their columns -- beside the rounds of the schedule from
// the matrix' low-priority stream, sliced); run_rest does what is left after the last panel.
template <typename T>
struct FinishPlan {
    const CholJob<T> *job = nullptr;
    int64_t done = 0;   // full SNB blocks whose solve-phase data have been issued
    bool skip_all = false, skip_mirror = false;   // timing experiments only (SSA_CHOL_DEBUG finish=0 / mirror=0)
    int64_t sblk = SNB;                           // rows of the solves' diagonal blocks: inverse levels below it are built

    void init(const CholJob<T> *j) {
        job = j;
        done = 0;
    }
    int64_t nfull() const { return job->n / SNB; }
    bool finished() const { return done < 0; }

    int run_blocks(int64_t upto, bool sliced, hipStream_t st) {
        const CholJob<T> &J = *job;
        if (upto > nfull()) upto = nfull();
        if (upto <= done) return SSA_OK;
        if (skip_all) {
            done = upto;
            return SSA_OK;
        }
        int rc = SSA_OK;
        for (int64_t h = 256; h < sblk; h *= 2) {
            rc = inverse_level(J, h, false, done, upto, st, sliced);
            if (rc != SSA_OK) return rc;
            rc = inverse_level(J, h, true, done, upto, st, sliced);
            if (rc != SSA_OK) return rc;
        }
        // (beside the rounds these two go out in pieces as well: a transpose launch of a whole block is 4 096 / 20 000
        // short workgroups that take every free slot for 0.2-0.7 ms -- the rounds' panel and strip launches waited
        // behind them for up to 0.9 ms)
        for (int64_t j = done; j < upto && rc == SSA_OK && !skip_mirror; j += (sliced ? 1 : upto - done))
            rc = inverse_transposes(J, j, sliced ? j + 1 : upto, st);
        if (rc != SSA_OK) return rc;
        const int64_t piece = sliced ? 512 : (upto - done) * SNB;
        for (int64_t c0 = done * SNB; c0 < upto * SNB && rc == SSA_OK && !skip_mirror; c0 += piece)
            rc = mirror_columns(J, c0, std::min(c0 + piece, upto * SNB), st);
        if (rc != SSA_OK) return rc;
        done = upto;
        return SSA_OK;
    }
    int run_rest(bool sliced, hipStream_t st) {
        const CholJob<T> &J = *job;
        const AuxLayout al = aux_layout(J.n);
        const int64_t first = done;
        int rc = SSA_OK;
        if (skip_all) {
            done = -1;
            return SSA_OK;
        }
        for (int64_t h = 256; h < sblk; h *= 2) {
            rc = inverse_level(J, h, false, first, al.nfull, st, sliced);
            if (rc != SSA_OK) return rc;
            rc = inverse_level(J, h, true, first, al.nfull, st, sliced);
            if (rc != SSA_OK) return rc;
        }
        // the last, partial SNB block: its solve blocks one by one
        for (int64_t r0 = al.nfull * SNB; r0 < J.n; r0 += sblk) {
            rc = build_block_inverse(J.A, J.lda, r0, std::min(sblk, J.n - r0), J.aux + al.inv + inv_block_offset(r0), SNB,
                                     J.aux + al.tmp + al.nfull * (SNB * SNB / 4), st);
            if (rc != SSA_OK) return rc;
        }
        if (!skip_mirror) {
            rc = inverse_transposes(J, first, al.nblk, st);
            if (rc != SSA_OK) return rc;
            rc = mirror_columns(J, first * SNB, J.n, st);
        }
        done = -1;
        return rc;
    }
};

// Debugging knobs of the schedule (never set in production; tools/chol_race_hunt.py): the environment variable
// SSA_CHOL_DEBUG holds comma-separated `name=value` pairs, read at every call:
//   split=0|1   trailing updates of >= 3 matrices on streams of their own (default) or all on the caller's stream
//   tail=N      switch to single-stream rounds at N trailing columns (default 10 240; 0: no rounds)
//   late=1      no finishing passes beside the schedule: everything after the last panel, on the caller's stream
//   delay=0     every trailing update applies ONE panel (K = 256)
//   sync=1      the host waits for the device after every outer step (serialises the streams: no overlap at all)
//   excl=0      round launches never ask for a CU per workgroup
//   early=0     no finishing passes during the STREAM part (the blocks that become final there wait for the rounds)
//   look=D      look-ahead depth of the stream part in block columns (default: 3 for a single matrix, else 1)
//   trace=1     every diagonal-block workgroup of a round stores the block AS IT READ IT (register images, 272 KB);
//               after the schedule the host waits and writes all of them to the file SSA_CHOL_TRACE_FILE:
//               [matrix][panel][34 816] float64 (float64 matrices only)
//   finish=0    TIMING EXPERIMENT, WRONG SOLVES: no finishing passes at all (what do they cost the schedule?)
//   mirror=0    TIMING EXPERIMENT, WRONG SOLVES: finishing passes without the L^T mirror and the inverse transposes
// A value that is set is reported once per distinct string on stderr: a stray variable must not change a production
// schedule silently.
struct CholDebug {
    int split = -1, late = 0, delay = 1, sync = 0, excl = 1, trace = 0, look = 0, early = 1, finish = 1, mirror = 1;
    int64_t tail = -1;
};
inline CholDebug chol_debug() {
    CholDebug d;
    const char *e = getenv("SSA_CHOL_DEBUG");
    if (e == nullptr || *e == 0) return d;
    std::string str(e);
    {
        static std::mutex seen_mutex;
        static std::vector<std::string> seen;
        std::lock_guard<std::mutex> lock(seen_mutex);
        if (std::find(seen.begin(), seen.end(), str) == seen.end()) {
            seen.push_back(str);
            fprintf(stderr, "superscreen_hip: SSA_CHOL_DEBUG=%s changes the Cholesky schedule (debugging aid)\n", e);
        }
    }
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find(',', pos);
        if (end == std::string::npos) end = str.size();
        const std::string item = str.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            const std::string key = item.substr(0, eq);
            const long long val = atoll(item.c_str() + eq + 1);
            if (key == "split") d.split = static_cast<int>(val);
            else if (key == "tail") d.tail = val;
            else if (key == "late") d.late = static_cast<int>(val);
            else if (key == "delay") d.delay = static_cast<int>(val);
            else if (key == "sync") d.sync = static_cast<int>(val);
            else if (key == "excl") d.excl = static_cast<int>(val);
            else if (key == "look") d.look = static_cast<int>(val);
            else if (key == "early") d.early = static_cast<int>(val);
            else if (key == "trace") d.trace = static_cast<int>(val);
            else if (key == "finish") d.finish = static_cast<int>(val);
            else if (key == "mirror") d.mirror = static_cast<int>(val);
        }
        pos = end + 1;
    }
    return d;
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
// Per matrix and outer step k:   block column C[:, 0:256] -= P P[0:256]^T (diagonal block and the strip below it in
//                                one product), diagonal-block kernel of panel k+1, panel k+1 = A21 W^T
//                                                                             (side stream: three dependent launches)
//                                rest    C[256:, 256:] -= P2 P2^T  (lower)    (caller's stream; for a large
//                                        trailing matrix every other step: P = the last two panels, K = 512)
//
// The chain is ONE stream per matrix (rounds 2 and 3 ran it as two, with up to six hand-offs per round; a dependent
// launch on the same stream costs 1-5 us, a hand-off between streams 25-130 us once four or more queues are active:
// tools/probes/gap_probe.hip, profiles/r03_gap_probe.txt).
//
// ROUNDS (round 4).  In the last third of a factorization an update takes less time than a round of a chain
// (0.3-0.5 ms beside running updates: the diagonal-block kernel shares its CU's vector ALUs with an update
// workgroup, its products wait for workgroup slots), the update stream idles and the schedule depends on which
// hardware queue the chains got.  From there on every film advances one panel per round and a round is three
// launches on the caller's stream, no events, no side streams:
//     round launch   workgroups 0 .. films-1: diagonal-block kernel of block c of each film; the others: the pending
//                    panel(s) onto the lower tiles BEHIND block column c -- the update the diagonal block does not
//                    need runs under it.  With few tiles the launch asks for a CU per workgroup: the diagonal-block
//                    kernel then takes the 175 us it takes alone;
//     panel launch   L21 = A21 W^T of all films, one workgroup per 32 rows;
//     strip launch   block column c + 256 of all films -= the new panel (it has the older ones from the round launch).
// A chain-bound round is 235 us (175 + 25 + 29 + launch gaps) against 300-570 us for a round of a chain of the stream
// schedule.  The finishing passes of the blocks that are final ride beside the chain-bound rounds on the films'
// low-priority streams, in slices (gemm_batched_sliced) so that the rounds' launches always find free slots.
template <typename T>
int potrf_batch(const CholJob<T> *jobs, int count, hipStream_t st, int64_t sblk = SNB) {
    if (count <= 0 || count > kMaxLanes || !valid_solve_block(sblk)) return SSA_ERR_INVALID_ARGUMENT;
    // the lanes (side streams + events) are shared per device: one schedule per device is enqueued at a time
    LaneSet *lane_set = nullptr;
    int rc = get_lanes(count, st, &lane_set);
    if (rc != SSA_OK) return rc;
    std::lock_guard<std::mutex> enqueue_lock(lane_set->enqueue);
    CholLane *lanes = lane_set->lanes;
    // With three or more matrices the trailing updates of matrix i go to a stream of their own instead of
    // alternating on the caller's stream: the partially filled last round of tiles of one update then overlaps
    // the next update of another matrix, and a matrix' chain no longer waits behind the updates of all the
    // others (4 x 30 301-vertex stack: 350 -> 317 ms; two matrices: no gain, 156.6 vs 158.7 ms, so they keep the
    // single stream).
    const CholDebug dbg = chol_debug();
    const bool split_updates = dbg.split < 0 ? count >= 3 : (dbg.split != 0 && count >= 2);
    // Once the trailing matrix of EVERY film is at most this order the schedule is bound by the panel chains (a
    // round of a chain takes 0.3-0.5 ms beside running updates, an update of that size less): the rest of the
    // factorization runs as single-stream rounds of batched launches (see the loop below).
    // (round 4, same-box runs: config H float64 99.5 / 97.0 / 96.8 / 95.9 / 96.9 / 99.2 / 101.3 ms for 0 / 6144 / 8192 /
    // 10240 / 12288 / 16384 / all columns, float32 57.7 -> 52.6, K = 81 60.8 -> 54.8, one film K = 64 18.0 -> 15.1,
    // K = 129 407 -> 405, four films 361 -> 350 ms.  Larger: the rounds' panel and strip launches are not hidden
    // behind an update the way the chains' are)
    const int64_t tail_round_cols = dbg.tail >= 0 ? dbg.tail : 10240;
    // a round launch with at most this many update tiles asks for a CU per workgroup (chain-bound rounds: the
    // diagonal-block workgroups then run alone on their CUs, 175 us instead of 220-300; 0 / 2500: + 0.5 ms)
    constexpr int64_t tail_excl_tiles = 1024;
    // the finishing passes of the blocks that are final go out beside the rounds once a round has at most this many
    // update tiles (before that the rounds are update bound and the passes would only lengthen them; 800 / 3000 /
    // always: the same within 0.5 ms, never: + 3 ms)
    constexpr int64_t fill_tiles = 1500;
    FinishPlan<T> plans[kMaxLanes];
    int64_t nmax = 0;
    // (debugging, trace=1: one slot of register images per matrix and panel, zero where no round ran)
    double *trace_buf = nullptr;
    struct TraceGuard {   // (every early return below frees the trace buffer; the device may still be writing it, so wait)
        double *&p;
        ~TraceGuard() {
            if (p != nullptr) {
                (void)hipDeviceSynchronize();
                (void)hipFree(p);
                p = nullptr;
            }
        }
    } trace_guard{trace_buf};
    int64_t trace_panels = 0;
    if (dbg.trace && sizeof(T) == 8) {
        for (int i = 0; i < count; ++i) trace_panels = std::max<int64_t>(trace_panels, jobs[i].n / CNB);
        const size_t bytes = static_cast<size_t>(count) * trace_panels * cholk2::kScratchElems * sizeof(double);
        if (hipMalloc(reinterpret_cast<void **>(&trace_buf), bytes) != hipSuccess ||
            hipMemsetAsync(trace_buf, 0, bytes, st) != hipSuccess)
            return SSA_ERR_HIP;
    }
    for (int i = 0; i < count; ++i) {
        const CholJob<T> &J = jobs[i];
        CholLane &ln = lanes[i];
        plans[i].init(&jobs[i]);
        plans[i].sblk = sblk;
        plans[i].skip_all = dbg.finish == 0;
        plans[i].skip_mirror = dbg.mirror == 0;
        if (J.n % CNB != 0) return SSA_ERR_INVALID_ARGUMENT;  // callers pad (potrf_padded_batch)
        if (J.lda > (int64_t(1) << 22)) return SSA_ERR_INVALID_ARGUMENT;  // 32-bit offsets inside a diagonal block
        if (J.n > nmax) nmax = J.n;
        if (hipMemsetAsync(J.info, 0, sizeof(int32_t), st) != hipSuccess ||
            hipMemsetAsync(J.aux, 0, static_cast<size_t>(aux_layout(J.n).tmp) * sizeof(T), st) != hipSuccess)
            return SSA_ERR_HIP;
        // fork: the side stream starts after everything enqueued so far on the caller's stream
        if (hipEventRecord(ln.ev_strip, st) != hipSuccess || hipStreamWaitEvent(ln.side, ln.ev_strip, 0) != hipSuccess)
            return SSA_ERR_HIP;
        if (split_updates && hipStreamWaitEvent(ln.upd, ln.ev_strip, 0) != hipSuccess) return SSA_ERR_HIP;
        rc = chol_panel_diag(J, 0, ln.side);
        if (rc != SSA_OK) return rc;
        rc = chol_panel_rows(J, 0, CNB, J.n, ln.side);
        if (rc != SSA_OK) return rc;
        if (hipEventRecord(ln.ev_panel, ln.side) != hipSuccess) return SSA_ERR_HIP;
    }
    bool detached[kMaxLanes] = {};
    bool on_finish[kMaxLanes] = {};    // finishing steps of this matrix have gone to its low-priority stream
    // State of the stream part, per matrix.  The trailing updates maintain the region T = A[tstart:, tstart:]: every
    // panel before pending_from has been applied to it.  Block columns between the panel being factored and tstart are
    // LOOK-AHEAD columns: they left T when an update was issued that no longer covered them, with the panels before
    // col_from[] applied, and take the rest from the chain's block-column product when their turn comes.
    //   look = 1: an update covers everything right of the panel being factored; the chain's next block-column
    //             product waits for the update issued a step earlier (two or more matrices fill that wait with each
    //             other's updates);
    //   look = 3: an update leaves the next TWO block columns to the chain (K = 768 / 1 024 products there instead
    //             of 256 / 512): the chain's products then wait for the update BEFORE the running one, i.e. never,
    //             and a single matrix' updates follow each other without the gap of two chain steps (round 5).
    // In float64 the factor does not depend on `look`, bit for bit: every tile still takes its panels in ascending order.
    constexpr int kColRing = 8;
    if (dbg.look > kColRing) return SSA_ERR_INVALID_ARGUMENT;   // (two outstanding look-ahead columns would share a ring slot)
    const int look = dbg.look > 0 ? dbg.look : (count == 1 ? 3 : 1);
    int64_t pending_from[kMaxLanes] = {};
    int64_t tstart[kMaxLanes];
    int64_t col_from[kMaxLanes][kColRing] = {};
    int col_writer[kMaxLanes][kColRing] = {};   // index of the last trailing update that covered the column (a slot is
                                                // only read after the update that retires its column has set it)
    int n_syrk[kMaxLanes] = {};                 // trailing updates issued so far; update k records ev_syrk / ev_syrk2 by k & 1
    for (int i = 0; i < count; ++i) tstart[i] = CNB;
    auto slot = [](int64_t col) { return static_cast<int>((col / CNB) % kColRing); };
    auto syrk_event = [&](int i, int k) { return (k & 1) ? lanes[i].ev_syrk2 : lanes[i].ev_syrk; };
    // trailing updates of a large trailing matrix are applied two panels at a time (K = 512): the C tiles
    // are then read and written once per 32 LDS stages instead of 16 (50 -> 63 TFLOP/s per launch,
    // tools/probes/syrk_k_probe.py); deeper (K = 768, 1024) leaves too little between the chains' products
    const int kDelayDepth = dbg.delay ? 2 : 1;            // (3 / 4 panels: 107 / 108 against 101.5-102.8 ms in round 3; with the rounds
                                              // behind the stream part, round 4: 96.4 / 100.4 against 94.6 ms)
    constexpr int64_t kDelayMinCols = 8192;   // (2048 ... 16384: flat within 1 %, rounds 2 and 3)
    hipStream_t cur_us[kMaxLanes];
    for (int i = 0; i < count; ++i) cur_us[i] = split_updates ? lanes[i].upd : st;
    bool in_rounds = false;
    for (int64_t k0 = 0; k0 + CNB < nmax; k0 += CNB) {
        const int64_t c = k0 + CNB;   // first column of the next panel
        if (dbg.sync && hipDeviceSynchronize() != hipSuccess) return SSA_ERR_HIP;
        // ---- films that have run out of panels while others still have some: their remaining finishing steps go to
        // their own low-priority stream, behind their last panel and behind what their update stream holds for them
        for (int i = 0; i < count; ++i) {
            const CholJob<T> &J = jobs[i];
            CholLane &ln = lanes[i];
            if (c < J.n || detached[i] || dbg.late) continue;
            detached[i] = true;
            if (hipEventRecord(ln.ev_fork, cur_us[i]) != hipSuccess || hipStreamWaitEvent(ln.finish, ln.ev_fork, 0) != hipSuccess ||
                hipStreamWaitEvent(ln.finish, ln.ev_panel, 0) != hipSuccess)
                return SSA_ERR_HIP;
            rc = plans[i].run_rest(in_rounds, ln.finish);   // (beside rounds: in slices)
            if (rc != SSA_OK) return rc;
            if (hipEventRecord(ln.ev_finish, ln.finish) != hipSuccess) return SSA_ERR_HIP;
            on_finish[i] = true;
        }
        // ---- switch to single-stream rounds: the chains and update streams join the caller's stream, and the next
        // block column of every film takes its pending panels (what the chain's first product of a step does)
        if (!in_rounds && tail_round_cols > 0 && nmax - c <= tail_round_cols) {
            SmallNtJob strips[kMaxLanes];
            int ns = 0;
            for (int i = 0; i < count; ++i) {
                const CholJob<T> &J = jobs[i];
                CholLane &ln = lanes[i];
                if (c >= J.n) continue;
                if (hipStreamWaitEvent(st, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;
                if (cur_us[i] != st && (hipEventRecord(ln.ev_upd, cur_us[i]) != hipSuccess ||
                                        hipStreamWaitEvent(st, ln.ev_upd, 0) != hipSuccess))
                    return SSA_ERR_HIP;
                cur_us[i] = st;
                const int64_t right = J.n - c;
                const int64_t upd0 = (c >= tstart[i]) ? pending_from[i] : col_from[i][slot(c)];
                const T *P = J.A + c * J.lda + upd0;
                strips[ns++] = SmallNtJob{P, P, nullptr, J.A + c * J.lda + c, J.lda, J.lda, J.lda, right, CNB, c - upd0,
                                          -1.0, 1.0, 0};
                // look-ahead columns right of c catch up with the region the updates maintained: from here on
                // A[c + 256:, c + 256:] is one trailing matrix again, every panel before pending_from applied
                for (int64_t b = c + CNB; b < tstart[i] && b < J.n; b += CNB) {
                    const int64_t f = col_from[i][slot(b)];
                    if (f >= pending_from[i]) continue;
                    const T *Pb = J.A + b * J.lda + f;
                    const SmallNtJob job{Pb, Pb, nullptr, J.A + b * J.lda + b, J.lda, J.lda, J.lda, J.n - b, CNB,
                                         pending_from[i] - f, -1.0, 1.0, 0};
                    rc = small_batch_t(static_cast<const T *>(nullptr), 1, &job, st);
                    if (rc != SSA_OK) return rc;
                }
            }
            rc = small_batch_t(static_cast<const T *>(nullptr), ns, strips, st);
            if (rc != SSA_OK) return rc;
            in_rounds = true;
        }
        if (in_rounds) {
            // ---- one round of every film on the caller's stream.  State of a film at this point: block column c has
            // every earlier panel applied, the trailing matrix behind it every panel before pending_from.
            //   round launch: diagonal-block kernel of block c (first workgroups of the launch)  |  the pending panels
            //                 onto the lower tiles of A[c + 256:, c + 256:] (the other workgroups)
            //   panel launch: L21 = A21 W^T, one workgroup per 32 rows
            //   strip launch: block column c + 256 -= L21 L21[0:256]^T
            TailRoundJob rj[kMaxLanes];
            SmallNtJob pj[kMaxLanes], sj[kMaxLanes];
            int nr = 0, np = 0;
            int64_t tiles = 0;
            for (int i = 0; i < count; ++i) {
                const CholJob<T> &J = jobs[i];
                if (c >= J.n) continue;
                const int64_t below = J.n - c - CNB;   // rows under the diagonal block = order of what is behind it
                const int64_t upd0 = pending_from[i];
                T *A21 = J.A + (c + CNB) * J.lda + c;
                TailRoundJob &r = rj[nr++];
                r.D = J.A + c * (J.lda + 1);
                r.W = chol_leaf(J, c);
                r.scratch = J.aux + aux_layout(J.n).scratch;
                r.info = J.info;
                r.lda = static_cast<int>(J.lda);
                r.ldw = static_cast<int>(SNB);
                r.col1 = static_cast<int>(c + 1);
                r.has_diag = 1;
                r.C = J.A + (c + CNB) * (J.lda + 1);
                r.P = J.A + (c + CNB) * J.lda + upd0;
                r.ldc = J.lda;
                r.M = below;
                r.K = c - upd0;
                if (trace_buf != nullptr)
                    r.trace = trace_buf + (static_cast<int64_t>(i) * trace_panels + c / CNB) * cholk2::kScratchElems;
                tiles += (below / 128) * (below / 128 + 1) / 2;
                pending_from[i] = c;
                if (below <= 0) continue;
                const T *W = chol_leaf(J, c);
                pj[np] = SmallNtJob{A21, W, W + 128 * SNB, A21, J.lda, SNB, J.lda, below, CNB, CNB, 1.0, 0.0, 1};
                sj[np] = SmallNtJob{A21, A21, nullptr, A21 + CNB, J.lda, J.lda, J.lda, below, CNB, CNB, -1.0, 1.0, 0};
                ++np;
            }
            rc = tail_round_t(static_cast<const T *>(nullptr), nr, rj, (dbg.excl && tiles <= tail_excl_tiles) ? 1 : 0, st);
            if (rc != SSA_OK) return rc;
            rc = small_batch_t(static_cast<const T *>(nullptr), np, pj, st);
            if (rc != SSA_OK) return rc;
            rc = small_batch_t(static_cast<const T *>(nullptr), np, sj, st);
            if (rc != SSA_OK) return rc;
            // the finishing passes of a film's blocks that are final (all their columns lie left of c) fill the chip
            // beside the chain-bound rounds from the film's low-priority stream: in slices, so that no launch of theirs
            // holds more than a quarter of the chip's workgroup slots or a slot for longer than a round takes
            if (tiles <= fill_tiles && !dbg.late) {
                for (int i = 0; i < count; ++i) {
                    FinishPlan<T> &fp = plans[i];
                    CholLane &ln = lanes[i];
                    const int64_t ready = c / SNB;
                    if (detached[i] || fp.finished() || ready <= fp.done || fp.done >= fp.nfull()) continue;
                    if (hipEventRecord(ln.ev_fork, st) != hipSuccess || hipStreamWaitEvent(ln.finish, ln.ev_fork, 0) != hipSuccess)
                        return SSA_ERR_HIP;
                    rc = fp.run_blocks(ready, true, ln.finish);
                    if (rc != SSA_OK) return rc;
                    if (hipEventRecord(ln.ev_finish, ln.finish) != hipSuccess) return SSA_ERR_HIP;
                    on_finish[i] = true;
                }
            }
            continue;
        }
        // ---- body: panel chains on side streams beside the updates
        for (int i = 0; i < count; ++i) {
            const CholJob<T> &J = jobs[i];
            CholLane &ln = lanes[i];
            if (c >= J.n) continue;
            const int64_t right = J.n - c;                    // order of the trailing matrix
            hipStream_t us = cur_us[i];                       // where this matrix' trailing updates run
            hipStream_t cs = ln.side;                         // the matrix' panel chain
            const int64_t nw = (right < CNB) ? right : CNB;   // width of the next panel
            // panels that the trailing region has not seen yet: everything after the last trailing update; the block
            // column of this step may have left that region earlier (look-ahead column) and then needs more
            const int64_t upd0 = pending_from[i], kp = c - upd0;
            const bool in_T = c >= tstart[i];
            const int64_t col0 = in_T ? upd0 : col_from[i][slot(c)];
            const int writer = in_T ? n_syrk[i] - 1 : col_writer[i][slot(c)];   // the last update that wrote this column
            const T *P = J.A + c * J.lda + col0;              // the column's pending panels, from its diagonal block down
            T *C = J.A + c * J.lda + c;
            // every other panel of a large trailing matrix keeps its update pending: the next one then
            // runs with K = 512, i.e. half the C-tile traffic per flop (50 -> 63 TFLOP/s per launch)
            // (the phase comes from the matrix' own size, not from its place in the batch: in float64, where the
            // tiles accumulate onto C in k order, the factor of a matrix does not depend on what else is factored
            // with it -- bit for bit; in float32, where a launch adds C once, it does to rounding: the switch to
            // rounds comes with the LARGEST matrix of the batch)
            const bool delay = kp < kDelayDepth * CNB && right > kDelayMinCols &&
                               ((k0 + J.n) / CNB) % kDelayDepth != kDelayDepth - 1;
            if (hipStreamWaitEvent(us, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;  // panel k done
            // the chain: pending panels onto the next block column (behind the last trailing update of THIS matrix
            // that wrote that column; updates of one matrix run in order on one stream, so waiting for a later one
            // than necessary is safe), diagonal-block kernel, the panel below it
            if (writer >= 0 && hipStreamWaitEvent(cs, syrk_event(i, writer), 0) != hipSuccess) return SSA_ERR_HIP;
            rc = gemm_op_t(0, 1, 0, right, nw, c - col0, -1.0, P, J.lda, P, J.lda, 1.0, C, J.lda, cs);
            if (rc != SSA_OK) return rc;
            rc = chol_panel_diag(J, c, cs);
            if (rc != SSA_OK) return rc;
            if (right > nw) {
                rc = chol_panel_rows(J, c, c + CNB, J.n, cs);
                if (rc != SSA_OK) return rc;
            }
            if (hipEventRecord(ln.ev_panel, cs) != hipSuccess) return SSA_ERR_HIP;
            if (tstart[i] < c + nw) tstart[i] = c + nw;       // (this column is no longer part of the trailing region)
            if (right > nw && !delay) {
                // rest of the trailing update: the lower tiles behind the look-ahead columns.  Columns that leave the
                // region now have seen the panels before upd0, written last by the update before this one.
                const int64_t rstart = std::max(tstart[i], std::min(J.n, c + static_cast<int64_t>(look) * CNB));
                for (int64_t b = tstart[i]; b < rstart; b += CNB) {
                    col_from[i][slot(b)] = upd0;
                    col_writer[i][slot(b)] = n_syrk[i] - 1;
                }
                tstart[i] = rstart;
                const int64_t M = J.n - rstart;
                if (M > 0) {
                    const T *P2 = J.A + rstart * J.lda + upd0;
                    rc = gemm_op_t(0, 1, 1, M, M, kp, -1.0, P2, J.lda, P2, J.lda, 1.0, J.A + rstart * (J.lda + 1), J.lda, us);
                    if (rc != SSA_OK) return rc;
                    if (hipEventRecord(syrk_event(i, n_syrk[i]), us) != hipSuccess) return SSA_ERR_HIP;
                    ++n_syrk[i];
                    pending_from[i] = c;
                }
                // (M == 0: nothing is left behind the look-ahead columns; they take their panels from the chain alone)
            }
            // the solve-phase data of the SNB blocks that have become final (all their columns lie left of c): whole
            // launches on the matrix' low-priority stream, behind the chain (beside the rounds they go out in slices)
            FinishPlan<T> &fp = plans[i];
            if (!dbg.late && dbg.early && c / SNB > fp.done && fp.done < fp.nfull()) {
                if (hipStreamWaitEvent(ln.finish, ln.ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;
                rc = fp.run_blocks(c / SNB, false, ln.finish);
                if (rc != SSA_OK) return rc;
                if (hipEventRecord(ln.ev_finish, ln.finish) != hipSuccess) return SSA_ERR_HIP;
                on_finish[i] = true;
            }
        }
    }
    for (int i = 0; i < count; ++i) {  // join, then the inverses of the diagonal blocks
        if (!in_rounds && hipStreamWaitEvent(st, lanes[i].ev_panel, 0) != hipSuccess) return SSA_ERR_HIP;
        if (on_finish[i] && hipStreamWaitEvent(st, lanes[i].ev_finish, 0) != hipSuccess) return SSA_ERR_HIP;
        if (cur_us[i] != st && (hipEventRecord(lanes[i].ev_upd, cur_us[i]) != hipSuccess ||
                                hipStreamWaitEvent(st, lanes[i].ev_upd, 0) != hipSuccess))
            return SSA_ERR_HIP;
    }
    for (int i = 0; i < count; ++i) {
        if (plans[i].finished()) continue;
        rc = plans[i].run_rest(false, st);
        if (rc != SSA_OK) return rc;
    }
    if (trace_buf != nullptr) {   // (debugging: the host waits here)
        const size_t elems = static_cast<size_t>(count) * trace_panels * cholk2::kScratchElems;
        std::vector<double> host(elems);
        const bool ok = hipStreamSynchronize(st) == hipSuccess &&
                        hipMemcpy(host.data(), trace_buf, elems * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(trace_buf);
        trace_buf = nullptr;
        const char *path = getenv("SSA_CHOL_TRACE_FILE");
        if (ok && path != nullptr) {
            if (FILE *f = fopen(path, "wb")) {
                fwrite(host.data(), sizeof(double), elems, f);
                fclose(f);
            }
        }
        if (!ok) return SSA_ERR_HIP;
    }
    return SSA_OK;
}

// L L^T X = B on the finished factor buffer (L below, L^T above the diagonal).  Per SNB block two
// launches: apply the inverted diagonal block, then update the rest of the right-hand side with the
// block column of L (forward) / of L^T (backward).  Single right-hand side: row-major GEMVs;
// several: the same recurrence on the MFMA GEMM.
template <typename T>
int potrs(const T *L, int64_t n, int64_t lda, const T *aux, T *B, int64_t nrhs, int64_t ldb, T *X,
          T *partial, hipStream_t st, int64_t sblk = SNB) {
    if (!valid_solve_block(sblk)) return SSA_ERR_INVALID_ARGUMENT;
    const AuxLayout al = aux_layout(n);
    const int64_t ldx = nrhs, nblk = ceil_div(n, sblk);
    const bool vec = (nrhs == 1 && ldb == 1);
    int rc;
    for (int64_t k = 0; k < nblk; ++k) {  // forward: L y = b
        const int64_t r0 = k * sblk, kb = (n - r0 < sblk) ? n - r0 : sblk;
        const int64_t below = n - r0 - kb;
        const T *inv = aux + al.inv + inv_block_offset(r0);
        if (vec) {
            rc = trmv_t(inv, kb, kb, SNB, B + r0, X + r0, 1.0, 0.0, 1, st);
            if (rc == SSA_OK && below > 0)
                rc = gemv_n_t(L + (r0 + kb) * lda + r0, below, kb, lda, X + r0, B + r0 + kb, -1.0, 1.0, st);
        } else {
            rc = gemm_rhs_t(kb, nrhs, kb, 1.0, inv, SNB, B + r0 * ldb, ldb, 0.0, X + r0 * ldx, ldx, 1, partial, st);
            if (rc == SSA_OK && below > 0)
                rc = gemm_rhs_t(below, nrhs, kb, -1.0, L + (r0 + kb) * lda + r0, lda, X + r0 * ldx, ldx, 1.0,
                                B + (r0 + kb) * ldb, ldb, 0, partial, st);
        }
        if (rc != SSA_OK) return rc;
    }
    for (int64_t k = nblk - 1; k >= 0; --k) {  // backward: L^T x = y   (y lives in X, x goes to B)
        const int64_t r0 = k * sblk, kb = (n - r0 < sblk) ? n - r0 : sblk;
        const T *invT = aux + al.invT + inv_block_offset(r0);
        const T *U = L + r0;  // rows 0 .. r0-1 of L^T, columns of this block
        if (vec) {
            rc = trmv_t(invT, kb, kb, SNB, X + r0, B + r0, 1.0, 0.0, 2, st);
            if (rc == SSA_OK && r0 > 0) rc = gemv_n_t(U, r0, kb, lda, B + r0, X, -1.0, 1.0, st);
        } else {
            rc = gemm_rhs_t(kb, nrhs, kb, 1.0, invT, SNB, X + r0 * ldx, ldx, 0.0, B + r0 * ldb, ldb, 2, partial, st);
            if (rc == SSA_OK && r0 > 0)
                rc = gemm_rhs_t(r0, nrhs, kb, -1.0, U, lda, B + r0 * ldb, ldb, 1.0, X, ldx, 0, partial, st);
        }
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

}  // namespace
}  // namespace ssa

using namespace ssa;

namespace ssa {
int chol_shutdown() { return destroy_lanes(); }
}  // namespace ssa

extern "C" size_t ssa_chol_aux_bytes(int64_t n, int dtype) {
    const int64_t np = ceil_div(n, CNB) * CNB;
    return static_cast<size_t>(aux_layout(np).total) * (dtype == SSA_F64 ? 8 : 4);
}

extern "C" int ssa_chol_chain_stream_costs(double *microseconds, int32_t *pipe_group, int capacity) {
    return chain_streams_costs(microseconds, pipe_group, capacity);
}

extern "C" int ssa_chol_chain_streams_invalidate(void) { return chain_streams_invalidate(); }

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
                       void *const *aux, hipStream_t st, int64_t sblk) {
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
    return potrf_batch<T>(jobs, count, st, sblk);
}
template <typename T>
int potrs_padded(const T *L, int64_t n, int64_t lda, const T *aux, T *B, int64_t nrhs, int64_t ldb,
                 T *ws, hipStream_t st, int64_t sblk = SNB) {
    const int64_t np = ssa_chol_padded_n(n);
    T *X = ws, *Bp = ws + np * nrhs;
    if (hipMemcpy2DAsync(Bp, nrhs * sizeof(T), B, ldb * sizeof(T), nrhs * sizeof(T), n, hipMemcpyDeviceToDevice,
                         st) != hipSuccess)
        return SSA_ERR_HIP;
    if (np > n && hipMemsetAsync(Bp + n * nrhs, 0, (np - n) * nrhs * sizeof(T), st) != hipSuccess)
        return SSA_ERR_HIP;
    const int rc = potrs<T>(L, np, lda, aux, Bp, nrhs, nrhs, X, ws + 2 * np * nrhs, st, sblk);
    if (rc != SSA_OK) return rc;
    if (hipMemcpy2DAsync(B, ldb * sizeof(T), Bp, nrhs * sizeof(T), nrhs * sizeof(T), n, hipMemcpyDeviceToDevice,
                         st) != hipSuccess)
        return SSA_ERR_HIP;
    return SSA_OK;
}
// Single right-hand-side solves of several factors (the films of a device in one Jacobi pass) in lockstep: step s of
// every solve in ONE launch per kind (blas1.hip gemv_batch_kernel) - the same launches as `count` potrs calls, side
// by side in one grid; a film with fewer blocks sits out the last steps.  Results bit-identical to separate solves.
constexpr int kSolveBatchMax = 8;
template <typename T>
int potrs_vec_batch(int count, const T *const *L, const int64_t *n, const int64_t *lda, const T *const *aux, T *const *B,
                    T *const *X, hipStream_t st, int64_t sblk = SNB) {
    if (!valid_solve_block(sblk)) return SSA_ERR_INVALID_ARGUMENT;
    int64_t nblk[kSolveBatchMax], maxblk = 0;
    for (int i = 0; i < count; ++i) {
        nblk[i] = ceil_div(n[i], sblk);
        if (nblk[i] > maxblk) maxblk = nblk[i];
    }
    const T *M[kSolveBatchMax];
    const T *x[kSolveBatchMax];
    T *y[kSolveBatchMax];
    int64_t nr[kSolveBatchMax], nc[kSolveBatchMax], ld[kSolveBatchMax];
    int rc;
    for (int64_t s = 0; s < maxblk; ++s) {   // forward: L y = b
        int m = 0;
        for (int i = 0; i < count; ++i) {
            if (s >= nblk[i]) continue;
            const AuxLayout al = aux_layout(n[i]);
            const int64_t r0 = s * sblk, kb = (n[i] - r0 < sblk) ? n[i] - r0 : sblk;
            M[m] = aux[i] + al.inv + inv_block_offset(r0);
            nr[m] = nc[m] = kb;
            ld[m] = SNB;
            x[m] = B[i] + r0;
            y[m] = X[i] + r0;
            ++m;
        }
        rc = gemv_batch_t(m, M, nr, nc, ld, x, y, 1.0, 0.0, 1, st);
        if (rc != SSA_OK) return rc;
        m = 0;
        for (int i = 0; i < count; ++i) {
            if (s >= nblk[i]) continue;
            const int64_t r0 = s * sblk, kb = (n[i] - r0 < sblk) ? n[i] - r0 : sblk, below = n[i] - r0 - kb;
            if (below <= 0) continue;
            M[m] = L[i] + (r0 + kb) * lda[i] + r0;
            nr[m] = below;
            nc[m] = kb;
            ld[m] = lda[i];
            x[m] = X[i] + r0;
            y[m] = B[i] + r0 + kb;
            ++m;
        }
        rc = gemv_batch_t(m, M, nr, nc, ld, x, y, -1.0, 1.0, 0, st);
        if (rc != SSA_OK) return rc;
    }
    for (int64_t s = 0; s < maxblk; ++s) {   // backward: L^T x = y   (y lives in X, x goes to B)
        int m = 0;
        for (int i = 0; i < count; ++i) {
            const int64_t k = nblk[i] - 1 - s;
            if (k < 0) continue;
            const AuxLayout al = aux_layout(n[i]);
            const int64_t r0 = k * sblk, kb = (n[i] - r0 < sblk) ? n[i] - r0 : sblk;
            M[m] = aux[i] + al.invT + inv_block_offset(r0);
            nr[m] = nc[m] = kb;
            ld[m] = SNB;
            x[m] = X[i] + r0;
            y[m] = B[i] + r0;
            ++m;
        }
        rc = gemv_batch_t(m, M, nr, nc, ld, x, y, 1.0, 0.0, 2, st);
        if (rc != SSA_OK) return rc;
        m = 0;
        for (int i = 0; i < count; ++i) {
            const int64_t k = nblk[i] - 1 - s;
            if (k <= 0) continue;
            const int64_t r0 = k * sblk, kb = (n[i] - r0 < sblk) ? n[i] - r0 : sblk;
            M[m] = L[i] + r0;   // rows 0 .. r0-1 of L^T, columns of this block
            nr[m] = r0;
            nc[m] = kb;
            ld[m] = lda[i];
            x[m] = B[i] + r0;
            y[m] = X[i];
            ++m;
        }
        rc = gemv_batch_t(m, M, nr, nc, ld, x, y, -1.0, 1.0, 0, st);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

template <typename T>
int potrs_padded_vec_batch(int count, const void *const *L, const int64_t *n, const int64_t *lda, const void *const *aux,
                           void *const *B, void *const *workspace, bool padded, hipStream_t st, int64_t sblk) {
    const T *Lp[kSolveBatchMax];
    const T *auxp[kSolveBatchMax];
    T *Bp[kSolveBatchMax], *Xp[kSolveBatchMax];
    int64_t np[kSolveBatchMax];
    for (int i = 0; i < count; ++i) {
        np[i] = ssa_chol_padded_n(n[i]);
        Lp[i] = static_cast<const T *>(L[i]);
        auxp[i] = static_cast<const T *>(aux[i]);
        Xp[i] = static_cast<T *>(workspace[i]);
        if (padded) {   // the caller's vector already is the padded right-hand side: solved where it is
            Bp[i] = static_cast<T *>(B[i]);
            continue;
        }
        Bp[i] = Xp[i] + np[i];
        if (hipMemcpyAsync(Bp[i], B[i], n[i] * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) return SSA_ERR_HIP;
        if (np[i] > n[i] && hipMemsetAsync(Bp[i] + n[i], 0, (np[i] - n[i]) * sizeof(T), st) != hipSuccess)
            return SSA_ERR_HIP;
    }
    const int rc = potrs_vec_batch<T>(count, Lp, np, lda, auxp, Bp, Xp, st, sblk);
    if (rc != SSA_OK || padded) return rc;
    for (int i = 0; i < count; ++i)
        if (hipMemcpyAsync(B[i], Bp[i], n[i] * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess) return SSA_ERR_HIP;
    return SSA_OK;
}
}  // namespace
}  // namespace ssa

extern "C" int ssa_chol_solve_batch(int count, const void *const *L, const int64_t *n, const int64_t *lda,
                                    const void *const *aux, void *const *B, int b_is_padded, int dtype,
                                    void *const *workspace, const size_t *workspace_bytes, void *stream) {
    return ssa_chol_solve_batch_blk(count, L, n, lda, aux, B, b_is_padded, dtype, workspace, workspace_bytes,
                                    SSA_CHOL_SOLVE_BLOCK_DEFAULT, stream);
}

extern "C" int ssa_chol_solve_batch_blk(int count, const void *const *L, const int64_t *n, const int64_t *lda,
                                        const void *const *aux, void *const *B, int b_is_padded, int dtype,
                                        void *const *workspace, const size_t *workspace_bytes, int solve_block,
                                        void *stream) {
    if (count <= 0 || !L || !n || !lda || !aux || !B || !workspace || !workspace_bytes) return SSA_ERR_INVALID_ARGUMENT;
    if (!valid_solve_block(solve_block)) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < count; ++i) {
        if (!L[i] || !aux[i] || !B[i] || n[i] <= 0 || lda[i] < n[i]) return SSA_ERR_INVALID_ARGUMENT;
        if (!workspace[i] || workspace_bytes[i] < ssa_chol_solve_workspace_bytes(n[i], 1, dtype))
            return SSA_ERR_WORKSPACE_TOO_SMALL;
    }
    for (int first = 0; first < count; first += kSolveBatchMax) {   // more solves than a launch holds: groups
        const int c = (count - first < kSolveBatchMax) ? count - first : kSolveBatchMax;
        const int rc = (dtype == SSA_F64)
                           ? potrs_padded_vec_batch<double>(c, L + first, n + first, lda + first, aux + first, B + first,
                                                            workspace + first, b_is_padded != 0, as_stream(stream), solve_block)
                           : potrs_padded_vec_batch<float>(c, L + first, n + first, lda + first, aux + first, B + first,
                                                           workspace + first, b_is_padded != 0, as_stream(stream), solve_block);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

extern "C" int ssa_chol_factor_batch(int count, void *const *A, const int64_t *n, const int64_t *lda,
                                     int32_t *const *info, void *const *aux, int dtype, void *stream) {
    return ssa_chol_factor_batch_blk(count, A, n, lda, info, aux, dtype, SSA_CHOL_SOLVE_BLOCK_DEFAULT, stream);
}

extern "C" int ssa_chol_factor_batch_blk(int count, void *const *A, const int64_t *n, const int64_t *lda,
                                         int32_t *const *info, void *const *aux, int dtype, int solve_block, void *stream) {
    if (count <= 0 || !A || !n || !lda || !info || !aux) return SSA_ERR_INVALID_ARGUMENT;
    if (!valid_solve_block(solve_block)) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < count; ++i)
        if (!A[i] || !info[i] || !aux[i] || n[i] <= 0 || lda[i] < ssa_chol_padded_n(n[i]))
            return SSA_ERR_INVALID_ARGUMENT;
    // more matrices than look-ahead lanes: groups of kMaxLanes, one after the other
    for (int first = 0; first < count; first += kMaxLanes) {
        const int c = (count - first < kMaxLanes) ? count - first : kMaxLanes;
        const int rc = (dtype == SSA_F64)
                           ? potrf_padded_batch<double>(c, A + first, n + first, lda + first, info + first,
                                                        aux + first, as_stream(stream), solve_block)
                           : potrf_padded_batch<float>(c, A + first, n + first, lda + first, info + first,
                                                       aux + first, as_stream(stream), solve_block);
        if (rc != SSA_OK) return rc;
    }
    return SSA_OK;
}

extern "C" int ssa_chol_factor(void *A, int64_t n, int64_t lda, int32_t *info, void *aux, int dtype,
                               void *stream) {
    return ssa_chol_factor_batch(1, &A, &n, &lda, &info, &aux, dtype, stream);
}

extern "C" size_t ssa_chol_solve_workspace_bytes(int64_t n, int64_t nrhs, int dtype) {
    // X, padded B, and (several right-hand sides) the partial products of the split-K / skinny GEMMs
    const size_t np = static_cast<size_t>(ssa_chol_padded_n(n));
    const size_t elems = 2 * np * static_cast<size_t>(nrhs) + (nrhs > 1 ? gemm_rhs_partial_elems(np, nrhs) : 0);
    return elems * (dtype == SSA_F64 ? 8 : 4) + 256;
}

extern "C" int ssa_chol_solve(const void *L, int64_t n, int64_t lda, const void *aux, void *B,
                              int64_t nrhs, int64_t ldb, int dtype, void *workspace,
                              size_t workspace_bytes, void *stream) {
    return ssa_chol_solve_blk(L, n, lda, aux, B, nrhs, ldb, dtype, workspace, workspace_bytes, SSA_CHOL_SOLVE_BLOCK_DEFAULT,
                              stream);
}

extern "C" int ssa_chol_solve_blk(const void *L, int64_t n, int64_t lda, const void *aux, void *B,
                                  int64_t nrhs, int64_t ldb, int dtype, void *workspace,
                                  size_t workspace_bytes, int solve_block, void *stream) {
    if (!L || !aux || !B || n <= 0 || nrhs <= 0 || lda < n || ldb < nrhs) return SSA_ERR_INVALID_ARGUMENT;
    if (!valid_solve_block(solve_block)) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < ssa_chol_solve_workspace_bytes(n, nrhs, dtype))
        return SSA_ERR_WORKSPACE_TOO_SMALL;
    if (dtype == SSA_F64)
        return potrs_padded<double>(static_cast<const double *>(L), n, lda, static_cast<const double *>(aux),
                                    static_cast<double *>(B), nrhs, ldb, static_cast<double *>(workspace),
                                    as_stream(stream), solve_block);
    return potrs_padded<float>(static_cast<const float *>(L), n, lda, static_cast<const float *>(aux),
                               static_cast<float *>(B), nrhs, ldb, static_cast<float *>(workspace),
                               as_stream(stream), solve_block);
}
