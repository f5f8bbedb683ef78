// Diagonal-block kernel of the blocked Cholesky (chol.hip): one workgroup factors the 256 x 256
// diagonal block D of an outer panel, D = L L^T, AND inverts the factor, W = L^-1.
//
// With W in hand the rest of the panel is one MFMA GEMM, L21 = A21 W^T, so an outer panel costs
// three launches (this kernel + two GEMMs) instead of four (sub-panel kernel + in-panel update)
// pairs.  W is also exactly the leaf the triangular solves want (inverted diagonal blocks), so it is
// written straight into the solve-phase `aux` buffer.
//
// The kernel is built from loops with small bodies on purpose.  The first version of the panel
// factorization was fully unrolled straight-line code (17k instructions, ~120 KB): executed once
// per launch it ran at instruction-fetch speed (the 64 KB instruction cache never hit), 3-4x
// slower again when a trailing update was streaming through the same L2.
//
// Structure (64 x 64 blocks, right-looking):
//   for s = 0..3:  [L_ss, W_ss] = ge64(D_ss)                    all 256 threads, registers
//                  L_is = D_is W_ss^T            (i > s)        MFMA, one 16-row slab per wave
//                  D_ij -= L_is L_js^T           (s < j <= i)   MFMA
//   for d = 1..3:  S_ij = sum_{t=j}^{i-1} L_it W_tj,  W_ij = -W_ii S_ij     (i - j = d)    MFMA
//
// ge64: Gaussian elimination without pivoting on [D_ss | I] (thread (r, q) holds columns q + 4 i of
// row r of both halves).  For a symmetric positive definite block the multipliers are the
// Cholesky factor up to the column scaling 1/sqrt(d_J), and the eliminated identity is
// L~^-1 (unit lower), so W_ss = diag(1/sqrt(d)) L~^-1.  One LDS hop and one barrier per column.
// The register array of the D half is rotated by one slot every four columns so that the loop
// body has compile-time register indices (no scratch) and no dependence on the column number.
//
// MFMA operand trick: v_mfma_*_16x16x4 wants A[i][k] in lane (i = lane & 15, k = lane >> 4).  Any
// permutation of k that is applied to both operands leaves the product unchanged, so lane group
// g = lane >> 4 takes k = 16 g + kk for the kk-th instruction: each lane then owns 16 CONSECUTIVE
// elements of one row, which are plain 16-byte global loads (the blocks are L2 resident).
#pragma once

#include "common.hpp"
#include "mfma_traits.hpp"

namespace ssa {
namespace cholk {

constexpr int DB = 256;  // diagonal block
constexpr int SB = 64;   // sub-block

__device__ __forceinline__ double rsqrt_acc(double x) { return rsqrt_f64(x); }
__device__ __forceinline__ float rsqrt_acc(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
}

// LDS of ge64: column J of D (2 parities x 4 residues x 32 slots, slots 16..31 stay zero for the
// rotated overrun) and row J of the eliminated identity (2 x 4 x 16).
template <typename T>
struct Ge64Smem {
    T cb[2][4][32];
    T mb[2][4][16];
};

// [L, W] of the 64 x 64 block at D (leading dimension ld); L overwrites the lower triangle of D,
// W (full block, zero above the diagonal) goes to Wout.  All 256 threads.
template <typename T>
__device__ __forceinline__ void ge64(T *D, int64_t ld, T *Wout, int64_t ldw, Ge64Smem<T> &sm, bool &bad) {
    const int t = threadIdx.x, r = t >> 2, q = t & 3;
    T a[16], m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = D[static_cast<int64_t>(r) * ld + q + 4 * i];
        m[i] = (q + 4 * i == r) ? T(1) : T(0);
    }
    for (int e = t; e < 2 * 4 * 32; e += 256) (&sm.cb[0][0][0])[e] = T(0);
    __syncthreads();
    T myinv = T(0);
#pragma unroll 1
    for (int I0 = 0; I0 < 16; ++I0) {
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const int J = 4 * I0 + S;
            const int par = S & 1;
            if (q == S) sm.cb[par][r & 3][r >> 2] = a[0];  // column J (rows < J: stale, never read)
            if (r == J) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sm.mb[par][q][i] = m[i];
            }
            __syncthreads();
            const T d = sm.cb[par][S][I0];
            bad = bad || !(d > T(0));
            const T inv = rsqrt_acc(d);
            const T own = sm.cb[par][r & 3][r >> 2];
            const T lr = (r >= J) ? own * inv : T(0);  // l_rJ
            const T gmul = lr * inv;                   // a_rJ / d_J
            const T f = (r > J) ? gmul : T(0);
            myinv = (r == J) ? inv : myinv;
            {   // slot 0 = column group I0: column c = q + 4 I0 is J iff q == S, right of J iff q > S
                const T upd = a[0] - gmul * sm.cb[par][q][I0];
                a[0] = (q == S) ? ((r >= J) ? lr : a[0]) : ((q > S) ? upd : a[0]);
            }
#pragma unroll
            for (int k = 1; k < 16; ++k) a[k] -= gmul * sm.cb[par][q][I0 + k];
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] -= f * sm.mb[par][q][i];
        }
        const int c = 4 * I0 + q;
        if (c <= r) D[static_cast<int64_t>(r) * ld + c] = a[0];
#pragma unroll
        for (int k = 0; k < 15; ++k) a[k] = a[k + 1];
        a[15] = T(0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Wout[static_cast<int64_t>(r) * ldw + q + 4 * i] = m[i] * myinv;
}

// acc[jt] += sign * A_slab * op(B) for one 16-row slab (rows given by Arow, 64 columns = K) and the
// four 16-column tiles of a 64 x 64 block B.  BT: op(B) = B^T (B[j][k] row-major), else B[k][j].
template <typename T, bool BT>
__device__ __forceinline__ void slab_gemm(typename Mfma<T>::acc_t (&acc)[4], const T *Arow, int64_t lda,
                                          const T *B, int64_t ldb, T sign, int lane) {
    const int i = lane & 15, g = lane >> 4;
    T a[16];
    const T *ap = Arow + static_cast<int64_t>(i) * lda + 16 * g;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = sign * ap[kk];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        T b[16];
        if (BT) {
            const T *bp = B + static_cast<int64_t>(16 * jt + i) * ldb + 16 * g;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) b[kk] = bp[kk];
        } else {
            const T *bp = B + static_cast<int64_t>(16 * g) * ldb + 16 * jt + i;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) b[kk] = bp[static_cast<int64_t>(kk) * ldb];
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc[jt] = Mfma<T>::run(a[kk], b[kk], acc[jt]);
    }
}

template <typename T>
__device__ __forceinline__ void slab_zero(typename Mfma<T>::acc_t (&acc)[4]) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jt][e] = T(0);
}
template <typename T>
__device__ __forceinline__ void slab_load(typename Mfma<T>::acc_t (&acc)[4], const T *Crow, int64_t ldc, int lane) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc[jt][e] = Crow[static_cast<int64_t>(Mfma<T>::row(lane, e)) * ldc + 16 * jt + (lane & 15)];
}
template <typename T>
__device__ __forceinline__ void slab_store(const typename Mfma<T>::acc_t (&acc)[4], T *Crow, int64_t ldc, int lane) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Crow[static_cast<int64_t>(Mfma<T>::row(lane, e)) * ldc + 16 * jt + (lane & 15)] = acc[jt][e];
}

// D: the 256 x 256 diagonal block (leading dimension lda), W: its inverse factor (ldw), both in
// global memory; `scratch`: 3 * 64 * 64 elements; `col1`: 1-based column of D[0][0] for `info`.
template <typename T>
__global__ __launch_bounds__(256) void chol_diag256_kernel(T *D, int64_t lda, T *W, int64_t ldw, T *scratch,
                                                           int32_t *info, int col1) {
    __shared__ Ge64Smem<T> sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using acc_t = typename Mfma<T>::acc_t;
    // a chain of short dependent steps that shares its SIMDs with the trailing update's MFMA waves
    __builtin_amdgcn_s_setprio(3);
    bool bad = false;
    auto blk = [&](T *base, int64_t ld, int bi, int bj) { return base + (static_cast<int64_t>(bi) * ld + bj) * SB; };

    for (int s = 0; s < 4; ++s) {
        ge64<T>(blk(D, lda, s, s), lda, blk(W, ldw, s, s), ldw, sm, bad);
        __syncthreads();
        // L_is = D_is W_ss^T
        for (int task = wave; task < (3 - s) * 4; task += 4) {
            const int i = s + 1 + task / 4, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            T *rowp = blk(D, lda, i, s) + static_cast<int64_t>(16 * slab) * lda;
            slab_gemm<T, true>(acc, rowp, lda, blk(W, ldw, s, s), ldw, T(1), lane);
            slab_store<T>(acc, rowp, lda, lane);
        }
        __syncthreads();
        // D_ij -= L_is L_js^T
        const int npairs = (3 - s) * (4 - s) / 2;
        for (int task = wave; task < npairs * 4; task += 4) {
            int p = task / 4;
            const int slab = task % 4;
            int i = s + 1;
            while (p > i - (s + 1)) {  // pairs in row i: j = s+1 .. i
                p -= i - s;
                ++i;
            }
            const int j = s + 1 + p;
            acc_t acc[4];
            T *crow = blk(D, lda, i, j) + static_cast<int64_t>(16 * slab) * lda;
            slab_load<T>(acc, crow, lda, lane);
            slab_gemm<T, true>(acc, blk(D, lda, i, s) + static_cast<int64_t>(16 * slab) * lda, lda,
                               blk(D, lda, j, s), lda, T(-1), lane);
            slab_store<T>(acc, crow, lda, lane);
        }
        __syncthreads();
    }
    if (bad && threadIdx.x == 0 && *info == 0) *info = col1;

    // off-diagonal blocks of W = L^-1, by distance from the diagonal
    for (int d = 1; d < 4; ++d) {
        const int npairs = 4 - d;
        for (int task = wave; task < npairs * 4; task += 4) {  // S_ij = sum_t L_it W_tj
            const int j = task / 4, i = j + d, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            for (int tt = j; tt < i; ++tt)
                slab_gemm<T, false>(acc, blk(D, lda, i, tt) + static_cast<int64_t>(16 * slab) * lda, lda,
                                    blk(W, ldw, tt, j), ldw, T(1), lane);
            slab_store<T>(acc, scratch + j * SB * SB + 16 * slab * SB, SB, lane);
        }
        __syncthreads();
        for (int task = wave; task < npairs * 4; task += 4) {  // W_ij = -W_ii S_ij
            const int j = task / 4, i = j + d, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            slab_gemm<T, false>(acc, blk(W, ldw, i, i) + static_cast<int64_t>(16 * slab) * ldw, ldw,
                                scratch + j * SB * SB, SB, T(-1), lane);
            slab_store<T>(acc, blk(W, ldw, i, j) + static_cast<int64_t>(16 * slab) * ldw, ldw, lane);
        }
        __syncthreads();
    }
}

}  // namespace cholk
}  // namespace ssa
