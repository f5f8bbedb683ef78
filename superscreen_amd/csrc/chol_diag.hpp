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
// rotated overrun) and row J of the eliminated identity (2 x 4 x 16).  A wave reads 4 distinct
// addresses per instruction (one per residue q), so the q stride must not be a multiple of the
// 256-byte bank period: +2 elements of padding (unpadded, the 4-way conflict cost 2x kernel time).
template <typename T>
struct Ge64Smem {
    T cb[2][4][32 + 2];
    T mb[2][4][16 + 2];
    T lout[64][64 + 1];  // the block on its way in, then the finished columns of L and W_ss in turn on their way out
};  // 36 KB (float64): small on purpose -- beside the CU-holding placeholder of the schedule (cu_hold_kernel,
    // common.hpp) the kernel must fit where a 64+ KB GEMM workgroup does not

// [L, W] of the 64 x 64 block at D (leading dimension ld); L overwrites the lower triangle of D,
// W (full block, zero above the diagonal) goes to Wout.  All 256 threads.
__device__ __forceinline__ int opaque(int x) {
    // Hides the value from loop-invariant code motion: without it the compiler hoists every
    // per-lane constant of every phase (identity columns, row offsets) to the top of the kernel and
    // then spills them, because the kernel is capped at 256 registers (see below).
    asm volatile("" : "+v"(x));
    return x;
}

template <typename T>
__device__ __forceinline__ void ge64(T *D, int ld, T *Wout, int ldw, Ge64Smem<T> &sm, bool &bad) {
    const int t = opaque(threadIdx.x), r = t >> 2, q = t & 3;
    const int lane = t & 63, wave = t >> 6;
    // Global memory is touched only in whole 512-byte rows, before and after the column loop, through
    // the LDS tiles: partial-line accesses (this thread layout gives 32-byte pieces) and stores
    // inside the loop both put memory acknowledgements on the critical path, which a trailing
    // update streaming through the same L2 / HBM stretches to microseconds.
    {   // all 16 row loads of a wave in flight at once: as a rolled loop the compiler waits for every load
        // before it issues the next (16 dependent L2 / HBM round trips per block: 8 us alone, 40 us and more
        // beside a trailing update that keeps the memory system busy)
        T tmp[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) tmp[it] = D[(wave + 4 * it) * ld + lane];
#pragma unroll
        for (int it = 0; it < 16; ++it) sm.lout[wave + 4 * it][lane] = tmp[it];
    }
    for (int e = t; e < 2 * 4 * (32 + 2); e += 256) (&sm.cb[0][0][0])[e] = T(0);
    __syncthreads();
    T a[16], m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = sm.lout[r][q + 4 * i];
        m[i] = (q + 4 * i == r) ? T(1) : T(0);
    }
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
#ifndef GE_NO_A
#pragma unroll
            for (int k = 1; k < 16; ++k) a[k] -= gmul * sm.cb[par][q][I0 + k];
#endif
#ifndef GE_NO_M
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] -= f * sm.mb[par][q][i];
#endif
        }
        sm.lout[r][4 * I0 + q] = a[0];
#pragma unroll
        for (int k = 0; k < 15; ++k) a[k] = a[k + 1];
        a[15] = T(0);
    }
    // results leave through the one LDS tile in whole 512-byte rows: L, then W
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = wave + 4 * it;
        if (lane <= rr) D[rr * ld + lane] = sm.lout[rr][lane];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) sm.lout[r][4 * i + q] = m[i] * myinv;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = wave + 4 * it;
        Wout[rr * ldw + lane] = sm.lout[rr][lane];
    }
}

// acc[jt] += sign * A_slab * op(B) for one 16-row slab (rows given by Arow, 64 columns = K) and the
// four 16-column tiles of a 64 x 64 block B.  BT: op(B) = B^T (B[j][k] row-major), else B[k][j].
template <typename T, bool BT>
__device__ __forceinline__ void slab_gemm(typename Mfma<T>::acc_t (&acc)[4], const T *Arow, int lda,
                                          const T *B, int ldb, T sign, int lane) {
    const int i = lane & 15, g = lane >> 4;
    T a[16];
    const T *ap = Arow + (i * lda + 16 * g);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = sign * ap[kk];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        T b[16];
        if (BT) {
            const T *bp = B + ((16 * jt + i) * ldb + 16 * g);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) b[kk] = bp[kk];
        } else {
            const T *bp = B + (16 * g * ldb + 16 * jt + i);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) b[kk] = bp[kk * ldb];
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
__device__ __forceinline__ void slab_load(typename Mfma<T>::acc_t (&acc)[4], const T *Crow, int ldc, int lane) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc[jt][e] = Crow[Mfma<T>::row(lane, e) * ldc + 16 * jt + (lane & 15)];
}
template <typename T>
__device__ __forceinline__ void slab_store(const typename Mfma<T>::acc_t (&acc)[4], T *Crow, int ldc, int lane) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Crow[Mfma<T>::row(lane, e) * ldc + 16 * jt + (lane & 15)] = acc[jt][e];
}

// D: the 256 x 256 diagonal block (leading dimension lda), W: its inverse factor (ldw), both in
// global memory (all offsets inside the blocks are 32-bit: 256 * ld elements must fit, the host
// checks); `scratch`: 3 * 64 * 64 elements; `col1`: 1-based column of D[0][0] for `info`.
// __launch_bounds__(256, 2) caps the kernel at 256 registers per lane: the trailing update that
// runs beside it (look-ahead) keeps two 256-register workgroups on every CU, and a workgroup that
// needs more than the 256 registers one of them frees is never placed until the whole update has
// drained (measured: 386 registers -> the kernel "took" as long as the SYRK it was meant to hide
// behind).
#ifdef CHOLK_TIMING
#define CHOLK_STAMP(i) do { if (threadIdx.x == 0) tstamp[i] = wall_clock64(); } while (0)
#else
#define CHOLK_STAMP(i) do { } while (0)
#endif
template <typename T>
__global__ __launch_bounds__(256, 2) void chol_diag256_kernel(T *D, int lda, T *W, int ldw, T *scratch,
                                                           int32_t *info, int col1
#ifdef CHOLK_TIMING
                                                           , long long *tstamp
#endif
) {
    extern __shared__ __attribute__((aligned(16))) char cholk_smem_raw[];  // sizeof(Ge64Smem<T>)
    Ge64Smem<T> &sm = *reinterpret_cast<Ge64Smem<T> *>(cholk_smem_raw);
    const int wave = threadIdx.x >> 6;
    using acc_t = typename Mfma<T>::acc_t;
    // a chain of short dependent steps that shares its SIMDs with the trailing update's MFMA waves
    __builtin_amdgcn_s_setprio(3);
    bool bad = false;
    auto blk = [&](T *base, int ld, int bi, int bj) { return base + (bi * ld + bj) * SB; };

    CHOLK_STAMP(0);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        ge64<T>(blk(D, lda, s, s), lda, blk(W, ldw, s, s), ldw, sm, bad);
        __syncthreads();
        const int lane = opaque(threadIdx.x) & 63;
        CHOLK_STAMP(1 + 3 * s);
        // L_is = D_is W_ss^T
#pragma unroll 1
        for (int task = wave; task < (3 - s) * 4; task += 4) {
            const int i = s + 1 + task / 4, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            T *rowp = blk(D, lda, i, s) + 16 * slab * lda;
            slab_gemm<T, true>(acc, rowp, lda, blk(W, ldw, s, s), ldw, T(1), lane);
            slab_store<T>(acc, rowp, lda, lane);
        }
        __syncthreads();
        CHOLK_STAMP(2 + 3 * s);
        // D_ij -= L_is L_js^T
        const int npairs = (3 - s) * (4 - s) / 2;
#pragma unroll 1
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
            T *crow = blk(D, lda, i, j) + 16 * slab * lda;
            slab_load<T>(acc, crow, lda, lane);
            slab_gemm<T, true>(acc, blk(D, lda, i, s) + 16 * slab * lda, lda,
                               blk(D, lda, j, s), lda, T(-1), lane);
            slab_store<T>(acc, crow, lda, lane);
        }
        __syncthreads();
        CHOLK_STAMP(3 + 3 * s);
    }
    if (bad && threadIdx.x == 0 && *info == 0) *info = col1;

    // off-diagonal blocks of W = L^-1, by distance from the diagonal
#pragma unroll 1
    for (int d = 1; d < 4; ++d) {
        const int npairs = 4 - d;
        const int lane = opaque(threadIdx.x) & 63;
#pragma unroll 1
        for (int task = wave; task < npairs * 4; task += 4) {  // S_ij = sum_t L_it W_tj
            const int j = task / 4, i = j + d, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
#pragma unroll 1
            for (int tt = j; tt < i; ++tt)
                slab_gemm<T, false>(acc, blk(D, lda, i, tt) + 16 * slab * lda, lda,
                                    blk(W, ldw, tt, j), ldw, T(1), lane);
            slab_store<T>(acc, scratch + j * SB * SB + 16 * slab * SB, SB, lane);
        }
        __syncthreads();
        CHOLK_STAMP(11 + 2 * d);
#pragma unroll 1
        for (int task = wave; task < npairs * 4; task += 4) {  // W_ij = -W_ii S_ij
            const int j = task / 4, i = j + d, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            slab_gemm<T, false>(acc, blk(W, ldw, i, i) + 16 * slab * ldw, ldw,
                                scratch + j * SB * SB, SB, T(-1), lane);
            slab_store<T>(acc, blk(W, ldw, i, j) + 16 * slab * ldw, ldw, lane);
        }
        __syncthreads();
        CHOLK_STAMP(12 + 2 * d);
    }
}

}  // namespace cholk
}  // namespace ssa
