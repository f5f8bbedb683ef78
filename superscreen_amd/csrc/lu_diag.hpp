// Diagonal-block kernel of the LU route without interchanges (lu.hip, getrf_np_batch): one workgroup factors
// the 256 x 256 diagonal block D of an outer panel, D = L U (L unit lower, no pivoting), AND inverts both
// factors, WL = L^-1, WU = U^-1, so that the rest of the panel is four in-place MFMA GEMMs
// (L21 = A21 WU, U12 = WL A12).  The non-symmetric sibling of chol_diag256_kernel (chol_diag.hpp), built from
// the same pieces -- loops with small bodies (the speculative sub-panel kernel of the pivoting route is
// straight-line code that runs at instruction-fetch speed: 70 us alone, 460 us beside a trailing update),
// rotating register arrays, MFMA block products with operands straight from L2:
//
//   for s = 0..3:  [L_ss \ U_ss, WL_ss, WU_ss] = ge64_lu(D_ss)   Gaussian elimination on [D_ss | I] in registers,
//                                                            one LDS hop + one barrier per column, with the
//                                                            substitution X U_ss = I fused into the same loop
//                  L_is = D_is WU_ss                (i > s)  MFMA
//                  U_sj = WL_ss D_sj                (j > s)  MFMA
//                  D_ij -= L_is U_sj             (i, j > s)  MFMA
//   for d = 1..3:  WL_ij = -WL_ii sum_t L_it WL_tj  (i - j = d),   WU_ij = -WU_ii sum_t U_it WU_tj  (j - i = d)
//
// (The first version ran the substitutions U_ss^-1 and D_is U_ss^-1 as ten more 64-step loops per launch:
// 1.1 ms per launch beside the trailing updates, the bottleneck of the whole route.)
//
// LAPACK's partial pivoting would keep every diagonal entry iff no multiplier exceeds 1 in magnitude; the
// multipliers are the entries of L, which np_check_kernel inspects on the finished factor, so nothing is
// checked here except an exactly zero pivot (reported through `bad`).
#pragma once

#include "chol_diag.hpp"

namespace ssa {
namespace luk {

using cholk::opaque;
using cholk::SB;
using cholk::slab_gemm;
using cholk::slab_load;
using cholk::slab_store;
using cholk::slab_zero;

template <typename T>
struct LuSmem {
    T ra[2][4][16 + 2];   // row J of the D half   (slot k of residue q: column 4 (I0 + k) + q)
    T mb[2][4][16 + 2];   // row J of the eliminated identity (absolute slot i: column 4 i + q)
    T lout[64][64 + 1];   // the block on its way in, then L \ U, WL, WU in turn on their way out
};  // 36 KB (float64): small on purpose -- beside the CU-holding placeholder of the schedule (cu_hold_kernel,
    // common.hpp) the kernel must fit where a 64+ KB GEMM workgroup does not

// broadcast of lane (4 * (lane / 4) + S) inside every quad (DPP quad_perm), 64- and 32-bit payloads
template <int S>
__device__ __forceinline__ int quad_bcast_i32(int x) {
    constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);
    return __builtin_amdgcn_update_dpp(0, x, ctrl, 0xf, 0xf, true);
}
template <int S>
__device__ __forceinline__ double quad_bcast(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = quad_bcast_i32<S>(static_cast<int>(b & 0xffffffffll));
    const int hi = quad_bcast_i32<S>(static_cast<int>(b >> 32));
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
template <int S>
__device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(quad_bcast_i32<S>(__float_as_int(x)));
}

// the same with the lane as a (loop-unrolled, hence constant-folded) argument
template <typename T>
__device__ __forceinline__ T quad_bcast_s(T x, int S) {
    switch (S) {
        case 0: return quad_bcast<0>(x);
        case 1: return quad_bcast<1>(x);
        case 2: return quad_bcast<2>(x);
        default: return quad_bcast<3>(x);
    }
}

__device__ __forceinline__ double rcp_acc(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ float rcp_acc(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
}

// L \ U of the 64 x 64 block at D (overwritten), WL = L^-1 (unit lower) to WLout and WU = U^-1 (upper) to WUout,
// both as full blocks with zeros in the other triangle.  All 256 threads: thread (r, q) holds columns q + 4 i
// of row r of [D | I | I].  The third part is the row-wise substitution X U = I fused into the elimination: at
// column J the row J of U that every thread reads anyway (sm.ra) is final, x_rJ = b_rJ / u_JJ is passed round the
// quad by DPP like the multiplier, and the rest of the row of X updated with the same LDS operands -- U^-1 costs
// 16 more FMAs per column instead of a second 64-step loop.  The register array of X is rotated like the D
// half, with the finished entry of every fourth column parked in the slot that falls off the end (the published
// row is zero there: the parked values are left alone), so that after 16 rotations b[i] is column 4 i + q.
template <typename T>
__device__ __forceinline__ void ge64_lu(T *D, int ld, T *WLout, T *WUout, int ldw, LuSmem<T> &sm, bool &bad) {
    const int t = opaque(threadIdx.x), r = t >> 2, q = t & 3;
    const int lane = t & 63, wave = t >> 6;
    {   // all 16 row loads of a wave in flight at once (a rolled loop waits for every load in turn)
        T tmp[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) tmp[it] = D[(wave + 4 * it) * ld + lane];
#pragma unroll
        for (int it = 0; it < 16; ++it) sm.lout[wave + 4 * it][lane] = tmp[it];
    }
    __syncthreads();
    T a[16], m[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = sm.lout[r][q + 4 * i];
        m[i] = (q + 4 * i == r) ? T(1) : T(0);
        b[i] = m[i];
    }
    __syncthreads();
#pragma unroll 1
    for (int I0 = 0; I0 < 16; ++I0) {
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const int J = 4 * I0 + S;
            const int par = S & 1;
            if (r == J) {  // the four threads of row J publish it (D half: rotated slots, identity half: absolute)
#pragma unroll
                for (int k = 0; k < 16; ++k) sm.ra[par][q][k] = a[k];
#pragma unroll
                for (int i = 0; i < 16; ++i) sm.mb[par][q][i] = m[i];
            }
            __syncthreads();
            const T pv = sm.ra[par][S][0];
            bad = bad || (pv == T(0));
            const T rp = rcp_acc(pv);
            const T acol = quad_bcast_s<T>(a[0], S);      // this row's entry in column J of D ...
            const T bcol = quad_bcast_s<T>(b[0], S);      // ... and of the right-hand side of X U = I
            const T l = (r > J) ? acol * rp : T(0);       // multiplier
            const T x = (r <= J) ? bcol * rp : T(0);      // x_rJ (rows below J: zero, X is upper triangular)
            {   // slot 0 = column group I0: column 4 I0 + q is J iff q == S, right of J iff q > S
                const T u0 = sm.ra[par][q][0];
                const T upd = a[0] - l * u0, updb = b[0] - x * u0;
                a[0] = (q > S) ? upd : ((q == S && r > J) ? l : a[0]);
                b[0] = (q > S) ? updb : ((q == S) ? x : b[0]);
            }
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const T u = sm.ra[par][q][k];
                a[k] -= l * u;
                b[k] -= x * u;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] -= l * sm.mb[par][q][i];
        }
        sm.lout[r][4 * I0 + q] = a[0];
        const T done = b[0];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            a[k] = a[k + 1];
            b[k] = b[k + 1];
        }
        a[15] = T(0);
        b[15] = done;
    }
    // results leave through the one LDS tile in whole 512-byte rows: L \ U, then WL, then WU
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = wave + 4 * it;
        D[rr * ld + lane] = sm.lout[rr][lane];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) sm.lout[r][4 * i + q] = m[i];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = wave + 4 * it;
        WLout[rr * ldw + lane] = sm.lout[rr][lane];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) sm.lout[r][4 * i + q] = b[i];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = wave + 4 * it;
        WUout[rr * ldw + lane] = sm.lout[rr][lane];
    }
    __syncthreads();
}

// D: the 256 x 256 diagonal block (leading dimension lda), overwritten with L \ U; WL, WU: the inverses of the
// two factors (leading dimension ldw, full 256 x 256; the 64 x 64 blocks of the other triangle are NOT written:
// the caller hands in zeroed memory -- the solve-phase buffer, zeroed once per factorization); scratch:
// 6 * 64 * 64 elements; a zero pivot sets *info = -2 (the caller falls back to the pivoting route).
#ifdef LUK_TIMING
#define LUK_STAMP(i) do { if (threadIdx.x == 0) tstamp[i] = wall_clock64(); } while (0)
#else
#define LUK_STAMP(i) do { } while (0)
#endif
template <typename T>
__device__ __forceinline__ void lu_diag256_body(T *D, int lda, T *WL, T *WU, int ldw, T *scratch, int32_t *info,
                                                char *luk_smem_raw
#ifdef LUK_TIMING
                                                , long long *tstamp
#endif
) {
    LuSmem<T> &sm = *reinterpret_cast<LuSmem<T> *>(luk_smem_raw);
    const int wave = threadIdx.x >> 6;
    using acc_t = typename Mfma<T>::acc_t;
    __builtin_amdgcn_s_setprio(3);
    bool bad = false;
    auto blk = [&](T *base, int ld, int bi, int bj) { return base + (bi * ld + bj) * SB; };

    LUK_STAMP(0);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        ge64_lu<T>(blk(D, lda, s, s), lda, blk(WL, ldw, s, s), blk(WU, ldw, s, s), ldw, sm, bad);
        LUK_STAMP(1 + 3 * s);
        const int lane = opaque(threadIdx.x) & 63;
        const int nb = 3 - s;
        // L_is = D_is WU_ss (i > s): a slab needs only its own rows, in place
#pragma unroll 1
        for (int task = wave; task < nb * 4; task += 4) {
            const int i = s + 1 + task / 4, slab = task % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            T *rowp = blk(D, lda, i, s) + 16 * slab * lda;
            slab_gemm<T, false>(acc, rowp, lda, blk(WU, ldw, s, s), ldw, T(1), lane);
            slab_store<T>(acc, rowp, lda, lane);
        }
        // U_sj = WL_ss D_sj (j > s): a slab of the result needs ALL rows of D_sj, so the four slabs of a block are
        // computed (one per wave), then stored behind a barrier
#pragma unroll 1
        for (int j = s + 1; j < 4; ++j) {
            acc_t acc[4];
            slab_zero<T>(acc);
            slab_gemm<T, false>(acc, blk(WL, ldw, s, s) + 16 * wave * ldw, ldw, blk(D, lda, s, j), lda, T(1), lane);
            __syncthreads();
            slab_store<T>(acc, blk(D, lda, s, j) + 16 * wave * lda, lda, lane);
            __syncthreads();
        }
        LUK_STAMP(2 + 3 * s);
        // D_ij -= L_is U_sj
#pragma unroll 1
        for (int task = wave; task < nb * nb * 4; task += 4) {
            const int p = task / 4, slab = task % 4;
            const int i = s + 1 + p / nb, j = s + 1 + p % nb;
            acc_t acc[4];
            T *crow = blk(D, lda, i, j) + 16 * slab * lda;
            slab_load<T>(acc, crow, lda, lane);
            slab_gemm<T, false>(acc, blk(D, lda, i, s) + 16 * slab * lda, lda, blk(D, lda, s, j), lda, T(-1), lane);
            slab_store<T>(acc, crow, lda, lane);
        }
        __syncthreads();
        LUK_STAMP(3 + 3 * s);
    }
    if (bad && threadIdx.x == 0) atomicMin(info, -2);

    // off-diagonal blocks of WL = L^-1 (below) and WU = U^-1 (above the diagonal), by distance from it
    T *scrL = scratch, *scrU = scratch + 3 * SB * SB;
#pragma unroll 1
    for (int d = 1; d < 4; ++d) {
        const int npairs = 4 - d;
        const int lane = opaque(threadIdx.x) & 63;
#pragma unroll 1
        for (int task = wave; task < 2 * npairs * 4; task += 4) {
            const bool upper = task >= npairs * 4;
            const int tt0 = upper ? task - npairs * 4 : task;
            const int p = tt0 / 4, slab = tt0 % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            if (!upper) {   // S_ij = sum_{t = j}^{i - 1} L_it WL_tj,   i = j + d
                const int j = p, i = j + d;
#pragma unroll 1
                for (int tt = j; tt < i; ++tt)
                    slab_gemm<T, false>(acc, blk(D, lda, i, tt) + 16 * slab * lda, lda, blk(WL, ldw, tt, j), ldw, T(1), lane);
                slab_store<T>(acc, scrL + p * SB * SB + 16 * slab * SB, SB, lane);
            } else {        // S_ij = sum_{t = i + 1}^{j} U_it WU_tj,   j = i + d
                const int i = p, j = i + d;
#pragma unroll 1
                for (int tt = i + 1; tt <= j; ++tt)
                    slab_gemm<T, false>(acc, blk(D, lda, i, tt) + 16 * slab * lda, lda, blk(WU, ldw, tt, j), ldw, T(1), lane);
                slab_store<T>(acc, scrU + p * SB * SB + 16 * slab * SB, SB, lane);
            }
        }
        __syncthreads();
        LUK_STAMP(11 + 2 * d);
#pragma unroll 1
        for (int task = wave; task < 2 * npairs * 4; task += 4) {
            const bool upper = task >= npairs * 4;
            const int tt0 = upper ? task - npairs * 4 : task;
            const int p = tt0 / 4, slab = tt0 % 4;
            acc_t acc[4];
            slab_zero<T>(acc);
            if (!upper) {   // WL_ij = -WL_ii S_ij
                const int j = p, i = j + d;
                slab_gemm<T, false>(acc, blk(WL, ldw, i, i) + 16 * slab * ldw, ldw, scrL + p * SB * SB, SB, T(-1), lane);
                slab_store<T>(acc, blk(WL, ldw, i, j) + 16 * slab * ldw, ldw, lane);
            } else {        // WU_ij = -WU_ii S_ij
                const int i = p, j = i + d;
                slab_gemm<T, false>(acc, blk(WU, ldw, i, i) + 16 * slab * ldw, ldw, scrU + p * SB * SB, SB, T(-1), lane);
                slab_store<T>(acc, blk(WU, ldw, i, j) + 16 * slab * ldw, ldw, lane);
            }
        }
        __syncthreads();
        LUK_STAMP(12 + 2 * d);
    }
}

// Elements of `scratch` (of the matrix type) the kernel needs.  float64: the six 64 x 64 blocks of the inversion.
// float32: the block is FACTORED AND INVERTED IN FLOAT64 (as the Cholesky route's diagonal blocks, chol_diag2.hpp):
// a float64 copy of D, float64 WL and WU and the six blocks, 1.77 MB.
template <typename T>
constexpr int64_t lu_diag_scratch_elems() {
    return sizeof(T) == 8 ? 6 * 64 * 64 : 2 * (3 * 256 * 256 + 6 * 64 * 64);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void lu_diag256_kernel(T *D, int lda, T *WL, T *WU, int ldw, T *scratch,
                                                            int32_t *info
#ifdef LUK_TIMING
                                                            , long long *tstamp
#endif
) {
    extern __shared__ __attribute__((aligned(16))) char luk_smem_raw[];   // LuSmem<double> for both precisions
    if constexpr (sizeof(T) == 8) {
        lu_diag256_body<T>(D, lda, WL, WU, ldw, scratch, info, luk_smem_raw
#ifdef LUK_TIMING
                           , tstamp
#endif
        );
    } else {
        double *Dd = reinterpret_cast<double *>(scratch), *WLd = Dd + 256 * 256, *WUd = WLd + 256 * 256;
        double *scr = WUd + 256 * 256;
        const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
        // rows of 256 floats as four 256-byte pieces per wave-instruction
#pragma unroll 4
        for (int r = wave; r < 256; r += 4)
#pragma unroll
            for (int c = lane; c < 256; c += 64) Dd[r * 256 + c] = static_cast<double>(D[r * lda + c]);
        __syncthreads();
        lu_diag256_body<double>(Dd, 256, WLd, WUd, 256, scr, info, luk_smem_raw
#ifdef LUK_TIMING
                                , tstamp
#endif
        );
        __syncthreads();
        // back to float32: L \ U whole, WL on and below, WU on and above the diagonal 64-blocks (the others are the
        // caller's zeros)
#pragma unroll 4
        for (int r = wave; r < 256; r += 4) {
            const int bi = r >> 6;
#pragma unroll
            for (int c = lane; c < 256; c += 64) {
                const int bj = c >> 6;
                D[r * lda + c] = static_cast<T>(Dd[r * 256 + c]);
                if (bj <= bi) WL[r * ldw + c] = static_cast<T>(WLd[r * 256 + c]);
                if (bj >= bi) WU[r * ldw + c] = static_cast<T>(WUd[r * 256 + c]);
            }
        }
    }
}

}  // namespace luk
}  // namespace ssa
