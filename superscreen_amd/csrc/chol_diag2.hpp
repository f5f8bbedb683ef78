// Diagonal-block kernel of the blocked Cholesky, second form: one workgroup of 4 waves factors the
// 256 x 256 diagonal block D of an outer panel, D = L L^T, and inverts the factor, W = L^-1 -- the step every
// panel of a factorization has to wait for (chol.hip: diagonal-block update -> this kernel -> first block row of
// the panel), and for the last third of a factorization the step that sets its pace.
//
// The first form (chol_diag.hpp) eliminates 64 columns at a time on [D | I] with one LDS hop and one workgroup
// barrier per COLUMN (0.53 us per column alone, 1.3 us beside a trailing update) and runs its MFMA block products
// with operands fetched from L2 by four waves.  Here everything is 16 x 16 tiles in the register layout of
// v_mfma_f64_16x16x4_f64 itself:
//
//   tile layout "S":  lane (i = lane & 15, g = lane >> 4), register r  <->  element [i][col(g, r)],
//                      col(g, r) = g + 4 r for float64, 4 g + r for float32 (the two instructions number the rows of
//                      their accumulators differently: mfma_traits.hpp)
//
//   * Z = X Y^T for tiles in S layout is four MFMAs on the registers as they stand, acc = mfma(Y.r, X.r, acc):
//     the contraction index of instruction r in lane group g is g + 4 r for BOTH operands (any bijection will do
//     as long as it is the same one), and the accumulator layout of the instruction (column = lane & 15,
//     row = (lane >> 4) + 4 reg) is the S layout of the transposed product.  Every product of a Cholesky
//     factorization has this form (L_ij = D_ij W_jj^T, D_ik -= L_ij L_kj^T), and so has the inversion when it is
//     carried out on the transposed tiles of W (U_i = -(sum_t U_t L_it^T) W_ii^T, U = W^T): no shuffles, no
//     operand staging, tiles travel between waves as 2 KB register images through LDS.
//   * the 16 x 16 diagonal tiles are factored AND inverted inside one wave, Gaussian elimination on [D | I] with
//     the pivot row broadcast inside each 16-lane row by ds_bpermute (no LDS memory, no barrier): 16 steps of
//     about 170 cycles.
//   * 16 dependent steps per 64 columns instead of 64, two barriers per 16 columns instead of one per column.
//
// Structure: 4 panels of 64 columns; a wave owns the tile rows w, 7 - w, 8 + w and 15 - w of the current panel in
// registers (16 tiles); per 16-column step: diagonal tile (one wave) | barrier | L_ij = D_ij W_jj^T (every wave, its rows) |
// barrier | the rest of the panel -= L_ij L_kj^T; after four steps the tiles right of the panel are updated in
// global memory (they stay in L2).  Then W: wave w inverts the block columns w, 15 - w, 7 - w, 8 + w by forward
// substitution on transposed tiles, previous tiles of a column kept as register images in a scratch buffer.
//
// 4 waves x <= 256 registers (the diagonal tile wants its dozen lane permutes per step in flight together, the
// products their operands prefetched) and 66 KB of LDS: the workgroup fits on a CU beside ONE resident 256-register
// trailing-update workgroup, like the first form (see the remark on __launch_bounds__ there).
#pragma once

#include "common.hpp"
#include "mfma_traits.hpp"

namespace ssa {
namespace cholk2 {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRows = 4;   // tile rows per wave
constexpr int NT = 16;                          // 16 x 16 tiles of 16 x 16
constexpr int kScratchElems = (NT * (NT + 1) / 2) * 256;  // the lower tiles of the block as register images

template <typename T>
using Tile = typename Mfma<T>::acc_t;   // the four registers of a lane

// column of register r in lane group g: the row numbering of the instruction's accumulator
template <typename T>
__device__ __forceinline__ constexpr int col_of(int g, int r) { return sizeof(T) == 8 ? g + 4 * r : 4 * g + r; }

template <typename T>
struct Smem {
    T w[256];             // W_jj of the current step (register image)
    T lp[NT][256];        // column j of L, one register image per tile row
    T wdiag[NT][256];     // every W_jj (the W phase needs them again)
};

// ---- tile primitives --------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void tile_zero(Tile<T> &t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = T(0);
}
// acc += X Y^T  (all three in S layout; acc passed and returned as the MFMA accumulator)
template <typename T>
__device__ __forceinline__ Tile<T> mma_xyT(Tile<T> acc, const Tile<T> &X, const Tile<T> &Y) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = Mfma<T>::run(Y[r], X[r], acc);
    return acc;
}
template <typename T>
__device__ __forceinline__ Tile<T> zero_tile() {
    return Tile<T>{T(0), T(0), T(0), T(0)};
}

template <typename T>
__device__ __forceinline__ void image_store(T *img, const Tile<T> &t, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) img[r * 64 + lane] = t[r];
}
template <typename T>
__device__ __forceinline__ Tile<T> image_load(const T *img, int lane) {
    Tile<T> t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = img[r * 64 + lane];
    return t;
}
// S layout of the TRANSPOSE of the tile whose register image is img: element [col(g, r)][i] of the tile, which the
// image holds in lane (col(g, r), g') register r' with col(g', r') = i
template <typename T>
__device__ __forceinline__ Tile<T> image_load_transposed(const T *img, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int gp = sizeof(T) == 8 ? (i & 3) : (i >> 2), rp = sizeof(T) == 8 ? (i >> 2) : (i & 3);
    Tile<T> t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = img[rp * 64 + col_of<T>(g, r) + 16 * gp];
    return t;
}
// tile (bi, bj) of a row-major matrix, S layout.  IO: the type the matrix is stored in (the tile holds T)
template <typename T, typename IO = T>
__device__ __forceinline__ Tile<T> global_load(const IO *A, int ld, int bi, int bj, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int off = (16 * bi + i) * ld + 16 * bj;   // 32-bit element offset from the (uniform) base
    Tile<T> t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = A[off + col_of<T>(g, r)];
    return t;
}
// a diagonal tile, read from its lower triangle only (element [i][k] with k > i comes from [k][i])
template <typename T, typename IO = T>
__device__ __forceinline__ Tile<T> global_load_sym(const IO *A, int ld, int b, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int base = (16 * b) * ld + 16 * b;
    Tile<T> t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = col_of<T>(g, r);
        t[r] = A[base + ((k > i) ? k * ld + i : i * ld + k)];
    }
    return t;
}
template <typename T, typename IO = T>
__device__ __forceinline__ void global_store(IO *A, int ld, int bi, int bj, const Tile<T> &t, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int off = (16 * bi + i) * ld + 16 * bj;
#pragma unroll
    for (int r = 0; r < 4; ++r) A[off + col_of<T>(g, r)] = static_cast<IO>(t[r]);
}
// the tile whose TRANSPOSE is held in S layout: element [col(g, r)][i] = t[r]
template <typename T, typename IO = T>
__device__ __forceinline__ void global_store_transposed(IO *A, int ld, int bi, int bj, const Tile<T> &t, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int off = (16 * bi) * ld + 16 * bj + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) A[off + col_of<T>(g, r) * ld] = static_cast<IO>(t[r]);
}

__device__ __forceinline__ double bperm(double v, int byte_addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bperm(float v, int byte_addr) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}
__device__ __forceinline__ double readlane(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
// 1 / x and 1 / sqrt(x) to working precision: hardware seed + one Newton step (float64: the cubic step of common.hpp)
__device__ __forceinline__ double recip(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ float recip(float x) {
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
}
__device__ __forceinline__ double rsqrt_w(double x) { return rsqrt_f64(x); }
__device__ __forceinline__ float rsqrt_w(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
}

// Tile rows of a wave: w, 7 - w, 8 + w, 15 - w (long rows paired with short ones); the inverse map for row j.
__device__ __forceinline__ int row_of(int wave, int h) {
    return h == 0 ? wave : (h == 1 ? 7 - wave : (h == 2 ? 8 + wave : 15 - wave));
}
__device__ __forceinline__ int slot_of(int j) { return j >> 2; }
__device__ __forceinline__ int wave_of(int j) {
    const int h = j >> 2;
    return h == 0 ? j : (h == 1 ? 7 - j : (h == 2 ? j - 8 : 15 - j));
}

// k-th product of a block column of the inversion, k = u (u - 1) / 2 + v: row offset u = 1 .. 15, previous tile v < u
constexpr int kTriCount = NT * (NT - 1) / 2;
struct TriTable {
    int u[kTriCount], v[kTriCount];
    constexpr TriTable() : u(), v() {
        int k = 0;
        for (int a = 1; a < NT; ++a)
            for (int b = 0; b < a; ++b) {
                u[k] = a;
                v[k] = b;
                ++k;
            }
    }
};
constexpr TriTable kTri{};
#define kTriU kTri.u
#define kTriV kTri.v

// The block lives in `scratch` as one REGISTER IMAGE per lower tile (2 KB, [r][lane]): every access of the kernel's
// inner phases is then four 512-byte wave-instructions per tile.  (Read straight from the row-major block, the S
// layout makes every wave-instruction touch 16 rows x 32 bytes: the panel loads, the updates right of a panel and
// the inversion ran at 3-6 times their MFMA time on exactly those accesses.)
__device__ __forceinline__ int img_of(int I, int K) { return (I * (I + 1) / 2 + K) * 256; }
// image of the L tile of product k of block column c of the inversion (row c + u, column c + v).  The prefetch ring
// runs ahead of the products that exist for a column (rows past the last tile row): those requests are clamped to
// a tile INSIDE the scratch (row and column; the values are never used).
__device__ __forceinline__ int tri_img(int c, int k) {
    const int I = min(c + kTriU[k], NT - 1);
    return img_of(I, min(c + kTriV[k], I));
}

// Workgroup barrier for data handed over through LDS: the LDS stores of this wave have completed, loads and stores to
// global memory stay in flight (__syncthreads() would also wait for the acknowledgement of every global store:
// about a microsecond, twice per 16-column step).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One wave: d (symmetric positive definite tile, both triangles, S layout) -> d = L (lower triangular, zeros
// above the diagonal), w = L^-1.  Gaussian elimination without pivoting on [D | I]: the multipliers are the
// Cholesky factor up to the column scaling 1 / sqrt(pivot), the eliminated identity is the inverse of the unit
// lower factor, W = diag(1 / sqrt(pivot)) times it.  bad: some pivot was not positive.
template <typename T>
__device__ __forceinline__ void chol16inv(Tile<T> &d, Tile<T> &w, bool &bad, int lane) {
    const int i = lane & 15, g = lane >> 4;
    T m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = (col_of<T>(g, r) == i) ? T(1) : T(0);
    T myinv = T(0);
    const int row_base = (lane & 48) << 2;   // byte address of lane 0 of this lane's 16-lane row
#pragma unroll
    for (int J = 0; J < 16; ++J) {
        // lane group and register of column J
        const int gJ = sizeof(T) == 8 ? (J & 3) : (J >> 2), rJ = sizeof(T) == 8 ? (J >> 2) : (J & 3);
        // every lane permute of the step is issued before anything waits for one of them (in source order the
        // compiler waited for each pair before it issued the next: six LDS round trips per step)
        const int src = row_base + (J << 2);                      // lane J of this 16-lane row: holds row J
        const T cJ = bperm(d[rJ], (i + 16 * gJ) << 2);            // D[i][J] of this lane's row
        T pr[4], pm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // float64: register r holds columns g + 4 r, so only r >= rJ (r <= rJ) can hold a column >= J (<= J)
            const bool need_d = sizeof(T) == 8 ? (r >= rJ) : true, need_m = sizeof(T) == 8 ? (r <= rJ) : true;
            pr[r] = need_d ? bperm(d[r], src) : T(0);             // D[J][col]: the pivot row, columns >= J
            pm[r] = need_m ? bperm(m[r], src) : T(0);             // row J of the eliminated identity
        }
        const T p = readlane(d[rJ], J + 16 * gJ);                 // pivot, wave-uniform
        bad = bad || !(p > T(0));
        // The dependent chain from one pivot to the next is  1 / p  ->  multiplier  ->  update  (a dependent FP64
        // instruction costs 13-17 ns alone, 40 ns beside another kernel's MFMAs: tools/probes/prim_probe.hip);
        // 1 / sqrt(p) only scales the finished column of L and the row of W and stays off that chain.
        const T ip = recip(p);
        const T rinv = rsqrt_w(p);
        int ii = i;
        asm volatile("" : "+v"(ii));   // lane predicates are computed here, not hoisted into 80 scalar registers
        const T lJ = cJ * rinv;                                    // L[i][J]  (i >= J)
        const T f = (ii > J) ? cJ * ip : T(0);                     // D[i][J] / pivot
        myinv = (ii == J) ? rinv : myinv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (sizeof(T) == 8) {
                if (r > rJ) {
                    d[r] = __builtin_fma(-f, pr[r], d[r]);
                } else if (r == rJ) {
                    const T upd = __builtin_fma(-f, pr[r], d[r]);
                    d[r] = (g == gJ) ? ((ii >= J) ? lJ : T(0)) : ((g > gJ) ? upd : d[r]);
                }
                if (r <= rJ) m[r] = __builtin_fma(-f, pm[r], m[r]);
            } else {
                const int kap = col_of<T>(g, r);                   // per-lane column of this register
                const T upd = d[r] - f * pr[r];
                d[r] = (kap == J) ? ((ii >= J) ? lJ : T(0)) : ((kap > J) ? upd : d[r]);
                m[r] = m[r] - f * pm[r];                           // (columns > J of row J of the identity are zero)
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (col_of<T>(g, r) > i) d[r] = T(0);
        w[r] = m[r] * myinv;
    }
}

// D: the 256 x 256 diagonal block (leading dimension lda; its lower triangle is read, L overwrites it), W: the
// inverse factor (leading dimension ldw; tiles on and below the diagonal are written, the rest is left alone: the
// caller keeps it zero), scratch: kScratchElems doubles; col1: 1-based column of D[0][0] for `info`.
#ifdef CHOLK2_TIMING
#define CHOLK2_STAMP(i) do { if (threadIdx.x == 0) tstamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define CHOLK2_TIMING_ARG , long long *tstamp
#else
#define CHOLK2_STAMP(i) do { } while (0)
#define CHOLK2_TIMING_ARG
#endif
// The body is a device function: the stand-alone kernel below runs it as a launch of its own (panel chains of the
// stream schedule), the tail rounds of the factorization run it as the first workgroups of a launch whose other
// workgroups are trailing-update tiles (gemm_ops.hip, chol_tail_round_kernel).
//
// T is the type the block is FACTORED in, IO the type D and W are stored in.  The float32 route runs <double, float>:
// the block's own factorization and inversion in float64 (the trailing updates that produced D, the panel product and
// everything else stay float32).  Measured on the 20 419-unknown film of config H: backward error of the whole
// factorization 8.0e-7 -> 4.7e-7 of max|S|, error of a float32 solve against float64 2.4 times smaller
// (tools/r04/f32_panel_emulation.py: the panel through the explicit inverse is NOT what costs accuracy, the rounding
// inside the diagonal blocks is); the kernel is latency-bound, so it costs little time.
template <typename T, typename IO = T>
__device__ __forceinline__ void chol_diag256_v2_body(IO *D, int lda, IO *W, int ldw, T *scratch, int32_t *info, int col1,
                                                     char *cholk2_smem_raw CHOLK2_TIMING_ARG, T *trace = nullptr) {
    Smem<T> &sm = *reinterpret_cast<Smem<T> *>(cholk2_smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __builtin_amdgcn_s_setprio(3);
    bool bad = false;
    CHOLK2_STAMP(0);
    // ---- the lower triangle as register images (the only strided reads of the kernel; a diagonal tile is read from
    // its lower half)
#pragma unroll 1
    for (int h = 0; h < kRows; ++h) {
        const int row = row_of(wave, h);
#pragma unroll 1
        for (int K = 0; K <= row; K += 4) {   // four tiles in flight
            Tile<T> t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int Kq = (K + q <= row) ? K + q : row;
                t[q] = (Kq == row) ? global_load_sym<T, IO>(D, lda, row, lane) : global_load<T, IO>(D, lda, row, Kq, lane);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (K + q <= row) image_store<T>(scratch + img_of(row, K + q), t[q], lane);
            if (trace != nullptr) {   // (debugging: what this launch read, kept apart from the working copy)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (K + q <= row) image_store<T>(trace + img_of(row, K + q), t[q], lane);
            }
        }
    }
    // (a wave reads back only images it wrote itself until the first barrier below)
    CHOLK2_STAMP(1);
    // tile rows of this wave: row_of(wave, 0..3) = w, 7 - w, 8 + w, 15 - w (long rows with short ones).  P[h][0] is
    // always the column of the current step: the columns are rotated left after every step, so that the step loop has
    // ONE body (one copy of the unrolled diagonal-tile code; straight-line code that is executed once runs at
    // instruction-fetch speed).
    Tile<T> P[kRows][4];

#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int c0 = 4 * s;   // first tile column of the panel
        // ---- load the panel: tiles (row, c0 + c), c0 + c <= row
#pragma unroll
        for (int h = 0; h < kRows; ++h) {
            const int row = row_of(wave, h);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (row >= c0 && c0 + c <= row) P[h][c] = image_load<T>(scratch + img_of(row, c0 + c), lane);
                else tile_zero<T>(P[h][c]);
            }
        }
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
            const int j = c0 + jj;                 // global tile column of this step
            const int hj = slot_of(j), wj = wave_of(j);   // slot / owner wave of tile row j
            if (wave == wj) {                      // diagonal tile: factor and invert
                Tile<T> dt, wt;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dt[r] = hj == 0 ? P[0][0][r] : (hj == 1 ? P[1][0][r] : (hj == 2 ? P[2][0][r] : P[3][0][r]));
#ifdef CHOLK2_TIMING
                const long long tb0 = __builtin_amdgcn_s_memtime();
#endif
                chol16inv<T>(dt, wt, bad, lane);
#ifdef CHOLK2_TIMING
                if (lane == 0) tstamp[16 + j] = __builtin_amdgcn_s_memtime() - tb0;
#endif
                global_store<T, IO>(D, lda, j, j, dt, lane);
                image_store<T>(sm.w, wt, lane);
                image_store<T>(sm.wdiag[j], wt, lane);
            }
            lds_barrier();
            {   // L_ij = D_ij W_jj^T for this wave's rows below j
                const Tile<T> wjj = image_load<T>(sm.w, lane);
#pragma unroll
                for (int h = 0; h < kRows; ++h) {
                    const int row = row_of(wave, h);
                    if (row > j) {
                        P[h][0] = mma_xyT<T>(zero_tile<T>(), P[h][0], wjj);
                        image_store<T>(sm.lp[row], P[h][0], lane);
                        image_store<T>(scratch + img_of(row, j), P[h][0], lane);
                        global_store<T, IO>(D, lda, row, j, P[h][0], lane);
                    }
                }
            }
            lds_barrier();
            // the rest of the panel: D_ic -= L_ij L_cj^T, c = j + 1 .. c0 + 3  (tile (j + 1, j + 1) first: the next
            // diagonal tile)
#pragma unroll
            for (int cc = 1; cc < 4; ++cc) {
                const int c = j + cc;
                if (jj + cc < 4) {
                    const Tile<T> nlc = -image_load<T>(sm.lp[c], lane);
#pragma unroll
                    for (int h = 0; h < kRows; ++h) {
                        const int row = row_of(wave, h);
                        if (row >= c) P[h][cc] = mma_xyT<T>(P[h][cc], P[h][0], nlc);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < kRows; ++h) {   // rotate: the next column becomes column 0
                P[h][0] = P[h][1];
                P[h][1] = P[h][2];
                P[h][2] = P[h][3];
            }
        }
        // ---- tiles right of the panel: D_IK -= sum_c L_Ic L_Kc^T, K = c0 + 4 .. I, on the register images (L2)
        __syncthreads();   // every L image of the panel is in memory
        CHOLK2_STAMP(2 + 3 * s);
#pragma unroll 1
        for (int h = 0; h < kRows; ++h) {
            const int row = row_of(wave, h);
            if (row < c0 + 4) continue;
            Tile<T> n[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) n[c] = -image_load<T>(scratch + img_of(row, c0 + c), lane);
            // Software pipeline over three register sets: the operands of the next two tiles are in flight while a
            // tile is computed; no register copies between rounds (a copy of a register that is being loaded is a wait
            // for the load).  Rounds past the last tile repeat it: same values to the same addresses.
            const int K0 = c0 + 4;
            auto trail_load = [&](int K, Tile<T> (&l)[4], Tile<T> &dk) {
                const int Kc = K < row ? K : row;
#pragma unroll
                for (int c = 0; c < 4; ++c) l[c] = image_load<T>(scratch + img_of(Kc, c0 + c), lane);
                dk = image_load<T>(scratch + img_of(row, Kc), lane);
            };
            auto trail_tile = [&](int K, const Tile<T> (&l)[4], const Tile<T> &dk) {
                Tile<T> acc0 = zero_tile<T>(), acc1 = zero_tile<T>();
                acc0 = mma_xyT<T>(acc0, n[0], l[0]);
                acc1 = mma_xyT<T>(acc1, n[1], l[1]);
                acc0 = mma_xyT<T>(acc0, n[2], l[2]);
                acc1 = mma_xyT<T>(acc1, n[3], l[3]);
                const Tile<T> res = dk + (acc0 + acc1);
                image_store<T>(scratch + img_of(row, K < row ? K : row), res, lane);
            };
            Tile<T> la[4], lb[4], lc[4], da, db, dc;
            trail_load(K0, la, da);
            trail_load(K0 + 1, lb, db);
#pragma unroll 1
            for (int K = K0; K <= row; K += 3) {
                trail_load(K + 2, lc, dc);
                trail_tile(K, la, da);
                trail_load(K + 3, la, da);
                trail_tile(K + 1, lb, db);
                trail_load(K + 4, lb, db);
                trail_tile(K + 2, lc, dc);
            }
        }
        __syncthreads();
        CHOLK2_STAMP(3 + 3 * s);
    }
    if (bad && lane == 0 && *info == 0) *info = col1;

    // ---- W = L^-1 by block columns on transposed tiles: U_c = W_cc^T, U_i = -(sum_{t=c}^{i-1} U_t L_it^T) W_ii^T.
    // The tiles of a column stay in registers (U[v] = U_{c + v}: the triangular loop nest is unrolled over the
    // OFFSETS u = i - c, v = t - c, so that every register index is a constant; the column c is a run-time value);
    // a column only reads L (global, final) and the W_ii (LDS): no stores to wait for, no barriers.
#pragma unroll 1
    for (int pass = 0; pass < kRows; ++pass) {
        const int c = row_of(wave, pass == 0 ? 0 : (pass == 1 ? 3 : (pass == 2 ? 1 : 2)));   // w, 15 - w, 7 - w, 8 + w
        Tile<T> U[NT];
        U[0] = image_load_transposed<T>(sm.wdiag[c], lane);                     // U_c
        const Tile<T> wcc = image_load<T>(sm.wdiag[c], lane);
        // the products of the column as one stream k = u (u - 1) / 2 + v; the L tile of product k + 8 is requested when
        // product k starts
        constexpr int kRing = 8;
        Tile<T> Q[kRing];
#pragma unroll
        for (int k = 0; k < kRing; ++k) Q[k] = image_load<T>(scratch + tri_img(c, k), lane);
#pragma unroll
        for (int u = 1; u < NT; ++u) {
            if (c + u < NT) {
                Tile<T> acc0 = zero_tile<T>(), acc1 = zero_tile<T>();
#pragma unroll
                for (int v = 0; v < u; ++v) {
                    const int k = u * (u - 1) / 2 + v;
                    const Tile<T> l = Q[k % kRing];
                    if (k + kRing < kTriCount)
                        Q[k % kRing] = image_load<T>(scratch + tri_img(c, k + kRing), lane);
                    if (v & 1) acc1 = mma_xyT<T>(acc1, U[v], l);
                    else acc0 = mma_xyT<T>(acc0, U[v], l);
                }
                const Tile<T> wii = image_load<T>(sm.wdiag[c + u], lane);
                const Tile<T> nacc = -(acc0 + acc1);
                U[u] = mma_xyT<T>(zero_tile<T>(), nacc, wii);
            }
        }
        global_store<T, IO>(W, ldw, c, c, wcc, lane);                               // W_cc as it is
#pragma unroll
        for (int u = 1; u < NT; ++u)
            if (c + u < NT) global_store_transposed<T, IO>(W, ldw, c + u, c, U[u], lane);
        if (wave == 0) CHOLK2_STAMP(13 + (pass & 1));
    }
    __syncthreads();
    CHOLK2_STAMP(15);
}

#ifdef CHOLK2_TIMING
#define CHOLK2_TIMING_PASS , tstamp
#else
#define CHOLK2_TIMING_PASS
#endif
// type a block stored as IO is factored in
template <typename IO>
struct FactorIn { using type = double; };
template <typename IO>
using factor_t = typename FactorIn<IO>::type;

template <typename IO>
__global__ __launch_bounds__(kThreads, 2) void chol_diag256_v2_kernel(IO *D, int lda, IO *W, int ldw, IO *scratch,
                                                                     int32_t *info, int col1 CHOLK2_TIMING_ARG) {
    extern __shared__ __attribute__((aligned(16))) char cholk2_smem_raw[];
    chol_diag256_v2_body<factor_t<IO>, IO>(D, lda, W, ldw, reinterpret_cast<factor_t<IO> *>(scratch), info, col1,
                                           cholk2_smem_raw CHOLK2_TIMING_PASS);
}

}  // namespace cholk2
}  // namespace ssa
