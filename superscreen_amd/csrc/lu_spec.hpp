// Helpers of the LU route (included by lu.hip): inverse of the unit-lower 64 x 64 diagonal
// blocks of a factored panel and the block triangular solve that uses them.  (The speculative
// sub-panel kernel itself lives in lu_spec3.hpp.)
#pragma once

namespace ssa {

// ---------------------------------------------------------------------------------------
// Inverse of the unit-lower 64 x 64 diagonal block(s) of a factored panel: thread c owns
// column c of the inverse (forward substitution, columns are independent).  One workgroup
// per block (blockIdx.x); reads the block from A (below-diagonal part = L11), writes a dense
// [64][64] inverse (zero above the diagonal).  kb < 64 (last partial block) is padded with
// the identity.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void trtri_lower64_kernel(const T *__restrict__ A, int64_t lda,
                                                            int64_t block_stride, int kb_last,
                                                            int nblocks, T *__restrict__ out,
                                                            const int *__restrict__ only_if_set) {
    if (only_if_set != nullptr && *only_if_set == 0) return;  // inverse already provided
    // Register-tiled Gauss-Jordan on [L | I]: thread (r, q) holds X[r][q + 4 i]; at step k the
    // (final) row k of X is broadcast through LDS and rows r > k subtract L[r][k] times it.
    __shared__ T L[64][65];
    __shared__ T rowbuf[2][64];
    const int tid = threadIdx.x;
    const int r = tid >> 2, q = tid & 3;
    const int b = blockIdx.x;
    const int kb = (b == nblocks - 1) ? kb_last : 64;
    const T *Ab = A + static_cast<int64_t>(b) * block_stride;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int i = e >> 6, c = e & 63;
        L[i][c] = (i < kb && c < i) ? Ab[static_cast<int64_t>(i) * lda + c] : T(0);
    }
    T x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (q + 4 * i == r) ? T(1) : T(0);
    __syncthreads();
    for (int k = 0; k < 63; ++k) {
        T *rb = rowbuf[k & 1];
        if (r == k) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rb[q + 4 * i] = x[i];
        }
        __syncthreads();
        if (r > k) {
            const T l = L[r][k];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] -= l * rb[q + 4 * i];
        }
    }
    T *ob = out + static_cast<int64_t>(b) * 64 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) ob[r * 64 + q + 4 * i] = x[i];
}

// ---------------------------------------------------------------------------------------
// Block triangular solve  L X = B  (unit lower, kb <= 256) using the inverses of the 64 x 64
// diagonal blocks: X_rb = inv(L_rb,rb) (B_rb - sum_{cb<rb} L_rb,cb X_cb).  Every block product
// is a 64 x 64 x 32 register-tiled multiply out of LDS -- no per-row barriers.  One workgroup
// per 32-column strip of B.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void trsm_lower_inv_kernel(const T *__restrict__ Lmat, int64_t ldl,
                                                             const T *__restrict__ dinv,
                                                             T *__restrict__ B, int64_t ldb, int kb,
                                                             int64_t N) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int SB = 32 + 1, SL = 64 + 1;
    T *Bs = reinterpret_cast<T *>(smem_raw);   // [256][SB]
    T *Ls = Bs + 256 * SB;                     // [64][SL]
    T *Ys = Ls + 64 * SL;                      // [64][SB] staging of a block result
    const int tid = threadIdx.x;
    const int c = tid & 31, rg = tid >> 5;
    const int64_t n0 = static_cast<int64_t>(blockIdx.x) * 32;
    const bool col_ok = (n0 + c < N);
    const int nblk = (kb + 63) / 64;
    for (int r = rg; r < nblk * 64; r += 8)
        Bs[r * SB + c] = (r < kb && col_ok) ? B[static_cast<int64_t>(r) * ldb + n0 + c] : T(0);

    auto load_block = [&](const T *src, int64_t ld, int rows_valid, int cols_valid) {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int i = e >> 6, k = e & 63;
            Ls[i * SL + k] = (i < rows_valid && k < cols_valid) ? src[static_cast<int64_t>(i) * ld + k] : T(0);
        }
    };
    auto block_mul = [&](int src_row0, T (&acc)[8]) {  // acc[t] = sum_k Ls[rg+8t][k] * Bs[src_row0+k][c]
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = T(0);
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {
            const T bv = Bs[(src_row0 + k) * SB + c];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += Ls[(rg + 8 * t) * SL + k] * bv;
        }
    };
    for (int rb = 0; rb < nblk; ++rb) {
        const int rows_here = min(64, kb - rb * 64);
        for (int cb = 0; cb < rb; ++cb) {
            __syncthreads();
            load_block(Lmat + static_cast<int64_t>(rb) * 64 * ldl + cb * 64, ldl, rows_here, 64);
            __syncthreads();
            T acc[8];
            block_mul(cb * 64, acc);
#pragma unroll
            for (int t = 0; t < 8; ++t) Bs[(rb * 64 + rg + 8 * t) * SB + c] -= acc[t];
        }
        __syncthreads();
        load_block(dinv + static_cast<int64_t>(rb) * 64 * 64, 64, 64, 64);
        __syncthreads();
        T acc[8];
        block_mul(rb * 64, acc);
#pragma unroll
        for (int t = 0; t < 8; ++t) Ys[(rg + 8 * t) * SB + c] = acc[t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) Bs[(rb * 64 + rg + 8 * t) * SB + c] = Ys[(rg + 8 * t) * SB + c];
    }
    __syncthreads();
    for (int r = rg; r < kb; r += 8)
        if (col_ok) B[static_cast<int64_t>(r) * ldb + n0 + c] = Bs[r * SB + c];
}

template <typename T>
constexpr size_t trsm_inv_smem_bytes() {
    return sizeof(T) * (256 * 33 + 64 * 65 + 64 * 33);
}

}  // namespace ssa
