// Speculative sub-panel factorization, second generation (included by lu.hip).
//
// Same contract as described in lu.hip ("Speculative sub-panel factorization"): the pivots of
// the 64-column sub-panel are assumed to lie in its 64 x 64 diagonal block; the assumption is
// verified exactly while the rows below are forward-substituted, and a failed check makes the
// cooperative kernel redo the sub-panel from the backup copy.  This version removes the long
// dependent LDS chains of the first one:
//   * the diagonal block is factored in REGISTERS: thread (r, q) holds the 16 entries
//     c = q + 4 i of row r; per column one LDS hop publishes the column, every wave finds the
//     pivot redundantly (no broadcast step), a second hop exchanges the two rows and hands the
//     pivot row to everybody; two barriers per column, all LDS traffic in independent batches;
//   * every row below is shared by two adjacent lanes that split the k-sum of the forward
//     substitution (even / odd k) and combine with one xor-shuffle; the k loop is unrolled
//     with independent accumulators;
//   * workgroup 0 additionally inverts the unit-lower diagonal block (inverse of L11) for the
//     block triangular solve that follows.
#pragma once

namespace ssa {

constexpr int kSpec2Rows = 128;                 // rows per workgroup (two lanes per row)
constexpr int kSpec2Stride = kSpec2Rows + 1;    // slab [PW][stride], column-major

template <typename T>
struct Spec2Args {
    T *A;
    int64_t lda;
    int64_t j0;
    int m;
    int jb;
    int32_t *ipiv;
    T *backup;            // [m][PW]
    int *spec_flag;
    int *zero_col;
    unsigned int *cnt;    // counters of the cooperative kernel, reset here
};

template <typename T>
__device__ __forceinline__ T select16(const T (&v)[16], int idx) {
    T out = v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) out = (idx == i) ? v[i] : out;
    return out;
}

template <typename T>
__global__ __launch_bounds__(256) void lu_panel_spec2_kernel(Spec2Args<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TS = PW + 1;
    T *slab = reinterpret_cast<T *>(smem_raw);       // [PW][kSpec2Stride]
    T *top = slab + PW * kSpec2Stride;               // [PW][TS] diagonal block, row-major
    T *colJ = top + PW * TS;                         // [2][PW]
    T *rowA = colJ + 2 * PW;                         // [PW] old row J
    T *rowB = rowA + PW;                             // [PW] pivot row
    int *lp = reinterpret_cast<int *>(rowB + PW);    // [PW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int row_base = g * kSpec2Rows;
    const int myrows = min(kSpec2Rows, a.m - row_base);
    const int jb = a.jb;
    const int nt = min(PW, a.m);
    T *Ap = a.A + a.j0 * a.lda + a.j0;

    if (g == 0) {
        for (int i = tid; i < kShards * 32 + 32; i += 256) a.cnt[i] = 0u;
    }
    for (int r = wave; r < myrows; r += 4) {
        if (lane < jb) {
            const T v = Ap[static_cast<int64_t>(row_base + r) * a.lda + lane];
            slab[lane * kSpec2Stride + r] = v;
            a.backup[static_cast<int64_t>(row_base + r) * PW + lane] = v;
        }
    }
    for (int r = wave; r < PW; r += 4) {
        top[r * TS + lane] = (r < nt && lane < jb) ? Ap[static_cast<int64_t>(r) * a.lda + lane] : T(0);
    }
    __syncthreads();

    // ---- diagonal block in registers -------------------------------------------------------
    const int r = tid >> 2, q = tid & 3;
    T v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = top[r * TS + q + 4 * i];
    int zero_col = 0;
    const int nsteps = min(jb, nt);
    for (int J = 0; J < nsteps; ++J) {
        T *cj = colJ + (J & 1) * PW;
        if (q == (J & 3)) cj[r] = select16(v, J >> 2);
        __syncthreads();
        double av = (lane >= J && lane < nt) ? fabs(static_cast<double>(cj[lane])) : -1.0;
        int p = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(av, off, 64);
            const int oi = __shfl_xor(p, off, 64);
            if (ov > av || (ov == av && oi < p)) { av = ov; p = oi; }
        }
        const T pv = cj[p];
        if (r == p) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rowB[q + 4 * i] = v[i];
        }
        if (r == J && p != J) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rowA[q + 4 * i] = v[i];
        }
        __syncthreads();
        if (p != J) {
            if (r == J) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = rowB[q + 4 * i];
            } else if (r == p) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = rowA[q + 4 * i];
            }
        }
        if (pv == T(0)) {
            if (zero_col == 0) zero_col = J + 1;
        } else if (r > J && r < nt) {
            const T aJ = (r == p) ? cj[J] : cj[r];  // column-J entry of the row now at position r
            const T l = aJ / pv;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = q + 4 * i;
                const T pr = rowB[c];
                if (c > J) v[i] -= l * pr;
                else if (c == J) v[i] = l;
            }
        }
        if (tid == 0) lp[J] = p;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) top[r * TS + q + 4 * i] = v[i];
    __syncthreads();

    // ---- rows below the block: forward substitution + exactness check ------------------------
    const int lr = wave * 32 + (lane >> 1);   // local row, two lanes per row
    const int half = lane & 1;
    const int prow = row_base + lr;
    const bool active = (lr < myrows && prow >= nt);
    bool viol = false;
    for (int J = 0; J < jb; ++J) {
        T p0 = T(0), p1 = T(0);
        if (active) {
            int k = half;
            for (; k + 2 < J; k += 4) {
                p0 += slab[k * kSpec2Stride + lr] * top[k * TS + J];
                p1 += slab[(k + 2) * kSpec2Stride + lr] * top[(k + 2) * TS + J];
            }
            for (; k < J; k += 2) p0 += slab[k * kSpec2Stride + lr] * top[k * TS + J];
        }
        T part = p0 + p1;
        part += __shfl_xor(part, 1, 64);
        if (active) {
            const T s = slab[J * kSpec2Stride + lr] - part;
            const T pv = top[J * TS + J];
            viol = viol || (fabs(static_cast<double>(s)) > fabs(static_cast<double>(pv)));
            if (half == 0) slab[J * kSpec2Stride + lr] = (pv != T(0)) ? s / pv : s;
        }
    }
    if (__any(viol) && lane == 0) atomicOr(a.spec_flag, 1);
    __syncthreads();

    // ---- write back -------------------------------------------------------------------------
    for (int rr = wave; rr < myrows; rr += 4) {
        const int pr = row_base + rr;
        if (lane < jb) {
            const T val = (pr < nt) ? top[pr * TS + lane] : slab[lane * kSpec2Stride + rr];
            Ap[static_cast<int64_t>(pr) * a.lda + lane] = val;
        }
    }
    if (g == 0) {
        if (tid < jb) a.ipiv[a.j0 + tid] = static_cast<int32_t>(a.j0 + (tid < nsteps ? lp[tid] : tid));
        if (tid == 0) *a.zero_col = zero_col;
    }
}

template <typename T>
constexpr size_t spec2_smem_bytes() {
    return sizeof(T) * (PW * kSpec2Stride + PW * (PW + 1) + 4 * PW) + sizeof(int) * PW + 64;
}

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
