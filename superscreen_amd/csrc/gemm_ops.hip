// C = alpha * op(A) * op(B) + beta * C with op in {N, T}, optionally restricted to the tiles on or
// below the diagonal of a square C ("lower": the SYRK-shaped trailing update of the Cholesky
// factorization, C -= L21 * L21^T, and nothing above the diagonal is touched or needed).
//
// Same tiling and LDS-DMA pipeline as gemm.hip (128 x 128 tile, 4 waves x (4 x 4) MFMA tiles of
// v_mfma_f64_16x16x4_f64, KC = 16 stages, accumulators initialised from C).  An operand slab
// (128 rows of m or n  x  16 k) is staged in one of two LDS images, depending on which index is
// contiguous in memory:
//   K-MAJOR  (k contiguous: A for op N, B for op T)   [128][16]  unpadded, XOR-swizzled k
//   K-MINOR  (m|n contiguous: A for op T, B for op N) [16][128 + 16]
// so NT (SYRK: both operands are row panels of the same matrix) uses two K-MAJOR images and TN
// (backward substitution with L^T on many right-hand sides) two K-MINOR images.
#include "chol_diag2.hpp"
#include "chol_tail.hpp"
#include "gemm_profile.hpp"
#include "mfma_traits.hpp"

namespace ssa {

enum : int { OP_N = 0, OP_T = 1 };

constexpr int KMINOR_STRIDE = BN + 16;
// 128 x 128 tiles of a launch below which the 32 x 128 tile is used.  (Round 3, update-bound factorization: 0 / 8 / 20 /
// 40 / 64 / 80 / 160 / 320 tiles -> 104.7 / 101.6 / 103.5 / 101.4 / 102.9 / 101.4 / 102.5-104.0 / 104.0 ms over two
// boxes: the fewer of the chains' products run on the small tile, the less they cost the update.)
constexpr int64_t kSmallTileMaxTiles = 64;
constexpr int IMG_ELEMS = KC * KMINOR_STRIDE;  // the larger of the two images

struct OpSmemF64 {
    double a[2][IMG_ELEMS];
    double b[2][IMG_ELEMS];
};

// ---- FULL path (f64, interior tiles) -------------------------------------------------------
// kmajor: element (r, k) lives at X[(x0 + r) * ld + k]; else at X[k * ld + x0 + r].
template <bool KMAJOR>
struct OperandF64 {
    const double *src[4];
    int64_t ld;
    __device__ __forceinline__ void init(const double *X, int64_t ldx, int64_t x0, int lane, int wave) {
        ld = ldx;
        if (KMAJOR) {
            const int sub = lane >> 3, kpos = (lane & 7) * 2;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = 8 * (wave + 4 * u) + sub;
                src[u] = X + (x0 + row) * ldx + (kpos ^ (2 * ((row >> 1) & 7)));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) src[u] = X + static_cast<int64_t>(wave + 4 * u) * ldx + x0 + 2 * lane;
        }
    }
    __device__ __forceinline__ void issue(int64_t k0, double *img, int wave) const {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KMAJOR) glds16(src[u] + k0, img + 8 * (wave + 4 * u) * KC);
            else glds16(src[u] + k0 * ld, img + (wave + 4 * u) * KMINOR_STRIDE);
        }
    }
    // fragment of sub-tile t (16 rows) of the wave's 64-row half `w`, k-step ks
    static __device__ __forceinline__ double frag(const double *img, int w, int t, int ks, int li, int lk,
                                                  int swz) {
        if (KMAJOR) return img[(w * 64 + t * 16 + li) * KC + ((ks * 4 + lk) ^ swz)];
        return img[(ks * 4 + lk) * KMINOR_STRIDE + w * 64 + t * 16 + li];
    }
};

template <int TA, int TB>
__device__ __forceinline__ void tile_full_f64(int64_t K, double alpha, const double *__restrict__ A,
                                              int64_t lda, const double *__restrict__ B, int64_t ldb,
                                              double beta, double *__restrict__ C, int64_t ldc,
                                              int64_t m0, int64_t n0, char *smem_raw) {
    using MF = Mfma<double>;
    using acc_t = MF::acc_t;
    constexpr bool A_KMAJOR = (TA == OP_N), B_KMAJOR = (TB == OP_T);
    OpSmemF64 &sm = *reinterpret_cast<OpSmemF64 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int swz = 2 * ((li >> 1) & 7);

    OperandF64<A_KMAJOR> opa;
    OperandF64<B_KMAJOR> opb;
    opa.init(A, lda, m0, lane, wave);
    opb.init(B, ldb, n0, lane, wave);
    opa.issue(0, sm.a[0], wave);
    opb.issue(0, sm.b[0], wave);

    acc_t acc[4][4];
    double *Cw = C + (m0 + wm * 64) * ldc + n0 + wn * 64 + li;
    if (beta != 0.0) {
        const double scale = beta / alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = scale * Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t nk = K / KC;
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = static_cast<int>(kt & 1);
        if (kt + 1 < nk) {
            opa.issue((kt + 1) * KC, sm.a[cur ^ 1], wave);
            opb.issue((kt + 1) * KC, sm.b[cur ^ 1], wave);
        }
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = OperandF64<A_KMAJOR>::frag(sm.a[cur], wm, i, ks, li, lk, swz);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = OperandF64<B_KMAJOR>::frag(sm.b[cur], wn, j, ks, li, lk, swz);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MF::run(fa[i], fb[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16] = alpha * acc[i][j][r];
}

// ---- FULL path, f32, NT (both operands K-MAJOR) -----------------------------------------------
// Byte for byte the f64 pipeline: a row segment of 32 floats is the same 128 bytes as 16 doubles,
// so the operands are staged through the SAME LDS-DMA / swizzle code on 8-byte views of the
// matrices (leading dimension and K counted in float pairs).  Every 8-byte LDS read then carries
// two consecutive k of one row and feeds two v_mfma_f32_16x16x4_f32 (the k permutation this
// implies is the same for A and B, so the product is unchanged).  Same time per byte as f64,
// i.e. twice the rate per element.
__device__ __forceinline__ void tile_full_f32_nt(int64_t K, float alpha, const float *__restrict__ A,
                                                 int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                 float beta, float *__restrict__ C, int64_t ldc, int64_t m0,
                                                 int64_t n0, char *smem_raw) {
    using MF = Mfma<float>;
    using acc_t = MF::acc_t;
    OpSmemF64 &sm = *reinterpret_cast<OpSmemF64 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int swz = 2 * ((li >> 1) & 7);

    OperandF64<true> opa, opb;
    opa.init(reinterpret_cast<const double *>(A), lda / 2, m0, lane, wave);
    opb.init(reinterpret_cast<const double *>(B), ldb / 2, n0, lane, wave);
    opa.issue(0, sm.a[0], wave);
    opb.issue(0, sm.b[0], wave);

    // float32: products summed from zero, C joins in the epilogue (see gemm_tile_full_f32, gemm.hip: with the
    // accumulators started at C every product is rounded at the magnitude of C); C is loaded here all the same
    acc_t acc[4][4], cin[4][4];
    float *Cw = C + (m0 + wm * 64) * ldc + n0 + wn * 64 + li;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = acc_t{0, 0, 0, 0};
            if (beta != 0.0f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cin[i][j][r] = Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16];
            } else {
                cin[i][j] = acc_t{0, 0, 0, 0};
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int64_t nk = (K / 2) / KC;  // stages of 16 float pairs
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = static_cast<int>(kt & 1);
        if (kt + 1 < nk) {
            opa.issue((kt + 1) * KC, sm.a[cur ^ 1], wave);
            opb.issue((kt + 1) * KC, sm.b[cur ^ 1], wave);
        }
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            float2 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double v = OperandF64<true>::frag(sm.a[cur], wm, i, ks, li, lk, swz);
                fa[i] = __builtin_bit_cast(float2, v);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = OperandF64<true>::frag(sm.b[cur], wn, j, ks, li, lk, swz);
                fb[j] = __builtin_bit_cast(float2, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = MF::run(fa[i].x, fb[j].x, acc[i][j]);
                    acc[i][j] = MF::run(fa[i].y, fb[j].y, acc[i][j]);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16] =
                    __builtin_fmaf(alpha, acc[i][j][r], beta * cin[i][j][r]);
}

// ---- SMALL-TILE path, NT: 32 x 128 tile per workgroup ------------------------------------------
// The chain of the Cholesky schedule (chol.hip) runs products whose whole grid is a fraction of the chip --
// strips C[M x 256] -= P P0^T and panel products L21 = A21 W^T with M a few thousand rows: with 128 x 128
// tiles such a launch takes as long as ONE tile (4096 MFMAs of 64 cycles on the 4 SIMDs of one CU: 30 us at
// K = 256, 60 us at K = 512, plus prologue and epilogue), and in the last third of a factorization three of
// them sit on the critical path of every panel round.  Here a workgroup computes 32 rows x 128 columns (a
// quarter of the MFMA work, four times as many workgroups); 128 columns per workgroup keep the in-place panel
// products safe (a workgroup reads and writes its own rows only, N = 128 is one tile).  4 waves side by
// side, 32 x 32 each (2 x 2 MFMA tiles); operands staged by LDS-DMA like the large tile.  float32 through the
// same 8-byte views as tile_full_f32_nt.
constexpr int SBM = 32;
constexpr int SNST = 4;   // LDS ring: with 16 MFMAs per wave and stage a stage computes in 0.5 us, a load takes 1-3 us:
                          // three stages are kept in flight (one-stage prefetch made the tile latency 16 x one memory
                          // round trip = 50 us whatever its size)
struct SmallSmem {
    double a[SNST][SBM * KC];
    double b[SNST][BN * KC];
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

template <typename T>
__device__ __forceinline__ void tile_small_nt(int64_t K8, T alpha, const double *__restrict__ A, int64_t lda8,
                                              const double *__restrict__ B, int64_t ldb8, T beta, T *__restrict__ C,
                                              int64_t ldc, int64_t m0, int64_t n0, char *smem_raw) {
    // A, B: 8-byte views (float32: pairs of consecutive k); K8, lda8, ldb8 in 8-byte units
    using MF = Mfma<T>;
    using acc_t = typename MF::acc_t;
    SmallSmem &sm = *reinterpret_cast<SmallSmem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int swz = 2 * ((li >> 1) & 7);
    const int sub = lane >> 3, kpos = (lane & 7) * 2;
    // LDS-DMA sources: A image = 32 rows (one 8-row group per wave), B image = 128 rows (four groups per wave):
    // 5 loads per wave and stage
    const int arow = 8 * wave + sub;
    const double *asrc = A + (m0 + arow) * lda8 + (kpos ^ (2 * ((arow >> 1) & 7)));
    const double *bsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = 8 * (wave + 4 * u) + sub;
        bsrc[u] = B + (n0 + row) * ldb8 + (kpos ^ (2 * ((row >> 1) & 7)));
    }
    auto issue = [&](int64_t kt) {
        const int stage = static_cast<int>(kt % SNST);
        glds16(asrc + kt * KC, sm.a[stage] + 8 * wave * KC);
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16(bsrc[u] + kt * KC, sm.b[stage] + 8 * (wave + 4 * u) * KC);
    };
    // the C tile first (oldest in the vmcnt order; raw values, scaled after the first wait so that nothing
    // forces them to arrive before the stages are issued), then the first SNST - 1 stages
    // (float32: the C tile is kept apart and joins the products in the epilogue, see gemm_tile_full_f32, gemm.hip)
    constexpr bool kSplitC = sizeof(T) == 4;
    acc_t acc[2][2], cin[2][2];
    T *Cw = C + m0 * ldc + n0 + wave * 32 + li;
    const bool has_c = (beta != T(0));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = acc_t{0, 0, 0, 0};
            cin[i][j] = acc_t{0, 0, 0, 0};
            if (has_c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T v = Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16];
                    if (kSplitC) cin[i][j][r] = v;
                    else acc[i][j][r] = v;
                }
            }
        }
    const int64_t nk = K8 / KC;
#pragma unroll
    for (int p = 0; p < SNST - 1; ++p)
        if (p < nk) issue(p);
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = static_cast<int>(kt % SNST);
        if (kt + SNST - 1 < nk) issue(kt + SNST - 1);   // into the slot that was read at step kt - 1 (barrier below)
        const int64_t ahead = nk - 1 - kt;               // stages issued after stage kt: min(SNST - 1, ahead)
        if (ahead >= 3) wait_vmcnt<15>();
        else if (ahead == 2) wait_vmcnt<10>();
        else if (ahead == 1) wait_vmcnt<5>();
        else wait_vmcnt<0>();
        // raw barrier: __syncthreads() carries a fence that drains every load in flight (vmcnt(0)), which is
        // exactly what the ring must not do; each wave has waited for its own share of stage kt above
        __builtin_amdgcn_s_barrier();
        if (kt == 0 && has_c && !kSplitC) {
            const T scale = beta / alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] *= scale;
        }
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = sm.a[cur][(i * 16 + li) * KC + ((ks * 4 + lk) ^ swz)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = sm.b[cur][(wave * 32 + j * 16 + li) * KC + ((ks * 4 + lk) ^ swz)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (sizeof(T) == 8) {
                        acc[i][j] = MF::run(fa[i], fb[j], acc[i][j]);
                    } else {
                        const float2 x = __builtin_bit_cast(float2, fa[i]), y = __builtin_bit_cast(float2, fb[j]);
                        acc[i][j] = MF::run(x.x, y.x, acc[i][j]);
                        acc[i][j] = MF::run(x.y, y.y, acc[i][j]);
                    }
                }
        }
        // Every wave is done reading slot `cur` before it is refilled (the next iteration's issue() writes it): the
        // reads must have COMPLETED, not just been issued, when a wave arrives here.  s_barrier alone does not wait for
        // them -- the compiler placed it in front of the `s_waitcnt lgkmcnt(0)` of the last k-step's fragments, and
        // beside LDS-heavy workgroups of another stream (the finishing passes' transposes next to the rounds) a DMA of
        // the next stage landed in the slot before a slow wave's read of it had executed: about one factorization in a
        // hundred of the four-film stack came out different in the 10th digit (round 5, tools/chol_race_hunt.py).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cw[static_cast<int64_t>(i * 16 + MF::row(lane, r)) * ldc + j * 16] =
                    kSplitC ? alpha * acc[i][j][r] + beta * cin[i][j][r] : alpha * acc[i][j][r];
}

template <typename T>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_small_kernel(int64_t K8, T alpha, const double *__restrict__ A,
                                                                    int64_t lda8, const double *__restrict__ B,
                                                                    int64_t ldb8, T beta, T *__restrict__ C, int64_t ldc,
                                                                    int64_t ntm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // These launches are the products of the factorization's panel chains: few workgroups, each on a CU it shares
    // with the trailing update's waves.  Instruction arbitration on a SIMD goes by priority, then by age -- and the
    // update's waves are always the older ones.
    __builtin_amdgcn_s_setprio(3);
    const int64_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t tm = wg % ntm, tn = wg / ntm;   // consecutive ids share the B panel
    tile_small_nt<T>(K8, alpha, A, lda8, B, ldb8, beta, C, ldc, tm * SBM, tn * BN, smem_raw);
}

// ---- EDGE path: any type, any shape, guarded element-wise staging --------------------------
template <typename T>
struct EdgeSmem {
    static constexpr int SA = KC + Mfma<T>::APAD;
    static constexpr int SB = BN + Mfma<T>::BPAD;
    T a[BM * SA];
    T b[KC * SB];
};

template <typename T, int TA, int TB>
__device__ __forceinline__ void tile_edge(int64_t M, int64_t N, int64_t K, T alpha, const T *__restrict__ A,
                                          int64_t lda, const T *__restrict__ B, int64_t ldb, T beta,
                                          T *__restrict__ C, int64_t ldc, int64_t m0, int64_t n0,
                                          char *smem_raw) {
    using MF = Mfma<T>;
    using acc_t = typename MF::acc_t;
    EdgeSmem<T> &sm = *reinterpret_cast<EdgeSmem<T> *>(smem_raw);
    constexpr int SA = EdgeSmem<T>::SA, SB = EdgeSmem<T>::SB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t mrem = M - (m0 + wm * 64), nrem = N - (n0 + wn * 64);
    const int mt_cnt = mrem <= 0 ? 0 : (mrem >= 64 ? 4 : static_cast<int>((mrem + 15) / 16));
    const int nt_cnt = nrem <= 0 ? 0 : (nrem >= 64 ? 4 : static_cast<int>((nrem + 15) / 16));

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    for (int64_t k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
        for (int e = tid; e < BM * KC; e += kGemmThreads) {
            int r, k;
            if (TA == OP_N) { r = e / KC; k = e % KC; } else { k = e / BM; r = e % BM; }
            const int64_t gr = m0 + r, gk = k0 + k;
            T v = T(0);
            if (gr < M && gk < K) v = (TA == OP_N) ? A[gr * lda + gk] : A[gk * lda + gr];
            sm.a[r * SA + k] = v;
        }
        for (int e = tid; e < KC * BN; e += kGemmThreads) {
            int c, k;
            if (TB == OP_N) { k = e / BN; c = e % BN; } else { c = e / KC; k = e % KC; }
            const int64_t gc = n0 + c, gk = k0 + k;
            T v = T(0);
            if (gc < N && gk < K) v = (TB == OP_N) ? B[gk * ldb + gc] : B[gc * ldb + gk];
            sm.b[k * SB + c] = v;
        }
        __syncthreads();
        const T *sa = &sm.a[(wm * 64 + li) * SA + lk];
        const T *sb = &sm.b[lk * SB + wn * 64 + li];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            T fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = sa[i * 16 * SA + ks * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = sb[ks * 4 * SB + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < mt_cnt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j < nt_cnt) acc[i][j] = MF::run(fa[i], fb[j], acc[i][j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= mt_cnt) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nt_cnt) continue;
            const int64_t col = n0 + wn * 64 + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * 64 + i * 16 + MF::row(lane, r);
                if (row < M && col < N) {
                    T *dst = C + row * ldc + col;
                    T v = alpha * acc[i][j][r];
                    if (beta != T(0)) v += beta * (*dst);
                    *dst = v;
                }
            }
        }
    }
}

// Lower-triangular tile enumeration in bands of 8 tile rows, column by column inside a band: 64 consecutive
// ids (= what one XCD runs at a time) touch 8 row blocks + 8 column blocks of the panel (4 MB at K = 256, the
// size of the XCD's L2) instead of 1 + 64.  Measured with FETCH_SIZE: row-by-row order re-read the panel ~60x
// through the fabric.  id in [0, ntm (ntm + 1) / 2) -> tile (tm, tn), tn <= tm < ntm.
__device__ __forceinline__ void lower_band_tile(int64_t id, int64_t ntm, int64_t &tm, int64_t &tn) {
    constexpr int64_t G = 8;
    // band b starts at id 8b (8b + 1) / 2
    int64_t b = static_cast<int64_t>((sqrt(8.0 * static_cast<double>(id) + 1.0) - 1.0) * 0.5) / G;
    while (G * b * (G * b + 1) / 2 > id) --b;
    while (G * (b + 1) * (G * (b + 1) + 1) / 2 <= id) ++b;
    const int64_t r0 = G * b;
    const int64_t R = (ntm - r0 < G) ? ntm - r0 : G;  // tile rows in this band
    int64_t l = id - r0 * (r0 + 1) / 2;
    if (l < r0 * R) {  // full columns left of the band's diagonal blocks
        tn = l / R;
        tm = r0 + l % R;
    } else {
        l -= r0 * R;
        int64_t j = 0;
        while (j + 1 < R && l >= R - j) {  // column r0 + j holds rows r0 + j .. r0 + R - 1  (j < R: an id past the last tile ends here)
            l -= R - j;
            ++j;
        }
        tn = r0 + j;
        tm = r0 + j + l;
    }
}

// LOWER is a template parameter so that the SYRK-shaped launches carry their own symbol in
// rocprofv3's kernel statistics (profiles/), apart from the skinny in-panel updates.
template <typename T, int TA, int TB, bool LOWER>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_op_kernel(
    int64_t M, int64_t N, int64_t K, T alpha, const T *__restrict__ A, int64_t lda,
    const T *__restrict__ B, int64_t ldb, T beta, T *__restrict__ C, int64_t ldc, int64_t ntm,
    int64_t ntn, int aligned) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int64_t tm, tn;
    if constexpr (!LOWER) __builtin_amdgcn_s_setprio(2);   // strips and panel products of the chains (see gemm_nt_small_kernel)
    if constexpr (LOWER) {
        lower_band_tile(xcd_contiguous(blockIdx.x, gridDim.x), ntm, tm, tn);
    } else {
        const int64_t wg = xcd_contiguous(blockIdx.x, ntm * ntn);
        constexpr int64_t G = 8;
        const int64_t per_group = G * ntn;
        const int64_t group = wg / per_group;
        const int64_t first_m = group * G;
        const int64_t gsize = (ntm - first_m < G) ? ntm - first_m : G;
        tm = first_m + (wg % per_group) % gsize;
        tn = (wg % per_group) / gsize;
    }
    const int64_t m0 = tm * BM, n0 = tn * BN;
    constexpr bool kHasFull = (sizeof(T) == 8) || (TA == OP_N && TB == OP_T);  // f32: NT only
    constexpr int64_t kStageK = (sizeof(T) == 8) ? KC : 2 * KC;
    const bool full = kHasFull && aligned && (m0 + BM <= M) && (n0 + BN <= N) && (K % kStageK == 0) &&
                      (K > 0) && alpha != T(0);
    if (full) {
        if constexpr (sizeof(T) == 8)
            tile_full_f64<TA, TB>(K, alpha, A, lda, B, ldb, beta, C, ldc, m0, n0, smem_raw);
        else if constexpr (TA == OP_N && TB == OP_T)
            tile_full_f32_nt(K, alpha, A, lda, B, ldb, beta, C, ldc, m0, n0, smem_raw);
    } else {
        tile_edge<T, TA, TB>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, m0, n0, smem_raw);
    }
}

template <typename T, int TA, int TB, bool LOWER>
int launch_op_l(int64_t M, int64_t N, int64_t K, double alpha, const T *A, int64_t lda, const T *B,
                int64_t ldb, double beta, T *C, int64_t ldc, hipStream_t st) {
    constexpr int lower = LOWER ? 1 : 0;
    if (M <= 0 || N <= 0) return SSA_OK;
    if (lower && M != N) return SSA_ERR_INVALID_ARGUMENT;
    const int64_t ntm = ceil_div(M, BM), ntn = ceil_div(N, BN);
    const size_t smem = sizeof(OpSmemF64) > sizeof(EdgeSmem<T>) ? sizeof(OpSmemF64) : sizeof(EdgeSmem<T>);
    const int aligned = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0) &&
                        ((lda * sizeof(T)) % 16 == 0) && ((ldb * sizeof(T)) % 16 == 0);
    const int64_t nwg = lower ? ntm * (ntm + 1) / 2 : ntm * ntn;
    // few tiles (a launch of the panel chain): the small-tile kernel, a quarter of the latency
    constexpr int64_t kStage = (sizeof(T) == 8) ? KC : 2 * KC;
    if (!lower && TA == OP_N && TB == OP_T && aligned && nwg <= kSmallTileMaxTiles && M % SBM == 0 && N % BN == 0 &&
        K % kStage == 0 && K > 0 && alpha != 0.0) {
        static DeviceFlags small_flags;
        if (raise_dynamic_lds(small_flags, {{reinterpret_cast<const void *>(&gemm_nt_small_kernel<T>), sizeof(SmallSmem)}}) !=
            SSA_OK)
            return SSA_ERR_HIP;
        const int64_t stm = M / SBM, stn = N / BN;
        constexpr int64_t per8 = 8 / sizeof(T);
        ProfileScope scope(sizeof(T) == 8, kProfileOpNT, 2.0 * static_cast<double>(K) * static_cast<double>(M) * N, st);
        hipLaunchKernelGGL((gemm_nt_small_kernel<T>), dim3(static_cast<unsigned>(stm * stn)), dim3(kGemmThreads),
                           sizeof(SmallSmem), st, K / per8, static_cast<T>(alpha), reinterpret_cast<const double *>(A),
                           lda / per8, reinterpret_cast<const double *>(B), ldb / per8, static_cast<T>(beta), C, ldc, stm);
        SSA_RETURN_IF_LAUNCH_FAILED();
        return SSA_OK;
    }
    static DeviceFlags lds_flags;
    if (raise_dynamic_lds(lds_flags, {{reinterpret_cast<const void *>(&gemm_op_kernel<T, TA, TB, LOWER>), smem}}) !=
        SSA_OK)
        return SSA_ERR_HIP;
    // algorithmic flops (lower: the M (M + 1) / 2 entries on/below the diagonal)
    const double flops = 2.0 * static_cast<double>(K) * (lower ? 0.5 * static_cast<double>(M) * (M + 1)
                                                                : static_cast<double>(M) * N);
    ProfileScope scope(aligned && sizeof(T) == 8 && TA == OP_N && TB == OP_T, lower ? kProfileSyrkLower : kProfileOpNT,
                       flops, st);
    hipLaunchKernelGGL((gemm_op_kernel<T, TA, TB, LOWER>), dim3(static_cast<unsigned>(nwg)),
                       dim3(kGemmThreads), smem, st, M, N, K, static_cast<T>(alpha), A, lda, B, ldb,
                       static_cast<T>(beta), C, ldc, ntm, ntn, aligned);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}

template <typename T, int TA, int TB>
int launch_op(int64_t M, int64_t N, int64_t K, double alpha, const T *A, int64_t lda, const T *B,
              int64_t ldb, double beta, T *C, int64_t ldc, int lower, hipStream_t st) {
    if (lower) return launch_op_l<T, TA, TB, true>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
    return launch_op_l<T, TA, TB, false>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}

template <typename T>
int gemm_op(int opA, int opB, int lower, int64_t M, int64_t N, int64_t K, double alpha, const T *A,
            int64_t lda, const T *B, int64_t ldb, double beta, T *C, int64_t ldc, hipStream_t st) {
    if (opA == OP_N && opB == OP_T)
        return launch_op<T, OP_N, OP_T>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, st);
    if (opA == OP_T && opB == OP_N)
        return launch_op<T, OP_T, OP_N>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, st);
    if (opA == OP_N && opB == OP_N)
        return launch_op<T, OP_N, OP_N>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, st);
    return launch_op<T, OP_T, OP_T>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, st);
}


// ---- batched launches of the single-stream rounds of the Cholesky schedule (chol_tail.hpp) --------------------
struct SmallBatchArgs {
    int njobs;
    int64_t wg_begin[kTailMaxFilms + 1];   // first workgroup of each job
    struct Job {
        const double *A, *B, *B2;          // 8-byte views
        void *C;
        int64_t lda8, ldb8, ldc, K8, ntm;
        double alpha, beta;
        int pair;
    } j[kTailMaxFilms];
};

template <typename T>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_small_batch_kernel(SmallBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __builtin_amdgcn_s_setprio(3);   // beside the finishing passes' tiles on their low-priority stream (see gemm_nt_small_kernel)
    const int64_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
    int f = 0;
    while (f + 1 < a.njobs && wg >= a.wg_begin[f + 1]) ++f;
    const SmallBatchArgs::Job &J = a.j[f];
    const int64_t l = wg - a.wg_begin[f];
    T *C = static_cast<T *>(J.C);
    if (J.pair) {   // one workgroup per 32 rows of a panel: columns 128 .. 255 from all 256, then 0 .. 127 from the first 128
        tile_small_nt<T>(J.K8, static_cast<T>(J.alpha), J.A, J.lda8, J.B2, J.ldb8, T(0), C + BN, J.ldc, l * SBM, 0, smem_raw);
        tile_small_nt<T>(J.K8 / 2, static_cast<T>(J.alpha), J.A, J.lda8, J.B, J.ldb8, T(0), C, J.ldc, l * SBM, 0, smem_raw);
    } else {
        const int64_t tm = l % J.ntm, tn = l / J.ntm;   // consecutive ids share the B panel
        tile_small_nt<T>(J.K8, static_cast<T>(J.alpha), J.A, J.lda8, J.B, J.ldb8, static_cast<T>(J.beta), C, J.ldc, tm * SBM,
                         tn * BN, smem_raw);
    }
}

template <typename T>
int gemm_nt_small_batch(int njobs, const SmallNtJob *jobs, hipStream_t st) {
    if (njobs < 0 || njobs > kTailMaxFilms) return SSA_ERR_INVALID_ARGUMENT;
    constexpr int64_t per8 = 8 / sizeof(T);
    constexpr int64_t kStage = (sizeof(T) == 8) ? KC : 2 * KC;
    SmallBatchArgs a{};
    int64_t total = 0;
    double flops = 0.0;
    for (int i = 0; i < njobs; ++i) {
        const SmallNtJob &s = jobs[i];
        if (s.M <= 0 || s.N <= 0) continue;
        const bool aligned = reinterpret_cast<uintptr_t>(s.A) % 16 == 0 && reinterpret_cast<uintptr_t>(s.B) % 16 == 0 &&
                             (s.lda * sizeof(T)) % 16 == 0 && (s.ldb * sizeof(T)) % 16 == 0 &&
                             (!s.pair || reinterpret_cast<uintptr_t>(s.B2) % 16 == 0);
        if (!aligned || s.M % SBM != 0 || s.N % BN != 0 || s.K % kStage != 0 || s.K <= 0 || s.alpha == 0.0 ||
            (s.pair && (s.N != 2 * BN || s.K != 2 * BN)))
            return SSA_ERR_INVALID_ARGUMENT;
        SmallBatchArgs::Job &J = a.j[a.njobs];
        J.A = static_cast<const double *>(s.A);
        J.B = static_cast<const double *>(s.B);
        J.B2 = static_cast<const double *>(s.B2);
        J.C = s.C;
        J.lda8 = s.lda / per8;
        J.ldb8 = s.ldb / per8;
        J.ldc = s.ldc;
        J.K8 = s.K / per8;
        J.ntm = s.M / SBM;
        J.alpha = s.alpha;
        J.beta = s.beta;
        J.pair = s.pair;
        a.wg_begin[a.njobs] = total;
        total += s.pair ? J.ntm : J.ntm * (s.N / BN);
        flops += (s.pair ? 1.5 : 2.0) * static_cast<double>(s.M) * static_cast<double>(s.N) * static_cast<double>(s.K);
        ++a.njobs;
    }
    if (total == 0) return SSA_OK;
    a.wg_begin[a.njobs] = total;
    static DeviceFlags flags;
    if (raise_dynamic_lds(flags, {{reinterpret_cast<const void *>(&gemm_nt_small_batch_kernel<T>), sizeof(SmallSmem)}}) !=
        SSA_OK)
        return SSA_ERR_HIP;
    ProfileScope scope(sizeof(T) == 8, kProfileSmallBatch, flops, st);
    hipLaunchKernelGGL((gemm_nt_small_batch_kernel<T>), dim3(static_cast<unsigned>(total)), dim3(kGemmThreads),
                       sizeof(SmallSmem), st, a);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
int gemm_nt_small_batch_f64(int njobs, const SmallNtJob *jobs, hipStream_t st) {
    return gemm_nt_small_batch<double>(njobs, jobs, st);
}
int gemm_nt_small_batch_f32(int njobs, const SmallNtJob *jobs, hipStream_t st) {
    return gemm_nt_small_batch<float>(njobs, jobs, st);
}

// The round launch: workgroups [0, ndiag) run the diagonal-block kernel of one film each (the lowest block ids of a
// grid are dispatched first, round-robin over the XCDs: each of them starts on a CU of its own), the others one
// 128 x 128 tile of a film's pending trailing update, C -= P P^T on the lower tiles, in the band order of the
// stand-alone update.
struct TailRoundArgs {
    int nfilms, ndiag;
    int diag_film[kTailMaxFilms];
    int64_t tile_begin[kTailMaxFilms + 1];
    TailRoundJob f[kTailMaxFilms];
};
// LDS request that leaves no room for a second 72 KiB tile workgroup on a 160 KiB CU
constexpr size_t kExclusiveLds = 89 * 1024;

template <typename T>
__global__ __launch_bounds__(kGemmThreads, 2) void chol_tail_round_kernel(TailRoundArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (static_cast<int>(blockIdx.x) < a.ndiag) {
        const TailRoundJob &F = a.f[a.diag_film[blockIdx.x]];
        cholk2::chol_diag256_v2_body<cholk2::factor_t<T>, T>(static_cast<T *>(F.D), F.lda, static_cast<T *>(F.W), F.ldw,
                                                             static_cast<cholk2::factor_t<T> *>(F.scratch), F.info, F.col1,
                                                             smem_raw, static_cast<cholk2::factor_t<T> *>(F.trace));
        return;
    }
    // (equal blockIdx.x % 8 = equal XCD holds for the shifted ids as well)
    const int64_t id = xcd_contiguous(static_cast<int64_t>(blockIdx.x) - a.ndiag, static_cast<int64_t>(gridDim.x) - a.ndiag);
    int f = 0;
    while (f + 1 < a.nfilms && id >= a.tile_begin[f + 1]) ++f;
    const TailRoundJob &F = a.f[f];
    int64_t tm, tn;
    lower_band_tile(id - a.tile_begin[f], F.M / BM, tm, tn);
    const T *P = static_cast<const T *>(F.P);
    if constexpr (sizeof(T) == 8)
        tile_full_f64<OP_N, OP_T>(F.K, -1.0, P, F.ldc, P, F.ldc, 1.0, static_cast<T *>(F.C), F.ldc, tm * BM, tn * BN, smem_raw);
    else
        tile_full_f32_nt(F.K, -1.0f, P, F.ldc, P, F.ldc, 1.0f, static_cast<T *>(F.C), F.ldc, tm * BM, tn * BN, smem_raw);
}

template <typename T>
int chol_tail_round(int nfilms, const TailRoundJob *jobs, int exclusive, hipStream_t st) {
    if (nfilms <= 0 || nfilms > kTailMaxFilms) return SSA_ERR_INVALID_ARGUMENT;
    constexpr int64_t kStage = (sizeof(T) == 8) ? KC : 2 * KC;
    TailRoundArgs a{};
    a.nfilms = nfilms;
    int64_t tiles = 0;
    double flops = 0.0;
    for (int i = 0; i < nfilms; ++i) {
        TailRoundJob J = jobs[i];
        if (J.M < 0 || J.M % BM != 0 || (J.M > 0 && (J.K <= 0 || J.K % kStage != 0))) return SSA_ERR_INVALID_ARGUMENT;
        if (J.M > 0 && (reinterpret_cast<uintptr_t>(J.P) % 16 != 0 || (J.ldc * sizeof(T)) % 16 != 0))
            return SSA_ERR_INVALID_ARGUMENT;
        if (J.has_diag) a.diag_film[a.ndiag++] = i;
        a.f[i] = J;
        a.tile_begin[i] = tiles;
        const int64_t ntm = J.M / BM;
        tiles += ntm * (ntm + 1) / 2;
        flops += static_cast<double>(J.K) * static_cast<double>(J.M) * static_cast<double>(J.M + 1);
    }
    a.tile_begin[nfilms] = tiles;
    if (a.ndiag + tiles == 0) return SSA_OK;
    constexpr size_t diag_lds = sizeof(cholk2::Smem<cholk2::factor_t<T>>);
    const size_t base = sizeof(OpSmemF64) > diag_lds ? sizeof(OpSmemF64) : diag_lds;
    static DeviceFlags flags;
    if (raise_dynamic_lds(flags, {{reinterpret_cast<const void *>(&chol_tail_round_kernel<T>), kExclusiveLds}}) != SSA_OK)
        return SSA_ERR_HIP;
    ProfileScope scope(sizeof(T) == 8, kProfileRound, flops, st);
    hipLaunchKernelGGL((chol_tail_round_kernel<T>), dim3(static_cast<unsigned>(a.ndiag + tiles)), dim3(kGemmThreads),
                       exclusive ? kExclusiveLds : base, st, a);
    SSA_RETURN_IF_LAUNCH_FAILED();
    return SSA_OK;
}
int chol_tail_round_f64(int nfilms, const TailRoundJob *jobs, int exclusive, hipStream_t st) {
    return chol_tail_round<double>(nfilms, jobs, exclusive, st);
}
int chol_tail_round_f32(int nfilms, const TailRoundJob *jobs, int exclusive, hipStream_t st) {
    return chol_tail_round<float>(nfilms, jobs, exclusive, st);
}

// used by chol.hip
int gemm_op_f64(int opA, int opB, int lower, int64_t M, int64_t N, int64_t K, double alpha,
                const double *A, int64_t lda, const double *B, int64_t ldb, double beta, double *C,
                int64_t ldc, hipStream_t st) {
    return gemm_op<double>(opA, opB, lower, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}
int gemm_op_f32(int opA, int opB, int lower, int64_t M, int64_t N, int64_t K, double alpha,
                const float *A, int64_t lda, const float *B, int64_t ldb, double beta, float *C,
                int64_t ldc, hipStream_t st) {
    return gemm_op<float>(opA, opB, lower, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st);
}

}  // namespace ssa

using namespace ssa;

extern "C" int ssa_gemm_ex(int opA, int opB, int lower_only, int64_t M, int64_t N, int64_t K,
                           double alpha, const void *A, int64_t lda, const void *B, int64_t ldb,
                           double beta, void *C, int64_t ldc, int dtype, void *stream) {
    if (M < 0 || N < 0 || K < 0 || !A || !B || !C) return SSA_ERR_INVALID_ARGUMENT;
    if ((opA != OP_N && opA != OP_T) || (opB != OP_N && opB != OP_T)) return SSA_ERR_INVALID_ARGUMENT;
    if (ldc < N) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype == SSA_F64)
        return gemm_op<double>(opA, opB, lower_only, M, N, K, alpha, static_cast<const double *>(A), lda,
                               static_cast<const double *>(B), ldb, beta, static_cast<double *>(C), ldc,
                               as_stream(stream));
    if (dtype == SSA_F32)
        return gemm_op<float>(opA, opB, lower_only, M, N, K, alpha, static_cast<const float *>(A), lda,
                              static_cast<const float *>(B), ldb, beta, static_cast<float *>(C), ldc,
                              as_stream(stream));
    return SSA_ERR_INVALID_ARGUMENT;
}
