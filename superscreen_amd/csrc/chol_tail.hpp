// Batched launches of the single-stream rounds of the Cholesky schedule (chol.hip, potrf_batch): every film of a
// device advances one panel per round and each round is three kinds of launch on ONE stream,
//
//   round launch   the diagonal-block kernels of all films as the FIRST workgroups of a launch whose other workgroups
//                  are the lower tiles of the films' pending trailing updates (chol_tail_round)
//   panel launch   L21 = A21 W^T of all films                                  (gemm_nt_small_batch, pair jobs)
//   strip launch   the next block column of all films -= the new panel         (gemm_nt_small_batch)
//
// so that the schedule of the chain-bound part of a factorization needs no side streams, no events and no luck
// with the command processor's queues.  Definitions: gemm_ops.hip.
#pragma once
#include "common.hpp"

namespace ssa {

constexpr int kTailMaxFilms = 16;

// One NT product on 32 x 128 tiles (tile_small_nt): C[M x N] = alpha A[M x K] B[N x K]^T + beta C, all row-major,
// M % 32 == 0, N % 128 == 0, K % 16 == 0 (float32: K % 32), 16-byte aligned operands.
// pair: the in-place panel product L21 = A21 W^T of the Cholesky panel as ONE workgroup per 32 rows: first
//   C[:, 128:256] = A[:, 0:256] B2^T  (B2 = rows 128 .. 255 of W),  then  C[:, 0:128] = A[:, 0:128] B^T  (rows 0 .. 127)
// (W is lower triangular; a workgroup reads and writes its own rows only, and the first product has read columns
// 0 .. 127 before the second overwrites them).
struct SmallNtJob {
    const void *A, *B, *B2;
    void *C;
    int64_t lda, ldb, ldc;   // elements
    int64_t M, N, K;         // pair: N = 256, K = 256
    double alpha, beta;
    int pair;
};
int gemm_nt_small_batch_f64(int njobs, const SmallNtJob *jobs, hipStream_t st);
int gemm_nt_small_batch_f32(int njobs, const SmallNtJob *jobs, hipStream_t st);

// One film in a round launch.  diag: factor and invert the 256 x 256 block at D (leading dimension lda), inverse to
// W (leading dimension ldw), `scratch` and `info` as for the stand-alone kernel; col1: 1-based column of D[0][0].
// update: C[M x M] -= P[M x K] P^T on the tiles on or below the diagonal (M % 128 == 0, K % 16 == 0; M = 0: none).
struct TailRoundJob {
    void *D, *W, *scratch;
    int32_t *info;
    int lda, ldw, col1, has_diag;
    void *C;
    const void *P;
    int64_t ldc, M, K;
    void *trace = nullptr;   // debugging (chol.hip, CholDebug::trace): the block as the kernel READ it, as register images
};
// exclusive: the launch asks for so much LDS that no second workgroup of an MFMA tile kernel fits on a CU beside one
// of its own: the diagonal-block workgroups then have their CUs to themselves (chain-bound rounds: few tiles);
// otherwise two workgroups per CU as in the stand-alone trailing update (update-bound rounds).
int chol_tail_round_f64(int nfilms, const TailRoundJob *jobs, int exclusive, hipStream_t st);
int chol_tail_round_f32(int nfilms, const TailRoundJob *jobs, int exclusive, hipStream_t st);

// ---- Fused rounds (round 6): ONE launch per round --------------------------------------------------------------
// A round of every film as one launch whose workgroups synchronise through device-side flags:
//
//   workgroups 0 .. films-1        diagonal-block kernel of block c of a film (L11, W = L11^-1); publishes FLAG_DIAG
//   tile workgroups                128 x 128 tiles (tm >= tn) of the tile columns [t0, t1) of a film:
//                                  C -= P P^T with the panels [level(tn), c) -- all of them finished before the launch, so
//                                  these need no flag; `level` is kept per 256-wide block column, so that the host can
//                                  serve far columns every other round with K = 512 (one pass over C per two panels,
//                                  as the stand-alone trailing updates of the stream part do); level = c: nothing to do
//   chain workgroups               one per 32 rows below the diagonal block: waits for FLAG_DIAG, computes its rows of
//                                  L21 = A21 W^T in place; the 8 workgroups of the first 256 rows ("head": the rows of
//                                  the next diagonal block) then publish to FLAG_HEAD; every chain workgroup waits for
//                                  the 8 heads and applies the panels [strip_from, c + 256) -- the new one included --
//                                  to its rows of block column c + 256 (the next diagonal block and what lies below it)
//
// Dependencies inside the launch: diag -> panel rows (W), head panel rows -> every strip (the B operand of the strip);
// everything else a workgroup reads was final before the launch and nothing it writes is read by a tile workgroup
// (tiles write columns >= c + 512 and read columns < c; the chain writes columns c .. c + 511).  The chain workgroups
// sit in the grid behind `chain_pos` tile workgroups -- about as many as are dispatched while the diagonal block is
// being factored -- so that they neither hold workgroup slots asleep nor start after the last tile.  Progress: a waiting
// workgroup only waits for workgroups with LOWER block ids (the diagonal blocks, heads first among the chain), which
// the dispatcher has started before it -- and every wait is bounded (a timeout sets info = -3: the host then takes the
// LU route).  Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility): producer = plain stores, every
// wave's s_waitcnt vmcnt(0), workgroup barrier, ONE lane: agent release fence, s_waitcnt vmcnt(0), relaxed agent
// atomic; consumer = ONE lane polls (relaxed agent loads + s_sleep), agent acquire fence, s_waitcnt vmcnt(0),
// workgroup barrier, then the loads.
constexpr int kRoundMaxBlockCols = 64;    // block columns behind the diagonal block of a round (tail <= 16 384 columns)
enum : int { kFlagDiag = 0, kFlagHead = 1, kFlagPanel = 2, kRoundFlagWords = 8 };

struct FusedRoundJob {
    void *A;                 // the matrix (n x n, leading dimension lda), n % 256 == 0
    int64_t lda, n;
    int64_t c;               // first column of the diagonal block of this round (c % 256 == 0, c < n)
    void *W, *scratch;       // as TailRoundJob
    int32_t *info;
    int ldw;
    void *trace = nullptr;
    int64_t strip_from;      // block column c + 256 has every panel before this column applied (< c + 256)
    int64_t tile_cols_end;   // tiles cover the columns [c + 512, tile_cols_end) (multiple of 256; <= n; c + 512 if none)
    int64_t level[kRoundMaxBlockCols];   // level[q]: block column c + 512 + 256 q has every panel before this column
    uint32_t *flags;         // kRoundFlagWords words of device memory, zeroed before the film's first round
    uint32_t round_no;       // 1, 2, ... per film: FLAG_DIAG reaches round_no when the diagonal block is done
    uint32_t panel_target;   // big_strips: FLAG_PANEL reaches this when the round's panel workgroups are done (a running
                             // total over the film's big-strip rounds: + (n - c - 256) / 32 per round)
    uint32_t head_target;    // small form: FLAG_HEAD reaches this when the round's head workgroups are done (a running
                             // total over the film's small-form rounds: + min(8, (n - c - 256) / 32) per round)
};
// exclusive: as chol_tail_round (one workgroup per CU).  chain_pos: tile workgroups in front of the chain workgroups
// (< 0 or more than there are: all of them).  big_strips: the chain in its form for update-bound rounds -- one workgroup
// per 32 panel rows that ONLY computes them (and adds to FLAG_PANEL), then block column c + 256 as 128 x 128 tiles of
// their own workgroups: the panels before c at once, the new one behind FLAG_PANEL.  Short workgroups that never hold
// a slot asleep for long; the small form's 32-row workgroups (panel rows, wait for the heads, two small strip tiles)
// are the low-latency form of the chain-bound rounds.
// wg_times (debugging, may be null): 3 words per workgroup of the launch {role 1 diag / 2 tile / 3 panel rows / 4 strip
// tile, start, end} in wall_clock64() ticks (100 MHz), written if the launch has at most wg_capacity workgroups.
int chol_fused_round_f64(int nfilms, const FusedRoundJob *jobs, int exclusive, int64_t chain_pos, int big_strips, hipStream_t st,
                         unsigned long long *wg_times = nullptr, int64_t wg_capacity = 0);
int chol_fused_round_f32(int nfilms, const FusedRoundJob *jobs, int exclusive, int64_t chain_pos, int big_strips, hipStream_t st,
                         unsigned long long *wg_times = nullptr, int64_t wg_capacity = 0);

}  // namespace ssa
