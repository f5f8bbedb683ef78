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

}  // namespace ssa
