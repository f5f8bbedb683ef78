/*
 * superscreen_hip.h -- C ABI of libsuperscreen_hip.so (MI355X / gfx950 only).
 *
 * The reference (loganbvh/superscreen v0.13.0) has no FFI of its own: its hot path is
 * Python calling numba-JIT kernels and scipy/LAPACK.  Each entry point below replaces one
 * of those call sites; the reference interface it stands in for is cited as file:line
 * relative to /root/reference/superscreen/.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add at each of these sites.
 *
 * Conventions
 *   - every function returns an int status: SSA_OK (0) or a negative SSA_ERR_* code;
 *     nothing throws across the boundary;
 *   - all pointers are DEVICE pointers (hipMalloc'd / torch CUDA tensors) unless the
 *     parameter is documented as host; buffers are caller-owned; no hidden allocation:
 *     scratch space is passed in explicitly and sized by the *_workspace_bytes queries;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only
 *     enqueue work, they never synchronise; one stream per call, re-entrant across
 *     streams and devices (the current HIP device of the calling thread is used);
 *   - matrices are row-major ("C order", numpy default) with an explicit leading
 *     dimension ld (elements between consecutive rows, ld >= number of columns);
 *     ld must be even for n x n outputs so that rows stay 16-byte aligned;
 *   - `dtype` selects the storage/solve precision (device.solve_dtype of the reference,
 *     device/device.py:57): SSA_F32 or SSA_F64.  Geometry (xy, vertex areas w, edge
 *     vector C, sparse operator values) is always float64, like mesh.sites in the
 *     reference (distance.py:101: Q is always evaluated in float64 and cast afterwards,
 *     solver/utils.py:291);
 *   - index arrays are int64 (reference: np.int64), LU pivots int32 0-based LAPACK-style
 *     row interchanges (scipy.linalg.lu_factor's `piv`).
 */
#ifndef SUPERSCREEN_HIP_H
#define SUPERSCREEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSA_F32 0
#define SSA_F64 1

#define SSA_OK 0
#define SSA_ERR_INVALID_ARGUMENT (-1)
#define SSA_ERR_HIP (-2)
#define SSA_ERR_WORKSPACE_TOO_SMALL (-3)
#define SSA_ERR_UNSUPPORTED_SIZE (-4)
#define SSA_ERR_RCCL (-5) /* librccl missing, or an RCCL call failed */

#define SSA_ABI_VERSION 6

/* Library / device introspection (host-side, no reference counterpart). */
int ssa_abi_version(void);
const char *ssa_error_string(int status);
/* Fills compute-unit count and total HBM bytes of the current device. */
int ssa_device_info(int *num_cus, size_t *hbm_bytes, char *arch_name, int arch_name_len);

/* ---------------------------------------------------------------------------------- */
/* (1) Kernel matrix Q                                                                 */
/* ---------------------------------------------------------------------------------- */

/*
 * Replaces  MeshOperators.Q_matrix(points, weights)  device/mesh.py:435-458, i.e.
 *   q = distance.q_matrix(points)        distance.py:87-115   q_ij = 1/(4 pi |r_i-r_j|^3)
 *   diag = -(C + einsum("ij,j->i", q, w)) / w ; fill_diagonal(q, diag) ; return -q
 * fused into one pass:  Q_ij = -q_ij (i != j),  Q_ii = (C_i + sum_{l != i} q_il w_l)/w_i.
 *   xy    [n,2] f64   mesh sites
 *   w     [n]   f64   vertex areas (mesh.operators.weights)
 *   C     [n]   f64   edge vector (MeshOperators.C_vector, device/mesh.py:401-432)
 *   Q     [n,ldq] dtype  out (may be NULL: only the diagonal is produced)
 *   qdiag [n]   f64   out, Q_ii in float64 (may be NULL)
 */
int ssa_q_assemble(const double *xy, const double *w, const double *C, int64_t n, void *Q,
                   int64_t ldq, int dtype, double *qdiag, void *stream);

/*
 * Replaces  _build_system_2d / _build_system_1d  solver/solve_film.py:285-305 together with
 * the dense operands they slice (film_info.kernel, film_info.laplacian -- the latter is
 * `laplacian.toarray()` in the reference, solver/utils.py:292; here it stays CSR):
 *   out[r,c] = sign * ( Q[i,j] * w[j] - Lambda[j] * Del2[i,j] ),  i = rows[r], j = cols[c]
 * with Q_ij regenerated on the fly from the sites (never read from HBM).
 *   rows/cols  int64 vertex indices (rows may be NULL = identity 0..nr-1)
 *   qdiag [n] f64 from ssa_q_assemble;  Lambda [n] f64 (cast to dtype inside, like
 *   solver/utils.py:269);  lap_* CSR of the mesh Laplacian (n x n, f64 values)
 *   sign = -1 writes -A directly, which is what gets LU-factored (solve_film.py:279).
 *   row_scale [n] f64 or NULL: out[r,c] is additionally multiplied by row_scale[rows[r]]; with
 *   row_scale = w this is S = diag(w) A, which is symmetric positive definite for a
 *   homogeneous film (see ssa_chol_factor).  lower_only != 0 (needs rows == cols): only the
 *   entries on or below the diagonal are guaranteed to be written.
 *   workspace: ssa_system_assemble_workspace_bytes(n, nr, nc)
 */
size_t ssa_system_assemble_workspace_bytes(int64_t n, int64_t nr, int64_t nc);
int ssa_system_assemble(const double *xy, const double *w, const double *qdiag,
                        const double *Lambda, int64_t n, const int64_t *lap_indptr,
                        const int64_t *lap_indices, const double *lap_data,
                        const int64_t *rows, int64_t nr, const int64_t *cols, int64_t nc,
                        double sign, const double *row_scale, int lower_only, void *out,
                        int64_t ldo, int dtype, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---------------------------------------------------------------------------------- */
/* (2) Dense LU factor / solve                                                         */
/* ---------------------------------------------------------------------------------- */

/*
 * Replaces  scipy.linalg.lu_factor(-A)  solver/solve_film.py:279 (LAPACK ?getrf):
 * in-place blocked right-looking LU with partial (row) pivoting, P*A = L*U, L unit lower.
 *   A    [n,lda] dtype  in: matrix, out: L\U
 *   ipiv [n] int32      out: row i was interchanged with row ipiv[i] (0-based)
 *   info [1] int32      out (device): 0, or k+1 if U[k,k] is exactly zero (LAPACK info)
 *   aux                 out: ssa_lu_aux_bytes(n, dtype) bytes that ssa_lu_solve needs
 *                       (inverses of the diagonal blocks of L and U)
 *   workspace           ssa_lu_factor_workspace_bytes(n, dtype), scratch
 */
size_t ssa_lu_factor_workspace_bytes(int64_t n, int dtype);
size_t ssa_lu_aux_bytes(int64_t n, int dtype);
int ssa_lu_factor(void *A, int64_t n, int64_t lda, int32_t *ipiv, int32_t *info, void *aux,
                  int dtype, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Turns LAPACK interchanges into a gather permutation: perm[i] = index of the original row
 * that ends up in row i (LU == A[perm]).  HOST arrays, pure C, no GPU work.
 */
int ssa_lu_pivots_to_permutation(const int32_t *ipiv_host, int64_t n, int64_t *perm_host);

/*
 * Replaces  scipy.linalg.lu_solve(lu_piv, h)  solver/solve_film.py:530 (LAPACK ?getrs)
 * for a right-hand side that has ALREADY been row-permuted (b = h[perm]); the permutation
 * is folded into the gather of ssa_film_rhs.  Solves L U X = B in place.
 *   B [n,ldb] dtype  in: permuted rhs, out: solution; nrhs >= 1 columns (batched sweeps)
 *   workspace: ssa_lu_solve_workspace_bytes(n, nrhs, dtype)
 */
size_t ssa_lu_solve_workspace_bytes(int64_t n, int64_t nrhs, int dtype);
int ssa_lu_solve(const void *LU, int64_t n, int64_t lda, const void *aux, void *B,
                 int64_t nrhs, int64_t ldb, int dtype, void *workspace,
                 size_t workspace_bytes, void *stream);

/*
 * Cholesky alternative to the pair above for homogeneous films.  The reference solves
 *   gf = lu_solve(lu_factor(-A), h)                       solver/solve_film.py:279, :530
 * with A = Q[ix,ix] w[ix] - Lambda Del2[ix,ix].  S = diag(w[ix]) A is symmetric positive
 * definite (Q is symmetric off its diagonal, Del2 = diag(1/w) L with L symmetric, and S is
 * strictly diagonally dominant with a positive diagonal), so  gf = -S^-1 (w[ix] .* h):
 *   The factorization works on S padded with identity rows/columns to np = ssa_chol_padded_n(n)
 *   (a multiple of 256, so that every MFMA tile is a full one): the S buffer must hold np rows
 *   with lda >= np; rows / columns >= n are initialised by ssa_chol_factor itself.
 *   ssa_chol_factor: in-place  S = L L^T  on the lower triangle of S [n,lda] (what
 *                    ssa_system_assemble(row_scale = w, lower_only = 1, sign = +1) writes);
 *                    (1/3) n^3 flops, trailing update = MFMA SYRK on the lower tiles only.
 *                    info [1] int32 (device): 0, or > 0 if a pivot was not positive (then the
 *                    caller must fall back to ssa_lu_factor on a freshly assembled -A).
 *                    On return the buffer holds L on/below and L^T above the diagonal (both
 *                    triangular solves then stream row-major), and aux (ssa_chol_aux_bytes)
 *                    the inverses of the 4096 x 4096 diagonal blocks of L and their transposes.
 *   ssa_chol_factor_batch: the films of one device (factorize_linear_systems loops over them,
 *                    solver/solve_film.py:174) factored in ONE interleaved schedule: all MFMA
 *                    trailing updates round-robin on the caller's stream, every film's panel
 *                    chain on its own internal side stream, hidden behind the other films'
 *                    updates.  A, n, lda, info, aux: HOST arrays of `count` entries with the
 *                    per-matrix arguments of ssa_chol_factor.  Results identical to `count`
 *                    separate ssa_chol_factor calls (float64: bit for bit; float32: to rounding, the
 *                    schedule's two-panel steps depend on the largest matrix of the batch).
 *   float32 (SSA_F32): storage, trailing updates, panels and solves are float32; the 256 x 256 diagonal
 *                    blocks are factored and inverted in float64 inside the diagonal-block kernel and
 *                    rounded back (ssa_lu_factor's route without interchanges does the same), which brings
 *                    the backward error of the factorization to LAPACK spotrf / sgetrf's.
 *   ssa_chol_solve:  L L^T X = B in place, nrhs >= 1; workspace ssa_chol_solve_workspace_bytes.
 *                    nrhs = 1: triangular GEMV chain at the HBM rate; 2..64 (float64): the factor is
 *                    streamed once by an MFMA kernel that takes its operands straight from global memory;
 *                    more: tiled MFMA GEMMs with split-K (applied-field scans, solve_sweep).
 */
int64_t ssa_chol_padded_n(int64_t n);
size_t ssa_chol_aux_bytes(int64_t n, int dtype);
int ssa_chol_factor(void *S, int64_t n, int64_t lda, int32_t *info, void *aux, int dtype,
                    void *stream);
int ssa_chol_factor_batch(int count, void *const *S, const int64_t *n, const int64_t *lda,
                          int32_t *const *info, void *const *aux, int dtype, void *stream);
size_t ssa_chol_solve_workspace_bytes(int64_t n, int64_t nrhs, int dtype);
int ssa_chol_solve(const void *L, int64_t n, int64_t lda, const void *aux, void *B, int64_t nrhs,
                   int64_t ldb, int dtype, void *workspace, size_t workspace_bytes, void *stream);
/* ssa_chol_solve_batch: `count` single-right-hand-side solves L_i L_i^T x_i = b_i (the films of a device in one
 * pass of the Jacobi loop, solver/solve.py:517-536 - they are independent of each other) in lockstep: every block
 * step of all solves is one launch, so that the short launches of the triangular GEMV chain (a 4096-row inverse
 * block, a film's last block column) run side by side in one grid.  L, n, lda, aux, B, workspace, workspace_bytes:
 * HOST arrays of `count` entries with the arguments of ssa_chol_solve for nrhs = 1, ldb = 1 (workspace_bytes[i] >=
 * ssa_chol_solve_workspace_bytes(n[i], 1, dtype)).  b_is_padded != 0: every B[i] holds ssa_chol_padded_n(n[i])
 * elements, zero from n[i] on, and is solved where it is (no staging copies; the padding stays zero) - what a caller
 * that runs many passes with the same buffers wants.  Results bit-identical to `count` ssa_chol_solve calls.
 * Every L[i] and aux[i] must be 16-byte aligned with lda[i] * sizeof(element) a multiple of 16 (what the package's
 * padded buffers are); otherwise SSA_ERR_INVALID_ARGUMENT - ssa_chol_solve, which has a scalar path, takes such
 * factors. */
int ssa_chol_solve_batch(int count, const void *const *L, const int64_t *n, const int64_t *lda,
                         const void *const *aux, void *const *B, int b_is_padded, int dtype, void *const *workspace,
                         const size_t *workspace_bytes, void *stream);
/* The `_blk` forms (ABI 6) take the row count of the solves' diagonal blocks: SSA_CHOL_SOLVE_BLOCK_DEFAULT (4096, what
 * the forms above use) or a smaller power-of-two multiple of 256.  A factorization only builds the levels of its block
 * inverses BELOW that size -- at 2048 a quarter of the block-inverse flops (config H: 3-4 ms of a 95 ms
 * factorization) -- and a solve then takes twice the dependent launches (config H: + 0.12 ms per pass): the choice
 * for a factorization that serves few solves, e.g. the one inside the reference's plain cold call
 * `solve(device=...)` (solver/solve.py:380-399).  The SAME value must be passed to the solves of a factorization.
 * aux layout and ssa_chol_aux_bytes do not depend on it. */
#define SSA_CHOL_SOLVE_BLOCK_DEFAULT 4096
int ssa_chol_factor_batch_blk(int count, void *const *S, const int64_t *n, const int64_t *lda,
                              int32_t *const *info, void *const *aux, int dtype, int solve_block, void *stream);
int ssa_chol_solve_blk(const void *L, int64_t n, int64_t lda, const void *aux, void *B, int64_t nrhs,
                       int64_t ldb, int dtype, void *workspace, size_t workspace_bytes, int solve_block, void *stream);
int ssa_chol_solve_batch_blk(int count, const void *const *L, const int64_t *n, const int64_t *lda,
                             const void *const *aux, void *const *B, int b_is_padded, int dtype,
                             void *const *workspace, const size_t *workspace_bytes, int solve_block, void *stream);
/*
 * Diagnostics of the factorization schedule (no reference counterpart).  The panel chains run on internal
 * high-priority streams; which hardware queue / command-processor pipe the runtime gives a stream decides what a
 * dependent launch on it costs beside the trailing updates (2-4 us, or 35-40 us when it shares the pipe of the
 * caller's stream or of another chain), so the first ssa_chol_factor* call on a device measures its chain streams
 * against the caller's stream and against each other (about 10 ms, once) and gives every film's chain a pipe of its
 * own as far as there are pipes.  ssa_chol_chain_stream_costs writes, for up to `capacity` chain streams of the
 * current device in the order in which the schedule uses them, the measured cost of a dependent launch
 * (microseconds, alone beside the caller's stream) and the group of streams it shares a pipe with (0 = the caller's
 * pipe, 1, 2, ... = others); either pointer may be NULL.  Returns the number of chain streams (0 before the first
 * factorization on this device).
 */
int ssa_chol_chain_stream_costs(double *microseconds, int32_t *pipe_group, int capacity);
/* Forgets the measurements of the current device (they are kept per caller stream, least recently used first out):
 * the next factorization measures again.  For callers that create or destroy streams between factorizations (the
 * runtime may move hardware queues between pipes then) or recycle stream handles. */
int ssa_chol_chain_streams_invalidate(void);

/*
 * Replaces the numba kernels _biot_savart_2d_z / _biot_savart_2d_vector
 * (sources/current.py:13-57, :60-110) behind biot_savart_2d (:113-199) and
 * Solution.screening_field_at_position / field_at_position (solution.py:611-831):
 * field of the sheet current of one film at arbitrary points,
 *   pref_k = a_k |r - r_k|^-3,  r - r_k = (dx, dy, dz),  r_k = (x_k, y_k, z0)
 *   vector == 0:  out[i]   = prefactor * sum_k pref_k (Jx_k dy - Jy_k dx)
 *   vector != 0:  out[i,:] = prefactor * sum_k pref_k (Jy_k dz, -Jx_k dz, Jx_k dy - Jy_k dx)
 * src_xy [ns,2], src_areas [ns], src_J [ns,2], eval_xyz [np,3], out [np] or [np,3]: float64, in the
 * caller's length / current units; prefactor = (mu_0 / 4 pi) * (A/m per current_unit/length_unit)
 * gives tesla.  Deterministic two-stage reduction; workspace ssa_sheet_field_workspace_bytes.
 *
 * ssa_sheet_potential: the in-plane vector potential of the same sheet
 * (Solution.vector_potential_at_position, solution.py:833-934; cdist + einsum there):
 *   out[i,0:2] = prefactor * sum_k a_k (Jx_k, Jy_k) / |r_i - r_k|      out [np,2] float64
 * workspace: ssa_sheet_field_workspace_bytes(np, 1).
 */
size_t ssa_sheet_field_workspace_bytes(int64_t np, int vector);
int ssa_sheet_field(const double *src_xy, const double *src_areas, const double *src_J, int64_t ns,
                    double z0, const double *eval_xyz, int64_t np, double prefactor, int vector,
                    double *out, void *workspace, size_t workspace_bytes, void *stream);
int ssa_sheet_potential(const double *src_xy, const double *src_areas, const double *src_J, int64_t ns,
                        double z0, const double *eval_xyz, int64_t np, double prefactor, double *out,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------- */
/* (3) Per-film vector kernels of solve_film                                           */
/* ---------------------------------------------------------------------------------- */

/*
 * Replaces BLAS gemv at  solver/solve_film.py:503 (A_hole @ g[ix]) and :565 (Q @ (w*g)):
 *   y[r] = alpha * sum_c M[r,c] * (xscale ? xscale[c] : 1) * x[xidx ? xidx[c] : c]
 *          + beta * y[r]
 *   M [nr,ldm] dtype; x, y dtype; xscale [nc] dtype or NULL; xidx int64 [nc] or NULL.
 * (Batched right-hand sides go through ssa_row_scale + ssa_gemm instead.)
 */
int ssa_gemv(const void *M, int64_t nr, int64_t nc, int64_t ldm, const void *x,
             const void *xscale, const int64_t *xidx, void *y, double alpha, double beta,
             int dtype, void *stream);

/* y[r,b] = s[r] * x[r,b]  (the w*g of solve_film.py:565 for batched g), row-major [nr,nvec]. */
int ssa_row_scale(const void *x, const void *s, void *y, int64_t nr, int64_t nvec, int dtype,
                  void *stream);

/*
 * Matrix-free form of  Q @ (w * g)  (solver/solve_film.py:565): regenerates q_ij from the
 * sites instead of reading n^2 stored entries.
 *   out[i] = alpha * ( qdiag[i] w[i] g[i] - sum_{j != i} q_ij w[j] g[j] )
 */
int ssa_self_field(const double *xy, const double *w, const double *qdiag, const void *g,
                   int64_t n, void *out, double alpha, int dtype, void *workspace,
                   size_t workspace_bytes, void *stream);
size_t ssa_self_field_workspace_bytes(int64_t n);

/*
 * The same quantity from O(n) data where that is possible.  For a row i of the film interior the
 * solved system  -A gf = h  (solve_film.py:529-531) IS the London equation
 *   H_applied[i] + H_other[i] + (Q (w g))[i] = (Laplacian (Lambda g))[i],
 * (A = Q w - Lambda Laplacian, the hole columns enter through Ha_eff: both sides carry the same terms),
 * so on those rows  self_field = Laplacian(Lambda g) - H_applied - H_other  with the sparse mesh
 * Laplacian.  Valid for a homogeneous film without vortices or terminals; agrees with the all-pairs sum to
 * the residual of the linear solve (1e-12 relative in float64).
 *   ssa_london_field_rows: out[rows[k]] = sum_j lap[r,j] Lambda[j] g[j] - applied[r] - other[r], r = rows[k]
 *                          (other may be NULL); g, applied, other, out: [n, nvec] dtype, row-major
 *   ssa_self_field_rows:   out[rows[k]] = the all-pairs value of ssa_self_field, for the remaining rows
 *                          (mesh vertices outside the film interior: boundary, vacuum buffer, holes);
 *                          workspace ssa_self_field_workspace_bytes(nr)
 * Other entries of out are left untouched.
 */
int ssa_self_field_rows(const double *xy, const double *w, const double *qdiag, const void *g, int64_t n,
                        const int64_t *rows, int64_t nr, void *out, double alpha, int dtype,
                        void *workspace, size_t workspace_bytes, void *stream);
int ssa_london_field_rows(const int64_t *lap_indptr, const int64_t *lap_indices, const double *lap_data,
                          const double *Lambda, const void *g, const void *applied, const void *other,
                          const int64_t *rows, int64_t nr, int64_t nvec, void *out, int dtype, void *stream);

/*
 * Replaces  h = Hz_applied[indices] - Ha_eff[indices]  solver/solve_film.py:486-488,526-529
 * with the LU row permutation folded in (see ssa_lu_solve):
 *   h[k,b] = applied[idx[k], b] + (other ? other[idx[k], b] : 0) - ha_eff[idx[k], b]
 * idx = film indices composed with the LU permutation; all arrays dtype, row-major [.,nvec].
 */
int ssa_film_rhs(const void *applied, const void *other, const void *ha_eff,
                 const int64_t *idx, int64_t ni, int64_t nvec, void *h, int dtype,
                 void *stream);

/* g[idx[k], b] += gf[k, b]   (solver/solve_film.py:531)  */
int ssa_scatter_add(void *g, const int64_t *idx, const void *gf, int64_t ni, int64_t nvec,
                    int dtype, void *stream);

/* g[idx[k], b] += value[b]  (hole boundary condition g[hole] = I_circ, solve_film.py:498-502);
 * value is a HOST array of nvec doubles. */
int ssa_index_add_scalar(void *g, const int64_t *idx, int64_t ni, const double *value_host,
                         int64_t nvec, int dtype, void *stream);

/*
 * Replaces the two scipy.sparse CSR SpMVs of  solver/solve_film.py:556
 *   J = [grad_y @ g, -(grad_x @ g)].T          (float64 operators => float64 J)
 * gx / gy share one CSR pattern (indptr, indices) with two value arrays.
 *   g [n,nvec] dtype;  J [n,nvec,2] f64 out.
 */
int ssa_current_density(const int64_t *indptr, const int64_t *indices, const double *gx_data,
                        const double *gy_data, const void *g, int64_t n, int64_t nvec,
                        double *J, int dtype, void *stream);

/* y[i] = alpha * x[i] (elementwise; the "/ field_conversion" of solve_film.py:566-574). */
int ssa_scale(const void *x, void *y, double alpha, int64_t count, int dtype, void *stream);

/* ---------------------------------------------------------------------------------- */
/* (4) Inter-film Biot-Savart coupling                                                 */
/* ---------------------------------------------------------------------------------- */

/*
 * Replaces  biot_savart_film_to_film  solver/solve.py:28-73 and the accumulation
 * `other_screening_fields[film] += ...` (:508):
 *   out[i] (+)= sum_{j in [src_begin, src_end)} (1/4pi) a_j (Jx_j dy - Jy_j dx)
 *                                               (dx^2 + dy^2 + dz^2)^(-3/2)
 *   dx = tgt_xy[i,0] - src_xy[j,0], dy likewise; dz = z0(target) - z0(source).
 *   src_areas [ns] dtype (film_info.weights), src_J [ns,2] f64, out [nt] dtype.
 *   accumulate != 0 adds into out (float64 arithmetic, one rounding to dtype at the end).
 * [src_begin, src_end) lets several GPUs each sum a slice of the sources; the partial
 * fields are then summed with one RCCL all-reduce (superscreen_amd.parallel).
 *   workspace: ssa_biot_savart_workspace_bytes(nt)
 */
size_t ssa_biot_savart_workspace_bytes(int64_t nt);
int ssa_biot_savart(const double *src_xy, const void *src_areas, const double *src_J,
                    int64_t ns, int64_t src_begin, int64_t src_end, const double *tgt_xy,
                    int64_t nt, double dz, void *out, int accumulate, int dtype,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------- */
/* (5) Building blocks exposed for tests and benchmarks                                */
/* ---------------------------------------------------------------------------------- */

/* C = alpha * A(MxK) * B(KxN) + beta * C, row-major, MFMA tiles (the LU trailing update). */
int ssa_gemm(int64_t M, int64_t N, int64_t K, double alpha, const void *A, int64_t lda,
             const void *B, int64_t ldb, double beta, void *C, int64_t ldc, int dtype,
             void *stream);

/*
 * C = alpha * op(A) * op(B) + beta * C with op = 0 (as stored) or 1 (transposed), row-major;
 * lower_only != 0 (square C) computes only the 128 x 128 tiles on or below the diagonal -- the
 * SYRK-shaped trailing update of the Cholesky factorization.
 */
int ssa_gemm_ex(int opA, int opB, int lower_only, int64_t M, int64_t N, int64_t K, double alpha,
                const void *A, int64_t lda, const void *B, int64_t ldb, double beta, void *C,
                int64_t ldc, int dtype, void *stream);

/*
 * Instrumentation for bench.py: between ssa_profile_begin() and ssa_profile_end() every launch
 * of the f64 MFMA GEMM kernels is bracketed by HIP events on its own stream.
 * ssa_profile_read(kind, ...) waits for the recorded events of one kernel kind and returns the
 * summed kernel time [ms], the summed algorithmic flops and the launch count:
 *   kind 0: gemm_kernel<double, true>                 (LU trailing / in-panel updates, 2 M N K)
 *   kind 1: gemm_op_kernel<double, N, T>, lower_only  (Cholesky trailing SYRK; algorithmic flops of the
 *           entries on / below the diagonal: K * M * (M + 1))
 *   kind 2: gemm_op_kernel<double, N, T>, all tiles   (the Cholesky chain's strips and L21 = A21 W^T
 *           panel products; 2 M N K)
 *   kind 3: chol_tail_round_kernel<double>            (the round launches of the Cholesky schedule's last part:
 *           diagonal-block kernels of all films + lower tiles of their pending updates; flops of the tiles)
 *   kind 4: gemm_nt_small_batch_kernel<double>        (the rounds' batched panel and strip products; 2 M N K,
 *           panel products 1.5 M 256^2)
 * ssa_profile_begin_kinds(mask) brackets only the kinds whose bit is set in `mask` (bit k = kind k): an
 * event pair costs a few microseconds on its stream, which matters for the ~ 1300 short launches of kind 2
 * per factorization and not for the ~ 100 trailing updates; ssa_profile_begin() = all kinds.
 */
int ssa_profile_begin(void);
int ssa_profile_begin_kinds(unsigned kinds_mask);
int ssa_profile_read(int kind, double *ms, double *flops, int64_t *launches);
int ssa_profile_end(void);

/*
 * Multi-vector forms of ssa_self_field and ssa_biot_savart for sweeps that carry nvec right-hand
 * sides through one factorization (BASELINE config 4; the solve itself then runs on the MFMA GEMM
 * path of ssa_chol_solve / ssa_lu_solve with nrhs = nvec).  Operands are row-major with the vector
 * index fastest: g, out [n, nvec]; src_J [ns, nvec, 2]; r^-3 is evaluated once per pair and reused
 * for 16 vectors at a time.  Same formulas, same deterministic two-stage reduction.
 *   workspace: ssa_pairwise_multi_workspace_bytes(number of targets)
 *   ssa_self_field_multi_rows / ssa_biot_savart_multi_rows: only the listed target rows (out[rows[k], :]),
 *   like ssa_self_field_rows; workspace for nr targets.
 */
size_t ssa_pairwise_multi_workspace_bytes(int64_t nt);
int ssa_self_field_multi(const double *xy, const double *w, const double *qdiag, const void *g,
                         int64_t n, int64_t nvec, void *out, double alpha, int dtype, void *workspace,
                         size_t workspace_bytes, void *stream);
int ssa_biot_savart_multi_rows(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                               const double *tgt_xy, int64_t nt, const int64_t *rows, int64_t nr, double dz,
                               int64_t nvec, void *out, int accumulate, int dtype, void *workspace,
                               size_t workspace_bytes, void *stream);
int ssa_self_field_multi_rows(const double *xy, const double *w, const double *qdiag, const void *g, int64_t n,
                              int64_t nvec, const int64_t *rows, int64_t nr, void *out, double alpha, int dtype,
                              void *workspace, size_t workspace_bytes, void *stream);
int ssa_biot_savart_multi(const double *src_xy, const void *src_areas, const double *src_J, int64_t ns,
                          const double *tgt_xy, int64_t nt, double dz, int64_t nvec, void *out,
                          int accumulate, int dtype, void *workspace, size_t workspace_bytes,
                          void *stream);

/* HBM write-bandwidth probe: fills `bytes` bytes with a 16-byte pattern, one workgroup per CU, grid-stride, plain
 * stores (the fastest fill shape measured on MI355X: what a pure store stream reaches on this box). */
int ssa_fill_probe(void *dst, size_t bytes, void *stream);

/*
 * FP64 matrix-pipe probe: every wave of a chip-filling grid issues `iters` x 4 independent
 * v_mfma_f64_16x16x4_f64 from registers (no memory traffic).  *flops_out (host) receives the
 * flops issued; time the launch to get the MFMA rate the chip sustains (clock under matrix load),
 * the practical ceiling next to the nominal 78.6 TFLOP/s.  sink: >= 8 bytes of device memory.
 */
int ssa_mfma_probe(int iters, void *sink, double *flops_out, void *stream);

/*
 * LU without row interchanges, with look-ahead, for a batch of matrices (the films of a device) -- the fast
 * form of  lu_factor(-A)  solver/solve_film.py:279 for the matrices this solver actually meets: the London
 * systems are strictly diagonally dominant by rows, LAPACK's partial pivoting never swaps (ipiv == arange).
 * Same schedule as ssa_chol_factor_batch: 256-column panels factored on a side stream while the previous
 * trailing update (one NN MFMA GEMM, K = 256 or 512) still runs, all matrices of the batch interleaved.
 * The result is what ?getrf returns IF AND ONLY IF no multiplier exceeds 1 in magnitude (then the diagonal is
 * the pivot LAPACK picks at every column, ties included); that condition is verified on the finished factor:
 *   info[i] =  0   factors and ipiv (= arange) are LAPACK's, to rounding
 *           = -2   a row interchange would have been needed (or a pivot block was singular): the buffer
 *                  holds garbage; assemble the matrix again and call ssa_lu_factor (partial pivoting)
 *   A[i]    [np, lda] dtype, np = ssa_lu_padded_n(n): the caller allocates np rows and lda >= np; rows and
 *           columns n .. np-1 are overwritten with the identity before factoring (full MFMA tiles everywhere);
 *           the leading n x n part of the result is the factorization of the n x n matrix
 *   ipiv[i] [np] int32,  aux[i]: ssa_lu_aux_bytes(n)  (for ssa_lu_solve with the TRUE n),
 *   workspace[i]: ssa_lu_factor_nopivot_workspace_bytes(n) each (distinct buffers).
 */
int64_t ssa_lu_padded_n(int64_t n);
size_t ssa_lu_factor_nopivot_workspace_bytes(int64_t n, int dtype);
int ssa_lu_factor_nopivot_batch(int count, void *const *A, const int64_t *n, const int64_t *lda,
                                int32_t *const *ipiv, int32_t *const *info, void *const *aux, int dtype,
                                void *const *workspace, const size_t *workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------- */
/* (7) Inter-film coupling across GPUs: RCCL all-reduce of the coupling vector         */
/* ---------------------------------------------------------------------------------- */

/*
 * The reference sums the field of every other film into other_screening_fields[film] in one process
 * (solver/solve.py:491-515).  With the ordered film pairs split by SOURCE SLICE over the ranks of a node
 * (ssa_biot_savart's src_begin / src_end), every rank holds a partial sum of the concatenated coupling
 * vector [sum_f n_f]; ONE in-place sum all-reduce per Jacobi iteration completes it (0.97 MB for the
 * 4 x 30 301 stack of BASELINE config 5: a single latency-bound ring pass over xGMI).
 *
 *   ssa_coupling_allreduce   buf <- sum over ranks of buf, in place, on `stream` (enqueue only).
 *       buf [count] dtype (device), rccl_comm: an ncclComm_t (from ssa_rccl_comm_create, or any
 *       communicator the caller already owns: RCCL's own handle type, passed as void*).
 *
 * Communicator helpers for callers that have no RCCL binding of their own (one process per GPU; the
 * 128-byte id travels from rank 0 to the others by whatever channel the caller has -- MPI, a file,
 * torch.distributed's store):
 *   ssa_rccl_unique_id      id_out: 128 bytes of HOST memory                    (ncclGetUniqueId)
 *   ssa_rccl_comm_create    collective over all ranks; uses the current device  (ncclCommInitRank)
 *   ssa_rccl_comm_destroy                                                       (ncclCommDestroy)
 * librccl.so is opened on first use (dlopen), so the library loads on machines without RCCL; these
 * calls then return SSA_ERR_RCCL.
 */
#define SSA_RCCL_UNIQUE_ID_BYTES 128
int ssa_rccl_unique_id(void *id_out);
int ssa_rccl_comm_create(void **comm_out, int nranks, int rank, const void *id);
int ssa_rccl_comm_destroy(void *comm);
int ssa_coupling_allreduce(void *buf, int64_t count, int dtype, void *rccl_comm, void *stream);

/*
 * Releases what the library created behind the caller's back on every device of the process: the
 * look-ahead side streams and events of the factorization schedules (created on first use, reused by
 * every later call) and the events of ssa_profile_*.  Waits for the side streams first.  The library
 * stays usable: the next factorization creates its lanes again.
 */
int ssa_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* SUPERSCREEN_HIP_H */
