// Internal launchers of rom_ops.hip shared with rom_basis.hip: raw device pointers, work only ENQUEUED on the context's
// compute stream (no host synchronisation), results left on the device.
#pragma once
#include "romhc_internal.h"

struct StencilGeom {
  int nr, nc, N, ncb, kblk;
  long long dim;
};
StencilGeom rom_make_geom(int nrb, int ncb, int N);

// C[m,n] = alpha * A[m,k] B[k,n] + beta C
int rom_launch_gemm_nn(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
// C[m,m] = A A^T (lower tiles on MFMA + mirror)
int rom_launch_gram(rom_ctx* ctx, int64_t m, int64_t k, const double* A, int64_t lda, double* C, int64_t ldc);
// Y[k,:] = A(coef) X[k,:]; d_coef: kblk block coefficients ON THE DEVICE, or null for the unit operator A_1
int rom_launch_stencil_apply(rom_fem* f, const double* d_coef, const double* X, int K, double* Y);
// the same on the interior mesh rows [row_lo, row_hi] only (Y is not written outside the slabs that cover them)
int rom_launch_stencil_apply_band(rom_fem* f, const double* d_coef, const double* X, int K, double* Y, int row_lo, int row_hi);
// Y[b, :] = A_b x for all kblk blocks in one launch (d_onehot: kblk x kblk identity on the device)
int rom_launch_stencil_apply_blocks(rom_fem* f, const double* d_onehot, const double* x, double* Y);
// d_out[k] = ||U_k - V_k||_{H10} (V may be null); squared norms if !take_sqrt.  Same kernels, same bits as rom_h10norm.
int rom_launch_h10norm(rom_fem* f, const double* U, const double* V, int K, double* d_out, bool take_sqrt);
// d_out[k] = ||U_k||_2 (or its square)
int rom_launch_l2norm(rom_ctx* ctx, const double* U, int K, int64_t dim, double* d_out, bool take_sqrt);
// d_out[k] = U_k . z  (rows of length dim against one vector)
int rom_launch_rowdot(rom_ctx* ctx, const double* U, int K, int64_t dim, const double* z, double* d_out);
// batched reduced solves, Ahat (kb, ldA, ldA) of which the leading n x n blocks are used; no status read-back
// (a non-positive pivot sets bit 0 of ctx->d_status, which the caller clears before and reads after)
int rom_launch_reduced_solve(rom_ctx* ctx, int n, int ldA, int kb, int M, const double* Ahat, const double* w,
                             const double* rhs, int rhs_per_system, double* c_out);
// X[row, :] *= fac[row] with the factors on the device
int rom_launch_rows_scale(rom_ctx* ctx, double* X, int rows, int64_t dim, const double* d_fac);
int rom_launch_center_rows(rom_ctx* ctx, double* X, int M, int64_t dim, double* d_mean);
int rom_launch_subtract_row(rom_ctx* ctx, double* X, int M, int64_t dim, const double* d_row);
int rom_launch_rows_sign_flip(rom_ctx* ctx, double* X, int rows, int64_t dim);
