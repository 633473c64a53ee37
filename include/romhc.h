/*
 * romhc.h -- C-ABI of libromhc.so, the MI355X (gfx950) implementation of the
 * ROMHighContrast snapshot-generation + reduced-basis hot path.
 *
 * The reference (agussomacal/ROMHighContrast) has no FFI layer: its boundary for this path is
 * the Python API of src/lib/SolutionsManagers.py and src/lib/ReducedBasis.py.  Each entry point
 * below names the reference interface it replaces (file:line, relative to the reference tree).
 * The Python shim (romhighcontrast_amd/lib, re-exported as src.lib / lib) binds these
 * with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; rom_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure on this thread.
 *   - all floating point is IEEE fp64.  Bulk data lives in device buffers (rom_buf*), created
 *     and destroyed explicitly; host arrays are caller-owned and only touched by
 *     rom_buf_upload / rom_buf_download and the few *_host helpers.
 *   - snapshot matrices are row-major (n_rows, dim): row = one FE vector over the inner
 *     vertices in row-major (r, c) order -- exactly the reference's `solutions` arrays.
 *   - parameters `a` are row-major (M, nrb*ncb): a[m][p*ncb+q], p = block row (y), q = block
 *     column (x) -- the reference's a[m][p][q].
 *   - all work is enqueued on the context's HIP stream; functions that return host data
 *     synchronise that stream first.  One context per process per GPU.
 */
#ifndef ROMHC_H
#define ROMHC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rom_ctx rom_ctx; /* one GPU + stream + workspace            */
typedef struct rom_buf rom_buf; /* fp64 device buffer                      */
typedef struct rom_fem rom_fem; /* FE space of one (blocks_geometry, N)    */

#define ROM_OK 0
#define ROM_ERR_INVALID 1  /* bad argument                                              */
#define ROM_ERR_HIP 2      /* HIP runtime failure (message holds hipGetErrorString)     */
#define ROM_ERR_NOT_SPD 3  /* a pivot was <= 0: maps to scipy.linalg.LinAlgError         */
#define ROM_ERR_COMM 4     /* RCCL failure / librccl not loadable                        */
#define ROM_ERR_NOMEM 5

const char* rom_last_error(void);
int rom_version(void);

/* ---- context ------------------------------------------------------------------------- */
int rom_device_count(int* n);
int rom_init(int device, rom_ctx** out);
int rom_shutdown(rom_ctx* ctx);
int rom_synchronize(rom_ctx* ctx);
/* workspace budget (bytes) for the factor storage of rom_solve_batch; default 24 GiB */
int rom_set_workspace_limit(rom_ctx* ctx, size_t bytes);
int rom_device_name(rom_ctx* ctx, char* out, size_t cap);

/* HIP-event stopwatch on the context stream (bench.py's timed region) */
int rom_timer_start(rom_ctx* ctx);
int rom_timer_stop(rom_ctx* ctx, double* elapsed_ms);

/* per-kernel HIP-event profiling: when enabled every kernel launch of the library is
 * bracketed by an event pair on the launch stream; rom_profile_query sums them by name.
 * While it is enabled rom_solve_batch* keeps a sweep on ONE stream (otherwise geometries whose
 * reduced solve is the tile Cholesky run as two concurrent sub-batches, ROMHC_STREAMS), so
 * that a bracket times its kernel alone. */
int rom_profile_enable(rom_ctx* ctx, int on);
int rom_profile_reset(rom_ctx* ctx);
int rom_profile_count(rom_ctx* ctx, int* n_kernels);
int rom_profile_query(rom_ctx* ctx, int idx, char* name, size_t cap, double* total_ms, long* launches,
                      double* flops, double* bytes);

/* ---- device buffers ------------------------------------------------------------------- */
int rom_buf_alloc(rom_ctx* ctx, size_t n_doubles, rom_buf** out);
int rom_buf_free(rom_buf* b);
int rom_buf_size(rom_buf* b, size_t* n_doubles);
int rom_buf_upload(rom_buf* b, size_t offset, const double* host, size_t n);
int rom_buf_download(rom_buf* b, size_t offset, double* host, size_t n);
/* Page-locked host arrays for large results (process-wide pool, thread safe): generate_solutions
 * (src/lib/SolutionsManagers.py:64-68) returns the (M, dim) rows to the host, and rom_buf_download into such an
 * array runs at the PCIe rate (48 vs 11-25 GB/s) -- but pinning 533 MB takes 110 ms, so the Python shim only pins for
 * sizes it has seen repeatedly and otherwise takes what the pool has; the NumPy arrays it wraps around the blocks
 * give them back when they are collected. */
/* pooled_only != 0: hand out a block of the pool or *out = NULL (status 0), never pin new memory */
int rom_host_alloc(size_t n, int pooled_only, double** out);
int rom_host_free(double* p);
int rom_buf_fill(rom_buf* b, size_t offset, size_t n, double value);
int rom_buf_copy(rom_buf* dst, size_t dst_off, rom_buf* src, size_t src_off, size_t n);
/* *equal_host = 1 if the n doubles A[a_off ...] and B[b_off ...] have the same bits (how the Python layer checks that rows
 * handed back to it are still the image of the interface vectors it kept for them), else 0 */
int rom_buf_equal(rom_buf* A, size_t a_off, rom_buf* B, size_t b_off, size_t n, int* equal_host);
/* dst[i, :] = src[rows[i], :] for row length `dim` (host index list; used by the greedy) */
int rom_buf_gather_rows(rom_buf* dst, rom_buf* src, const int64_t* rows, int n_rows, size_t dim);

/* ---- FE space: SolutionsManagerFEM.__init__ (src/lib/SolutionsManagers.py:146-219) ------ */
/* Builds the parameter-independent tables of the substructured operator on the device
 * (unit-block sine basis, harmonic-extension matrix, Dirichlet-to-Neumann blocks, symbolic
 * tile Cholesky of the interface system).  Replaces the dense A_preassembled tensor. */
int rom_fem_create(rom_ctx* ctx, int nrb, int ncb, int N, rom_fem** out);
int rom_fem_destroy(rom_fem* fem);
int rom_fem_dims(rom_fem* fem, int* nr, int* nc, int64_t* dim, int* n_interface, int* n_tiles);
/* B_total (:177-185): dim doubles, host */
int rom_fem_load_vector_host(rom_fem* fem, double* B_out);

/* einsum('pqij,pq->ij') in stencil form (:19-23 / :187-215): for each of M parameters the
 * three stencil arrays diag[M,nr,nc], east[M,nr,nc-1], north[M,nr-1,nc]. */
int rom_assemble_batch(rom_fem* fem, rom_buf* a, int M, rom_buf* diag, rom_buf* east, rom_buf* north);

/* generate_solutions (:64-68) = map(galerkin (:17-40)) : U[row0+m, :] = A(a_m)^{-1} B_total.
 * Exact direct method (interface Schur complement + tile Cholesky + harmonic extension). */
int rom_solve_batch(rom_fem* fem, rom_buf* a, int M, rom_buf* U, int64_t row0);
/* The same sweep, only enqueued (no host synchronisation); a non-positive pivot -- the reference's
 * scipy LinAlgError case -- is latched on the device and reported by the next rom_solve_status()
 * (or rom_solve_batch()), which waits for the compute stream. */
int rom_solve_batch_async(rom_fem* fem, rom_buf* a, int M, rom_buf* U, int64_t row0);
int rom_solve_status(rom_ctx* ctx);
/* The sweep in two stages, for the multi-GPU exchange: a snapshot row is a fixed linear image of
 * its system's "interface vector" (reduced unknowns + coefficient blocks, rom_fem_reduced_stride() doubles:
 * 784 instead of 65,025 at 256x256 / 2x2).  Ranks all-gather the interface vectors and every rank expands all
 * of them; the expansion is deterministic, so the gathered snapshot block is bit-identical on every rank.
 * Both calls only enqueue work on the compute stream (rom_solve_status() reports a non-positive pivot). */
int rom_fem_reduced_stride(rom_fem* fem, int64_t* stride);
/* positions [nodal_begin, nodal_end) of an interface vector are outputs of the expansion (nodal edge values);
 * only the rest is read by it */
int rom_fem_reduced_layout(rom_fem* fem, int64_t* nodal_begin, int64_t* nodal_end);
/* 1 if rom_expand_batch_async is a LINEAR map of the interface vectors (it then ignores `a`): U = Y B^T with a fixed
 * B, so Gram matrices, means and POD modes of snapshots can be formed from Y alone (every geometry whose
 * closed-form edges are all kept in compressed form, e.g. 2x2/N>=16, 3x3/N=171, 4x4/N=256). */
int rom_fem_expansion_is_linear(rom_fem* fem, int* flag);
/* Compact form for the exchange (the all-gather before the SVD): the entries of an interface vector outside
 * its nodal part, rom_fem_compact_stride() doubles (272 of 784 at 256x256 / 2x2) -- all that has to travel, since the
 * expansion recomputes the nodal part.  pack: Yc[c_row0+m] <- Y[y_row0+m]; unpack: the inverse (nodal part zeroed).
 * Both only enqueue on the compute stream. */
int rom_fem_compact_stride(rom_fem* fem, int64_t* stride);
int rom_fem_pack_reduced_async(rom_fem* fem, rom_buf* Y, int64_t y_row0, int M, rom_buf* Yc, int64_t c_row0);
int rom_fem_unpack_reduced_async(rom_fem* fem, rom_buf* Yc, int64_t c_row0, int M, rom_buf* Y, int64_t y_row0);
int rom_solve_reduced_async(rom_fem* fem, rom_buf* a, int M, rom_buf* Y, int64_t y_row0);
int rom_expand_batch_async(rom_fem* fem, rom_buf* a, int M, rom_buf* Y, int64_t y_row0, rom_buf* U, int64_t row0);
/* flops / HBM bytes of the library's own algorithm for one snapshot solve, and the canonical
 * banded-Cholesky figures of SURVEY.md 8(d) for comparison */
int rom_solve_work(rom_fem* fem, double* flops_own, double* bytes_own, double* flops_banded,
                   double* bytes_banded);

/* Y[k,:] = A(coef) X[k,:], K rows.  mode 0: coef = a_one (k doubles, host) general blocks;
 * mode 1: unit coefficient (A_preassembled4h1_norm, :49).  (the C A_pq contractions, :93-101) */
int rom_stencil_apply(rom_fem* fem, const double* a_one_host, int unit, rom_buf* X, int64_t x_row0,
                      int K, rom_buf* Y, int64_t y_row0);

/* H10norm (:56-58): out[k] = sqrt(u_k^T A_1 u_k); out_host has K doubles.
 * If V != NULL computes the norm of (U[u_row0+k] - V[v_row0+k]) instead (greedy residuals,
 * src/lib/ReducedBasis.py:129). */
int rom_h10norm(rom_fem* fem, rom_buf* U, int64_t u_row0, rom_buf* V, int64_t v_row0, int K,
                double* out_host);
/* l2norm (:60-62) */
int rom_l2norm(rom_ctx* ctx, rom_buf* U, int64_t row0, int K, int64_t dim, double* out_host);

/* ---- dense fp64 contractions on MFMA (v_mfma_f64_16x16x4_f64) --------------------------- */
/* C[m,n] = alpha * sum_k A[m,k] B[n,k] + beta*C   (row-major, "NT": Gram / C A U^T) */
int rom_gemm_nt(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, rom_buf* A, size_t a_off,
                int64_t lda, rom_buf* B, size_t b_off, int64_t ldb, double beta, rom_buf* C, size_t c_off,
                int64_t ldc);
/* G[m,m] = A A^T (row-major A[m,k]): the snapshot Gram matrix of the POD; only the lower tiles are
 * computed on MFMA, the strict upper triangle is mirrored */
int rom_gram(rom_ctx* ctx, int64_t m, int64_t k, rom_buf* A, size_t a_off, int64_t lda, rom_buf* C, size_t c_off,
             int64_t ldc);
/* C[m,n] = alpha * sum_k A[m,k] B[k,n] + beta*C   (row-major, "NN": lift c_i . basis, :106/:139) */
int rom_gemm_nn(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, rom_buf* A, size_t a_off,
                int64_t lda, rom_buf* B, size_t b_off, int64_t ldb, double beta, rom_buf* C, size_t c_off,
                int64_t ldc);

/* batched reduced solves: for m<M: (sum_b w[m,b] * Ahat[b]) c_m = rhs[m or 0]  (n x n SPD).
 * Ahat: (kb, n, n); w: (M, kb); rhs: (M, n) if rhs_per_system else (n); out c: (M, n).
 * The reduced `galerkin` calls of generate_fm_solutions (:104-105) and project_solutions
 * (:135-138). */
int rom_reduced_solve_batch(rom_ctx* ctx, int n, int kb, int M, rom_buf* Ahat, rom_buf* w, rom_buf* rhs,
                            int rhs_per_system, rom_buf* c_out);

/* ---- helpers of the basis builders ------------------------------------------------------ */
int rom_buf_scale(rom_buf* b, size_t offset, size_t n, double alpha);
/* X[row0+m, :] -= mean over m (column means, kept in `mean`, dim doubles): the centring step of
 * sklearn PCA.fit called at src/lib/ReducedBasis.py:196 */
int rom_center_rows(rom_ctx* ctx, rom_buf* X, int64_t row0, int M, int64_t dim, rom_buf* mean);
/* X[row0+i, :] *= factors_host[i], i < rows: the 1/sigma scaling of the lifted POD modes (components_ of the PCA at
 * src/lib/ReducedBasis.py:196-197 have unit norm) */
int rom_rows_scale(rom_ctx* ctx, rom_buf* X, int64_t row0, int rows, int64_t dim, const double* factors_host);
/* every row times the sign of its entry of largest magnitude: sklearn's svd_flip(u_based_decision=False) inside the
 * same PCA call, which fixes the signs of `components_` */
int rom_rows_sign_flip(rom_ctx* ctx, rom_buf* X, int64_t row0, int rows, int64_t dim);
/* evaluate_solutions (src/lib/SolutionsManagers.py:221-244): P1 interpolation of K FE vectors at
 * npts points.  ix/iy = cell index of each point (searchsorted(points_c/points_r) - 1), tx/ty its
 * local coordinates in the cell; out_host is (K, npts). */
int rom_evaluate_points(rom_fem* fem, rom_buf* U, int64_t row0, int K, int npts, const int* ix_host,
                        const int* iy_host, const double* tx_host, const double* ty_host, double* out_host);

/* ---- the basis stage as single calls (rom_project_h10, rom_galerkin_rom, rom_greedy, rom_pod) --------
 * Each call enqueues all its kernels on the context's stream and waits for it once at its end (status word /
 * results); small dense problems (Gram matrices of the basis, projected eigenproblems, argmax of the greedy) are solved
 * on the device.  rom_pod additionally reads a few dozen spectrum values per pass to decide how many modes to accept. */
/* project_solutions (src/lib/SolutionsManagers.py:108-139): OUT[out_row0+m] = H^1_0-orthogonal projection of
 * U[u_row0+m] onto the span of the n rows C[c_row0 ...] (any full-rank rows; n = 0: zeros, :109-111).
 * ROM_ERR_NOT_SPD if C A_1 C^T is not positive definite (dependent rows). */
int rom_project_h10(rom_fem* fem, rom_buf* U, int64_t u_row0, int M, rom_buf* C, int64_t c_row0, int n, rom_buf* OUT,
                    int64_t out_row0);
/* generate_fm_solutions (:88-106): OUT[out_row0+m] = Galerkin reduced-order solution for a[m] (M x nrb*ncb) in the span
 * of the n rows of C (n = 0: zeros, :89-91) */
int rom_galerkin_rom(rom_fem* fem, rom_buf* a, int M, rom_buf* C, int64_t c_row0, int n, rom_buf* OUT, int64_t out_row0);
/* orthonormalize_base (src/lib/ReducedBasis.py:18-21): rows of X -> Euclidean-orthonormal rows of Q spanning the same
 * nested subspaces (the thin QR at :19 up to the sign of each row; a dependent row becomes zero).  Q may be X. */
int rom_orthonormalize_rows(rom_ctx* ctx, rom_buf* X, int64_t x_row0, int n, int64_t dim, rom_buf* Q, int64_t q_row0);
/* ReducedBasisGreedy.build (src/lib/ReducedBasis.py:112-139): strong greedy over the M training snapshots U[u_row0 ...]
 * in relative H^1_0 error; mode 0 = error of the H^1_0 projection (:122), 1 = error of the Galerkin ROM (:124, needs
 * the training parameters a, M x nrb*ncb).  h1norm_host: the M normalisations (solutions2train_h1norm, :129).
 * picks_out[i] = training index chosen in iteration i (first maximum, like np.argmax), max_err_out[i] = its relative
 * error, i < n.  Iteration 0 has the empty basis: with h1norm = rom_h10norm(U) every error is exactly 1.0 and the
 * pick is index 0, as in the reference. */
int rom_greedy(rom_fem* fem, rom_buf* U, int64_t u_row0, int M, rom_buf* a, const double* h1norm_host, int mode, int n,
               int64_t* picks_out, double* max_err_out);
/* PCA(n_components = n).fit (src/lib/ReducedBasis.py:196): leading n right singular vectors of the (M, dim) block
 * X[x_row0 ...] -- OVERWRITTEN when center != 0 (the column means are subtracted in place) -- into V[v_row0 ...] (n x dim, orthonormal rows,
 * scikit-learn's svd_flip(u_based_decision=False) signs) and their singular values into sigma_host (n).  Randomised
 * range-finder passes over the implicitly deflated block (thin products 2 b M dim, one power step, rows orthonormalised on
 * both sides of it: a pass resolves modes over seven orders of magnitude to LAPACK's own noise bound eps sigma_1 / sigma)
 * with a convergence rule per pass; when the first pass shows a spectrum that decays too slowly for that (its modes do not
 * separate from what lies beyond the sketch) the leading modes come from the M x M Gram matrix instead (MFMA, eigenpairs
 * iterated to convergence in M space) and the passes continue below its reach (1e-5 sigma_1).  Rayleigh-Ritz over the
 * collected modes; modes below 1e-13 sigma_1 do not exist in fp64 data and are completed with orthonormal directions of
 * singular value 0.  info_host (8 doubles or NULL): resolved modes, completed modes, Gram passes (0 or 1), sketch passes,
 * executed flops, 8 n M dim (the four thin products of a pass for the n modes alone), subspace iterations of the Gram
 * route, stop reason (0: all n modes resolved; 1: the spectrum reached the floor -- the completed modes are not determined
 * by the data; 2: a pass accepted nothing although the floor was not reached -- modes above it may be missing). */
int rom_pod(rom_ctx* ctx, rom_buf* X, int64_t x_row0, int M, int64_t dim, int n, int center, rom_buf* V, int64_t v_row0,
            double* sigma_host, double* info_host);
/* the same with the floor chosen by the caller: modes with sigma <= rel_floor * sigma_1 are not looked for (every sketch
 * pass over the block buys seven orders of magnitude; a reduced basis that is used to 1e-6 needs one pass).
 * rel_floor <= 1e-13 is rom_pod. */
int rom_pod_ex(rom_ctx* ctx, rom_buf* X, int64_t x_row0, int M, int64_t dim, int n, int center, double rel_floor, rom_buf* V,
               int64_t v_row0, double* sigma_host, double* info_host);
/* ---- the basis stage on a snapshot block held in FACTORED form ---------------------------------------------------------
 * A sweep gathered from several GPUs exists on every rank as interface vectors, not as rows (rom_comm_allgather_packed_
 * async); when rom_fem_expansion_is_linear() the rows are U = Y B^T with a fixed B, so the builders below work on the
 * (M, Kc) block of COMPACT interface vectors (Kc = rom_fem_compact_stride(); rom_fem_pack_reduced_async makes them) and
 * never form a row except the basis vectors they return.  Same reference bodies as the row calls above
 * (src/lib/ReducedBasis.py:112-139, :189-200); same contracts, same picks / modes up to rounding.
 * rom_fem_energy_map builds (once per FE space, cached on the fem) the geometry of the snapshots in those coordinates:
 * parts 1 = H^1_0 inner product (norms, greedy), 2 = the block forms u^T A_b v and the load functional (Galerkin greedy),
 * 4 = Euclidean inner product (POD); the other calls build what they need themselves.  k_h10 / k_l2 (may be NULL): ranks. */
int rom_fem_energy_map(rom_fem* fem, int parts, int* k_h10, int* k_l2);
/* H10norm (src/lib/SolutionsManagers.py:56-58) of M snapshots from their compact interface vectors Yc[c_row0 ...] */
int rom_h10norm_factored(rom_fem* fem, rom_buf* Yc, int64_t c_row0, int M, double* out_host);
/* rom_greedy on compact interface vectors (M x Kc); a: M x nrb*ncb parameters (mode 1) */
int rom_greedy_factored(rom_fem* fem, rom_buf* Yc, int64_t c_row0, int M, rom_buf* a, const double* h1norm_host, int mode,
                        int n, int64_t* picks_out, double* max_err_out);
/* rom_pod on compact interface vectors (M x Kc, not modified): V[v_row0 ...] receives the n modes as ROWS (n x dim) */
int rom_pod_factored(rom_fem* fem, rom_buf* Yc, int64_t c_row0, int M, int n, int center, rom_buf* V, int64_t v_row0,
                     double* sigma_host, double* info_host);
/* n nearly orthonormal rows of V -> orthonormal rows, each as close as possible to what it was: V <- (V V^T)^(-1/2) V */
int rom_symmetric_orthonormalize(rom_ctx* ctx, rom_buf* V, int64_t v_row0, int n, int64_t dim);
/* rows V[v_row0+found .. +found+rest) <- deterministic pseudo-random directions, orthonormal and orthogonal to the
 * orthonormal rows V[v_row0 .. +found): how rom_pod completes a request beyond what the data determine */
int rom_complete_orthonormal(rom_ctx* ctx, rom_buf* V, int64_t v_row0, int found, int rest, int64_t dim);
/* the device eigen-solver the calls above use for their small symmetric problems (cyclic Jacobi, one workgroup), for
 * n x n host matrices, n <= 1024: mode 0 T = eigenvector rows (eigenvalues descending in lam_host); 1 T = whitening
 * transform Lambda^-1/2 Q^T (rows with lambda <= rel_tol lambda_max zero); 2 T = Q Lambda^-1/2 Q^T; 3 (n <= 96) T =
 * the rank-revealing whitening transform [L_r^-1 0] P of the pivoted Cholesky factorisation P A P^T = L L^T (lam_host:
 * squared pivots), what the orthonormalisations of rom_pod use.
 * gram_like != 0: A is a Gram matrix of explicit rows (entries accurate relative to sqrt(a_pp a_qq): small eigenvalues
 * of graded matrices come out to high relative accuracy); 0: general symmetric matrix (absolute accuracy).  Test hook. */
int rom_small_eig_host(rom_ctx* ctx, int n, const double* A_host, int mode, double rel_tol, int gram_like,
                       double* lam_host, double* T_host);

/* ---- multi-GPU: RCCL all-gather of the snapshot block --------------------- */
/* id_out: 128 bytes (ncclUniqueId).  librccl is dlopen()ed on first use. */
int rom_comm_unique_id(char* id_out, size_t cap);
int rom_comm_init(rom_ctx* ctx, const char* id, size_t id_len, int rank, int nranks);
int rom_comm_destroy(rom_ctx* ctx);
/* recv[(r*count) ...] = send of rank r; count doubles per rank */
int rom_comm_allgather(rom_ctx* ctx, rom_buf* send, size_t send_off, rom_buf* recv, size_t recv_off,
                       size_t count);
/* same collective on the context's communication stream, ordered after the work enqueued so far on the
 * compute stream but not blocking it (the next sweep step overlaps the exchange); rom_comm_wait makes the
 * compute stream (and the host if host_sync != 0) wait for the outstanding collectives */
int rom_comm_allgather_async(rom_ctx* ctx, rom_buf* send, size_t send_off, rom_buf* recv, size_t recv_off,
                             size_t count, int slot /* 0|1: double-buffer slot of the send buffer */);
/* The exchange of one step of a sharded sweep in ONE call: pack the interface vectors Y[y_row0 .. +M) of the own shard
 * into their compact form (into `send`, M x rom_fem_compact_stride() doubles of scratch) and all-gather them into
 * recv[recv_off ...] (nranks x M x compact stride) -- both on the communication stream, after everything enqueued so
 * far on the compute stream, which is not blocked.  Slots as in rom_comm_allgather_async. */
int rom_comm_allgather_packed_async(rom_fem* fem, rom_buf* Y, int64_t y_row0, int M, rom_buf* send, rom_buf* recv,
                                    size_t recv_off, int slot);
int rom_comm_wait(rom_ctx* ctx, int host_sync);
/* compute stream waits for the collective last issued with `slot` (before its send buffer is rewritten) */
int rom_comm_wait_slot(rom_ctx* ctx, int slot);
/* in-place max / sum all-reduce of n host doubles staged through the device (control plane:
 * barrier + max-over-ranks timing of bench.py) */
int rom_comm_allreduce_host(rom_ctx* ctx, double* vals, int n, int op /*0=sum,1=max*/);

#ifdef __cplusplus
}
#endif
#endif /* ROMHC_H */
