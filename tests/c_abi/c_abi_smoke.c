/* A plain C caller of the C-ABI (no Python, no C++): what a maintainer binding libromhc from another host language sees.
 * Builds with  gcc -std=c99 -I include tests/c_abi/c_abi_smoke.c -L romhighcontrast_amd/csrc -lromhc -lm
 * `c_abi_smoke --symbols` only touches entry points that need no GPU (used by the CPU test suite);
 * without arguments it runs, on device 0: FE space (2 x 2 blocks, N = 8) -> sweep of M = 5 parameters -> H^1_0 norms ->
 * strong greedy (n = 3) -> POD (2 modes) -> H^1_0 projection onto the picked snapshots, and checks identities that hold
 * for the exact result whatever the arithmetic path:
 *   - every snapshot satisfies the stencil equation A(a_m) u_m = B (rom_stencil_apply with the row's own coefficients);
 *   - the greedy's first pick is row 0 with relative error exactly 1.0, its errors are non-increasing;
 *   - a picked snapshot is reproduced by its own projection (relative H^1_0 error < 1e-11);
 *   - POD modes are orthonormal and sigma_0 >= sigma_1 > 0.
 *   - (2 x 2 blocks, N = 48, where snapshots are a linear image of their interface vectors) the basis stage on the factored
 *     block agrees with the one on rows: rom_h10norm_factored with rom_h10norm (1e-11), rom_greedy_factored with rom_greedy
 *     (same picks, errors to 1e-10), rom_pod_factored with rom_pod (singular values to 1e-9 sigma_0).
 * (the parity tests proper -- against the oracle -- are tests/test_gpu_parity.py) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "romhc.h"

#define CK(call)                                                                          \
  do {                                                                                    \
    int st_ = (call);                                                                     \
    if (st_) {                                                                            \
      fprintf(stderr, "%s -> %d: %s\n", #call, st_, rom_last_error());                    \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)
#define REQUIRE(cond, ...)                                                                \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      fprintf(stderr, "FAILED %s: ", #cond);                                              \
      fprintf(stderr, __VA_ARGS__);                                                       \
      fprintf(stderr, "\n");                                                              \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

int main(int argc, char** argv) {
  printf("libromhc version %d\n", rom_version());
  if (argc > 1 && strcmp(argv[1], "--symbols") == 0) {
    int n = -1;
    int st = rom_device_count(&n); /* may fail without a GPU: it must then say why */
    printf("rom_device_count -> %d (%d devices)%s%s\n", st, n, st ? ": " : "", st ? rom_last_error() : "");
    return 0;
  }
  enum { NRB = 2, NCB = 2, N = 8, M = 5, K = NRB * NCB, NPICK = 3, NMODES = 2 };
  rom_ctx* ctx = NULL;
  rom_fem* fem = NULL;
  CK(rom_init(0, &ctx));
  CK(rom_fem_create(ctx, NRB, NCB, N, &fem));
  int nr, nc, ni, nt;
  int64_t dim;
  CK(rom_fem_dims(fem, &nr, &nc, &dim, &ni, &nt));
  REQUIRE(nr == NRB * N - 1 && nc == NCB * N - 1 && dim == (int64_t)nr * nc, "dims %d x %d, dim %lld", nr, nc, (long long)dim);

  double a[M * K];
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) a[m * K + k] = 1.0 + 9.0 * fabs(sin(1.0 + 3.0 * m + 7.0 * k)); /* 1 .. 10 */
  rom_buf *a_d, *U, *AU, *V, *P, *C;
  CK(rom_buf_alloc(ctx, M * K, &a_d));
  CK(rom_buf_alloc(ctx, (size_t)M * dim, &U));
  CK(rom_buf_alloc(ctx, (size_t)dim, &AU));
  CK(rom_buf_alloc(ctx, (size_t)NMODES * dim, &V));
  CK(rom_buf_alloc(ctx, (size_t)M * dim, &P));
  CK(rom_buf_alloc(ctx, (size_t)NPICK * dim, &C));
  CK(rom_buf_upload(a_d, 0, a, M * K));
  CK(rom_solve_batch(fem, a_d, M, U, 0));

  /* residual of the stencil equation, row by row */
  double* B = (double*)malloc(sizeof(double) * dim);
  double* r = (double*)malloc(sizeof(double) * dim);
  double* u = (double*)malloc(sizeof(double) * dim);
  CK(rom_fem_load_vector_host(fem, B));
  for (int m = 0; m < M; ++m) {
    CK(rom_stencil_apply(fem, a + m * K, 0, U, m, 1, AU, 0));
    CK(rom_buf_download(AU, 0, r, dim));
    CK(rom_buf_download(U, (size_t)m * dim, u, dim));
    double worst = 0.0, umax = 0.0, bmax = 0.0;
    for (int64_t i = 0; i < dim; ++i) {
      worst = fmax(worst, fabs(r[i] - B[i]));
      umax = fmax(umax, fabs(u[i]));
      bmax = fmax(bmax, fabs(B[i]));
    }
    REQUIRE(worst <= 1e-11 * (40.0 * umax + bmax), "row %d: |A u - B|_inf = %.3e (|u|_inf %.3e, |B|_inf %.3e)", m, worst, umax, bmax);
  }

  double h1[M];
  CK(rom_h10norm(fem, U, 0, NULL, 0, M, h1));
  for (int m = 0; m < M; ++m) REQUIRE(h1[m] > 0.0 && isfinite(h1[m]), "H10 norm of row %d = %g", m, h1[m]);

  int64_t picks[NPICK];
  double err[NPICK];
  CK(rom_greedy(fem, U, 0, M, NULL, h1, 0, NPICK, picks, err));
  REQUIRE(picks[0] == 0 && err[0] == 1.0, "first pick %lld with error %.17g", (long long)picks[0], err[0]);
  for (int i = 1; i < NPICK; ++i) {
    REQUIRE(picks[i] >= 0 && picks[i] < M && err[i] <= err[i - 1] * (1.0 + 1e-12), "pick %d = %lld, error %g after %g", i, (long long)picks[i], err[i], err[i - 1]);
    for (int j = 0; j < i; ++j) REQUIRE(picks[i] != picks[j] || err[i] < 1e-10, "row %lld picked twice", (long long)picks[i]);
  }

  /* projection onto the picked snapshots reproduces them */
  CK(rom_buf_gather_rows(C, U, picks, NPICK, (size_t)dim));
  CK(rom_project_h10(fem, U, 0, M, C, 0, NPICK, P, 0));
  double d[M];
  CK(rom_h10norm(fem, P, 0, U, 0, M, d));
  for (int i = 0; i < NPICK; ++i) REQUIRE(d[picks[i]] <= 1e-11 * h1[picks[i]], "picked row %lld: projection error %.3e", (long long)picks[i], d[picks[i]] / h1[picks[i]]);

  /* POD of a copy (rom_pod centres its input in place) */
  rom_buf* X;
  CK(rom_buf_alloc(ctx, (size_t)M * dim, &X));
  CK(rom_buf_copy(X, 0, U, 0, (size_t)M * dim));
  double sigma[NMODES], info[8];
  CK(rom_pod(ctx, X, 0, M, dim, NMODES, 1, V, 0, sigma, info));
  REQUIRE(sigma[0] >= sigma[1] && sigma[1] > 0.0, "sigma = %g, %g", sigma[0], sigma[1]);
  double* v0 = u;
  double* v1 = r;
  CK(rom_buf_download(V, 0, v0, dim));
  CK(rom_buf_download(V, (size_t)dim, v1, dim));
  double n0 = 0.0, n1 = 0.0, d01 = 0.0;
  for (int64_t i = 0; i < dim; ++i) {
    n0 += v0[i] * v0[i];
    n1 += v1[i] * v1[i];
    d01 += v0[i] * v1[i];
  }
  REQUIRE(fabs(n0 - 1.0) < 1e-12 && fabs(n1 - 1.0) < 1e-12 && fabs(d01) < 1e-12, "modes: |v0|^2 %.15g |v1|^2 %.15g v0.v1 %.3e", n0, n1, d01);

  printf("C-ABI smoke: %d snapshots of dim %lld, picks %lld %lld %lld, errors %.3e %.3e %.3e, sigma %.6e %.6e\n", M, (long long)dim,
         (long long)picks[0], (long long)picks[1], (long long)picks[2], err[0], err[1], err[2], sigma[0], sigma[1]);

  /* ---- the same basis stage on the FACTORED block: interface vectors instead of rows (what a multi-GPU sweep leaves on
   * every rank).  A geometry whose expansion is linear (edges kept in compressed form): 2 x 2 blocks, N = 48. */
  {
    enum { N2 = 48, M2 = 6 };
    rom_fem* fem2 = NULL;
    CK(rom_fem_create(ctx, NRB, NCB, N2, &fem2));
    int lin = 0, nr2, nc2, ni2, nt2;
    int64_t dim2, ky, kc;
    CK(rom_fem_expansion_is_linear(fem2, &lin));
    CK(rom_fem_dims(fem2, &nr2, &nc2, &dim2, &ni2, &nt2));
    CK(rom_fem_reduced_stride(fem2, &ky));
    CK(rom_fem_compact_stride(fem2, &kc));
    REQUIRE(lin == 1 && kc > 0 && kc <= ky, "expansion linear %d, strides %lld / %lld", lin, (long long)ky, (long long)kc);
    double a2[M2 * K];
    for (int m = 0; m < M2; ++m)
      for (int k = 0; k < K; ++k) a2[m * K + k] = 1.0 + 9.0 * fabs(sin(2.0 + 5.0 * m + 3.0 * k));
    rom_buf *a2_d, *U2, *Y2, *Yc, *V2, *X2, *V3;
    CK(rom_buf_alloc(ctx, M2 * K, &a2_d));
    CK(rom_buf_alloc(ctx, (size_t)M2 * dim2, &U2));
    CK(rom_buf_alloc(ctx, (size_t)M2 * ky, &Y2));
    CK(rom_buf_alloc(ctx, (size_t)M2 * kc, &Yc));
    CK(rom_buf_alloc(ctx, (size_t)NMODES * dim2, &V2));
    CK(rom_buf_alloc(ctx, (size_t)M2 * dim2, &X2));
    CK(rom_buf_alloc(ctx, (size_t)NMODES * dim2, &V3));
    CK(rom_buf_upload(a2_d, 0, a2, M2 * K));
    CK(rom_solve_batch(fem2, a2_d, M2, U2, 0));
    CK(rom_solve_reduced_async(fem2, a2_d, M2, Y2, 0));
    CK(rom_solve_status(ctx));
    CK(rom_fem_pack_reduced_async(fem2, Y2, 0, M2, Yc, 0));
    int k_h10 = 0, k_l2 = 0;
    CK(rom_fem_energy_map(fem2, 1 | 4, &k_h10, &k_l2));
    REQUIRE(k_h10 > 0 && k_h10 <= kc && k_l2 > 0 && k_l2 <= kc, "ranks of the energy map: %d, %d of %lld", k_h10, k_l2, (long long)kc);
    double hr[M2], hf[M2];
    CK(rom_h10norm(fem2, U2, 0, NULL, 0, M2, hr));
    CK(rom_h10norm_factored(fem2, Yc, 0, M2, hf));
    for (int m = 0; m < M2; ++m) REQUIRE(fabs(hf[m] - hr[m]) <= 1e-11 * hr[m], "H10 norm of snapshot %d: %.15g from rows, %.15g factored", m, hr[m], hf[m]);
    int64_t pr[NPICK], pf[NPICK];
    double er[NPICK], ef[NPICK];
    CK(rom_greedy(fem2, U2, 0, M2, NULL, hr, 0, NPICK, pr, er));
    CK(rom_greedy_factored(fem2, Yc, 0, M2, NULL, hf, 0, NPICK, pf, ef));
    for (int i = 0; i < NPICK; ++i)
      REQUIRE(pr[i] == pf[i] && fabs(er[i] - ef[i]) <= 1e-10, "greedy step %d: row %lld (error %.3e) on rows, %lld (%.3e) factored", i, (long long)pr[i], er[i], (long long)pf[i], ef[i]);
    double sr[NMODES], sf[NMODES];
    CK(rom_buf_copy(X2, 0, U2, 0, (size_t)M2 * dim2));
    CK(rom_pod(ctx, X2, 0, M2, dim2, NMODES, 1, V2, 0, sr, NULL));
    CK(rom_pod_factored(fem2, Yc, 0, M2, NMODES, 1, V3, 0, sf, NULL));
    for (int i = 0; i < NMODES; ++i) REQUIRE(fabs(sr[i] - sf[i]) <= 1e-9 * sr[0], "singular value %d: %.15g on rows, %.15g factored", i, sr[i], sf[i]);
    printf("factored stage: %d snapshots of dim %lld from %lld-double interface vectors, energy ranks %d / %d, picks %lld %lld %lld, sigma %.6e %.6e\n",
           M2, (long long)dim2, (long long)kc, k_h10, k_l2, (long long)pf[0], (long long)pf[1], (long long)pf[2], sf[0], sf[1]);
    rom_buf_free(V3); rom_buf_free(X2); rom_buf_free(V2); rom_buf_free(Yc); rom_buf_free(Y2); rom_buf_free(U2); rom_buf_free(a2_d);
    CK(rom_fem_destroy(fem2));
  }
  free(B); free(r); free(u);
  rom_buf_free(X); rom_buf_free(C); rom_buf_free(P); rom_buf_free(V); rom_buf_free(AU); rom_buf_free(U); rom_buf_free(a_d);
  CK(rom_fem_destroy(fem));
  CK(rom_shutdown(ctx));
  printf("OK\n");
  return 0;
}
