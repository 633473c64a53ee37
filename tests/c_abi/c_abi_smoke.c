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
  free(B); free(r); free(u);
  rom_buf_free(X); rom_buf_free(C); rom_buf_free(P); rom_buf_free(V); rom_buf_free(AU); rom_buf_free(U); rom_buf_free(a_d);
  CK(rom_fem_destroy(fem));
  CK(rom_shutdown(ctx));
  printf("OK\n");
  return 0;
}
