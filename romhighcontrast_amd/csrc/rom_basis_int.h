// Pieces of the basis stage shared by rom_basis.hip and rom_factored.hip (internal).
#pragma once
#include <algorithm>

#include "rom_ops.h"

namespace {

// ---- temporaries from the context's caching allocator ---------------------------------------------------------------
struct Tmp {
  rom_buf* b = nullptr;
  Tmp() = default;
  Tmp(const Tmp&) = delete;
  Tmp& operator=(const Tmp&) = delete;
  ~Tmp() { release(); }
  void release() {
    if (b) rom_buf_free(b);
    b = nullptr;
  }
  int get(rom_ctx* ctx, size_t n) {
    release();
    return rom_buf_alloc(ctx, std::max<size_t>(n, 1), &b);
  }
  double* p() const { return b->p; }
  operator double*() const { return b->p; }
};

int read_status(rom_ctx* ctx, const char* who) {
  int status = 0;
  ROM_HIP(hipMemcpyAsync(&status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (status) {
    // (reported once: the word is cleared, or the next call on this context -- whatever it is -- would report this failure again)
    ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    rom_set_error("%s: reduced matrix not positive definite", who);
    return ROM_ERR_NOT_SPD;
  }
  return ROM_OK;
}

int download(rom_ctx* ctx, const double* d, double* h, size_t n) {
  if (n == 0) return ROM_OK;
  ROM_HIP(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  return ROM_OK;
}

}  // namespace

// kernels of rom_basis.hip that the factored greedy launches as well
__global__ void kb_greedy_select(int M, const double* __restrict__ err2, const double* __restrict__ extra2,
                                 const double* __restrict__ h1, int it, int* __restrict__ picks, double* __restrict__ maxerr);
__global__ void kb_grow_ahat(double* __restrict__ Ahat, int k, int ld, int j, const double* __restrict__ col,
                             const int* __restrict__ degenerate, int it);
__global__ void kb_galerkin_gap(int M, int n, const double* __restrict__ P, const double* __restrict__ c, double* __restrict__ extra2);
__global__ void kb_ints_to_doubles(const int* __restrict__ src, double* __restrict__ dst, int n);

// ---- small dense problems and orthonormalisation helpers of rom_basis.hip, shared with rom_pod.hip ------------------
enum { SE_EIG = 0, SE_WHITEN = 1, SE_LOWDIN = 2 };
constexpr int SE_LDS_MAX = 96, SE_MAX = 1024;
// (the grid-wide Jacobi takes larger orders -- four buffers of n^2 doubles, n launches per sweep: ~1 s at 2048 -- as the
// whole-matrix fallback of the POD's Gram route; SE_MAX stays the limit of the public entries and of the modes per request)
constexpr int SE_GRID_MAX = 2048;
__global__ void kb_rows_axpy(double* __restrict__ out, const double* __restrict__ x, const double* __restrict__ y,
                             const double* __restrict__ f, double s, long long dim);
int romb_fill_random(rom_ctx* ctx, double* p, size_t n, unsigned long long seed, bool gaussian);
int romb_small_eig(rom_ctx* ctx, int n, const double* A, int lda, double* lam, double* T, int ldt, int mode, double rel_tol,
                   bool gram_like = true);
int romb_pivchol_whiten(rom_ctx* ctx, int n, const double* A, int lda, double* lam, double* T, int ldt, double rel_tol);
int romb_gram_transform(rom_ctx* ctx, double* X, double* Y, int b, int64_t dim, int mode, double rel_tol, int rounds);
int romb_orthonormalize_against(rom_ctx* ctx, double* V, int found, int take, int64_t dim);
