// The basis stage on a snapshot block held in FACTORED form (round 4: behind the C-ABI, on the device).
//
// A snapshot row is a fixed linear image of its system's interface vector, U = Y B^T (rom_fem_expansion_is_linear); the
// expansion reads only the Kc "compact" coordinates of a vector (rom_fem_compact_stride: 272 of 784 at 256 x 256 / 2 x 2) --
// what the ranks of a multi-GPU sweep exchange.  Everything the basis stage needs from the (M, dim) block can therefore be
// formed from the (M, Kc) block Yc:
//     <u, v>_{H^1_0} = yc_u^T S1 yc_v,  S1 = B^T A_1 B      <u, v>_2 = yc_u^T S yc_v,  S = B^T B      u^T A_b v = yc_u^T S_b yc_v
// The reference has no counterpart of the factored form (its snapshots are NumPy rows, src/lib/SolutionsManagers.py:64-68);
// the operations are its basis builders: ReducedBasisGreedy.build (src/lib/ReducedBasis.py:112-139) and the PCA fit of
// ReducedBasisPCA.build (:189-200), on the same snapshots.
//
// Geometry without cancellation.  Quadratic forms y^T S1 y lose small residual norms to the spread of S1's entries (the
// slot of h^2 / a_b carries L^-1 1 ~ N^2, others O(1)), so the block is mapped ONCE to coordinates in which the inner
// product is Euclidean: S1 is equilibrated (unit diagonal), factorised by a diagonally PIVOTED CHOLESKY that stops at the
// numerical rank (pivot <= 1e-14: directions the fp64 tables do not resolve), P (D^-1 S1 D^-1) P^T = L L^T, and
// xi = yc E, E = D P^T L (Kc x k').  Norms are sums of squares from then on.  Round 3 did the same with an eigen-
// decomposition in NumPy on the host (0.95 s at 3 x 3 / N = 171, romhighcontrast_amd/factored.py); the factor does the same
// job, needs no eigen-solver for matrices of a thousand rows, and its triangle gives the way back (POD modes) by one
// substitution.  Built once per FE space, cached on the rom_fem.
#include <cmath>
#include <cstring>
#include <vector>

#include "rom_basis_int.h"

struct rom_factored_map {
  int Kc = 0;
  int k1 = 0;               // rank of the H^1_0 geometry
  double* ET1 = nullptr;    // (k1, Kc): rows of E1^T
  bool have_galerkin = false;
  double* Sb = nullptr;     // (kblk, Kc, Kc): B^T A_b B
  double* bt = nullptr;     // (Kc): B^T B_total
  int k2 = 0;               // rank of the Euclidean geometry
  double* ET2 = nullptr;    // (k2, Kc)
  double* LT2 = nullptr;    // (k2, Kc): row c = column c of the Cholesky factor, over ORIGINAL coordinate indices
  int* piv2 = nullptr;      // (k2): pivot order
  double* d2 = nullptr;     // (Kc): equilibration
};

void rom_factored_map_free(void* p) {
  auto* m = static_cast<rom_factored_map*>(p);
  if (!m) return;
  for (void* q : {(void*)m->ET1, (void*)m->Sb, (void*)m->bt, (void*)m->ET2, (void*)m->LT2, (void*)m->piv2, (void*)m->d2})
    if (q) hipFree(q);
  delete m;
}

namespace {

constexpr int PC_MAX = 3072;      // coordinates the one-workgroup factorisation takes (its LDS: 63 KB)
constexpr double PC_TOL = 1e-14;  // pivots of the equilibrated matrix below this are rounding of the tables

// Diagonally pivoted Cholesky of the equilibrated matrix D^-1 S D^-1 (d_i = sqrt(S_ii)), left-looking, ONE workgroup:
// step j picks the largest remaining diagonal entry (first index on ties), forms column j from row `piv` of S and the j
// earlier columns (a thread owns rows tid, tid + 1024, ...: coalesced reads of LT, the pivot row broadcast from LDS) and
// downdates the diagonal.  LT[c * n + i] = L[i][c] over ORIGINAL indices i (rows pivoted before step c hold 0): no row is
// ever swapped.  Stops when the pivot falls to `tol`: rank_out[0] columns.
__global__ __launch_bounds__(1024) void kf_pivchol(int n, const double* __restrict__ S, double tol, double* __restrict__ LT,
                                                   int* __restrict__ piv, double* __restrict__ dscale, int* __restrict__ rank_out) {
  __shared__ double diag[PC_MAX];
  __shared__ double rowp[PC_MAX];
  __shared__ unsigned char used[PC_MAX];
  __shared__ double rv[1024];
  __shared__ int ri[1024];
  __shared__ int s_p;
  __shared__ double s_ljj;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 1024) {
    const double sii = S[size_t(i) * n + i];
    const bool ok = sii > 0.0;
    dscale[i] = ok ? sqrt(sii) : 1.0;
    diag[i] = ok ? 1.0 : 0.0;
    used[i] = 0;
  }
  __syncthreads();
  int j = 0;
  for (; j < n; ++j) {
    double best = -1.0;
    int at = 0;
    for (int i = tid; i < n; i += 1024)
      if (!used[i] && diag[i] > best) { best = diag[i]; at = i; }
    rv[tid] = best;
    ri[tid] = at;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if (tid < s) {
        const double o = rv[tid + s];
        const int oi = ri[tid + s];
        if (o > rv[tid] || (o == rv[tid] && oi < ri[tid])) { rv[tid] = o; ri[tid] = oi; }
      }
      __syncthreads();
    }
    if (!(rv[0] > tol)) break;  // (uniform: everybody reads the same value)
    if (tid == 0) {
      s_p = ri[0];
      s_ljj = sqrt(rv[0]);
      piv[j] = ri[0];
    }
    __syncthreads();
    const int p = s_p;
    const double ljj = s_ljj, dp = dscale[p];
    for (int k = tid; k < j; k += 1024) rowp[k] = LT[size_t(k) * n + p];
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
      double l = 0.0;
      if (i == p) {
        l = ljj;
      } else if (!used[i]) {
        double v = S[size_t(p) * n + i] / (dscale[i] * dp);
        for (int k = 0; k < j; ++k) v -= LT[size_t(k) * n + i] * rowp[k];
        l = v / ljj;
        diag[i] = fmax(diag[i] - l * l, 0.0);
      }
      LT[size_t(j) * n + i] = l;
    }
    __syncthreads();
    if (tid == 0) used[p] = 1;
    __syncthreads();
  }
  if (tid == 0) rank_out[0] = j;
}

// ET[c][i] = d[i] * LT[c][i]
__global__ void kf_scale_cols(int k, int n, const double* __restrict__ LT, const double* __restrict__ d, double* __restrict__ ET) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx < (long long)k * n) ET[idx] = LT[idx] * d[idx % n];
}

__global__ void kf_set_identity(double* __restrict__ A, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[size_t(i) * n + i] = 1.0;
}

// The way back from the factor's coordinates: row m of V (k numbers) -> the coordinate vector w with w E = v supported
// on the pivot coordinates, w[piv[r]] = x[r] / d[piv[r]], x L1 = v (L1 = the k x k triangle of the pivot rows; back
// substitution from the last column).  One workgroup per row; W (rows x n) is zero-filled by the caller.
__global__ __launch_bounds__(256) void kf_solve_triangle(int k, int n, const double* __restrict__ LT, const int* __restrict__ piv,
                                                         const double* __restrict__ d, const double* __restrict__ V,
                                                         double* __restrict__ W) {
  extern __shared__ double xs[];  // k doubles + k ints + 256 doubles
  int* pv = reinterpret_cast<int*>(xs + k);
  double* red = reinterpret_cast<double*>(pv + k + (k & 1));
  const int m = blockIdx.x, tid = threadIdx.x;
  for (int r = tid; r < k; r += 256) pv[r] = piv[r];
  __syncthreads();
  for (int r = k - 1; r >= 0; --r) {
    const double* col = LT + size_t(r) * n;  // L[.][r]
    double s = 0.0;
    for (int q = r + 1 + tid; q < k; q += 256) s += xs[q] * col[pv[q]];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) red[tid] += red[tid + st];
      __syncthreads();
    }
    if (tid == 0) xs[r] = (V[size_t(m) * k + r] - red[0]) / col[pv[r]];
    __syncthreads();
  }
  for (int r = tid; r < k; r += 256) W[size_t(m) * n + pv[r]] = xs[r] / d[pv[r]];
}

// iteration 0 of the greedy: with the snapshots' own norms as normalisation the reference gets exactly 1.0 for every
// snapshot and argmax takes index 0 (src/lib/ReducedBasis.py:129); norms formed in the factor's coordinates agree with a
// caller's stencil norms to rounding only, so quotients within 1e-10 of 1 ARE that tie (sqrt(h1^2) / h1 is exactly 1)
__global__ void kf_snap_unit(int M, const double* __restrict__ norm0, const double* __restrict__ h1, double* __restrict__ err2) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double rel = sqrt(norm0[m]) / h1[m];
  err2[m] = fabs(rel - 1.0) <= 1e-10 ? h1[m] * h1[m] : norm0[m];
}

__device__ inline double block_sum_1024(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

// Basis vector j from the pick of iteration `it`: its residual (already orthogonal to the basis), normalised, once more
// orthogonalised against Q and renormalised -- in the factor's coordinates (q, k1 numbers) and, with the same
// coefficients, as a coordinate vector of the FE space (w, Kc numbers: what the reduced tensors are built from).
// A pick whose residual is at roundoff of its snapshot (a duplicate) gives q = w = 0 and degenerate[it] = 1.
__global__ __launch_bounds__(1024) void kf_new_direction(int j, int it, int k1, int Kc, int M, const double* __restrict__ R,
                                                         const double* __restrict__ Yc, const int* __restrict__ picks,
                                                         const double* __restrict__ err2, const double* __restrict__ norm0,
                                                         const double* __restrict__ P, double* __restrict__ Q,
                                                         double* __restrict__ W, int* __restrict__ degenerate) {
  __shared__ double red[1024];
  __shared__ double tcoef[2048];
  const int tid = threadIdx.x;
  const int p = picks[it];
  const double e2 = err2[p];
  const bool dead = !(e2 > 1e-26 * norm0[p]) || !(e2 > 0.0);
  double* q = Q + size_t(j) * k1;
  double* w = W + size_t(j) * Kc;
  if (dead) {
    for (int c = tid; c < k1; c += 1024) q[c] = 0.0;
    for (int c = tid; c < Kc; c += 1024) w[c] = 0.0;
    if (tid == 0) degenerate[it] = 1;
    return;
  }
  if (tid == 0) degenerate[it] = 0;
  const double a = 1.0 / sqrt(e2);
  for (int c = tid; c < k1; c += 1024) q[c] = a * R[size_t(p) * k1 + c];
  __syncthreads();
  // t_i = <Q_i, q>, one WAVE per product (round 4: the whole workgroup took them one after the other, a tree of eleven
  // barriers each -- 3.6 us times j, 4.3 of the call's 4.9 ms at n = 50)
  for (int i = tid >> 6; i < j; i += 16) {
    double s = 0.0;
    for (int c = tid & 63; c < k1; c += 64) s += Q[size_t(i) * k1 + c] * q[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) tcoef[i] = s;
  }
  __syncthreads();
  double s2 = 0.0;
  for (int c = tid; c < k1; c += 1024) {
    double v = q[c];
    for (int i = 0; i < j; ++i) v -= tcoef[i] * Q[size_t(i) * k1 + c];
    q[c] = v;
    s2 += v * v;
  }
  const double nrm2 = block_sum_1024(s2, red);
  const double b = nrm2 > 0.0 ? 1.0 / sqrt(nrm2) : 0.0;
  for (int c = tid; c < k1; c += 1024) q[c] *= b;
  // the same combination of coordinate vectors: residual of the pick = yc_p - sum_i P[i][p] w_i
  for (int c = tid; c < Kc; c += 1024) {
    double v = Yc[size_t(p) * Kc + c];
    for (int i = 0; i < j; ++i) v -= P[size_t(i) * M + p] * W[size_t(i) * Kc + c];
    v *= a;
    for (int i = 0; i < j; ++i) v -= tcoef[i] * W[size_t(i) * Kc + c];
    w[c] = v * b;
  }
}

// out[i * nb + b] = <A_i, B_b> for a handful of rows (na x nb products of length n), one wave each: the general GEMM spends
// 56 us on such a launch (split-K bookkeeping for a 50 x 9 result), this one a few
__global__ __launch_bounds__(256) void kf_small_dots(int na, int nb, int n, const double* __restrict__ A, const double* __restrict__ B,
                                                     double* __restrict__ out) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pair >= na * nb) return;
  const double* a = A + size_t(pair / nb) * n;
  const double* b = B + size_t(pair % nb) * n;
  double s = 0.0;
  for (int c = lane; c < n; c += 64) s += a[c] * b[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[pair] = s;
}

// One pass over the residuals: p_m = <R_m, q>, R_m -= p_m q, err2[m] = |R_m|^2 (a sum of squares: exact to rounding).
__global__ __launch_bounds__(256) void kf_pass(int k1, double* __restrict__ R, const double* __restrict__ q,
                                               double* __restrict__ pj, double* __restrict__ err2) {
  __shared__ double red[256];
  const int m = blockIdx.x, tid = threadIdx.x;
  double* r = R + size_t(m) * k1;
  double s = 0.0;
  for (int c = tid; c < k1; c += 256) s += r[c] * q[c];
  red[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const double dot = red[0];
  __syncthreads();
  double s2 = 0.0;
  for (int c = tid; c < k1; c += 256) {
    const double v = r[c] - dot * q[c];
    r[c] = v;
    s2 += v * v;
  }
  red[tid] = s2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    pj[m] = dot;
    err2[m] = red[0];
  }
}

// T[b][i] = sum_k Sb[b][k][i] w[k]   (S_b symmetric: the column walk is the coalesced one)
__global__ __launch_bounds__(256) void kf_sb_apply(int Kc, const double* __restrict__ Sb, const double* __restrict__ w,
                                                   double* __restrict__ T) {
  __shared__ double ws[256];
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const double* S = Sb + size_t(b) * Kc * Kc;
  double acc = 0.0;
  for (int k0 = 0; k0 < Kc; k0 += 256) {
    __syncthreads();
    ws[threadIdx.x] = k0 + int(threadIdx.x) < Kc ? w[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int kn = min(256, Kc - k0);
    if (i < Kc)
      for (int k = 0; k < kn; ++k) acc += S[size_t(k0 + k) * Kc + i] * ws[k];
  }
  if (i < Kc) T[size_t(b) * Kc + i] = acc;
}

// equilibrated pivoted Cholesky of the n x n matrix S: LT (n x n scratch), piv, d on the device; returns the rank
int pivoted_factor(rom_ctx* ctx, int n, const double* S, double* LT, int* piv, double* d, int* rank_host) {
  ROM_CHECK(n <= PC_MAX, "factored map: %d coordinates, at most %d", n, PC_MAX);
  Tmp rk;
  ROM_TRY(rk.get(ctx, 1));
  kf_pivchol<<<1, 1024, 0, ctx->stream>>>(n, S, PC_TOL, LT, piv, d, reinterpret_cast<int*>(rk.p()));
  ROM_HIP(hipGetLastError());
  ROM_HIP(hipMemcpyAsync(rank_host, rk.p(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  return ROM_OK;
}

template <class T>
int dev_alloc(T** p, size_t n) {
  ROM_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T)));
  return ROM_OK;
}

// B^T restricted to the compact coordinates: (Kc, dim) = the expansion of the Kc unit vectors
int basis_rows(rom_fem* f, int Kc, double* Bt) {
  rom_ctx* ctx = f->ctx;
  const int kblk = f->nrb * f->ncb;
  Tmp I, Y, ones;
  ROM_TRY(I.get(ctx, size_t(Kc) * Kc));
  ROM_TRY(Y.get(ctx, size_t(Kc) * f->nGp));
  ROM_TRY(ones.get(ctx, size_t(Kc) * kblk));
  ROM_HIP(hipMemsetAsync(I.p(), 0, size_t(Kc) * Kc * sizeof(double), ctx->stream));
  kf_set_identity<<<unsigned((Kc + 255) / 256), 256, 0, ctx->stream>>>(I.p(), Kc);
  ROM_HIP(hipGetLastError());
  {
    std::vector<double> one(size_t(Kc) * kblk, 1.0);
    ROM_HIP(hipMemcpyAsync(ones.p(), one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  rom_buf bt{ctx, Bt, size_t(Kc) * size_t(f->dim)};
  ROM_TRY(rom_fem_unpack_reduced_async(f, I.b, 0, Kc, Y.b, 0));
  ROM_TRY(rom_expand_batch_async(f, ones.b, Kc, Y.b, 0, &bt, 0));  // (a linear expansion does not read the parameters)
  return rom_solve_status(ctx);
}

int ensure_map(rom_fem* f, int parts) {
  int lin = 0;
  ROM_TRY(rom_fem_expansion_is_linear(f, &lin));
  ROM_CHECK(lin, "this geometry recovers some edges node by node: its expansion is not a linear map of the interface vectors "
                 "(use snapshot rows)");
  rom_ctx* ctx = f->ctx;
  ROM_HIP(hipSetDevice(ctx->device));
  if (!f->fmap) f->fmap = new rom_factored_map();
  auto* mp = static_cast<rom_factored_map*>(f->fmap);
  const int Kc = f->nGp - (f->xb0 - f->nGa);
  mp->Kc = Kc;
  const int kblk = f->nrb * f->ncb;
  const bool need1 = (parts & 1) && !mp->ET1, need2 = (parts & 2) && !mp->have_galerkin, need4 = (parts & 4) && !mp->ET2;
  if (!need1 && !need2 && !need4) return ROM_OK;
  const int64_t dim = f->dim;
  Tmp Bt, ABt, Sm, LT, pv, dd;
  ROM_TRY(Bt.get(ctx, size_t(Kc) * dim));
  ROM_TRY(basis_rows(f, Kc, Bt));
  ROM_TRY(Sm.get(ctx, size_t(Kc) * Kc));
  ROM_TRY(LT.get(ctx, size_t(Kc) * Kc));
  ROM_TRY(pv.get(ctx, Kc));
  ROM_TRY(dd.get(ctx, Kc));
  if (need1 || need2) ROM_TRY(ABt.get(ctx, size_t(Kc) * dim));
  if (need1) {
    ROM_TRY(rom_launch_stencil_apply(f, nullptr, Bt, Kc, ABt));                                          // rows: A_1 B e_i
    ROM_TRY(rom_launch_gemm_nt_ex(ctx, Kc, Kc, dim, 1.0, Bt, dim, ABt, dim, 0.0, Sm, Kc, "gemm_nt", 1));  // S1 = B^T A_1 B (symmetric: lower tiles + mirror)
    int rank = 0;
    ROM_TRY(pivoted_factor(ctx, Kc, Sm, LT, reinterpret_cast<int*>(pv.p()), dd, &rank));
    ROM_CHECK(rank >= 1, "factored map: the H^1_0 form of the expansion has rank 0");
    ROM_TRY(dev_alloc(&mp->ET1, size_t(rank) * Kc));
    kf_scale_cols<<<unsigned((size_t(rank) * Kc + 255) / 256), 256, 0, ctx->stream>>>(rank, Kc, LT, dd, mp->ET1);
    ROM_HIP(hipGetLastError());
    mp->k1 = rank;
  }
  if (need2) {
    ROM_TRY(dev_alloc(&mp->Sb, size_t(kblk) * Kc * Kc));
    ROM_TRY(dev_alloc(&mp->bt, size_t(Kc)));
    Tmp coef, Btot;
    ROM_TRY(coef.get(ctx, size_t(kblk) * kblk));
    ROM_TRY(Btot.get(ctx, dim));
    std::vector<double> eye(size_t(kblk) * kblk, 0.0);
    for (int b = 0; b < kblk; ++b) eye[size_t(b) * kblk + b] = 1.0;
    ROM_HIP(hipMemcpyAsync(coef.p(), eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < kblk; ++b) {
      ROM_TRY(rom_launch_stencil_apply(f, coef.p() + size_t(b) * kblk, Bt, Kc, ABt));                    // rows: A_b B e_i
      ROM_TRY(rom_launch_gemm_nt_ex(ctx, Kc, Kc, dim, 1.0, Bt, dim, ABt, dim, 0.0, mp->Sb + size_t(b) * Kc * Kc, Kc, "gemm_nt", 1));
    }
    {
      std::vector<double> bt(size_t(dim), 1.0 / (double(f->N) * f->N));  // B_total (:177-185: every inner entry h^2)
      ROM_HIP(hipMemcpyAsync(Btot.p(), bt.data(), bt.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
    }
    ROM_TRY(rom_launch_rowdot(ctx, Bt, Kc, dim, Btot, mp->bt));
    mp->have_galerkin = true;
  }
  if (need4) {
    ROM_TRY(rom_launch_gram(ctx, Kc, dim, Bt, dim, Sm, Kc));                                             // S = B^T B
    int rank = 0;
    ROM_TRY(pivoted_factor(ctx, Kc, Sm, LT, reinterpret_cast<int*>(pv.p()), dd, &rank));
    ROM_CHECK(rank >= 1, "factored map: the expansion has rank 0");
    ROM_TRY(dev_alloc(&mp->ET2, size_t(rank) * Kc));
    ROM_TRY(dev_alloc(&mp->LT2, size_t(rank) * Kc));
    ROM_TRY(dev_alloc(&mp->piv2, size_t(rank)));
    ROM_TRY(dev_alloc(&mp->d2, size_t(Kc)));
    kf_scale_cols<<<unsigned((size_t(rank) * Kc + 255) / 256), 256, 0, ctx->stream>>>(rank, Kc, LT, dd, mp->ET2);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipMemcpyAsync(mp->LT2, LT.p(), size_t(rank) * Kc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipMemcpyAsync(mp->piv2, pv.p(), size_t(rank) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipMemcpyAsync(mp->d2, dd.p(), size_t(Kc) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    mp->k2 = rank;
  }
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  return ROM_OK;
}

}  // namespace

// parts: 1 = H^1_0 geometry (norms, greedy), 2 = the block forms u^T A_b v and the load functional (Galerkin greedy),
// 4 = Euclidean geometry (POD).  Builds what is missing, once per FE space; ranks of the two geometries on return.
extern "C" int rom_fem_energy_map(rom_fem* f, int parts, int* k_h10, int* k_l2) {
  ROM_CHECK(f, "rom_fem_energy_map: null fem");
  ROM_TRY(ensure_map(f, parts));
  auto* mp = static_cast<rom_factored_map*>(f->fmap);
  if (k_h10) *k_h10 = mp->k1;
  if (k_l2) *k_l2 = mp->k2;
  return ROM_OK;
}

// H^1_0 norms of M snapshots from the compact coordinates of their interface vectors (= rom_h10norm of their rows)
extern "C" int rom_h10norm_factored(rom_fem* f, rom_buf* Yc, int64_t c_row0, int M, double* out_host) {
  ROM_CHECK(f && Yc && (out_host || M == 0), "rom_h10norm_factored: null argument");
  ROM_CHECK(M >= 0 && c_row0 >= 0, "rom_h10norm_factored: negative size");
  if (M == 0) return ROM_OK;
  ROM_TRY(ensure_map(f, 1));
  auto* mp = static_cast<rom_factored_map*>(f->fmap);
  rom_ctx* ctx = f->ctx;
  const int Kc = mp->Kc, k1 = mp->k1;
  ROM_CHECK(size_t(c_row0 + M) * Kc <= Yc->n, "rom_h10norm_factored: buffer too small");
  Tmp Xi, nr;
  ROM_TRY(Xi.get(ctx, size_t(M) * k1));
  ROM_TRY(nr.get(ctx, M));
  ROM_TRY(rom_launch_gemm_nt(ctx, M, k1, Kc, 1.0, Yc->p + c_row0 * Kc, Kc, mp->ET1, Kc, 0.0, Xi, k1, "gemm_nt"));
  ROM_TRY(rom_launch_l2norm(ctx, Xi, M, k1, nr, true));
  return download(ctx, nr, out_host, M);
}

// ReducedBasisGreedy.build (src/lib/ReducedBasis.py:112-139) on a training block held as compact interface vectors
// Yc[c_row0 ...] (M x Kc).  Same contract as rom_greedy: mode 0 = error of the H^1_0 projection, 1 = of the Galerkin ROM
// (needs the parameters a); picks_out / max_err_out: n entries.  In the factor's coordinates the projection is Euclidean
// (residuals updated by one vector per iteration, their norms exact sums of squares), the Galerkin error is
// |R_m|^2 + sum_j (p_mj - c_mj)^2 with c_m from reduced systems whose tensor grows by one row per iteration.
extern "C" int rom_greedy_factored(rom_fem* f, rom_buf* Yc, int64_t c_row0, int M, rom_buf* a, const double* h1norm_host,
                                   int mode, int n, int64_t* picks_out, double* max_err_out) {
  ROM_CHECK(f && Yc && h1norm_host && (picks_out || n == 0) && (max_err_out || n == 0), "rom_greedy_factored: null argument");
  ROM_CHECK(mode == 0 || mode == 1, "rom_greedy_factored: mode must be 0 (H10 projection) or 1 (Galerkin)");
  ROM_CHECK(mode == 0 || a, "rom_greedy_factored: the Galerkin greedy needs the training parameters");
  ROM_CHECK(M >= 1 && M <= 65535 && n >= 0 && c_row0 >= 0, "rom_greedy_factored: bad sizes (1 <= M <= 65535)");
  ROM_CHECK(n <= 2048, "rom_greedy_factored: at most 2048 basis vectors");
  if (n == 0) return ROM_OK;
  ROM_TRY(ensure_map(f, mode == 1 ? 3 : 1));
  auto* mp = static_cast<rom_factored_map*>(f->fmap);
  rom_ctx* ctx = f->ctx;
  const int Kc = mp->Kc, k1 = mp->k1, k = f->nrb * f->ncb;
  ROM_CHECK(size_t(c_row0 + M) * Kc <= Yc->n && (!a || size_t(M) * k <= a->n), "rom_greedy_factored: buffers too small");
  const double* yc = Yc->p + c_row0 * Kc;
  const int nb = std::max(n - 1, 1);
  Tmp R, Q, W, P, err2, norm0, h1, ipick, idead, maxerr, Ahat, bhat, cg, extra, T, col;
  ROM_TRY(R.get(ctx, size_t(M) * k1));
  ROM_TRY(Q.get(ctx, size_t(nb) * k1));
  ROM_TRY(W.get(ctx, size_t(nb) * Kc));
  ROM_TRY(P.get(ctx, size_t(nb) * M));
  ROM_TRY(err2.get(ctx, M));
  ROM_TRY(norm0.get(ctx, M));
  ROM_TRY(h1.get(ctx, M));
  ROM_TRY(ipick.get(ctx, n));
  ROM_TRY(idead.get(ctx, n));
  ROM_TRY(maxerr.get(ctx, n));
  int* d_picks = reinterpret_cast<int*>(ipick.p());
  int* d_dead = reinterpret_cast<int*>(idead.p());
  if (mode == 1) {
    ROM_TRY(Ahat.get(ctx, size_t(k) * nb * nb));
    ROM_TRY(bhat.get(ctx, nb));
    ROM_TRY(cg.get(ctx, size_t(M) * nb));
    ROM_TRY(extra.get(ctx, M));
    ROM_TRY(T.get(ctx, size_t(k) * Kc));
    ROM_TRY(col.get(ctx, size_t(nb) * k));
    ROM_HIP(hipMemsetAsync(Ahat.p(), 0, size_t(k) * nb * nb * sizeof(double), ctx->stream));
  }
  ROM_HIP(hipMemcpyAsync(h1.p(), h1norm_host, size_t(M) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));  // the caller's array is free again
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  ROM_HIP(hipMemsetAsync(ipick.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemsetAsync(idead.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemsetAsync(maxerr.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  // the block in the factor's coordinates; empty basis: the error of snapshot m is |u_m| / h1_m
  ROM_TRY(rom_launch_gemm_nt(ctx, M, k1, Kc, 1.0, yc, Kc, mp->ET1, Kc, 0.0, R, k1, "gemm_nt"));
  ROM_TRY(rom_launch_l2norm(ctx, R, M, k1, norm0, false));
  kf_snap_unit<<<unsigned((M + 255) / 256), 256, 0, ctx->stream>>>(M, norm0, h1, err2);
  kb_greedy_select<<<1, 1024, 0, ctx->stream>>>(M, err2, nullptr, h1, 0, d_picks, maxerr);
  ROM_HIP(hipMemcpyAsync(err2.p(), norm0.p(), size_t(M) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  ROM_HIP(hipGetLastError());
  for (int it = 1; it < n; ++it) {
    const int j = it - 1;
    kf_new_direction<<<1, 1024, 0, ctx->stream>>>(j, it - 1, k1, Kc, M, R, yc, d_picks, err2, norm0, P, Q, W, d_dead);
    double* pj = P.p() + size_t(j) * M;
    kf_pass<<<M, 256, 0, ctx->stream>>>(k1, R, Q.p() + size_t(j) * k1, pj, err2);
    ROM_HIP(hipGetLastError());
    if (mode == 1) {
      const double* wj = W.p() + size_t(j) * Kc;
      kf_sb_apply<<<dim3(unsigned((Kc + 255) / 256), unsigned(k)), 256, 0, ctx->stream>>>(Kc, mp->Sb, wj, T);      // T[b] = S_b w_j
      ROM_HIP(hipGetLastError());
      kf_small_dots<<<unsigned(((j + 1) * k + 3) / 4), 256, 0, ctx->stream>>>(j + 1, k, Kc, W, T, col);                // col[i, b] = w_i . S_b w_j
      ROM_HIP(hipGetLastError());
      kb_grow_ahat<<<unsigned(((j + 1) * k + 255) / 256), 256, 0, ctx->stream>>>(Ahat, k, nb, j, col, d_dead, it - 1);
      ROM_HIP(hipGetLastError());
      ROM_TRY(rom_launch_rowdot(ctx, wj, 1, Kc, mp->bt, bhat.p() + j));                                               // w_j . B^T B_total
      ROM_TRY(rom_launch_reduced_solve(ctx, j + 1, nb, k, M, Ahat, a->p, bhat, 0, cg));
      kb_galerkin_gap<<<unsigned((M + 255) / 256), 256, 0, ctx->stream>>>(M, j + 1, P, cg, extra);
      ROM_HIP(hipGetLastError());
    }
    kb_greedy_select<<<1, 1024, 0, ctx->stream>>>(M, err2, mode == 1 ? extra.p() : nullptr, h1, it, d_picks, maxerr);
    ROM_HIP(hipGetLastError());
  }
  Tmp outd;
  ROM_TRY(outd.get(ctx, 2 * size_t(n)));
  kb_ints_to_doubles<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(d_picks, outd, n);
  ROM_HIP(hipMemcpyAsync(outd.p() + n, maxerr.p(), size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  std::vector<double> host(2 * size_t(n));
  ROM_TRY(download(ctx, outd, host.data(), host.size()));
  for (int i = 0; i < n; ++i) {
    picks_out[i] = int64_t(host[i]);
    max_err_out[i] = host[n + i];
  }
  return read_status(ctx, "rom_greedy_factored");
}

// PCA(n_components = n).fit (src/lib/ReducedBasis.py:196) of a snapshot block held as compact interface vectors
// Yc[c_row0 ...] (M x Kc, not modified): the rows of Z = Yc E2 (M x k2, k2 <= Kc) have the Euclidean geometry of the snapshot
// rows, so the POD of the block IS the POD of Z (rom_pod, on a matrix dim / k2 times narrower); a mode is taken back to
// coordinates by one triangular substitution and expanded like any interface vector.  V[v_row0 ...]: (n, dim) rows;
// sigma_host / info_host as rom_pod (info: of the inner call; completed modes include those beyond the rank of the map).
extern "C" int rom_pod_factored(rom_fem* f, rom_buf* Yc, int64_t c_row0, int M, int n, int center, rom_buf* V, int64_t v_row0,
                                double* sigma_host, double* info_host) {
  ROM_CHECK(f && Yc && V && (sigma_host || n == 0), "rom_pod_factored: null argument");
  ROM_CHECK(M >= 1 && n >= 0 && c_row0 >= 0 && v_row0 >= 0, "rom_pod_factored: bad sizes");
  const int64_t dim = f->dim;
  ROM_CHECK(n <= std::min<int64_t>(M, dim), "rom_pod_factored: %d modes requested from a %d x %lld block", n, M, (long long)dim);
  ROM_CHECK(size_t(v_row0 + n) * dim <= V->n, "rom_pod_factored: mode buffer too small");
  ROM_TRY(ensure_map(f, 4));
  auto* mp = static_cast<rom_factored_map*>(f->fmap);
  rom_ctx* ctx = f->ctx;
  const int Kc = mp->Kc, k2 = mp->k2, kblk = f->nrb * f->ncb;
  ROM_CHECK(size_t(c_row0 + M) * Kc <= Yc->n, "rom_pod_factored: buffer too small");
  for (int i = 0; i < n; ++i) sigma_host[i] = 0.0;
  if (n == 0) return ROM_OK;
  Tmp Ycc, mean, Z, Vz, Wc, Yfull, ones;
  ROM_TRY(Ycc.get(ctx, size_t(M) * Kc));
  ROM_HIP(hipMemcpyAsync(Ycc.p(), Yc->p + c_row0 * Kc, size_t(M) * Kc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (center) {  // the expansion is linear: the mean row is the expansion of the mean vector
    ROM_TRY(mean.get(ctx, Kc));
    ROM_TRY(rom_launch_center_rows(ctx, Ycc, M, Kc, mean));
  }
  ROM_TRY(Z.get(ctx, size_t(M) * k2));
  ROM_TRY(rom_launch_gemm_nt(ctx, M, k2, Kc, 1.0, Ycc, Kc, mp->ET2, Kc, 0.0, Z, k2, "gemm_nt"));
  const int nz = std::min(n, std::min(k2, M));
  ROM_TRY(Vz.get(ctx, size_t(nz) * k2));
  double info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ROM_TRY(rom_pod(ctx, Z.b, 0, M, k2, nz, 0, Vz.b, 0, sigma_host, info));
  // modes as coordinate vectors, then as rows
  ROM_TRY(Wc.get(ctx, size_t(nz) * Kc));
  ROM_HIP(hipMemsetAsync(Wc.p(), 0, size_t(nz) * Kc * sizeof(double), ctx->stream));
  {
    const size_t lds = size_t(k2) * 8 + size_t(k2 + (k2 & 1)) * 4 + 256 * 8;
    ROM_CHECK(lds <= 64 * 1024, "rom_pod_factored: map of rank %d too large for the substitution kernel", k2);
    kf_solve_triangle<<<nz, 256, lds, ctx->stream>>>(k2, Kc, mp->LT2, mp->piv2, mp->d2, Vz, Wc);
    ROM_HIP(hipGetLastError());
  }
  ROM_TRY(Yfull.get(ctx, size_t(nz) * f->nGp));
  ROM_TRY(ones.get(ctx, size_t(nz) * kblk));
  {
    std::vector<double> one(size_t(nz) * kblk, 1.0);
    ROM_HIP(hipMemcpyAsync(ones.p(), one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  ROM_TRY(rom_fem_unpack_reduced_async(f, Wc.b, 0, nz, Yfull.b, 0));
  ROM_TRY(rom_expand_batch_async(f, ones.b, nz, Yfull.b, 0, V, v_row0));
  ROM_TRY(rom_solve_status(ctx));
  // the map is orthonormal to the accuracy of its factor (~1e-9 relative to the leading directions): clean up
  ROM_TRY(rom_symmetric_orthonormalize(ctx, V, v_row0, nz, dim));
  if (nz < n) {  // more modes requested than the snapshot manifold has dimensions: completed like rom_pod completes
    ROM_TRY(rom_complete_orthonormal(ctx, V, v_row0, nz, n - nz, dim));
    info[1] += n - nz;
    if (info[7] == 0.0) info[7] = 1.0;   // (not "filled": the completed modes lie beyond the rank of the snapshot manifold)
  }
  ROM_TRY(rom_launch_rows_sign_flip(ctx, V->p + v_row0 * dim, n, dim));  // svd_flip(u_based_decision=False)
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (info_host) {
    info[4] += 2.0 * M * double(Kc) * k2 + 2.0 * nz * double(Kc) * double(dim);
    memcpy(info_host, info, sizeof(info));
  }
  return ROM_OK;
}
