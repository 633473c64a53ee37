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
  double* Li2 = nullptr;    // (k2, k2): inverse of the pivot rows' triangle L1 (L1[r][c] = LT2[c][piv2[r]]): x L1 = v is x = v Li2
};

void rom_factored_map_free(void* p) {
  auto* m = static_cast<rom_factored_map*>(p);
  if (!m) return;
  for (void* q : {(void*)m->ET1, (void*)m->Sb, (void*)m->bt, (void*)m->ET2, (void*)m->LT2, (void*)m->piv2, (void*)m->d2, (void*)m->Li2})
    if (q) hipFree(q);
  delete m;
}

namespace {

constexpr double PC_TOL = 1e-14;  // pivots of the equilibrated matrix below this are rounding of the tables
constexpr int PC_PANEL = 32;      // pivots per panel of the blocked factorisation

// A = D^-1 S D^-1 (d_i = sqrt(S_ii); a non-positive diagonal entry: d_i = 1 and the coordinate is never pivoted), diag = its
// diagonal (1 or 0), nobody used yet
__global__ void kf_equilibrate(int n, const double* __restrict__ S, double* __restrict__ A, double* __restrict__ d,
                               double* __restrict__ diag, int* __restrict__ used, int* __restrict__ state) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx == 0) { state[0] = 0; state[1] = 0; }
  if (idx >= (long long)n * n) return;
  const int i = int(idx / n), j = int(idx - (long long)i * n);
  const double sii = S[size_t(i) * n + i], sjj = S[size_t(j) * n + j];
  const double di = sii > 0.0 ? sqrt(sii) : 1.0, dj = sjj > 0.0 ? sqrt(sjj) : 1.0;
  A[idx] = S[idx] / (di * dj);
  if (i == j) {
    d[i] = di;
    diag[i] = sii > 0.0 ? 1.0 : 0.0;
    used[i] = 0;
  }
}

// Diagonally pivoted Cholesky of the equilibrated matrix, BLOCKED (round 5; round 4 ran all steps in one workgroup, every
// column against all earlier ones: 44 ms for 769 pivots of 928 coordinates, and its LDS arrays capped the map at 3072
// coordinates).  One panel = PC_PANEL pivots in ONE workgroup on the matrix A that already carries the downdates of all
// earlier panels: step j picks the largest remaining diagonal entry (first index on ties), forms column j from row `piv` of A
// and the columns of THIS panel, downdates the diagonal.  Between panels A -= P^T P (the panel's rows of LT) is one MFMA
// product on the whole chip.  LT[c * n + i] = L[i][c] over ORIGINAL indices i (rows pivoted before step c hold 0): no row is
// ever swapped.  state[0] = columns so far, state[1] = 1 once the pivot has fallen to `tol` (later panels return at once).
__global__ __launch_bounds__(1024) void kf_pivchol_panel(int n, const double* __restrict__ A, double tol, double* __restrict__ LT,
                                                         int* __restrict__ piv, double* __restrict__ diag, int* __restrict__ used,
                                                         int* __restrict__ state, int j0) {
  __shared__ double rv[16];
  __shared__ int ri[16];
  __shared__ double lp[PC_PANEL];
  __shared__ int s_p;
  __shared__ double s_best;
  const int tid = threadIdx.x;
  if (state[1]) return;
  int j = j0;
  const int jend = min(n, j0 + PC_PANEL);
  bool done = false;
  for (; j < jend; ++j) {
    double best = -1.0;
    int at = 0x7fffffff;
    for (int i = tid; i < n; i += 1024)
      if (!used[i] && diag[i] > best) { best = diag[i]; at = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_down(best, o, 64);
      const int oa = __shfl_down(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = at; }
    __syncthreads();
    if (tid == 0) {
      double bb = rv[0];
      int ba = ri[0];
      for (int w = 1; w < 16; ++w)
        if (rv[w] > bb || (rv[w] == bb && ri[w] < ba)) { bb = rv[w]; ba = ri[w]; }
      s_best = bb;
      s_p = ba;
    }
    __syncthreads();
    const double pv = s_best;
    const int p = s_p;
    if (!(pv > tol)) { done = true; break; }  // (uniform)
    const double ljj = sqrt(pv);
    if (tid < j - j0) lp[tid] = LT[size_t(j0 + tid) * n + p];
    __syncthreads();
    const int kk = j - j0;
    for (int i = tid; i < n; i += 1024) {
      double l = 0.0;
      if (i == p) {
        l = ljj;
      } else if (!used[i]) {
        double v = A[size_t(p) * n + i];
        for (int k = 0; k < kk; ++k) v -= LT[size_t(j0 + k) * n + i] * lp[k];
        l = v / ljj;
        diag[i] = fmax(diag[i] - l * l, 0.0);
      }
      LT[size_t(j) * n + i] = l;
    }
    if (tid == 0) {
      piv[j] = p;
      used[p] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) {
    state[0] = j;
    if (done || j >= n) state[1] = 1;
  }
}

// Pt (n x PC_PANEL) = the transpose of the panel's rows of LT (zero columns behind the rows the panel wrote)
__global__ void kf_panel_transpose(int n, const double* __restrict__ LT, int j0, const int* __restrict__ state, double* __restrict__ Pt,
                                   double* __restrict__ Pz) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)n * PC_PANEL) return;
  const int i = int(idx / PC_PANEL), k = int(idx - (long long)i * PC_PANEL);
  const double v = j0 + k < state[0] ? LT[size_t(j0 + k) * n + i] : 0.0;   // (rows behind the rank hold stale numbers: zero)
  Pt[idx] = v;
  Pz[size_t(k) * n + i] = v;
}

// OUT[k][i * w + j] = IN[k][(r0 + i) * ld + c0 + j]: the part of K FE vectors on a rectangle of the mesh
__global__ void kf_gather_rect(const double* __restrict__ IN, long long dim, int ld, int r0, int c0, int h, int w,
                               double* __restrict__ OUT) {
  const long long hw = (long long)h * w;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= hw) return;
  const int i = int(idx / w), j = int(idx - (long long)i * w);
  OUT[blockIdx.y * hw + idx] = IN[blockIdx.y * dim + (long long)(r0 + i) * ld + c0 + j];
}

__global__ void kf_add_inplace(double* __restrict__ acc, const double* __restrict__ x, long long n) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx < n) acc[idx] += x[idx];
}

// L1[r][c] = LT[c * n + piv[r]] (c <= r; 0 above the diagonal): the k x k triangle of the pivot rows, dense and row-major
__global__ void kf_gather_triangle(int k, int n, const double* __restrict__ LT, const int* __restrict__ piv, double* __restrict__ L1) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)k * k) return;
  const int r = int(idx / k), c = int(idx - (long long)r * k);
  L1[idx] = c <= r ? LT[size_t(c) * n + piv[r]] : 0.0;
}

// X = L1^-1 (k x k lower triangular, row-major): ONE WAVE per column j walks down it by forward substitution, the row's dot
// product spread over the lanes; the column lives in LDS until it is written out.  (A thread per column -- 769 columns of
// 300k dependent multiply-adds each -- took 55 ms at C4.)
__global__ __launch_bounds__(256) void kf_tri_inverse(int k, const double* __restrict__ L1, double* __restrict__ X) {
  extern __shared__ double xcol[];   // blockDim.x / 64 columns of k doubles
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * int(blockDim.x >> 6) + w;
  if (j >= k) return;
  double* x = xcol + size_t(w) * k;
  for (int i = j; i < k; ++i) {
    const double* row = L1 + size_t(i) * k;
    double s = 0.0;
    for (int q = j + lane; q < i; q += 64) s += row[q] * x[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double xi = ((i == j ? 1.0 : 0.0) - s) / row[i];
    if (lane == 0) x[i] = xi;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // (lgkmcnt(0): the store to LDS is in before the next row reads it; one wave, no barrier)
  }
  for (int i = lane; i < k; i += 64) X[size_t(i) * k + j] = i >= j ? x[i] : 0.0;
}

// ET[c][i] = d[i] * LT[c][i]
__global__ void kf_scale_cols(int k, int n, const double* __restrict__ LT, const double* __restrict__ d, double* __restrict__ ET) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx < (long long)k * n) ET[idx] = LT[idx] * d[idx % n];
}

__global__ void kf_fill_const(double* __restrict__ p, long long n, double v) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void kf_set_identity(double* __restrict__ A, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[size_t(i) * n + i] = 1.0;
}

// The way back from the factor's coordinates: x L1 = v (L1 = the k x k triangle of the pivot rows) gives the coordinate
// vector w with w E = v supported on the pivot coordinates, w[piv[r]] = x[r] / d[piv[r]].  W (rows x n) is zero-filled by
// the caller; x = v L1^-1 is a product with the inverse kept in the map.
__global__ void kf_scatter_pivots(int k, int n, const int* __restrict__ piv, const double* __restrict__ d, const double* __restrict__ X,
                                  double* __restrict__ W) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (r < k) W[size_t(m) * n + piv[r]] = X[size_t(m) * k + r] / d[piv[r]];
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
  // compensated: the products exactly (fma), the running sum with its rounding error carried along (TwoSum)
  double s = 0.0, comp = 0.0;
  for (int c = lane; c < n; c += 64) {
    const double pr = a[c] * b[c], pe = fma(a[c], b[c], -pr);
    const double t = s + pr, bb = t - s;
    comp += ((s - (t - bb)) + (pr - bb)) + pe;
    s = t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double os = __shfl_xor(s, off), oc = __shfl_xor(comp, off);
    const double t = s + os, bb = t - s;
    comp += ((s - (t - bb)) + (os - bb)) + oc;
    s = t;
  }
  if (lane == 0) out[pair] = s + comp;
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


// equilibrated pivoted Cholesky of the n x n matrix S (blocked, see kf_pivchol_panel): LT (n x n scratch), piv, d on the
// device; returns the rank.  No limit on n beyond memory.
int pivoted_factor(rom_ctx* ctx, int n, const double* S, double* LT, int* piv, double* d, int* rank_host) {
  Tmp A, diag, used, st, Pt, Pz;
  ROM_TRY(A.get(ctx, size_t(n) * n));
  ROM_TRY(diag.get(ctx, n));
  ROM_TRY(used.get(ctx, n));
  ROM_TRY(st.get(ctx, 2));
  ROM_TRY(Pt.get(ctx, size_t(n) * PC_PANEL));
  ROM_TRY(Pz.get(ctx, size_t(n) * PC_PANEL));
  int* d_used = reinterpret_cast<int*>(used.p());
  int* d_state = reinterpret_cast<int*>(st.p());
  kf_equilibrate<<<unsigned((size_t(n) * n + 255) / 256), 256, 0, ctx->stream>>>(n, S, A, d, diag, d_used, d_state);
  ROM_HIP(hipGetLastError());
  int state[2] = {0, 0};
  int panel = 0;
  for (int j0 = 0; j0 < n; j0 += PC_PANEL, ++panel) {
    {
      ROM_PROF(ctx, "pivchol_panel", 2.0 * n * PC_PANEL * PC_PANEL, 8.0 * n * PC_PANEL);
      kf_pivchol_panel<<<1, 1024, 0, ctx->stream>>>(n, A, PC_TOL, LT, piv, diag, d_used, d_state, j0);
    }
    ROM_HIP(hipGetLastError());
    if (j0 + PC_PANEL >= n) break;
    if (panel % 8 == 7) {  // (every eighth panel the host looks whether the pivots have reached the tolerance)
      ROM_HIP(hipMemcpyAsync(state, d_state, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
      if (state[1]) break;
    }
    kf_panel_transpose<<<unsigned((size_t(n) * PC_PANEL + 255) / 256), 256, 0, ctx->stream>>>(n, LT, j0, d_state, Pt, Pz);
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_gemm_nn(ctx, n, n, PC_PANEL, -1.0, Pt, PC_PANEL, Pz, n, 1.0, A, n));   // A -= P^T P
  }
  ROM_HIP(hipMemcpyAsync(state, d_state, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  *rank_host = state[0];
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

// What a part of the map owns on the device while it is being built: freed on any early return, handed to the map only
// when the whole part (ranks included) is complete -- a failure halfway leaves the map as it was.
struct DevOwner {
  std::vector<void*> ptrs;
  ~DevOwner() {
    for (void* p : ptrs)
      if (p) hipFree(p);
  }
  template <class T>
  int alloc(T** p, size_t n) {
    ROM_TRY(dev_alloc(p, n));
    ptrs.push_back(*p);
    return ROM_OK;
  }
  void release() { ptrs.clear(); }
};

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
  const int kblk = f->nrb * f->ncb, N = f->N;
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
  bool have_s1 = false;   // Sm holds S1 = B^T A_1 B
  if (need2) {
    // The block forms S_b = B^T A_b B.  A_b B e_i lives on the closure of block b -- (N + 1)^2 of the dim vertices -- so the
    // product runs over THAT rectangle of both operands (gathered into compact rows): nine products of a ninth of the
    // length at 3 x 3 instead of nine of the full length (round 4: 60 of the map's 127 ms).  S1 = sum_b S_b (A_1 = sum_b A_b).
    DevOwner own;
    double *Sb = nullptr, *bt = nullptr;
    ROM_TRY(own.alloc(&Sb, size_t(kblk) * Kc * Kc));
    ROM_TRY(own.alloc(&bt, size_t(Kc)));
    Tmp coef, Btot, Bc, Zc;
    ROM_TRY(coef.get(ctx, size_t(kblk) * kblk));
    ROM_TRY(Btot.get(ctx, dim));
    const size_t rect = size_t(N + 1) * (N + 1);
    ROM_TRY(Bc.get(ctx, size_t(Kc) * rect));
    ROM_TRY(Zc.get(ctx, size_t(Kc) * rect));
    std::vector<double> eye(size_t(kblk) * kblk, 0.0);
    for (int b = 0; b < kblk; ++b) eye[size_t(b) * kblk + b] = 1.0;
    ROM_HIP(hipMemcpyAsync(coef.p(), eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < kblk; ++b) {
      const int p = b / f->ncb, q = b % f->ncb;
      // inner vertices (0-based rows r, columns c) touched by the cells of block (p, q): r in [p N - 1, (p + 1) N - 1] clipped
      const int r0 = std::max(0, p * N - 1), r1 = std::min(f->nr - 1, (p + 1) * N - 1);
      const int c0 = std::max(0, q * N - 1), c1 = std::min(f->nc - 1, (q + 1) * N - 1);
      const int h = r1 - r0 + 1, w = c1 - c0 + 1;
      ROM_TRY(rom_launch_stencil_apply_band(f, coef.p() + size_t(b) * kblk, Bt, Kc, ABt, r0, r1));       // rows: A_b B e_i on the block's band
      const dim3 gg(unsigned((size_t(h) * w + 255) / 256), unsigned(Kc));
      kf_gather_rect<<<gg, 256, 0, ctx->stream>>>(Bt, dim, f->nc, r0, c0, h, w, Bc);
      kf_gather_rect<<<gg, 256, 0, ctx->stream>>>(ABt, dim, f->nc, r0, c0, h, w, Zc);
      ROM_HIP(hipGetLastError());
      ROM_TRY(rom_launch_gemm_nt_ex(ctx, Kc, Kc, int64_t(h) * w, 1.0, Bc, int64_t(h) * w, Zc, int64_t(h) * w, 0.0, Sb + size_t(b) * Kc * Kc, Kc,
                                    "gemm_nt", 1));
      if (need1) {
        if (b == 0) ROM_HIP(hipMemcpyAsync(Sm.p(), Sb, size_t(Kc) * Kc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        else kf_add_inplace<<<unsigned((size_t(Kc) * Kc + 255) / 256), 256, 0, ctx->stream>>>(Sm, Sb + size_t(b) * Kc * Kc, (long long)Kc * Kc);
        ROM_HIP(hipGetLastError());
      }
    }
    {
      std::vector<double> bt_h(size_t(dim), 1.0 / (double(f->N) * f->N));  // B_total (:177-185: every inner entry h^2)
      ROM_HIP(hipMemcpyAsync(Btot.p(), bt_h.data(), bt_h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
    }
    ROM_TRY(rom_launch_rowdot(ctx, Bt, Kc, dim, Btot, bt));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    mp->Sb = Sb;
    mp->bt = bt;
    mp->have_galerkin = true;
    own.release();
    have_s1 = need1;
  }
  if (need1) {
    if (!have_s1) {
      ROM_TRY(rom_launch_stencil_apply(f, nullptr, Bt, Kc, ABt));                                          // rows: A_1 B e_i
      ROM_TRY(rom_launch_gemm_nt_ex(ctx, Kc, Kc, dim, 1.0, Bt, dim, ABt, dim, 0.0, Sm, Kc, "gemm_nt", 1));  // S1 = B^T A_1 B (symmetric: lower tiles + mirror)
    }
    int rank = 0;
    ROM_TRY(pivoted_factor(ctx, Kc, Sm, LT, reinterpret_cast<int*>(pv.p()), dd, &rank));
    ROM_CHECK(rank >= 1, "factored map: the H^1_0 form of the expansion has rank 0");
    DevOwner own;
    double* ET1 = nullptr;
    ROM_TRY(own.alloc(&ET1, size_t(rank) * Kc));
    kf_scale_cols<<<unsigned((size_t(rank) * Kc + 255) / 256), 256, 0, ctx->stream>>>(rank, Kc, LT, dd, ET1);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    mp->ET1 = ET1;
    mp->k1 = rank;
    own.release();
  }
  if (need4) {
    ROM_TRY(rom_launch_gram(ctx, Kc, dim, Bt, dim, Sm, Kc));                                             // S = B^T B
    int rank = 0;
    ROM_TRY(pivoted_factor(ctx, Kc, Sm, LT, reinterpret_cast<int*>(pv.p()), dd, &rank));
    ROM_CHECK(rank >= 1, "factored map: the expansion has rank 0");
    DevOwner own;
    double *ET2 = nullptr, *LT2 = nullptr, *d2 = nullptr, *Li2 = nullptr;
    int* piv2 = nullptr;
    ROM_TRY(own.alloc(&ET2, size_t(rank) * Kc));
    ROM_TRY(own.alloc(&LT2, size_t(rank) * Kc));
    ROM_TRY(own.alloc(&piv2, size_t(rank)));
    ROM_TRY(own.alloc(&d2, size_t(Kc)));
    ROM_TRY(own.alloc(&Li2, size_t(rank) * rank));
    kf_scale_cols<<<unsigned((size_t(rank) * Kc + 255) / 256), 256, 0, ctx->stream>>>(rank, Kc, LT, dd, ET2);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipMemcpyAsync(LT2, LT.p(), size_t(rank) * Kc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipMemcpyAsync(piv2, pv.p(), size_t(rank) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipMemcpyAsync(d2, dd.p(), size_t(Kc) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    // the way back from the factor's coordinates needs L1^-1 (L1 = the pivot rows' triangle): once here, a product per call
    {
      Tmp L1;
      ROM_TRY(L1.get(ctx, size_t(rank) * rank));
      kf_gather_triangle<<<unsigned((size_t(rank) * rank + 255) / 256), 256, 0, ctx->stream>>>(rank, Kc, LT2, piv2, L1);
      const int cpw = rank <= 1920 ? 4 : rank <= 3840 ? 2 : 1;   // columns (waves) per workgroup: their LDS stays below 64 KB
      ROM_CHECK(size_t(cpw) * rank * sizeof(double) <= 62 * 1024, "factored map: rank %d beyond what the inverse kernel's LDS columns hold", rank);
      kf_tri_inverse<<<unsigned((rank + cpw - 1) / cpw), 64 * cpw, size_t(cpw) * rank * sizeof(double), ctx->stream>>>(rank, L1, Li2);
      ROM_HIP(hipGetLastError());
      ROM_HIP(hipStreamSynchronize(ctx->stream));   // (L1 is a temporary)
    }
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    mp->ET2 = ET2;
    mp->LT2 = LT2;
    mp->piv2 = piv2;
    mp->d2 = d2;
    mp->Li2 = Li2;
    mp->k2 = rank;
    own.release();
  }
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
      // T[b] = S_b w_j: S_b is symmetric, so entry i is the product of ROW i with w_j -- a wave per entry, compensated
      // (kf_small_dots; the column walk of a thread per entry read the same 43 MB at a third of the rate)
      kf_small_dots<<<unsigned((k * Kc + 3) / 4), 256, 0, ctx->stream>>>(k * Kc, 1, Kc, mp->Sb, wj, T);
      ROM_HIP(hipGetLastError());
      kf_small_dots<<<unsigned(((j + 1) * k + 3) / 4), 256, 0, ctx->stream>>>(j + 1, k, Kc, W, T, col);                // col[i, b] = w_i . S_b w_j
      ROM_HIP(hipGetLastError());
      kb_grow_ahat<<<unsigned(((j + 1) * k + 255) / 256), 256, 0, ctx->stream>>>(Ahat, k, nb, j, col, d_dead, it - 1);
      ROM_HIP(hipGetLastError());
      kf_small_dots<<<1, 256, 0, ctx->stream>>>(1, 1, Kc, wj, mp->bt, bhat.p() + j);                                      // w_j . B^T B_total (compensated)
      ROM_HIP(hipGetLastError());
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
    // x L1 = v  =>  x = v L1^-1 (the inverse is part of the map); w[piv[r]] = x[r] / d[piv[r]]
    Tmp Xs;
    ROM_TRY(Xs.get(ctx, size_t(nz) * k2));
    ROM_TRY(rom_launch_gemm_nn(ctx, nz, k2, k2, 1.0, Vz, k2, mp->Li2, k2, 0.0, Xs, k2));
    kf_scatter_pivots<<<dim3(unsigned((k2 + 255) / 256), unsigned(nz)), 256, 0, ctx->stream>>>(k2, Kc, mp->piv2, mp->d2, Xs, Wc);
    ROM_HIP(hipGetLastError());
  }
  ROM_TRY(Yfull.get(ctx, size_t(nz) * f->nGp));
  ROM_TRY(ones.get(ctx, size_t(nz) * kblk));
  kf_fill_const<<<unsigned((size_t(nz) * kblk + 255) / 256), 256, 0, ctx->stream>>>(ones, (long long)nz * kblk, 1.0);
  ROM_HIP(hipGetLastError());
  ROM_TRY(rom_fem_unpack_reduced_async(f, Wc.b, 0, nz, Yfull.b, 0));
  ROM_TRY(rom_expand_batch_async(f, ones.b, nz, Yfull.b, 0, V, v_row0));
  // the map is orthonormal to the accuracy of its factor (~1e-9 relative to the leading directions): clean up (symmetric
  // orthonormalisation through the eigen-decomposition of the n x n Gram matrix: exact for any defect, one round)
  {
    Tmp Ys;
    ROM_TRY(Ys.get(ctx, size_t(nz) * dim));
    ROM_TRY(romb_gram_transform(ctx, V->p + v_row0 * dim, Ys, nz, dim, SE_LOWDIN, 1e-30, 1));
  }
  if (nz < n) {  // more modes requested than the snapshot manifold has dimensions: completed like rom_pod completes
    ROM_TRY(rom_complete_orthonormal(ctx, V, v_row0, nz, n - nz, dim));
    info[1] += n - nz;
    if (info[7] == 0.0) info[7] = 1.0;   // (not "filled": the completed modes lie beyond the rank of the snapshot manifold)
  }
  ROM_TRY(rom_launch_rows_sign_flip(ctx, V->p + v_row0 * dim, n, dim));  // svd_flip(u_based_decision=False)
  ROM_TRY(rom_solve_status(ctx));   // (synchronises; the status word of the expansion)
  if (info_host) {
    info[4] += 2.0 * M * double(Kc) * k2 + 2.0 * nz * double(Kc) * double(dim);
    memcpy(info_host, info, sizeof(info));
  }
  return ROM_OK;
}
