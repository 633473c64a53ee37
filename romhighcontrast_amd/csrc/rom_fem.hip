// FE space + batched parametric solve (the snapshot sweep).
//
// Replaces SolutionsManagerFEM.__init__ (src/lib/SolutionsManagers.py:146-219), galerkin (:17-40)
// and generate_solutions (:64-68) of the reference.
//
// Algorithm (exact direct method, fp64): the coefficient is constant a_b on each unit block b, so
// inside block b the operator is a_b * L with L the Dirichlet 5-point Laplacian of the block --
// parameter independent.  Eliminating all block interiors leaves an SPD system on the interface
// vertices (edges between blocks + cross points)
//        S(a) u_G = g,      S(a) = A_GG(a) - sum_b a_b T_b,     g parameter independent,
// with T_b the (dense) Dirichlet-to-Neumann blocks of the unit square, identical for all blocks
// up to the side pairing (16 tables T[sr][sc]).  Per parameter the work is: assemble S(a) tile by
// tile (never stored: each tile is built in registers when it is factored), a left-looking 64x64
// tile Cholesky on MFMA with the forward substitution fused in, a backward substitution, and the
// harmonic extension  u_I,b = (h^2/a_b) W + sum_sides H_s u_G|side   as one batched MFMA GEMM
// that writes the snapshot rows straight into the caller's (M, dim) matrix.
//
// Setup tables come from the sine (DST-I) eigenbasis of the block:  H_0[(i,j),k] =
// sum_m Q[j,m] rho_m(i) Q[k,m], rho_m(i) = sinh((N-i) phi_m)/sinh(N phi_m), cosh phi_m = 2-cos(pi m/N);
// the other three sides are row permutations of H_0.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>

#include "rom_mma.h"

// ============================================================================================
// device-side view of a rom_fem
// ============================================================================================
struct FemDev {
  int nrb, ncb, N, n1, n1p, nr, nc, nGp, nGa, T, nslots, kblk, npre, nrhs;
  long long dim;
  const double* R;
  const double* vec;
  const RhsTerm* rhs;
  const PreEdge* pre;
  const double* A0;    // (n1*n1) x n1p : Q[j,mode] rho_mode(i), harmonic extension in the sine basis
  const double* Qp;    // n1p x n1p sine matrix (zero padded)
  const int* kmax;     // [N+1] modes (multiple of 16) that matter at distance d from a side
  const int* epos;     // position of every edge's n1p block in the interface vectors
  double* yhat;        // [Mc][nGp] sine coefficients of the interface values
  const double* Tm;
  const double* W;
  const double* g;
  const TileDesc* desc;
  const TileExtra* extra;
  const int* kptr;
  const int* kpair;
  const int* colptr;
  const int* colrow;
  const int* colti;
  const BlockSide* sides;
  const int* vmap;
  double* L;     // [Mc][nslots][64*64]
  double* invL;  // [Mc][T][64*64]
  double* y;     // [Mc][nGp]
  int* status;
};

static FemDev make_dev(const rom_fem* f) {
  FemDev d;
  d.nrb = f->nrb; d.ncb = f->ncb; d.N = f->N; d.n1 = f->n1; d.n1p = f->n1p; d.nr = f->nr; d.nc = f->nc;
  d.nGp = f->nGp; d.nGa = f->nGa; d.npre = f->npre; d.nrhs = f->nrhs; d.R = f->d_R; d.vec = f->d_vec;
  d.rhs = f->d_rhs; d.pre = f->d_pre; d.T = f->T; d.nslots = f->nslots; d.kblk = f->nrb * f->ncb; d.dim = f->dim;
  d.A0 = f->d_A0; d.Qp = f->d_Qp; d.kmax = f->d_kmax; d.epos = f->d_epos; d.yhat = f->d_yhat; d.Tm = f->d_Tm; d.W = f->d_W; d.g = f->d_g; d.desc = f->d_desc; d.extra = f->d_extra;
  d.kptr = f->d_kptr; d.kpair = f->d_kpair; d.colptr = f->d_colptr; d.colrow = f->d_colrow;
  d.colti = f->d_colti; d.sides = f->d_sides; d.vmap = f->d_vmap; d.L = f->d_L; d.invL = f->d_invL;
  d.y = f->d_y; d.status = f->ctx->d_status;
  return d;
}

// ============================================================================================
// setup kernels
// ============================================================================================
// A0[((i-1)*n1 + (j-1)) * n1p + m] = Q[j-1][m] * rho[m][i]   (rho stored [m][i], i = 0..N)
__global__ void k_build_A0(double* A0, const double* Qp, const double* rho, int n1, int n1p, int N) {
  size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  size_t total = size_t(n1) * n1 * n1p;
  if (idx >= total) return;
  int m = int(idx % n1p);
  size_t ij = idx / n1p;
  int j = int(ij % n1) + 1, i = int(ij / n1) + 1;
  A0[idx] = (m < n1) ? Qp[size_t(j - 1) * n1p + m] * rho[size_t(m) * (N + 1) + i] : 0.0;
}

// row of H0 that holds the extension from side s evaluated at interior vertex (i,j), 1-based
__host__ __device__ inline int h0_row(int s, int i, int j, int N, int n1) {
  int ii, jj;
  switch (s) {
    case 0: ii = i; jj = j; break;
    case 1: ii = N - i; jj = j; break;
    case 2: ii = j; jj = i; break;
    default: ii = N - j; jj = i; break;
  }
  return (ii - 1) * n1 + (jj - 1);
}

// Tm[sr*4+sc][t][k] = H_sc[adjacent interior vertex of node t+1 on side sr][k]
__global__ void k_build_Tm(double* Tm, const double* H0, int n1, int n1p, int N) {
  size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  size_t per = size_t(n1p) * n1p;
  if (idx >= 16 * per) return;
  int tm = int(idx / per);
  int t = int((idx % per) / n1p), k = int(idx % n1p);
  int sr = tm >> 2, sc = tm & 3;
  double v = 0.0;
  if (t < n1 && k < n1) {
    int i, j;
    switch (sr) {
      case 0: i = 1; j = t + 1; break;
      case 1: i = N - 1; j = t + 1; break;
      case 2: i = t + 1; j = 1; break;
      default: i = t + 1; j = N - 1; break;
    }
    v = H0[size_t(h0_row(sc, i, j, N, n1)) * n1p + k];
  }
  Tm[idx] = v;
}

// ============================================================================================
// interface tile assembly  (A_GG(a) - sum_b a_b T_b, one entry)
// ============================================================================================
// Per-system scalar coefficients of one tile (uniform over the workgroup, computed once per thread):
// the tile is a linear combination of parameter-independent tables with these weights.
struct TileCoef {
  double cT[2];                             // - a_b                              (Schur terms)
  double cEE[4], cEX[4], cXE[4], cXX[4];    // pre-eliminated edge terms by (row kind, col kind), negated
  double dg, off;                           // tridiagonal A_GammaGamma part of a same-edge tile
};

__device__ inline void tile_coefs(TileCoef& tc, const TileDesc& d, const double* __restrict__ am) {
#pragma unroll
  for (int t = 0; t < 2; ++t) tc.cT[t] = t < d.nterms ? -am[d.term[t].blk] : 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    tc.cEE[t] = tc.cEX[t] = tc.cXE[t] = tc.cXX[t] = 0.0;
    if (t < d.npre) {
      const PreTerm& pt = d.pre[t];
      const double se = am[pt.e0] + am[pt.e1];
      const double ar = pt.brow >= 0 ? am[pt.brow] : 0.0, ac = pt.bcol >= 0 ? am[pt.bcol] : 0.0;
      tc.cEE[t] = -(ar * ac / se);  // c_row c_col / s_e with c = a_b on edge nodes, s_e / 2 on cross slots
      tc.cEX[t] = -(ar / 2);
      tc.cXE[t] = -(ac / 2);
      tc.cXX[t] = -(se / 4);
    }
  }
  tc.dg = tc.off = 0.0;
  if (d.same_edge) {
    const double a0 = am[d.b0], a1 = am[d.b1];
    // oracle order: k[r-1,c-1] + k[r-1,c] + k[r,c-1] + k[r,c]
    tc.dg = d.hv == 0 ? ((a0 + a0) + a1) + a1 : ((a0 + a1) + a0) + a1;
    tc.off = -(a1 + a0) / 2;
  }
}

// This thread's share of the assembled interface tile: row (t >> 2), 16 consecutive columns starting
// at (t & 3) * 16.  In this layout every table is read with 16-byte loads, 128 contiguous bytes per
// thread and table; the values wait in registers while the MFMA k-loop runs.
struct STile {
  double v[16];
};

__device__ inline void s_tile_load(STile& st, const TileDesc& d, const FemDev& f, const double* __restrict__ am) {
  TileCoef tc;
  tile_coefs(tc, d, am);
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
#pragma unroll
  for (int x = 0; x < 16; ++x) st.v[x] = 0.0;
  const bool re = r < d.nvr;
  if (r < d.ndr) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < d.npre) {
        const PreTerm& pt = d.pre[t];
        const double2* src = reinterpret_cast<const double2*>(f.R + (size_t(pt.table) * f.n1p + (pt.r0 + r)) * f.n1p +
                                                              (pt.c0 + c0));
        const double ce_coef = re ? tc.cEE[t] : tc.cXE[t], cx_coef = re ? tc.cEX[t] : tc.cXX[t];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const double2 w = src[x];
          const int c = c0 + 2 * x;
          st.v[2 * x] += (c < d.nvc ? ce_coef : (c < d.ndc ? cx_coef : 0.0)) * w.x;
          st.v[2 * x + 1] += (c + 1 < d.nvc ? ce_coef : (c + 1 < d.ndc ? cx_coef : 0.0)) * w.y;
        }
      }
  }
  if (re) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < d.nterms) {
        const TileTerm& tt = d.term[t];
        const double2* src = reinterpret_cast<const double2*>(f.Tm + (size_t(tt.tmat) * f.n1p + (tt.r0 + r)) * f.n1p +
                                                              (tt.c0 + c0));
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const double2 w = src[x];  // the tables are zero beyond the edge nodes: no column mask needed
          st.v[2 * x] += tc.cT[t] * w.x;
          st.v[2 * x + 1] += tc.cT[t] * w.y;
        }
      }
    if (d.same_edge) {
      const int gr = d.lr0 + r;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int c = c0 + x, gc = d.lc0 + c;
        if (c < d.nvc) st.v[x] += gr == gc ? tc.dg : ((gr - gc == 1 || gc - gr == 1) ? tc.off : 0.0);
      }
    }
  }
  if (d.diag && r >= d.ndr) {
#pragma unroll
    for (int x = 0; x < 16; ++x)
      if (c0 + x == r) st.v[x] = 1.0;  // padding unknowns: identity
  }
}

// C(LDS tile) = S_tile - acc ; then the sparse cross-point extras
__device__ inline void tile_from_acc(double* Cb, const Acc& acc, const STile& st, const TileDesc& d, const FemDev& f,
                                     const double* __restrict__ am, const WavePos& wp) {
  {
    double2* dst = reinterpret_cast<double2*>(Cb + (threadIdx.x >> 2) * LDC + (threadIdx.x & 3) * 16);
#pragma unroll
    for (int x = 0; x < 8; ++x) dst[x] = double2{st.v[2 * x], st.v[2 * x + 1]};
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) Cb[acc_row(wp, i, g) * LDC + acc_col(wp, j)] -= acc.c[i][j][g];
  __syncthreads();
  for (int x = d.x0 + threadIdx.x; x < d.x1; x += blockDim.x) {
    const TileExtra& e = f.extra[x];
    double v = e.kind == 0 ? -(am[e.b[0]] + am[e.b[1]]) / 2
                           : ((am[e.b[0]] + am[e.b[1]]) + am[e.b[2]]) + am[e.b[3]];
    Cb[e.r * LDC + e.c] += v;
  }
  __syncthreads();
}

// LDS carve-up of the factor kernels: B staging (2 buffers) | union { A staging (2 buffers), C tile }
constexpr int FACT_LDS_DOUBLES = 2 * STAGE_DOUBLES + TILE_DOUBLES;  // 6528 doubles = 52,224 B -> 3 WG / CU
constexpr int KP_MAX = 64;                                            // k-pairs cached in LDS per pass

// acc += sum over the k-list of `slot` of L[slotA] * L[slotB]^T, one merged pipelined loop
template <class FP>
__device__ inline void accumulate_klist(const FemDev& f, int slot, const double* Lm, int* kp, FP active, Acc& acc,
                                        double* stA, double* stB, const WavePos& wp) {
  const int e0 = f.kptr[slot], e1 = f.kptr[slot + 1];
  const int srow = stage_row(), sseg = stage_seg();
  for (int eb = e0; eb < e1; eb += KP_MAX) {
    const int np = min(KP_MAX, e1 - eb);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * np; i += blockDim.x) kp[i] = f.kpair[2 * eb + i];
    __syncthreads();
    const double* base = Lm + srow * 64 + sseg;
    gemm_loop2(
        4 * np, [&](int ch, double* v) { load4_aligned(base + size_t(kp[2 * (ch >> 2)]) * 4096 + (ch & 3) * BK, v); },
        [&](int ch, double* v) { load4_aligned(base + size_t(kp[2 * (ch >> 2) + 1]) * 4096 + (ch & 3) * BK, v); },
        active, acc, stA, stB, wp);
  }
}

// ============================================================================================
// factorisation kernels
// ============================================================================================
__global__ void k_init_rhs(FemDev f, int Mc) {
  size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (idx >= size_t(Mc) * f.nGp) return;
  f.y[idx] = f.g[idx % f.nGp];
}

// rhs of the condensed system: g_f + sum_e diag(c_f) X_fe K^-1 g_e / s_e   (grid: 64 threads x (tile rows, Mc))
__global__ void k_rhs_pre(FemDev f, const double* __restrict__ a, int ntile_rows) {
  const int m = blockIdx.y, tile = blockIdx.x, r = threadIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double acc = 0.0;
  for (int t = 0; t < f.nrhs; ++t) {
    const RhsTerm& rt = f.rhs[t];
    if (rt.tile != tile || r >= rt.ndr) continue;
    const double se = am[rt.e0] + am[rt.e1];
    const double cr = r < rt.nvr ? (rt.brow >= 0 ? am[rt.brow] : 0.0) : se / 2;
    acc += cr / se * f.vec[rt.qoff + rt.lr0 + r];
  }
  f.y[size_t(m) * f.nGp + tile * 64 + r] += acc;
}

// Back substitution of the pre-eliminated edges: x_e = (w_e + sum_f B_fe (c_f . x_f)) / s_e as one batched
// MFMA GEMM: tile rows = systems, tile cols = nodes of e, K = positions on the neighbours.
// grid (n1p/64, ceil(Mc/64), npre)
__global__ __launch_bounds__(256) void k_back_pre(FemDev f, const double* __restrict__ a, int Mc) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  const WavePos wp;
  const PreEdge& pe = f.pre[blockIdx.z];
  const int srow = stage_row(), sseg = stage_seg();
  const int mA = blockIdx.y * 64 + srow;
  const int iB = blockIdx.x * 64 + srow;  // node of e handled by this thread's B row
  const double* am = mA < Mc ? a + size_t(mA) * f.kblk : nullptr;
  const double seA = am ? am[pe.e0] + am[pe.e1] : 0.0;
  Acc acc;
  acc_zero(acc);
  const int cps = f.n1p / BK;
  for (int q = 0; q < pe.nnb; ++q) {
    const PreNb nb = pe.nb[q];
    const double* pA = am ? f.y + size_t(mA) * f.nGp + nb.fpos + sseg : nullptr;
    const double* pB = f.R + (size_t(nb.bt) * f.n1p + iB) * f.n1p + sseg;
    const double cedge = (am && nb.blk >= 0) ? am[nb.blk] : 0.0;
    gemm_loop(
        cps,
        [&](int ch, double* v) {
          load4_aligned(pA ? pA + ch * BK : nullptr, v);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int k = ch * BK + sseg + x;
            v[x] *= k < f.n1 ? cedge : (k < f.n1 + nb.nused ? seA / 2 : 0.0);
          }
        },
        [&](int ch, double* v) { load4_aligned(pB + ch * BK, v); }, acc, lds, wp);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + acc_row(wp, i, g);
      if (m >= Mc) continue;
      const double* amr = a + size_t(m) * f.kblk;
      const double inv = 1.0 / (amr[pe.e0] + amr[pe.e1]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int node = blockIdx.x * 64 + acc_col(wp, jb);
        f.y[size_t(m) * f.nGp + pe.pos + node] = node < f.n1 ? (f.vec[pe.woff + node] + acc.c[i][jb][g]) * inv : 0.0;
      }
    }
}

// X_fe (n1p x n1p): rows = positions on the active edge f (edge nodes, then cross slots), cols = nodes of e
__global__ void k_build_X(double* X, const double* Tm, int n1, int n1p, int tmat, int nx, const int* xrow,
                          const int* xcol) {
  size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (idx >= size_t(n1p) * n1p) return;
  int r = int(idx / n1p), c = int(idx % n1p);
  double v = 0.0;
  if (tmat >= 0 && r < n1 && c < n1) v = Tm[(size_t(tmat) * n1p + r) * n1p + c];
  for (int x = 0; x < nx; ++x)
    if (r == xrow[x] && c == xcol[x]) v = 1.0;
  X[idx] = v;
}

// Diagonal tile j, step 1 of 3 (MFMA): C = S_jj - sum_k L_jk L_jk^T, written to the tile's L slot.
// Only the lower triangle is consumed by the factorisation: the wave owning the upper-right
// quadrant skips its MFMAs.
__global__ __launch_bounds__(256) void k_diag_update(FemDev f, const double* __restrict__ a, int slot) {
  // 36.9 KB: four workgroups per CU, i.e. all 1024 systems of a C2 step resident in one round
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  __shared__ int kp[2 * KP_MAX];
  double* stB = lds;
  double* stA = lds + 2 * STAGE_DOUBLES;
  double* Cb = lds;  // the C tile aliases the whole staging area (used after the k-loop only)
  static_assert(TILE_DOUBLES <= STAGE_TOTAL, "C tile must fit in the staging area");
  const int m = blockIdx.x;
  const WavePos wp;
  const TileDesc& d = f.desc[slot];
  const double* am = a + size_t(m) * f.kblk;
  double* Lm = f.L + size_t(m) * f.nslots * 4096;
  STile st;
  s_tile_load(st, d, f, am);  // table reads fly under the MFMAs below
  Acc acc;
  acc_zero(acc);
  const bool lower = !(wp.wr == 0 && wp.wc == 1);
  accumulate_klist(f, slot, Lm, kp, [&](int) { return lower; }, acc, stA, stB, wp);
  tile_from_acc(Cb, acc, st, d, f, am, wp);
  double* Lout = Lm + size_t(slot) * 4096;
  for (int idx = threadIdx.x; idx < 4096; idx += 256) Lout[idx] = Cb[(idx >> 6) * LDC + (idx & 63)];
}

// 1/sqrt(d) for a positive normal d: hardware seed (v_rsq_f64, ~2^-26 relative error) + two Newton
// steps -> within 1-2 ulp; a fraction of the dependent-instruction chain of 1.0 / sqrt(d).
__device__ inline double rsqrt_newton(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * fma(-hd * y, y, 1.5);
  y = y * fma(-hd * y, y, 1.5);
  return y;
}

__device__ inline double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Diagonal tile j, step 2 of 3: in-register Cholesky, ONE WAVE per system (all systems of the batch
// resident at once, one wave per SIMD).  Lane r keeps row r of the 64x64 tile in registers; the
// elimination is fully unrolled (static register indices), column j of L is broadcast through LDS
// each step; no barriers.  (Steps 2 and 3 are separate kernels because hipcc's register allocation
// collapses into scratch when the two fully unrolled phases share one function.)
__global__ __launch_bounds__(64) void k_diag_potrf(FemDev f, int slot) {
  __shared__ __align__(16) double Ls[64 * LDC];
  __shared__ __align__(16) double lv[2][64];
  const int m = blockIdx.x, lane = threadIdx.x;
  double* Lt = f.L + (size_t(m) * f.nslots + slot) * 4096;
  for (int i = 0; i < 64; ++i) Ls[i * LDC + lane] = Lt[i * 64 + lane];
  __syncthreads();
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; c += 2) {
    double2 v = *reinterpret_cast<const double2*>(&Ls[lane * LDC + c]);
    a[c] = v.x;
    a[c + 1] = v.y;
  }
  bool bad = false;
#pragma unroll
  for (int jj = 0; jj < 64; ++jj) {
    const double dj = readlane_f64(a[jj], jj);
    bad = bad || !(dj > 0.0);
    const double rs = rsqrt_newton(dj);
    const double l = a[jj] * rs;  // L[lane][jj] for lane >= jj
    a[jj] = l;
    if (jj < 63) {
      double* bv = lv[jj & 1];
      bv[lane] = l;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 64; ++c)
        if (c > jj) a[c] -= l * bv[c];  // constant trip count so that both loops unroll fully
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (bad && lane == 0) atomicOr(f.status, 1);
  __syncthreads();
  // L (lower, zero above the diagonal) back through LDS, coalesced to HBM
#pragma unroll
  for (int c = 0; c < 64; ++c) Ls[lane * LDC + c] = c <= lane ? a[c] : 0.0;
  __syncthreads();
  for (int i = 0; i < 64; ++i) Lt[i * 64 + lane] = Ls[i * LDC + lane];
}

// Diagonal tile j, step 3 of 3 (one wave per system): y_j <- L_jj^-1 y_j by column-oriented
// substitution, and X = L_jj^-1 with lane c owning column c of X in registers:
// X[r][c] = (delta_rc - sum_{k<r} L[r][k] X[k][c]) / L[r][r]; L is read from LDS with wave-uniform
// addresses (broadcast), every row of X is stored coalesced.
__global__ __launch_bounds__(64) void k_diag_inverse(FemDev f, int slot, int j) {
  __shared__ __align__(16) double Ls[64 * LDC];
  __shared__ double rinv[64];
  const int m = blockIdx.x, lane = threadIdx.x;
  const double* Lt = f.L + (size_t(m) * f.nslots + slot) * 4096;
  for (int i = 0; i < 64; ++i) Ls[i * LDC + lane] = Lt[i * 64 + lane];
  __syncthreads();
  rinv[lane] = 1.0 / Ls[lane * LDC + lane];
  __syncthreads();
  double g = f.y[size_t(m) * f.nGp + j * 64 + lane];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const double yk = readlane_f64(g, k) * rinv[k];
    if (lane == k) g = yk;
    else if (lane > k) g -= Ls[lane * LDC + k] * yk;
  }
  f.y[size_t(m) * f.nGp + j * 64 + lane] = g;
  double* It = f.invL + (size_t(m) * f.T + j) * 4096;
  double x[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    // four independent partial sums: the dot product is otherwise one dependent FMA chain of length r
    double s0 = (r == lane) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
      if (k < r) s0 -= Ls[r * LDC + k] * x[k];
      if (k + 1 < r) s1 -= Ls[r * LDC + k + 1] * x[k + 1];
      if (k + 2 < r) s2 -= Ls[r * LDC + k + 2] * x[k + 2];
      if (k + 3 < r) s3 -= Ls[r * LDC + k + 3] * x[k + 3];
    }
    x[r] = ((s0 + s1) + (s2 + s3)) * rinv[r];
    It[r * 64 + lane] = x[r];
  }
}

// Sub-diagonal tiles of column j: C = S_ij - sum_k L_ik L_jk^T ; L_ij = C invL_jj^T ;
// y_i -= L_ij y_j.  invL_jj is lower triangular: the waves owning output columns 0..31 only need
// k < 32 of the second product.
__global__ __launch_bounds__(256) void k_factor_panel(FemDev f, const double* __restrict__ a, int j, int Mc) {
  __shared__ __align__(16) double lds[FACT_LDS_DOUBLES];
  __shared__ int kp[2 * KP_MAX];
  __shared__ double yj[64];
  double* stB = lds;
  double* stA = lds + 2 * STAGE_DOUBLES;
  double* Cb = lds + 2 * STAGE_DOUBLES;
  // XCD-aware mapping: block ids are dealt round-robin to the 8 XCDs (each with its own L2); all row
  // tiles of one system are given ids of the same residue mod 8 so that the L_jk / invL_jj tiles they
  // share are served by one L2.  (Placement only affects speed, never correctness.)
  const int nrows = f.colptr[j + 1] - f.colptr[j];
  int m, row;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int full = (Mc >> 3) << 3;  // systems covered by complete groups of 8
    m = xcd + 8 * (idx / nrows);
    row = idx % nrows;
    if (m >= full) {  // ragged tail (Mc % 8 systems): plain order
      const int tb = b - full * nrows;
      m = full + tb / nrows;
      row = tb % nrows;
    }
  }
  const int ent = f.colptr[j] + row;
  const int slot = f.colrow[ent];
  const int ti = f.colti[ent];
  const WavePos wp;
  const TileDesc& d = f.desc[slot];
  const double* am = a + size_t(m) * f.kblk;
  double* Lm = f.L + size_t(m) * f.nslots * 4096;
  const int t = threadIdx.x;
  STile st;
  s_tile_load(st, d, f, am);
  if (t < 64) yj[t] = f.y[size_t(m) * f.nGp + j * 64 + t];
  Acc acc;
  acc_zero(acc);
  accumulate_klist(f, slot, Lm, kp, [](int) { return true; }, acc, stA, stB, wp);
  tile_from_acc(Cb, acc, st, d, f, am, wp);

  // X = C * invL_jj^T
  acc_zero(acc);
  const double* I = f.invL + (size_t(m) * f.T + j) * 4096 + stage_row() * 64 + stage_seg();
  const int kmax = (wp.wc + 1) * 32;  // invL[c][k] = 0 for k > c
  gemm_loop_Atile(Cb, 4, [&](int ch, double* v) { load4_aligned(I + ch * BK, v); },
                  [&](int ch) { return ch * BK < kmax; }, acc, stB, wp);

  double* Lout = Lm + size_t(slot) * 4096;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int r = acc_row(wp, i, g), c = acc_col(wp, jb);
        double v = acc.c[i][jb][g];
        Lout[r * 64 + c] = v;
        Cb[r * LDC + c] = v;
      }
  __syncthreads();
  if (t < 64) {
    double s = 0.0;
    for (int k = 0; k < 64; ++k) s += Cb[t * LDC + k] * yj[k];
    f.y[size_t(m) * f.nGp + ti * 64 + t] -= s;
  }
}

// x = L^{-T} y, one workgroup per system, x kept in LDS, written back over y
__global__ __launch_bounds__(256) void k_backsolve(FemDev f) {
  extern __shared__ __align__(16) double xs[];  // nGa
  __shared__ double red[4][64];
  __shared__ double vs[64];
  const int m = blockIdx.x;
  const int t = threadIdx.x, c = t & 63, part = t >> 6;
  const double* Lm = f.L + size_t(m) * f.nslots * 4096;
  double* ym = f.y + size_t(m) * f.nGp;
  for (int j = f.T - 1; j >= 0; --j) {
    double s = 0.0;
    for (int e = f.colptr[j]; e < f.colptr[j + 1]; ++e) {
      const double* Lt = Lm + size_t(f.colrow[e]) * 4096;
      const double* xi = xs + f.colti[e] * 64;
#pragma unroll 4
      for (int rr = part * 16; rr < part * 16 + 16; ++rr) s += Lt[rr * 64 + c] * xi[rr];
    }
    red[part][c] = s;
    __syncthreads();
    if (t < 64) vs[t] = ym[j * 64 + t] - (red[0][t] + red[1][t] + red[2][t] + red[3][t]);
    __syncthreads();
    const double* It = f.invL + (size_t(m) * f.T + j) * 4096;
    s = 0.0;
#pragma unroll 4
    for (int rr = part * 16; rr < part * 16 + 16; ++rr) s += It[rr * 64 + c] * vs[rr];
    red[part][c] = s;
    __syncthreads();
    if (t < 64) xs[j * 64 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    __syncthreads();
  }
  for (int v = t; v < f.nGa; v += 256) ym[v] = xs[v];
}

// ============================================================================================
// harmonic extension + scatter: writes the snapshot rows
// ============================================================================================
// Sine transform of the interface values, edge by edge:  yhat[m, pos_e + mode] = sum_k y[m, pos_e + k] Q[k, mode]
// (Q symmetric).  grid (n1p/64, ceil(Mc/64), n_edges); tile rows = systems, tile cols = modes.
__global__ __launch_bounds__(256) void k_edge_transform(FemDev f, int Mc) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  const WavePos wp;
  const int pos = f.epos[blockIdx.z];
  const int srow = stage_row(), sseg = stage_seg();
  const int mA = blockIdx.y * 64 + srow;
  const double* pA = mA < Mc ? f.y + size_t(mA) * f.nGp + pos + sseg : nullptr;
  const double* pB = f.Qp + size_t(blockIdx.x * 64 + srow) * f.n1p + sseg;
  Acc acc;
  acc_zero(acc);
  gemm_loop(
      f.n1p / BK, [&](int ch, double* v) { load4_aligned(pA ? pA + ch * BK : nullptr, v); },
      [&](int ch, double* v) { load4_aligned(pB + ch * BK, v); }, acc, lds, wp);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + acc_row(wp, i, g);
      if (m >= Mc) continue;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
        f.yhat[size_t(m) * f.nGp + pos + blockIdx.x * 64 + acc_col(wp, jb)] = acc.c[i][jb][g];
    }
}

// Harmonic extension in the sine basis of each side:
//   U_I,b[m,(i,j)] = (h^2/a_b) W[i,j] + sum_{sides s} sum_mode yhat_s[m, mode] * A0[pi_s(i,j)][mode],
//   A0[(i',j'), mode] = Q[j', mode] rho_mode(i'),   rho_mode(i') ~ exp(-i' phi_mode).
// Far from a side only the low modes survive in fp64: kmax[d] (multiple of 16) is the number of modes with
// rho_mode(d) above 1e-24, so a tile whose vertices are at distance >= d from side s stops its K loop
// there (terms below 1e-24 of the leading ones cannot change an fp64 sum).
// Tile columns = a 4 x 16 patch of interior vertices (so that the distance to all four sides is bounded
// below per tile and rows are still written in 128-byte segments); tile rows = systems.
// grid (patches, ceil(Mc/64), nblocks)
__global__ __launch_bounds__(256) void k_extend(FemDev f, const double* __restrict__ a, int Mc,
                                                double* __restrict__ U, long long row0) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  double* stage = lds;
  const WavePos wp;
  const int b = blockIdx.z;
  const int p = b / f.ncb, q = b % f.ncb;
  const int n1 = f.n1, N = f.N;
  const BlockSide sd = f.sides[b];
  const int srow = stage_row(), sseg = stage_seg();
  const int npj = (n1 + 15) / 16;                     // patches per patch row
  const int pi = blockIdx.x / npj, pj = blockIdx.x % npj;
  const int i0 = 4 * pi + 1, j0 = 16 * pj + 1;        // first vertex of the patch (1-based)
  const int i1 = min(i0 + 3, n1), j1 = min(j0 + 15, n1);

  const int mA = blockIdx.y * 64 + srow;
  const double* Arow = mA < Mc ? f.yhat + size_t(mA) * f.nGp + sseg : nullptr;
  const int iB = i0 + (srow >> 4), jB = j0 + (srow & 15);  // vertex of this thread's B row
  const bool vB = iB <= n1 && jB <= n1;

  Acc acc;
  acc_zero(acc);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = s == 0 ? sd.off[0] : s == 1 ? sd.off[1] : s == 2 ? sd.off[2] : sd.off[3];
    if (off < 0) continue;  // side on the domain boundary (uniform branch)
    const int dist = s == 0 ? i0 : s == 1 ? N - i1 : s == 2 ? j0 : N - j1;  // closest vertex of the patch
    const int nch = f.kmax[dist] / BK;
    const double* pA = Arow ? Arow + off : nullptr;
    const double* pB = vB ? f.A0 + size_t(h0_row(s, iB, jB, N, n1)) * f.n1p + sseg : nullptr;
    gemm_loop(
        nch, [&](int ch, double* v) { load4_aligned(pA ? pA + ch * BK : nullptr, v); },
        [&](int ch, double* v) { load4_aligned(pB ? pB + ch * BK : nullptr, v); }, acc, stage, wp);
  }

  const double h2 = 1.0 / (double(N) * double(N));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int m = blockIdx.y * 64 + acc_row(wp, i, g);
      if (m >= Mc) continue;
      double sc = h2 / a[size_t(m) * f.kblk + b];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int cidx = acc_col(wp, jb);
        const int ii = i0 + (cidx >> 4) - 1, jj = j0 + (cidx & 15) - 1;  // 0-based interior indices
        if (ii >= n1 || jj >= n1) continue;
        long long gidx = (long long)(p * N + ii) * f.nc + (q * N + jj);
        U[(row0 + m) * f.dim + gidx] = acc.c[i][jb][g] + sc * f.W[ii * n1 + jj];
      }
    }
}

__global__ void k_scatter_interface(FemDev f, int Mc, double* __restrict__ U, long long row0) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  int m = blockIdx.y;
  if (v >= f.nGp || m >= Mc) return;
  int gi = f.vmap[v];
  if (gi >= 0) U[(row0 + m) * f.dim + gi] = f.y[size_t(m) * f.nGp + v];
}

// stencil arrays for the API (einsum('pqij,pq->ij') in stencil form)
__global__ void k_assemble_stencil(FemDev f, const double* __restrict__ a, int M, double* __restrict__ diag,
                                   double* __restrict__ east, double* __restrict__ north) {
  __shared__ double am[64];  // coefficients of this parameter staged in LDS
  const int m = blockIdx.y;
  for (int i = threadIdx.x; i < f.kblk; i += blockDim.x) am[i] = a[size_t(m) * f.kblk + i];
  __syncthreads();
  long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= f.dim) return;
  int r = int(idx / f.nc) + 1, c = int(idx % f.nc) + 1;  // 1-based vertex coordinates
  int N = f.N, ncb = f.ncb;
  // kappa[line, col] = a[line / N][col / N]; the four cells around vertex (r, c)
  double k00 = am[((r - 1) / N) * ncb + (c - 1) / N];
  double k01 = am[((r - 1) / N) * ncb + c / N];
  double k10 = am[(r / N) * ncb + (c - 1) / N];
  double k11 = am[(r / N) * ncb + c / N];
  diag[size_t(m) * f.dim + idx] = ((k00 + k01) + k10) + k11;
  if (c < f.nc) east[size_t(m) * f.nr * (f.nc - 1) + size_t(r - 1) * (f.nc - 1) + (c - 1)] = -(k11 + k01) / 2;
  if (r < f.nr) north[size_t(m) * (f.nr - 1) * f.nc + size_t(r - 1) * f.nc + (c - 1)] = -(k11 + k10) / 2;
}

// ============================================================================================
// host: geometry, symbolic tile Cholesky, tables
// ============================================================================================
namespace {

struct Edge {
  int hv, p, q;  // hv 0: horizontal (r = pN, c in block column q); 1: vertical (c = qN, r in block row p)
  int b0, b1;    // up/dn or lf/rt block indices
};

template <class Tp>
int upload(Tp** dptr, const std::vector<Tp>& h) {
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(Tp);
  ROM_HIP(hipMalloc(dptr, bytes));
  if (!h.empty()) ROM_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
  return ROM_OK;
}

}  // namespace

extern "C" int rom_fem_destroy(rom_fem* f) {
  if (!f) return ROM_OK;
  hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->d_A0, f->d_Qp, f->d_kmax, f->d_epos, f->d_yhat, f->d_Tm, f->d_W, f->d_g, f->d_desc, f->d_extra, f->d_slot_of, f->d_kptr, f->d_kpair,
                  f->d_colptr, f->d_colrow, f->d_colti, f->d_sides, f->d_vmap, f->d_L, f->d_invL, f->d_y, f->d_R, f->d_vec,
                  f->d_rhs, f->d_pre};
  for (void* p : ptrs)
    if (p) hipFree(p);
  delete f;
  return ROM_OK;
}

extern "C" int rom_fem_create(rom_ctx* ctx, int nrb, int ncb, int N, rom_fem** out) {
  ROM_CHECK(ctx && out, "rom_fem_create: null argument");
  ROM_CHECK(nrb >= 1 && ncb >= 1 && N >= 2, "rom_fem_create: need nrb,ncb >= 1 and N >= 2 (got %d,%d,%d)", nrb, ncb, N);
  ROM_CHECK(nrb * ncb <= 64, "rom_fem_create: at most 64 blocks supported (got %d)", nrb * ncb);
  ROM_HIP(hipSetDevice(ctx->device));
  rom_fem* f = new rom_fem();
  f->ctx = ctx;
  f->nrb = nrb; f->ncb = ncb; f->N = N;
  const int n1 = N - 1;
  f->n1 = n1;
  f->tpe = (n1 + TB - 1) / TB;
  f->n1p = f->tpe * TB;
  f->nr = nrb * N - 1;
  f->nc = ncb * N - 1;
  f->dim = int64_t(f->nr) * f->nc;
  const int n1p = f->n1p, tpe = f->tpe;

  // ---- edges, crosses ------------------------------------------------------------------------
  std::vector<Edge> edges;
  std::map<std::pair<int, int>, int> hid, vid, xid;
  for (int p = 1; p < nrb; ++p)
    for (int q = 0; q < ncb; ++q) {
      hid[{p, q}] = int(edges.size());
      edges.push_back({0, p, q, (p - 1) * ncb + q, p * ncb + q});
    }
  for (int q = 1; q < ncb; ++q)
    for (int p = 0; p < nrb; ++p) {
      vid[{p, q}] = int(edges.size());
      edges.push_back({1, p, q, p * ncb + (q - 1), p * ncb + q});
    }
  std::vector<std::pair<int, int>> crosses;
  for (int p = 1; p < nrb; ++p)
    for (int q = 1; q < ncb; ++q) {
      xid[{p, q}] = int(crosses.size());
      crosses.push_back({p, q});
    }
  const int E = int(edges.size());
  const int ncross = int(crosses.size());
  f->nG = E * n1 + ncross;
  // block -> side -> edge id
  std::vector<std::array<int, 4>> bside(nrb * ncb);
  for (int p = 0; p < nrb; ++p)
    for (int q = 0; q < ncb; ++q) {
      auto& s = bside[p * ncb + q];
      s[0] = p >= 1 ? hid[{p, q}] : -1;
      s[1] = p + 1 < nrb ? hid[{p + 1, q}] : -1;
      s[2] = q >= 1 ? vid[{p, q}] : -1;
      s[3] = q + 1 < ncb ? vid[{p, q + 1}] : -1;
    }
  auto side_of = [&](int blk, int e) {
    for (int s = 0; s < 4; ++s)
      if (bside[blk][s] == e) return s;
    return -1;
  };

  // ---- cross <-> edge-end couplings ----------------------------------------------------------------
  struct XCpl { int cross, edge, node; };  // node: 0-based local node on the edge
  std::vector<XCpl> xc;
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    if (ed.hv == 0) {
      if (ed.q >= 1) xc.push_back({xid[{ed.p, ed.q}], e, 0});
      if (ed.q + 1 < ncb) xc.push_back({xid[{ed.p, ed.q + 1}], e, n1 - 1});
    } else {
      if (ed.p >= 1) xc.push_back({xid[{ed.p, ed.q}], e, 0});
      if (ed.p + 1 < nrb) xc.push_back({xid[{ed.p + 1, ed.q}], e, n1 - 1});
    }
  }
  // block adjacency of the edges
  std::vector<std::set<int>> adj(E);
  for (auto& s : bside)
    for (int x = 0; x < 4; ++x)
      for (int y = 0; y < 4; ++y)
        if (x != y && s[x] >= 0 && s[y] >= 0) adj[s[x]].insert(s[y]);
  auto shared_block = [&](int e1, int e2) {  // the one block two distinct edges can share, or -1
    for (int b1 : {edges[e1].b0, edges[e1].b1})
      for (int b2 : {edges[e2].b0, edges[e2].b1})
        if (b1 == b2) return b1;
    return -1;
  };
  const int free_per_edge = n1p - n1;

  // Everything below depends on which edges are eliminated in closed form (`pre`); if the cross points
  // cannot all be hosted in padding slots of active edges the set is dropped and the layout redone.
  std::vector<char> is_pre(E, 0);
  if (free_per_edge >= 1 && !getenv("ROMHC_NO_PREELIM")) {
    // maximal set of edges no two of which touch the same block (greedy): their self-interaction is
    // (a_b0 + a_b1) K with K parameter independent and they do not couple to each other
    std::vector<char> busy(nrb * ncb, 0);
    for (int e = 0; e < E; ++e)
      if (!busy[edges[e].b0] && !busy[edges[e].b1]) { is_pre[e] = 1; busy[edges[e].b0] = busy[edges[e].b1] = 1; }
  }
  std::vector<int> order, tile0, epos, used, xpos, overflow, pre_list, ppos;
  int nact = 0, xt0 = 0, nxt = 0, T = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    pre_list.clear();
    for (int e = 0; e < E; ++e)
      if (is_pre[e]) pre_list.push_back(e);
    // ordering graph of the active edges: shared block, or common neighbour of an eliminated edge
    std::vector<std::set<int>> g(E);
    for (int e = 0; e < E; ++e)
      if (!is_pre[e])
        for (int x : adj[e])
          if (!is_pre[x]) g[e].insert(x);
    for (int e : pre_list)
      for (int x : adj[e])
        for (int y : adj[e])
          if (x != y && !is_pre[x] && !is_pre[y]) g[x].insert(y);
    order.clear();
    std::vector<char> done(E, 0);
    nact = E - int(pre_list.size());
    for (int step = 0; step < nact; ++step) {  // greedy minimum degree
      int best = -1;
      size_t bd = 0;
      for (int e = 0; e < E; ++e) {
        if (done[e] || is_pre[e]) continue;
        if (best < 0 || g[e].size() < bd) { best = e; bd = g[e].size(); }
      }
      done[best] = 1;
      order.push_back(best);
      std::vector<int> nb(g[best].begin(), g[best].end());
      for (int x : nb) {
        g[x].erase(best);
        for (int y : nb)
          if (x != y) g[x].insert(y);
      }
    }
    tile0.assign(E, -1);
    epos.assign(E, -1);
    for (int pos = 0; pos < nact; ++pos) { tile0[order[pos]] = pos * tpe; epos[order[pos]] = pos; }
    // cross points go into the padding slots behind an adjacent active edge (the one eliminated last)
    used.assign(E, 0);
    xpos.assign(ncross, -1);
    overflow.clear();
    for (int x = 0; x < ncross; ++x) {
      int best = -1;
      for (auto& c : xc)
        if (c.cross == x && !is_pre[c.edge] && used[c.edge] < free_per_edge &&
            (best < 0 || epos[c.edge] > epos[best]))
          best = c.edge;
      if (best >= 0) xpos[x] = tile0[best] * TB + n1 + used[best]++;
      else overflow.push_back(x);
    }
    if (!overflow.empty() && !pre_list.empty()) {  // keep the closed-form elimination simple: no overflow tiles
      std::fill(is_pre.begin(), is_pre.end(), 0);
      continue;
    }
    break;
  }
  xt0 = nact * tpe;                                    // first overflow tile
  nxt = (int(overflow.size()) + TB - 1) / TB;          // overflow tiles
  for (size_t o = 0; o < overflow.size(); ++o) xpos[overflow[o]] = xt0 * TB + int(o);
  T = xt0 + nxt;
  const int npre = int(pre_list.size());
  f->T = T;
  f->nGa = T * TB;
  f->npre = npre;
  f->nGp = f->nGa + npre * n1p;
  ppos.assign(E, -1);  // position of an edge's n1p block in the interface vectors
  for (int e = 0; e < E; ++e)
    if (!is_pre[e]) ppos[e] = tile0[e] * TB;
  for (int i = 0; i < npre; ++i) ppos[pre_list[i]] = f->nGa + i * n1p;
  // tile -> (edge id or -1 for an overflow tile, local tile index, number of defined rows)
  std::vector<int> tile_edge(T, -1), tile_loc(T, 0), tile_ndr(T, 0);
  for (int e = 0; e < E; ++e) {
    if (is_pre[e]) continue;
    for (int x = 0; x < tpe; ++x) {
      tile_edge[tile0[e] + x] = e;
      tile_loc[tile0[e] + x] = x;
      tile_ndr[tile0[e] + x] = std::max(0, std::min(TB, n1 + used[e] - x * TB));
    }
  }
  for (int x = 0; x < nxt; ++x) {
    tile_loc[xt0 + x] = x;
    tile_ndr[xt0 + x] = std::min(TB, int(overflow.size()) - x * TB);
  }

  // ---- neighbours of every eliminated edge: active edges sharing a block, or hosting a cross point
  //      that touches one of its ends -------------------------------------------------------------------
  struct Nb { int f; int blk; std::vector<std::pair<int, int>> xs; };  // xs: (row on f, end node on e)
  std::vector<std::vector<Nb>> nbs(npre);
  for (int i = 0; i < npre; ++i) {
    const int e = pre_list[i];
    std::map<int, Nb> m;
    for (int x : adj[e])
      if (!is_pre[x]) m[x] = Nb{x, shared_block(e, x), {}};
    for (auto& c : xc) {
      if (c.edge != e) continue;
      const int host = tile_edge[xpos[c.cross] / TB];
      if (!m.count(host)) m[host] = Nb{host, -1, {}};
      m[host].xs.push_back({xpos[c.cross] - tile0[host] * TB, c.node});
    }
    for (auto& kv : m) nbs[i].push_back(kv.second);
    if (nbs[i].size() > 8) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
  }
  auto nb_tiles = [&](const Nb& nb) {  // tiles of the neighbour that actually couple to e
    std::set<int> t;
    if (nb.blk >= 0)
      for (int x = 0; x < tpe; ++x) t.insert(tile0[nb.f] + x);
    for (auto& xr : nb.xs) t.insert(tile0[nb.f] + xr.first / TB);
    return t;
  };

  // ---- tile mask + symbolic fill --------------------------------------------------------------------
  std::vector<char> mask(size_t(T) * T, 0);
  auto M_ = [&](int i, int j) -> char& { return mask[size_t(i) * T + j]; };
  for (int e = 0; e < E; ++e)
    for (int e2 = 0; e2 < E; ++e2)
      if (!is_pre[e] && !is_pre[e2] && (e == e2 || adj[e].count(e2)))
        for (int x = 0; x < tpe; ++x)
          for (int y = 0; y < tpe; ++y) M_(tile0[e] + x, tile0[e2] + y) = 1;
  for (int x = 0; x < ncross; ++x) M_(xpos[x] / TB, xpos[x] / TB) = 1;
  for (auto& c : xc) {
    if (is_pre[c.edge]) continue;
    int ti = xpos[c.cross] / TB, tj = tile0[c.edge] + c.node / TB;
    M_(ti, tj) = M_(tj, ti) = 1;
  }
  for (int i = 0; i < npre; ++i) {  // fill created by the closed-form elimination
    std::set<int> ts;
    for (auto& nb : nbs[i])
      for (int t : nb_tiles(nb)) ts.insert(t);
    for (int x : ts)
      for (int y : ts) M_(x, y) = 1;
  }
  for (int k = 0; k < T; ++k)
    for (int i = k + 1; i < T; ++i)
      if (M_(i, k))
        for (int j = k + 1; j <= i; ++j)
          if (M_(j, k)) M_(i, j) = M_(j, i) = 1;

  f->slot_of.assign(size_t(T) * T, -1);
  std::vector<std::pair<int, int>> slots;
  f->colptr.assign(T + 1, 0);
  f->diag_slot.assign(T, -1);
  for (int j = 0; j < T; ++j) {
    f->diag_slot[j] = int(slots.size());
    f->slot_of[size_t(j) * T + j] = int(slots.size());
    slots.push_back({j, j});
    for (int i = j + 1; i < T; ++i)
      if (M_(i, j)) {
        f->slot_of[size_t(i) * T + j] = int(slots.size());
        f->colrow.push_back(int(slots.size()));
        f->colti.push_back(i);
        slots.push_back({i, j});
      }
    f->colptr[j + 1] = int(f->colrow.size());
  }
  f->nslots = int(slots.size());
  f->kptr.assign(f->nslots + 1, 0);
  double flops = 0;
  for (int s = 0; s < f->nslots; ++s) {
    int i = slots[s].first, j = slots[s].second;
    for (int k = 0; k < j; ++k)
      if (M_(i, k) && M_(j, k)) {
        f->kpair.push_back(f->slot_of[size_t(i) * T + k]);
        f->kpair.push_back(f->slot_of[size_t(j) * T + k]);
        flops += 2.0 * TB * TB * TB;
      }
    f->kptr[s + 1] = int(f->kpair.size() / 2);
    flops += (i == j) ? TB * double(TB) * TB / 3.0 : 2.0 * TB * TB * TB;  // potrf | trsm-as-gemm
  }

  // ---- tables of the eliminated edges: which R = X_f K^-1 X_f'^T are needed -----------------------------
  // table ids: R tables first (one per (pre edge, f, f') actually used by a tile), then the B^T tables
  std::map<std::array<int, 3>, int> rid;  // (pre index, f, f') -> table id
  auto nb_index = [&](int i, int fedge) {
    for (size_t q = 0; q < nbs[i].size(); ++q)
      if (nbs[i][q].f == fedge) return int(q);
    return -1;
  };

  // ---- tile descriptors + extras ------------------------------------------------------------------------
  std::vector<TileExtra> extras;
  f->desc.resize(f->nslots);
  for (int s = 0; s < f->nslots; ++s) {
    TileDesc d;
    memset(&d, 0, sizeof(d));
    d.ti = slots[s].first;
    d.tj = slots[s].second;
    d.diag = d.ti == d.tj;
    int er = tile_edge[d.ti], ec = tile_edge[d.tj];
    d.lr0 = tile_loc[d.ti] * TB;
    d.lc0 = tile_loc[d.tj] * TB;
    d.nvr = er >= 0 ? std::max(0, std::min(TB, n1 - d.lr0)) : 0;  // edge nodes in the tile rows / cols
    d.nvc = ec >= 0 ? std::max(0, std::min(TB, n1 - d.lc0)) : 0;
    d.ndr = tile_ndr[d.ti];                                        // + cross slots: rows that are unknowns
    d.ndc = tile_ndr[d.tj];
    if (er >= 0 && ec >= 0) {
      // blocks adjacent to both edges
      int cand[2] = {edges[er].b0, edges[er].b1};
      for (int cb : cand) {
        int sr = side_of(cb, er), sc = side_of(cb, ec);
        if (sr >= 0 && sc >= 0) {
          if (d.nterms >= 2) { rom_set_error("internal: more than 2 Schur terms per tile"); return ROM_ERR_INVALID; }
          d.term[d.nterms++] = TileTerm{cb, sr * 4 + sc, d.lr0, d.lc0};
        }
      }
      if (er == ec) {
        d.same_edge = 1;
        d.hv = edges[er].hv;
        d.b0 = edges[er].b0;
        d.b1 = edges[er].b1;
      }
      // closed-form contributions of the eliminated edges that see both er and ec
      for (int i = 0; i < npre; ++i) {
        int qr = nb_index(i, er), qc = nb_index(i, ec);
        if (qr < 0 || qc < 0) continue;
        if (!nb_tiles(nbs[i][qr]).count(d.ti) || !nb_tiles(nbs[i][qc]).count(d.tj)) continue;
        std::array<int, 3> key{i, er, ec};
        if (!rid.count(key)) { int id = int(rid.size()); rid[key] = id; }
        if (d.npre >= 4) { rom_set_error("internal: more than 4 pre-elimination terms per tile"); return ROM_ERR_INVALID; }
        const Edge& pe = edges[pre_list[i]];
        d.pre[d.npre++] = PreTerm{rid[key], d.lr0, d.lc0, nbs[i][qr].blk, nbs[i][qc].blk, pe.b0, pe.b1};
      }
    }
    d.x0 = int(extras.size());
    // entries that involve cross points (placed by interface position)
    for (int x = 0; x < ncross; ++x) {
      if (d.diag && xpos[x] / TB == d.ti) {
        int p = crosses[x].first, q = crosses[x].second, l = xpos[x] % TB;
        extras.push_back(TileExtra{l, l, 1, {(p - 1) * ncb + (q - 1), (p - 1) * ncb + q, p * ncb + (q - 1), p * ncb + q}});
      }
    }
    for (auto& c : xc) {
      if (is_pre[c.edge]) continue;  // folded into the R tables
      int pa = xpos[c.cross], pb = tile0[c.edge] * TB + c.node;
      // coupling value uses the two blocks of the edge: -(k[r,c] + k[r-1,c])/2 resp. -(k[r,c]+k[r,c-1])/2
      const int ta = pa / TB, tb = pb / TB;
      const TileExtra rc{pa % TB, pb % TB, 0, {edges[c.edge].b1, edges[c.edge].b0, 0, 0}};
      const TileExtra cr{pb % TB, pa % TB, 0, {edges[c.edge].b1, edges[c.edge].b0, 0, 0}};
      if (ta == tb) {
        if (d.diag && d.ti == ta) { extras.push_back(rc); extras.push_back(cr); }
      } else if (d.ti == ta && d.tj == tb) {
        extras.push_back(rc);
      } else if (d.ti == tb && d.tj == ta) {
        extras.push_back(cr);
      }
    }
    d.x1 = int(extras.size());
    f->desc[s] = d;
  }

  // ---- block sides, vmap ------------------------------------------------------------------------------------
  f->sides.resize(nrb * ncb);
  for (int b = 0; b < nrb * ncb; ++b)
    for (int s = 0; s < 4; ++s) f->sides[b].off[s] = bside[b][s] >= 0 ? ppos[bside[b][s]] : -1;
  std::vector<int> vmap(std::max(f->nGp, 1), -1);
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    for (int t = 0; t < n1; ++t) {
      int r, c;  // 1-based inner vertex coordinates
      if (ed.hv == 0) { r = ed.p * N; c = ed.q * N + t + 1; }
      else { r = ed.p * N + t + 1; c = ed.q * N; }
      vmap[ppos[e] + t] = (r - 1) * f->nc + (c - 1);
    }
  }
  for (int x = 0; x < ncross; ++x) {
    int r = crosses[x].first * N, c = crosses[x].second * N;
    vmap[xpos[x]] = (r - 1) * f->nc + (c - 1);
  }

  // ---- unit-block tables in long double ----------------------------------------------------------------
  typedef long double ld;
  const ld PI = acosl(-1.0L);
  std::vector<ld> Q(size_t(n1) * n1), lam(n1);
  for (int j = 1; j <= n1; ++j) {
    lam[j - 1] = 2.0L - 2.0L * cosl(PI * j / N);
    for (int m = 1; m <= n1; ++m) Q[size_t(j - 1) * n1 + (m - 1)] = sqrtl(2.0L / N) * sinl(PI * j * m / (ld)N);
  }
  std::vector<double> rho(size_t(n1) * (N + 1));
  for (int m = 0; m < n1; ++m) {
    ld phi = acoshl(1.0L + lam[m] / 2.0L);
    ld den = -expm1l(-2.0L * N * phi);  // 1 - exp(-2 N phi)
    for (int i = 0; i <= N; ++i) {
      ld num = expl(-phi * i) * (-expm1l(-2.0L * (N - i) * phi));
      rho[size_t(m) * (N + 1) + i] = double(num / den);
    }
  }
  // W = L^{-1} 1 = Q (s s^T / (lam_l + lam_m)) Q
  std::vector<ld> sv(n1, 0.0L), Z(size_t(n1) * n1), ZQ(size_t(n1) * n1);
  for (int m = 0; m < n1; ++m)
    for (int j = 0; j < n1; ++j) sv[m] += Q[size_t(j) * n1 + m];
  for (int l = 0; l < n1; ++l)
    for (int m = 0; m < n1; ++m) Z[size_t(l) * n1 + m] = sv[l] * sv[m] / (lam[l] + lam[m]);
  for (int l = 0; l < n1; ++l)
    for (int j = 0; j < n1; ++j) {
      ld s = 0;
      for (int m = 0; m < n1; ++m) s += Z[size_t(l) * n1 + m] * Q[size_t(j) * n1 + m];
      ZQ[size_t(l) * n1 + j] = s;
    }
  std::vector<double> W(size_t(n1) * n1);
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n1; ++j) {
      ld s = 0;
      for (int l = 0; l < n1; ++l) s += Q[size_t(i) * n1 + l] * ZQ[size_t(l) * n1 + j];
      W[size_t(i) * n1 + j] = double(s);
    }
  const double h2 = 1.0 / (double(N) * double(N));
  f->g_host.assign(std::max(f->nGp, 1), 0.0);
  auto Wat = [&](int i, int j) { return W[size_t(i - 1) * n1 + (j - 1)]; };
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    for (int t = 1; t <= n1; ++t) {
      double w = ed.hv == 0 ? Wat(N - 1, t) + Wat(1, t) : Wat(t, N - 1) + Wat(t, 1);
      f->g_host[ppos[e] + t - 1] = h2 * (1.0 + w);
    }
  }
  for (int x = 0; x < ncross; ++x) f->g_host[xpos[x]] = h2;

  // ---- device tables -------------------------------------------------------------------------------------
  std::vector<double> Qp(size_t(n1p) * n1p, 0.0);
  for (int j = 0; j < n1; ++j)
    for (int m = 0; m < n1; ++m) Qp[size_t(j) * n1p + m] = double(Q[size_t(j) * n1 + m]);
  double *d_Qp = nullptr, *d_rho = nullptr, *d_A0 = nullptr, *d_H0 = nullptr;
  ROM_TRY(upload(&d_Qp, Qp));
  ROM_TRY(upload(&d_rho, rho));
  const size_t hrows = size_t(n1) * n1;
  ROM_HIP(hipMalloc(&d_A0, std::max<size_t>(hrows * n1p, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&d_H0, std::max<size_t>(hrows * n1p, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&f->d_Tm, size_t(16) * n1p * n1p * sizeof(double)));
  {
    size_t total = hrows * n1p;
    k_build_A0<<<unsigned((total + 255) / 256), 256, 0, ctx->stream>>>(d_A0, d_Qp, d_rho, n1, n1p, N);
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_gemm_nt(ctx, int64_t(hrows), n1p, n1p, 1.0, d_A0, n1p, d_Qp, n1p, 0.0, d_H0, n1p,
                               "setup_gemm_H0"));
    size_t tt = size_t(16) * n1p * n1p;
    k_build_Tm<<<unsigned((tt + 255) / 256), 256, 0, ctx->stream>>>(f->d_Tm, d_H0, n1, n1p, N);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  hipFree(d_rho);
  hipFree(d_H0);  // only needed for the Dirichlet-to-Neumann tables; the extension runs in the sine basis (A0)
  f->d_A0 = d_A0;
  {
    // kmax[d]: modes with rho_mode(d) >= 1e-24, rounded up to the K chunk
    std::vector<int> kmax(N + 1, n1p);
    for (int dd = 1; dd <= N; ++dd) {
      int last = -1;
      for (int m = 0; m < n1; ++m)
        if (rho[size_t(m) * (N + 1) + std::min(dd, N)] >= 1e-24) last = m;
      kmax[dd] = std::min(n1p, std::max(BK, (last + 1 + BK - 1) / BK * BK));
    }
    kmax[0] = n1p;
    ROM_TRY(upload(&f->d_kmax, kmax));
    std::vector<int> eposv(std::max(E, 1), 0);
    for (int e = 0; e < E; ++e) eposv[e] = ppos[e];
    ROM_TRY(upload(&f->d_epos, eposv));
    f->n_edges = E;
    // flops of the truncated extension, per system (for the work accounting)
    double fl = 0;
    const int npj = (n1 + 15) / 16, npi = (n1 + 3) / 4;
    for (int b = 0; b < nrb * ncb; ++b)
      for (int pi = 0; pi < npi; ++pi)
        for (int pj = 0; pj < npj; ++pj) {
          int i0 = 4 * pi + 1, j0 = 16 * pj + 1, i1 = std::min(i0 + 3, n1), j1 = std::min(j0 + 15, n1);
          int dist[4] = {i0, N - i1, j0, N - j1};
          for (int sdx = 0; sdx < 4; ++sdx)
            if (bside[b][sdx] >= 0) fl += 2.0 * 64 * kmax[dist[sdx]];
        }
    f->ext_flops = fl + 2.0 * E * double(n1p) * n1p;
  }
  ROM_TRY(upload(&f->d_W, W));
  ROM_TRY(upload(&f->d_g, f->g_host));

  // ---- closed-form elimination tables ------------------------------------------------------------------------
  // K = tridiag(-1/2, 2, -1/2) - T_same = Q diag(1 + lam_m/2 - rho_m(1)) Q   =>   K^-1 = Q diag(1/kappa_m) Q
  std::vector<RhsTerm> rhs_terms;
  std::vector<PreEdge> pre_edges(npre);
  double pre_flops = 0;
  if (npre > 0) {
    std::vector<double> Kinv(size_t(n1p) * n1p, 0.0);
    {
      std::vector<ld> QD(size_t(n1) * n1);
      for (int j = 0; j < n1; ++j)
        for (int m = 0; m < n1; ++m) {
          ld kappa = 1.0L + lam[m] / 2.0L - (ld)rho[size_t(m) * (N + 1) + 1];
          QD[size_t(j) * n1 + m] = Q[size_t(j) * n1 + m] / kappa;
        }
      for (int i = 0; i < n1; ++i)
        for (int j = 0; j <= i; ++j) {
          ld sacc = 0;
          for (int m = 0; m < n1; ++m) sacc += QD[size_t(i) * n1 + m] * Q[size_t(j) * n1 + m];
          Kinv[size_t(i) * n1p + j] = Kinv[size_t(j) * n1p + i] = double(sacc);
        }
    }
    double* d_Kinv = nullptr;
    ROM_TRY(upload(&d_Kinv, Kinv));
    const size_t tsz = size_t(n1p) * n1p;
    // per (pre edge, neighbour): X_fe, B_fe = X_fe K^-1 ; tables kept: B^T (back substitution), R (tiles)
    int nbt = 0;
    for (int i = 0; i < npre; ++i) nbt += int(nbs[i].size());
    const int nR = int(rid.size());
    ROM_HIP(hipMalloc(&f->d_R, std::max<size_t>(size_t(nR + nbt) * tsz, 1) * sizeof(double)));
    int nvec = 0;
    for (int i = 0; i < npre; ++i) nvec += 1 + int(nbs[i].size());
    ROM_HIP(hipMalloc(&f->d_vec, std::max<size_t>(size_t(nvec) * n1p, 1) * sizeof(double)));
    ROM_HIP(hipMemsetAsync(f->d_vec, 0, size_t(nvec) * n1p * sizeof(double), ctx->stream));
    double *d_X = nullptr, *d_B = nullptr;
    int maxnb = 0;
    for (int i = 0; i < npre; ++i) maxnb = std::max(maxnb, int(nbs[i].size()));
    ROM_HIP(hipMalloc(&d_X, size_t(maxnb) * tsz * sizeof(double)));
    ROM_HIP(hipMalloc(&d_B, size_t(maxnb) * tsz * sizeof(double)));
    int* d_xrc = nullptr;
    ROM_HIP(hipMalloc(&d_xrc, 64 * sizeof(int)));
    int bt_next = nR, vec_next = 0;
    for (int i = 0; i < npre; ++i) {
      const int e = pre_list[i];
      const Edge& pe = edges[e];
      PreEdge& P = pre_edges[i];
      memset(&P, 0, sizeof(P));
      P.pos = ppos[e];
      P.e0 = pe.b0;
      P.e1 = pe.b1;
      P.nnb = int(nbs[i].size());
      // w_e = K^-1 g_e
      P.woff = (vec_next++) * n1p;
      ROM_TRY(rom_launch_gemm_nt(ctx, n1p, 1, n1p, 1.0, d_Kinv, n1p, f->d_g + ppos[e], n1p, 0.0, f->d_vec + P.woff, 1,
                                 "setup_gemm"));
      for (int q = 0; q < P.nnb; ++q) {
        const Nb& nb = nbs[i][q];
        int tmat = -1;
        if (nb.blk >= 0) tmat = side_of(nb.blk, nb.f) * 4 + side_of(nb.blk, e);
        std::vector<int> xrc;
        for (auto& xr : nb.xs) xrc.push_back(xr.first);
        for (auto& xr : nb.xs) xrc.push_back(xr.second);
        if (xrc.size() > 64) { rom_set_error("internal: too many cross points on one edge"); return ROM_ERR_INVALID; }
        if (!xrc.empty()) ROM_HIP(hipMemcpy(d_xrc, xrc.data(), xrc.size() * sizeof(int), hipMemcpyHostToDevice));
        double* Xq = d_X + size_t(q) * tsz;
        double* Bq = d_B + size_t(q) * tsz;
        k_build_X<<<unsigned((tsz + 255) / 256), 256, 0, ctx->stream>>>(Xq, f->d_Tm, n1, n1p, tmat, int(nb.xs.size()),
                                                                      d_xrc, d_xrc + nb.xs.size());
        ROM_HIP(hipGetLastError());
        ROM_HIP(hipStreamSynchronize(ctx->stream));  // d_xrc is reused
        // B_fe = X_fe K^-1 (K^-1 symmetric)
        ROM_TRY(rom_launch_gemm_nt(ctx, n1p, n1p, n1p, 1.0, Xq, n1p, d_Kinv, n1p, 0.0, Bq, n1p, "setup_gemm"));
        // B^T table for the back substitution: (K^-1 X_fe^T)[node of e][position on f]
        const int bt = bt_next++;
        ROM_TRY(rom_launch_gemm_nt(ctx, n1p, n1p, n1p, 1.0, d_Kinv, n1p, Xq, n1p, 0.0, f->d_R + size_t(bt) * tsz, n1p,
                                   "setup_gemm"));
        P.nb[q] = PreNb{ppos[nb.f], nb.blk, used[nb.f], bt};
        // q_{e,f} = B_fe g_e : rhs correction of the active unknowns on f
        const int qoff = (vec_next++) * n1p;
        ROM_TRY(rom_launch_gemm_nt(ctx, n1p, 1, n1p, 1.0, Bq, n1p, f->d_g + ppos[e], n1p, 0.0, f->d_vec + qoff, 1,
                                   "setup_gemm"));
        for (int t : nb_tiles(nb)) {
          const int lr0 = tile_loc[t] * TB;
          rhs_terms.push_back(RhsTerm{t, lr0, std::max(0, std::min(TB, n1 - lr0)), tile_ndr[t], qoff, nb.blk, pe.b0, pe.b1});
        }
        pre_flops += 2.0 * n1p * double(n1p);  // back substitution GEMM share of this neighbour
      }
      // R tables of this edge
      for (auto& kv : rid) {
        if (kv.first[0] != i) continue;
        const int qr = nb_index(i, kv.first[1]), qc = nb_index(i, kv.first[2]);
        ROM_TRY(rom_launch_gemm_nt(ctx, n1p, n1p, n1p, 1.0, d_B + size_t(qr) * tsz, n1p, d_X + size_t(qc) * tsz, n1p, 0.0,
                                   f->d_R + size_t(kv.second) * tsz, n1p, "setup_gemm"));
      }
      ROM_HIP(hipStreamSynchronize(ctx->stream));  // d_X / d_B are reused by the next edge
    }
    hipFree(d_X);
    hipFree(d_B);
    hipFree(d_xrc);
    hipFree(d_Kinv);
  }
  f->d_Qp = d_Qp;
  f->nrhs = int(rhs_terms.size());
  ROM_TRY(upload(&f->d_rhs, rhs_terms));
  ROM_TRY(upload(&f->d_pre, pre_edges));
  ROM_TRY(upload(&f->d_desc, f->desc));
  ROM_TRY(upload(&f->d_extra, extras));
  ROM_TRY(upload(&f->d_slot_of, f->slot_of));
  ROM_TRY(upload(&f->d_kptr, f->kptr));
  ROM_TRY(upload(&f->d_kpair, f->kpair));
  ROM_TRY(upload(&f->d_colptr, f->colptr));
  ROM_TRY(upload(&f->d_colrow, f->colrow));
  ROM_TRY(upload(&f->d_colti, f->colti));
  ROM_TRY(upload(&f->d_sides, f->sides));
  ROM_TRY(upload(&f->d_vmap, vmap));

  // ---- work accounting of this algorithm, per snapshot solve ------------------------------------------------
  const double ext_flops = f->ext_flops;
  double back_flops = 2.0 * 4096.0 * (f->nslots + T);
  f->flops_solve = flops + ext_flops + back_flops + pre_flops;
  // HBM bytes: factor tiles written once + read once by the back substitution, inverse tiles w+r,
  // the snapshot row written once, the coefficients read.
  f->bytes_solve = 8.0 * (2.0 * 4096.0 * f->nslots + 2.0 * 4096.0 * T + double(f->dim) + nrb * ncb);
  *out = f;
  return ROM_OK;
}

extern "C" int rom_fem_dims(rom_fem* f, int* nr, int* nc, int64_t* dim, int* n_interface, int* n_tiles) {
  ROM_CHECK(f, "null fem");
  if (nr) *nr = f->nr;
  if (nc) *nc = f->nc;
  if (dim) *dim = f->dim;
  if (n_interface) *n_interface = f->nG;
  if (n_tiles) *n_tiles = f->nslots;
  return ROM_OK;
}

extern "C" int rom_fem_load_vector_host(rom_fem* f, double* B) {
  ROM_CHECK(f && B, "bad arguments");
  // (:177-185) every inner vertex collects area/6 + area/3 + area/3 + area/6 in this order
  double area = (1.0 / f->N) * (1.0 / f->N);
  double v = 0.0;
  v += area / 6;
  v += area / 3;
  v += area / 3;
  v += area / 6;
  for (int64_t i = 0; i < f->dim; ++i) B[i] = v;
  return ROM_OK;
}

extern "C" int rom_solve_work(rom_fem* f, double* flops_own, double* bytes_own, double* flops_banded,
                              double* bytes_banded) {
  ROM_CHECK(f, "null fem");
  if (flops_own) *flops_own = f->flops_solve;
  if (bytes_own) *bytes_own = f->bytes_solve;
  double b = std::min(f->nr, f->nc), dim = double(f->dim);
  double nnzL = dim * (b + 1) - b * (b + 1) / 2;
  if (bytes_banded) *bytes_banded = 8.0 * (3 * nnzL + 5 * dim);
  if (flops_banded) *flops_banded = dim * b * b + 4 * dim * b;
  return ROM_OK;
}

extern "C" int rom_assemble_batch(rom_fem* f, rom_buf* a, int M, rom_buf* diag, rom_buf* east, rom_buf* north) {
  ROM_CHECK(f && a && diag && east && north, "rom_assemble_batch: null argument");
  ROM_CHECK(M >= 0, "rom_assemble_batch: negative M");
  const int kblk = f->nrb * f->ncb;
  ROM_CHECK(a->n >= size_t(M) * kblk, "rom_assemble_batch: `a` holds %zu doubles, need %zu", a->n, size_t(M) * kblk);
  ROM_CHECK(diag->n >= size_t(M) * f->dim && east->n >= size_t(M) * f->nr * (f->nc - 1) &&
                north->n >= size_t(M) * (f->nr - 1) * f->nc,
            "rom_assemble_batch: output buffers too small");
  if (M == 0) return ROM_OK;
  FemDev d = make_dev(f);
  dim3 grid(unsigned((f->dim + 255) / 256), M);
  {
    ROM_PROF(f->ctx, "assemble_stencil", 7.0 * f->dim * M, 24.0 * f->dim * M);
    k_assemble_stencil<<<grid, 256, 0, f->ctx->stream>>>(d, a->p, M, diag->p, east->p, north->p);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

static int ensure_workspace(rom_fem* f, int Mc) {
  if (f->ws_M >= Mc) return ROM_OK;
  ROM_HIP(hipStreamSynchronize(f->ctx->stream));
  if (f->d_L) hipFree(f->d_L);
  if (f->d_invL) hipFree(f->d_invL);
  if (f->d_y) hipFree(f->d_y);
  if (f->d_yhat) hipFree(f->d_yhat);
  f->d_L = f->d_invL = f->d_y = f->d_yhat = nullptr;
  f->ws_M = 0;
  ROM_HIP(hipMalloc(&f->d_L, std::max<size_t>(size_t(Mc) * f->nslots * 4096, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&f->d_invL, std::max<size_t>(size_t(Mc) * f->T * 4096, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&f->d_y, std::max<size_t>(size_t(Mc) * f->nGp, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&f->d_yhat, std::max<size_t>(size_t(Mc) * f->nGp, 1) * sizeof(double)));
  f->ws_M = Mc;
  return ROM_OK;
}

// enqueue every kernel of one sub-batch (Mc systems, workspace pointers already offset) on `st`
static int enqueue_solve(rom_fem* f, const FemDev& d, const double* am, int Mc, double* U, long long row,
                         hipStream_t st, size_t lds_back) {
  rom_ctx* ctx = f->ctx;
  const int kblk = f->nrb * f->ncb;
  static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-column kernel names
  char nm[4][48];
  if (f->nGp > 0) {
    {
      ROM_PROF(ctx, "init_rhs", 0, 8.0 * Mc * f->nGp);
      size_t tot = size_t(Mc) * f->nGp;
      k_init_rhs<<<unsigned((tot + 255) / 256), 256, 0, st>>>(d, Mc);
      if (f->nrhs > 0) k_rhs_pre<<<dim3(f->T, Mc), 64, 0, st>>>(d, am, f->T);
    }
    for (int j = 0; j < f->T; ++j) {
      {
        int slot = f->diag_slot[j];
        double nk = f->kptr[slot + 1] - f->kptr[slot];
        const char* base[4] = {"diag_update", "diag_potrf", "diag_inverse", "factor_panel"};
        for (int q = 0; q < 4; ++q) detail ? snprintf(nm[q], 48, "%s_j%02d", base[q], j) : snprintf(nm[q], 48, "%s", base[q]);
        {
          ROM_PROF(ctx, nm[0], Mc * nk * 2.0 * 262144, Mc * 8.0 * 4096 * (1 + 2 * nk));
          k_diag_update<<<Mc, 256, 0, st>>>(d, am, slot);
        }
        {
          ROM_PROF(ctx, nm[1], Mc * (262144 / 3.0), Mc * 8.0 * 4096 * 2);
          k_diag_potrf<<<Mc, 64, 0, st>>>(d, slot);
        }
        {
          ROM_PROF(ctx, nm[2], Mc * (262144 / 3.0 + 4096.0), Mc * 8.0 * 4096 * 2);
          k_diag_inverse<<<Mc, 64, 0, st>>>(d, slot, j);
        }
      }
      int nrows = f->colptr[j + 1] - f->colptr[j];
      if (nrows > 0) {
        double nk = 0;
        for (int e = f->colptr[j]; e < f->colptr[j + 1]; ++e) nk += f->kptr[f->colrow[e] + 1] - f->kptr[f->colrow[e]];
        ROM_PROF(ctx, nm[3], Mc * (nk + nrows) * 2.0 * 262144, Mc * 8.0 * 4096 * (2 * nk + 2 * nrows));
        k_factor_panel<<<nrows * Mc, 256, 0, st>>>(d, am, j, Mc);
      }
    }
    if (f->T > 0) {
      ROM_PROF(ctx, "backsolve", Mc * 2.0 * 4096 * (f->nslots + f->T), Mc * 8.0 * 4096 * (f->nslots + f->T));
      k_backsolve<<<Mc, 256, lds_back, st>>>(d);
    }
    if (f->npre > 0) {
      ROM_PROF(ctx, "back_pre", Mc * 2.0 * f->n1p * double(f->n1p) * 3.0 * f->npre, 8.0 * Mc * f->n1p * 4.0 * f->npre);
      k_back_pre<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->npre), 256, 0, st>>>(d, am, Mc);
    }
  }
  {
    const int nij = f->n1 * f->n1;
    if (nij > 0) {
      if (f->n_edges > 0) {
        ROM_PROF(ctx, "edge_transform", Mc * 2.0 * f->n_edges * double(f->n1p) * f->n1p, 16.0 * Mc * f->n_edges * f->n1p);
        k_edge_transform<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->n_edges), 256, 0, st>>>(d, Mc);
      }
      const int npatch = ((f->n1 + 3) / 4) * ((f->n1 + 15) / 16);
      dim3 grid(npatch, (Mc + 63) / 64, kblk);
      ROM_PROF(ctx, "extend", (f->ext_flops - 2.0 * f->n_edges * double(f->n1p) * f->n1p) * Mc, 8.0 * Mc * double(kblk) * nij);
      k_extend<<<grid, 256, 0, st>>>(d, am, Mc, U, row);
    }
    if (f->nGp > 0) {
      ROM_PROF(ctx, "scatter_interface", 0, 16.0 * Mc * f->nG);
      k_scatter_interface<<<dim3((f->nGp + 255) / 256, Mc), 256, 0, st>>>(d, Mc, U, row);
    }
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_solve_batch(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0) {
  ROM_CHECK(f && a && U, "rom_solve_batch: null argument");
  ROM_CHECK(M >= 0 && row0 >= 0, "rom_solve_batch: negative M or row offset");
  const int kblk = f->nrb * f->ncb;
  ROM_CHECK(a->n >= size_t(M) * kblk, "rom_solve_batch: `a` holds %zu doubles, need %zu", a->n, size_t(M) * kblk);
  ROM_CHECK(U->n >= size_t(row0 + M) * f->dim, "rom_solve_batch: U holds %zu doubles, need %zu", U->n,
            size_t(row0 + M) * f->dim);
  if (M == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  ROM_HIP(hipSetDevice(ctx->device));
  // chunk the sweep so that the factor workspace respects the budget
  size_t per_sys = (size_t(f->nslots) * 4096 + size_t(f->T) * 4096 + 2 * size_t(f->nGp)) * sizeof(double);
  int Mc_max = int(std::max<size_t>(1, std::min<size_t>(size_t(M), ctx->ws_limit / std::max<size_t>(per_sys, 1))));
  if (f->ws_M > 0 && f->ws_M < Mc_max && f->ws_M >= 256) Mc_max = f->ws_M;  // reuse what we have
  ROM_TRY(ensure_workspace(f, Mc_max));
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  const size_t lds_back = size_t(std::max(f->nGa, 1)) * sizeof(double);
  ROM_CHECK(lds_back <= 60 * 1024, "rom_solve_batch: interface too large for the LDS-resident back substitution");
  // Sub-batches run on separate HIP streams: the wave-per-system diagonal kernels are latency bound
  // (one wave per SIMD), the MFMA kernels of another sub-batch fill the chip meanwhile.
  const int nsub = std::max(1, std::min(ctx->n_streams, (M + 255) / 256));
  ROM_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  for (int s = 1; s < nsub; ++s) ROM_HIP(hipStreamWaitEvent(ctx->aux[s - 1], ctx->ev_fork, 0));
  for (int m0 = 0; m0 < M; m0 += Mc_max) {
    const int Mchunk = std::min(Mc_max, M - m0);
    const int per = ((Mchunk + nsub - 1) / nsub + 63) / 64 * 64;
    for (int s = 0; s < nsub; ++s) {
      const int off = s * per;
      if (off >= Mchunk) break;
      const int Mc = std::min(per, Mchunk - off);
      hipStream_t st = s == 0 ? ctx->stream : ctx->aux[s - 1];
      ctx->prof_stream = st;
      FemDev d = make_dev(f);
      d.L += size_t(off) * f->nslots * 4096;
      d.invL += size_t(off) * f->T * 4096;
      d.y += size_t(off) * f->nGp;
      d.yhat += size_t(off) * f->nGp;
      ROM_TRY(enqueue_solve(f, d, a->p + size_t(m0 + off) * kblk, Mc, U->p, (long long)(row0 + m0 + off), st, lds_back));
    }
    ctx->prof_stream = nullptr;
    if (m0 + Mc_max < M) {  // the workspace is reused by the next chunk: join first
      for (int s = 1; s < nsub; ++s) {
        ROM_HIP(hipEventRecord(ctx->ev_join[s - 1], ctx->aux[s - 1]));
        ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[s - 1], 0));
      }
      ROM_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
      for (int s = 1; s < nsub; ++s) ROM_HIP(hipStreamWaitEvent(ctx->aux[s - 1], ctx->ev_fork, 0));
    }
  }
  for (int s = 1; s < nsub; ++s) {
    ROM_HIP(hipEventRecord(ctx->ev_join[s - 1], ctx->aux[s - 1]));
    ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[s - 1], 0));
  }
  hipStream_t st = ctx->stream;
  int status = 0;
  ROM_HIP(hipMemcpyAsync(&status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
  ROM_HIP(hipStreamSynchronize(st));
  if (status != 0) {
    rom_set_error("rom_solve_batch: interface matrix not positive definite (non-positive pivot); "
                  "all block coefficients must be > 0");
    return ROM_ERR_NOT_SPD;
  }
  return ROM_OK;
}
