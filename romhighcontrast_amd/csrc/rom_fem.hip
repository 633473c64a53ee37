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
// up to the side pairing (16 tables T[sr][sc]).  Two further reductions are parameter independent
// up to scalar weights and therefore tabulated once per FE space:
//   (1) an independent set of edges (no two on the same block) has the self block (a_p + a_q) K with K
//       fixed, so it is eliminated in closed form (tables X K^-1 X^T);
//   (2) on every remaining ("active") edge f all couplings to the rest of the interface act through a
//       numerically low-rank range W_f (the smooth traces of the neighbouring sides + the end nodes
//       that touch cross points), so u_f = P_f z_f + p0_f / s_f with z_f = W_f^T u_f of dimension
//       rank(W_f) ~ 30 at N = 128, and the system that is actually factorised couples only the z_f
//       and the cross points.
// Per parameter the work is: assemble the reduced system tile by tile (never stored: each tile is
// built in registers when it is factored), a left-looking 64x64 tile Cholesky on MFMA with the
// forward substitution fused in, a backward substitution, the expansion z -> edge values, the back
// substitution of the closed-form edges, and the harmonic extension
//        u_I,b = (h^2/a_b) W + sum_sides H_s u_G|side
// as one batched MFMA GEMM that writes the snapshot rows straight into the caller's (M, dim) matrix.
//
// Setup tables come from the sine (DST-I) eigenbasis of the block:  H_0[(i,j),k] =
// sum_m Q[j,m] rho_m(i) Q[k,m], rho_m(i) = sinh((N-i) phi_m)/sinh(N phi_m), cosh phi_m = 2-cos(pi m/N);
// the other three sides are row permutations of H_0.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include "rom_hostla.h"
#include "rom_mma.h"

// ============================================================================================
// device-side view of a rom_fem
// ============================================================================================
struct FemDev {
  int nrb, ncb, N, n1, n1p, nr, nc, nGp, nGa, T, nslots, kblk, npre, nrhs, nexp, ncross, xb0;
  long long dim;
  const double* pool;  // 64x64 tables of the tile terms
  const GenTerm* terms;
  const double* Bt;    // back substitution tables of the closed-form edges
  const double* P;     // expansion tables of the active edges
  const double* vec;
  const RhsTerm* rhs;
  const PreEdge* pre;
  const ExpEdge* exp;
  const int* xred;
  const int* scb;         // scalar block: (b0, b1) per entry
  int spos0, nsc, sblk0;  // its position / length in the interface vector, position of the h^2/a_b part
  const RowEnt* rowent;
  int nrowent;
  const DenseGroup* dgroups;  // single-tile path: coefficient blocks of the closed-form edges as one dense product
  const int* dweight;
  const int* ditem_group;
  const int* ditem_k;
  const double* dmat;
  int ndg, ndi;
  const CoefGroup* groups;
  const double* cm;
  const int* item_group;
  const int* item_k;
  int ncoef;
  const double* G;     // extension tables of the compressed edges: per table (n1*n1) x (rank+1 padded)
  const double* A0;    // (n1*n1) x n1p : Q[j,mode] rho_mode(i), harmonic extension in the sine basis
  const double* Qp;    // n1p x n1p sine matrix (zero padded)
  const int* kmax;     // [N+1] modes (multiple of 16) that matter at distance d from a side
  const int* epos;     // nodal n1p block of every edge whose sine coefficients are needed
  double* yhat;        // [Mc][nGp] sine coefficients of the interface values
  const double* W;
  const double* g;
  const TileDesc* desc;
  const int* kptr;
  const int* kpair;
  const int* colptr;
  const int* colrow;
  const int* colti;
  const BlockSide* sides;
  const int* lr_blocks;   // blocks whose sides are all compressed: extended in mesh-row tiles
  const int* gen_blocks;  // all others: 4 x 16 patches
  const int* vmap;
  const int* scat;  // interface positions copied to the snapshot rows by k_scatter_interface
  int nscat;
  double* L;     // [Mc][nslots][64*64]
  double* invL;  // [Mc][T][64*64]
  double* y;     // [Mc][nGp]: reduced unknowns | nodal edge blocks | cross block
  int* status;
};

static FemDev make_dev(const rom_fem* f) {
  FemDev d;
  d.nrb = f->nrb; d.ncb = f->ncb; d.N = f->N; d.n1 = f->n1; d.n1p = f->n1p; d.nr = f->nr; d.nc = f->nc;
  d.nGp = f->nGp; d.nGa = f->nGa; d.npre = f->npre; d.nrhs = f->nrhs; d.nexp = f->nexp; d.ncross = f->ncross;
  d.xb0 = f->xb0; d.pool = f->d_pool; d.terms = f->d_terms; d.Bt = f->d_Bt; d.P = f->d_P; d.vec = f->d_vec;
  d.rhs = f->d_rhs; d.pre = f->d_pre; d.exp = f->d_exp; d.xred = f->d_xred; d.scb = f->d_scb; d.spos0 = f->spos0; d.nsc = f->nsc;
  d.sblk0 = f->spos0 + f->n_all_edges; d.groups = f->d_groups; d.cm = f->d_cm; d.item_group = f->d_item_group;
  d.item_k = f->d_item_k; d.ncoef = f->ncoef; d.rowent = f->d_rowent; d.nrowent = f->nrowent; d.dgroups = f->d_dgroups; d.dweight = f->d_dweight;
  d.ditem_group = f->d_ditem_group; d.ditem_k = f->d_ditem_k; d.dmat = f->d_dmat; d.ndg = f->ndg; d.ndi = f->ndi; d.T = f->T; d.nslots = f->nslots;
  d.kblk = f->nrb * f->ncb; d.dim = f->dim;
  d.G = f->d_G; d.A0 = f->d_A0; d.Qp = f->d_Qp; d.kmax = f->d_kmax; d.epos = f->d_epos; d.yhat = f->d_yhat; d.W = f->d_W;
  d.g = f->d_g; d.desc = f->d_desc;
  d.kptr = f->d_kptr; d.kpair = f->d_kpair; d.colptr = f->d_colptr; d.colrow = f->d_colrow;
  d.colti = f->d_colti; d.sides = f->d_sides; d.lr_blocks = f->d_lr_blocks; d.gen_blocks = f->d_gen_blocks; d.vmap = f->d_vmap; d.scat = f->d_scat; d.nscat = f->nscat; d.L = f->d_L; d.invL = f->d_invL;
  d.y = f->d_y; d.status = f->ctx->d_status;
  return d;
}

// ============================================================================================
// setup kernel
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

// ============================================================================================
// reduced-system tile assembly
// ============================================================================================
// weight of one table, from the block coefficients of this system
__device__ inline double term_coef(const GenTerm& g, const double* __restrict__ am) {
  switch (g.kind) {
    case 0: return -am[g.b[0]];                                             // Schur coupling through block b0
    case 1: return am[g.b[0]] + am[g.b[1]];                                 // edge self block s_f K~
    case 2: return -(am[g.b[0]] + am[g.b[1]]) / 2;                          // cross point <-> end node of an edge
    case 3: return ((am[g.b[0]] + am[g.b[1]]) + am[g.b[2]]) + am[g.b[3]];   // cross point diagonal
    case 4: return -(am[g.b[0]] * am[g.b[1]] / (am[g.b[2]] + am[g.b[3]]));  // closed-form edge: edge x edge
    case 5: return -(am[g.b[0]] / 2);                                       //                   edge x cross
    default: return -((am[g.b[2]] + am[g.b[3]]) / 4);                       //                   cross x cross
  }
}

// This thread's share of the assembled tile: row (t >> 2), 16 consecutive columns starting at
// (t & 3) * 16.  Every table is read with 16-byte loads, 128 contiguous bytes per thread and table; the
// values wait in registers while the MFMA k-loop runs.
struct STile {
  double v[16];
};

constexpr int COEF_MAX = 64;   // term weights cached in LDS per pass
constexpr int DENSE_GROUPS_MAX = 8;  // closed-form edges whose coefficient blocks k_solve1 builds
constexpr int ROW_BATCH = 64;  // loads in flight per wave in the single-tile assembly

__device__ inline void s_tile_load(STile& st, const TileDesc& d, const FemDev& f, const double* __restrict__ am,
                                   double* coef) {
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
#pragma unroll
  for (int x = 0; x < 16; ++x) st.v[x] = 0.0;
  for (int tb = d.t0; tb < d.t1; tb += COEF_MAX) {
    const int nt = min(COEF_MAX, d.t1 - tb);
    __syncthreads();
    if (int(threadIdx.x) < nt) coef[threadIdx.x] = term_coef(f.terms[tb + threadIdx.x], am);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const GenTerm& g = f.terms[tb + t];
      if (r < g.r_lo || r >= g.r_hi || c0 >= g.c_hi || c0 + 16 <= g.c_lo) continue;
      const double cf = coef[t];
      const double2* src = reinterpret_cast<const double2*>(f.pool + size_t(g.tab) * 4096 + r * 64 + c0);
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const double2 w = src[x];  // tables are zero outside their rectangle: no masks needed
        st.v[2 * x] += cf * w.x;
        st.v[2 * x + 1] += cf * w.y;
      }
    }
  }
  if (d.diag && r >= d.ndr) {
#pragma unroll
    for (int x = 0; x < 16; ++x) st.v[x] = (c0 + x == r) ? 1.0 : 0.0;  // padding unknowns: identity
  }
}

// C(LDS tile) = S_tile - acc
__device__ inline void tile_from_acc(double* Cb, const Acc& acc, const STile& st, const WavePos& wp) {
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
// rhs of the reduced system: the parameter-independent part plus the contributions of the closed-form
// edges.  One workgroup per system; the terms are applied one after another (they overlap).
__global__ __launch_bounds__(256) void k_rhs(FemDev f, const double* __restrict__ a) {
  const int m = blockIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double* y = f.y + size_t(m) * f.nGp;
  for (int v = threadIdx.x; v < f.nGa; v += blockDim.x) y[v] = f.g[v];
  for (int t = 0; t < f.nrhs; ++t) {
    __syncthreads();
    const RhsTerm& rt = f.rhs[t];
    const double coef = rt.kind == 0 ? am[rt.b0] / (am[rt.e0] + am[rt.e1]) : 0.5;
    for (int i = threadIdx.x; i < rt.len; i += blockDim.x) y[rt.pos + i] += coef * f.vec[rt.voff + i];
  }
}

// Coefficient blocks read by the extension and the expansion, and the nodal copy of the cross points.
//   active edge f:      [z_f, 1/s_f, 0...]
//   closed-form edge e: [c_e / s_e, 1/s_e, 0...],  s_e K u_e = g_e + W_e c_e,
//                       c_e = sum_u a_u (M_eu z_u + m_eu / s_u) + (s_e/2) sum_x W_e[node_x,:]^T u_x
// one workgroup per system, one thread per entry
__global__ __launch_bounds__(256) void k_coef(FemDev f, const double* __restrict__ a) {
  const int m = blockIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double* y = f.y + size_t(m) * f.nGp;
  for (int x = threadIdx.x; x < f.ncross; x += blockDim.x) y[f.xb0 + x] = y[f.xred[x]];
  for (int i = threadIdx.x; i < f.nsc; i += blockDim.x) {
    const int b0 = f.scb[2 * i], b1 = f.scb[2 * i + 1];
    y[f.spos0 + i] = b1 >= 0 ? 1.0 / (am[b0] + am[b1]) : (1.0 / (double(f.N) * double(f.N))) / am[b0];
  }
  for (int it = threadIdx.x; it < f.ncoef; it += blockDim.x) {
    const CoefGroup& cg = f.groups[f.item_group[it]];
    const int k = f.item_k[it];
    const double s = am[cg.b0] + am[cg.b1];
    double out = 0.0;
    if (k == cg.r) {
      out = 1.0 / s;
    } else if (k < cg.r) {
      if (cg.kind == 0) {
        out = y[cg.zpos + k];
      } else {
        double acc = 0.0;
        for (int t = 0; t < cg.nterm; ++t) {
          const CoefTerm& ct = cg.t[t];
          const double* Mt = f.cm + ct.moff + k;
          const double* src = y + ct.src;
          double dot = 0.0;
#pragma unroll 16
          for (int j = 0; j < ct.len; ++j) dot += Mt[size_t(j) * cg.r] * src[j];
          if (ct.voff >= 0) dot += f.vec[ct.voff + k] / (am[ct.u0] + am[ct.u1]);
          acc += (ct.blk >= 0 ? am[ct.blk] : s / 2) * dot;
        }
        out = acc / s;
      }
    }
    y[cg.cpos + k] = out;
  }
}

// 4 doubles from an address that is only 8-byte aligned (pointer may be null -> zeros)
__device__ inline void load4_any(const double* __restrict__ p, double v[4]) {
  if (p) {
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
  } else {
    v[0] = v[1] = v[2] = v[3] = 0.0;
  }
}

// Edge values of the active edges from the reduced solution, u_f = P_f z_f + p0_f / s_f, as one batched
// MFMA GEMM (tile rows = systems, tile cols = nodes of f, K = compressed index); closed-form edges kept in
// compressed form enter the same way with z = c_e / s_e, P = K^-1 W_e, p0 = K^-1 g_e.  The values go to the
// snapshot rows directly (and to the nodal blocks of the interface vector).   grid (n1p/64, ceil(Mc/64), nexp)
__global__ __launch_bounds__(256) void k_expand(FemDev f, const double* __restrict__ a, int Mc, double* __restrict__ U,
                                                long long row0) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  const WavePos wp;
  const ExpEdge ee = f.exp[blockIdx.z];
  const int srow = stage_row(), sseg = stage_seg();
  const int mA = blockIdx.y * 64 + srow;
  const double* pA = mA < Mc ? f.y + size_t(mA) * f.nGp + ee.zpos + sseg : nullptr;
  const double* pB = f.P + (size_t(ee.ptab) * f.n1p + blockIdx.x * 64 + srow) * f.n1p + sseg;
  Acc acc;
  acc_zero(acc);
  gemm_loop(
      ee.nch, [&](int ch, double* v) { load4_any(pA ? pA + ch * BK : nullptr, v); },
      [&](int ch, double* v) { load4_aligned(pB + ch * BK, v); }, acc, lds, wp);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + acc_row(wp, i, g);
      if (m >= Mc) continue;
      const double inv = f.y[size_t(m) * f.nGp + ee.spos];  // 1 / (a_b0 + a_b1), from the scalar block
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int node = blockIdx.x * 64 + acc_col(wp, jb);
        const double v = node < f.n1 ? acc.c[i][jb][g] + f.vec[ee.p0off + node] * inv : 0.0;
        f.y[size_t(m) * f.nGp + ee.npos + node] = v;  // (read again by the node-by-node paths, if any)
        if (node < f.n1) U[(row0 + m) * f.dim + f.vmap[ee.npos + node]] = v;
      }
    }
}

// Back substitution of the closed-form edges: x_e = (w_e + sum_u B_ue^T (c_u . x_u)) / s_e as one batched
// MFMA GEMM: tile rows = systems, tile cols = nodes of e, K = positions in the neighbours' nodal blocks.
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
  for (int q = 0; q < pe.nnb; ++q) {
    const PreNb nb = pe.nb[q];
    const double* pA = am ? f.y + size_t(mA) * f.nGp + nb.fpos + sseg : nullptr;
    const double* pB = f.Bt + (size_t(nb.bt) * f.n1p + iB) * f.n1p + sseg;
    const double cu = !am ? 0.0 : (nb.blk >= 0 ? am[nb.blk] : seA / 2);
    gemm_loop(
        nb.nch,
        [&](int ch, double* v) {
          load4_aligned(pA ? pA + ch * BK : nullptr, v);
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] *= cu;
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

// Diagonal tile j, step 1 of 3 (MFMA): C = S_jj - sum_k L_jk L_jk^T, written to the tile's L slot.
// Only the lower triangle is consumed by the factorisation: the wave owning the upper-right
// quadrant skips its MFMAs.
__global__ __launch_bounds__(256) void k_diag_update(FemDev f, const double* __restrict__ a, int slot) {
  // 36.9 KB: four workgroups per CU, i.e. all 1024 systems of a C2 step resident in one round
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  __shared__ int kp[2 * KP_MAX];
  __shared__ double coef[COEF_MAX];
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
  s_tile_load(st, d, f, am, coef);  // table reads fly under the MFMAs below
  Acc acc;
  acc_zero(acc);
  const bool lower = !(wp.wr == 0 && wp.wc == 1);
  accumulate_klist(f, slot, Lm, kp, [&](int) { return lower; }, acc, stA, stB, wp);
  tile_from_acc(Cb, acc, st, wp);
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

// Whole reduced solve of a system whose reduced matrix is ONE tile (e.g. 2x2 blocks at N = 128: 2 x 31
// compressed unknowns + the cross point), one wave per system, nothing but the solution leaves the CU:
//   assemble (lane c accumulates column c of the upper triangle = row c of the lower one, coalesced table reads) ->
//   in-register Cholesky with the forward substitution fused in -> L through LDS -> back substitution ->
//   coefficient blocks for the extension (what k_coef does on the general path).
__global__ __launch_bounds__(64) void k_solve1(FemDev f, const double* __restrict__ a) {
  __shared__ __align__(16) double Ls[64 * LDC];
  __shared__ __align__(16) double lv[2][64];
  __shared__ double zs[64];
  __shared__ double wz[DENSE_GROUPS_MAX * 64];
  const int m = blockIdx.x, lane = threadIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double* ym = f.y + size_t(m) * f.nGp;
  const TileDesc& d = f.desc[0];
  // Assembly, upper triangle only (row <= col), by the host-built row program: lane c owns column c, entries
  // are (table row segment, term) pairs sorted by tile row; ROW_BATCH independent coalesced loads are in flight
  // before the first is consumed (one wave per SIMD: nothing else hides the latency).
  for (int r = 0; r < 64; ++r) Ls[r * LDC + lane] = 0.0;
  const double mycoef = lane < d.t1 - d.t0 ? term_coef(f.terms[d.t0 + lane], am) : 0.0;  // lane t: weight of term t
  __syncthreads();
  {
    static_assert(ROW_BATCH == 64, "one row-program entry per lane and batch");
    double acc = 0.0;
    const int4* ents = reinterpret_cast<const int4*>(f.rowent);
    int4 mine = f.nrowent > 0 ? ents[lane] : int4{0, 0, 0, 0};  // lane i holds entry i of the batch
    for (int e0 = 0; e0 < f.nrowent; e0 += ROW_BATCH) {
      const int4 cur = mine;
      if (e0 + ROW_BATCH < f.nrowent) mine = ents[e0 + ROW_BATCH + lane];  // next batch's entries fly meanwhile
      double v[ROW_BATCH];
#pragma unroll
      for (int i = 0; i < ROW_BATCH; ++i) {
        const int off = __builtin_amdgcn_readlane(cur.x, i);
        const int c_lo = __builtin_amdgcn_readlane(cur.z, i), c_hi = __builtin_amdgcn_readlane(cur.w, i);
        v[i] = (lane >= c_lo && lane < c_hi) ? f.pool[off + lane] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < ROW_BATCH; ++i) {
        const int meta = __builtin_amdgcn_readlane(cur.y, i);  // r | term << 8 | last << 16
        acc += readlane_f64(mycoef, (meta >> 8) & 0xff) * v[i];  // (an LDS lookup here costs its full latency per entry)
        if (meta >> 16) {
          Ls[(meta & 0xff) * LDC + lane] = acc;
          acc = 0.0;
        }
      }
    }
  }
  __syncthreads();
  double arow[64];  // row `lane` of the symmetric tile = column `lane` of its upper triangle
#pragma unroll
  for (int c = 0; c < 64; ++c) arow[c] = c <= lane ? Ls[c * LDC + lane] : 0.0;
  if (lane >= d.ndr) {
#pragma unroll
    for (int c = 0; c < 64; ++c) arow[c] = (c == lane) ? 1.0 : 0.0;  // padding unknowns: identity
  }
  // rhs of the reduced system (k_rhs)
  double y = f.g[lane];
  for (int t0 = 0; t0 < f.nrhs; t0 += 8) {  // eight terms at a time: their vector loads are in flight together
    double rv[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      rv[x] = 0.0;
      if (t0 + x < f.nrhs) {
        const RhsTerm& rt = f.rhs[t0 + x];
        if (lane >= rt.pos && lane < rt.pos + rt.len) rv[x] = f.vec[rt.voff + lane - rt.pos];
      }
    }
#pragma unroll
    for (int x = 0; x < 8; ++x)
      if (t0 + x < f.nrhs) {
        const RhsTerm& rt = f.rhs[t0 + x];
        y += (rt.kind == 0 ? am[rt.b0] / (am[rt.e0] + am[rt.e1]) : 0.5) * rv[x];
      }
  }
  // Cholesky (as k_diag_potrf) with y carried along: after step jj, y holds L^-1 g in lanes <= jj
  bool bad = false;
  double myrs = 0.0;
#pragma unroll
  for (int jj = 0; jj < 64; ++jj) {
    const double dj = readlane_f64(arow[jj], jj);
    bad = bad || !(dj > 0.0);
    const double rs = rsqrt_newton(dj);
    const double l = arow[jj] * rs;  // L[lane][jj] for lane >= jj
    arow[jj] = l;
    const double yj = readlane_f64(y, jj) * rs;
    if (lane == jj) {
      y = yj;
      myrs = rs;  // (an LDS store here makes hipcc spill the whole tile)
    } else if (lane > jj) {
      y -= l * yj;
    }
    if (jj < 63) {
      double* bv = lv[jj & 1];
      bv[lane] = l;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 64; ++c)
        if (c > jj) arow[c] -= l * bv[c];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (bad && lane == 0) atomicOr(f.status, 1);
#pragma unroll
  for (int c = 0; c < 64; ++c) Ls[lane * LDC + c] = c <= lane ? arow[c] : 0.0;
  __syncthreads();
  // back substitution x = L^-T y.  Column `lane` of L is fetched from LDS in one batch (conflict free), then the
  // chain x_j = y_j / L_jj ; y_i -= L_ji x_j (i < j) runs on registers and readlane broadcasts only.
  double lcol[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) lcol[j] = Ls[j * LDC + lane];
#pragma unroll
  for (int j = 63; j >= 0; --j) {
    const double xj = readlane_f64(y, j) * readlane_f64(myrs, j);
    if (lane == j) y = xj;
    else if (lane < j) y -= lcol[j] * xj;
  }
  ym[lane] = y;
  zs[lane] = y;
  __syncthreads();
  // coefficient blocks + nodal copy of the cross points (what k_coef does on the general path).  The blocks of
  // the closed-form edges are one dense product here: out[it] = sum_j D[j][it] * (w_g(j) z_j), D = all their
  // matrices side by side (64 x items, coalesced in `it`), w_g(j) the weight of source j for group g.
  for (int x = lane; x < f.ncross; x += 64) ym[f.xb0 + x] = zs[f.xred[x]];
  for (int i = lane; i < f.nsc; i += 64) {
    const int b0 = f.scb[2 * i], b1 = f.scb[2 * i + 1];
    ym[f.spos0 + i] = b1 >= 0 ? 1.0 / (am[b0] + am[b1]) : (1.0 / (double(f.N) * double(f.N))) / am[b0];
  }
  for (int idx = lane; idx < f.ndg * 64; idx += 64) {
    const DenseGroup& dg = f.dgroups[idx >> 6];
    const int wd = f.dweight[idx];
    const double wgt = wd >= 0 ? am[wd] : (wd == -1 ? (am[dg.b0] + am[dg.b1]) / 2 : 0.0);
    wz[idx] = wgt * zs[idx & 63];
  }
  __syncthreads();
  for (int it = lane; it < f.ndi; it += 64) {
    const int g = f.ditem_group[it], k = f.ditem_k[it];
    const DenseGroup& dg = f.dgroups[g];
    const double* D = f.dmat + it;
    const double* wg = wz + g * 64;
    double acc = 0.0;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) acc += D[size_t(j) * f.ndi] * wg[j];
    for (int v = 0; v < dg.nv; ++v) acc += am[dg.vblk[v]] / (am[dg.vu0[v]] + am[dg.vu1[v]]) * f.vec[dg.voff[v] + k];
    ym[dg.cpos + k] = acc / (am[dg.b0] + am[dg.b1]);
  }
  for (int it = lane; it < f.ncoef; it += 64) {
    const CoefGroup& cg = f.groups[f.item_group[it]];
    const int k = f.item_k[it];
    if (cg.kind == 1 && k < cg.r) continue;  // done above
    ym[cg.cpos + k] = k == cg.r ? 1.0 / (am[cg.b0] + am[cg.b1]) : (k < cg.r ? zs[cg.zpos + k] : 0.0);
  }
}

// Sub-diagonal tiles of column j: C = S_ij - sum_k L_ik L_jk^T ; L_ij = C invL_jj^T ;
// y_i -= L_ij y_j.  invL_jj is lower triangular: the waves owning output columns 0..31 only need
// k < 32 of the second product.
__global__ __launch_bounds__(256) void k_factor_panel(FemDev f, const double* __restrict__ a, int j, int Mc) {
  __shared__ __align__(16) double lds[FACT_LDS_DOUBLES];
  __shared__ int kp[2 * KP_MAX];
  __shared__ double yj[64];
  __shared__ double coef[COEF_MAX];
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
  s_tile_load(st, d, f, am, coef);
  if (t < 64) yj[t] = f.y[size_t(m) * f.nGp + j * 64 + t];
  Acc acc;
  acc_zero(acc);
  accumulate_klist(f, slot, Lm, kp, [](int) { return true; }, acc, stA, stB, wp);
  tile_from_acc(Cb, acc, st, wp);

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

// value of the neighbouring lane (lane ^ 1), by DPP quad permutation
__device__ inline double lane_swap1(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// two doubles at an address that is only 8-byte aligned (snapshot rows have odd length)
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

// Harmonic extension, one batched MFMA GEMM over all blocks:
//   U_I,b[m,(i,j)] = (h^2/a_b) W[i,j] + sum_{sides s} sum_k c_s[m, k] * Tab_s[pi_s(i,j)][k]
// with one of two parameter-independent representations per side (ExtSide::mode):
//   1  sine modes:  c = yhat_s (sine coefficients of the edge values), Tab = A0[(i',j'), mode] = Q[j', mode] rho_mode(i').
//      rho_mode(i') ~ exp(-i' phi_mode): far from a side only the low modes survive in fp64; kmax[d] (multiple
//      of 16) is the number of modes with rho_mode(d) above 1e-18, so a tile whose vertices are at distance >= d
//      from side s stops its K loop there (terms below 1e-18 of the leading ones cannot change an fp64 sum).
//   2  compressed edge:  c = [z_f, 1/s_f] (the reduced unknowns themselves), Tab = H_0 [P_f, p0_f]: K = rank + 1,
//      no edge values or sine transform needed.
// Tile rows = systems; tile columns = 64 interior vertices of one block, either a 4 x 16 patch (pw_log2 = 4:
// the distance to all four sides is bounded below per tile, which is what the mode truncation needs) or
// 64 consecutive vertices of one mesh row (pw_log2 = 6, blocks whose sides are all compressed: K does not
// depend on the position, and 512 contiguous bytes per system are written -- measured 2.8 instead of
// 2.1 TB/s for the store stream alone, tools/hbm_write_bw.hip).
// grid (patches, ceil(Mc/64), blocks in the list)
__global__ __launch_bounds__(256) void k_extend(FemDev f, const double* __restrict__ a, int Mc,
                                                double* __restrict__ U, long long row0, const int* __restrict__ blocks,
                                                int pw_log2) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  __shared__ double scs[64];  // h^2 / a_b of the tile's systems
  double* stage = lds;
  const WavePos wp;
  const int b = blocks[blockIdx.z];
  const int p = b / f.ncb, q = b % f.ncb;
  const int n1 = f.n1, N = f.N;
  const BlockSide& sd = f.sides[b];
  const int srow = stage_row(), sseg = stage_seg();
  const int pw = 1 << pw_log2, ph = 64 >> pw_log2;    // patch width / height in vertices
  const int npj = (n1 + pw - 1) >> pw_log2;           // patches per patch row
  const int pi = blockIdx.x / npj, pj = blockIdx.x % npj;
  const int i0 = ph * pi + 1, j0 = pw * pj + 1;       // first vertex of the patch (1-based)
  const int i1 = min(i0 + ph - 1, n1), j1 = min(j0 + pw - 1, n1);

  const int mA = blockIdx.y * 64 + srow;
  const bool vA = mA < Mc;
  const int iB = i0 + (srow >> pw_log2), jB = j0 + (srow & (pw - 1));  // vertex of this thread's B row
  const bool vB = iB <= n1 && jB <= n1;

  // one merged, pipelined K loop over the chunks of all four sides
  int cend[4];
  const double* pAs[4];
  const double* pBs[4];
  int tot = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const ExtSide es = sd.s[s];
    int nch = 0;
    pAs[s] = pBs[s] = nullptr;
    if (es.mode != 0) {  // (mode 0: side on the domain boundary)
      const int hrow = vB ? h0_row(s, iB, jB, N, n1) : 0;
      if (es.mode == 1) {
        const int dist = s == 0 ? i0 : s == 1 ? N - i1 : s == 2 ? j0 : N - j1;  // closest vertex of the patch
        nch = f.kmax[dist] / BK;
        if (vA) pAs[s] = f.yhat + size_t(mA) * f.nGp + es.off + sseg;
        if (vB) pBs[s] = f.A0 + size_t(hrow) * f.n1p + sseg;
      } else {
        nch = es.nch;
        if (vA) pAs[s] = f.y + size_t(mA) * f.nGp + es.off + sseg;
        if (vB) pBs[s] = f.G + es.gtab + size_t(hrow) * (es.nch * BK) + sseg;
      }
    }
    tot += nch;
    cend[s] = tot;
  }
  if (threadIdx.x < 64) {
    const int m = blockIdx.y * 64 + threadIdx.x;
    scs[threadIdx.x] = m < Mc ? f.y[size_t(m) * f.nGp + f.sblk0 + b] : 0.0;  // h^2 / a_b, from the scalar block
  }
  __syncthreads();
  Acc acc;
  acc_zero(acc);
  auto pick = [&](int ch, const double* const* ps) -> const double* {
    const int s = (ch >= cend[0]) + (ch >= cend[1]) + (ch >= cend[2]);
    const int lc = ch - (s == 0 ? 0 : s == 1 ? cend[0] : s == 2 ? cend[1] : cend[2]);
    const double* p = s == 0 ? ps[0] : s == 1 ? ps[1] : s == 2 ? ps[2] : ps[3];
    return p ? p + lc * BK : nullptr;
  };
  gemm_loop(
      tot, [&](int ch, double* v) { load4_aligned(pick(ch, pAs), v); },
      [&](int ch, double* v) { load4_aligned(pick(ch, pBs), v); }, acc, stage, wp);

  // Epilogue.  The MFMA result layout gives a lane one vertex of each of its two 16-vertex column blocks; lane
  // pairs swap one value each (DPP) so that every lane owns two ADJACENT vertices of one block and the row is
  // written with 16-byte stores: the store stream of this kernel is issue bound, half the instructions matter.
  // (everything the epilogue needs from memory is fetched before the first store: a load after a store would
  // make its s_waitcnt vmcnt wait for the stores as well -- one counter, in order)
  const bool odd = wp.lane & 1;
  const int cfirst = acc_col(wp, odd ? 1 : 0) - (odd ? 1 : 0);  // first of this lane's two tile columns
  const int ii = i0 + (cfirst >> pw_log2) - 1, jj = j0 + (cfirst & (pw - 1)) - 1;  // 0-based interior indices
  const bool v0 = ii < n1 && jj < n1, v1 = ii < n1 && jj + 1 < n1;
  const long long gidx = (long long)(p * N + ii) * f.nc + (q * N + jj);
  double w_own[2];  // particular solution at this lane's own accumulator columns
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int cidx = acc_col(wp, jb);
    const int wi = i0 + (cidx >> pw_log2) - 1, wj = j0 + (cidx & (pw - 1)) - 1;
    w_own[jb] = (wi < n1 && wj < n1) ? f.W[wi * n1 + wj] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + acc_row(wp, i, g);
      const double sc = scs[acc_row(wp, i, g)];
      const double x0 = acc.c[i][0][g] + sc * w_own[0], x1 = acc.c[i][1][g] + sc * w_own[1];
      const double got = lane_swap1(odd ? x0 : x1);  // even lanes give away block 1, odd lanes block 0
      if (m >= Mc) continue;
      double* dst = U + (row0 + m) * f.dim + gidx;
      const double lo = odd ? got : x0, hi = odd ? x1 : got;
      if (v1) *reinterpret_cast<double2_u*>(dst) = double2_u{lo, hi};
      else if (v0) dst[0] = lo;
    }
}

// The same extension for blocks whose sides are all compressed, with 128 x 128 workgroup tiles (128 systems x
// one mesh row of up to 128 interior vertices; every wave a 64 x 64 quadrant = 4 x 4 MFMA accumulators): K is
// only sum(rank + 1) ~ 64, so a 64 x 64 tile spends most of its life in its prologue and epilogue; four times
// the outputs per workgroup amortise them and every LDS fragment feeds four MFMAs instead of two.
// grid (mesh rows x column tiles, ceil(Mc/128), lr blocks); LDS 74,752 B -> 2 workgroups per CU
constexpr int X128_STAGE = 128 * LDK;

__global__ __launch_bounds__(256, 2) void k_extend128(FemDev f, const double* __restrict__ a, int Mc,
                                                      double* __restrict__ U, long long row0) {
  __shared__ __align__(16) double lds[4 * X128_STAGE];  // {A,B} x 2 buffers
  __shared__ double scs[128];                            // h^2 / a_b of the workgroup's systems
  {
    // The MFMA phase and the store phase of a workgroup take about equally long (the stores drain at the HBM
    // write rate) and do not overlap within it.  Two workgroups share a CU; started together they stay in
    // lockstep -- all computing, then all storing.  The second half of the first round therefore starts one
    // MFMA phase late (about 64 cycles per MFMA), so that from then on one workgroup of a CU computes while
    // the other drains: measured 274 -> 245 us at 256x256 / 2x2 / 1024 systems.  Placement only affects speed.
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin >= 256u && lin < 512u)
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);  // 2 x 127 x 64 cycles
  }
  const int b = f.lr_blocks[blockIdx.z];
  const int p = b / f.ncb, q = b % f.ncb;
  const int n1 = f.n1, N = f.N;
  const BlockSide& sd = f.sides[b];
  const int nct = (n1 + 127) / 128;
  const int iv = blockIdx.x / nct + 1;           // mesh row (1-based interior index)
  const int jv0 = 128 * (blockIdx.x % nct) + 1;  // first vertex of the tile
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  // staging: thread t -> row t >> 1, eight consecutive k starting at (t & 1) * 8
  const int srow = threadIdx.x >> 1, sseg = (threadIdx.x & 1) * 8;
  const int mA = blockIdx.y * 128 + srow;
  const bool vA = mA < Mc;
  const int jB = jv0 + srow;
  const bool vB = jB <= n1;
  if (threadIdx.x < 128) {
    const int m = blockIdx.y * 128 + threadIdx.x;
    scs[threadIdx.x] = m < Mc ? f.y[size_t(m) * f.nGp + f.sblk0 + b] : 0.0;  // h^2 / a_b (visible after the first barrier below)
  }
  int cend[4];
  const double* pAs[4];
  const double* pBs[4];
  int tot = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const ExtSide es = sd.s[s];
    pAs[s] = pBs[s] = nullptr;
    if (es.mode == 2) {
      tot += es.nch;
      if (vA) pAs[s] = f.y + size_t(mA) * f.nGp + es.off + sseg;
      if (vB) pBs[s] = f.G + es.gtab + size_t(h0_row(s, iv, jB, N, n1)) * (es.nch * BK) + sseg;
    }
    cend[s] = tot;
  }
  auto pick = [&](int ch, const double* const* ps) -> const double* {
    const int s = (ch >= cend[0]) + (ch >= cend[1]) + (ch >= cend[2]);
    const int lc = ch - (s == 0 ? 0 : s == 1 ? cend[0] : s == 2 ? cend[1] : cend[2]);
    const double* ptr = s == 0 ? ps[0] : s == 1 ? ps[1] : s == 2 ? ps[2] : ps[3];
    return ptr ? ptr + lc * BK : nullptr;
  };
  auto load8 = [&](const double* ptr, double* v) {
    load4_aligned(ptr, v);
    load4_aligned(ptr ? ptr + 4 : nullptr, v + 4);
  };
  auto store8 = [&](double* sbuf, const double* v) {
    double2* dst = reinterpret_cast<double2*>(sbuf + srow * LDK + sseg);  // 144-byte rows, 64-byte segments
#pragma unroll
    for (int x = 0; x < 4; ++x) dst[x] = double2{v[2 * x], v[2 * x + 1]};
  };
  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  double va[8], vb[8];
  if (tot > 0) {
    load8(pick(0, pAs), va);
    load8(pick(0, pBs), vb);
  }
  const int fr = lane & 15, kq = lane >> 4;
  // everything the epilogue needs from memory is fetched before the first store: a load after a store would
  // make its s_waitcnt vmcnt wait for the stores as well (one counter, in order)
  double w_own[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int jj = jv0 + wc * 64 + j * 16 + fr;  // 1-based
    w_own[j] = jj <= n1 ? f.W[(iv - 1) * n1 + (jj - 1)] : 0.0;
  }
  for (int ch = 0; ch < tot; ++ch) {
    double* sA = lds + (ch & 1) * 2 * X128_STAGE;
    double* sB = sA + X128_STAGE;
    store8(sA, va);
    store8(sB, vb);
    __syncthreads();
    if (ch + 1 < tot) {
      load8(pick(ch + 1, pAs), va);
      load8(pick(ch + 1, pBs), vb);
    }
    const double* pa = sA + (wr * 64 + fr) * LDK + kq;
    const double* pb = sB + (wc * 64 + fr) * LDK + kq;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = pa[i * 16 * LDK + kk];
        bf[i] = pb[i * 16 * LDK + kk];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  if (tot == 0) __syncthreads();  // scs
  // epilogue: add the particular solution, swap between lane pairs so that every lane owns two adjacent
  // vertices of one 16-vertex block, 16-byte stores (see k_extend)
  const bool odd = lane & 1;
  const long long grow = (long long)(p * N + iv - 1) * f.nc + q * N - 1;
  int jcol[2];
  bool ok0[2], ok1[2];
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {  // pair hp of column blocks: (0,1) and (2,3); even lanes take the first, odd the second
    jcol[hp] = jv0 + wc * 64 + (2 * hp + (odd ? 1 : 0)) * 16 + fr - (odd ? 1 : 0);  // first of the two vertices, 1-based
    ok0[hp] = jcol[hp] <= n1;
    ok1[hp] = jcol[hp] + 1 <= n1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ml = wr * 64 + i * 16 + kq + 4 * g;
      const int m = blockIdx.y * 128 + ml;
      const double sc = scs[ml];
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const double x0 = acc[i][2 * hp][g] + sc * w_own[2 * hp], x1 = acc[i][2 * hp + 1][g] + sc * w_own[2 * hp + 1];
        const double got = lane_swap1(odd ? x0 : x1);
        if (m >= Mc) continue;
        double* dst = U + (row0 + m) * f.dim + grow + jcol[hp];
        const double lo = odd ? got : x0, hi = odd ? x1 : got;
        if (ok1[hp]) *reinterpret_cast<double2_u*>(dst) = double2_u{lo, hi};
        else if (ok0[hp]) dst[0] = lo;
      }
    }
}

// interface values that k_expand does not write: cross points and the edges recovered node by node
__global__ void k_scatter_interface(FemDev f, int Mc, double* __restrict__ U, long long row0) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int m = blockIdx.y;
  if (i >= f.nscat || m >= Mc) return;
  const int v = f.scat[i];
  U[(row0 + m) * f.dim + f.vmap[v]] = f.y[size_t(m) * f.nGp + v];
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
// host: geometry, compression, symbolic tile Cholesky, tables
// ============================================================================================
namespace {

using hostla::ld;
using hostla::Mat;

struct Edge {
  int hv, p, q;  // hv 0: horizontal (r = pN, c in block column q); 1: vertical (c = qN, r in block row p)
  int b0, b1;    // up/dn or lf/rt block indices
};

// one parameter-independent block of the reduced matrix: coef(kind, b) * tab at (rpos, cpos)
struct Small {
  int rpos, cpos;
  Mat tab;
  int kind;
  std::array<int, 4> b;
};

// compressed representation of an active edge (shared by all edges with the same surroundings)
struct Comp {
  int r = 0;
  Mat W;               // n1 x r   orthonormal basis of the coupling range
  Mat Kt;              // r x r    (W^T K^-1 W)^-1
  Mat P;               // n1 x r   K^-1 W Kt
  std::vector<ld> gt;  // r        Kt W^T K^-1 g_f
  std::vector<ld> p0;  // n1       (K^-1 - P W^T K^-1) g_f
  Mat KiW;             // n1 x r   K^-1 W          (closed-form edges: u_e = (KiW c_e + wK) / s_e)
  std::vector<ld> wK;  // n1       K^-1 g_f
};

struct TermAcc {
  std::array<int, 5> key;
  std::vector<double> tab;
  int r_lo, r_hi, c_lo, c_hi;
};

template <class Tp>
int upload(Tp** dptr, const std::vector<Tp>& h) {
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(Tp);
  ROM_HIP(hipMalloc(dptr, bytes));
  if (!h.empty()) ROM_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
  return ROM_OK;
}

// n1p x n1p fp64 table (zero padded) from a long double matrix of at most that size
void put_table(std::vector<double>& pool, size_t idx, int n1p, const Mat& A, bool transposed) {
  double* dst = pool.data() + idx * size_t(n1p) * n1p;
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < A.c; ++j) {
      if (transposed) dst[size_t(j) * n1p + i] = double(A(i, j));
      else dst[size_t(i) * n1p + j] = double(A(i, j));
    }
}

}  // namespace

extern "C" int rom_fem_destroy(rom_fem* f) {
  if (!f) return ROM_OK;
  hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->d_A0, f->d_G, f->d_Qp, f->d_kmax, f->d_epos, f->d_yhat, f->d_W, f->d_g, f->d_desc, f->d_terms, f->d_pool,
                  f->d_kptr, f->d_kpair, f->d_colptr, f->d_colrow, f->d_colti, f->d_sides, f->d_vmap, f->d_L,
                  f->d_invL, f->d_y, f->d_Bt, f->d_P, f->d_vec, f->d_rhs, f->d_pre, f->d_exp, f->d_xred, f->d_groups, f->d_cm,
                  f->d_item_group, f->d_item_k, f->d_rowent, f->d_lr_blocks, f->d_gen_blocks, f->d_scat, f->d_dgroups, f->d_dweight, f->d_ditem_group,
                  f->d_ditem_k, f->d_dmat, f->d_scb};
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
  f->n1p = (n1 + TB - 1) / TB * TB;
  f->nr = nrb * N - 1;
  f->nc = ncb * N - 1;
  f->dim = int64_t(f->nr) * f->nc;
  const int n1p = f->n1p;

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
  f->ncross = ncross;
  // block -> side -> edge id   (sides: 0 top, 1 bottom, 2 left, 3 right)
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

  // ---- edges eliminated in closed form: a maximal set no two of which touch the same block (greedy);
  //      their self-interaction is (a_b0 + a_b1) K with K parameter independent and they do not couple
  //      to each other ------------------------------------------------------------------------------------
  std::vector<char> is_pre(E, 0);
  if (!getenv("ROMHC_NO_PREELIM")) {
    std::vector<char> busy(nrb * ncb, 0);
    for (int e = 0; e < E; ++e)
      if (!busy[edges[e].b0] && !busy[edges[e].b1]) { is_pre[e] = 1; busy[edges[e].b0] = busy[edges[e].b1] = 1; }
  }
  std::vector<int> pre_list, pre_index(E, -1);
  for (int e = 0; e < E; ++e)
    if (is_pre[e]) { pre_index[e] = int(pre_list.size()); pre_list.push_back(e); }
  const int npre = int(pre_list.size());
  const int nact = E - npre;

  // ---- elimination order of the active edges (greedy minimum degree on the graph: shared block, or
  //      common neighbour of a closed-form edge) ----------------------------------------------------------
  std::vector<int> order, ord_of(E, -1);
  {
    std::vector<std::set<int>> g(E);
    for (int e = 0; e < E; ++e)
      if (!is_pre[e])
        for (int x : adj[e])
          if (!is_pre[x]) g[e].insert(x);
    for (int e : pre_list)
      for (int x : adj[e])
        for (int y : adj[e])
          if (x != y && !is_pre[x] && !is_pre[y]) g[x].insert(y);
    std::vector<char> done(E, 0);
    for (int step = 0; step < nact; ++step) {
      int best = -1;
      size_t bd = 0;
      for (int e = 0; e < E; ++e) {
        if (done[e] || is_pre[e]) continue;
        if (best < 0 || g[e].size() < bd) { best = e; bd = g[e].size(); }
      }
      done[best] = 1;
      ord_of[best] = int(order.size());
      order.push_back(best);
      std::vector<int> nb(g[best].begin(), g[best].end());
      for (int x : nb) {
        g[x].erase(best);
        for (int y : nb)
          if (x != y) g[x].insert(y);
      }
    }
  }

  // ---- unit-block tables in long double ----------------------------------------------------------------
  const ld PI = acosl(-1.0L);
  Mat Q(n1, n1), rho(n1, N + 1);
  std::vector<ld> lam(n1), kappa(n1);
  for (int j = 1; j <= n1; ++j) {
    lam[j - 1] = 2.0L - 2.0L * cosl(PI * j / N);
    for (int m = 1; m <= n1; ++m) Q(j - 1, m - 1) = sqrtl(2.0L / N) * sinl(PI * j * m / (ld)N);
  }
  for (int m = 0; m < n1; ++m) {
    ld phi = acoshl(1.0L + lam[m] / 2.0L);
    ld den = -expm1l(-2.0L * N * phi);  // 1 - exp(-2 N phi)
    for (int i = 0; i <= N; ++i) rho(m, i) = expl(-phi * i) * (-expm1l(-2.0L * (N - i) * phi)) / den;
    kappa[m] = 1.0L + lam[m] / 2.0L - rho(m, 1);
  }
  std::vector<double> rho_d(rho.v.size());
  for (size_t i = 0; i < rho.v.size(); ++i) rho_d[i] = double(rho.v[i]);
  // W = L^{-1} 1 = Q (s s^T / (lam_l + lam_m)) Q
  std::vector<double> Wd(size_t(n1) * n1);
  Mat Wl(n1, n1);
  {
    std::vector<ld> sv(n1, 0.0L);
    Mat Z(n1, n1);
    for (int m = 0; m < n1; ++m)
      for (int j = 0; j < n1; ++j) sv[m] += Q(j, m);
    for (int l = 0; l < n1; ++l)
      for (int m = 0; m < n1; ++m) Z(l, m) = sv[l] * sv[m] / (lam[l] + lam[m]);
    Wl = hostla::mul(Q, hostla::mul_nt(Z, Q));
    for (size_t i = 0; i < Wd.size(); ++i) Wd[i] = double(Wl.v[i]);
  }
  const double h2 = 1.0 / (double(N) * double(N));
  // interface rhs of an edge: h^2 (1 + W at the two adjacent interior lines), by orientation
  std::vector<ld> gE[2];
  for (int hv = 0; hv < 2; ++hv) {
    gE[hv].resize(n1);
    for (int t = 0; t < n1; ++t)
      gE[hv][t] = (ld)h2 * (1.0L + (hv == 0 ? Wl(N - 2, t) + Wl(0, t) : Wl(t, N - 2) + Wl(t, 0)));
  }
  // K = tridiag(-1/2, 2, -1/2) - T_same = Q diag(kappa) Q, kappa_m = 1 + lam_m/2 - rho_m(1)
  Mat Kmat(n1, n1), Kinv(n1, n1);
  if (E > 0) {
    Mat QK(n1, n1), QKi(n1, n1);
    for (int j = 0; j < n1; ++j)
      for (int m = 0; m < n1; ++m) {
        QK(j, m) = Q(j, m) * kappa[m];
        QKi(j, m) = Q(j, m) / kappa[m];
      }
    Kmat = hostla::mul_nt(QK, Q);
    Kinv = hostla::mul_nt(QKi, Q);
  }
  // Dirichlet-to-Neumann tables T[sr*4+sc][t][k] = H_sc[interior vertex next to node t of side sr][k]
  // and their products with K^-1, built on first use
  std::array<Mat, 16> Tm_, TK_;
  std::array<char, 16> haveT{}, haveTK{};
  auto Tm = [&](int id) -> const Mat& {
    if (!haveT[id]) {
      const int sr = id >> 2, sc = id & 3;
      Mat V(n1, n1);
      for (int t = 0; t < n1; ++t) {
        int i, j;
        switch (sr) {
          case 0: i = 1; j = t + 1; break;
          case 1: i = N - 1; j = t + 1; break;
          case 2: i = t + 1; j = 1; break;
          default: i = t + 1; j = N - 1; break;
        }
        const int hr = h0_row(sc, i, j, N, n1);
        const int ii = hr / n1 + 1, jj = hr % n1 + 1;
        for (int m = 0; m < n1; ++m) V(t, m) = Q(jj - 1, m) * rho(m, ii);
      }
      Tm_[id] = hostla::mul_nt(V, Q);
      haveT[id] = 1;
    }
    return Tm_[id];
  };
  auto TK = [&](int id) -> const Mat& {
    if (!haveTK[id]) {
      TK_[id] = hostla::mul(Tm(id), Kinv);
      haveTK[id] = 1;
    }
    return TK_[id];
  };

  // ---- compression of the edges (shared by all edges with the same surroundings) -------------------------------
  const bool compress = !getenv("ROMHC_NO_COMPRESS");
  ld ctol = 1e-17L;
  if (const char* s = getenv("ROMHC_COMPRESS_TOL")) ctol = (ld)atof(s);
  std::map<std::vector<int>, int> sig_id;
  std::vector<Comp> comps;
  std::vector<int> comp_of(E, -1);
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    std::vector<int> sig{ed.hv};
    for (int blk : {ed.b0, ed.b1}) {
      const int sf = side_of(blk, e);
      for (int s2 = 0; s2 < 4; ++s2)
        if (s2 != sf && bside[blk][s2] >= 0) sig.push_back(sf * 4 + s2);
    }
    bool x0 = false, x1 = false;
    for (auto& c : xc)
      if (c.edge == e) (c.node == 0 ? x0 : x1) = true;
    sig.push_back(100 + (x0 ? 1 : 0) + (x1 ? 2 : 0));
    auto it = sig_id.find(sig);
    if (it != sig_id.end()) { comp_of[e] = it->second; continue; }
    Comp cp;
    Mat Wb;
    if (compress) {
      const int ntab = int(sig.size()) - 2;
      Mat C(n1, ntab * n1 + 2);
      for (int t = 0; t < ntab; ++t) {
        const Mat& Tt = Tm(sig[1 + t]);
        ld mx = 0;
        for (ld v : Tt.v) mx = std::max(mx, fabsl(v));
        if (mx == 0.0L) mx = 1.0L;
        for (int i = 0; i < n1; ++i)
          for (int k = 0; k < n1; ++k) C(i, t * n1 + k) = Tt(i, k) / mx;
      }
      if (x0) C(0, ntab * n1) = 1.0L;
      if (x1) C(n1 - 1, ntab * n1 + 1) = 1.0L;
      Wb = hostla::range_basis(C, ctol);
    }
    if (!compress || Wb.c >= n1) {  // nothing to gain: nodal unknowns
      cp.r = n1;
      cp.W = hostla::identity(n1);
      cp.Kt = Kmat;
      cp.P = hostla::identity(n1);
      cp.gt = gE[ed.hv];
      cp.p0.assign(n1, 0.0L);
      cp.KiW = Kinv;
      cp.wK = hostla::matvec(Kinv, gE[ed.hv]);
    } else {
      cp.r = Wb.c;
      cp.W = Wb;
      Mat KiW = hostla::mul(Kinv, Wb);
      Mat G = hostla::mul_tn(Wb, KiW);
      for (int i = 0; i < G.r; ++i)
        for (int j = 0; j < i; ++j) G(i, j) = G(j, i) = (G(i, j) + G(j, i)) / 2;
      if (!hostla::spd_inverse(G, cp.Kt)) { rom_set_error("internal: compressed edge block not positive definite"); return ROM_ERR_INVALID; }
      cp.P = hostla::mul(KiW, cp.Kt);
      std::vector<ld> v = hostla::matvec(Kinv, gE[ed.hv]);
      std::vector<ld> wv = hostla::matvec(hostla::transpose(Wb), v);
      cp.gt = hostla::matvec(cp.Kt, wv);
      std::vector<ld> pw = hostla::matvec(cp.P, wv);
      cp.p0.resize(n1);
      for (int i = 0; i < n1; ++i) cp.p0[i] = v[i] - pw[i];
      cp.KiW = KiW;
      cp.wK = v;
    }
    comp_of[e] = int(comps.size());
    sig_id[sig] = comp_of[e];
    comps.push_back(std::move(cp));
  }


  // kmax[d]: sine modes with rho_mode(d) >= 1e-18 (rounded up to the K chunk): what the extension needs at
  // distance d from a side.  A compressed edge enters the extension through its reduced unknowns instead
  // when rank + 1 (padded) is not much above the average mode count.
  std::vector<int> kmax(N + 1, n1p);
  double kavg = 0;
  for (int dd = 1; dd <= N; ++dd) {
    int last = -1;
    for (int m = 0; m < n1; ++m)
      if (rho_d[size_t(m) * (N + 1) + std::min(dd, N)] >= 1e-18) last = m;
    kmax[dd] = std::min(n1p, std::max(BK, (last + 1 + BK - 1) / BK * BK));
    if (dd <= n1) kavg += kmax[dd] / double(std::max(n1, 1));
  }
  kmax[0] = n1p;
  std::vector<int> rp(comps.size(), 0);
  std::vector<char> use_lr(comps.size(), 0);
  for (size_t c = 0; c < comps.size(); ++c) {
    rp[c] = (comps[c].r + 1 + BK - 1) / BK * BK;
    // (up to a quarter more K than the truncated modes is still a gain: flat K, wide tiles, no edge transforms)
    use_lr[c] = comps[c].r < n1 && rp[c] <= 1.25 * kavg && n1 > 0 && !getenv("ROMHC_NO_LOWRANK_EXT");
  }

  // ---- layout of the reduced vector: edge groups in elimination order, every cross point right behind
  //      the adjacent active edge that is eliminated last ------------------------------------------------------
  std::vector<int> zpos(E, -1), rk(E, 0), xred(ncross, -1), xhost(ncross, -1);
  for (int x = 0; x < ncross; ++x)
    for (auto& c : xc)
      if (c.cross == x && !is_pre[c.edge] && (xhost[x] < 0 || ord_of[c.edge] > ord_of[xhost[x]])) xhost[x] = c.edge;
  int nred = 0;
  {
    // one tile in total: cross points first, so that their couplings are table ROWS of the upper triangle
    // (the single-tile assembly reads row segments; a cross behind its edges would cost one 8-byte read per
    // edge row instead)
    int total = ncross;
    for (int e : order) total += comps[comp_of[e]].r;
    if (total <= TB)
      for (int x = 0; x < ncross; ++x) xred[x] = nred++;
  }
  for (int e : order) {
    zpos[e] = nred;
    rk[e] = comps[comp_of[e]].r;
    nred += rk[e];
    f->ranks.push_back(rk[e]);
    for (int x = 0; x < ncross; ++x)
      if (xhost[x] == e && xred[x] < 0) xred[x] = nred++;
  }
  for (int x = 0; x < ncross; ++x)
    if (xred[x] < 0) xred[x] = nred++;
  const int T = (nred + TB - 1) / TB;
  f->nred = nred;
  f->T = T;
  f->nGa = T * TB;
  // nodal layout behind the reduced part: one n1p block per edge, then the cross block
  std::vector<int> npos(E, -1);
  for (int e = 0; e < E; ++e) npos[e] = f->nGa + e * n1p;
  f->xb0 = f->nGa + E * n1p;
  f->nGp = f->xb0 + (ncross > 0 ? (ncross + TB - 1) / TB * TB : 0);
  std::vector<int> cpos(E, -1);  // [z_f, 1/s_f] blocks of the edges that enter the extension in compressed form
  for (int e : order)
    if (use_lr[comp_of[e]]) {
      cpos[e] = f->nGp;
      f->nGp += rp[comp_of[e]];
    }
  for (int e : pre_list)  // closed-form edges kept in compressed form: [c_e / s_e, 1/s_e]
    if (use_lr[comp_of[e]]) {
      cpos[e] = f->nGp;
      f->nGp += rp[comp_of[e]];
    }
  // scalar block: 1/(a_p + a_q) of every edge, then h^2/a_b of every block -- with it the expansion stage is a
  // LINEAR map of the interface vector (it never reads the parameters)
  f->spos0 = f->nGp;
  f->n_all_edges = E;
  f->nsc = E + nrb * ncb;
  f->nGp += (f->nsc + BK - 1) / BK * BK;

  // ---- blocks of the reduced matrix --------------------------------------------------------------------
  std::vector<Small> smalls;
  auto add_small = [&](int rpos, int cpos, const Mat& tab, int kind, std::array<int, 4> b) {
    smalls.push_back(Small{rpos, cpos, tab, kind, b});
    if (rpos != cpos) smalls.push_back(Small{cpos, rpos, hostla::transpose(tab), kind, b});
  };
  for (int e : order) {
    const Edge& ed = edges[e];
    const Comp& ce = comps[comp_of[e]];
    add_small(zpos[e], zpos[e], ce.Kt, 1, {ed.b0, ed.b1, 0, 0});
    for (int e2 : adj[e]) {
      if (is_pre[e2] || e2 <= e) continue;
      const int blk = shared_block(e, e2);
      const int id = side_of(blk, e) * 4 + side_of(blk, e2);
      const Comp& c2 = comps[comp_of[e2]];
      add_small(zpos[e], zpos[e2], hostla::mul(hostla::mul_tn(ce.W, Tm(id)), c2.W), 0, {blk, 0, 0, 0});
    }
  }
  for (auto& c : xc) {
    if (is_pre[c.edge]) continue;  // folded into the closed-form tables
    const Comp& ce = comps[comp_of[c.edge]];
    Mat row(1, ce.r);
    for (int k = 0; k < ce.r; ++k) row(0, k) = ce.W(c.node, k);
    add_small(xred[c.cross], zpos[c.edge], row, 2, {edges[c.edge].b1, edges[c.edge].b0, 0, 0});
  }
  for (int x = 0; x < ncross; ++x) {
    const int p = crosses[x].first, q = crosses[x].second;
    Mat one(1, 1);
    one(0, 0) = 1.0L;
    add_small(xred[x], xred[x], one, 3, {(p - 1) * ncb + (q - 1), (p - 1) * ncb + q, p * ncb + (q - 1), p * ncb + q});
  }

  // ---- closed-form edges: neighbours, reduced-matrix blocks, rhs terms, back substitution ---------------------------
  std::vector<double> vecs;  // vector table
  auto push_vec = [&](const std::vector<ld>& v, int padded) {
    const int off = int(vecs.size());
    for (ld x : v) vecs.push_back(double(x));
    for (int i = int(v.size()); i < padded; ++i) vecs.push_back(0.0);
    return off;
  };
  std::vector<RhsTerm> rhs_terms;
  std::vector<PreEdge> pre_edges;   // closed-form edges recovered node by node (k_back_pre)
  std::vector<CoefGroup> groups;    // coefficient blocks built by k_coef
  std::vector<double> cm;           // matrices of the closed-form edges kept in compressed form
  std::map<int, int> bt_of_id;               // T table id -> B^T table index
  std::vector<std::pair<int, Mat>> bt_extra;  // cross-block tables (index, n1 x ncross)
  int nbt = 0;
  double pre_flops = 0;
  for (int i = 0; i < npre; ++i) {
    const int e = pre_list[i];
    const Edge& pe = edges[e];
    const bool lr_e = cpos[e] >= 0;
    PreEdge P;
    memset(&P, 0, sizeof(P));
    CoefGroup cg;
    memset(&cg, 0, sizeof(cg));
    const Comp& cpe = comps[comp_of[e]];
    cg.kind = 1; cg.cpos = cpos[e]; cg.r = cpe.r; cg.w = rp[comp_of[e]]; cg.b0 = pe.b0; cg.b1 = pe.b1;
    auto add_cterm = [&](int src, int blk, const Mat& Mt, int voff, int u0, int u1) {  // Mt: len x r_e
      if (cg.nterm >= 8) return false;
      cg.t[cg.nterm++] = CoefTerm{src, Mt.r, blk, int(cm.size()), voff, u0, u1};
      for (ld v : Mt.v) cm.push_back(double(v));
      return true;
    };
    P.pos = npos[e];
    P.e0 = pe.b0;
    P.e1 = pe.b1;
    const std::vector<ld> we = hostla::matvec(Kinv, gE[pe.hv]);
    P.woff = push_vec(we, n1p);
    struct Ent { int pos, len, blk; Mat X, Y; };  // X: len x n1 coupling to e (without its weight), Y = X K^-1
    std::vector<Ent> ents;
    for (int u : adj[e]) {
      const int blk = shared_block(e, u);
      const int id = side_of(blk, u) * 4 + side_of(blk, e);
      const Comp& cu = comps[comp_of[u]];
      Ent en{zpos[u], cu.r, blk, hostla::mul_tn(cu.W, Tm(id)), hostla::mul_tn(cu.W, TK(id))};
      ents.push_back(std::move(en));
      if (lr_e) {
        // c_e += a_blk * (W_e^T T^(e,u) P_u z_u + W_e^T T^(e,u) p0_u / s_u)
        const Mat WT = hostla::mul_tn(cpe.W, Tm(side_of(blk, e) * 4 + side_of(blk, u)));  // r_e x n1
        const Mat Mt = hostla::transpose(hostla::mul(WT, cu.P));                           // r_u x r_e
        const int voff = push_vec(hostla::matvec(WT, cu.p0), cpe.r);
        if (!add_cterm(zpos[u], blk, Mt, voff, edges[u].b0, edges[u].b1)) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
        pre_flops += 2.0 * cpe.r * double(cu.r);
      } else {
        if (!bt_of_id.count(id)) bt_of_id[id] = nbt++;
        if (P.nnb >= 8) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
        P.nb[P.nnb++] = PreNb{npos[u], blk, n1p / BK, bt_of_id[id]};
        pre_flops += 2.0 * n1p * double(n1p);
      }
    }
    Mat Btx(n1, std::max(ncross, 1));
    bool has_x = false;
    for (auto& c : xc) {
      if (c.edge != e) continue;
      Ent en{xred[c.cross], 1, -1, Mat(1, n1), Mat(1, n1)};
      en.X(0, c.node) = 1.0L;
      for (int k = 0; k < n1; ++k) {
        en.Y(0, k) = Kinv(c.node, k);
        Btx(k, c.cross) = Kinv(k, c.node);
      }
      ents.push_back(std::move(en));
      has_x = true;
      if (lr_e) {  // c_e += (s_e / 2) W_e[node, :]^T u_x
        Mat Mt(1, cpe.r);
        for (int k = 0; k < cpe.r; ++k) Mt(0, k) = cpe.W(c.node, k);
        if (!add_cterm(xred[c.cross], -1, Mt, -1, 0, 0)) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
      }
    }
    if (lr_e) groups.push_back(cg);
    if (has_x && !lr_e) {
      if (P.nnb >= 8) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
      P.nb[P.nnb++] = PreNb{f->xb0, -1, (ncross + BK - 1) / BK, nbt};
      bt_extra.push_back({nbt++, Btx});
      pre_flops += 2.0 * n1p * double((ncross + BK - 1) / BK * BK);
    }
    for (size_t u = 0; u < ents.size(); ++u) {
      const Ent& eu = ents[u];
      rhs_terms.push_back(RhsTerm{eu.pos, eu.len, push_vec(hostla::matvec(eu.Y, gE[pe.hv]), eu.len), eu.blk >= 0 ? 0 : 1,
                                  std::max(eu.blk, 0), pe.b0, pe.b1});
      for (size_t v = 0; v <= u; ++v) {
        const Ent& ev = ents[v];
        Mat R = hostla::mul_nt(eu.Y, ev.X);
        if (u == v)
          for (int a = 0; a < R.r; ++a)
            for (int b = 0; b < a; ++b) R(a, b) = R(b, a) = (R(a, b) + R(b, a)) / 2;
        if (eu.blk >= 0 && ev.blk >= 0)
          add_small(eu.pos, ev.pos, R, 4, {std::min(eu.blk, ev.blk), std::max(eu.blk, ev.blk), pe.b0, pe.b1});
        else if (eu.blk >= 0 || ev.blk >= 0)
          add_small(eu.pos, ev.pos, R, 5, {std::max(eu.blk, ev.blk), 0, 0, 0});
        else
          add_small(eu.pos, ev.pos, R, 6, {0, 0, pe.b0, pe.b1});
      }
    }
    if (!lr_e) pre_edges.push_back(P);
  }
  f->npre = int(pre_edges.size());

  // ---- tile mask + symbolic fill --------------------------------------------------------------------
  std::vector<char> mask(size_t(T) * T, 0);
  auto M_ = [&](int i, int j) -> char& { return mask[size_t(i) * T + j]; };
  for (int t = 0; t < T; ++t) M_(t, t) = 1;
  for (const Small& s : smalls)
    for (int tr = s.rpos / TB; tr <= (s.rpos + s.tab.r - 1) / TB; ++tr)
      for (int tc = s.cpos / TB; tc <= (s.cpos + s.tab.c - 1) / TB; ++tc) M_(tr, tc) = M_(tc, tr) = 1;
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

  // ---- distribute the blocks over the tiles: one 64x64 table per (tile, coefficient formula) ---------------------
  std::vector<std::vector<TermAcc>> slot_terms(f->nslots);
  for (const Small& s : smalls)
    for (int tr = s.rpos / TB; tr <= (s.rpos + s.tab.r - 1) / TB; ++tr)
      for (int tc = s.cpos / TB; tc <= (s.cpos + s.tab.c - 1) / TB && tc <= tr; ++tc) {
        const int slot = f->slot_of[size_t(tr) * T + tc];
        if (slot < 0) { rom_set_error("internal: reduced-matrix block outside the tile mask"); return ROM_ERR_INVALID; }
        const std::array<int, 5> key{s.kind, s.b[0], s.b[1], s.b[2], s.b[3]};
        TermAcc* ta = nullptr;
        for (auto& cand : slot_terms[slot])
          if (cand.key == key) ta = &cand;
        if (!ta) {
          slot_terms[slot].push_back(TermAcc{key, std::vector<double>(4096, 0.0), TB, 0, TB, 0});
          ta = &slot_terms[slot].back();
        }
        const int i0 = std::max(s.rpos, tr * TB), i1 = std::min(s.rpos + s.tab.r, (tr + 1) * TB);
        const int j0 = std::max(s.cpos, tc * TB), j1 = std::min(s.cpos + s.tab.c, (tc + 1) * TB);
        for (int i = i0; i < i1; ++i)
          for (int j = j0; j < j1; ++j)
            ta->tab[size_t(i - tr * TB) * TB + (j - tc * TB)] += double(s.tab(i - s.rpos, j - s.cpos));
        ta->r_lo = std::min(ta->r_lo, i0 - tr * TB);
        ta->r_hi = std::max(ta->r_hi, i1 - tr * TB);
        ta->c_lo = std::min(ta->c_lo, j0 - tc * TB);
        ta->c_hi = std::max(ta->c_hi, j1 - tc * TB);
      }
  std::vector<GenTerm> terms;
  std::vector<double> pool;
  f->desc.resize(f->nslots);
  for (int s = 0; s < f->nslots; ++s) {
    TileDesc d;
    memset(&d, 0, sizeof(d));
    d.ti = slots[s].first;
    d.tj = slots[s].second;
    d.diag = d.ti == d.tj;
    d.ndr = std::max(0, std::min(TB, nred - d.ti * TB));
    d.t0 = int(terms.size());
    for (auto& ta : slot_terms[s]) {
      GenTerm g;
      g.tab = int(pool.size() / 4096);
      g.r_lo = short(ta.r_lo); g.r_hi = short(ta.r_hi); g.c_lo = short(ta.c_lo); g.c_hi = short(ta.c_hi);
      g.kind = ta.key[0];
      for (int q = 0; q < 4; ++q) g.b[q] = ta.key[1 + q];
      terms.push_back(g);
      pool.insert(pool.end(), ta.tab.begin(), ta.tab.end());
    }
    d.t1 = int(terms.size());
    f->desc[s] = d;
  }
  slot_terms.clear();
  smalls.clear();
  // row program of the single-tile solve (k_solve1): per tile row, the table row segments with col >= row
  std::vector<RowEnt> rowents;
  f->fused1 = T == 1 && f->desc[0].t1 - f->desc[0].t0 <= COEF_MAX && f->nGa == TB;
  if (f->fused1) {
    const TileDesc& d0 = f->desc[0];
    for (int r = 0; r < TB; ++r) {
      const size_t first = rowents.size();
      for (int t = d0.t0; t < d0.t1; ++t) {
        const GenTerm& g = terms[t];
        if (r < g.r_lo || r >= g.r_hi) continue;
        const double* rowp = pool.data() + size_t(g.tab) * 4096 + size_t(r) * TB;
        int lo = TB, hi = 0;
        for (int c = std::max<int>(r, g.c_lo); c < g.c_hi; ++c)
          if (rowp[c] != 0.0) { lo = std::min(lo, c); hi = c + 1; }
        if (hi <= lo) continue;
        rowents.push_back(RowEnt{int(size_t(g.tab) * 4096 + size_t(r) * TB), r | ((t - d0.t0) << 8), lo, hi});
      }
      if (rowents.size() > first) rowents.back().meta |= 1 << 16;
    }
    while (rowents.size() % ROW_BATCH) rowents.push_back(RowEnt{0, 0, 0, 0});  // no-ops
  }
  f->nrowent = int(rowents.size());
  ROM_TRY(upload(&f->d_rowent, rowents));

  // ---- block sides, vmap, parameter-independent part of the reduced rhs --------------------------------------------
  std::vector<int> vmap(std::max(f->nGp, 1), -1);
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    for (int t = 0; t < n1; ++t) {
      int r, c;  // 1-based inner vertex coordinates
      if (ed.hv == 0) { r = ed.p * N; c = ed.q * N + t + 1; }
      else { r = ed.p * N + t + 1; c = ed.q * N; }
      vmap[npos[e] + t] = (r - 1) * f->nc + (c - 1);
    }
  }
  for (int x = 0; x < ncross; ++x) {
    int r = crosses[x].first * N, c = crosses[x].second * N;
    vmap[f->xb0 + x] = (r - 1) * f->nc + (c - 1);
  }
  std::vector<double> g_red(std::max(f->nGa, 1), 0.0);
  for (int e : order) {
    const Comp& ce = comps[comp_of[e]];
    for (int k = 0; k < ce.r; ++k) g_red[zpos[e] + k] = double(ce.gt[k]);
  }
  for (int x = 0; x < ncross; ++x) g_red[xred[x]] = h2;

  // ---- expansion tables of the active edges, back substitution tables of the closed-form ones ------------------------
  const size_t tsz = size_t(n1p) * n1p;
  // table variants of a compressed-edge type: 0 = active edge (P, p0), 1 = closed-form edge (K^-1 W, K^-1 g)
  std::map<std::pair<int, int>, int> ptab_of, p0_of;
  std::vector<std::pair<int, int>> ptab_list;
  auto variant = [&](int c, int v) {
    if (!ptab_of.count({c, v})) {
      ptab_of[{c, v}] = int(ptab_list.size());
      ptab_list.push_back({c, v});
      p0_of[{c, v}] = push_vec(v == 0 ? comps[c].p0 : comps[c].wK, n1p);
    }
    return ptab_of[{c, v}];
  };
  std::vector<ExpEdge> exps;
  for (int e : order) {
    const int c = comp_of[e], pt = variant(c, 0);
    exps.push_back(ExpEdge{zpos[e], (rk[e] + BK - 1) / BK, npos[e], pt, p0_of[{c, 0}], f->spos0 + e});
    if (cpos[e] >= 0) {
      CoefGroup cg;
      memset(&cg, 0, sizeof(cg));
      cg.kind = 0; cg.cpos = cpos[e]; cg.r = rk[e]; cg.w = rp[c]; cg.b0 = edges[e].b0; cg.b1 = edges[e].b1; cg.zpos = zpos[e];
      groups.push_back(cg);
    }
  }
  for (int e : pre_list)
    if (cpos[e] >= 0) {
      const int c = comp_of[e], pt = variant(c, 1);
      exps.push_back(ExpEdge{cpos[e], (comps[c].r + BK - 1) / BK, npos[e], pt, p0_of[{c, 1}], f->spos0 + e});
    }
  f->nexp = int(exps.size());
  std::vector<int> item_group, item_k;
  for (size_t g = 0; g < groups.size(); ++g)
    for (int k = 0; k < groups[g].w; ++k) { item_group.push_back(int(g)); item_k.push_back(k); }
  f->ncoef = int(item_group.size());
  {
    // single-tile path: the coefficient blocks of the closed-form edges as one dense product (k_solve1)
    std::vector<DenseGroup> dgroups;
    std::vector<int> dweight, ditem_group, ditem_k;
    std::vector<std::pair<int, int>> dsrc;  // (group index in `groups`, first item)
    for (size_t g = 0; g < groups.size(); ++g)
      if (groups[g].kind == 1) {
        DenseGroup dg;
        memset(&dg, 0, sizeof(dg));
        dg.cpos = groups[g].cpos; dg.r = groups[g].r; dg.b0 = groups[g].b0; dg.b1 = groups[g].b1;
        dsrc.push_back({int(g), int(ditem_group.size())});
        for (int k = 0; k < groups[g].r; ++k) { ditem_group.push_back(int(dgroups.size())); ditem_k.push_back(k); }
        dgroups.push_back(dg);
      }
    const int ndi = int(ditem_group.size());
    std::vector<double> dmat(size_t(TB) * std::max(ndi, 1), 0.0);
    dweight.assign(dgroups.size() * TB, -2);
    bool ok = f->fused1 && int(dgroups.size()) <= DENSE_GROUPS_MAX;
    for (size_t dgi = 0; dgi < dgroups.size() && ok; ++dgi) {
      const CoefGroup& cg = groups[dsrc[dgi].first];
      for (int t = 0; t < cg.nterm && ok; ++t) {
        const CoefTerm& ct = cg.t[t];
        for (int j = 0; j < ct.len; ++j) {
          if (ct.src + j >= TB) { ok = false; break; }
          dweight[dgi * TB + ct.src + j] = ct.blk >= 0 ? ct.blk : -1;
          for (int k = 0; k < cg.r; ++k) dmat[size_t(ct.src + j) * ndi + dsrc[dgi].second + k] = cm[ct.moff + size_t(j) * cg.r + k];
        }
        if (ct.voff >= 0) {
          DenseGroup& dg = dgroups[dgi];
          if (dg.nv >= 4) { ok = false; break; }
          dg.voff[dg.nv] = ct.voff; dg.vblk[dg.nv] = ct.blk; dg.vu0[dg.nv] = ct.u0; dg.vu1[dg.nv] = ct.u1;
          ++dg.nv;
        }
      }
    }
    if (!ok) f->fused1 = false;
    f->ndg = f->fused1 ? int(dgroups.size()) : 0;
    f->ndi = f->fused1 ? ndi : 0;
    ROM_TRY(upload(&f->d_dgroups, dgroups));
    ROM_TRY(upload(&f->d_dweight, dweight));
    ROM_TRY(upload(&f->d_ditem_group, ditem_group));
    ROM_TRY(upload(&f->d_ditem_k, ditem_k));
    ROM_TRY(upload(&f->d_dmat, dmat));
  }
  {
    std::vector<double> Ptab(std::max<size_t>(ptab_list.size() * tsz, 1), 0.0);
    for (size_t t = 0; t < ptab_list.size(); ++t)
      put_table(Ptab, t, n1p, ptab_list[t].second == 0 ? comps[ptab_list[t].first].P : comps[ptab_list[t].first].KiW, false);
    ROM_TRY(upload(&f->d_P, Ptab));
    std::vector<double> Bt(std::max<size_t>(size_t(nbt) * tsz, 1), 0.0);
    for (auto& kv : bt_of_id) put_table(Bt, kv.second, n1p, TK(kv.first), true);  // (T K^-1)^T: row = node of e
    for (auto& kv : bt_extra) put_table(Bt, kv.first, n1p, kv.second, false);
    ROM_TRY(upload(&f->d_Bt, Bt));
  }

  // ---- device tables of the harmonic extension ---------------------------------------------------------------
  std::vector<double> Qp(size_t(n1p) * n1p, 0.0);
  for (int j = 0; j < n1; ++j)
    for (int m = 0; m < n1; ++m) Qp[size_t(j) * n1p + m] = double(Q(j, m));
  double* d_rho = nullptr;
  ROM_TRY(upload(&f->d_Qp, Qp));
  ROM_TRY(upload(&d_rho, rho_d));
  const size_t hrows = size_t(n1) * n1;
  ROM_HIP(hipMalloc(&f->d_A0, std::max<size_t>(hrows * n1p, 1) * sizeof(double)));
  if (hrows > 0) {
    size_t total = hrows * n1p;
    k_build_A0<<<unsigned((total + 255) / 256), 256, 0, ctx->stream>>>(f->d_A0, f->d_Qp, d_rho, n1, n1p, N);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  hipFree(d_rho);
  {
    ROM_TRY(upload(&f->d_kmax, kmax));

    // Representation of every block side in the extension.  A compressed edge enters through its reduced
    // unknowns when that is cheaper than the distance-truncated sine modes: table G_c = H_0 [P_c, p0_c]
    // = A0 (Q [P_c, p0_c]), one (n1*n1) x rp_c table per compressed-edge type.
    // one table per (compressed-edge type, variant) that a block side actually uses
    std::map<std::pair<int, int>, long long> goff;
    long long gtotal = 0;
    for (int e = 0; e < E; ++e)
      if (cpos[e] >= 0 && !goff.count({comp_of[e], int(is_pre[e])})) {
        goff[{comp_of[e], int(is_pre[e])}] = gtotal;
        gtotal += (long long)hrows * rp[comp_of[e]];
      }
    ROM_CHECK(gtotal < (1ll << 31), "rom_fem_create: extension tables too large");
    ROM_HIP(hipMalloc(&f->d_G, std::max<size_t>(size_t(gtotal), 1) * sizeof(double)));
    for (auto& kv : goff) {
      const int c = kv.first.first;
      const Comp& cp = comps[c];
      const Mat& Pm = kv.first.second == 0 ? cp.P : cp.KiW;
      const std::vector<ld>& pv = kv.first.second == 0 ? cp.p0 : cp.wK;
      Mat Bm = hostla::mul_tn(Pm, Q);  // r x n1
      std::vector<double> Bh(size_t(rp[c]) * n1p, 0.0);
      for (int k = 0; k < cp.r; ++k)
        for (int m = 0; m < n1; ++m) Bh[size_t(k) * n1p + m] = double(Bm(k, m));
      for (int m = 0; m < n1; ++m) {
        ld sacc = 0;
        for (int t = 0; t < n1; ++t) sacc += pv[t] * Q(t, m);
        Bh[size_t(cp.r) * n1p + m] = double(sacc);
      }
      double* d_B = nullptr;
      ROM_TRY(upload(&d_B, Bh));
      ROM_TRY(rom_launch_gemm_nt(ctx, int64_t(hrows), rp[c], n1p, 1.0, f->d_A0, n1p, d_B, n1p, 0.0, f->d_G + kv.second,
                                 rp[c], "setup_gemm_G"));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
      hipFree(d_B);
    }
    f->sides.resize(nrb * ncb);
    std::vector<char> need_tr(E, 0);
    double fl = 0;
    const int npj = (n1 + 15) / 16, npi = (n1 + 3) / 4;
    for (int b = 0; b < nrb * ncb; ++b)
      for (int sdx = 0; sdx < 4; ++sdx) {
        ExtSide& es = f->sides[b].s[sdx];
        memset(&es, 0, sizeof(es));
        const int e = bside[b][sdx];
        if (e < 0) continue;
        const int c = comp_of[e];
        if (cpos[e] >= 0) {
          es = ExtSide{2, cpos[e], rp[c] / BK, comps[c].r, int(goff[{c, int(is_pre[e])}]), edges[e].b0, edges[e].b1};
          fl += 2.0 * 64 * rp[c] * npi * npj;
        } else {
          es.mode = 1;
          es.off = npos[e];
          need_tr[e] = 1;
          for (int pi = 0; pi < npi; ++pi)
            for (int pj = 0; pj < npj; ++pj) {
              int i0 = 4 * pi + 1, j0 = 16 * pj + 1, i1 = std::min(i0 + 3, n1), j1 = std::min(j0 + 15, n1);
              int dist[4] = {i0, N - i1, j0, N - j1};
              fl += 2.0 * 64 * kmax[dist[sdx]];
            }
        }
      }
    std::vector<int> lr_blocks, gen_blocks;
    f->lr_nch = 0;
    for (int b = 0; b < nrb * ncb; ++b) {
      int nlr = 0, nother = 0, nch = 0;
      for (int sdx = 0; sdx < 4; ++sdx) {
        const ExtSide& es = f->sides[b].s[sdx];
        if (es.mode == 2) { ++nlr; nch += es.nch; }
        else if (es.mode != 0) ++nother;
      }
      if (nlr > 0 && nother == 0 && !getenv("ROMHC_NO_EXT_LR")) {
        lr_blocks.push_back(b);
        f->lr_nch = std::max(f->lr_nch, nch);
      } else {
        gen_blocks.push_back(b);
      }
    }
    f->n_lr_blocks = int(lr_blocks.size());
    f->n_gen_blocks = int(gen_blocks.size());
    ROM_TRY(upload(&f->d_lr_blocks, lr_blocks));
    ROM_TRY(upload(&f->d_gen_blocks, gen_blocks));
    std::vector<int> eposv;
    for (int e = 0; e < E; ++e)
      if (need_tr[e]) eposv.push_back(npos[e]);
    f->n_edges = int(eposv.size());
    if (eposv.empty()) eposv.push_back(0);
    ROM_TRY(upload(&f->d_epos, eposv));
    // flops of the extension, per system (for the work accounting)
    f->ext_flops = fl + 2.0 * f->n_edges * double(n1p) * n1p;
  }
  ROM_TRY(upload(&f->d_W, Wd));
  ROM_TRY(upload(&f->d_g, g_red));
  ROM_TRY(upload(&f->d_vec, vecs));
  ROM_TRY(upload(&f->d_pool, pool));
  ROM_TRY(upload(&f->d_terms, terms));
  f->nrhs = int(rhs_terms.size());
  ROM_TRY(upload(&f->d_rhs, rhs_terms));
  ROM_TRY(upload(&f->d_pre, pre_edges));
  ROM_TRY(upload(&f->d_exp, exps));
  ROM_TRY(upload(&f->d_groups, groups));
  ROM_TRY(upload(&f->d_cm, cm));
  ROM_TRY(upload(&f->d_item_group, item_group));
  ROM_TRY(upload(&f->d_item_k, item_k));
  ROM_TRY(upload(&f->d_xred, xred));
  {
    std::vector<int> scb;  // (b0, b1) per scalar: an edge's two blocks, or (block, -1)
    for (int e = 0; e < E; ++e) { scb.push_back(edges[e].b0); scb.push_back(edges[e].b1); }
    for (int b = 0; b < nrb * ncb; ++b) { scb.push_back(b); scb.push_back(-1); }
    ROM_TRY(upload(&f->d_scb, scb));
  }
  ROM_TRY(upload(&f->d_desc, f->desc));
  ROM_TRY(upload(&f->d_kptr, f->kptr));
  ROM_TRY(upload(&f->d_kpair, f->kpair));
  ROM_TRY(upload(&f->d_colptr, f->colptr));
  ROM_TRY(upload(&f->d_colrow, f->colrow));
  ROM_TRY(upload(&f->d_colti, f->colti));
  ROM_TRY(upload(&f->d_sides, f->sides));
  ROM_TRY(upload(&f->d_vmap, vmap));
  {
    std::vector<char> expanded(E, 0);
    for (int e : order) expanded[e] = 1;
    for (int e : pre_list)
      if (cpos[e] >= 0) expanded[e] = 1;
    std::vector<int> scat;
    for (int e = 0; e < E; ++e)
      if (!expanded[e])
        for (int t = 0; t < n1; ++t) scat.push_back(npos[e] + t);
    for (int x = 0; x < ncross; ++x) scat.push_back(f->xb0 + x);
    f->nscat = int(scat.size());
    ROM_TRY(upload(&f->d_scat, scat));
  }

  if (getenv("ROMHC_VERBOSE")) {
    fprintf(stderr, "romhc: %dx%d blocks N=%d: %d edges (%d closed-form, %d of them compressed), reduced size %d -> %d tiles, "
                    "%d slots, %zu terms, kavg %.1f\n", nrb, ncb, N, E, int(pre_list.size()), int(pre_list.size()) - f->npre, nred, T,
            f->nslots, terms.size(), kavg);
    for (size_t c = 0; c < comps.size(); ++c) {
      int cnt = 0;
      for (int e = 0; e < E; ++e) cnt += comp_of[e] == int(c);
      fprintf(stderr, "romhc:   edge type %zu: rank %d (padded %d), %d edges, extension %s\n", c, comps[c].r, rp[c], cnt,
              use_lr[c] ? "from the reduced unknowns" : "sine modes");
    }
    fprintf(stderr, "romhc:   blocks extended by the 128-tile kernel: %d, general kernel: %d\n", f->n_lr_blocks, f->n_gen_blocks);
  }

  // ---- work accounting of this algorithm, per snapshot solve ------------------------------------------------
  double exp_flops = 0;
  for (int e : order) exp_flops += 2.0 * n1p * double((rk[e] + BK - 1) / BK * BK);
  const double back_flops = 2.0 * 4096.0 * (f->nslots + T);
  f->flops_solve = flops + f->ext_flops + back_flops + pre_flops + exp_flops;
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
  // zeroed once: padding slots are read (against zero table entries) before anything writes them
  ROM_HIP(hipMemset(f->d_y, 0, std::max<size_t>(size_t(Mc) * f->nGp, 1) * sizeof(double)));
  ROM_HIP(hipMalloc(&f->d_yhat, std::max<size_t>(size_t(Mc) * f->nGp, 1) * sizeof(double)));
  f->ws_M = Mc;
  return ROM_OK;
}

// enqueue every kernel of one sub-batch (Mc systems, workspace pointers already offset) on `st`
// `stages`: 1 = reduced solve (interface vector: reduced unknowns, cross points, coefficient blocks),
//           2 = expansion of the interface vector into snapshot rows, 3 = both
static int enqueue_solve(rom_fem* f, const FemDev& d, const double* am, int Mc, double* U, long long row,
                         hipStream_t st, size_t lds_back, int stages) {
  rom_ctx* ctx = f->ctx;
  const int kblk = f->nrb * f->ncb;
  static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-column kernel names
  char nm[4][48];
  const bool no_fused = getenv("ROMHC_NO_FUSED") != nullptr;  // (read per call: the tests toggle it)
  const bool fused1 = f->fused1 && !no_fused;  // the whole reduced solve in one wave-per-system kernel
  if (f->nGp > 0 && fused1 && (stages & 1)) {
    ROM_PROF(ctx, "solve1", Mc * (262144 / 3.0 + 3 * 4096.0), Mc * 8.0 * 4096 * 3);
    k_solve1<<<Mc, 64, 0, st>>>(d, am);
  }
  if (f->nGp > 0 && !fused1 && (stages & 1)) {
    {
      ROM_PROF(ctx, "rhs", 0, 8.0 * Mc * f->nGa);
      k_rhs<<<Mc, 256, 0, st>>>(d, am);
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
    {
      ROM_PROF(ctx, "coef", 0, 8.0 * Mc * f->ncoef);
      k_coef<<<Mc, 256, 0, st>>>(d, am);
    }
  }
  if (!(stages & 2)) {
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  if (f->nGp > 0) {
    if (f->nexp > 0) {
      ROM_PROF(ctx, "expand", Mc * 2.0 * f->n1p * 32.0 * f->nexp, 8.0 * Mc * f->n1p * f->nexp);
      k_expand<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->nexp), 256, 0, st>>>(d, am, Mc, U, row);
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
      const double fl_ext = (f->ext_flops - 2.0 * f->n_edges * double(f->n1p) * f->n1p) * Mc / kblk;  // per block (average)
      if (f->n_gen_blocks > 0) {
        dim3 grid(npatch, (Mc + 63) / 64, f->n_gen_blocks);
        ROM_PROF(ctx, "extend", fl_ext * f->n_gen_blocks, 8.0 * Mc * double(f->n_gen_blocks) * nij);
        k_extend<<<grid, 256, 0, st>>>(d, am, Mc, U, row, d.gen_blocks, 4);
      }
      if (f->n_lr_blocks > 0) {
        const bool no128 = getenv("ROMHC_NO_EXT128") != nullptr;
        ROM_PROF(ctx, "extend_lr", fl_ext * f->n_lr_blocks, 8.0 * Mc * double(f->n_lr_blocks) * nij);
        if (f->n1 >= 96 && Mc >= 128 && !no128) {  // wide tiles need enough vertices per mesh row and systems to fill them
          dim3 grid(f->n1 * ((f->n1 + 127) / 128), (Mc + 127) / 128, f->n_lr_blocks);
          k_extend128<<<grid, 256, 0, st>>>(d, am, Mc, U, row);
        } else {
          dim3 grid(f->n1 * ((f->n1 + 63) / 64), (Mc + 63) / 64, f->n_lr_blocks);
          k_extend<<<grid, 256, 0, st>>>(d, am, Mc, U, row, d.lr_blocks, 6);
        }
      }
    }
    if (f->nscat > 0) {
      ROM_PROF(ctx, "scatter_interface", 0, 16.0 * Mc * f->nscat);
      k_scatter_interface<<<dim3((f->nscat + 255) / 256, Mc), 256, 0, st>>>(d, Mc, U, row);
    }
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// enqueue the sweep; `check` = also wait for it and report a non-positive pivot.  `Y` (optional): the caller's
// interface vectors (rows y_row0 .. of stride nGp) instead of the internal workspace; `stages` as in enqueue_solve.
static int solve_batch_impl(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0, bool check, rom_buf* Y = nullptr,
                            int64_t y_row0 = 0, int stages = 3) {
  ROM_CHECK(f && a && (U || !(stages & 2)), "rom_solve_batch: null argument");
  ROM_CHECK(M >= 0 && row0 >= 0 && y_row0 >= 0, "rom_solve_batch: negative M or row offset");
  const int kblk = f->nrb * f->ncb;
  ROM_CHECK(a->n >= size_t(M) * kblk, "rom_solve_batch: `a` holds %zu doubles, need %zu", a->n, size_t(M) * kblk);
  if (stages & 2)
    ROM_CHECK(U->n >= size_t(row0 + M) * f->dim, "rom_solve_batch: U holds %zu doubles, need %zu", U->n,
              size_t(row0 + M) * f->dim);
  if (Y)
    ROM_CHECK(Y->n >= size_t(y_row0 + M) * f->nGp, "rom_solve_batch: the interface-vector buffer holds %zu doubles, need %zu",
              Y->n, size_t(y_row0 + M) * f->nGp);
  if (M == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  ROM_HIP(hipSetDevice(ctx->device));
  // chunk the sweep so that the factor workspace respects the budget
  size_t per_sys = (size_t(f->nslots) * 4096 + size_t(f->T) * 4096 + 2 * size_t(f->nGp)) * sizeof(double);
  int Mc_max = int(std::max<size_t>(1, std::min<size_t>(size_t(M), ctx->ws_limit / std::max<size_t>(per_sys, 1))));
  if (f->ws_M > 0 && f->ws_M < Mc_max && f->ws_M >= 256) Mc_max = f->ws_M;  // reuse what we have
  ROM_TRY(ensure_workspace(f, Mc_max));
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
      if (Y) d.y = Y->p + size_t(y_row0 + m0 + off) * f->nGp;
      else d.y += size_t(off) * f->nGp;
      d.yhat += size_t(off) * f->nGp;
      ROM_TRY(enqueue_solve(f, d, a->p + size_t(m0 + off) * kblk, Mc, U ? U->p : nullptr, (long long)(row0 + m0 + off), st,
                            lds_back, stages));
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
  return check ? rom_solve_status(ctx) : ROM_OK;
}

extern "C" int rom_solve_batch(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0) {
  return solve_batch_impl(f, a, M, U, row0, true);
}

// Same without the host round trip: the sweep is only enqueued on the compute stream; a non-positive pivot
// is remembered on the device until rom_solve_status() is asked.
extern "C" int rom_solve_batch_async(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0) {
  return solve_batch_impl(f, a, M, U, row0, false);
}

extern "C" int rom_fem_expansion_is_linear(rom_fem* f, int* flag) {
  ROM_CHECK(f && flag, "rom_fem_expansion_is_linear: null argument");
  *flag = f->npre == 0 ? 1 : 0;  // no edge is recovered node by node (k_back_pre weights its inputs with the parameters)
  return ROM_OK;
}

extern "C" int rom_fem_reduced_stride(rom_fem* f, int64_t* stride) {
  ROM_CHECK(f && stride, "rom_fem_reduced_stride: null argument");
  *stride = f->nGp;
  return ROM_OK;
}

// Stage 1 only: Y[y_row0 + m, :] = interface vector of system m (reduced unknowns, cross points, the coefficient
// blocks the extension reads): everything that depends on the solve, 1/85 of a snapshot row at 256x256 / 2x2.
extern "C" int rom_solve_reduced_async(rom_fem* f, rom_buf* a, int M, rom_buf* Y, int64_t y_row0) {
  ROM_CHECK(f && Y, "rom_solve_reduced_async: null argument");
  ROM_CHECK(M >= 0 && y_row0 >= 0 && Y->n >= size_t(y_row0 + M) * f->nGp, "rom_solve_reduced_async: Y too small");
  if (M > 0 && f->nGp > 0)  // padding slots are read against zero table entries: they must hold finite numbers
    ROM_HIP(hipMemsetAsync(Y->p + size_t(y_row0) * f->nGp, 0, size_t(M) * f->nGp * sizeof(double), f->ctx->stream));
  return solve_batch_impl(f, a, M, nullptr, 0, false, Y, y_row0, 1);
}

// Stage 2 only: snapshot rows U[row0 + m, :] from the interface vectors Y[y_row0 + m, :] (of this or any other
// rank: the expansion is deterministic, so every rank reproduces the owner's rows bit for bit).
extern "C" int rom_expand_batch_async(rom_fem* f, rom_buf* a, int M, rom_buf* Y, int64_t y_row0, rom_buf* U, int64_t row0) {
  ROM_CHECK(f && Y && U, "rom_expand_batch_async: null argument");
  return solve_batch_impl(f, a, M, U, row0, false, Y, y_row0, 2);
}

extern "C" int rom_solve_status(rom_ctx* ctx) {
  ROM_CHECK(ctx, "rom_solve_status: null context");
  int status = 0;
  ROM_HIP(hipMemcpyAsync(&status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (status != 0) {
    ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    rom_set_error("rom_solve_batch: interface matrix not positive definite (non-positive pivot); "
                  "all block coefficients must be > 0");
    return ROM_ERR_NOT_SPD;
  }
  return ROM_OK;
}
