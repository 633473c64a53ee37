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
#include "rom_fem_dev.h"

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

// Segment-major copy of one extension table for k_extend128: Gs[(seg * n1^2 + row') * 8 + kk] = G[row * ld + 8 seg + kk],
// row = distance-from-the-side * n1 + position-along-it; row' = row for the sides a mesh row runs along (orient 0) and
// the transposed numbering, position * n1 + distance, for the sides it runs away from (orient 1) -- either way the 128
// vertices of a tile are adjacent rows, and one global_load_lds_dwordx4 of the kernel (16 rows x 64 bytes) reads one
// contiguous kilobyte = 8 cache lines instead of pieces of 16 (probe: -6 % at C2, profiles/r02_extend128_kloop_probes.txt)
__global__ void k_repack_table(const double* __restrict__ G, int ld, int nseg, int n1, int orient, double* __restrict__ Gs) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long hrows = (long long)n1 * n1;
  if (idx >= (long long)nseg * hrows * 8) return;
  const int kk = int(idx & 7);
  const long long rp = (idx >> 3) % hrows, seg = (idx >> 3) / hrows;
  const long long row = orient ? (rp % n1) * n1 + rp / n1 : rp;
  const int k = int(seg) * 8 + kk;
  Gs[idx] = k < ld ? G[row * ld + k] : 0.0;
}


__device__ inline double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
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

// This thread's share of the assembled tile (round 4): lane l of wave w holds, for x = 0 .. 7, the two entries
// (row 16 w + 2 x + (l >> 5), columns 2 (l & 31), + 1): v[2 x], v[2 x + 1].  One load instruction of the wave then reads
// two whole rows of a table = ONE contiguous kilobyte (eight cache lines), and the wave's 16 rows take eight of them.
// (Before: a thread owned 16 consecutive columns of one row and read them with eight 16-byte loads; every instruction
// of the wave touched 64 different cache lines for 16 bytes each.  The coalesced share takes the same time -- the assembly,
// 8 % of a C4 step, is a chain of memory round trips with two terms in flight, not line traffic: profiles/r04_tile_cholesky.txt
// -- but leaves k_factor_panel without register spills.)
// The values wait in registers while the MFMA k-loop runs.
struct STile {
  double v[16];
};


// NS systems (consecutive rows of `a`) are assembled from ONE pass over the tables: the assembly costs what it reads
// (profiles/r02_tile_cholesky_probes.txt), and the tables are the same for every system.
// coef: LDS, NS * COEF_MAX doubles of term weights + TERM_DESC_DOUBLES of term descriptors.
// The descriptors of a pass are staged in LDS next to the weights (one coalesced load); every WAVE walks the terms whose
// rectangle meets its band of 16 rows (a uniform walk), fetches only the row pairs inside the rectangle, and the rows of
// the next term are in flight while the current one is added.  Same terms in the same order per entry as ever: the tiles
// are bit-identical (entries outside a term's rectangle are zero in its table: adding them or not is the same).
constexpr int TERM_DESC_DOUBLES = COEF_MAX * 3 / 2;  // 3 ints per term: table, rows lo | hi << 16, columns lo | hi << 16
template <int NS>
__device__ inline void s_tile_load(STile (&st)[NS], const TileDesc& d, const FemDev& f, const double* __restrict__ am0,
                                   int nsys, double* coef) {
  const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6), l = threadIdx.x & 63;
  const int band = 16 * w;                   // this wave's rows: band .. band + 15
  const int rb = band + (l >> 5), cu = 2 * (l & 31);  // the lane's row at x = 0 (row of x: rb + 2 x) and its first column
  int* desc = reinterpret_cast<int*>(coef + NS * COEF_MAX);
#pragma unroll
  for (int q = 0; q < NS; ++q)
#pragma unroll
    for (int x = 0; x < 16; ++x) st[q].v[x] = 0.0;
  for (int tb = d.t0; tb < d.t1; tb += COEF_MAX) {
    const int nt = min(COEF_MAX, d.t1 - tb);
    __syncthreads();
    if (int(threadIdx.x) < nt) {
      const GenTerm g = f.terms[tb + threadIdx.x];
#pragma unroll
      for (int q = 0; q < NS; ++q) coef[q * COEF_MAX + threadIdx.x] = term_coef(g, am0 + size_t(q < nsys ? q : 0) * f.kblk);
      desc[3 * threadIdx.x] = g.tab;
      desc[3 * threadIdx.x + 1] = int(g.r_lo) | int(g.r_hi) << 16;
      desc[3 * threadIdx.x + 2] = int(g.c_lo) | int(g.c_hi) << 16;
    }
    __syncthreads();
    // the wave's next term at or behind t whose rows meet its band (nt: none)
    auto next_term = [&](int t) {
      for (; t < nt; ++t) {
        const int rr = __builtin_amdgcn_readfirstlane(desc[3 * t + 1]);
        if (band < (rr >> 16) && band + 16 > (rr & 0xffff)) break;
      }
      return t;
    };
    // row pairs (band + 2 x, + 1) of term t inside its rectangle: bit x of the mask; their kilobytes are fetched
    auto fetch = [&](int t, double2 (&wv)[8]) -> int {
      const int rr = __builtin_amdgcn_readfirstlane(desc[3 * t + 1]);
      const int lo = rr & 0xffff, hi = rr >> 16;
      const double2* src = reinterpret_cast<const double2*>(f.pool + size_t(__builtin_amdgcn_readfirstlane(desc[3 * t])) * 4096 + rb * 64 + cu);
      int mask = 0;
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (band + 2 * x + 2 > lo && band + 2 * x < hi) {  // (uniform)
          wv[x] = src[x * 64];  // tables are zero outside their rectangle: no lane masks needed
          mask |= 1 << x;
        }
      return mask;
    };
    double2 wa[8], wb[8];
    int ta = next_term(0), ma = 0;
    if (ta < nt) ma = fetch(ta, wa);
    while (ta < nt) {
      const int tn = next_term(ta + 1);
      int mb = 0;
      if (tn < nt) mb = fetch(tn, wb);
      double cf[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) cf[q] = coef[q * COEF_MAX + ta];
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (ma >> x & 1) {
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            st[q].v[2 * x] += cf[q] * wa[x].x;
            st[q].v[2 * x + 1] += cf[q] * wa[x].y;
          }
        }
#pragma unroll
      for (int x = 0; x < 8; ++x) wa[x] = wb[x];
      ta = tn;
      ma = mb;
    }
  }
  if (d.diag) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int r = rb + 2 * x;
      if (r >= d.ndr) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          st[q].v[2 * x] = cu == r ? 1.0 : 0.0;  // padding unknowns: identity
          st[q].v[2 * x + 1] = cu + 1 == r ? 1.0 : 0.0;
        }
      }
    }
  }
  __syncthreads();  // (`coef` may alias memory the caller writes next)
}

// C(LDS tile) = S_tile - acc
__device__ inline void tile_from_acc(double* Cb, const Acc& acc, const STile& st, const WavePos& wp) {
  {
    const int l = threadIdx.x & 63;  // (the share of s_tile_load: rows 16 w + 2 x + (l >> 5), columns 2 (l & 31), + 1)
    double* dst = Cb + (16 * (threadIdx.x >> 6) + (l >> 5)) * LDC + 2 * (l & 31);
#pragma unroll
    for (int x = 0; x < 8; ++x) *reinterpret_cast<double2*>(dst + 2 * x * LDC) = double2{st.v[2 * x], st.v[2 * x + 1]};
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

// The same assembly as a STREAM (round 4): S_tile for NS systems straight into LDS tiles T[q] (64 x LDC).
// s_tile_load walks a tile's ~10 terms with two terms in flight -- a chain of memory round trips that is 8 % of a C4 step
// (profiles/r04_tile_cholesky.txt) and cannot be deepened: the tile itself sits in 32 of the 128 registers four resident
// workgroups per CU allow.  Here the host has flattened, per tile and wave, the (row pair, term) pieces that wave needs
// (FemDev::alist: element offset of the piece in the pool | meta = x + (term << 8) + (last piece of position x) << 16; a piece =
// 8 rows x 16 columns of a table = eight whole cache lines, position x = one of the 2 x 4 such pieces of the wave's 16 rows),
// sorted by position, the terms of a position in their order: a ring of TILE_RING kilobytes is in flight whatever term
// they belong to, a position's sum lives in two registers per system and goes to LDS when its last piece is in, and the
// tile never occupies registers.  Weights: lane t holds the weight of term t (two registers for up to 128 terms), fetched
// with v_readlane.  Every entry adds up the same products in the same order as s_tile_load: same bits.
// Called by all 256 threads; ends with a barrier (the tiles are complete and visible).
constexpr int TILE_RING = 8;
// what the stream needs before its first piece -- the wave's first 64 list entries and the weights of the terms (lane t: terms t
// and 64 + t) -- is fetched AHEAD of the k-loop (s_tile_stream_begin): two dependent memory round trips less behind it
template <int NS>
struct TileStream {
  int e0, ne;
  int2 mine;
  double c0[NS], c1[NS];
};
template <int NS>
__device__ inline void s_tile_stream_begin(TileStream<NS>& ts, int slot, const TileDesc& d, const FemDev& f, const double* __restrict__ am0, int nsys) {
  const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6), l = threadIdx.x & 63;
  ts.e0 = f.aoff[slot * 4 + w];
  ts.ne = f.aoff[slot * 4 + w + 1] - ts.e0;  // this wave's pieces (a multiple of TILE_RING: padded with no-ops)
  ts.mine = (reinterpret_cast<const int2*>(f.alist) + ts.e0)[l];  // lane i: piece i of the first group of 64 (the list carries 128 no-ops behind its end)
  const int nt = d.t1 - d.t0;
  // (a fill-only tile has no direct terms: nt = 0 -- the index stays inside the list and the coefficients are zero)
  const GenTerm g0 = f.terms[d.t0 + max(min(l, nt - 1), 0)], g1 = f.terms[d.t0 + max(min(64 + l, nt - 1), 0)];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const double* am = am0 + size_t(q < nsys ? q : 0) * f.kblk;
    ts.c0[q] = l < nt ? term_coef(g0, am) : 0.0;
    ts.c1[q] = 64 + l < nt ? term_coef(g1, am) : 0.0;
  }
}
template <int NS>
__device__ inline void s_tile_to_lds(double* const (&T)[NS], const TileStream<NS>& ts, const TileDesc& d, const FemDev& f) {
  const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6), l = threadIdx.x & 63;
  const int band = 16 * w, rsub = l >> 3, csub = 2 * (l & 7);  // piece x = 4 pr + cs: rows band + 8 pr + rsub, columns 16 cs + csub, + 1
  const int ne = ts.ne;
  const int2* ent = reinterpret_cast<const int2*>(f.alist) + ts.e0;
  const double2* pool2 = reinterpret_cast<const double2*>(f.pool) + rsub * 32 + (l & 7);  // (a piece: 8 rows x 16 columns = eight whole cache lines)
  double2 v[TILE_RING];
  int2 mine = ts.mine;
  auto issue = [&](int2 e_lane, int i, double2& dst) { dst = pool2[__builtin_amdgcn_readlane(e_lane.x, i) >> 1]; };
#pragma unroll
  for (int u = 0; u < TILE_RING; ++u) issue(mine, u, v[u]);
  double c0[NS], c1[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { c0[q] = ts.c0[q]; c1[q] = ts.c1[q]; }
  // positions that no term touches stay zero
#pragma unroll
  for (int q = 0; q < NS; ++q)
    for (int i = threadIdx.x; i < 64 * LDC / 2; i += 256) reinterpret_cast<double2*>(T[q])[i] = double2{0.0, 0.0};
  __syncthreads();
  double2 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) acc[q] = double2{0.0, 0.0};
  for (int base = 0; base < ne; base += 64) {
    const int2 cur = mine;
    const int2 nxt = ent[base + 64 + l];
    const int cnt = min(64, ne - base);
#pragma unroll
    for (int i0 = 0; i0 < 64; i0 += TILE_RING) {  // (unrolled: straight-line code keeps the compiler's wait counts exact)
      if (i0 >= cnt) break;
#pragma unroll
      for (int u = 0; u < TILE_RING; ++u) {
        const int i = i0 + u;
        const int meta = __builtin_amdgcn_readlane(cur.y, i);  // x | term << 8 | (last piece of row pair x) << 16 | no-op << 17
        const int t = (meta >> 8) & 0xff;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double c = t < 64 ? readlane_f64(c0[q], t) : readlane_f64(c1[q], t - 64);
          if (meta >> 17) c = 0.0;
          acc[q].x = __builtin_fma(c, v[u].x, acc[q].x);
          acc[q].y = __builtin_fma(c, v[u].y, acc[q].y);
        }
        if ((meta >> 16) & 1) {  // the pieces are sorted by row pair: its sum is complete
          const int x = meta & 0xff, row = band + 8 * (x >> 2) + rsub, col = 16 * (x & 3) + csub;
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            *reinterpret_cast<double2*>(T[q] + row * LDC + col) = acc[q];
            acc[q] = double2{0.0, 0.0};
          }
        }
        if (i + TILE_RING < 64) issue(cur, i + TILE_RING, v[u]);
        else issue(nxt, i + TILE_RING - 64, v[u]);
      }
    }
    mine = nxt;
  }
  if (d.diag) {  // padding unknowns: identity
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int r = band + 8 * (x >> 2) + rsub, c = 16 * (x & 3) + csub;
      if (r >= d.ndr) {
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<double2*>(T[q] + r * LDC + c) = double2{c == r ? 1.0 : 0.0, c + 1 == r ? 1.0 : 0.0};
      }
    }
  }
  __syncthreads();
}

// T (LDS tile holding S) -= acc
// The lower triangle of a symmetric 64 x 64 tile as ten 16 x 16 blocks dealt to the four waves 3 / 3 / 2 / 2 (k_diag_update: with
// the quadrants of `Acc` the wave of the upper-right quadrant idles and two others compute an upper block nobody reads -- 4 / 4 / 4 / 0
// MFMAs per k-step; the k-loop is two thirds of that kernel and runs at the rate of the matrix pipe, in-kernel stamps,
// profiles/r05_tile_cholesky_probes.txt).  Wave w, pair p -> block (row block, column block):
//   w = 0: (0,0) (1,0) (1,1)    w = 1: (2,0) (2,1) (2,2)    w = 2: (3,0) (3,1)    w = 3: (3,2) (3,3)
// (a wave with two pairs repeats its second one as pair 2: addresses stay valid, the MFMA is skipped)
// `ws`: the wave's place in that deal = its index.  (Reversed in every other group of 256 workgroups -- workgroups that share a
// CU are mostly 256 apart in launch order, a SIMD would then hold a 3-block and a 2-block wave instead of two of a kind --
// measured: no difference at C4, slower at C5; profiles/r05_tile_cholesky_probes.txt)
struct SymAcc {
  d4_t c[3];
  int ws;
};
__device__ inline int sym_pairs(int w) { return w < 2 ? 3 : 2; }
__device__ inline int sym_ib(int w, int p) { return w == 0 ? (p == 0 ? 0 : 1) : w == 1 ? 2 : 3; }
__device__ inline int sym_jb(int w, int p) { return w == 0 ? (p == 2 ? 1 : 0) : w == 1 ? p : w == 2 ? min(p, 1) : 2 + min(p, 1); }
__device__ inline void sym_acc_zero(SymAcc& a) {
#pragma unroll
  for (int p = 0; p < 3; ++p) a.c[p] = d4_t{0.0, 0.0, 0.0, 0.0};
  a.ws = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}
// Cb (lower blocks only) -= acc; the blocks above the diagonal keep what the assembly left there: nobody reads them
__device__ inline void tile_minus_acc_sym(double* Cb, const SymAcc& acc, const WavePos& wp) {
  const int w = acc.ws;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    if (p >= sym_pairs(w)) break;
    const int r0 = 16 * sym_ib(w, p) + (wp.lane >> 4), c = 16 * sym_jb(w, p) + (wp.lane & 15);
#pragma unroll
    for (int g = 0; g < 4; ++g) Cb[(r0 + 4 * g) * LDC + c] -= acc.c[p][g];
  }
  __syncthreads();
}
__device__ inline void tile_minus_acc(double* Cb, const Acc& acc, const WavePos& wp) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) Cb[acc_row(wp, i, g) * LDC + acc_col(wp, j)] -= acc.c[i][j][g];
  __syncthreads();
}

// LDS carve-up of the factor kernels: B staging (2 buffers) | union { A staging (2 buffers), C tile }
constexpr int LDK8 = 10;                                        // row stride of a staged [64][8] chunk (16-byte aligned rows)
constexpr int FACT_LDS_DOUBLES = TILE_DOUBLES + 64 * LDK8;      // C tile | one 8-wide chunk of invL: 4864 doubles = 38,912 B -> 4 WG / CU

// acc += sum over the k-list of `slot` of L[slotA] * L[slotB]^T
// a pointer that is the same in every lane, in scalar registers whatever the compiler thinks of it
__device__ inline const char* x128_uniform(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const char*>((unsigned long long)hi << 32 | lo);
}
__device__ inline unsigned long long x128_uniform(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return (unsigned long long)hi << 32 | lo;
}

// The operand chunks go from global memory straight into LDS (global_load_lds_dwordx4, the main loop of k_gram128 /
// k_extend128 at 64 x 64): no staging registers, no ds_write (the register-staged loop of round 1 gave the same bits;
// equal at C4, 2-3 % slower at C5).  A chunk = 16 k of both tiles =
// {A 64 rows x 128 B | B 64 rows x 128 B}; one DMA instruction writes 64 lanes x 16 B back to back = 8 rows, the eight
// 16-byte units of a row stored at position u ^ ((row >> 1) & 7) (conflict-free fragment reads without padding); wave w
// fetches rows 16 w .. 16 w + 15 of both operands (4 instructions per chunk); one barrier per chunk.  `lds`: TD_LDS_BYTES;
// ends with a barrier (the area may be reused right after).  (Rounds 2-3: three slots, chunks ch + 1 and ch + 2 in flight
// behind a counted s_waitcnt vmcnt(4).)
// Round 4: TWO slots (chunk ch + 1 in flight under chunk ch, plain vmcnt(0) in front of the barrier).  The third slot bought
// its deeper prefetch with 16 KB of LDS per workgroup: with two, a diagonal-update workgroup needs 33.8 KB and a panel
// workgroup 39.4 KB -- FOUR of them per CU instead of three, and these kernels live on resident workgroups
// (profiles/r03_overlap_and_chunk_probes.txt: -19 % from three to two).
constexpr int TD_SLOT = 2 * 64 * 128;     // bytes
constexpr int TD_NSLOT = 2;
constexpr int TD_LDS_BYTES = TD_NSLOT * TD_SLOT;  // 32,768
// SYM (AccT = SymAcc): the product is a symmetric tile (both operands the same tile pairs), the waves own the lower blocks
// as dealt by sym_ib / sym_jb; every block sees the same MFMAs in the same order as in the quadrant form: same bits.
template <bool SYM = false, class FP, class AccT>
__device__ inline void accumulate_klist_dma(const FemDev& f, int slot, const double* Lm, FP active, AccT& acc, char* lds,
                                            const WavePos& wp) {
  const int e0 = f.kptr[slot], np = f.kptr[slot + 1] - e0;
  const int n = 4 * np;  // chunks
  if (n <= 0) return;
  const int lane = wp.lane, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds));
  // fragment addressing: lane (fr, kq) reads row fr (+ 16 i) at k = 4 kki + kq: unit (2 kki + (kq >> 1)) ^ (fr >> 1)
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kki = 0; kki < 4; ++kki) {
    const unsigned uo = unsigned((((2 * kki) ^ (kq >> 1) ^ (fr >> 1)) << 4) + (kq & 1) * 8);
    fa[kki] = unsigned((wp.wr * 32 + fr) * 128) + uo;
    fb[kki] = unsigned(8192 + (wp.wc * 32 + fr) * 128) + uo;
  }
  // SYM: block row ib of A at ib * 2048, block column jb of B at 8192 + jb * 2048, + the lane's (fr * 128 + unit) below
  unsigned ua[4], offA[3], offB[3];
  int npairs = 0;
#pragma unroll
  for (int kki = 0; kki < 4; ++kki) ua[kki] = unsigned(fr * 128) + unsigned((((2 * kki) ^ (kq >> 1) ^ (fr >> 1)) << 4) + (kq & 1) * 8);
  if constexpr (SYM) {
    npairs = sym_pairs(acc.ws);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      offA[q] = unsigned(sym_ib(acc.ws, q) * 2048);
      offB[q] = unsigned(8192 + sym_jb(acc.ws, q) * 2048);
    }
  }
  // DMA addressing: lane -> row 16 w + 8 q + (lane >> 3) of the tile (512-byte rows), stored unit lane & 7 = logical
  // unit (lane & 7) ^ ((4 q + (lane >> 4)) & 7)
  unsigned vo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    vo[q] = unsigned((16 * w + 8 * q + (lane >> 3)) * 512) + unsigned(((lane & 7) ^ ((4 * q + (lane >> 4)) & 7)) * 16);
  // the k-list: lane l holds pair l of the current batch of 64
  int ida = 0, idb = 0;
  auto load_pairs = [&](int p0) {
    const int p = min(p0 + lane, np - 1);
    ida = f.kpair[2 * (e0 + p)];
    idb = f.kpair[2 * (e0 + p) + 1];
  };
  load_pairs(0);
  const char* const Lb = reinterpret_cast<const char*>(Lm);
  auto dma = [](unsigned lds_addr, const char* base, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                 "s"(x128_uniform(base))
                 : "memory", "m0");
  };
  auto issue = [&](int ch, unsigned slot_off) {
    const int p = ch >> 2, c = ch & 3;
    if (c == 0 && (p & 63) == 0 && p > 0) load_pairs(p);
    const int ia = __builtin_amdgcn_readlane(ida, p & 63), ib = __builtin_amdgcn_readlane(idb, p & 63);
    const char* ba = Lb + size_t(ia) * 32768 + c * 128;
    const char* bb = Lb + size_t(ib) * 32768 + c * 128;
    const unsigned sl = lds0 + slot_off + unsigned(w) * 2048;
    dma(sl, ba, vo[0]);
    dma(sl + 1024, ba, vo[1]);
    dma(sl + 8192, bb, vo[0]);
    dma(sl + 8192 + 1024, bb, vo[1]);
  };
  unsigned s_cur = 0, s_nxt = TD_SLOT;  // slots of chunks ch, ch + 1
  issue(0, s_cur);
  for (int ch = 0; ch < n; ++ch) {
    // chunk ch has landed for everybody, and everybody has left the slot of chunk ch - 1 (= that of chunk ch + 1)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (ch + 1 < n) issue(ch + 1, s_nxt);
    if constexpr (SYM) {
      if (active(ch)) {
        const char* sp = lds + s_cur;
        double af[2][3], bf[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          af[0][q] = *reinterpret_cast<const double*>(sp + offA[q] + ua[0]);
          bf[0][q] = *reinterpret_cast<const double*>(sp + offB[q] + ua[0]);
        }
#pragma unroll
        for (int kki = 0; kki < 4; ++kki) {
          const int pb = kki & 1;
          if (kki < 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              af[pb ^ 1][q] = *reinterpret_cast<const double*>(sp + offA[q] + ua[kki + 1]);
              bf[pb ^ 1][q] = *reinterpret_cast<const double*>(sp + offB[q] + ua[kki + 1]);
            }
          }
          acc.c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][0], bf[pb][0], acc.c[0], 0, 0, 0);
          acc.c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][1], bf[pb][1], acc.c[1], 0, 0, 0);
          if (npairs > 2) acc.c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][2], bf[pb][2], acc.c[2], 0, 0, 0);
        }
      }
    } else if (active(ch)) {
      const char* sp = lds + s_cur;
      double af[2][2], bf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[0][i] = *reinterpret_cast<const double*>(sp + fa[0] + i * 2048);
        bf[0][i] = *reinterpret_cast<const double*>(sp + fb[0] + i * 2048);
      }
#pragma unroll
      for (int kki = 0; kki < 4; ++kki) {
        const int pb = kki & 1;
        if (kki < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[pb ^ 1][i] = *reinterpret_cast<const double*>(sp + fa[kki + 1] + i * 2048);
            bf[pb ^ 1][i] = *reinterpret_cast<const double*>(sp + fb[kki + 1] + i * 2048);
          }
        }
        acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][0], bf[pb][0], acc.c[0][0], 0, 0, 0);
        acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][0], bf[pb][1], acc.c[0][1], 0, 0, 0);
        acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][1], bf[pb][0], acc.c[1][0], 0, 0, 0);
        acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][1], bf[pb][1], acc.c[1][1], 0, 0, 0);
      }
    }
    const unsigned t = s_cur;
    s_cur = s_nxt;
    s_nxt = t;
  }
  __syncthreads();
}

// ============================================================================================
// factorisation kernels
// ============================================================================================
// rhs of the reduced system: the parameter-independent part plus the contributions of the closed-form
// edges.  One workgroup per system; the terms overlap and are summed in their order, entry by entry: a thread owns
// its entries of y and walks the terms that cover them (one store per entry, no barrier and no read-modify-write
// of global memory per term: 23 -> 13 us per 1024 systems at C4, 45 -> 22 at C5).
__global__ __launch_bounds__(256) void k_rhs(FemDev f, const double* __restrict__ a) {
  __shared__ double coef[256];
  const int m = blockIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double* y = f.y + size_t(m) * f.nGp;
  double acc[4];  // entries threadIdx.x + 256 q, q < 4; interfaces of more than 1024 unknowns: the loop at the end
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = threadIdx.x + 256 * q;
    acc[q] = v < f.nGa ? f.g[v] : 0.0;
  }
  for (int t0 = 0; t0 < f.nrhs; t0 += 256) {
    const int nt = min(256, f.nrhs - t0);
    __syncthreads();
    if (int(threadIdx.x) < nt) {
      const RhsTerm& rt = f.rhs[t0 + threadIdx.x];
      coef[threadIdx.x] = rt.kind == 0 ? am[rt.b0] / (am[rt.e0] + am[rt.e1]) : 0.5;
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const RhsTerm& rt = f.rhs[t0 + t];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = threadIdx.x + 256 * q - rt.pos;
        if (i >= 0 && i < rt.len) acc[q] += coef[t] * f.vec[rt.voff + i];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int v = threadIdx.x + 256 * q;
    if (v < f.nGa) y[v] = acc[q];
  }
  for (int v = threadIdx.x + 1024; v < f.nGa; v += 256) {
    double s = f.g[v];
    for (int t = 0; t < f.nrhs; ++t) {
      const RhsTerm& rt = f.rhs[t];
      const int i = v - rt.pos;
      if (i >= 0 && i < rt.len) s += (rt.kind == 0 ? am[rt.b0] / (am[rt.e0] + am[rt.e1]) : 0.5) * f.vec[rt.voff + i];
    }
    y[v] = s;
  }
}

// Coefficient blocks read by the extension and the expansion, and the nodal copy of the cross points.
//   active edge f:      [z_f, 1/s_f, 0...]
//   closed-form edge e: [c_e / s_e, 1/s_e, 0...],  s_e K u_e = g_e + W_e c_e,
//                       c_e = sum_u a_u (M_eu z_u + m_eu / s_u) + (s_e/2) sum_x W_e[node_x,:]^T u_x
// one workgroup per system.  Round 4: the dot products of the closed-form blocks -- (entry k, term t): a column of a
// coefficient matrix against the neighbour's reduced unknowns -- are spread over the workgroup as TASKS (the host lists
// them, FemDev::ctask, ordered so that neighbouring threads read neighbouring matrix entries) and parked in LDS; an entry's
// thread then adds its terms up in their order.  Before, the thread of entry k walked its ~6 terms one after the other,
// each a chain of memory round trips, while three quarters of the workgroup had nothing to do: 44 us at C4 whatever the
// traffic (four systems sharing the matrix reads: 41 us).  Same dot products, same sums: same bits.
__global__ __launch_bounds__(1024) void k_coef(FemDev f, const double* __restrict__ a) {
  const int m = blockIdx.x;
  const double* am = a + size_t(m) * f.kblk;
  double* y = f.y + size_t(m) * f.nGp;
  extern __shared__ double ys[];  // the nGa reduced unknowns of the system (what the coefficient blocks are built from) | the tasks' dot products
  double* dots = f.gdots ? f.gdots + size_t(m) * f.ncf * 8 : ys + f.nGa;   // (global only where ncf * 64 bytes outgrow the LDS)
  for (int v = threadIdx.x; v < f.nGa; v += blockDim.x) ys[v] = y[v];
  __syncthreads();
  for (int x = threadIdx.x; x < f.ncross; x += blockDim.x) y[f.xb0 + x] = ys[f.xred[x]];
  for (int i = threadIdx.x; i < f.nsc; i += blockDim.x) {
    const int b0 = f.scb[2 * i], b1 = f.scb[2 * i + 1];
    y[f.spos0 + i] = b1 >= 0 ? 1.0 / (am[b0] + am[b1]) : (1.0 / (double(f.N) * double(f.N))) / am[b0];
  }
  for (int task = threadIdx.x; task < f.nctask; task += blockDim.x) {
    // a flat record per task (no descriptor to chase): {matrix offset of entry k, source, length, row stride | vector entry or -1, u0, u1, slot}
    const int4 t0 = reinterpret_cast<const int4*>(f.ctask)[2 * task], t1 = reinterpret_cast<const int4*>(f.ctask)[2 * task + 1];
    const int len = t0.z, ldm = t0.w;
    const double* Mt = f.cm + t0.x;
    const double* src = ys + t0.y;
    double dot = 0.0;
#pragma unroll 16
    for (int j = 0; j < len; ++j) dot += Mt[size_t(j) * ldm] * src[j];
    if (t1.x >= 0) dot += f.vec[t1.x] / (am[t1.y] + am[t1.z]);
    dots[t1.w] = dot;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < f.ncoef; it += blockDim.x) {
    const CoefGroup& cg = f.groups[f.item_group[it]];
    const int k = f.item_k[it];
    const double s = am[cg.b0] + am[cg.b1];
    double out = 0.0;
    if (k == cg.r) {
      out = 1.0 / s;
    } else if (k < cg.r) {
      if (cg.kind == 0) {
        out = ys[cg.zpos + k];
      } else {
        const double* dk = dots + size_t(f.item_cf[it]) * 8;  // this entry's dot products, one per term
        double acc = 0.0;
        for (int t = 0; t < cg.nterm; ++t) acc += (cg.t[t].blk >= 0 ? am[cg.t[t].blk] : s / 2) * dk[t];
        out = acc / s;
      }
    }
    y[cg.cpos + k] = out;
  }
}

__global__ __launch_bounds__(256) void k_expand(FemDev f, const double* __restrict__ a, int Mc, double* __restrict__ U,
                                                long long row0) {
  __shared__ __align__(16) double lds[STAGE_TOTAL];
  expand_tile(f, Mc, U, row0, lds, blockIdx.x, blockIdx.y, blockIdx.z);
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
template <int NS>
__device__ inline void diag_update_body(const FemDev& f, const double* __restrict__ am0, int m0, int nsys, int slot, double* lds,
                                        double* coef) {
  const WavePos wp;
  const TileDesc& d = f.desc[slot];
  const bool lower = !(wp.wr == 0 && wp.wc == 1);
  if (f.tile_stream && d.t1 - d.t0 <= 128) {
    // the k-loops first (their DMA slots alias the first tile), then ONE pass over the tables assembles S for all NS systems
    // straight into their LDS tiles, minus the accumulators, out
    TileStream<NS> ts;
    s_tile_stream_begin<NS>(ts, slot, d, f, am0, nsys);
    SymAcc acc[NS];  // (the ten lower blocks over the four waves: no idle quadrant)
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      sym_acc_zero(acc[q]);
      if (q >= nsys) continue;
      if (q > 0) __syncthreads();  // (the previous k-loop's slots)
      accumulate_klist_dma<true>(f, slot, f.L + size_t(m0 + q) * f.nslots * 4096, [](int) { return true; }, acc[q], reinterpret_cast<char*>(lds), wp);
    }
    __syncthreads();
    double* T[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) T[q] = lds + q * TILE_DOUBLES;
    s_tile_to_lds<NS>(T, ts, d, f);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (q >= nsys) break;
      tile_minus_acc_sym(T[q], acc[q], wp);
      double* Lout = f.L + (size_t(m0 + q) * f.nslots + slot) * 4096;
      for (int idx = threadIdx.x; idx < 4096; idx += 256) Lout[idx] = T[q][(idx >> 6) * LDC + (idx & 63)];
    }
    return;
  }
  double* Cb = lds;  // the C tile aliases the DMA slots (used after the k-loop only)
  STile st[NS];
  s_tile_load<NS>(st, d, f, am0, nsys, coef);  // (the values wait in registers)
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (q >= nsys) break;
    double* Lm = f.L + size_t(m0 + q) * f.nslots * 4096;
    Acc acc;
    acc_zero(acc);
    if (q > 0) __syncthreads();  // (the copy-out of the previous system reads the tile)
    accumulate_klist_dma(f, slot, Lm, [&](int) { return lower; }, acc, reinterpret_cast<char*>(lds), wp);
    tile_from_acc(Cb, acc, st[q], wp);
    double* Lout = Lm + size_t(slot) * 4096;
    for (int idx = threadIdx.x; idx < 4096; idx += 256) Lout[idx] = Cb[(idx >> 6) * LDC + (idx & 63)];
  }
}
// NS = 2: two systems per workgroup (one pass over the term tables for both)
template <int NS>
__global__ __launch_bounds__(256, NS == 1 ? 4 : 2) void k_diag_update(FemDev f, const double* __restrict__ a, int slot, int Mc) {
  __shared__ __align__(16) double lds[NS * TILE_DOUBLES];  // 33.8 KB per system (the two DMA slots alias the first tile): NS = 1: four workgroups per CU
  __shared__ double coef[NS * COEF_MAX + TERM_DESC_DOUBLES];
  static_assert(TD_LDS_BYTES <= TILE_DOUBLES * 8, "the DMA slots fit under the tile");
  const int m0 = blockIdx.x * NS;
  diag_update_body<NS>(f, a + size_t(m0) * f.kblk, m0, min(NS, Mc - m0), slot, lds, coef);
}
template __global__ void k_diag_update<1>(FemDev, const double*, int, int);
template __global__ void k_diag_update<2>(FemDev, const double*, int, int);

// 1/sqrt(d) for a positive normal d: hardware seed (v_rsq_f64, ~2^-26 relative error) + two Newton
// steps -> within 1-2 ulp; a fraction of the dependent-instruction chain of 1.0 / sqrt(d).
__device__ inline double rsqrt_newton(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * fma(-hd * y, y, 1.5);
  y = y * fma(-hd * y, y, 1.5);
  return y;
}


// k_solve1 keeps its factor packed by rows in whole panels of four columns: rows 4 g .. 4 g + 3 have 4 (g + 1) entries each
__device__ __host__ constexpr int s1_lrow(int r) { return 8 * (r >> 2) * ((r >> 2) + 1) + (r & 3) * 4 * ((r >> 2) + 1); }

// One 64 x 4 panel of the blocked Cholesky of k_solve1 / k_diag_factor, in row-per-lane form (row r of the panel at
// Pn[4 r]; c0 = its first column): on return v = row `lane` of L in these columns, y / myrs carry the forward substitution.
// Round 4: every lane factorises the 4 x 4 diagonal block for itself (LDS broadcast reads of rows c0 .. c0 + 3, fetched
// beside its own row) instead of receiving pivots and multipliers through v_readlane one column at a time -- 14
// cross-lane broadcasts per panel, each at the end of a dependent chain (31 k of a wave's 82 k cycles), become straight
// VALU code.  Same operations on the same numbers in the same order as the broadcast form: same bits.
// A non-positive (or NaN) pivot shows in its 1 / sqrt: lane j's myrs = 1 / L_jj is then not a finite positive number, and
// every later one is NaN.  (Tested once, behind the factorisation: an or-chain over the 64 pivots kept them all alive.)
__device__ inline bool chol_pivots_bad(double myrs) {
  return __builtin_amdgcn_ballot_w64(!(myrs > 0.0 && myrs < 1.0e300)) != 0;
}
// (the arithmetic; a00 .. a33: the lower triangle of the panel's 4 x 4 diagonal block, the same numbers in every lane)
__device__ inline void chol_panel_core(int lane, int c0, double (&v)[4], double& y, double& myrs, double a00, double a10,
                                       double a11, double a20, double a21, double a22, double a30, double a31, double a32, double a33) {
  const double y0 = readlane_f64(y, c0);
  double y1 = readlane_f64(y, c0 + 1), y2 = readlane_f64(y, c0 + 2), y3 = readlane_f64(y, c0 + 3);
  // column c0
  const double rs0 = rsqrt_newton(a00);
  const double l10 = a10 * rs0, l20 = a20 * rs0, l30 = a30 * rs0;
  a11 -= l10 * l10; a21 -= l20 * l10; a31 -= l30 * l10;
  a22 -= l20 * l20; a32 -= l30 * l20; a33 -= l30 * l30;
  const double yd0 = y0 * rs0;
  y1 -= l10 * yd0; y2 -= l20 * yd0; y3 -= l30 * yd0;
  {
    const double l = lane >= c0 ? v[0] * rs0 : 0.0;  // L[lane][c0]
    v[0] = l;
    if (lane == c0) {
      y = yd0;
      myrs = rs0;
    } else if (lane > c0) {
      y -= l * yd0;
    }
    v[1] -= l * l10; v[2] -= l * l20; v[3] -= l * l30;
  }
  // column c0 + 1
  const double rs1 = rsqrt_newton(a11);
  const double l21 = a21 * rs1, l31 = a31 * rs1;
  a22 -= l21 * l21; a32 -= l31 * l21; a33 -= l31 * l31;
  const double yd1 = y1 * rs1;
  y2 -= l21 * yd1; y3 -= l31 * yd1;
  {
    const double l = lane >= c0 + 1 ? v[1] * rs1 : 0.0;
    v[1] = l;
    if (lane == c0 + 1) {
      y = yd1;
      myrs = rs1;
    } else if (lane > c0 + 1) {
      y -= l * yd1;
    }
    v[2] -= l * l21; v[3] -= l * l31;
  }
  // column c0 + 2
  const double rs2 = rsqrt_newton(a22);
  const double l32 = a32 * rs2;
  a33 -= l32 * l32;
  const double yd2 = y2 * rs2;
  y3 -= l32 * yd2;
  {
    const double l = lane >= c0 + 2 ? v[2] * rs2 : 0.0;
    v[2] = l;
    if (lane == c0 + 2) {
      y = yd2;
      myrs = rs2;
    } else if (lane > c0 + 2) {
      y -= l * yd2;
    }
    v[3] -= l * l32;
  }
  // column c0 + 3
  const double rs3 = rsqrt_newton(a33);
  const double yd3 = y3 * rs3;
  {
    const double l = lane >= c0 + 3 ? v[3] * rs3 : 0.0;
    v[3] = l;
    if (lane == c0 + 3) {
      y = yd3;
      myrs = rs3;
    } else if (lane > c0 + 3) {
      y -= l * yd3;
    }
  }
}
__device__ inline void chol_panel_rows(const double* Pn, int lane, int c0, double (&v)[4], double& y, double& myrs) {
  {
    const double2 v01 = *reinterpret_cast<const double2*>(&Pn[lane * 4]);
    const double2 v23 = *reinterpret_cast<const double2*>(&Pn[lane * 4 + 2]);
    v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
  }
  const double a00 = Pn[c0 * 4];
  const double2 r1 = *reinterpret_cast<const double2*>(&Pn[(c0 + 1) * 4]);
  const double2 r2 = *reinterpret_cast<const double2*>(&Pn[(c0 + 2) * 4]);
  const double a22 = Pn[(c0 + 2) * 4 + 2];
  const double2 r3a = *reinterpret_cast<const double2*>(&Pn[(c0 + 3) * 4]);
  const double2 r3b = *reinterpret_cast<const double2*>(&Pn[(c0 + 3) * 4 + 2]);
  chol_panel_core(lane, c0, v, y, myrs, a00, r1.x, r1.y, r2.x, r2.y, a22, r3a.x, r3a.y, r3b.x, r3b.y);
}
// the same with the panel already in registers (v: row `lane`); the diagonal block comes from its four lanes
__device__ inline void chol_panel_regs(int lane, int c0, double (&v)[4], double& y, double& myrs) {
  chol_panel_core(lane, c0, v, y, myrs, readlane_f64(v[0], c0), readlane_f64(v[0], c0 + 1), readlane_f64(v[1], c0 + 1),
                  readlane_f64(v[0], c0 + 2), readlane_f64(v[1], c0 + 2), readlane_f64(v[2], c0 + 2), readlane_f64(v[0], c0 + 3),
                  readlane_f64(v[1], c0 + 3), readlane_f64(v[2], c0 + 3), readlane_f64(v[3], c0 + 3));
}

// A [64][LDC] tile from LDS to a [64][64] tile in HBM, a row (512 B) per store instruction.  In-kernel stamps
// (tools/dev/gpu_diagf_stamps.py) put each of k_diag_factor's two tile stores at 5-6 k of its cycles: all systems store at the
// same time, 33 MB in ~2.5 us.  Sixteen LDS reads in flight before the stores: the same; two rows per instruction
// (16 bytes per lane): 21 k cycles -- four times slower; the rows of L stored panel by panel while the factorisation runs:
// the same cycles move into the panels (a store instruction holds the wave's issue ~80 cycles wherever it stands).
// The plain loop it is (profiles/r05_tile_cholesky_probes.txt).
__device__ inline void tile_rows_to_global(double* __restrict__ dst, const double* Ls, int lane) {
  for (int i = 0; i < 64; ++i) dst[i * 64 + lane] = Ls[i * LDC + lane];
}

// Diagonal tile j, steps 2 and 3 in one kernel on the matrix cores (one wave per system, LDS 36 KB: all 1024 systems
// of a step resident at once):
//   rank-4 blocked right-looking Cholesky of the tile held in MFMA accumulator layout (the scheme of k_solve1: a
//   64 x 4 panel goes through LDS into row-per-lane form, is factorised with readlane broadcasts while the forward
//   substitution y_j <- L_jj^-1 y_j rides along, and returns as A and B operand of the rank-4 trailing update);
//   then X = L_jj^-1 in place, blocked 16 x 16: the four diagonal blocks by the column sweep of an unblocked
//   in-place triangular inverse (lane = (block, row), its row of X in registers, the untouched columns of L
//   broadcast from LDS), the six blocks below them as  X_ij = -sum_{k=j+1..i} X_ik (L_kj X_jj)  on MFMA.
// (A column-by-column version on the vector pipe, two kernels, took 2 x 32 us per 1024 systems against 33 us.)
// (one wave; Ls: 64 * LDC, Pn: 256, rinv: 64 doubles of LDS; the only synchronisation is between the lanes of the wave)
// (dev build -DROMHC_DIAGF_STAMPS: cycle stamps of the kernel's phases, printed by two of the waves -- tools/dev/gpu_diagf_stamps.py)
#ifdef ROMHC_DIAGF_STAMPS
#define DF_STAMP(i) df_stamp[i] = __builtin_readcyclecounter()
#else
#define DF_STAMP(i)
#endif
__device__ inline void diag_factor_body(const FemDev& f, int m, int slot, int j, double* Ls, double* Pn, double* rinv) {
#ifdef ROMHC_DIAGF_STAMPS
  unsigned long long df_stamp[8];
#endif
  DF_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15, l4 = lane >> 4;
  double* Lt = f.L + (size_t(m) * f.nslots + slot) * 4096;
  // the lower blocks of the (symmetric) tile in accumulator layout; only its lower triangle is valid in memory
  d4_t C[4][4];
#pragma unroll
  for (int ib = 0; ib < 4; ++ib)
#pragma unroll
    for (int jb = 0; jb <= ib; ++jb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int R = 16 * ib + 4 * g + l4, Cc = 16 * jb + l16;
        C[ib][jb][g] = Lt[max(R, Cc) * 64 + min(R, Cc)];
      }
  double y = f.y[size_t(m) * f.nGp + j * 64 + lane];
  double myrs = 0.0;
#ifdef ROMHC_DIAGF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  DF_STAMP(1);
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int jb = p >> 2, co = 4 * (p & 3), c0 = 4 * p;
    if ((l16 >> 2) == (p & 3)) {
#pragma unroll
      for (int ib = jb; ib < 4; ++ib)
#pragma unroll
        for (int g = 0; g < 4; ++g) Pn[(16 * ib + 4 * g + l4) * 4 + (l16 - co)] = C[ib][jb][g];
    }
    __builtin_amdgcn_wave_barrier();
    double v[4];
    chol_panel_rows(Pn, lane, c0, v, y, myrs);
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<double2*>(&Pn[lane * 4]) = double2{v[0], v[1]};
    *reinterpret_cast<double2*>(&Pn[lane * 4 + 2]) = double2{v[2], v[3]};
    *reinterpret_cast<double2*>(&Ls[lane * LDC + c0]) = double2{v[0], v[1]};
    *reinterpret_cast<double2*>(&Ls[lane * LDC + c0 + 2]) = double2{v[2], v[3]};
    __builtin_amdgcn_wave_barrier();
    if (p < 15) {
      double frag[4];
#pragma unroll
      for (int x = jb; x < 4; ++x) frag[x] = Pn[(16 * x + l16) * 4 + l4];
#pragma unroll
      for (int jb2 = jb; jb2 < 4; ++jb2)
#pragma unroll
        for (int ib = jb2; ib < 4; ++ib)
          C[ib][jb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag[ib], frag[jb2], C[ib][jb2], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  DF_STAMP(2);
  if (chol_pivots_bad(myrs) && lane == 0) atomicOr(f.status, 1);
  f.y[size_t(m) * f.nGp + j * 64 + lane] = y;
  rinv[lane] = myrs;  // 1 / L[lane][lane]
  __builtin_amdgcn_wave_barrier();
  // L (lower, zero above the diagonal) to HBM, coalesced
  tile_rows_to_global(Lt, Ls, lane);
  DF_STAMP(3);
  // ---- X = L^-1 in place ----
  // (1) diagonal blocks: lane = (block bi, row r) sweeps the columns c = 15 .. 0 of its block:
  //     X[r][c] = -rinv_c * sum_{k=c+1..r} X[r][k] L[k][c]  (r > c),  X[c][c] = rinv_c
  {
    const int bi = lane >> 4, r = lane & 15;
    const double* Lb = Ls + (16 * bi) * LDC + 16 * bi;
    double xr[16];
#pragma unroll
    for (int c = 15; c >= 0; --c) {
      // (column c of L is fetched whole, for every lane, before the sums, and the sums run over all k: xr[k] = 0 for k > r, a
      //  term that adds nothing.  With `if (k <= r)` around each term the compiler emitted a branch, an LDS read and a wait per
      //  term -- 120 LDS latencies in a row, 14 k of the kernel's 67 k cycles.  Same operations in the same order on the terms
      //  that count: same bits.)
      double lc[16];
#pragma unroll
      for (int k = c + 1; k < 16; ++k) lc[k] = Lb[k * LDC + c];
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = c + 1; k < 16; k += 2) {
        s0 = __builtin_fma(xr[k], lc[k], s0);
        if (k + 1 < 16) s1 = __builtin_fma(xr[k + 1], lc[k + 1], s1);
      }
      const double rc = rinv[16 * bi + c];
      xr[c] = r == c ? rc : (r > c ? -rc * (s0 + s1) : 0.0);
    }
    __builtin_amdgcn_wave_barrier();  // every lane has read its block before anybody overwrites it
#pragma unroll
    for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2*>(&Ls[lane * LDC + 16 * bi + c]) = double2{xr[c], xr[c + 1]};
    __builtin_amdgcn_wave_barrier();
  }
  DF_STAMP(4);
  // (2) blocks below the diagonal, block column jb = 2, 1, 0: W_i = L_i,jb X_jb,jb (in place of L_i,jb), then
  //     X_i,jb = -sum_{k=jb+1..i} X_ik W_k.  Fragments: A(row = l16, k = l4), B(k = l4, col = l16).
#pragma unroll
  for (int jb = 2; jb >= 0; --jb) {
    d4_t W[4];
#pragma unroll
    for (int ib = jb + 1; ib < 4; ++ib) {
      W[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) {
        const double af = Ls[(16 * ib + l16) * LDC + 16 * jb + kk + l4];
        const double bf = Ls[(16 * jb + kk + l4) * LDC + 16 * jb + l16];
        W[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, W[ib], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ib = jb + 1; ib < 4; ++ib)
#pragma unroll
      for (int g = 0; g < 4; ++g) Ls[(16 * ib + 4 * g + l4) * LDC + 16 * jb + l16] = W[ib][g];
    __builtin_amdgcn_wave_barrier();
    d4_t X[4];
#pragma unroll
    for (int ib = jb + 1; ib < 4; ++ib) {
      X[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kb = jb + 1; kb <= ib; ++kb)
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
          const double af = Ls[(16 * ib + l16) * LDC + 16 * kb + kk + l4];
          const double bf = Ls[(16 * kb + kk + l4) * LDC + 16 * jb + l16];
          X[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(-af, bf, X[ib], 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ib = jb + 1; ib < 4; ++ib)
#pragma unroll
      for (int g = 0; g < 4; ++g) Ls[(16 * ib + 4 * g + l4) * LDC + 16 * jb + l16] = X[ib][g];
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_wave_barrier();
  DF_STAMP(5);
  double* It = f.invL + (size_t(m) * f.T + j) * 4096;
  tile_rows_to_global(It, Ls, lane);
  DF_STAMP(6);
#ifdef ROMHC_DIAGF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DF_STAMP(7);
  if (lane == 0 && (m == 0 || m == 513))
    printf("diagf m %d j %d: load %llu panels %llu store_L %llu inv_diag %llu inv_offdiag %llu store_X_issue %llu store_X_drain %llu total %llu\n", m, j,
           df_stamp[1] - df_stamp[0], df_stamp[2] - df_stamp[1], df_stamp[3] - df_stamp[2], df_stamp[4] - df_stamp[3],
           df_stamp[5] - df_stamp[4], df_stamp[6] - df_stamp[5], df_stamp[7] - df_stamp[6], df_stamp[7] - df_stamp[0]);
#endif
}
#undef DF_STAMP
constexpr int DIAGF_LDS_DOUBLES = 64 * LDC + 64 * 4 + 64;
__global__ __launch_bounds__(64) void k_diag_factor(FemDev f, int slot, int j) {
  __shared__ __align__(16) double lds[DIAGF_LDS_DOUBLES];
  diag_factor_body(f, blockIdx.x, slot, j, lds, lds + 64 * LDC, lds + 64 * LDC + 256);
}


// Whole reduced solve of a system whose reduced matrix is ONE tile (e.g. 2x2 blocks at N = 128: 2 x 31
// compressed unknowns + the cross point); nothing but the interface vector leaves the CU:
//   assemble the lower 16x16 blocks in MFMA accumulator layout (term by term) ->
//   rank-4 blocked Cholesky (panel in row-per-lane form, trailing update on MFMA) with the forward substitution
//   carried along -> back substitution on registers ->
//   coefficient blocks for the extension (what k_coef does on the general path).
// Round 4: FOUR systems per workgroup, TWO waves per system, and every piece of the kernel moved to where it waits least.
//  * Assembly split BY BLOCKS instead of by systems.  Every system adds up the same (term, block) table pieces with its own
//    weights; with a wave per system each wave pulled all 117 KB of them through the CU's load path for itself (25 k of a
//    wave's 82 k cycles: 39 GB/s per CU with 64 KB in flight; sharing the pieces through an LDS ring moved the same bytes
//    through the LDS instead and got 12 k).  Now factor wave w adds up ITS blocks (a quarter of the pairs, balanced by the
//    host) for all four systems: a piece is fetched once per workgroup (two 16-byte loads per lane -- the pieces are stored
//    in accumulator layout, [g pair][lane][2], in the order the waves walk them: no index to wait for), multiplied by four
//    weights, and the finished block sums go to the LDS area of the system they belong to.  A block's sum runs over the
//    same pairs in the same order as before: same bits.
//  * Waves 0 .. 3 ("factor" waves, one per system): weights, their share of the assembly, the rhs, the factorisation of the
//    Cholesky's panels in row form with the forward substitution, the back substitution.  Waves 4 .. 7 ("update" waves,
//    one per system): everything that does not sit on that chain -- they hold the accumulators of the Cholesky, do its
//    MFMAs and layout changes while the factor wave factorises, fetch the descriptors of the tail and compute what of it
//    does not depend on the solution while the assembly runs, and build the coefficient blocks once the solution is there.
//    One instruction stream did all of this in turn before (59 k cycles a system; the Cholesky alone 29 k at ~270
//    instructions per panel, issue-bound: software pipelining inside ONE wave changed nothing).
// LDS (dynamic): term weights of the four systems | per system: the block sums of the assembly (20 KB), then the factor PACKED
// by rows (s1_lrow), then the weighted unknowns of the coefficient blocks in the same area; two panels; z; a; two panels on
// their way from the accumulators to row form | the matrix of the dense product of the tail
__global__ __launch_bounds__(512) void k_solve1(FemDev f, const double* __restrict__ a, int Mc) {
  extern __shared__ __align__(16) char s1_dyn[];
  const int w8 = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6), w = w8 & 3, lane = threadIdx.x & 63;
  const bool upd = w8 >= 4;  // the system's update wave
  double* const coefs = reinterpret_cast<double*>(s1_dyn);  // [system][term]
  const double* const Dl = reinterpret_cast<const double*>(s1_dyn + S1_COEF_BYTES + 4 * S1_WAVE_BYTES);  // the matrix of the dense product, 64 x ndi
  double* const Ls = reinterpret_cast<double*>(s1_dyn + S1_COEF_BYTES + w * S1_WAVE_BYTES);  // assembly: the ten lower blocks [block][g][lane]; Cholesky on: L packed by rows
  double* const Pn = Ls + 40 * 64;  // the 64 x 4 panel of the even steps of the Cholesky
  double* const zs = Pn + 64 * 4;
  double* const aL = zs + 64;       // the system's block coefficients a_m[0 .. kblk)
  double* const Pn1 = aL + 64;      // the panel of the odd steps (the update wave reads one while the other is written)
  double* const PnB = Pn1 + 64 * 4; // two 64 x 4 panels on their way from the accumulators to row form
  double* const wz = Ls;            // (the factor is dead once the back substitution is through)
  static_assert(DENSE_GROUPS_MAX * 64 <= 40 * 64 && 8 * 16 * 17 <= 40 * 64, "the shared area holds each of its three tenants");
  static_assert(S1_WAVE_BYTES == (40 * 64 + 64 * 4 + 64 + 64 + 3 * 64 * 4) * 8 && S1_COEF_BYTES == 4 * 64 * 8, "LDS areas");
  const int m_raw = 4 * int(blockIdx.x) + w;
  const bool live = m_raw < Mc;  // (the last workgroup may hold waves without a system: they walk along -- the barriers count them -- and store nothing)
  const int m = live ? m_raw : Mc - 1;
  const double* am = a + size_t(m) * f.kblk;
  double* ym = f.y + size_t(m) * f.nGp;
#ifdef ROMHC_SOLVE1_STAMPS
  unsigned long long stamp[10];
#define S1_STAMP(i) stamp[i] = __builtin_readcyclecounter()
  S1_STAMP(0);
#else
#define S1_STAMP(i)
#endif
// LDS traffic of ONE wave needs no barrier (a wave's LDS instructions execute in order); this keeps the compiler in line
#define S1_WAVE_SYNC()                                   \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)
// a workgroup barrier behind this wave's LDS traffic
#define S1_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  const int l16 = lane & 15, l4 = lane >> 4;

  if (upd) {
    // ================= the update wave of system w =================
    // the flat records of the tail (one per lane and item: rom_fem_setup.hip) are asked for first and wait in registers
    int4 irec[6], crec[S1_ITEM_PASSES];
    int wd[DENSE_GROUPS_MAX];
    {
      const int4* ir = reinterpret_cast<const int4*>(f.s1_items) + size_t(min(lane, max(f.ndi, 1) - 1)) * 6;
#pragma unroll
      for (int x = 0; x < 6; ++x) irec[x] = ir[x];
#pragma unroll
      for (int r = 0; r < S1_ITEM_PASSES; ++r) crec[r] = reinterpret_cast<const int4*>(f.s1_citems)[min(lane + 64 * r, max(f.ncoef, 1) - 1)];
#pragma unroll
      for (int g = 0; g < DENSE_GROUPS_MAX; ++g) wd[g] = g < f.ndg ? f.dweight[g * 64 + lane] : -2;
    }
    const int xr_first = lane < f.ncross ? f.xred[lane] : 0;
    const int2 sc_b = lane < f.nsc ? reinterpret_cast<const int2*>(f.scb)[lane] : int2{0, 0};
    const RhsTerm rt = f.rhs[min(lane, max(f.nrhs, 1) - 1)];  // lane t: rhs term t
    const double gvec = f.g[lane];
    __syncthreads();  // (1) weights, zeroed block areas, and the system's coefficients aL
    {  // rhs of the reduced system (k_rhs): y = g + sum_t weight_t * (vector t placed at its rows), the terms in their order;
       // parked in LDS for the factor wave (it needs it with the first panel, and has the assembly to do meanwhile)
      const double rcoef = lane < f.nrhs ? (rt.kind == 0 ? aL[rt.b0] / (aL[rt.e0] + aL[rt.e1]) : 0.5) : 0.0;  // lane t: weight of rhs term t
      double y = gvec;
      for (int t0 = 0; t0 < f.nrhs; t0 += 8) {  // eight terms at a time: their vector loads are in flight together
        double rv[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          rv[x] = 0.0;
          if (t0 + x < f.nrhs) {
            const int pos = __builtin_amdgcn_readlane(rt.pos, t0 + x), len = __builtin_amdgcn_readlane(rt.len, t0 + x);
            const int voff = __builtin_amdgcn_readlane(rt.voff, t0 + x);
            if (lane >= pos && lane < pos + len) rv[x] = f.vec[voff + lane - pos];
          }
        }
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (t0 + x < f.nrhs) y += readlane_f64(rcoef, t0 + x) * rv[x];
      }
      zs[lane] = y;
    }
    // Everything of the tail that does not depend on the solution, while the factor waves assemble.
    // The alignment gaps of the interface vector are read (against zero table entries) by the extension: they must hold
    // finite numbers whatever buffer the caller handed in -- zeroed here, ahead of every other store of this wave to its
    // vector (a wave's stores to one address stay in order; ALL stores to the vector are this wave's), instead of by a
    // memset launch in front of every sweep
    if (live)
      for (int i = lane; i < f.nGp; i += 64) ym[i] = 0.0;
    if (live && lane < f.nsc) ym[f.spos0 + lane] = sc_b.y >= 0 ? 1.0 / (aL[sc_b.x] + aL[sc_b.y]) : (1.0 / (double(f.N) * double(f.N))) / aL[sc_b.x];
    for (int i = lane + 64; live && i < f.nsc; i += 64) {
      const int b0 = f.scb[2 * i], b1 = f.scb[2 * i + 1];
      ym[f.spos0 + i] = b1 >= 0 ? 1.0 / (aL[b0] + aL[b1]) : (1.0 / (double(f.N) * double(f.N))) / aL[b0];
    }
    double wgt[DENSE_GROUPS_MAX];
#pragma unroll
    for (int g = 0; g < DENSE_GROUPS_MAX; ++g) {
      wgt[g] = 0.0;
      if (g < f.ndg) {
        const DenseGroup& dg = f.dgroups[g];
        wgt[g] = wd[g] >= 0 ? aL[wd[g]] : (wd[g] == -1 ? (aL[dg.b0] + aL[dg.b1]) / 2 : 0.0);
      }
    }
    const bool has_it = lane < f.ndi;  // this lane's item of the first pass of the dense product
    const int it_g = irec[0].x, it_pos = irec[0].y, it_nv = irec[0].z;
    double it_den = 1.0, it_cv[4] = {0.0, 0.0, 0.0, 0.0}, it_vv[4] = {0.0, 0.0, 0.0, 0.0};
    if (has_it) {
      const int voff[4] = {irec[1].y, irec[1].z, irec[1].w, irec[2].x}, vblk[4] = {irec[2].y, irec[2].z, irec[2].w, irec[3].x};
      const int vu0[4] = {irec[3].y, irec[3].z, irec[3].w, irec[4].x}, vu1[4] = {irec[4].y, irec[4].z, irec[4].w, irec[5].x};
      it_den = aL[irec[0].w] + aL[irec[1].x];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v < it_nv) {
          it_cv[v] = aL[vblk[v]] / (aL[vu0[v]] + aL[vu1[v]]);
          it_vv[v] = f.vec[voff[v]];
        }
    }
    // coefficient items of the first S1_ITEM_PASSES passes: constants go out now, copies of the solution are remembered
    int ci_dst[S1_ITEM_PASSES], ci_src[S1_ITEM_PASSES];
#pragma unroll
    for (int r = 0; r < S1_ITEM_PASSES; ++r) {
      ci_src[r] = -1;
      ci_dst[r] = crec[r].x & 0xfffffff;
      if (lane + 64 * r < f.ncoef) {
        const int code = crec[r].x >> 28;
        if (code == 2) ci_src[r] = crec[r].y;
        else if (code != 0 && live) ym[ci_dst[r]] = code == 1 ? 1.0 / (aL[crec[r].z] + aL[crec[r].w]) : 0.0;
      }
    }
    __syncthreads();  // (2) the block sums are complete
    // ---- the accumulators of the Cholesky: the ten lower blocks in MFMA accumulator layout
    d4_t C[4][4];
    {
      int q = 0;
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb <= ib; ++jb, ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int R = 16 * ib + 4 * g + l4, Cc = 16 * jb + l16;
            const double x = Ls[(q * 4 + g) * 64 + lane];
            C[ib][jb][g] = (R >= f.s1_ndr || Cc >= f.s1_ndr) ? (R == Cc ? 1.0 : 0.0) : x;  // padding unknowns: identity
          }
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) PnB[x * 64 + lane] = 0.0;
    S1_BARRIER();  // (3) the block sums have been read: the factor may move into their place
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
    unsigned long long uwait = 0;
    const unsigned long long ut0 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int p = 0; p < 15; ++p) {
      const int jb = p >> 2;
      {  // the columns of panel p + 1 as the accumulators have them (updates of the panels before p) -> row form via LDS
        const int p1 = p + 1, jb1 = p1 >> 2, co1 = 4 * (p1 & 3);
        double* dstp = PnB + (p1 & 1) * 256;
        if ((l16 >> 2) == (p1 & 3)) {
#pragma unroll
          for (int ib = jb1; ib < 4; ++ib)
#pragma unroll
            for (int g = 0; g < 4; ++g) dstp[(16 * ib + 4 * g + l4) * 4 + (l16 - co1)] = C[ib][jb1][g];
        }
      }
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
      { const unsigned long long t0_ = __builtin_readcyclecounter(); S1_BARRIER(); uwait += __builtin_readcyclecounter() - t0_; }
#else
      S1_BARRIER();  // panel p is factorised and in LDS; the base of panel p + 1 is in LDS
#endif
      const double* Pp = (p & 1) ? Pn1 : Pn;
      double frag[4];
#pragma unroll
      for (int x = jb; x < 4; ++x) frag[x] = Pp[(16 * x + l16) * 4 + l4];  // same map for A (row, k) and B (k, col)
#pragma unroll
      for (int jb2 = jb; jb2 < 4; ++jb2)
#pragma unroll
        for (int ib = jb2; ib < 4; ++ib)
          C[ib][jb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag[ib], frag[jb2], C[ib][jb2], 0, 0, 0);
    }
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
    const unsigned long long ut1 = __builtin_readcyclecounter();
#endif
    S1_BARRIER();  // (4) the solution z is in LDS
    if (!live) return;
    // ---- coefficient blocks + nodal copy of the cross points (what k_coef does on the general path).  The blocks of
    // the closed-form edges are one dense product here: out[it] = sum_j D[j][it] * (w_g(j) z_j), D = all their
    // matrices side by side (64 x items, in LDS since the start of the kernel), w_g(j) the weight of source j for group g.
    const double z = zs[lane];
    ym[lane] = z;
    if (lane < f.ncross) ym[f.xb0 + lane] = zs[xr_first];
    for (int x = lane + 64; x < f.ncross; x += 64) ym[f.xb0 + x] = zs[f.xred[x]];
#pragma unroll
    for (int g = 0; g < DENSE_GROUPS_MAX; ++g)
      if (g < f.ndg) wz[g * 64 + lane] = wgt[g] * z;
    S1_WAVE_SYNC();
    if (has_it) {
      const double* wg = wz + it_g * 64;
      const double* D = Dl + lane;
      double acc = 0.0;
#pragma unroll 1
      for (int j0 = 0; j0 < 64; j0 += 16) {  // (16 at a time: the whole column in flight at once costs 256 registers)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += D[(j0 + j) * f.ndi] * wg[j0 + j];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v < it_nv) acc += it_cv[v] * it_vv[v];
      ym[it_pos] = acc / it_den;
    }
#pragma unroll
    for (int r = 0; r < S1_ITEM_PASSES; ++r)
      if (ci_src[r] >= 0) ym[ci_dst[r]] = zs[ci_src[r]];
    for (int it = lane + 64 * S1_ITEM_PASSES; it < f.ncoef; it += 64) {
      const CoefGroup& cg = f.groups[f.item_group[it]];
      const int k = f.item_k[it];
      if (cg.kind == 1 && k < cg.r) continue;  // done above
      ym[cg.cpos + k] = k == cg.r ? 1.0 / (aL[cg.b0] + aL[cg.b1]) : (k < cg.r ? zs[cg.zpos + k] : 0.0);
    }
#ifdef ROMHC_SOLVE1_STAMPS
    if (lane < 12) ym[lane] = PnB[lane];  // (dev build: the factor wave's stamps, parked in LDS, replace the first unknowns)
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
    if (lane == 12) ym[12] = double(uwait);
    if (lane == 13) ym[13] = double(ut1 - ut0);
#endif
#endif
    return;
  }

  // ================= the factor wave of system w =================
  // The matrix of the dense product of the tail (64 x ndi doubles, the same for every system) goes to LDS once per
  // workgroup, by LDS-DMA, before anything else: it lands while the assembly runs, and the barrier behind the assembly
  // publishes it.  (Each wave used to pull its own copy through the load path AFTER the back substitution: 64 eight-byte
  // loads per lane at the 40 GB/s per CU that path gives such loads, 7 k cycles.)
  {
    const unsigned d0 = unsigned(size_t((__attribute__((address_space(3))) char*)s1_dyn)) + unsigned(S1_COEF_BYTES + 4 * S1_WAVE_BYTES);
    const int nchunk = (f.ndi + 1) >> 1;  // kilobytes (the host pads the matrix to whole ones)
    const char* src = reinterpret_cast<const char*>(f.dmat);
    for (int c = w; c < nchunk; c += 4)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(d0 + unsigned(c) * 1024u)),
                   "v"(unsigned(lane) * 16u), "s"(x128_uniform(src + size_t(c) * 1024))
                   : "memory", "m0");
  }
  // Assembly in the MFMA accumulator layout of the Cholesky below: element g of block q = (ib, jb), ib >= jb, is
  // (row 16 ib + 4 g + (lane >> 4), column 16 jb + (lane & 15)).  The host lists the (term, block) pairs whose
  // rectangle and block intersect (57 at 2x2 / N=128), block by block, and deals the blocks to the four factor waves; a
  // pair is exactly two loads, so a ring of PAIR_RING pairs keeps 2 * PAIR_RING loads in flight with statically known
  // wait counts.  The sums of a block (one per system) are kept in registers and stored to the LDS copy of the blocks
  // of their system when the block's last pair is done (the target block of a pair is a run-time index, which registers
  // cannot have).
  {
    const int p0 = f.wp0[w], np = f.wp0[w + 1] - p0;  // this wave's pairs: pieces p0 .. of pool_acc, metas p0 .. of wmeta
    const double2* pbase = reinterpret_cast<const double2*>(f.pool_acc) + size_t(p0) * 128 + lane;
    double2 v[PAIR_RING][2];
    auto issue = [&](int i, double2 (&dst)[2]) {
      dst[0] = pbase[size_t(i) * 128];       // g = 0, 1
      dst[1] = pbase[size_t(i) * 128 + 64];  // g = 2, 3
    };
    // Everything that depends on nothing is asked for at once, the small things first (a wave's loads return in order):
    // the system's coefficients, the descriptors of the terms (lane t: term t), the metas, then the first PAIR_RING pieces.  The coefficients are parked in LDS: what used to be chains of dependent global loads
    // (descriptor -> a_m[block] -> divide, 6 k + 7 k cycles at the head of the kernel and of the rhs) are LDS reads.
    const double a_l = lane < f.kblk ? am[lane] : 1.0;
    const GenTerm gt = f.terms[f.s1_t0 + min(lane, f.s1_nterm - 1)];
    int mine = f.wmeta[p0 + lane];  // lane i holds the meta of pair i of the current group of 64
#pragma unroll
    for (int u = 0; u < PAIR_RING; ++u) issue(u, v[u]);
#pragma unroll
    for (int x = 0; x < 40; ++x) Ls[x * 64 + lane] = 0.0;
    aL[lane] = a_l;
    S1_WAVE_SYNC();
    coefs[w * 64 + lane] = lane < f.s1_nterm ? term_coef(gt, aL) : 0.0;  // lane t: weight of term t
    __syncthreads();  // (1) every system's weights are there; every system's block area is zeroed
    double cf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cf[s] = coefs[s * 64 + lane];
    double acc4[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc4[s][g] = 0.0;
    double* const Cl0 = reinterpret_cast<double*>(s1_dyn + S1_COEF_BYTES) + lane;  // system s: + s * (S1_WAVE_BYTES / 8)
    S1_STAMP(1);
    for (int base = 0; base < np; base += 64) {
      const int cur = mine;
      const int nxt = f.wmeta[p0 + base + 64 + lane];  // (the list carries 128 no-ops behind its end)
      const int cnt = min(64, np - base);  // (a multiple of PAIR_RING: the host pads every wave's list with no-ops)
#pragma unroll
      for (int i0 = 0; i0 < 64; i0 += PAIR_RING) {  // (unrolled: straight-line code keeps the compiler's wait counts exact)
        if (i0 >= cnt) break;
#pragma unroll
        for (int u = 0; u < PAIR_RING; ++u) {
          const int i = i0 + u;
          const int meta = __builtin_amdgcn_readlane(cur, i);  // term | q << 8 | (last pair of block q) << 16
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double c = readlane_f64(cf[s], meta & 0xff);
            acc4[s][0] = __builtin_fma(c, v[u][0].x, acc4[s][0]);
            acc4[s][1] = __builtin_fma(c, v[u][0].y, acc4[s][1]);
            acc4[s][2] = __builtin_fma(c, v[u][1].x, acc4[s][2]);
            acc4[s][3] = __builtin_fma(c, v[u][1].y, acc4[s][3]);
          }
          if (meta >> 16) {  // the pairs are sorted by block: its sums are complete
            double* dst = Cl0 + ((meta >> 8) & 0xff) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                dst[s * (S1_WAVE_BYTES / 8) + g * 64] = acc4[s][g];
                acc4[s][g] = 0.0;
              }
          }
          issue(base + i + PAIR_RING, v[u]);
        }
      }
      mine = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's share of the dense product's matrix has landed)
    __syncthreads();  // (2) the four waves' blocks are in every system's area
  }
  S1_STAMP(2);
  double nv[4];  // the current panel of the Cholesky in row form (row `lane`), all updates applied
  {  // panel 0 straight from the block sums: row r = lane sits in block (r >> 4, 0), element g = (r >> 2) & 3
    const int ib = lane >> 4;
    const double* src = Ls + ((ib * (ib + 1) / 2) * 4 + ((lane >> 2) & 3)) * 64 + (lane & 3) * 16;
    const double2 x0 = *reinterpret_cast<const double2*>(src), x1 = *reinterpret_cast<const double2*>(src + 2);
    nv[0] = x0.x; nv[1] = x0.y; nv[2] = x1.x; nv[3] = x1.y;
#pragma unroll
    for (int k = 0; k < 4; ++k)  // padding unknowns: identity
      if (lane >= f.s1_ndr || k >= f.s1_ndr) nv[k] = lane == k ? 1.0 : 0.0;
  }
  S1_BARRIER();  // (3) the block sums have been read (by this wave and by the update wave): the factor may move in
  S1_STAMP(3);
  double y = zs[lane];  // the rhs of the reduced system, built by the update wave while the assembly ran
  // Blocked right-looking Cholesky, 16 panels of 4 columns, on the system's two waves.  This wave keeps the current panel
  // in row form (nv) and factorises it (chol_panel_regs: every lane factorises the 4 x 4 diagonal block for itself; the
  // forward substitution of y rides along); the update wave holds the accumulators, applies the rank-4 updates to them
  // with MFMAs (K = 4 is exactly one panel) and hands the NEXT panel's columns over in row form as the panels BEFORE the
  // current one left them (`base`) -- this wave applies the current panel's update to them itself, on the vector pipe,
  // k in the order of the matrix instruction (same bits as the all-MFMA form).  One barrier per panel: factorised panel
  // and base are in LDS.
  double myrs = 0.0;
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
  unsigned long long fwait = 0;
#endif
  S1_STAMP(4);
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int c0 = 4 * p;
    chol_panel_regs(lane, c0, nv, y, myrs);
    {  // the factorised panel: operand array of the update wave's MFMAs ...
      double* Pp = (p & 1) ? Pn1 : Pn;
      *reinterpret_cast<double2*>(&Pp[lane * 4]) = double2{nv[0], nv[1]};
      *reinterpret_cast<double2*>(&Pp[lane * 4 + 2]) = double2{nv[2], nv[3]};
    }
    if ((lane >> 2) >= p) {  // ... and row `lane` of L, packed in whole panels: rows 4 g .. 4 g + 3 hold panels 0 .. g (zeros right of the diagonal)
      double* lrow = Ls + s1_lrow(lane) + c0;
      *reinterpret_cast<double2*>(lrow) = double2{nv[0], nv[1]};
      *reinterpret_cast<double2*>(lrow + 2) = double2{nv[2], nv[3]};
    }
    if (p < 15) {
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
      { const unsigned long long t0_ = __builtin_readcyclecounter(); S1_BARRIER(); fwait += __builtin_readcyclecounter() - t0_; }
#else
      S1_BARRIER();
#endif
      const double* bp = PnB + ((p + 1) & 1) * 256 + lane * 4;
      const double2 b01 = *reinterpret_cast<const double2*>(bp), b23 = *reinterpret_cast<const double2*>(bp + 2);
      double nn[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int k = 0; k < 4; ++k) nn[kk] = __builtin_fma(-nv[k], readlane_f64(nv[k], c0 + 4 + kk), nn[kk]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) nv[kk] = nn[kk];
    }
  }
  if (chol_pivots_bad(myrs) && live && lane == 0) atomicOr(f.status, 1);
  S1_WAVE_SYNC();
  S1_STAMP(5);
  // back substitution x = L^-T y.  Column `lane` of L is fetched from LDS in batches (conflict free), then the
  // chain x_j = y_j / L_jj ; y_i -= L_ji x_j (i < j) runs on registers and readlane broadcasts only.
#pragma unroll
  for (int h = 1; h >= 0; --h) {  // (in two halves: 64 registers of L at a time)
    double lcol[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) lcol[j] = Ls[s1_lrow(32 * h + j) + lane];  // L[j][lane] for lane <= j (beyond: padding or another row's entry, never used)
#pragma unroll
    for (int j = 31; j >= 0; --j) {
      const int jj = 32 * h + j;
      const double xj = readlane_f64(y, jj) * readlane_f64(myrs, jj);
      if (lane == jj) y = xj;
      else if (lane < jj) y -= lcol[j] * xj;
    }
  }
  zs[lane] = y;
  S1_STAMP(6);
#ifdef ROMHC_SOLVE1_STAMPS
#pragma unroll
  for (int k = 0; k < 7; ++k)
    if (lane == k) PnB[k] = double(stamp[k] - stamp[0]);  // (dev build: the update wave stores them in place of the first unknowns)
  if (lane >= 7 && lane < 12) PnB[lane] = 0.0;
#ifdef ROMHC_SOLVE1_PANEL_STAMPS
  if (lane == 10) PnB[10] = double(fwait);
#endif
#endif
  S1_BARRIER();  // (4) the solution z is in LDS: the update wave builds the coefficient blocks
#undef S1_STAMP
#undef S1_WAVE_SYNC
#undef S1_BARRIER
}

// Sub-diagonal tiles of column j: C = S_ij - sum_k L_ik L_jk^T ; L_ij = C invL_jj^T ;
// y_i -= L_ij y_j.  invL_jj is lower triangular: the waves owning output columns 0..31 only need
// k < 32 of the second product.
template <int NS>
__device__ inline void panel_body(const FemDev& f, const double* __restrict__ am0, int m0, int nsys, int j, int ent, double* lds,
                                  double* yj, double* coef) {
  double* Cb = lds;                  // (the DMA slots of the k loop alias it)
  double* stB = lds + TILE_DOUBLES;  // one 8-wide chunk of invL_jj
  const int slot = f.colrow[ent];
  const int ti = f.colti[ent];
  const WavePos wp;
  const TileDesc& d = f.desc[slot];
  const int t = threadIdx.x;
  const bool stream = NS == 1 && f.tile_stream && d.t1 - d.t0 <= 128;  // (the assembly as a stream of kilobytes behind the k-loop: s_tile_to_lds)
  STile st[NS];
  if (!stream) s_tile_load<NS>(st, d, f, am0, nsys, coef);
  static_assert(TD_LDS_BYTES <= FACT_LDS_DOUBLES * 8, "the DMA slots alias the staging area and the C tile");
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (q >= nsys) break;
    const int m = m0 + q;
    double* Lm = f.L + size_t(m) * f.nslots * 4096;
    if (q > 0) __syncthreads();  // (the rhs update of the previous system reads the tile and yj)
    if (t < 64) yj[t] = f.y[size_t(m) * f.nGp + j * 64 + t];
    TileStream<1> ts;
    if (stream) s_tile_stream_begin<1>(ts, slot, d, f, am0 + size_t(q) * f.kblk, 1);
    Acc acc;
    acc_zero(acc);
    accumulate_klist_dma(f, slot, Lm, [](int) { return true; }, acc, reinterpret_cast<char*>(lds), wp);
    if (stream) {
      __syncthreads();  // (the k-loop's slots)
      double* T[1] = {Cb};
      s_tile_to_lds<1>(T, ts, d, f);
      tile_minus_acc(Cb, acc, wp);
    } else {
      tile_from_acc(Cb, acc, st[q], wp);
    }

    // X = C * invL_jj^T
    acc_zero(acc);
    // (8-wide chunks of invL through ONE 5 KB staging buffer: 16 barriers instead of 5, and 13 KB less LDS per workgroup)
    const double* I = f.invL + (size_t(m) * f.T + j) * 4096 + (t >> 2) * 64 + (t & 3) * 2;
    const int kmax = (wp.wc + 1) * 32;  // invL[c][k] = 0 for k > c
    {
      const int fr = wp.lane & 15, kq = wp.lane >> 4;
      double2 vb = *reinterpret_cast<const double2*>(I);
      for (int ch = 0; ch < 8; ++ch) {
        *reinterpret_cast<double2*>(stB + (t >> 2) * LDK8 + (t & 3) * 2) = vb;
        __syncthreads();
        if (ch + 1 < 8) vb = *reinterpret_cast<const double2*>(I + (ch + 1) * 8);
        if (ch * 8 < kmax) {
          const double* pa = Cb + (wp.wr * 32 + fr) * LDC + ch * 8 + kq;
          const double* pb = stB + (wp.wc * 32 + fr) * LDK8 + kq;
#pragma unroll
          for (int kk = 0; kk < 8; kk += 4) {
            const double a0 = pa[kk], a1 = pa[16 * LDC + kk];
            const double b0 = pb[kk], b1 = pb[16 * LDK8 + kk];
            acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
            acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
            acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
            acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    }

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
      double sum = 0.0;
      for (int k = 0; k < 64; ++k) sum += Cb[t * LDC + k] * yj[k];
      f.y[size_t(m) * f.nGp + ti * 64 + t] -= sum;
    }
  }
}

// XCD-aware mapping of (system, row of the column) for `nrows` row tiles per system: block ids are dealt round-robin
// to the 8 XCDs (each with its own L2); all row tiles of one system are given ids of the same residue mod 8 so that
// the L_jk / invL_jj tiles they share are served by one L2.  (Placement only affects speed, never correctness.)
__device__ inline void panel_block(int b, int nrows, int Mc, int& m, int& row) {
  const int xcd = b & 7, idx = b >> 3;
  const int full = (Mc >> 3) << 3;  // systems covered by complete groups of 8
  m = xcd + 8 * (idx / nrows);
  row = idx % nrows;
  if (m >= full) {  // ragged tail (Mc % 8 systems): plain order
    const int tb = b - full * nrows;
    m = full + tb / nrows;
    row = tb % nrows;
  }
}

// (NS = 2, a workgroup doing its row tile for two systems with one pass over the term tables, pays in the diagonal
// update -- 20 % -- but not here: C4 0.69 ms either way, C5 7 % slower; profiles/r02_tile_cholesky_probes.txt)
template <int NS>
__global__ __launch_bounds__(256, 4) void k_factor_panel(FemDev f, const double* __restrict__ a, int j, int Mc) {  // (<= 128 VGPRs: four waves per SIMD)
  __shared__ __align__(16) double lds[FACT_LDS_DOUBLES];  // (+ yj = 39,424 B: four workgroups per CU)
  __shared__ double yj[64];
  double* coef = lds;  // NS * COEF_MAX term weights: only the assembly reads them, before the k loop writes the area
  int mp, row;
  panel_block(blockIdx.x, f.colptr[j + 1] - f.colptr[j], (Mc + NS - 1) / NS, mp, row);
  const int m0 = mp * NS;
  panel_body<NS>(f, a + size_t(m0) * f.kblk, m0, min(NS, Mc - m0), j, f.colptr[j] + row, lds, yj, coef);
}
template __global__ void k_factor_panel<1>(FemDev, const double*, int, int);

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

// (lane_swap1, double2_u: rom_fem_dev.h)

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
// one mesh row of up to 128 interior vertices): K is only sum(rank + 1) ~ 64, so a 64 x 64 tile spends most of its
// life in its prologue and epilogue; four times the outputs per workgroup amortise them.  Eight waves, 2 x 4 wave
// tiles of 64 systems x 32 vertices = 4 x 2 MFMA accumulators (106 VGPRs): two waves of the workgroup on every
// SIMD, and the two halves of the workgroup take turns fetching the chunks -- a wave held up at the issue of its
// loads (a fifth of the launch goes there while the neighbour workgroup's stores fill the CU's memory pipeline,
// profiles/r02_extend128_kloop_probes.txt) leaves the MFMA pipe to its twin.  (Four waves of 64 x 64 with 184 VGPRs,
// the shape of round 1: +1 % at C2, +3.5 % at C4.)
// grid (mesh rows x column tiles, ceil(Mc/128), lr blocks); LDS 66,560 B -> 2 workgroups per CU

// Main loop (round 2, after the Gram kernel): the K chunks of both operands go from global memory straight into
// LDS with global_load_lds_dwordx4 -- no staging registers, no ds_write, 8 instructions per fetching wave and chunk, chunk
// ch + 1 in flight under the 32 MFMAs per wave of chunk ch, the fragments of k-step j + 1 read before the MFMAs of step
// j.  A chunk slot is {A k 0..7 | A k 8..15 | B k 0..7 | B k 8..15}, 128 rows of 64 bytes each; a DMA instruction
// writes 64 lanes x 16 bytes back to back = 16 rows of ONE half, i.e. of one block side (scalar base + 32-bit lane
// offset); the four 16-byte units of a row are stored at position u ^ ((row >> 2) & 3), which makes the MFMA fragment
// reads conflict free.  Rows that do not exist (systems >= Mc, vertices behind the end of the mesh row / the block) are
// clamped to the last one that does: their products are not stored.
constexpr int X128_SLOT = 4 * 128 * 64;  // bytes

// FLAT: the 128 vertices of a tile are consecutive in the block's row-major vertex numbering instead of lying in
// one mesh row -- no padding when n1 is not close to a multiple of 128 (n1 = 170: 226 tiles per block instead of
// 340); a pair of adjacent vertices may then straddle two mesh rows and is stored as two 8-byte halves.
template <bool FLAT>
__global__ __launch_bounds__(512, 2) void k_extend128(FemDev f, X128Args xa, const double* __restrict__ a, int Mc,
                                                      double* __restrict__ U, long long row0, int with_expand,
                                                      int sys_fast) {
  __shared__ __align__(16) char lds_bytes[2 * X128_SLOT];  // two chunk slots = 65,536 B: two workgroups per CU
  __shared__ double scs[128];                               // h^2 / a_b of the workgroup's systems
  double* const lds = reinterpret_cast<double*>(lds_bytes);
  const int n1 = f.n1, N = f.N;
  const int nct = (n1 + 127) / 128;
  const int nvert = n1 * n1;
  const int ntile = FLAT ? (nvert + 127) / 128 : n1 * nct;
  // Workgroup order.  Classic (sys_fast = 0): grid (tiles, system groups, blocks) -- consecutive workgroups walk the
  // vertex tiles of ONE system group: they share the interface vectors (small) and each reads its own slab of the tables.
  // sys_fast = number of system groups: 1-D grid in x, the system group varies fastest -- the workgroups in flight at
  // any time share a few table slabs (128 vertices x K x 8 B each), which then come out of L2 / the Infinity Cache
  // instead of HBM once per system group; pays where the tables outgrow the caches (C4 / C5).  Placement only: the rows
  // are the same bits.
  // sys_fast < 0: the same with the XCDs in mind (workgroup i runs on XCD i % 8): units of 8 tiles x all -sys_fast system
  // groups, tile = position % 8 inside the unit -- every slab is fetched by ONE XCD's L2 instead of all eight.
  int bx, by, ngy;
  if (sys_fast > 0) {
    bx = int(blockIdx.x) / sys_fast; by = int(blockIdx.x) % sys_fast; ngy = sys_fast;
  } else if (sys_fast < 0) {
    ngy = -sys_fast;
    const int unit = int(blockIdx.x) / (8 * ngy), r = int(blockIdx.x) % (8 * ngy);
    bx = unit * 8 + (r & 7); by = r >> 3;
  } else {
    bx = int(blockIdx.x); by = int(blockIdx.y); ngy = int(gridDim.y);
  }
  if (bx >= ntile + with_expand) return;  // (padding of the last unit)
  if (bx >= ntile) {
    // with_expand: the (small) expansion of the edge values rides in `with_expand` extra workgroups per (y, z) cell
    // of this launch -- it depends on nothing here and nothing here depends on it
    static_assert(STAGE_TOTAL * sizeof(double) <= 2 * X128_SLOT, "expansion staging must fit");
    const int nx = f.n1p / 64, ny = (Mc + 63) / 64;
    const int item = (bx - ntile) + with_expand * (by + ngy * blockIdx.z);
    if (threadIdx.x >= 256) return;  // (the expansion is written for four waves)
    if (item < nx * ny * (f.nexp + 1)) expand_tile(f, Mc, U, row0, lds, item % nx, (item / nx) % ny, item / (nx * ny));
    return;
  }
  const int bz = blockIdx.z;
  {
    // Two workgroups share a CU; started together they run in lockstep (both loading / multiplying, then both in
    // the epilogue).  The second half of the first round starts one MFMA phase late (about 64 cycles per MFMA)
    // so that the phases of the two interleave: measured 274 -> 245 us at 256x256 / 2x2 / 1024 systems.
    // Placement only affects speed.
    const unsigned lin = __builtin_amdgcn_readfirstlane(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * bz));
    if (lin >= 256u && lin < 512u)
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);  // 2 x 127 x 64 cycles
  }
  const int b = xa.blocks[bz];
  const int p = b / f.ncb, q = b % f.ncb;
  const BlockSide sd = xa.sides[bz];  // by value and read without branches below: one batch of scalar loads
  const int iv = bx / nct + 1;           // mesh row (1-based interior index)       (!FLAT)
  const int jv0 = 128 * (bx % nct) + 1;  // first vertex of the tile                (!FLAT)
  const int vt0 = 128 * bx;              // first vertex of the tile, block-local   (FLAT)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = w >> 2, wc = w & 3;
  constexpr int NJ = 2;  // 16-vertex column blocks per wave (wave tile 64 x 32)
  const int fr = lane & 15, kq = lane >> 4;
  double my_sc = 0.0;  // h^2 / a_b of system threadIdx.x: requested now, parked in LDS after the k loop
  if (threadIdx.x < 128) {
    const int m = by * 128 + threadIdx.x;
    if (m < Mc) my_sc = f.y[size_t(m) * f.nGp + f.sblk0 + b];
  }
  // everything the epilogue needs from memory is fetched before the first store: a load after a store would
  // make its s_waitcnt vmcnt wait for the stores as well (one counter, in order)
  double w_own[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (FLAT) {
      const int v = vt0 + wc * (16 * NJ) + j * 16 + fr;
      w_own[j] = v < nvert ? f.W[v] : 0.0;
    } else {
      const int jj = jv0 + wc * (16 * NJ) + j * 16 + fr;  // 1-based
      w_own[j] = jj <= n1 ? f.W[(iv - 1) * n1 + (jj - 1)] : 0.0;
    }
  }
  // K runs over the sides' rank + 1 coefficients in segments of 8: the tables are padded to 16 per side, but only
  // ceil((rank + 1) / 8) segments of a side hold anything -- the second half of a side's last chunk is handed to the
  // next side (7-17 % fewer MFMAs where rank + 1 is just above a multiple of 16; products with the zero padding add
  // nothing, so the sums do not change).  A chunk = two consecutive segments of that walk.
  const int cnt0 = sd.s[0].mode == 2 ? min(2 * sd.s[0].nch, (sd.s[0].r + 1 + 7) / 8) : 0;
  const int cnt1 = sd.s[1].mode == 2 ? min(2 * sd.s[1].nch, (sd.s[1].r + 1 + 7) / 8) : 0;
  const int cnt2 = sd.s[2].mode == 2 ? min(2 * sd.s[2].nch, (sd.s[2].r + 1 + 7) / 8) : 0;
  const int cnt3 = sd.s[3].mode == 2 ? min(2 * sd.s[3].nch, (sd.s[3].r + 1 + 7) / 8) : 0;
  const int nseg = cnt0 + cnt1 + cnt2 + cnt3;
  const int tot = (nseg + 1) / 2;
  // ---- fragment addressing: lane (fr, kq) reads row fr (+ 16 i) at k = 4 kki + kq: half kki >> 1, unit
  // (2 (kki & 1) + (kq >> 1)) ^ (fr >> 2), byte (kq & 1) * 8
  const unsigned fx = unsigned((kq >> 1) ^ (fr >> 2));
  const unsigned fa0 = unsigned(fr * 64) + (fx << 4) + unsigned(kq & 1) * 8u;
  const unsigned fa1 = unsigned(fr * 64) + ((fx ^ 2u) << 4) + unsigned(kq & 1) * 8u;
#define X_FRAGS(SLOT_, KKI_, AF_, BF_)                                                                             \
  do {                                                                                                             \
    const char* pf_ = lds_bytes + (SLOT_) * X128_SLOT + (((KKI_)&1) ? fa1 : fa0);                                  \
    const char* pa_ = pf_ + ((KKI_) >> 1) * 8192 + wr * 4096;                                                      \
    const char* pb_ = pf_ + 16384 + ((KKI_) >> 1) * 8192 + wc * (1024 * NJ);                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) AF_[i_] = *reinterpret_cast<const double*>(pa_ + i_ * 1024);  \
    _Pragma("unroll") for (int i_ = 0; i_ < NJ; ++i_) BF_[i_] = *reinterpret_cast<const double*>(pb_ + i_ * 1024); \
  } while (0)
  // ---- DMA addressing: wave w of the fetching half takes rows 32 (w & 3) .. + 31 of both operands, 16 rows of one half
  // per instruction: lane -> row 32 (w & 3) + 16 g + (lane >> 2), stored unit lane & 3 = logical unit (lane & 3) ^ ((lane >> 4) & 3)
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds_bytes));
  const unsigned du16 = unsigned(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const unsigned ybytes_row = unsigned(f.nGp) * 8u;
  const int m0 = by * 128;
  unsigned voA[2];             // lane offsets behind ybytes + first system + side block
  int vi[2], vj[2];            // the lanes' vertices (1-based) in the wave's two B row groups
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int rl = 32 * (w & 3) + 16 * g + (lane >> 2);
    voA[g] = unsigned(max(0, min(rl, Mc - 1 - m0))) * ybytes_row + du16;
    if (FLAT) {
      const int v = min(vt0 + rl, nvert - 1);
      vi[g] = v / n1 + 1;
      vj[g] = v % n1 + 1;
    } else {
      vi[g] = iv;
      vj[g] = min(jv0 + rl, n1);
    }
  }
  const char* const ybase = reinterpret_cast<const char*>(f.y) + size_t(min(m0, Mc - 1)) * ybytes_row;
  const char* const gsbase = reinterpret_cast<const char*>(f.Gs);
  const size_t seg_stride = size_t(n1) * n1 * 64;  // bytes between the K segments of a table
  const char* const zbase = reinterpret_cast<const char*>(f.W + size_t(n1) * n1);  // EXT_ZERO_PAGE doubles of zeros
  const unsigned voZ = unsigned(lane) * 16u;
  // the walk over the segments (uniform): side, segments left in it, its two running pointers; lanes: table rows
  int c_side = -1, c_left = 0, c_segs = nseg;
  const char* pA = zbase;
  const char* pB = zbase;
  unsigned voB[2] = {0, 0};
#define X_NEXT_SIDE()                                                                                              \
  do {                                                                                                             \
    do {                                                                                                           \
      ++c_side;                                                                                                    \
      c_left = c_side == 0 ? cnt0 : c_side == 1 ? cnt1 : c_side == 2 ? cnt2 : cnt3;                                \
    } while (c_left == 0 && c_side < 3);                                                                           \
    const int off_ = c_side == 0 ? sd.s[0].off : c_side == 1 ? sd.s[1].off : c_side == 2 ? sd.s[2].off : sd.s[3].off; \
    const int gseg_ = c_side == 0 ? sd.s[0].gseg : c_side == 1 ? sd.s[1].gseg : c_side == 2 ? sd.s[2].gseg : sd.s[3].gseg; \
    /* row of the side's segment-major table (k_repack_table) = ci i + cj j + k0, i / j the 1-based interior indices: */ \
    /* side 0: n1 (i - 1) + (j - 1); 1: n1 (n1 - i) + (j - 1); 2: n1 (i - 1) + (j - 1); 3: n1 (i - 1) + (n1 - j)  */     \
    const int ci_ = c_side == 1 ? -n1 : n1, cj_ = c_side == 3 ? -1 : 1;                                             \
    const int k0_ = c_side == 1 ? n1 * n1 - 1 : c_side == 3 ? -n1 + n1 : -n1 - 1;                                   \
    pA = ybase + size_t(off_) * 8;                                                                                 \
    pB = gsbase + size_t(gseg_) * 8;                                                                               \
    voB[0] = unsigned(ci_ * vi[0] + cj_ * vj[0] + k0_) * 64u + du16;                                               \
    voB[1] = unsigned(ci_ * vi[1] + cj_ * vj[1] + k0_) * 64u + du16;                                               \
  } while (0)
#define X_DMA(LDS_, BASE_, VOFF_)                                                                                  \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),   \
               "v"(VOFF_), "s"(x128_uniform(BASE_))                                                                \
               : "memory", "m0")
  // the 4 loads of the segment under the cursor into half H_ of slot SLOT_, then the cursor moves on
#define X_ISSUE_HALF(SLOT_, H_, MINE_)                                                                             \
  do {                                                                                                             \
    const unsigned sb_ = lds0 + unsigned(SLOT_) * X128_SLOT + (H_) * 8192 + unsigned(w & 3) * 2048;                \
    if (c_segs > 0) {                                                                                              \
      if (MINE_) {                                                                                                 \
        X_DMA(sb_, pA, voA[0]);                                                                                    \
        X_DMA(sb_ + 1024, pA, voA[1]);                                                                             \
        X_DMA(sb_ + 16384, pB, voB[0]);                                                                            \
        X_DMA(sb_ + 16384 + 1024, pB, voB[1]);                                                                     \
      }                                                                                                            \
      pA += 64;                                                                                                    \
      pB += seg_stride; /* the next 8-wide K segment of the table */                                              \
      --c_segs;                                                                                                    \
      if (--c_left == 0 && c_segs > 0) X_NEXT_SIDE();                                                              \
    } else if (MINE_) { /* zeros: the odd half of the last chunk */                                                \
      X_DMA(sb_, zbase, voZ);                                                                                      \
      X_DMA(sb_ + 1024, zbase, voZ);                                                                               \
      X_DMA(sb_ + 16384, zbase, voZ);                                                                              \
      X_DMA(sb_ + 16384 + 1024, zbase, voZ);                                                                       \
    }                                                                                                              \
  } while (0)
  d4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  if (tot > 0) {
    X_NEXT_SIDE();
    X_ISSUE_HALF(0, 0, w < 4);
    X_ISSUE_HALF(0, 1, w < 4);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    double af[2][4], bf[2][NJ];
    X_FRAGS(0, 0, af[0], bf[0]);
    for (int ch = 0; ch < tot; ++ch) {
      const int slot = ch & 1;
      if (ch + 1 < tot) {  // (everybody left that slot at the barrier behind chunk ch - 1)
        const bool mine = (w >> 2) == ((ch + 1) & 1);
        X_ISSUE_HALF(slot ^ 1, 0, mine);
        X_ISSUE_HALF(slot ^ 1, 1, mine);
      }
#pragma unroll
      for (int kki = 0; kki < 4; ++kki) {
        const int pb = kki & 1;
        if (kki < 3) X_FRAGS(slot, kki + 1, af[pb ^ 1], bf[pb ^ 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][i], bf[pb][j], acc[i][j], 0, 0, 0);
      }
      {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // chunk ch + 1 is in LDS for everybody
      }
      if (ch + 1 < tot) X_FRAGS(slot ^ 1, 0, af[0], bf[0]);
    }
  }
#undef X_ISSUE_HALF
#undef X_DMA
#undef X_NEXT_SIDE
#undef X_FRAGS
#ifndef ROMHC_EPI_PRIO
#define ROMHC_EPI_PRIO 3
#endif
// Cache policy of the row stores (probe builds: -DX128_STORE_POLICY='" nt"' / '" sc1"').  Plain write-back stores it is: the
// pieces of a mesh row that neighbouring tiles write meet in L2 and leave it as whole lines -- streaming stores ("nt") cost
// 10 % of the kernel, write-through ("sc1", "sc0 sc1") 46 % -- although they would spare the 5.8 us every kernel boundary
// behind this kernel spends writing back the 32 MB of dirty lines it leaves in the eight L2s (profiles/r04_extend_lines_probes.txt)
#ifndef X128_STORE_POLICY
#define X128_STORE_POLICY ""
#endif
  // The epilogue runs at raised wave priority: while one workgroup of a CU stores and the other multiplies, neither
  // makes full progress (tools/mfma_store_overlap.hip, `waves`); letting the stores go first shortens that phase
  // (C2: 0.234 -> 0.220 ms; the other way round, priority to the k loop: 0.243 ms; priority to the prologue as
  // well: 0.237 ms; profiles/r02_extend128_wave_priority_ab.txt).
  __builtin_amdgcn_s_setprio(ROMHC_EPI_PRIO);
  if (threadIdx.x < 128) scs[threadIdx.x] = my_sc;
  __syncthreads();
  // epilogue: add the particular solution, swap between lane pairs so that every lane owns two adjacent vertices of
  // one 16-vertex block, 16-byte stores.  The stores of a wave walk down its 64 systems four rows at a time (rows
  // 16 i + 4 g + kq, i and g ascending): the lane's pointer advances by a constant, what a lane stores
  // where (16-byte pair / single first vertex / -- FLAT, a pair straddling two mesh rows -- single second vertex) is
  // decided once per tile as three lane masks, and the stores are issued under those masks without branches.
  const bool odd = lane & 1;
  constexpr int NHP = NJ / 2;
  char* sp[NHP];            // where the lane's next store of pair hp goes
  long long d1[NHP] = {};  // (FLAT) second vertex - first vertex, bytes
  unsigned long long k16[NHP], k8[NHP], k8b[NHP];
  {
    char* const rowp = reinterpret_cast<char*>(U) + size_t(row0 + m0 + wr * 64 + kq) * size_t(f.dim) * 8;
#pragma unroll
    for (int hp = 0; hp < NHP; ++hp) {  // pair hp of column blocks: (0,1) and (2,3); even lanes take the first, odd the second
      const int t = wc * (16 * NJ) + (2 * hp + (odd ? 1 : 0)) * 16 + fr - (odd ? 1 : 0);  // first of the two vertices, tile-local
      long long off0, off1;
      bool ok0, ok1;
      if (FLAT) {
        const int v = vt0 + t, i0 = v / n1, j0 = v - i0 * n1;
        off0 = (long long)(p * N + i0) * f.nc + q * N + j0;
        off1 = j0 + 1 < n1 ? off0 + 1 : (long long)(p * N + i0 + 1) * f.nc + q * N;
        ok0 = v < nvert;
        ok1 = v + 1 < nvert;
      } else {
        const int jcol = jv0 + t;  // 1-based
        off0 = (long long)(p * N + iv - 1) * f.nc + q * N - 1 + jcol;
        off1 = off0 + 1;
        ok0 = jcol <= n1;
        ok1 = jcol + 1 <= n1;
      }
      const bool straddle = FLAT && off1 != off0 + 1;  // the pair lies in two mesh rows
      k16[hp] = __builtin_amdgcn_ballot_w64(ok1 && !straddle);
      k8[hp] = __builtin_amdgcn_ballot_w64(ok0 && (!ok1 || straddle));
      k8b[hp] = __builtin_amdgcn_ballot_w64(ok1 && straddle);
      sp[hp] = rowp + off0 * 8;
      d1[hp] = (off1 - off0) * 8;
    }
  }
  const long long step = 4ll * f.dim * 8;          // four rows down
  const bool full = m0 + 128 <= Mc;                // all systems of the tile exist
  const int mrow = m0 + wr * 64 + kq;              // the lane's first system
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double sc = scs[wr * 64 + i * 16 + kq + 4 * g];
      unsigned long long in = ~0ull;
      if (!full) in = __builtin_amdgcn_ballot_w64(mrow + i * 16 + 4 * g < Mc);
#pragma unroll
      for (int hp = 0; hp < NHP; ++hp) {
        const double x0 = acc[i][2 * hp][g] + sc * w_own[2 * hp], x1 = acc[i][2 * hp + 1][g] + sc * w_own[2 * hp + 1];
        const double got = lane_swap1(odd ? x0 : x1);
        const double2_u pr = double2_u{odd ? got : x0, odd ? x1 : got};
        const unsigned long long m16 = x128_uniform(k16[hp] & in), m8 = x128_uniform(k8[hp] & in), m8b = x128_uniform(k8b[hp] & in);
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %2, %3, off" X128_STORE_POLICY "\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(m16), "v"(sp[hp]), "v"(pr));
        if (m8 != 0)  // (uniform: only the wave that holds the end of a mesh row has such lanes)
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx2 %2, %3, off" X128_STORE_POLICY "\n\ts_mov_b64 exec, %0"
                       : "=&s"(sv)
                       : "s"(m8), "v"(sp[hp]), "v"(pr.x));
        if (FLAT && m8b != 0)
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx2 %2, %3, off" X128_STORE_POLICY "\n\ts_mov_b64 exec, %0"
                       : "=&s"(sv)
                       : "s"(m8b), "v"(sp[hp] + d1[hp]), "v"(pr.y));
        sp[hp] += step;
      }
    }
}
template __global__ void k_extend128<false>(FemDev, X128Args, const double*, int, double*, long long, int, int);
template __global__ void k_extend128<true>(FemDev, X128Args, const double*, int, double*, long long, int, int);

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

