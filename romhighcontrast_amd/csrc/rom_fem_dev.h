// Device-side view of a rom_fem, shared constants and the kernels of the sweep (gfx950 only).
// rom_fem_kernels.hip defines the kernels, rom_fem_setup.hip builds the tables (rom_fem_create),
// rom_fem_solve.hip enqueues them (rom_solve_batch and friends).
#pragma once
#include "rom_mma.h"

// ============================================================================================
// device-side view of a rom_fem
// ============================================================================================
struct FemDev {
  int nrb, ncb, N, n1, n1p, nr, nc, nGp, nGa, T, nslots, kblk, npre, nrhs, nexp, ncross, xb0;
  long long dim;
  const double* pool;  // 64x64 tables of the tile terms
  const int* alist;    // tile assembly as a stream (s_tile_to_lds): per tile slot and wave, pieces {element offset of a row pair's kilobyte in the pool, x | term << 8 | last << 16 | no-op << 17}
  const int* aoff;     // 4 * slots + 1 offsets into it (in pieces)
  int tile_stream;     // 0: tiles are assembled in registers (s_tile_load; ROMHC_NO_TILE_STREAM, and any tile of more than 128 terms)
  const int* pairs;    // single-tile solve: (term, block) pairs of the assembly, two ints each
  int npairs;          // multiple of 64 (no-op padded), followed by 64 more no-ops
  // k_solve1 (four systems per workgroup, the blocks dealt to its four waves): wave w walks pairs wp0[w] .. wp0[w + 1] - 1
  const double* pool_acc;  // their 16x16 table pieces, 256 doubles each, in accumulator layout [g pair][lane][2] (+ zero pieces behind the end)
  const int* wmeta;        // term | q << 8 | (last pair of block q) << 16 per pair (+ 128 no-ops behind the end)
  int wp0[5];
  int s1_t0, s1_nterm, s1_ndr;  // the single tile's terms and unknowns (desc[0], by value)
  const int* s1_items;    // flat records of the dense items: {group, position, nv, b0, b1, voff[4] + k, vblk[4], vu0[4], vu1[4], -} (24 ints)
  const int* s1_citems;   // ... of the coefficient items: {position | code << 28, source, b0, b1}; code 0 dense product's, 1 one over (a_b0 + a_b1), 2 copy, 3 zero
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
  const int* item_cf;   // k_coef: index of an entry among the closed-form entries (its dot products: 8 slots from 8 * item_cf), -1 for the others
  const int* ctask;     // k_coef: a flat record of 8 ints per dot product of the closed-form blocks: {matrix offset of the entry's column, source, length, row stride | vector entry or -1, u0, u1, slot}
  int nctask, ncf;
  double* gdots;        // k_coef: the tasks' dot products in global memory ([system][ncf * 8]) where they outgrow the LDS; else null
  int ncoef;
  const double* G;     // extension tables of the compressed edges: per table (n1*n1) x (rank+1 padded)
  const double* Gs;    // segment-major copies: [8-wide K segment][row][8], rows ordered so that the vertices of a mesh
                       // row are adjacent (sides 2, 3: the transposed table) -- one DMA instruction = 1 KB contiguous
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
  int n_lr;               // blocks in lr_blocks
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


FemDev make_dev(const rom_fem* f);

constexpr int COEF_MAX = 64;   // term weights cached in LDS per pass
#ifndef PAIR_RING_
#define PAIR_RING_ 8
#endif
constexpr int PAIR_RING = PAIR_RING_;  // (term, block) pairs in flight in the single-tile assembly
// k_solve1's dynamic LDS: the term weights of its four systems, four per-wave areas
constexpr int S1_COEF_BYTES = 4 * COEF_MAX * 8, S1_WAVE_BYTES = (40 * 64 + 64 * 4 + 64 + 64 + 3 * 64 * 4) * 8;
constexpr int S1_ITEM_PASSES = 4;  // coefficient items (64 per pass) whose descriptors k_solve1 reads ahead of its Cholesky
constexpr int S1_DENSE_BYTES = 64 * 64 * 8;  // the matrix of the dense product of the tail (64 x ndi, ndi <= 64)
constexpr int S1_LDS_BYTES = S1_COEF_BYTES + 4 * S1_WAVE_BYTES + S1_DENSE_BYTES;
constexpr int DENSE_GROUPS_MAX = 8;  // closed-form edges whose coefficient blocks k_solve1 builds

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

// ---- kernels (rom_fem_kernels.hip) ----------------------------------------------------------------
__global__ void k_repack_table(const double* __restrict__ G, int ld, int nseg, int n1, int orient, double* __restrict__ Gs);
__global__ void k_build_A0(double* A0, const double* Qp, const double* rho, int n1, int n1p, int N);
__global__ void k_rhs(FemDev f, const double* __restrict__ a);
__global__ void k_coef(FemDev f, const double* __restrict__ a);
__global__ void k_expand(FemDev f, const double* __restrict__ a, int Mc, double* __restrict__ U, long long row0);
__global__ void k_back_pre(FemDev f, const double* __restrict__ a, int Mc);
template <int NS>
__global__ void k_diag_update(FemDev f, const double* __restrict__ a, int slot, int Mc);
__global__ void k_diag_factor(FemDev f, int slot, int j);
__global__ void k_solve1(FemDev f, const double* __restrict__ a, int Mc);
template <int NS>
__global__ void k_factor_panel(FemDev f, const double* __restrict__ a, int j, int Mc);
__global__ void k_backsolve(FemDev f);
__global__ void k_edge_transform(FemDev f, int Mc);
__global__ void k_extend(FemDev f, const double* __restrict__ a, int Mc, double* __restrict__ U, long long row0, const int* __restrict__ blocks, int pw_log2);
// Descriptors of up to X128_BLOCKS blocks for one k_extend128 launch, passed BY VALUE: kernel arguments are read
// with scalar loads in one batch, whereas f.lr_blocks[z] -> f.sides[b] -> fields compiled into a chain of 14
// dependent vector loads (each with its own s_waitcnt vmcnt(0)) at the head of every workgroup.
constexpr int X128_BLOCKS = 16;
struct X128Args {
  int blocks[X128_BLOCKS];
  BlockSide sides[X128_BLOCKS];
};

template <bool FLAT>
__global__ void k_extend128(FemDev f, X128Args xa, const double* __restrict__ a, int Mc, double* __restrict__ U, long long row0, int with_expand, int sys_fast);
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
// snapshot rows directly (and to the nodal blocks of the interface vector).   grid (n1p/64, ceil(Mc/64), nexp [+ 1])
// one workgroup of the expansion: (64 nodes bx) x (64 systems by) of edge bz; bz == nexp: the slice that copies
// the interface values that need no expansion (cross points) from the interface vector to the snapshot rows
__device__ inline void expand_tile(const FemDev& f, int Mc, double* __restrict__ U, long long row0, double* lds,
                                   int bx, int by, int bz) {
  if (bz == f.nexp) {
    if (bx != 0) return;
    for (int idx = threadIdx.x; idx < 64 * f.nscat; idx += 256) {  // (four waves, whatever the launch has)
      const int m = by * 64 + idx / f.nscat, v = f.scat[idx % f.nscat];
      if (m < Mc) U[(row0 + m) * f.dim + f.vmap[v]] = f.y[size_t(m) * f.nGp + v];
    }
    return;
  }
  const WavePos wp;
  const ExpEdge ee = f.exp[bz];
  const int srow = stage_row(), sseg = stage_seg();
  const int mA = by * 64 + srow;
  const double* pA = mA < Mc ? f.y + size_t(mA) * f.nGp + ee.zpos + sseg : nullptr;
  const double* pB = f.P + (size_t(ee.ptab) * f.n1p + bx * 64 + srow) * f.n1p + sseg;
  Acc acc;
  acc_zero(acc);
  gemm_loop(
      ee.nch, [&](int ch, double* v) { load4_any(pA ? pA + ch * BK : nullptr, v); },
      [&](int ch, double* v) { load4_aligned(pB + ch * BK, v); }, acc, lds, wp);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = by * 64 + acc_row(wp, i, g);
      if (m >= Mc) continue;
      const double inv = f.y[size_t(m) * f.nGp + ee.spos];  // 1 / (a_b0 + a_b1), from the scalar block
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int node = bx * 64 + acc_col(wp, jb);
        const double v = node < f.n1 ? acc.c[i][jb][g] + f.vec[ee.p0off + node] * inv : 0.0;
        f.y[size_t(m) * f.nGp + ee.npos + node] = v;  // (read again by the node-by-node paths, if any)
        if (node < f.n1) U[(row0 + m) * f.dim + f.vmap[ee.npos + node]] = v;
      }
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

constexpr int EXT_ZERO_PAGE = 256;  // doubles of zeros behind FemDev::W (target of the lanes of k_extend128 that have nothing to load)
__global__ void k_scatter_interface(FemDev f, int Mc, double* __restrict__ U, long long row0);
__global__ void k_assemble_stencil(FemDev f, const double* __restrict__ a, int M, double* __restrict__ diag, double* __restrict__ east, double* __restrict__ north);
