// fp64 MFMA tile engine shared by the kernels of libromhc (gfx950 / CDNA4, wave64).
//
// One 256-thread workgroup (4 waves, 2x2) owns a 64x64 fp64 output tile; each wave owns a 32x32
// quadrant as 2x2 v_mfma_f64_16x16x4_f64 accumulators.  Operands are staged through LDS in
// [64][BK=16] chunks (row stride LDK=18 doubles: conflict-free ds_read_b64 for the MFMA operand
// pattern lane -> (row = lane&15, k = lane>>4)), double buffered with a register prefetch so the
// global loads of chunk c+1 fly under the MFMAs of chunk c.
//
// Product computed:  acc[r][c] += sum_k A[r][k] * B[c][k]   ("NT": both operands K-contiguous)
//
// MFMA f64 16x16x4 operand/result maps (guide: cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15]
//   C/D: 4 doubles per lane, reg g -> (row = (l>>4) + 4*g, col = l&15)
#pragma once
#include <hip/hip_runtime.h>

#include "romhc_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

struct Acc {
  d4_t c[2][2];
};

__device__ inline void acc_zero(Acc& a) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) a.c[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
}

// LDS carve-up (doubles).  Staging: 2 buffers x {A,B} x 64 x LDK.
constexpr int STAGE_DOUBLES = 64 * LDK;                 // one operand chunk
constexpr int STAGE_TOTAL = 4 * STAGE_DOUBLES;          // 4608 doubles = 36,864 B
constexpr int TILE_DOUBLES = 64 * LDC;                  // 4224 doubles = 33,792 B

// wave coordinates inside the workgroup tile
struct WavePos {
  int lane, wr, wc;
  __device__ WavePos() {
    lane = threadIdx.x & 63;
    int w = threadIdx.x >> 6;
    wr = w >> 1;
    wc = w & 1;
  }
};

// row/col of accumulator element (i, j, g) inside the 64x64 tile
__device__ inline int acc_row(const WavePos& wp, int i, int g) { return wp.wr * 32 + i * 16 + (wp.lane >> 4) + 4 * g; }
__device__ inline int acc_col(const WavePos& wp, int j) { return wp.wc * 32 + j * 16 + (wp.lane & 15); }

// 16 MFMAs on one staged chunk (K = BK)
__device__ inline void mma_chunk(const double* __restrict__ sA, const double* __restrict__ sB, Acc& acc,
                                 const WavePos& wp) {
  const int r = wp.lane & 15, kq = wp.lane >> 4;
  const double* pa = sA + (wp.wr * 32 + r) * LDK + kq;
  const double* pb = sB + (wp.wc * 32 + r) * LDK + kq;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 4) {
    double a0 = pa[kk], a1 = pa[16 * LDK + kk];
    double b0 = pb[kk], b1 = pb[16 * LDK + kk];
    acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
    acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
    acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
    acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
  }
}

// Same, with the A operand read straight from a full [64][LDC] LDS tile at column offset k0.
__device__ inline void mma_chunk_Atile(const double* __restrict__ tileA, int k0, const double* __restrict__ sB,
                                       Acc& acc, const WavePos& wp) {
  const int r = wp.lane & 15, kq = wp.lane >> 4;
  const double* pa = tileA + (wp.wr * 32 + r) * LDC + k0 + kq;
  const double* pb = sB + (wp.wc * 32 + r) * LDK + kq;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 4) {
    double a0 = pa[kk], a1 = pa[16 * LDC + kk];
    double b0 = pb[kk], b1 = pb[16 * LDK + kk];
    acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
    acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
    acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
    acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
  }
}

// Thread t stages 4 consecutive k of row (t>>2): segment (t&3)*4.
__device__ inline int stage_row() { return threadIdx.x >> 2; }
__device__ inline int stage_seg() { return (threadIdx.x & 3) * 4; }

__device__ inline void stage_store(double* s, const double v[4]) {
  double* p = s + stage_row() * LDK + stage_seg();
  // LDK*8 = 144 B rows, seg*8 in {0,32,64,96}: 16-B aligned -> two ds_write_b128
  *reinterpret_cast<double2*>(p) = double2{v[0], v[1]};
  *reinterpret_cast<double2*>(p + 2) = double2{v[2], v[3]};
}

// 32-byte aligned global load of 4 doubles (pointer may be null -> zeros)
__device__ inline void load4_aligned(const double* __restrict__ p, double v[4]) {
  if (p) {
    double2 x = *reinterpret_cast<const double2*>(p);
    double2 y = *reinterpret_cast<const double2*>(p + 2);
    v[0] = x.x; v[1] = x.y; v[2] = y.x; v[3] = y.y;
  } else {
    v[0] = v[1] = v[2] = v[3] = 0.0;
  }
}

// Generic pipelined loop over `nchunks` K-chunks.  loadA(ch, v) / loadB(ch, v) fetch this thread's
// 4 doubles of chunk ch.  stA / stB each point at 2*STAGE_DOUBLES doubles of LDS (two buffers).
// `active(ch)` is a wave-uniform predicate: a wave whose quadrant does not need chunk ch skips the
// MFMAs (it still takes part in staging and barriers).  Ends with a barrier, so the staging area
// may be reused by the caller right after.
template <class FA, class FB, class FP>
__device__ inline void gemm_loop2(int nchunks, FA loadA, FB loadB, FP active, Acc& acc, double* stA, double* stB,
                                  const WavePos& wp) {
  if (nchunks <= 0) return;
  double va[4], vb[4];
  loadA(0, va);
  loadB(0, vb);
  for (int ch = 0; ch < nchunks; ++ch) {
    double* sA = stA + (ch & 1) * STAGE_DOUBLES;
    double* sB = stB + (ch & 1) * STAGE_DOUBLES;
    stage_store(sA, va);
    stage_store(sB, vb);
    __syncthreads();
    if (ch + 1 < nchunks) {
      loadA(ch + 1, va);
      loadB(ch + 1, vb);
    }
    if (active(ch)) mma_chunk(sA, sB, acc, wp);
    // the buffer written next iteration is the other one; the barrier of that iteration orders
    // its readers (iteration ch-1 ... already passed the barrier of iteration ch) -> no 2nd barrier
  }
  __syncthreads();
}

// Staging map for rows that are only 8-byte aligned (odd leading dimension, e.g. snapshot rows of odd
// length): lane -> k (t & 15), so that one wave-instruction reads four rows x 128 contiguous bytes;
// thread t handles rows (t >> 4) + 16 x, x = 0..3, of the 64-row chunk.
__device__ inline int kmajor_k() { return threadIdx.x & 15; }
__device__ inline int kmajor_row(int x) { return (threadIdx.x >> 4) + 16 * x; }
__device__ inline void stage_store_kmajor(double* s, const double v[4]) {
#pragma unroll
  for (int x = 0; x < 4; ++x) s[kmajor_row(x) * LDK + kmajor_k()] = v[x];
}

template <class FA, class FB>
__device__ inline void gemm_loop_kmajor(int nchunks, FA loadA, FB loadB, Acc& acc, double* stage, const WavePos& wp) {
  if (nchunks <= 0) return;
  double va[4], vb[4];
  loadA(0, va);
  loadB(0, vb);
  for (int ch = 0; ch < nchunks; ++ch) {
    double* sA = stage + (ch & 1) * STAGE_DOUBLES;
    double* sB = stage + 2 * STAGE_DOUBLES + (ch & 1) * STAGE_DOUBLES;
    stage_store_kmajor(sA, va);
    stage_store_kmajor(sB, vb);
    __syncthreads();
    if (ch + 1 < nchunks) {
      loadA(ch + 1, va);
      loadB(ch + 1, vb);
    }
    mma_chunk(sA, sB, acc, wp);
  }
  __syncthreads();
}

template <class FA, class FB>
__device__ inline void gemm_loop(int nchunks, FA loadA, FB loadB, Acc& acc, double* stage, const WavePos& wp) {
  gemm_loop2(nchunks, loadA, loadB, [](int) { return true; }, acc, stage, stage + 2 * STAGE_DOUBLES, wp);
}

// Variant: A operand is a resident [64][LDC] LDS tile (K = 64), B streamed through stB (2 buffers).
template <class FB, class FP>
__device__ inline void gemm_loop_Atile(const double* tileA, int nchunks, FB loadB, FP active, Acc& acc, double* stB,
                                       const WavePos& wp) {
  double vb[4];
  loadB(0, vb);
  for (int ch = 0; ch < nchunks; ++ch) {
    double* sB = stB + (ch & 1) * STAGE_DOUBLES;
    stage_store(sB, vb);
    __syncthreads();
    if (ch + 1 < nchunks) loadB(ch + 1, vb);
    if (active(ch)) mma_chunk_Atile(tileA, ch * BK, sB, acc, wp);
  }
  __syncthreads();
}
