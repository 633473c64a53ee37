// POD of a snapshot block behind the C-ABI (rom_pod / rom_pod_ex): the PCA fit inside ReducedBasisPCA.build
// (src/lib/ReducedBasis.py:189-200).  Randomised range finder passes over the implicitly deflated block -- thin products
// 2 b M dim each, never the M^2 dim of a Gram matrix -- with one power step per pass, orthonormalised on both sides of
// it; Rayleigh-Ritz on the collected subspace; every small dense problem runs on the device.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rom_mma.h"
#include "rom_ops.h"

#include "rom_basis_int.h"
#include "rom_small_dense.h"

// =====================================================================================================================
// POD (the PCA fit of ReducedBasisPCA.build, src/lib/ReducedBasis.py:189-200)
// =====================================================================================================================
// ---------------------------------------------------------------------------------------------------------------------
// Fused small kernels of the sketch passes.  A pass is four thin products over the snapshot block plus ~70 small dense
// operations on b <= 32 rows; launched one by one (Gram product, split-K reduction, mirror, factorisation, apply, copy
// back ...) they cost more than the products.  Two kernels replace most of them:
//   kp_combine_rows  OUT = T1 Y + alpha2 T2 V2 for row blocks of any length, in place when OUT == Y (applies a transform,
//                    subtracts a projection, lifts modes: no scratch block, no copy back)
//   kp_tall_svd      the Rayleigh-Ritz rounds on a b x M factor (Gram, Jacobi, rotation, accumulated rotation), one
//                    workgroup from the first round to the last
// ---------------------------------------------------------------------------------------------------------------------
// OUT (bo x ncols) = T1 (bo x b1) Y (b1 x ncols) + alpha2 T2 (bo x b2) V2 (b2 x ncols).  T1 == nullptr: the identity
// (bo == b1).  OUT may be Y: a thread owns a column and reads all of its inputs before it writes.  Coefficients are staged
// through LDS in chunks of KC input rows, [k][BO] so that the BO coefficients of an input row are one broadcast run.
template <int BO>
__global__ __launch_bounds__(256) void kp_combine_rows(int bo, int b1, int b2, const double* __restrict__ T1, int ldt1,
                                                       const double* __restrict__ T2, int ldt2, double alpha2,
                                                       const double* Y, long long ldy, const double* __restrict__ V2, long long ldv,
                                                       double* OUT, long long ldo, long long ncols) {
  constexpr int KC = 32;
  __shared__ double Tc[KC * BO];
  const long long j = blockIdx.x * 256LL + threadIdx.x;
  const bool on = j < ncols;
  double acc[BO];
#pragma unroll
  for (int i = 0; i < BO; ++i) acc[i] = 0.0;
  if (!T1) {
#pragma unroll
    for (int i = 0; i < BO; ++i)
      if (i < bo && on) acc[i] = Y[i * ldy + j];
  }
  const int ktot = (T1 ? b1 : 0) + b2;
  for (int k0 = 0; k0 < ktot; k0 += KC) {
    const int kc = min(KC, ktot - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < KC * BO; idx += 256) {
      const int k = idx / BO, i = idx - k * BO, kg = k0 + k;
      double v = 0.0;
      if (k < kc && i < bo) {
        if (T1 && kg < b1) v = T1[i * ldt1 + kg];
        else v = alpha2 * T2[i * ldt2 + (kg - (T1 ? b1 : 0))];
      }
      Tc[idx] = v;
    }
    __syncthreads();
    if (on) {
      // all input rows of the chunk in flight before the first product (a column's inputs are one latency, not kc of them)
      double x[KC];
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        const int kg = k0 + u;
        x[u] = 0.0;
        if (u < kc) x[u] = (T1 && kg < b1) ? Y[kg * ldy + j] : V2[(kg - (T1 ? b1 : 0)) * ldv + j];
      }
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        if (u < kc) {
          const double* tc = Tc + u * BO;
#pragma unroll
          for (int i = 0; i < BO; ++i) acc[i] += tc[i] * x[u];
        }
      }
    }
  }
  if (on) {
#pragma unroll
    for (int i = 0; i < BO; ++i)
      if (i < bo) OUT[i * ldo + j] = acc[i];
  }
}

// Rayleigh-Ritz rounds on a tall factor given TRANSPOSED, Tt (b x M, ld M, b <= 32), one workgroup of 256 threads:
// per round  H = Tt Tt^T (MFMA, one 16 x 16 output block per wave),  H = S^T diag(sig2) S (jacobi32_run),  Tt <- S Tt,
// Rt <- S Rt.  The first round rotates the rows towards the singular directions; from then on H is graded and nearly
// diagonal, where Jacobi resolves the small eigenvalues to high relative accuracy.  Out: Rt (b x b), sig2 (b), Tt rotated.
__global__ __launch_bounds__(256) void kp_tall_svd(int b, int M, double* __restrict__ Tt, int rounds, double* __restrict__ Rt,
                                                   double* __restrict__ sig2) {
  __shared__ Jacobi32Lds L;
  __shared__ double St[32 * J32_LD], Rs[32 * J32_LD], Rn[32 * J32_LD];
  __shared__ double Hp[4 * 3 * 256];   // partial H blocks of the four waves
  __shared__ double red[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int fr = lane & 15, kq = lane >> 4;
  for (int idx = t; idx < 32 * J32_LD; idx += 256) {
    Rs[idx] = 0.0;
    St[idx] = 0.0;   // (columns behind b stay zero: the rotation below runs over 32)
  }
  __syncthreads();
  if (t < b) Rs[t * J32_LD + t] = 1.0;
  for (int round = 0; round < rounds; ++round) {
    // H = Tt Tt^T: wave w takes the columns [w Mq, (w + 1) Mq) of Tt for the three 16 x 16 blocks (0,0), (1,0), (1,1) --
    // a quarter of the loads per wave of a block-per-wave split, 16 k-steps (32 loads) in flight; partial blocks are
    // added up over the waves in wave order
    {
      const int Mq = ((M + 3) / 4 + 3) / 4 * 4, kb0 = w * Mq, kb1 = min(M, kb0 + Mq);
      const bool v0 = fr < b, v1 = 16 + fr < b;
      const double* p0 = Tt + size_t(v0 ? fr : 0) * M + kq;
      const double* p1 = Tt + size_t(v1 ? 16 + fr : 0) * M + kq;
      d4_t h00 = d4_t{0.0, 0.0, 0.0, 0.0}, h10 = h00, h11 = h00;
      for (int k = kb0; k < kb1; k += 64) {
        double a0[16], a1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int kk = k + 4 * u + kq;
          a0[u] = (v0 && kk < kb1) ? p0[k + 4 * u] : 0.0;
          a1[u] = (v1 && kk < kb1) ? p1[k + 4 * u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          h00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a0[u], h00, 0, 0, 0);
          h10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], a0[u], h10, 0, 0, 0);
          h11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], a1[u], h11, 0, 0, 0);
        }
      }
      double* hp = Hp + w * (3 * 256);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int e = ((lane >> 4) + 4 * g) * 16 + (lane & 15);
        hp[e] = h00[g];
        hp[256 + e] = h10[g];
        hp[512 + e] = h11[g];
      }
    }
    __syncthreads();
    for (int idx = t; idx < 3 * 256; idx += 256) {
      const int blk = idx >> 8, e = idx & 255, i = e >> 4, j = e & 15;
      const double s = ((Hp[idx] + Hp[768 + idx]) + Hp[2 * 768 + idx]) + Hp[3 * 768 + idx];
      const int r = (blk == 0 ? 0 : 16) + i, c = (blk == 2 ? 16 : 0) + j;
      if (r < b && c < b) {
        L.As[r * J32_LD + c] = s;
        if (blk == 1) L.As[c * J32_LD + r] = s;
      }
    }
    __syncthreads();
    double dmax = 0.0;
    for (int idx = t; idx < b * b; idx += 256) {
      const int r = idx / b, c = idx - r * b;
      L.Vt[r * J32_LD + c] = r == c ? 1.0 : 0.0;
      if (r == c) dmax = fmax(dmax, fabs(L.As[r * J32_LD + r]));
    }
    // (symmetrise: the two triangles come from different waves with the same products in the same order -- equal bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
    if (lane == 0) red[w] = dmax;
    __syncthreads();
    dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const bool rotated = jacobi32_run<256>(b, L, 1, dmax);
    for (int idx = t; idx < b * b; idx += 256) {
      const int r = idx / b, c = idx - r * b;
      St[r * J32_LD + c] = L.Vt[L.perm[r] * J32_LD + c];
    }
    __syncthreads();
    // Rt <- S Rt
    for (int idx = t; idx < b * b; idx += 256) {
      const int r = idx / b, c = idx - r * b;
      double s = 0.0;
      for (int k = 0; k < b; ++k) s += St[r * J32_LD + k] * Rs[k * J32_LD + c];
      Rn[r * J32_LD + c] = s;
    }
    // Tt <- S Tt on the matrix cores: a wave owns 16-column blocks of Tt (all b rows of them: read before written), the
    // A operand (S, from LDS) is the same for every block and stays in registers, two blocks in flight
    {
      double sa[2][8];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) sa[rb][ks] = St[(rb * 16 + fr) * J32_LD + 4 * ks + kq];
      const int ncb = (M + 15) / 16;
      for (int cb = w; cb < ncb; cb += 8) {
        double xb[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = (cb + 4 * h) * 16 + fr;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const int k = 4 * ks + kq;
            xb[h][ks] = (k < b && j < M) ? Tt[size_t(k) * M + j] : 0.0;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          d4_t o0 = d4_t{0.0, 0.0, 0.0, 0.0}, o1 = o0;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[0][ks], xb[h][ks], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[1][ks], xb[h][ks], o1, 0, 0, 0);
          }
          const int j = (cb + 4 * h) * 16 + (lane & 15);
          if (j < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int r = (lane >> 4) + 4 * g;
              if (r < b) Tt[size_t(r) * M + j] = o0[g];
              if (16 + r < b) Tt[size_t(16 + r) * M + j] = o1[g];
            }
          }
        }
      }
    }
    __syncthreads();
    for (int idx = t; idx < b * b; idx += 256) {
      const int r = idx / b, c = idx - r * b;
      Rs[r * J32_LD + c] = Rn[r * J32_LD + c];
    }
    __syncthreads();
    // H was diagonal by Jacobi's own criterion (S = a sorting permutation, applied above): the rows ARE the singular
    // directions to working accuracy, and a further round would compute the same H and rotate nothing
    if (!rotated) break;
  }
  for (int idx = t; idx < b * b; idx += 256) {
    const int r = idx / b, c = idx - r * b;
    Rt[r * b + c] = Rs[r * J32_LD + c];
  }
  if (t < b) sig2[t] = L.ev[L.perm[t]];
}

// Sketch rows for a block that is still to be centred: rows 0 .. b - 2 of Om (b x M) lose their own mean -- a row with zero
// sum gives the same product with the block as with its centred rows, Om' X = Om' (X - 1 mu^T), and nothing of the sketch
// is lost: the columns of the centred block are orthogonal to the vector of ones -- and row b - 1 becomes 1 / M, so that the
// last row of Om X IS the column mean mu: the first product of the first pass replaces the pass over X that the means cost.
__global__ __launch_bounds__(256) void kp_zero_sum_rows(double* __restrict__ Om, int M, int b) {
  __shared__ double red[4];
  double* row = Om + size_t(blockIdx.x) * M;
  const int t = threadIdx.x;
  if (int(blockIdx.x) == b - 1) {
    for (int m = t; m < M; m += 256) row[m] = 1.0 / double(M);
    return;
  }
  double s = 0.0;
  for (int m = t; m < M; m += 256) s += row[m];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const double mean = (((red[0] + red[1]) + red[2]) + red[3]) / double(M);
  for (int m = t; m < M; m += 256) row[m] -= mean;
}

namespace {

// How far below its largest singular value a sketch pass accepts modes.  A pass orthonormalises its rows through their
// b x b Gram matrix (pivoted Cholesky); a row whose direction lies r = sigma_top / sigma below the top keeps eps r of the
// strong directions, and the product with the block multiplies that share by r again: the coefficient rows stay dominated
// by their own direction -- and their orthonormalisation in M space (sketch_pass) then removes the rest -- as long as
// eps r^2 < 1.  Measured on a spectrum of one mode per third of a decade (tools/dev/pod_synth.py): modes down to
// 2e-8 of the top of a pass come out at LAPACK's own noise bound eps sigma_1 / sigma, at 5e-9 the angle is 200 x that, at
// 5e-10 the mode is lost.  1e-7 keeps a factor 100 on eps r^2; what lies below is the next pass's.
constexpr double SKETCH_ACCEPT = 1e-7;
constexpr double NOISE_FLOOR = 1e-13;    // modes below this fraction of sigma_1 are fp64 noise of the snapshots
constexpr int PASS_MODES = 24;           // modes asked of one sketch pass (+ 8 rows of oversampling = 32: the one-workgroup small kernels)

struct PodInfo {
  int sketch_passes = 0, completed = 0, resolved = 0;
  double executed = 0.0;
};

// ---- launchers of the fused kernels --------------------------------------------------------------------------------
constexpr int FUSED_ROWS = 32;   // row blocks up to this size take the fused kernels (one Jacobi / Cholesky wave, LDS resident)

// OUT (bo x ncols) = T1 Y + alpha2 T2 V2 (see kp_combine_rows); bo <= 64
int combine_rows(rom_ctx* ctx, int bo, int b1, const double* T1, int ldt1, int b2, const double* T2, int ldt2, double alpha2,
                 const double* Y, int64_t ldy, const double* V2, int64_t ldv, double* OUT, int64_t ldo, int64_t ncols) {
  if (bo <= 0 || ncols <= 0) return ROM_OK;
  ROM_CHECK(bo <= 64, "combine_rows: %d output rows", bo);
  const unsigned grid = unsigned((ncols + 255) / 256);
  ROM_PROF(ctx, "combine_rows", 2.0 * bo * ((T1 ? b1 : 0) + b2) * double(ncols), 8.0 * double(ncols) * (bo + (T1 ? b1 : bo) + b2));
  if (bo <= 32) kp_combine_rows<32><<<grid, 256, 0, ctx->stream>>>(bo, b1, b2, T1, ldt1, T2, ldt2, alpha2, Y, ldy, V2, ldv, OUT, ldo, ncols);
  else kp_combine_rows<64><<<grid, 256, 0, ctx->stream>>>(bo, b1, b2, T1, ldt1, T2, ldt2, alpha2, Y, ldy, V2, ldv, OUT, ldo, ncols);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// rows of X (b x dim) <- orthonormal rows spanning its numerical row space, graded, zero rows behind the rank (the
// whitening of romb_gram_transform with SE_WHITEN): per round the Gram product + its split-K sum, the pivoted Cholesky,
// the transform in place (no scratch block, no copy back).  Tm: b x b, lam: b (device scratch of the caller).
// (Measured and dropped: ONE launch for Gram matrix + factorisation -- the last split-K workgroup to arrive adds up and
// factorises: 110-160 us against 16 + 5 + 40 for the three launches, the device-scope release of 450 workgroups is a
// write-back of an XCD's L2 each; the pivoted Cholesky as one wave without barriers: 45-100 us against 30-55 for the
// 256-thread kernel -- a lone wave pays every LDS round trip in full.)
int whiten_rows(rom_ctx* ctx, double* X, int b, int64_t dim, double rel_tol, int rounds, double* Tm, double* lam) {
  if (b <= 0) return ROM_OK;
  if (b > FUSED_ROWS) {
    Tmp scr;
    ROM_TRY(scr.get(ctx, size_t(b) * dim));
    return romb_gram_transform(ctx, X, scr, b, dim, SE_WHITEN, rel_tol, rounds);
  }
  Tmp G;
  ROM_TRY(G.get(ctx, size_t(b) * b));
  for (int r = 0; r < rounds; ++r) {
    ROM_TRY(rom_launch_gemm_nt(ctx, b, b, dim, 1.0, X, dim, X, dim, 0.0, G, b, "gram_small"));
    ROM_TRY(romb_pivchol_whiten(ctx, b, G, b, lam, Tm, b, r == 0 ? rel_tol : 1e-8));
    ROM_TRY(combine_rows(ctx, b, b, Tm, b, 0, nullptr, 0, 0.0, X, dim, nullptr, 0, X, dim, dim));
  }
  return ROM_OK;
}

// nearly orthonormal rows X (b x dim) <- (X X^T)^(-1/2) X in place; T_out (b x b, device): the transform that was applied
int lowdin_rows(rom_ctx* ctx, double* X, int b, int64_t dim, double* G, double* lam, double* T_out) {
  if (b <= 0) return ROM_OK;
  ROM_TRY(rom_launch_gram(ctx, b, dim, X, dim, G, b));
  ROM_TRY(romb_small_eig(ctx, b, G, b, lam, T_out, b, SE_LOWDIN, 1e-30));
  if (b <= 64) return combine_rows(ctx, b, b, T_out, b, 0, nullptr, 0, 0.0, X, dim, nullptr, 0, X, dim, dim);
  Tmp Y;
  ROM_TRY(Y.get(ctx, size_t(b) * dim));
  ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, b, 1.0, T_out, b, X, dim, 0.0, Y, dim));
  ROM_HIP(hipMemcpyAsync(X, Y.p(), size_t(b) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return ROM_OK;
}

// Right singular vectors / singular values of a tall factor given TRANSPOSED, Tt (b x M, ld M): Rayleigh-Ritz rounds on
// the b x b Gram matrix Tt Tt^T.  The first round rotates the rows towards the singular directions; from then on the
// Gram matrix is graded and nearly diagonal, where Jacobi resolves the small eigenvalues to high relative accuracy --
// nothing is lost to the squaring that a single eigen-decomposition of an ungraded Gram matrix would lose.
// Rt (b x b): accumulated rotation (rows = right singular vectors in the coordinates Tt came in); sig2: b values.
int tall_svd_rotation(rom_ctx* ctx, double* Tt, int b, int M, double* Rt, double* sig2, int rounds = 3) {
  if (b <= FUSED_ROWS && M <= 65536) {   // one workgroup from the first round to the last (b x M doubles: L2 resident)
    ROM_PROF(ctx, "tall_svd", rounds * (4.0 * b * b * M + 30.0 * b * b * b), 8.0 * rounds * 2.0 * b * M);
    kp_tall_svd<<<1, 256, 0, ctx->stream>>>(b, M, Tt, rounds, Rt, sig2);
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  Tmp H, St, T2, R2;
  ROM_TRY(H.get(ctx, size_t(b) * b));
  ROM_TRY(St.get(ctx, size_t(b) * b));
  ROM_TRY(T2.get(ctx, size_t(b) * M));
  ROM_TRY(R2.get(ctx, size_t(b) * b));
  for (int r = 0; r < rounds; ++r) {
    ROM_TRY(rom_launch_gemm_nt(ctx, b, b, M, 1.0, Tt, M, Tt, M, 0.0, H, b, "gemm_nt"));
    ROM_TRY(romb_small_eig(ctx, b, H, b, sig2, St, b, SE_EIG, 0.0));
    ROM_TRY(rom_launch_gemm_nn(ctx, b, M, b, 1.0, St, b, Tt, M, 0.0, T2, M));
    ROM_HIP(hipMemcpyAsync(Tt, T2.p(), size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (r == 0) {
      ROM_HIP(hipMemcpyAsync(Rt, St.p(), size_t(b) * b * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      ROM_TRY(rom_launch_gemm_nn(ctx, b, b, b, 1.0, St, b, Rt, b, 0.0, R2, b));
      ROM_HIP(hipMemcpyAsync(Rt, R2.p(), size_t(b) * b * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  return ROM_OK;
}

// The first product of a sketch pass, Q = Omega X (b x dim), does not depend on what the passes before it found -- only its
// rank-`found` correction does -- so it is started on a SECOND STREAM while the small dense problems in front of the pass
// (one-workgroup factorisations, Jacobi, a host read-back) leave the chip idle.  Same seeds, same kernels, same bits as
// the product enqueued in line; a guess of b that turns out wrong is discarded.
struct SketchAhead {
  rom_ctx* ctx = nullptr;
  Tmp Om, Q;
  int b = 0, seed = 0;
  bool pending = false;
  ~SketchAhead() {   // (an error return with the product still in flight: its buffers go back to the allocator only when it is done)
    if (pending && ctx && ctx->aux[0]) hipStreamSynchronize(ctx->aux[0]);
  }
};

int sketch_ahead_start(rom_ctx* ctx, SketchAhead& sa, const double* X, int M, int64_t dim, int b, int seed) {
  if (ctx->profile || !ctx->aux[0]) return ROM_OK;   // (per-kernel profiling keeps everything on one stream)
  sa.ctx = ctx;
  ROM_TRY(sa.Om.get(ctx, size_t(b) * M));
  ROM_TRY(sa.Q.get(ctx, size_t(b) * dim));
  ROM_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  ROM_HIP(hipStreamWaitEvent(ctx->aux[0], ctx->ev_fork, 0));
  hipStream_t main_stream = ctx->stream;
  ctx->stream = ctx->aux[0];
  int st = romb_fill_random(ctx, sa.Om, size_t(b) * M, 0xabcd0000ull + unsigned(seed) * 7919u, true);
  if (st == ROM_OK) st = rom_launch_gemm_nn(ctx, b, dim, M, 1.0, sa.Om, M, X, dim, 0.0, sa.Q, dim);
  ctx->stream = main_stream;
  ROM_TRY(st);
  ROM_HIP(hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
  sa.b = b;
  sa.seed = seed;
  sa.pending = true;
  return ROM_OK;
}

// the main stream waits for the product in flight (always, so that nothing is left running); true if it is the one asked for
int sketch_ahead_take(rom_ctx* ctx, SketchAhead& sa, int b, int seed, bool& hit) {
  hit = false;
  if (!sa.pending) return ROM_OK;
  ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[0], 0));
  sa.pending = false;
  hit = b >= 1 && sa.b >= b && sa.seed == seed;   // (the leading b rows of a larger product: the fill is a function of the index)
  return ROM_OK;
}

// One sketch pass over the DEFLATED remainder of the (M, dim) block X: a randomised range finder with one power iteration,
// thin products (2 b M dim flops each) instead of the 2 M^2 dim of a Gram matrix.
// The block is deflated IMPLICITLY: X_d = X - Bt^T V with the `found` modes accepted so far (V: found x dim, orthonormal
// rows; Bt: found x M, row j = X v_j).  Every product with X_d is the product with X followed by a rank-`found`
// correction -- X itself is never rewritten, which saves a read + write of the whole block per accepted batch of modes.
// The rounding error is what the explicit subtraction leaves in X_d as well: eps x sigma_1.
// Out: Q (b x dim): orthonormal rows spanning the sketch; Rt (b x b): rows = right singular vectors of Q X_d^T in Q's
// coordinates (mode i = row i of Rt Q); Traw (b x M) = Q X^T, the coefficients of the UNDEFLATED block (mode i's
// coefficient row X v_i = row i of Rt Traw: the caller's next deflation needs no pass over X); ss_host: b singular values.
template <class Hook>
int sketch_pass(rom_ctx* ctx, const double* X, int M, int64_t dim, const double* V, const double* Bt, int found, int b, int seed,
                double* Q, const double* Om_ready, double* Rt, double* Traw, std::vector<double>& ss_host, PodInfo& info,
                Hook before_rotation, int power = 1) {
  Tmp Om_own, Tt, Cc, Tm, lam, s2;
  if (!Om_ready) ROM_TRY(Om_own.get(ctx, size_t(b) * M));
  const double* Om = Om_ready ? Om_ready : Om_own.p();
  ROM_TRY(Tt.get(ctx, size_t(b) * M));
  ROM_TRY(Cc.get(ctx, size_t(b) * std::max(found, 1)));
  ROM_TRY(Tm.get(ctx, size_t(b) * b));
  ROM_TRY(lam.get(ctx, b));
  ROM_TRY(s2.get(ctx, b));
  const bool fused = found <= 512;   // (the correction kernel walks the found rows one by one: fine for any POD request)
  // out (b x dim) -= (left (b x M) Bt^T) V
  auto correct_rows = [&](double* out, const double* left) -> int {
    if (found == 0) return ROM_OK;
    ROM_TRY(rom_launch_gemm_nt(ctx, b, found, M, 1.0, left, M, Bt, M, 0.0, Cc, found, "gemm_nt"));
    if (fused && b <= 64) return combine_rows(ctx, b, b, nullptr, 0, found, Cc, found, -1.0, out, dim, V, dim, out, dim, dim);
    return rom_launch_gemm_nn(ctx, b, dim, found, -1.0, Cc, found, V, dim, 1.0, out, dim);
  };
  if (!Om_ready) {   // (else Q = Omega X is there already: started ahead on the second stream)
    ROM_TRY(romb_fill_random(ctx, Om_own, size_t(b) * M, 0xabcd0000ull + unsigned(seed) * 7919u, true));
    ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Om, M, X, dim, 0.0, Q, dim));               // Q = Omega X_d
  }
  ROM_TRY(correct_rows(Q, Om));
  info.executed += 2.0 * b * M * double(dim);
  double* Tdefl = Tt;
  for (int it = 0; it <= power; ++it) {
    // orthonormal rows (rank may drop: zero rows), one round of Cholesky whitening each time.  Before the power step the
    // rows only have to span the sketch and be aligned with its principal directions.  After it they are T' X_d with
    // orthonormal T': graded and nearly orthogonal, the case in which a pivoted Cholesky factor of their Gram matrix is
    // accurate relative to each row (its error follows the condition number of the SCALED matrix), so one round leaves
    // them orthonormal to ~1e-15 (measured: the same singular values, angles and orthonormality as two rounds on the C2
    // block and on spectra of 1 and 3 modes per decade over 13 orders -- tools/dev/pod_synth.py); the accepted modes are
    // orthonormalised again by the caller
    ROM_TRY(whiten_rows(ctx, Q, b, dim, 1e-26, 1, Tm, lam));
    ROM_TRY(rom_launch_gemm_nt(ctx, b, M, dim, 1.0, Q, dim, X, dim, 0.0, Traw, M, "gemm_nt"));    // Q X^T  (b, M)
    if (found) {                                                                                  // Q X_d^T = Q X^T - (Q V^T) Bt
      ROM_TRY(rom_launch_gemm_nt(ctx, b, found, dim, 1.0, Q, dim, V, dim, 0.0, Cc, found, "gemm_nt"));
      if (fused && b <= 64) {
        ROM_TRY(combine_rows(ctx, b, b, nullptr, 0, found, Cc, found, -1.0, Traw, M, Bt, M, Tt, M, M));
      } else {
        ROM_HIP(hipMemcpyAsync(Tt.p(), Traw, size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        ROM_TRY(rom_launch_gemm_nn(ctx, b, M, found, -1.0, Cc, found, Bt, M, 1.0, Tt, M));
      }
    } else {
      ROM_HIP(hipMemcpyAsync(Tt.p(), Traw, size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    info.executed += 2.0 * b * M * double(dim) + 4.0 * b * b * double(dim);
    if (it == power) break;
    // the coefficient rows orthonormalised in M space BEFORE the second product: a row dominated by its own direction (see
    // SKETCH_ACCEPT) is cleaned of the strong directions to eps here, so the second product leaves eps r of them instead of
    // the eps r^3 that limited a pass to four orders of magnitude -- one pass now reaches seven
    ROM_TRY(whiten_rows(ctx, Tdefl, b, M, 1e-26, 1, Tm, lam));
    ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Tdefl, M, X, dim, 0.0, Q, dim));              // Q X_d^T X_d (Q is rebuilt)
    ROM_TRY(correct_rows(Q, Tdefl));
    info.executed += 2.0 * b * M * double(dim);
  }
  ROM_TRY(before_rotation());   // (the last pass over the block is enqueued: what may run beside the small problems starts here)
  // X_d ~ T Q: the right singular vectors of the small factor rotate Q into the modes (Tt is rotated along, not used again)
  // (one round: the pass only has to separate its leading directions from the rest and to rank them for the accept rule --
  // the Rayleigh-Ritz step over ALL collected modes at the end of rom_pod_ex iterates to convergence)
  ROM_TRY(tall_svd_rotation(ctx, Tt, b, M, Rt, s2, 1));
  ss_host.resize(b);
  ROM_TRY(download(ctx, s2, ss_host.data(), b));
  for (double& v : ss_host) v = std::sqrt(std::max(v, 0.0));
  return ROM_OK;
}

}  // namespace

// Leading n right singular vectors / singular values of the (M, dim) block X (overwritten when it is centred).
// center != 0: subtract the column means first (sklearn PCA.fit).  V: (n, dim) rows = modes, sign convention of
// sklearn's svd_flip(u_based_decision=False); sigma_host: n singular values (0 for completed modes);
// info_host (8 doubles, may be null): resolved modes, completed modes, 0 (Gram passes: none), sketch passes, executed
// flops, 8 n M dim (the thin products for the requested modes alone), 0 (subspace iterations: none), stop reason (0 filled,
// 1 floor reached, 2 budget).
// rel_floor: modes with sigma <= rel_floor * sigma_1 are not looked for (<= 0 or below the fp64 noise floor of the snapshots,
// 1e-13: that floor -- what a full LAPACK SVD of the block resolves)
extern "C" int rom_pod_ex(rom_ctx* ctx, rom_buf* Xb, int64_t x_row0, int M, int64_t dim, int n, int center, double rel_floor,
                          rom_buf* Vb, int64_t v_row0, double* sigma_host, double* info_host) {
  ROM_CHECK(ctx && Xb && Vb && (sigma_host || n == 0), "rom_pod: null argument");
  const double floor_rel = rel_floor > NOISE_FLOOR ? rel_floor : NOISE_FLOOR;
  ROM_CHECK(M >= 1 && dim >= 1 && n >= 0 && x_row0 >= 0 && v_row0 >= 0, "rom_pod: bad sizes");
  ROM_CHECK(n <= std::min<int64_t>(M, dim), "rom_pod: %d modes requested from a %d x %lld block", n, M, (long long)dim);
  ROM_CHECK(n + 12 <= SE_MAX, "rom_pod: at most %d modes", SE_MAX - 12);
  ROM_CHECK(size_t(x_row0 + M) * dim <= Xb->n && size_t(v_row0 + n) * dim <= Vb->n, "rom_pod: buffers too small");
  double* X = Xb->p + x_row0 * dim;
  double* V = Vb->p + v_row0 * dim;
  PodInfo info;
  // centring: the column means come out of the first pass's first product (kp_zero_sum_rows) when there is a pass
  const bool mean_from_sketch = center && n > 0 && std::min<int64_t>(M, dim) >= 2;
  if (center && !mean_from_sketch) {
    Tmp mean;
    ROM_TRY(mean.get(ctx, dim));
    ROM_TRY(rom_launch_center_rows(ctx, X, M, dim, mean));
  }
  for (int i = 0; i < n; ++i) sigma_host[i] = 0.0;
  int found = 0;
  double sigma_1 = 0.0;
  Tmp Bt;  // coefficients of the accepted modes, (n, M): row j = X v_j
  ROM_TRY(Bt.get(ctx, size_t(std::max(n, 1)) * M));
  SketchAhead ahead;
  // (blocks from 64 MB: below, the product is shorter than the stream hand-over.  dim >= 1024, M >= 128: the product then
  // takes the thin LDS-DMA kernel, which needs no scratch -- the context's scratch area belongs to the kernels of the main stream)
  const bool worth_ahead = size_t(M) * dim * sizeof(double) >= (size_t(64) << 20) && dim >= 1024 && M >= 128;
  // the sketch passes run until the request is filled or the spectrum has reached the floor; the budget below only guards
  // against a pass that makes no progress (every pass accepts at least one mode or ends the loop)
  const int passes = n + 2;
  bool at_floor_stop = false;
  for (int p = 1; p < passes; ++p) {
    if (found >= n) break;
    std::vector<double> ss;
    // a pass accepts modes over seven orders of magnitude (SKETCH_ACCEPT) -- two dozen of them in a spectrum that decays
    // like the snapshot blocks' do -- so it asks for at most PASS_MODES (+ 8 of oversampling): the thin products scale with
    // b, the small dense problems with b^3, and 32 rows is what the one-workgroup kernels hold
    // (a request with hundreds of modes left asks for more per pass: 24 per pass would be n / 24 passes over the block)
    const int left = n - found;
    const int want = std::min(left, std::max(PASS_MODES, left / 4));
    int b = int(std::min<int64_t>(std::min<int64_t>(M, dim), want + 8));
    Tmp Q, Om, Rt, Traw;
    bool hit = false;
    ROM_TRY(sketch_ahead_take(ctx, ahead, b, p, hit));
    if (hit) {   // Q = Omega X of this pass is already there
      std::swap(Q.b, ahead.Q.b);
      std::swap(Om.b, ahead.Om.b);
    } else {
      ROM_TRY(Q.get(ctx, size_t(b) * dim));
    }
    if (p == 1 && mean_from_sketch) {
      // the first product on the block as it came, with zero-sum sketch rows and a row of 1 / M: its last row is the column
      // mean, which is then subtracted from the block (the one pass over X that the centring still costs); the pass goes on
      // with the b - 1 sketch rows.  (The product sees the uncentred values: its rounding is eps x THEIR size -- a sketch
      // only has to span the range, and every later product of the pass works on the centred block)
      ROM_TRY(Om.get(ctx, size_t(b) * M));
      ROM_TRY(romb_fill_random(ctx, Om, size_t(b) * M, 0xabcd0000ull + unsigned(p) * 7919u, true));
      kp_zero_sum_rows<<<b, 256, 0, ctx->stream>>>(Om, M, b);
      ROM_HIP(hipGetLastError());
      ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Om, M, X, dim, 0.0, Q, dim));
      ROM_TRY(rom_launch_subtract_row(ctx, X, M, dim, Q.p() + size_t(b - 1) * dim));
      hit = true;
      b -= 1;
    }
    ROM_TRY(Rt.get(ctx, size_t(b) * b));
    ROM_TRY(Traw.get(ctx, size_t(b) * M));
    // the next pass's product beside this pass's Rayleigh-Ritz rounds, when its b is certain whatever this pass accepts:
    // 16 <= modes still wanted afterwards <= 64  =>  16 + 8 rows
    // (a next pass is certain -- unless the floor is reached -- when this one cannot fill the request; it asks for at most
    // PASS_MODES + 8 rows as long as fewer than 4 PASS_MODES modes are left)
    auto start_next = [&]() -> int {
      if (left - want >= 1 && left - 1 <= 4 * PASS_MODES && worth_ahead)
        return sketch_ahead_start(ctx, ahead, X, M, dim, int(std::min<int64_t>(std::min<int64_t>(M, dim), PASS_MODES + 8)), p + 1);
      return ROM_OK;
    };
    ROM_TRY(sketch_pass(ctx, X, M, dim, V, Bt, found, b, p, Q, hit ? Om.p() : nullptr, Rt, Traw, ss, info, start_next));
    info.sketch_passes += 1;
    if (found == 0) sigma_1 = ss.empty() ? 0.0 : ss[0];
    int take = 0;
    while (take < std::min(b, want) && ss[take] > SKETCH_ACCEPT * ss[0] && ss[take] > floor_rel * sigma_1) ++take;
    if (take == 0) {
      at_floor_stop = b == 0 || ss[0] <= floor_rel * sigma_1;
      break;
    }
    double* Vn = V + size_t(found) * dim;
    double* Bn = Bt.p() + size_t(found) * M;
    // the accepted modes = the first `take` rows of Rt Q, written where they belong; their coefficient rows X v_i = Rt Traw
    if (b <= 64 && take <= 64) {
      ROM_TRY(combine_rows(ctx, take, b, Rt, b, 0, nullptr, 0, 0.0, Q, dim, nullptr, 0, Vn, dim, dim));
      ROM_TRY(combine_rows(ctx, take, b, Rt, b, 0, nullptr, 0, 0.0, Traw, M, nullptr, 0, Bn, M, M));
    } else {
      ROM_TRY(rom_launch_gemm_nn(ctx, take, dim, b, 1.0, Rt, b, Q, dim, 0.0, Vn, dim));
      ROM_TRY(rom_launch_gemm_nn(ctx, take, M, b, 1.0, Rt, b, Traw, M, 0.0, Bn, M));
    }
    info.executed += 2.0 * take * b * double(dim);
    // orthogonal to the earlier modes (block Gram-Schmidt, twice), orthonormal among themselves (symmetric
    // orthonormalisation); every step is a small linear map of the new rows and of the old ones, so the coefficient rows
    // follow by the same maps in M space: X (v - C V_old)^T = X v^T - C Bt_old -- no pass over the block for the deflation
    {
      Tmp C, Gs, ls, Ts;
      ROM_TRY(C.get(ctx, size_t(take) * std::max(found, 1)));
      ROM_TRY(Gs.get(ctx, size_t(take) * take));
      ROM_TRY(ls.get(ctx, take));
      ROM_TRY(Ts.get(ctx, size_t(take) * take));
      for (int r = 0; r < 2 && found > 0; ++r) {
        ROM_TRY(rom_launch_gemm_nt(ctx, take, found, dim, 1.0, Vn, dim, V, dim, 0.0, C, found, "gemm_nt"));
        if (take <= 64) {
          ROM_TRY(combine_rows(ctx, take, take, nullptr, 0, found, C, found, -1.0, Vn, dim, V, dim, Vn, dim, dim));
          ROM_TRY(combine_rows(ctx, take, take, nullptr, 0, found, C, found, -1.0, Bn, M, Bt, M, Bn, M, M));
        } else {
          ROM_TRY(rom_launch_gemm_nn(ctx, take, dim, found, -1.0, C, found, V, dim, 1.0, Vn, dim));
          ROM_TRY(rom_launch_gemm_nn(ctx, take, M, found, -1.0, C, found, Bt, M, 1.0, Bn, M));
        }
      }
      ROM_TRY(lowdin_rows(ctx, Vn, take, dim, Gs, ls, Ts));
      if (take <= 64) {
        ROM_TRY(combine_rows(ctx, take, take, Ts, take, 0, nullptr, 0, 0.0, Bn, M, nullptr, 0, Bn, M, M));
      } else {
        Tmp B2;
        ROM_TRY(B2.get(ctx, size_t(take) * M));
        ROM_TRY(rom_launch_gemm_nn(ctx, take, M, take, 1.0, Ts, take, Bn, M, 0.0, B2, M));
        ROM_HIP(hipMemcpyAsync(Bn, B2.p(), size_t(take) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      }
    }
    const bool at_floor = take < b && ss[take] <= floor_rel * sigma_1;
    found += take;
    if (at_floor) {  // the spectrum has reached the floor: nothing left to find
      at_floor_stop = true;
      break;
    }
  }
  {
    bool unused = false;
    ROM_TRY(sketch_ahead_take(ctx, ahead, -1, -1, unused));   // (a product started for a pass that did not happen)
  }
  if (found) {
    // Rayleigh-Ritz on the collected subspace: X ~ B V  ->  the SVD of B orders / rotates the modes
    Tmp Rt, s2, Vr;
    ROM_TRY(Rt.get(ctx, size_t(found) * found));
    ROM_TRY(s2.get(ctx, found));
    ROM_TRY(tall_svd_rotation(ctx, Bt, found, M, Rt, s2));
    if (found <= 64) {
      ROM_TRY(combine_rows(ctx, found, found, Rt, found, 0, nullptr, 0, 0.0, V, dim, nullptr, 0, V, dim, dim));   // in place
    } else {
      ROM_TRY(Vr.get(ctx, size_t(found) * dim));
      ROM_TRY(rom_launch_gemm_nn(ctx, found, dim, found, 1.0, Rt, found, V, dim, 0.0, Vr, dim));
      ROM_HIP(hipMemcpyAsync(V, Vr.p(), size_t(found) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    info.executed += 2.0 * found * found * double(dim);
    std::vector<double> s(found);
    ROM_TRY(download(ctx, s2, s.data(), found));
    for (int i = 0; i < found; ++i) sigma_host[i] = std::sqrt(std::max(s[i], 0.0));
  }
  if (found < n) {
    // complete the basis: random directions orthonormalised against the modes; they carry no variance (LAPACK and
    // scikit-learn return SOME orthonormal directions there too).  Seeded by the count of resolved modes: deterministic.
    const int rest = n - found;
    if (rest <= FUSED_ROWS && found <= 512) {   // (the fused row kernels; same seed as rom_complete_orthonormal)
      double* Vn = V + size_t(found) * dim;
      Tmp C, Tm, lam;
      ROM_TRY(C.get(ctx, size_t(rest) * std::max(found, 1)));
      ROM_TRY(Tm.get(ctx, size_t(rest) * rest));
      ROM_TRY(lam.get(ctx, rest));
      ROM_TRY(romb_fill_random(ctx, Vn, size_t(rest) * dim, 0xc0de0000ull + unsigned(found), false));
      for (int r = 0; r < 2 && found > 0; ++r) {
        ROM_TRY(rom_launch_gemm_nt(ctx, rest, found, dim, 1.0, Vn, dim, V, dim, 0.0, C, found, "gemm_nt"));
        ROM_TRY(combine_rows(ctx, rest, rest, nullptr, 0, found, C, found, -1.0, Vn, dim, V, dim, Vn, dim, dim));
      }
      ROM_TRY(whiten_rows(ctx, Vn, rest, dim, 1e-26, 1, Tm, lam));
    } else {
      ROM_TRY(rom_complete_orthonormal(ctx, Vb, v_row0, found, rest, dim));
    }
    info.completed = rest;
  }
  info.resolved = found;
  if (n > 0) ROM_TRY(rom_launch_rows_sign_flip(ctx, V, n, dim));  // svd_flip(u_based_decision=False)
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (info_host) {
    info_host[0] = info.resolved;
    info_host[1] = info.completed;
    info_host[2] = 0.0;   // (Gram passes: none since round 5 -- the slot keeps its place in the ABI)
    info_host[3] = info.sketch_passes;
    info_host[4] = info.executed;
    info_host[5] = 8.0 * n * M * double(dim);   // the four thin products of a pass for the n requested modes alone (no oversampling)
    info_host[6] = 0.0;   // (subspace iterations of the Gram stage: retired with it)
    // why the call stopped short of n modes: 0 request filled, 1 the spectrum reached the floor (the completed modes are
    // not determined by the data), 2 no accepted mode in a pass / pass budget (modes above the floor may be missing)
    info_host[7] = found >= n ? 0.0 : (at_floor_stop || sigma_1 == 0.0 ? 1.0 : 2.0);
  }
  return ROM_OK;
}

extern "C" int rom_pod(rom_ctx* ctx, rom_buf* Xb, int64_t x_row0, int M, int64_t dim, int n, int center, rom_buf* Vb,
                       int64_t v_row0, double* sigma_host, double* info_host) {
  return rom_pod_ex(ctx, Xb, x_row0, M, dim, n, center, 0.0, Vb, v_row0, sigma_host, info_host);
}
