// POD of a snapshot block behind the C-ABI (rom_pod / rom_pod_ex): the PCA fit inside ReducedBasisPCA.build
// (src/lib/ReducedBasis.py:189-200).  Randomised range finder passes over the implicitly deflated block -- thin products
// 2 b M dim each -- with one power step per pass, orthonormalised on both sides of it, and a convergence rule per pass;
// the Gram route (M x M Gram matrix on MFMA, eigenpairs iterated in M space) when the first pass shows a spectrum that
// decays too slowly for the passes; Rayleigh-Ritz on the collected subspace; every small dense problem runs on the device.
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
// =====================================================================================================================
// The Gram route (slowly decaying spectra): Gram matrix on MFMA, leading eigenpairs by a pivoted-Cholesky low-rank factor or
// by subspace iteration IN M SPACE -- an iteration costs 2 M^2 b flops there instead of two passes over the block
// =====================================================================================================================
__global__ void kb_next_block(double* __restrict__ Zs, const double* __restrict__ Zr, const double* __restrict__ Yr,
                              const double* __restrict__ theta, long long M) {
  // rows: the rotated power step G y_i / theta_i for the resolvable pairs, the Ritz vector itself at the noise floor
  const double th = theta[blockIdx.y], t0 = fabs(theta[0]);
  const bool ok = th > 1e-13 * t0;
  const double a = ok ? 1.0 / th : 0.0;
  const long long o = blockIdx.y * M;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < M; j += (long long)gridDim.x * blockDim.x)
    Zs[o + j] = ok ? a * Zr[o + j] : Yr[o + j];
}

// out[i] = 1 / x[i] (0 where x[i] is not positive)
__global__ void kb_inv(const double* __restrict__ x, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] > 0.0 ? 1.0 / x[i] : 0.0;
}



// ---------------------------------------------------------------------------------------------------------------------
// Low-rank factor of a symmetric PSD matrix by diagonally pivoted Cholesky, ONE workgroup of 1024 threads:
//   G (M x M) ~ Lt^T Lt,  Lt (rcap x M; row k = column k of the factor), stopped at the first pivot <= tol x the first one
// (rank r) or at r = rcap.  A snapshot Gram matrix whose spectrum falls below the fp64 resolution of its entries within a few
// dozen directions -- the block of a low-dimensional parameter sweep -- is REPRODUCED by that factor to rounding: the
// trace of what is left of the diagonal bounds ||G - Lt^T Lt||_2, and the eigenpairs of G follow from the r x r problem
// Lt Lt^T with no subspace iteration.  Every step is one row of G (contiguous) and the rows of Lt so far (L2 resident).
// `slow` (checked every 16 steps): the decay so far extrapolates to more than rcap steps -- give up early.
// out[0] = r, out[1] = trace of the remaining diagonal (>= 0), out[2] = first pivot, out[3] = 1 if stopped by tol.
// dwork: M doubles (remaining diagonal).
__global__ __launch_bounds__(1024) void kp_pivchol_lowrank(int M, const double* __restrict__ G, long long ldg, int rcap, double tol,
                                                           double* __restrict__ Lt, double* __restrict__ dwork,
                                                           double* __restrict__ out) {
  __shared__ double red_v[16];
  __shared__ int red_i[16];
  __shared__ double s_best;
  __shared__ int s_piv;
  const int t = threadIdx.x;
  for (int i = t; i < M; i += 1024) dwork[i] = G[size_t(i) * ldg + i];
  __syncthreads();
  double first = 0.0;
  int r = 0, by_tol = 0;
  for (int k = 0; k < rcap; ++k) {
    double best = -1e300;
    int at = 0x7fffffff;
    for (int i = t; i < M; i += 1024) {
      const double v = dwork[i];
      if (v > best) { best = v; at = i; }   // ascending i per thread: first maximum
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_down(best, o, 64);
      const int oa = __shfl_down(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((t & 63) == 0) { red_v[t >> 6] = best; red_i[t >> 6] = at; }
    __syncthreads();
    if (t == 0) {
      double bb = red_v[0];
      int ba = red_i[0];
      for (int w = 1; w < 16; ++w)
        if (red_v[w] > bb || (red_v[w] == bb && red_i[w] < ba)) { bb = red_v[w]; ba = red_i[w]; }
      s_best = bb;
      s_piv = ba;
    }
    __syncthreads();
    const double piv = s_best;
    const int p = s_piv;
    if (k == 0) first = piv;
    if (!(piv > tol * first) || !(piv > 0.0)) { by_tol = 1; break; }
    // (slow decay: after k steps the pivots have fallen by piv / first; at that rate tol is more than rcap steps away)
    if ((k & 15) == 0 && k >= 16 && log(piv / first) * double(rcap) > log(tol) * double(k)) break;
    const double s = 1.0 / sqrt(piv);
    for (int i = t; i < M; i += 1024) {
      double c = G[size_t(p) * ldg + i];
      // (the pivot's own entries Lt[j][p] are wave-uniform loads issued with the column's: one latency, no LDS stage)
      for (int j = 0; j < k; ++j) c -= Lt[size_t(j) * M + i] * Lt[size_t(j) * M + p];
      c = (i == p) ? sqrt(piv) : c * s;
      Lt[size_t(k) * M + i] = c;
      dwork[i] = (i == p) ? -1e300 : dwork[i] - c * c;
    }
    r = k + 1;
    __syncthreads();
  }
  // what is left of the diagonal (pivots excluded; rounding may leave entries slightly negative)
  double tr = 0.0;
  for (int i = t; i < M; i += 1024) {
    const double v = dwork[i];
    if (v > 0.0) tr += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tr += __shfl_down(tr, o, 64);
  __syncthreads();
  if ((t & 63) == 0) red_v[t >> 6] = tr;
  __syncthreads();
  if (t == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red_v[w];
    out[0] = double(r);
    out[1] = s;
    out[2] = first;
    out[3] = double(by_tol);
  }
}

// Eigen-decomposition of the leading n x n block of A (nmax x nmax, nmax <= 32) with n read FROM THE DEVICE (n_dev[0], the
// rank a kernel in front of this one has just found): the host need not know n to launch it.  lam (nmax) and T (nmax x nmax)
// are zero behind n.
__global__ __launch_bounds__(256) void kp_jacobi32_devn(int nmax, const double* __restrict__ n_dev, const double* __restrict__ A,
                                                        double* __restrict__ lam, double* __restrict__ T) {
  __shared__ Jacobi32Lds L;
  __shared__ double red[4];
  const int t = threadIdx.x, n = max(0, min(nmax, int(n_dev[0])));
  for (int idx = t; idx < nmax * nmax; idx += 256) T[idx] = 0.0;
  if (t < nmax) lam[t] = 0.0;
  if (n == 0) return;
  double dmax = 0.0;
  for (int idx = t; idx < n * n; idx += 256) {
    const int r = idx / n, c = idx - r * n;
    const double v = 0.5 * (A[size_t(r) * nmax + c] + A[size_t(c) * nmax + r]);
    L.As[r * J32_LD + c] = v;
    L.Vt[r * J32_LD + c] = r == c ? 1.0 : 0.0;
    if (r == c) dmax = fmax(dmax, fabs(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
  if ((t & 63) == 0) red[t >> 6] = dmax;
  __syncthreads();
  dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  jacobi32_run<256>(n, L, 1, dmax);
  __syncthreads();   // (the zero fill of T above is complete for everybody)
  if (t < n) lam[t] = L.ev[L.perm[t]];
  for (int idx = t; idx < n * n; idx += 256) {
    const int r = idx / n, c = idx - r * n;
    T[size_t(r) * nmax + c] = L.Vt[L.perm[r] * J32_LD + c];
  }
}

// rows of W (r x M) scaled by 1 / sqrt(lam_i) where lam_i > floor_rel * lam_0, zeroed elsewhere
__global__ void kp_scale_eigvec_rows(double* __restrict__ W, long long M, const double* __restrict__ lam, double floor_rel) {
  const double l = lam[blockIdx.y], l0 = lam[0];
  const double a = (l > floor_rel * l0 && l > 0.0) ? 1.0 / sqrt(l) : 0.0;
  double* row = W + blockIdx.y * M;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < M; j += (long long)gridDim.x * blockDim.x) row[j] *= a;
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused small kernels of the sketch passes.  A pass is four thin products over the snapshot block plus ~70 small dense
// operations on b <= 32 rows; launched one by one (Gram product, split-K reduction, mirror, factorisation, apply, copy
// back ...) they cost more than the products.  Two kernels replace most of them:
//   kp_combine_rows_mma  OUT = T1 Y + alpha2 T2 V2 for row blocks of any length, in place when OUT == Y (applies a
//                    transform, subtracts a projection, lifts modes: no scratch block, no copy back)
//   kp_tall_svd      the Rayleigh-Ritz rounds on a b x M factor (Gram, Jacobi, rotation, accumulated rotation), one
//                    workgroup from the first round to the last
// ---------------------------------------------------------------------------------------------------------------------
// OUT (bo x ncols) = T1 (bo x b1) Y (b1 x ncols) + alpha2 T2 (bo x b2) V2 (b2 x ncols).  T1 == nullptr: the identity
// (bo == b1).  OUT may be Y: a wave owns 64 columns -- four 16-column blocks -- and reads every input row of them before it
// writes one.  On the matrix cores: the coefficients of a 32-row chunk of inputs are the wave's A fragments (from LDS, once
// per chunk), the inputs its B fragments (lane = (column, k mod 4): 16 columns x 4 input rows per load instruction, all 32
// loads of a chunk in flight before the first product), 8 NRB MFMAs per column block and chunk; NRB = ceil(bo / 16) row
// blocks of outputs.  (The first form of this kernel was a thread per column on the vector pipe: bo x b products per column
// on a wave that is alone on its SIMD and pays ~9 cycles per instruction it issues, a third of them LDS reads of
// coefficients -- 15-17 us for 32 x 32 x 65 025; this one 12.  A form with the coefficients as scalar operands, read
// through the scalar cache, took 45: 512 s_loads per chunk.)
template <int NRB>
__global__ __launch_bounds__(256) void kp_combine_rows_mma(int bo, int b1, int b2, const double* __restrict__ T1, int ldt1,
                                                           const double* __restrict__ T2, int ldt2, double alpha2,
                                                           const double* Y, long long ldy, const double* __restrict__ V2,
                                                           long long ldv, double* OUT, long long ldo, long long ncols) {
  constexpr int KC = 32, LDT = KC + 1;
  __shared__ double Tc[NRB * 16 * LDT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, kq = lane >> 4;
  const long long c0 = blockIdx.x * 256LL + w * 64;
  d4_t acc[NRB][4];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc[rb][cb] = d4_t{0.0, 0.0, 0.0, 0.0};
      if (!T1) {
        const long long col = c0 + cb * 16 + fr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = rb * 16 + kq + 4 * g;
          if (row < bo && col < ncols) acc[rb][cb][g] = Y[row * ldy + col];
        }
      }
    }
  const int n1 = T1 ? b1 : 0, ktot = n1 + b2;
  for (int k0 = 0; k0 < ktot; k0 += KC) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < NRB * 16 * KC; idx += 256) {
      const int i = idx / KC, k = idx - i * KC, kg = k0 + k;
      double v = 0.0;
      if (i < bo && kg < ktot) v = kg < n1 ? T1[i * ldt1 + kg] : alpha2 * T2[i * ldt2 + (kg - n1)];
      Tc[i * LDT + k] = v;
    }
    __syncthreads();
    double xb[4][8];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const long long col = c0 + cb * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kg = k0 + 4 * ks + kq;
        xb[cb][ks] = 0.0;
        if (kg < ktot && col < ncols) xb[cb][ks] = kg < n1 ? Y[kg * ldy + col] : V2[(kg - n1) * ldv + col];
      }
    }
    double sa[NRB][8];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) sa[rb][ks] = Tc[(rb * 16 + fr) * LDT + 4 * ks + kq];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[rb][ks], xb[cb][ks], acc[rb][cb], 0, 0, 0);
  }
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const long long col = c0 + cb * 16 + fr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = rb * 16 + kq + 4 * g;
        if (row < bo && col < ncols) OUT[row * ldo + col] = acc[rb][cb][g];
      }
    }
}


// Rayleigh-Ritz rounds on a tall factor given TRANSPOSED, Tt (b x M, ld M, b <= 32), one workgroup of 256 threads:
// per round  H = Tt Tt^T (MFMA, one 16 x 16 output block per wave),  H = S^T diag(sig2) S (jacobi32_run),  Tt <- S Tt,
// Rt <- S Rt.  The first round rotates the rows towards the singular directions; from then on H is graded and nearly
// diagonal, where Jacobi resolves the small eigenvalues to high relative accuracy.  Out: Rt (b x b), sig2 (b); Tt is scratch:
// rotated by every round but the last one executed (a one-round call does not write it).
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
    // A operand (S, from LDS) is the same for every block and stays in registers, two blocks in flight.  Only for the sake
    // of a further round: the last one leaves Tt as it found it (the caller gets the rotation in Rt; a one-round call does
    // not write Tt at all)
    if (rotated && round + 1 < rounds) {
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

// out[0] = bits of the largest |x| among the finite entries, out[1] = number of entries that are not finite
__global__ void kp_block_amax(const double* __restrict__ X, size_t count, unsigned long long* __restrict__ out) {
  unsigned long long best = 0ull, bad = 0ull;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
    const double v = fabs(X[i]);
    if (v <= 1.7976931348623157e308) best = max(best, (unsigned long long)__double_as_longlong(v));   // (positive doubles order like their bits)
    else ++bad;
  }
  atomicMax(&out[0], best);
  if (bad) atomicAdd(&out[1], bad);
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
  int gram_passes = 0, sketch_passes = 0, completed = 0, resolved = 0, eig_iterations = 0, lowrank = 0, full_eig = 0, unconverged = 0;
  double executed = 0.0;
};

constexpr double GRAM_ACCEPT = 1e-10;    // eigenvalues of a Gram matrix are taken down to this fraction of its largest one
constexpr int LOWRANK_CAP = 96;          // most steps of the pivoted Cholesky that replaces the subspace iteration
constexpr double LOWRANK_TOL = 1e-14;    // its stopping pivot, relative to the first one
constexpr double LOWRANK_RESIDUAL = 2e-14;  // accepted ||G - L L^T|| (bounded by the trace of the remaining diagonal) / lambda_1

// The leading eigenpairs of a numerically low-rank PSD matrix without iteration: G ~ Lt^T Lt by pivoted Cholesky (rank r,
// one launch), the r x r problem Lt Lt^T = Q diag(theta) Q^T, eigenvectors w_i = Lt^T q_i / sqrt(theta_i).  The error
// against the eigenpairs of G itself is that of a perturbation of norm <= trace(remaining diagonal), which the kernel
// reports; the caller falls back to the subspace iteration when that bound is above LOWRANK_RESIDUAL x theta_0 or the
// factor did not end within LOWRANK_CAP steps (a slowly decaying spectrum).  done = 1 on success.
int lowrank_eigenpairs(rom_ctx* ctx, const double* G, int M, int nev, double* W, std::vector<double>& theta_host, double* theta_dev,
                       PodInfo& info, int& done) {
  done = 0;
  const int rcap = std::min(M, LOWRANK_CAP);
  if (M <= rcap) return ROM_OK;  // (the full space: one exact Ritz step of the general path)
  Tmp Lt, dw, out, H, St, lam, Wr;
  ROM_TRY(dw.get(ctx, M));
  ROM_TRY(out.get(ctx, 4));
  {
    // First a factor of at most 32 steps with everything behind it enqueued BEFORE the host knows the rank (the small
    // eigenproblem reads it from the device): ONE host synchronisation for the whole Gram stage when the block's Gram
    // matrix has numerical rank <= 32 -- a sweep over a handful of parameters -- instead of two round trips.
    constexpr int R32 = 32;
    ROM_TRY(Lt.get(ctx, size_t(R32) * M));
    ROM_TRY(H.get(ctx, size_t(R32) * R32));
    ROM_TRY(St.get(ctx, size_t(R32) * R32));
    ROM_TRY(lam.get(ctx, R32));
    ROM_TRY(Wr.get(ctx, size_t(R32) * M));
    ROM_HIP(hipMemsetAsync(Lt.p(), 0, size_t(R32) * M * sizeof(double), ctx->stream));
    {
      ROM_PROF(ctx, "pivchol_lowrank", double(R32) * R32 * M, 8.0 * R32 * M);
      kp_pivchol_lowrank<<<1, 1024, 0, ctx->stream>>>(M, G, M, R32, LOWRANK_TOL, Lt, dw, out);
    }
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_gemm_nt(ctx, R32, R32, M, 1.0, Lt, M, Lt, M, 0.0, H, R32, "gemm_nt"));
    {
      ROM_PROF(ctx, "small_eig", 30.0 * R32 * R32 * R32, 16.0 * R32 * R32);
      kp_jacobi32_devn<<<1, 256, 0, ctx->stream>>>(R32, out, H, lam, St);
    }
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_gemm_nn(ctx, R32, M, R32, 1.0, St, R32, Lt, M, 0.0, Wr, M));
    kp_scale_eigvec_rows<<<dim3(unsigned(std::min((M + 255) / 256, 64)), R32), 256, 0, ctx->stream>>>(Wr, M, lam, 0.0);
    ROM_HIP(hipGetLastError());
    double o[4 + R32];
    ROM_HIP(hipMemcpyAsync(o, out.p(), 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ROM_HIP(hipMemcpyAsync(o + 4, lam.p(), R32 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    const int r = int(o[0]);
    if (r < 1) return ROM_OK;
    if (o[3] != 0.0 && o[4] > 0.0 && o[1] <= LOWRANK_RESIDUAL * o[4]) {
      const int ncopy = std::min(nev, r);
      ROM_HIP(hipMemcpyAsync(W, Wr.p(), size_t(ncopy) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      if (ncopy < nev) ROM_HIP(hipMemsetAsync(W + size_t(ncopy) * M, 0, size_t(nev - ncopy) * M * sizeof(double), ctx->stream));
      ROM_HIP(hipMemsetAsync(theta_dev, 0, size_t(nev) * sizeof(double), ctx->stream));
      ROM_HIP(hipMemcpyAsync(theta_dev, lam.p(), size_t(ncopy) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      theta_host.assign(nev, 0.0);
      for (int i = 0; i < ncopy; ++i) theta_host[i] = std::max(o[4 + i], 0.0);
      info.eig_iterations = 0;
      info.lowrank = r;
      done = 1;
      return ROM_OK;
    }
    if (o[3] != 0.0) return ROM_OK;   // (ended by tolerance but the residual bound failed: the general path)
  }
  // (rank above 32: the factor again up to LOWRANK_CAP steps, sizes known to the host before the small problems are launched)
  ROM_TRY(Lt.get(ctx, size_t(rcap) * M));
  {
    ROM_PROF(ctx, "pivchol_lowrank", double(rcap) * rcap * M, 8.0 * rcap * M);
    kp_pivchol_lowrank<<<1, 1024, 0, ctx->stream>>>(M, G, M, rcap, LOWRANK_TOL, Lt, dw, out);
  }
  ROM_HIP(hipGetLastError());
  double o[4];
  ROM_TRY(download(ctx, out, o, 4));
  const int r = int(o[0]);
  if (r < 1 || o[3] == 0.0) return ROM_OK;   // nothing there, or not low rank within the cap
  ROM_TRY(H.get(ctx, size_t(r) * r));
  ROM_TRY(St.get(ctx, size_t(r) * r));
  ROM_TRY(lam.get(ctx, r));
  ROM_TRY(Wr.get(ctx, size_t(r) * M));
  ROM_TRY(rom_launch_gemm_nt(ctx, r, r, M, 1.0, Lt, M, Lt, M, 0.0, H, r, "gemm_nt"));
  ROM_TRY(romb_small_eig(ctx, r, H, r, lam, St, r, SE_EIG, 0.0, true));   // (graded: the factor's rows fall off like the pivots)
  std::vector<double> th(r);
  ROM_TRY(download(ctx, lam, th.data(), r));
  if (!(th[0] > 0.0) || o[1] > LOWRANK_RESIDUAL * th[0]) return ROM_OK;
  ROM_TRY(rom_launch_gemm_nn(ctx, r, M, r, 1.0, St, r, Lt, M, 0.0, Wr, M));
  const int ncopy = std::min(nev, r);
  kp_scale_eigvec_rows<<<dim3(unsigned(std::min((M + 255) / 256, 64)), ncopy), 256, 0, ctx->stream>>>(Wr, M, lam, 0.0);
  ROM_HIP(hipGetLastError());
  ROM_HIP(hipMemcpyAsync(W, Wr.p(), size_t(ncopy) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (ncopy < nev) ROM_HIP(hipMemsetAsync(W + size_t(ncopy) * M, 0, size_t(nev - ncopy) * M * sizeof(double), ctx->stream));
  ROM_HIP(hipMemsetAsync(theta_dev, 0, size_t(nev) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemcpyAsync(theta_dev, lam.p(), size_t(ncopy) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  theta_host.assign(nev, 0.0);
  for (int i = 0; i < ncopy; ++i) theta_host[i] = std::max(th[i], 0.0);
  info.eig_iterations = 0;
  info.lowrank = r;
  done = 1;
  return ROM_OK;
}

// Leading nev eigenpairs of the symmetric PSD matrix G (M x M) by subspace iteration with Rayleigh-Ritz; the projected
// b x b problems are solved on the device.  theta_host / theta_dev: nev values (host / device); W: (nev, M) rows = eigenvectors.
int top_eigenpairs(rom_ctx* ctx, const double* G, int M, int nev, double* W, std::vector<double>& theta_host, double* theta_dev,
                   PodInfo& info,
                   int oversample = 12, double tol = 2e-14, int max_iter = 300, double accept = GRAM_ACCEPT) {
  {
    int done = 0;
    ROM_TRY(lowrank_eigenpairs(ctx, G, M, nev, W, theta_host, theta_dev, info, done));
    if (done) return ROM_OK;
  }
  // the whole Gram matrix diagonalised at once (Jacobi: the eigenvalues to the accuracy of the matrix entries whatever the
  // gaps): when the request is most of the spectrum -- an iteration would diagonalise a block of nearly that order EVERY
  // step -- and as the last resort of an iteration that did not converge (below)
  auto full_eig = [&]() -> int {
    Tmp lamf, Tf;
    ROM_TRY(lamf.get(ctx, M));
    ROM_TRY(Tf.get(ctx, size_t(M) * M));
    ROM_TRY(romb_small_eig(ctx, M, G, M, lamf, Tf, M, SE_EIG, 0.0, true));
    std::vector<double> lf(M);
    ROM_TRY(download(ctx, lamf, lf.data(), M));
    theta_host.assign(lf.begin(), lf.begin() + nev);
    ROM_HIP(hipMemcpyAsync(theta_dev, lamf.p(), size_t(nev) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipMemcpyAsync(W, Tf.p(), size_t(nev) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));   // (Tf / lamf go back to the pool when this scope ends)
    info.full_eig = 1;
    return ROM_OK;
  };
  if (M <= SE_GRID_MAX && 2 * (nev + oversample) >= M) return full_eig();
  const int b0 = std::min(M, nev + oversample);
  int b = b0;  // rows in play: shrinks once the spectrum shows how many pairs the caller can use (see below)
  Tmp Y, Z, H, St, lam, Yr, Zr, Res, res, Zs, scr, nrm;
  ROM_TRY(Y.get(ctx, size_t(b) * M));
  ROM_TRY(Z.get(ctx, size_t(b) * M));
  ROM_TRY(H.get(ctx, size_t(b) * b));
  ROM_TRY(St.get(ctx, size_t(b) * b));
  ROM_TRY(lam.get(ctx, 2 * size_t(b)));
  ROM_TRY(Yr.get(ctx, size_t(b) * M));
  ROM_TRY(Zr.get(ctx, size_t(b) * M));
  ROM_TRY(Res.get(ctx, size_t(b) * M));
  ROM_TRY(Zs.get(ctx, size_t(b) * M));
  ROM_TRY(scr.get(ctx, size_t(b) * M));
  ROM_TRY(nrm.get(ctx, b));
  double* d_res = lam.p() + b0;
  ROM_TRY(romb_fill_random(ctx, Y, size_t(b) * M, 0x5eed0000ull + unsigned(b) * 131u + unsigned(M), true));
  std::vector<double> th(2 * size_t(b0), 0.0);
  double best = 1e300, worst = 0.0;
  int stall = 0;
  std::vector<double> hist;   // the residual after every step
  if (b == M) {
    ROM_TRY(romb_gram_transform(ctx, Y, scr, b, M, SE_WHITEN, 1e-30, 2));  // (the full space: one exact Ritz step below)
  } else {
    // Iteration 0 is a plain power step: Y_1 = orthonormalised rows of (Gaussian block) G.  A Rayleigh-Ritz step on a
    // random block only rotates noise -- it costs a b x b eigenproblem and an orthonormalisation of the start block and
    // leaves the same subspace.
    ROM_TRY(rom_launch_gemm_nt(ctx, b, M, M, 1.0, Y, M, G, M, 0.0, Zs, M, "gemm_nt"));
    ROM_TRY(rom_launch_l2norm(ctx, Zs, b, M, nrm, true));
    kb_inv<<<unsigned((b + 255) / 256), 256, 0, ctx->stream>>>(nrm, nrm, b);
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_rows_scale(ctx, Zs, b, M, nrm));
    ROM_TRY(romb_gram_transform(ctx, Zs, scr, b, M, SE_WHITEN, 1e-30, 2));
    ROM_HIP(hipMemcpyAsync(Y.p(), Zs.p(), size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  for (int it = 0; it < max_iter; ++it) {
    ROM_TRY(rom_launch_gemm_nt(ctx, b, M, M, 1.0, Y, M, G, M, 0.0, Z, M, "gemm_nt"));   // Z = Y G (G symmetric)
    ROM_TRY(rom_launch_gemm_nt(ctx, b, b, M, 1.0, Z, M, Y, M, 0.0, H, b, "gemm_nt"));   // H = Y G Y^T
    ROM_TRY(romb_small_eig(ctx, b, H, b, lam, St, b, SE_EIG, 0.0, false));                   // rows of St: Ritz rotations
    ROM_TRY(rom_launch_gemm_nn(ctx, b, M, b, 1.0, St, b, Y, M, 0.0, Yr, M));            // Ritz vectors
    info.eig_iterations = it + 1;
    if (b == M) {  // full space: exact after one Ritz step
      ROM_TRY(download(ctx, lam, th.data(), b));
      break;
    }
    ROM_TRY(rom_launch_gemm_nn(ctx, b, M, b, 1.0, St, b, Z, M, 0.0, Zr, M));            // G applied to them
    kb_rows_axpy<<<dim3(unsigned(std::min((M + 255) / 256, 64)), b), 256, 0, ctx->stream>>>(Res, Zr, Yr, lam, -1.0, M);
    ROM_HIP(hipGetLastError());
    const int ncheck = std::min(nev, b);
    ROM_TRY(rom_launch_l2norm(ctx, Res, ncheck, M, d_res, true));
    ROM_TRY(download(ctx, lam, th.data(), size_t(b0) + ncheck));
    for (int i = b; i < b0; ++i) th[i] = 0.0;  // (pairs dropped from the block)
    const double t0 = std::max(std::fabs(th[0]), 1e-300);
    worst = 0.0;
    for (int i = 0; i < ncheck; ++i)
      if (th[i] > accept * std::fabs(th[0])) worst = std::max(worst, th[b0 + i] / t0);
    if (worst < 0.7 * best) { best = worst; stall = 0; } else { ++stall; }
    // (three steps without a gain of 30 % end the iteration only close to its tolerance -- the residual has reached the
    // rounding level of the Gram matrix; further out a slow gain is a flat spectrum, (lambda_{b+1} / lambda_k) per step, and
    // the iteration goes on: 300 steps cost less than one diagonalisation of the whole matrix)
    if (worst <= tol || (stall >= 3 && worst <= 1e-10) || it == max_iter - 1) break;
    // (a matrix that can be diagonalised whole -- 0.3 s at 900 rows, 0.85 s at 2048 -- is not iterated on in vain: after a dozen
    // steps the gain per step of the last six says how many more the tolerance would take; more than the budget, or steps that
    // each diagonalise a block too large for the one-workgroup kernel, and the iteration ends here)
    hist.push_back(worst);
    if (M <= SE_GRID_MAX && it >= 11 && worst > 1e-10) {
      const double rate = std::pow(worst / hist[hist.size() - 7], 1.0 / 6.0);
      const double more = rate < 1.0 ? std::log(1e-10 / worst) / std::log(rate) : 1e9;
      if (it + more > max_iter || (b > SE_LDS_MAX && more > 20)) break;
    }
    // The caller only takes pairs with theta_i > accept * theta_0.  Once a Rayleigh-Ritz step on an orthonormal block
    // has shown how many there can be (two orders of magnitude of slack on the threshold), the block is cut down to
    // those + the oversampling: the rows are Ritz vectors in descending order, so the cut keeps the leading ones.
    // (A snapshot block with 16 usable pairs out of 50 requested then iterates with 28 rows instead of 62.)
    {
      int count = 0;
      while (count < b && th[count] > 1e-2 * accept * std::fabs(th[0])) ++count;
      b = std::min(b, std::min(nev, count) + oversample);
    }
    kb_next_block<<<dim3(unsigned(std::min((M + 255) / 256, 64)), b), 256, 0, ctx->stream>>>(Zs, Zr, Yr, lam, M);
    ROM_HIP(hipGetLastError());
    ROM_TRY(romb_gram_transform(ctx, Zs, scr, b, M, SE_WHITEN, 1e-30, 1));  // (rows are rotated Ritz vectors: nearly orthonormal)
    ROM_HIP(hipMemcpyAsync(Y.p(), Zs.p(), size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (b < M && worst > 1e-10) {
    // The iteration ran out of steps far from its tolerance (a spectrum so flat that 300 steps of (lambda_{b+1} / lambda_k)
    // do not get there: a large block of independent random rows).  Up to SE_GRID_MAX rows the whole matrix is diagonalised
    // instead; beyond, the caller is told (info.unconverged -> stop reason 2).
    if (M <= SE_GRID_MAX) return full_eig();
    info.unconverged = 1;
  }
  theta_host.assign(th.begin(), th.begin() + nev);
  for (int i = b; i < nev; ++i) theta_host[i] = 0.0;
  const int ncopy = std::min(nev, b);
  ROM_HIP(hipMemsetAsync(theta_dev, 0, size_t(nev) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemcpyAsync(theta_dev, lam.p(), size_t(ncopy) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  ROM_HIP(hipMemcpyAsync(W, Yr.p(), size_t(ncopy) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (ncopy < nev) ROM_HIP(hipMemsetAsync(W + size_t(ncopy) * M, 0, size_t(nev - ncopy) * M * sizeof(double), ctx->stream));
  return ROM_OK;
}


// ---- launchers of the fused kernels --------------------------------------------------------------------------------
constexpr int FUSED_ROWS = 32;   // row blocks up to this size take the fused kernels (one Jacobi / Cholesky wave, LDS resident)

// OUT (bo x ncols) = T1 Y + alpha2 T2 V2 (see kp_combine_rows_mma); bo <= 64
int combine_rows(rom_ctx* ctx, int bo, int b1, const double* T1, int ldt1, int b2, const double* T2, int ldt2, double alpha2,
                 const double* Y, int64_t ldy, const double* V2, int64_t ldv, double* OUT, int64_t ldo, int64_t ncols) {
  if (bo <= 0 || ncols <= 0) return ROM_OK;
  ROM_CHECK(bo <= 64, "combine_rows: %d output rows", bo);
  const unsigned grid = unsigned((ncols + 255) / 256);
  ROM_PROF(ctx, "combine_rows", 2.0 * bo * ((T1 ? b1 : 0) + b2) * double(ncols), 8.0 * double(ncols) * (bo + (T1 ? b1 : bo) + b2));
  if (bo <= 16) {
    kp_combine_rows_mma<1><<<grid, 256, 0, ctx->stream>>>(bo, b1, b2, T1, ldt1, T2, ldt2, alpha2, Y, ldy, V2, ldv, OUT, ldo, ncols);
  } else if (bo <= 32) {
    kp_combine_rows_mma<2><<<grid, 256, 0, ctx->stream>>>(bo, b1, b2, T1, ldt1, T2, ldt2, alpha2, Y, ldy, V2, ldv, OUT, ldo, ncols);
  } else {
    kp_combine_rows_mma<4><<<grid, 256, 0, ctx->stream>>>(bo, b1, b2, T1, ldt1, T2, ldt2, alpha2, Y, ldy, V2, ldv, OUT, ldo, ncols);
  }
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
// (whether the one-workgroup kernel takes the call: it rotates Tt for the sake of a further round only -- a one-round call
// leaves Tt as it was; the parallel path rotates it in every round)
static bool tall_svd_is_fused(int b, int M) { return b <= FUSED_ROWS && M <= 4096; }
int tall_svd_rotation(rom_ctx* ctx, double* Tt, int b, int M, double* Rt, double* sig2, int rounds = 3) {
  if (tall_svd_is_fused(b, M)) {   // one workgroup from the first round to the last (beyond: its reads of the b x M factor, ~80 GB/s for ONE workgroup, outweigh the launches saved)
    ROM_PROF(ctx, "tall_svd", rounds * (4.0 * b * b * M + 30.0 * b * b * b), 8.0 * rounds * 2.0 * b * M);
    kp_tall_svd<<<1, 256, 0, ctx->stream>>>(b, M, Tt, rounds, Rt, sig2);
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  Tmp H, St, T2, R2;
  ROM_TRY(H.get(ctx, size_t(b) * b));
  ROM_TRY(St.get(ctx, size_t(b) * b));
  if (b > 64) ROM_TRY(T2.get(ctx, size_t(b) * M));
  ROM_TRY(R2.get(ctx, size_t(b) * b));
  for (int r = 0; r < rounds; ++r) {
    ROM_TRY(rom_launch_gemm_nt(ctx, b, b, M, 1.0, Tt, M, Tt, M, 0.0, H, b, "gemm_nt"));
    ROM_TRY(romb_small_eig(ctx, b, H, b, sig2, St, b, SE_EIG, 0.0));
    if (b <= 64) {
      ROM_TRY(combine_rows(ctx, b, b, St, b, 0, nullptr, 0, 0.0, Tt, M, nullptr, 0, Tt, M, M));   // Tt <- S Tt in place
    } else {
      ROM_TRY(rom_launch_gemm_nn(ctx, b, M, b, 1.0, St, b, Tt, M, 0.0, T2, M));
      ROM_HIP(hipMemcpyAsync(Tt, T2.p(), size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
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
// `pilot` (first pass only; returns true to abandon the pass -- ss_host then comes back empty): called after the first two
// products with estimates of the sketch's singular values, the pivots of the M-space orthonormalisation.
template <class Hook, class Pilot>
int sketch_pass(rom_ctx* ctx, const double* X, int M, int64_t dim, const double* V, const double* Bt, int found, int b, int seed,
                double* Q, const double* Om_ready, double* Rt, double* Traw, std::vector<double>& ss_host, PodInfo& info,
                Hook before_rotation, int power, Pilot pilot) {
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
    // Q X^T (b, M).  Without deflation it IS the factor of the step: before the power step it goes straight to Tt (nobody
    // needs the undeflated copy of that step); at the end to Traw, which the one-workgroup Rayleigh-Ritz kernel reads without
    // writing (one round), so that no copy is needed there either
    const bool last = it == power;
    double* Tout = (!found && !last) ? Tt.p() : Traw;
    ROM_TRY(rom_launch_gemm_nt(ctx, b, M, dim, 1.0, Q, dim, X, dim, 0.0, Tout, M, "gemm_nt"));
    if (found) {                                                                                  // Q X_d^T = Q X^T - (Q V^T) Bt
      ROM_TRY(rom_launch_gemm_nt(ctx, b, found, dim, 1.0, Q, dim, V, dim, 0.0, Cc, found, "gemm_nt"));
      if (fused && b <= 64) {
        ROM_TRY(combine_rows(ctx, b, b, nullptr, 0, found, Cc, found, -1.0, Traw, M, Bt, M, Tt, M, M));
      } else {
        ROM_HIP(hipMemcpyAsync(Tt.p(), Traw, size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        ROM_TRY(rom_launch_gemm_nn(ctx, b, M, found, -1.0, Cc, found, Bt, M, 1.0, Tt, M));
      }
    } else if (last && !tall_svd_is_fused(b, M)) {
      ROM_HIP(hipMemcpyAsync(Tt.p(), Traw, size_t(b) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    info.executed += 2.0 * b * M * double(dim) + 4.0 * b * b * double(dim);
    if (last) break;
    // the coefficient rows orthonormalised in M space BEFORE the second product: a row dominated by its own direction (see
    // SKETCH_ACCEPT) is cleaned of the strong directions to eps here, so the second product leaves eps r of them instead of
    // the eps r^3 that limited a pass to four orders of magnitude -- one pass now reaches seven
    ROM_TRY(whiten_rows(ctx, Tdefl, b, M, 1e-26, 1, Tm, lam));
    if (it == 0) {
      // (lam: the squared pivots of that orthonormalisation, in pivot order = the squared norms of the coefficient rows with
      // the stronger rows taken out: sigma_k^2 of the sketch to a small factor, before any power step)
      std::vector<double> est(b);
      bool abandon = false;
      ROM_TRY(pilot(lam, est, abandon));
      if (abandon) {
        ss_host.clear();
        return ROM_OK;
      }
    }
    ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Tdefl, M, X, dim, 0.0, Q, dim));              // Q X_d^T X_d (Q is rebuilt)
    ROM_TRY(correct_rows(Q, Tdefl));
    info.executed += 2.0 * b * M * double(dim);
  }
  ROM_TRY(before_rotation());   // (the last pass over the block is enqueued: what may run beside the small problems starts here)
  // X_d ~ T Q: the right singular vectors of the small factor rotate Q into the modes
  // (one round: the pass only has to separate its leading directions from the rest and to rank them for the accept rule --
  // the Rayleigh-Ritz step over ALL collected modes at the end of rom_pod_ex iterates to convergence)
  ROM_TRY(tall_svd_rotation(ctx, (!found && tall_svd_is_fused(b, M)) ? Traw : Tt.p(), b, M, Rt, s2, 1));
  ss_host.resize(b);
  ROM_TRY(download(ctx, s2, ss_host.data(), b));
  for (double& v : ss_host) v = std::sqrt(std::max(v, 0.0));
  return ROM_OK;
}

}  // namespace

// Leading n right singular vectors / singular values of the (M, dim) block X (overwritten when it is centred).
// center != 0: subtract the column means first (sklearn PCA.fit).  V: (n, dim) rows = modes, sign convention of
// sklearn's svd_flip(u_based_decision=False); sigma_host: n singular values (0 for completed modes);
// info_host (8 doubles, may be null): resolved modes, completed modes, Gram passes (0 / 1), sketch passes, executed flops,
// 8 n M dim (the thin products for the requested modes alone), subspace iterations of the Gram route, stop reason
// (0 filled, 1 floor reached, 2 budget).
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
  auto deflate = [&](int lo, int take) -> int {
    // coefficients of the modes V[lo : lo + take] into Bt (row j = X v_j; the modes are orthogonal to the earlier ones, so
    // X and the deflated block give the same coefficients).  X is not touched: the deflation is implicit (sketch_pass)
    ROM_TRY(rom_launch_gemm_nt(ctx, take, M, dim, 1.0, V + size_t(lo) * dim, dim, X, dim, 0.0, Bt.p() + size_t(lo) * M, M, "gemm_nt"));
    info.executed += 2.0 * take * M * double(dim);
    return ROM_OK;
  };
  // The Gram route, for a spectrum that decays too slowly for the sketch passes (see the loop below): the leading modes
  // from the M x M Gram matrix of the (centred) block -- eigenpairs to convergence in M space, modes down to 1e-5 sigma_1
  // (GRAM_ACCEPT) lifted, orthonormalised, deflated; the passes below go on from there.
  auto gram_route = [&]() -> int {
    Tmp G, W, fac;
    ROM_TRY(G.get(ctx, size_t(M) * M));
    ROM_TRY(W.get(ctx, size_t(n) * M));
    ROM_TRY(fac.get(ctx, n));
    ROM_TRY(rom_launch_gram(ctx, M, dim, X, dim, G, M));
    info.gram_passes = 1;
    info.executed += double(M) * (M + 1) * double(dim);
    std::vector<double> lam;
    ROM_TRY(top_eigenpairs(ctx, G, M, n, W, lam, fac, info));
    G.release();
    for (double& v : lam) v = std::max(v, 0.0);
    sigma_1 = lam.empty() ? 0.0 : std::sqrt(lam[0]);
    int take = 0;
    // (the caller's floor applies here as well as in the sketch passes: modes below rel_floor x sigma_1 are not looked for)
    while (take < n && take < int(lam.size()) && lam[take] > GRAM_ACCEPT * lam[0] && lam[take] > 0 &&
           lam[take] > floor_rel * floor_rel * lam[0]) ++take;
    found = 0;
    if (take) {
      // rows of W / sigma_i, with the eigenvalues top_eigenpairs left on the device (no upload, no host synchronisation)
      kp_scale_eigvec_rows<<<dim3(unsigned(std::min((M + 255) / 256, 64)), take), 256, 0, ctx->stream>>>(W, M, fac, 0.0);
      ROM_HIP(hipGetLastError());
      ROM_TRY(rom_launch_gemm_nn(ctx, take, dim, M, 1.0, W, M, X, dim, 0.0, V, dim));   // V = S^-1 W^T Xc
      info.executed += 2.0 * take * M * double(dim);
      {  // (lifted Gram modes are orthonormal to ~1e-6 at worst; the symmetric orthonormalisation is second order in that defect)
        Tmp Gs, ls, Ts;
        ROM_TRY(Gs.get(ctx, size_t(take) * take));
        ROM_TRY(ls.get(ctx, take));
        ROM_TRY(Ts.get(ctx, size_t(take) * take));
        ROM_TRY(lowdin_rows(ctx, V, take, dim, Gs, ls, Ts));
      }
      ROM_TRY(deflate(0, take));
      found = take;
    }
    return ROM_OK;
  };
  // How many of the `take` leading modes of a pass have converged.  The range finder with one power step leaves
  // (sigma_{b+1} / sigma_k)^3 of the directions beyond its b rows in mode k; sigma_{b+1} is not known, the smallest Ritz
  // value of the sketch stands for it with a factor 10 (a Ritz value underestimates).  Wanted: LAPACK's own bound
  // eps sigma_1 / sigma_k (x 10), or 1e-10 where that is smaller.  A sketch that spans the whole range is exact.
  auto converged_prefix = [&](const std::vector<double>& ss, int b, int take, int power) -> int {
    if (b + found >= std::min<int64_t>(M, dim) - (center ? 1 : 0) || take == 0) return take;
    const double tail = 10.0 * ss[b - 1];
    int k = 0;
    while (k < take) {
      const double rho = tail / ss[k];
      if (std::pow(rho, 2 * power + 1) > std::max(1e-10, 10.0 * 1.1e-16 * sigma_1 / ss[k])) break;   // (q power steps: rho^(2 q + 1))
      ++k;
    }
    return k;
  };
  bool gram_done = false;
  int power = 1;   // power steps per pass: 2 once a pass has shown a spectrum too flat for one (best-effort regime below)
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
    // (never fewer than 32 rows: the thin products are bound by the read of the block and the small kernels hold 32 rows at
    // the same cost, and every extra row lowers sigma_{b+1} -- what the accuracy of the accepted modes is measured against)
    int b = int(std::min<int64_t>(std::min<int64_t>(M, dim), std::max(want + 8, FUSED_ROWS)));
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
    // the next pass's product beside this pass's Rayleigh-Ritz kernel: a next pass is certain -- unless the floor is reached --
    // when this one cannot fill the request, and it asks for at most PASS_MODES + 8 rows as long as fewer than 4 PASS_MODES
    // modes are left (32 rows are computed, the pass uses the leading b of them)
    auto start_next = [&]() -> int {
      if (left - want >= 1 && left - 1 <= 4 * PASS_MODES && worth_ahead)
        return sketch_ahead_start(ctx, ahead, X, M, dim, int(std::min<int64_t>(std::min<int64_t>(M, dim), PASS_MODES + 8)), p + 1);
      return ROM_OK;
    };
    // The first pass is a PILOT as well: after its first two products the pivots of the M-space orthonormalisation estimate the
    // sketch's singular values.  If by them NO mode would meet the convergence rule (even without its safety factor) and the
    // cost model prefers the Gram route, the pass is abandoned there -- its other two products over the block would be wasted
    // (C5: 15 ms of 380).  A spectrum that decays fast never gets here: one read-back of b numbers.
    auto pilot = [&](const double* d_lam, std::vector<double>& est, bool& abandon) -> int {
      abandon = false;
      if (p != 1 || found != 0 || gram_done || b + found >= std::min<int64_t>(M, dim) - (center ? 1 : 0)) return ROM_OK;
      const double t_prod = double(M) * double(dim) * 8.0 / 4.5e12 + 30e-6, t_pass = 4.0 * t_prod + 0.5e-3;
      const double t_gram = double(M) * double(M) * double(dim) / 55e12 + 1.5e-3;
      if (2.0 * t_prod < 1e-3 || t_gram > 50.0 * t_pass) return ROM_OK;   // (nothing to save / the Gram route is not an option)
      ROM_TRY(download(ctx, d_lam, est.data(), b));
      for (double& v : est) v = std::sqrt(std::max(v, 0.0));
      std::sort(est.begin(), est.end(), std::greater<double>());
      if (!(est[0] > 0.0)) return ROM_OK;
      const double rho = est[b - 1] / est[0];
      abandon = rho * rho * rho > std::max(1e-10, 10.0 * 1.1e-16);   // (the rule of converged_prefix for the STRONGEST mode, no safety factor)
      return ROM_OK;
    };
    ROM_TRY(sketch_pass(ctx, X, M, dim, V, Bt, found, b, p, Q, hit ? Om.p() : nullptr, Rt, Traw, ss, info, start_next, power, pilot));
    if (ss.empty()) {   // the pilot said: a slowly decaying spectrum
      gram_done = true;
      ROM_TRY(gram_route());
      if (found == 0) break;
      continue;
    }
    info.sketch_passes += 1;
    if (found == 0 && !gram_done && !(ss[0] > 0.0 && ss[0] <= 1.7976931348623157e308)) {
      // nothing came out of the first pass.  A block of zeros is one reason (the floor logic below deals with it); the others
      // are not silent: NaN / Inf entries (scikit-learn's PCA raises on those) and entries so large or small that their
      // squares leave the range of fp64 -- every inner product of the passes is then Inf or 0.  One more look at the block
      // tells them apart; no call with a first singular value pays for it.
      Tmp st;
      ROM_TRY(st.get(ctx, 2));
      unsigned long long* d_st = reinterpret_cast<unsigned long long*>(st.p());
      ROM_HIP(hipMemsetAsync(d_st, 0, 2 * sizeof(unsigned long long), ctx->stream));
      kp_block_amax<<<1024, 256, 0, ctx->stream>>>(X, size_t(M) * dim, d_st);
      ROM_HIP(hipGetLastError());
      unsigned long long h_st[2] = {0, 0};
      ROM_HIP(hipMemcpyAsync(h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, ctx->stream));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
      double amax;
      memcpy(&amax, &h_st[0], sizeof(double));
      ROM_CHECK(h_st[1] == 0, "rom_pod: the block contains %llu NaN / Inf entries", h_st[1]);
      ROM_CHECK(amax == 0.0 || (amax < 1e140 && amax > 1e-140),
                "rom_pod: entries of magnitude %.3g -- their squares leave the range of fp64; rescale the block", amax);
    }
    // (the small eigenproblems compare squares of entries of Gram matrices, sigma^4: outside 1e-70 ... 1e70 those leave the
    // range of fp64 and the rotations stop silently -- LAPACK would rescale; this library says so)
    ROM_CHECK(found > 0 || gram_done || ss[0] == 0.0 || (ss[0] < 1e70 && ss[0] > 1e-70),
              "rom_pod: singular values of magnitude %.3g -- their fourth powers leave the range of fp64; rescale the block", ss[0]);
    if (found == 0) sigma_1 = ss.empty() ? 0.0 : ss[0];
    int take = 0;
    while (take < std::min(b, want) && ss[take] > SKETCH_ACCEPT * ss[0] && ss[take] > floor_rel * sigma_1) ++take;
    if (take == 0) {
      at_floor_stop = b == 0 || ss[0] <= floor_rel * sigma_1;
      break;
    }
    {
      // A slowly decaying spectrum: only a prefix of the modes has converged (the rest stays in the deflated block for the
      // next pass, at the top of its sketch).  When the first pass shows that the passes would cost more than the Gram
      // route -- whose iterations run in M space, 2 M^2 b flops each instead of two passes over the block -- its result is
      // dropped and the Gram route takes over; later passes without a converged mode accept what the rules above give.
      const int good = converged_prefix(ss, b, take, power);
      if (good < take && found == 0 && !gram_done) {
        const double t_prod = double(M) * double(dim) * 8.0 / 4.5e12 + 30e-6, t_pass = 4.0 * t_prod + 0.5e-3;
        const double t_gram = double(M) * double(M) * double(dim) / 55e12 + 1.5e-3;
        const double passes_left = good > 0 ? std::ceil(double(left) / good) : 1e9;
        if (passes_left * t_pass > t_gram + t_pass) {
          gram_done = true;
          ROM_TRY(gram_route());
          if (found == 0) break;   // (a zero block)
          continue;
        }
      }
      if (good >= 1) {
        take = good;
      } else {
        // no mode of this pass meets the bound (a slowly decaying spectrum below the reach of the Gram route): best effort --
        // the modes at least a factor 4 above the bottom of the sketch (their share of the directions beyond the sketch is
        // ~(1/4)^3), at least one; the rest is left for the next pass, where it sits at the top
        int k = 0;
        while (k < take && ss[k] >= 4.0 * ss[b - 1]) ++k;
        // (none: a plateau -- typically the rounding noise of a block with a large mean, above the floor because the floor
        // is relative to sigma_1 of the CENTRED block -- whose directions no method tells apart: a quarter of the sketch per pass)
        take = k > 0 ? k : std::max(1, std::min(take, b / 4));
        power = 2;   // (and the passes from here on take a second power step: (1/4)^5 instead of (1/4)^3)
      }
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
    // the first value NOT accepted says "floor reached" only if the pass resolves values that small: below 1e-8 of its top
    // (eps r^2 ~ 1, see SKETCH_ACCEPT) a Ritz value is rounding noise -- a cliff of more than eight orders behind the accepted
    // modes hides whatever lies between the cliff's foot and the floor, and the next pass, on the deflated block, looks there
    const bool at_floor = take < b && ss[take] <= floor_rel * sigma_1 && floor_rel * sigma_1 >= 1e-8 * ss[0];
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
    info_host[2] = info.gram_passes;
    info_host[3] = info.sketch_passes;
    info_host[4] = info.executed;
    info_host[5] = 8.0 * n * M * double(dim);   // the four thin products of a pass for the n requested modes alone (no oversampling)
    info_host[6] = info.eig_iterations;
    // why the call stopped short of n modes: 0 request filled, 1 the spectrum reached the floor (the completed modes are
    // not determined by the data), 2 no accepted mode in a pass / pass budget (modes above the floor may be missing)
    info_host[7] = found >= n ? (info.unconverged ? 2.0 : 0.0) : (at_floor_stop || sigma_1 == 0.0 ? 1.0 : 2.0);
  }
  return ROM_OK;
}

extern "C" int rom_pod(rom_ctx* ctx, rom_buf* Xb, int64_t x_row0, int M, int64_t dim, int n, int center, rom_buf* Vb,
                       int64_t v_row0, double* sigma_host, double* info_host) {
  return rom_pod_ex(ctx, Xb, x_row0, M, dim, n, center, 0.0, Vb, v_row0, sigma_host, info_host);
}
