// POD of a snapshot block behind the C-ABI (rom_pod / rom_pod_ex): the PCA fit inside ReducedBasisPCA.build
// (src/lib/ReducedBasis.py:189-200).  Gram matrix on MFMA, leading eigenpairs by a pivoted-Cholesky low-rank factor (or
// subspace iteration when the spectrum decays slowly), implicitly deflated randomised sketches below the Gram matrix's
// resolution; every small dense problem runs on the device.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rom_ops.h"

#include "rom_basis_int.h"

// =====================================================================================================================
// POD (the PCA fit of ReducedBasisPCA.build, src/lib/ReducedBasis.py:189-200)
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

// out[i] = lam[i] > 0 ? 1 / sqrt(lam[i]) : 0
__global__ void kb_inv_sqrt(const double* __restrict__ lam, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = lam[i] > 0.0 ? 1.0 / sqrt(lam[i]) : 0.0;
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
  __shared__ double lp[128];
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
    if (t < k && t < 128) lp[t] = Lt[size_t(t) * M + p];
    __syncthreads();
    const double s = 1.0 / sqrt(piv);
    for (int i = t; i < M; i += 1024) {
      double c = G[size_t(p) * ldg + i];
      for (int j = 0; j < k; ++j) c -= Lt[size_t(j) * M + i] * lp[j];
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

// rows of W (r x M) scaled by 1 / sqrt(lam_i) where lam_i > floor_rel * lam_0, zeroed elsewhere
__global__ void kp_scale_eigvec_rows(double* __restrict__ W, long long M, const double* __restrict__ lam, double floor_rel) {
  const double l = lam[blockIdx.y], l0 = lam[0];
  const double a = (l > floor_rel * l0 && l > 0.0) ? 1.0 / sqrt(l) : 0.0;
  double* row = W + blockIdx.y * M;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < M; j += (long long)gridDim.x * blockDim.x) row[j] *= a;
}

namespace {

constexpr double GRAM_ACCEPT = 1e-10;    // eigenvalues of a Gram matrix are taken down to this fraction of its largest one
// singular values of a sketch are taken down to this fraction of its largest one: the power step of the range finder
// weighs a direction with sigma^3, so what lies four orders below the top of a pass is still resolved to ~1e-16 / 1e-12
// of itself, what lies six orders below is not (measured: angle 1.4e-3 instead of 7e-6 for a mode at 1e-11 sigma_1)
constexpr double SKETCH_ACCEPT = 1e-4;
constexpr double NOISE_FLOOR = 1e-13;    // modes below this fraction of sigma_1 are fp64 noise of the snapshots

struct PodInfo {
  int gram_passes = 0, sketch_passes = 0, completed = 0, resolved = 0, eig_iterations = 0, lowrank = 0;
  double executed = 0.0;
};

constexpr int LOWRANK_CAP = 96;          // most steps of the pivoted Cholesky that replaces the subspace iteration
constexpr double LOWRANK_TOL = 1e-14;    // its stopping pivot, relative to the first one
constexpr double LOWRANK_RESIDUAL = 2e-14;  // accepted ||G - L L^T|| (bounded by the trace of the remaining diagonal) / lambda_1

// The leading eigenpairs of a numerically low-rank PSD matrix without iteration: G ~ Lt^T Lt by pivoted Cholesky (rank r,
// one launch), the r x r problem Lt Lt^T = Q diag(theta) Q^T, eigenvectors w_i = Lt^T q_i / sqrt(theta_i).  The error
// against the eigenpairs of G itself is that of a perturbation of norm <= trace(remaining diagonal), which the kernel
// reports; the caller falls back to the subspace iteration when that bound is above LOWRANK_RESIDUAL x theta_0 or the
// factor did not end within LOWRANK_CAP steps (a slowly decaying spectrum).  done = 1 on success.
int lowrank_eigenpairs(rom_ctx* ctx, const double* G, int M, int nev, double* W, std::vector<double>& theta_host, PodInfo& info,
                       int& done) {
  done = 0;
  const int rcap = std::min(M, LOWRANK_CAP);
  if (M <= rcap) return ROM_OK;  // (the full space: one exact Ritz step of the general path)
  Tmp Lt, dw, out, H, St, lam, Wr;
  ROM_TRY(Lt.get(ctx, size_t(rcap) * M));
  ROM_TRY(dw.get(ctx, M));
  ROM_TRY(out.get(ctx, 4));
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
  theta_host.assign(nev, 0.0);
  for (int i = 0; i < ncopy; ++i) theta_host[i] = std::max(th[i], 0.0);
  info.eig_iterations = 0;
  info.lowrank = r;
  done = 1;
  return ROM_OK;
}

// Leading nev eigenpairs of the symmetric PSD matrix G (M x M) by subspace iteration with Rayleigh-Ritz; the projected
// b x b problems are solved on the device.  theta_host: nev values; W: (nev, M) rows = eigenvectors.
int top_eigenpairs(rom_ctx* ctx, const double* G, int M, int nev, double* W, std::vector<double>& theta_host, PodInfo& info,
                   int oversample = 12, double tol = 2e-14, int max_iter = 30, double accept = GRAM_ACCEPT) {
  {
    static const bool no_lowrank = getenv("ROMHC_POD_NO_LOWRANK") != nullptr;   // dev A/B switch
    int done = 0;
    if (!no_lowrank) ROM_TRY(lowrank_eigenpairs(ctx, G, M, nev, W, theta_host, info, done));
    if (done) return ROM_OK;
  }
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
  double best = 1e300;
  int stall = 0;
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
    double worst = 0.0;
    for (int i = 0; i < ncheck; ++i)
      if (th[i] > accept * std::fabs(th[0])) worst = std::max(worst, th[b0 + i] / t0);
    if (worst < 0.7 * best) { best = worst; stall = 0; } else { ++stall; }
    if (worst <= tol || stall >= 3 || it == max_iter - 1) break;
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
  theta_host.assign(th.begin(), th.begin() + nev);
  for (int i = b; i < nev; ++i) theta_host[i] = 0.0;
  const int ncopy = std::min(nev, b);
  ROM_HIP(hipMemcpyAsync(W, Yr.p(), size_t(ncopy) * M * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (ncopy < nev) ROM_HIP(hipMemsetAsync(W + size_t(ncopy) * M, 0, size_t(nev - ncopy) * M * sizeof(double), ctx->stream));
  return ROM_OK;
}

// Right singular vectors / singular values of a tall factor given TRANSPOSED, Tt (b x M, ld M): Rayleigh-Ritz rounds on
// the b x b Gram matrix Tt Tt^T.  The first round rotates the rows towards the singular directions; from then on the
// Gram matrix is graded and nearly diagonal, where Jacobi resolves the small eigenvalues to high relative accuracy --
// nothing is lost to the squaring that a single eigen-decomposition of an ungraded Gram matrix would lose.
// Rt (b x b): accumulated rotation (rows = right singular vectors in the coordinates Tt came in); sig2: b values.
int tall_svd_rotation(rom_ctx* ctx, double* Tt, int b, int M, double* Rt, double* sig2, int rounds = 3) {
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

// Leading k right singular vectors / singular values of the (M, dim) block X by a randomised range finder with one power
// iteration: thin GEMMs (2 b M dim flops each) instead of the 2 M^2 dim of a Gram matrix.  Used for the DEFLATED
// remainder of a snapshot block.  Vs: (b, dim) block, its first k rows are the modes; ss_host: b singular values.
// The block is DEFLATED IMPLICITLY: X_d = X - Bt^T V with the `found` modes accepted so far (V: found x dim, orthonormal
// rows; Bt: found x M, row j = X v_j).  Every product with X_d is the product with X followed by a rank-`found`
// correction (two small GEMMs) -- X itself is never rewritten, which saves a read + write of the whole block per
// accepted batch of modes.  The rounding error is what the explicit subtraction leaves in X_d as well: eps x sigma_1.
int sketched_modes(rom_ctx* ctx, const double* X, int M, int64_t dim, const double* V, const double* Bt, int found, int k,
                   int seed, double* Vs, std::vector<double>& ss_host, int& b_out, PodInfo& info, int oversample = 8,
                   int power = 1) {
  const int b = int(std::min<int64_t>(std::min<int64_t>(M, dim), k + oversample));
  b_out = b;
  Tmp Om, Y, scr, Tt, Rt, s2, Cc;
  ROM_TRY(Cc.get(ctx, size_t(b) * std::max(found, 1)));
  // out (b x ncols, ld ncols... ) -= ((lhs (b x kk) rhs_t^T (found x kk)) ) other: the two correction shapes below
  auto correct_rows = [&](double* out /* b x dim */, const double* left /* b x M */) -> int {   // out -= (left Bt^T) V
    if (found == 0) return ROM_OK;
    ROM_TRY(rom_launch_gemm_nt(ctx, b, found, M, 1.0, left, M, Bt, M, 0.0, Cc, found, "gemm_nt"));
    return rom_launch_gemm_nn(ctx, b, dim, found, -1.0, Cc, found, V, dim, 1.0, out, dim);
  };
  ROM_TRY(Om.get(ctx, size_t(b) * M));
  ROM_TRY(Y.get(ctx, size_t(b) * dim));
  ROM_TRY(scr.get(ctx, size_t(b) * dim));
  ROM_TRY(Tt.get(ctx, size_t(b) * M));
  ROM_TRY(Rt.get(ctx, size_t(b) * b));
  ROM_TRY(s2.get(ctx, b));
  ROM_TRY(romb_fill_random(ctx, Om, size_t(b) * M, 0xabcd0000ull + unsigned(seed) * 7919u + unsigned(b), true));
  ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Om, M, X, dim, 0.0, Y, dim));                 // Y = Omega X_d
  ROM_TRY(correct_rows(Y, Om));
  info.executed += 2.0 * b * M * double(dim);
  for (int it = 0; it <= power; ++it) {
    // Q (rank may drop: zero rows).  Before the power step one round is enough: the rows only have to span the sketch
    // and be aligned with its principal directions -- their residual non-orthogonality (eps x the condition number of
    // the sketch's Gram matrix) does not change what the power step spans; the basis that is USED is whitened twice
    ROM_TRY(romb_gram_transform(ctx, Y, scr, b, dim, SE_WHITEN, 1e-26, it == power ? 2 : 1));
    ROM_TRY(rom_launch_gemm_nt(ctx, b, M, dim, 1.0, Y, dim, X, dim, 0.0, Tt, M, "gemm_nt"));      // Tt = Q X_d^T  (b, M)
    if (found) {                                                                                  // ... - (Q V^T) Bt
      ROM_TRY(rom_launch_gemm_nt(ctx, b, found, dim, 1.0, Y, dim, V, dim, 0.0, Cc, found, "gemm_nt"));
      ROM_TRY(rom_launch_gemm_nn(ctx, b, M, found, -1.0, Cc, found, Bt, M, 1.0, Tt, M));
    }
    info.executed += 2.0 * b * M * double(dim) + 4.0 * b * b * double(dim);
    if (it == power) break;
    ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, M, 1.0, Tt, M, X, dim, 0.0, scr, dim));               // Q X_d^T X_d
    ROM_TRY(correct_rows(scr, Tt));
    ROM_HIP(hipMemcpyAsync(Y.p(), scr.p(), size_t(b) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    info.executed += 2.0 * b * M * double(dim);
  }
  // X ~ T Q: the right singular vectors of the small factor rotate Q into the modes
  ROM_TRY(tall_svd_rotation(ctx, Tt, b, M, Rt, s2));
  ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, b, 1.0, Rt, b, Y, dim, 0.0, Vs, dim));                  // modes = R^T Q
  info.executed += 2.0 * b * b * double(dim);
  ss_host.resize(b);
  ROM_TRY(download(ctx, s2, ss_host.data(), b));
  for (double& v : ss_host) v = std::sqrt(std::max(v, 0.0));
  return ROM_OK;
}

}  // namespace

// Leading n right singular vectors / singular values of the (M, dim) block X (overwritten when it is centred).
// center != 0: subtract the column means first (sklearn PCA.fit).  V: (n, dim) rows = modes, sign convention of
// sklearn's svd_flip(u_based_decision=False); sigma_host: n singular values (0 for completed modes);
// info_host (8 doubles, may be null): resolved modes, completed modes, Gram passes, sketch passes, executed flops,
// useful flops, subspace iterations, stop reason (0 filled, 1 floor reached, 2 budget).
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
  if (center) {
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
    // X and the deflated block give the same coefficients).  X is not touched: the deflation is implicit (sketched_modes)
    ROM_TRY(rom_launch_gemm_nt(ctx, take, M, dim, 1.0, V + size_t(lo) * dim, dim, X, dim, 0.0, Bt.p() + size_t(lo) * M, M, "gemm_nt"));
    info.executed += 2.0 * take * M * double(dim);
    return ROM_OK;
  };
  // the sketch passes run until the request is filled or the spectrum has reached the floor; the budget below only guards
  // against a pass that makes no progress (every pass accepts at least one mode or ends the loop)
  const int passes = n + 2;
  bool at_floor_stop = false;
  if (n > 0) {
    Tmp G, W, fac;
    ROM_TRY(G.get(ctx, size_t(M) * M));
    ROM_TRY(W.get(ctx, size_t(n) * M));
    ROM_TRY(fac.get(ctx, n));
    ROM_TRY(rom_launch_gram(ctx, M, dim, X, dim, G, M));
    info.gram_passes = 1;
    info.executed += double(M) * (M + 1) * double(dim);
    std::vector<double> lam;
    ROM_TRY(top_eigenpairs(ctx, G, M, n, W, lam, info));
    G.release();
    for (double& v : lam) v = std::max(v, 0.0);
    sigma_1 = lam.empty() ? 0.0 : std::sqrt(lam[0]);
    int take = 0;
    while (take < n && take < int(lam.size()) && lam[take] > GRAM_ACCEPT * lam[0] && lam[take] > 0) ++take;
    if (take) {
      std::vector<double> inv(take);
      for (int i = 0; i < take; ++i) inv[i] = 1.0 / std::sqrt(lam[i]);
      ROM_HIP(hipMemcpyAsync(fac.p(), inv.data(), size_t(take) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
      ROM_TRY(rom_launch_rows_scale(ctx, W, take, M, fac));
      ROM_TRY(rom_launch_gemm_nn(ctx, take, dim, M, 1.0, W, M, X, dim, 0.0, V, dim));   // V = S^-1 W^T Xc
      info.executed += 2.0 * take * M * double(dim);
      ROM_TRY(romb_orthonormalize_against(ctx, V, 0, take, dim));
      ROM_TRY(deflate(0, take));
      found = take;
    }
  }
  for (int p = 1; p < passes; ++p) {
    if (found >= n || found == 0) break;
    Tmp Vs;
    std::vector<double> ss;
    int b = 0;
    // a pass accepts modes over four orders of magnitude -- a dozen of them in a spectrum that decays like the snapshot
    // blocks' do -- so it asks for at most 16 (+ 8 of oversampling): the thin products scale with b, the small dense
    // problems with b^3; a spectrum that decays more slowly takes more passes
    // (a request with hundreds of modes left asks for more per pass: 16 per pass would be n / 16 passes over the block)
    const int want = std::min(n - found, std::max(16, (n - found) / 4));
    const int bmax = int(std::min<int64_t>(std::min<int64_t>(M, dim), want + 8));
    ROM_TRY(Vs.get(ctx, size_t(bmax) * dim));
    ROM_TRY(sketched_modes(ctx, X, M, dim, V, Bt, found, want, p, Vs, ss, b, info));
    info.sketch_passes += 1;
    int take = 0;
    while (take < std::min(b, want) && ss[take] > SKETCH_ACCEPT * ss[0] && ss[take] > floor_rel * sigma_1) ++take;
    if (take == 0) {
      at_floor_stop = b == 0 || ss[0] <= floor_rel * sigma_1;
      break;
    }
    ROM_HIP(hipMemcpyAsync(V + size_t(found) * dim, Vs.p(), size_t(take) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ROM_TRY(romb_orthonormalize_against(ctx, V, found, take, dim));
    const bool at_floor = take < b && ss[take] <= floor_rel * sigma_1;
    ROM_TRY(deflate(found, take));
    found += take;
    if (at_floor) {  // the spectrum has reached the floor: nothing left to find
      at_floor_stop = true;
      break;
    }
  }
  if (found) {
    // Rayleigh-Ritz on the collected subspace: X ~ B V  ->  the SVD of B orders / rotates the modes
    Tmp Rt, s2, Vr;
    ROM_TRY(Rt.get(ctx, size_t(found) * found));
    ROM_TRY(s2.get(ctx, found));
    ROM_TRY(Vr.get(ctx, size_t(found) * dim));
    ROM_TRY(tall_svd_rotation(ctx, Bt, found, M, Rt, s2));
    ROM_TRY(rom_launch_gemm_nn(ctx, found, dim, found, 1.0, Rt, found, V, dim, 0.0, Vr, dim));
    ROM_HIP(hipMemcpyAsync(V, Vr.p(), size_t(found) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    info.executed += 2.0 * found * found * double(dim);
    std::vector<double> s(found);
    ROM_TRY(download(ctx, s2, s.data(), found));
    for (int i = 0; i < found; ++i) sigma_host[i] = std::sqrt(std::max(s[i], 0.0));
  }
  if (found < n) {
    // complete the basis: random directions orthonormalised against the modes; they carry no variance (LAPACK and
    // scikit-learn return SOME orthonormal directions there too).  Seeded by the count of resolved modes: deterministic.
    const int rest = n - found;
    ROM_TRY(rom_complete_orthonormal(ctx, Vb, v_row0, found, rest, dim));
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
    info_host[5] = double(M) * (M + 1) * double(dim) + 2.0 * n * M * double(dim);
    info_host[6] = info.eig_iterations;
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
