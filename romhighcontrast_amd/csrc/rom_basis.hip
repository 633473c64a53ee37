// The basis stage behind the C-ABI: single-call operations for the projectors, the greedy and the POD.
//
// Reference counterparts: project_solutions (src/lib/SolutionsManagers.py:108-139), generate_fm_solutions (:88-106),
// orthonormalize_base (src/lib/ReducedBasis.py:18-21), ReducedBasisGreedy.build (:112-139), the PCA fit inside
// ReducedBasisPCA.build (:189-200).  Everything here runs on the device: small dense problems (symmetric eigenproblems
// of at most a few hundred unknowns, Cholesky-free whitening) are solved by a one-workgroup Jacobi kernel, selections
// (argmax of the greedy) stay in device memory, norms feed the next kernel through device scalars.  The host sees a
// status word at the end of a call -- and, in the POD, the handful of spectrum values its acceptance decisions need.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rom_ops.h"

#include "rom_basis_int.h"
#include "rom_small_dense.h"

// =====================================================================================================================
// small elementwise / indexing kernels
// =====================================================================================================================
__global__ void kb_fill(double* p, size_t n, double v) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = v;
}

static int fill(rom_ctx* ctx, double* p, size_t n, double v) {
  if (n == 0) return ROM_OK;
  if (v == 0.0) {
    ROM_HIP(hipMemsetAsync(p, 0, n * sizeof(double), ctx->stream));
    return ROM_OK;
  }
  kb_fill<<<unsigned(std::min<size_t>((n + 255) / 256, 2048)), 256, 0, ctx->stream>>>(p, n, v);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// counter-based generator (splitmix64 of seed and index): the same numbers whatever the launch shape
__device__ inline unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double u01(unsigned long long h) { return (double(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// gaussian != 0: standard normal (Box-Muller); else uniform in (-0.5, 0.5)
__global__ void kb_fill_random(double* p, size_t n, unsigned long long seed, int gaussian) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const unsigned long long h1 = mix64(seed * 0xD1342543DE82EF95ull + 2 * i), h2 = mix64(seed * 0xD1342543DE82EF95ull + 2 * i + 1);
    p[i] = gaussian ? sqrt(-2.0 * log(u01(h1))) * cos(6.283185307179586 * u01(h2)) : u01(h1) - 0.5;
  }
}

int romb_fill_random(rom_ctx* ctx, double* p, size_t n, unsigned long long seed, bool gaussian) {
  if (n == 0) return ROM_OK;
  kb_fill_random<<<unsigned(std::min<size_t>((n + 255) / 256, 4096)), 256, 0, ctx->stream>>>(p, n, seed, gaussian ? 1 : 0);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// out[i, :] = x[i, :] + s * f[i] * y[i, :]    (rows x dim; x may be null = 0)
__global__ void kb_rows_axpy(double* __restrict__ out, const double* __restrict__ x, const double* __restrict__ y,
                             const double* __restrict__ f, double s, long long dim) {
  const double a = s * f[blockIdx.y];
  const long long o = blockIdx.y * dim;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < dim; j += (long long)gridDim.x * blockDim.x)
    out[o + j] = (x ? x[o + j] : 0.0) + a * y[o + j];
}

// dst (cols x rows, ld ldd) = transpose of src (rows x cols, ld lds): small matrices only
__global__ void kb_transpose(double* __restrict__ dst, long long ldd, const double* __restrict__ src, long long lds,
                             int rows, int cols) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)rows * cols) return;
  const int r = int(idx / cols), c = int(idx % cols);
  dst[c * ldd + r] = src[r * lds + c];
}

static int transpose(rom_ctx* ctx, double* dst, long long ldd, const double* src, long long lds, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return ROM_OK;
  kb_transpose<<<unsigned(((long long)rows * cols + 255) / 256), 256, 0, ctx->stream>>>(dst, ldd, src, lds, rows, cols);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// =====================================================================================================================
// symmetric eigenproblem of a small matrix: cyclic Jacobi, one workgroup
// =====================================================================================================================
// A (n x n, leading dimension lda, symmetrised on entry) = sum_i lam_i q_i q_i^T.  Rotations of a round act on disjoint
// index pairs (round-robin tournament ordering), so a round is three workgroup-wide phases: angles, row rotations of A
// and of the accumulated eigenvector rows, column rotations of A.  A rotation is skipped when |a_pq| <= eps sqrt(a_pp a_qq)
// -- the criterion under which Jacobi computes the small eigenvalues of a graded positive definite matrix to high
// RELATIVE accuracy (Demmel-Veselic), which is what the whitening of nearly dependent sketches and the Rayleigh-Ritz
// rounds of the POD rely on.  The matrix and the eigenvector rows live in LDS up to n = SE_LDS_MAX (160 KB opted in),
// beyond that in a global workspace (gws; L2 resident).
// Output (mode): eigenvalues descending in lam, and in T (ld ldt)
//   SE_EIG     rows = eigenvectors q_i^T
//   SE_WHITEN  row i = q_i^T / sqrt(lam_i) for lam_i > rel_tol * lam_0, zero rows otherwise
//              (T X has orthonormal rows spanning the numerical row space of X when A = X X^T)
//   SE_LOWDIN  sum_i q_i q_i^T / sqrt(lam_i) over the same i: the symmetric inverse square root (nearest orthonormal rows)

// threads of the eigen-solver's workgroup: a template parameter (small matrices are latency bound per round: fewer waves,
// cheaper barriers)

// GWS: matrix and eigenvector rows in the global workspace (n > SE_LDS_MAX); else in LDS -- a compile-time choice, so that the
// LDS form addresses them with ds_read / ds_write (a pointer that may be either is a FLAT access: several times the latency)
template <int SE_TPB, bool GWS>
__global__ __launch_bounds__(SE_TPB) void kb_small_eig(int n, const double* __restrict__ A, int lda, double* __restrict__ lam,
                                                       double* __restrict__ T, int ldt, int mode, double rel_tol, int gram_like,
                                                       double* __restrict__ gws, double* __restrict__ ns_ws) {
  extern __shared__ __align__(16) double sm[];
  const int t = threadIdx.x, ld = n | 1, ne = n + (n & 1), half = ne / 2;
  double* As = GWS ? gws : sm;
  double* Vt = As + size_t(n) * ld;
  double* vec = GWS ? sm : Vt + size_t(n) * ld;  // [cs half | sn half | ev n | nu2 n | red 8 | (int) pp half, qq half, perm n, flag]
  double* cs = vec;
  double* sn = vec + half;
  double* ev = vec + 2 * half;
  double* nu2 = ev + n;
  double* red = nu2 + n;
  int* pp = reinterpret_cast<int*>(red + 8);
  int* qq = pp + half;
  int* perm = qq + half;
  int* flag = perm + n;
  constexpr int NW = SE_TPB / 64;
  auto block_max = [&](double v) -> double {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmax(m, red[w]);
    return m;
  };
  double dmax = 0.0;
  for (int idx = t; idx < n * n; idx += SE_TPB) {
    const int r = idx / n, c = idx % n;
    const double v = 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]);
    As[r * ld + c] = v;
    Vt[r * ld + c] = r == c ? 1.0 : 0.0;
    if (r == c) dmax = fmax(dmax, fabs(v));
  }
  dmax = block_max(dmax);
  if (mode != SE_EIG && dmax > 0.0 && !GWS) {
    // Rows that are orthogonal up to a moderate defect -- G = g (I + E), g = mean diagonal, ||E|| < 1/2: rotated Ritz
    // vectors of a subspace iteration, lifted modes, a Gaussian start block -- need no eigen-decomposition: the inverse
    // square root comes from the coupled Newton-Schulz iteration
    //     W = (3 I - Z Y) / 2 ,  Y <- Y W ,  Z <- W Z        (Y_0 = I + E, Z_0 = I;  Y -> (I+E)^(1/2), Z -> (I+E)^(-1/2))
    // which converges quadratically: a handful of n^3 products in LDS instead of ten Jacobi sweeps of n - 1 dependent
    // rounds each.  The transform is symmetric, so it serves the whitening and the symmetric orthonormalisation alike.
    // Eigenvalues reported: the diagonal of G.  (W lives in the caller's T, which is overwritten at the end anyway.)
    double tr = 0.0;
    for (int i = t; i < n; i += SE_TPB) tr += As[i * ld + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tr += __shfl_down(tr, o, 64);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = tr;
    __syncthreads();
    double gmean = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) gmean += red[w];
    gmean /= double(n);
    double esum = 0.0;  // max row sum of |E|: an upper bound of ||E||_2
    for (int r = t; r < n; r += SE_TPB) {
      double rs = 0.0;
      for (int c = 0; c < n; ++c) rs += fabs(As[r * ld + c] / gmean - (r == c ? 1.0 : 0.0));
      esum = fmax(esum, rs);
    }
    esum = block_max(esum);
    // (||E||_2 < 1 is what the iteration needs; the row-sum bound is sufficient, not necessary: up to 3 it is tried and
    // abandoned -- matrix reloaded, Jacobi below -- if the defect ||Z Y - I|| ever grows)
    if (gmean > 0.0 && esum < 3.0) {
      bool diverged = false;
      double dev_prev = 1e300;
      for (int i = t; i < n; i += SE_TPB) lam[i] = As[i * ld + i];
      __syncthreads();
      for (int idx = t; idx < n * n; idx += SE_TPB) {
        const int r = idx / n, c = idx % n;
        As[r * ld + c] /= gmean;
      }
      __syncthreads();
      double* Yn = ns_ws;                    // new Y, Z of an iteration (global scratch, L2 resident)
      double* Zn = ns_ws + size_t(n) * n;
      for (int itn = 0; itn < 20; ++itn) {
        double dev = 0.0;
        for (int idx = t; idx < n * n; idx += SE_TPB) {
          const int r = idx / n, c = idx % n;
          double acc = 0.0;
          for (int k = 0; k < n; ++k) acc += Vt[r * ld + k] * As[k * ld + c];   // (Z Y)_rc
          dev = fmax(dev, fabs(acc - (r == c ? 1.0 : 0.0)));
          T[size_t(r) * ldt + c] = 0.5 * ((r == c ? 3.0 : 0.0) - acc);
        }
        dev = block_max(dev);   // ||Z Y - I||_max; its barriers also order the writes of W before the reads below
        if (dev < 4e-16 * n) break;
        if (!(dev < dev_prev) || itn == 19) { diverged = true; break; }
        dev_prev = dev;
        for (int idx = t; idx < n * n; idx += SE_TPB) {
          const int r = idx / n, c = idx % n;
          double ay = 0.0, az = 0.0;
          for (int k = 0; k < n; ++k) {
            ay += As[r * ld + k] * T[size_t(k) * ldt + c];   // (Y W)_rc
            az += T[size_t(r) * ldt + k] * Vt[k * ld + c];   // (W Z)_rc
          }
          Yn[idx] = ay;
          Zn[idx] = az;
        }
        __syncthreads();   // every thread has read the old Y, Z (and written its part of the new ones)
        for (int idx = t; idx < n * n; idx += SE_TPB) {
          const int r = idx / n, c = idx % n;
          As[r * ld + c] = Yn[idx];
          Vt[r * ld + c] = Zn[idx];
        }
        __syncthreads();
      }
      if (!diverged) {
        const double sc = 1.0 / sqrt(gmean);
        for (int idx = t; idx < n * n; idx += SE_TPB) {
          const int r = idx / n, c = idx % n;
          T[size_t(r) * ldt + c] = sc * 0.5 * (Vt[r * ld + c] + Vt[c * ld + r]);
        }
        return;
      }
      __syncthreads();
      for (int idx = t; idx < n * n; idx += SE_TPB) {  // start over for the eigen-decomposition
        const int r = idx / n, c = idx % n;
        As[r * ld + c] = 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]);
        Vt[r * ld + c] = r == c ? 1.0 : 0.0;
      }
      __syncthreads();
    }
  }
  // A rotation is applied when |a_pq| > tol sqrt(|a_pp a_qq|) -- the relative criterion under which graded matrices keep
  // their small eigenvalues -- AND |a_pq| > tol nu_p nu_q, the rounding noise of the entry: nu_i^2 is what a_ii would be
  // had no cancellation happened (initially |a_ii| for a Gram-like matrix; under a rotation c^2 nu_p^2 + s^2 nu_q^2).
  // A direction that rank deficiency has cancelled to nothing keeps its nu, so the rounding residue that couples it to
  // the rest is recognised as such and the sweeps end; without it they never do (each rotation re-creates the residue).
  // gram_like == 0 (a general symmetric matrix such as Y G Y^T: all entries carry eps ||A||): nu_i^2 = max |a_ii|.
  // single-wave form (SE_TPB == 64, n <= 32: at most 256 blocks and 512 eigenvector items per round): the items of a lane
  int bk[4] = {0, 0, 0, 0}, bl[4] = {0, 0, 0, 0}, vk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vj[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (SE_TPB == 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = u * 64 + t;
      bk[u] = idx / half;
      bl[u] = idx - bk[u] * half;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 64 + t;
      vk[u] = idx / n;
      vj[u] = idx - vk[u] * n;
    }
  }
  const double tol = double(n > 8 ? n : 8) * 1.1e-16, tol2 = tol * tol, floor_abs = fmax(1e-300, 1e-40 * dmax);
  for (int i = t; i < n; i += SE_TPB) nu2[i] = gram_like ? fabs(As[i * ld + i]) : dmax;
  __syncthreads();
  for (int sweep = 0; sweep < 40; ++sweep) {
    if (t == 0) *flag = 0;
    __syncthreads();
    for (int r = 0; r < ne - 1; ++r) {
      if (t < half) {
        int p = t == 0 ? ne - 1 : (r + t) % (ne - 1), q = t == 0 ? r : (r - t + ne - 1) % (ne - 1);
        if (p > q) { const int x = p; p = q; q = x; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double app = As[p * ld + p], aqq = As[q * ld + q], apq = As[p * ld + q];
          const double np2 = nu2[p], nq2 = nu2[q], apq2 = apq * apq;
          if (apq2 > tol2 * fabs(app * aqq) && fabs(apq) > floor_abs && apq2 > tol2 * np2 * nq2) {
            // t = tan(phi) = sign(a) b / (|a| + sqrt(a^2 + b^2)), a = a_qq - a_pp, b = 2 a_pq (the smaller root of
            // t^2 + 2 theta t - 1 = 0, theta = a / b); c = 1 / sqrt(1 + t^2), s = t c
            const double a = aqq - app, bb = 2.0 * apq;
            const double tt = (a >= 0 ? bb : -bb) / (fabs(a) + sqrt(a * a + bb * bb));
            c = se_rsqrt(1.0 + tt * tt);
            s = tt * c;
            nu2[p] = c * c * np2 + s * s * nq2;
            nu2[q] = s * s * np2 + c * c * nq2;
            *flag = 1;
          }
        } else {
          q = -1;  // p is paired with the dummy index of an odd n: no rotation, but its row / column still takes the others'
        }
        cs[t] = c;
        sn[t] = s;
        pp[t] = p;
        qq[t] = q;
      }
      __syncthreads();
      // A <- J^T A J as disjoint 2 x 2 blocks: block (k, l) = rows (p_k, q_k) x columns (p_l, q_l) becomes R_k B R_l^T;
      // one pass, no barrier between the row and the column rotations.  (c, s) = (1, 0) for pairs that do not rotate.
      // Items are taken four at a time -- all loads, then all stores: the blocks are disjoint, but the compiler cannot
      // know that, and one item at a time is a chain of LDS round trips.
      for (int base = 0; base < half * half; base += 4 * SE_TPB) {
        double o00[4], o01[4], o10[4], o11[4];
        int a00[4], a01[4], a10[4], a11[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * SE_TPB + t;
          a00[u] = a01[u] = a10[u] = a11[u] = -1;
          if (idx < half * half) {
            // (one wave, n <= 32: the item -> (k, l) map is the same in every round and was divided out once)
            const int k = SE_TPB == 64 ? bk[u] : idx / half, l = SE_TPB == 64 ? bl[u] : idx - k * half;
            const double sk = sn[k], sl = sn[l];
            if (sk != 0.0 || sl != 0.0) {  // (s, not c, is the test: c rounds to 1 for tiny angles)
              const int pk = pp[k], qk = qq[k], pl = pp[l], ql = qq[l];
              const double ck = cs[k], cl = cs[l];
              const bool vk = qk >= 0, vl = ql >= 0;  // (a pair with the dummy index has one real row / column)
              const double b00 = As[pk * ld + pl], b01 = vl ? As[pk * ld + ql] : 0.0;
              const double b10 = vk ? As[qk * ld + pl] : 0.0, b11 = (vk && vl) ? As[qk * ld + ql] : 0.0;
              const double r00 = ck * b00 - sk * b10, r01 = ck * b01 - sk * b11;   // rows
              const double r10 = sk * b00 + ck * b10, r11 = sk * b01 + ck * b11;
              a00[u] = pk * ld + pl;
              o00[u] = cl * r00 - sl * r01;                                         // columns
              if (vl) { a01[u] = pk * ld + ql; o01[u] = sl * r00 + cl * r01; }
              if (vk) { a10[u] = qk * ld + pl; o10[u] = cl * r10 - sl * r11; }
              if (vk && vl) { a11[u] = qk * ld + ql; o11[u] = sl * r10 + cl * r11; }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (a00[u] >= 0) As[a00[u]] = o00[u];
          if (a01[u] >= 0) As[a01[u]] = o01[u];
          if (a10[u] >= 0) As[a10[u]] = o10[u];
          if (a11[u] >= 0) As[a11[u]] = o11[u];
        }
      }
      // eigenvector rows: Vt <- J^T Vt
      auto vt_items = [&](int base, int slot) {
        double op[4], oq[4];
        int ap[4], aq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * SE_TPB + t;
          ap[u] = -1;
          if (idx < half * n) {
            const int k = SE_TPB == 64 ? vk[slot * 4 + u] : idx / n;
            const int j = SE_TPB == 64 ? vj[slot * 4 + u] : idx - k * n;
            const double s = sn[k];
            if (s != 0.0) {  // (also every pair with the dummy index)
              const int p = pp[k], q = qq[k];
              const double c = cs[k];
              const double vp = Vt[p * ld + j], vq = Vt[q * ld + j];
              ap[u] = p * ld + j;
              aq[u] = q * ld + j;
              op[u] = c * vp - s * vq;
              oq[u] = s * vp + c * vq;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ap[u] >= 0) {
            Vt[ap[u]] = op[u];
            Vt[aq[u]] = oq[u];
          }
      };
      if (SE_TPB == 64) {
        vt_items(0, 0);
        if (256 < half * n) vt_items(256, 1);
      } else {
        for (int base = 0; base < half * n; base += 4 * SE_TPB) vt_items(base, 0);
      }
      __syncthreads();
    }
    const int any = *flag;
    __syncthreads();
    if (!any) break;
  }
  // eigenvalues, descending order (stable: ties by index)
  for (int i = t; i < n; i += SE_TPB) ev[i] = As[i * ld + i];
  __syncthreads();
  for (int i = t; i < n; i += SE_TPB) {
    int rank = 0;
    const double v = ev[i];
    for (int j = 0; j < n; ++j) rank += (ev[j] > v || (ev[j] == v && j < i)) ? 1 : 0;
    perm[rank] = i;
  }
  __syncthreads();
  for (int i = t; i < n; i += SE_TPB) lam[i] = ev[perm[i]];
  const double lmax = ev[perm[0]];
  if (mode == SE_LOWDIN) {
    for (int idx = t; idx < n * n; idx += SE_TPB) {
      const int r = idx / n, c = idx % n;
      double s = 0.0;
      for (int i = 0; i < n; ++i) {
        const double l = ev[i];
        if (l > rel_tol * lmax && l > 0.0) s += Vt[i * ld + r] * Vt[i * ld + c] / sqrt(l);
      }
      T[size_t(r) * ldt + c] = s;
    }
  } else {
    for (int idx = t; idx < n * n; idx += SE_TPB) {
      const int r = idx / n, c = idx % n, src = perm[r];
      double v = Vt[src * ld + c];
      if (mode == SE_WHITEN) {
        const double l = ev[src];
        v = (l > rel_tol * lmax && l > 0.0) ? v / sqrt(l) : 0.0;
      }
      T[size_t(r) * ldt + c] = v;
    }
  }
}

// Eigen-decomposition (SE_EIG) of a symmetric matrix of order n <= 32: jacobi32_run (rom_small_dense.h), 256 threads.
// The eigenproblems of the POD's Rayleigh-Ritz steps (b = 20 ... 32, graded, 5 ... 10 sweeps) are latency chains of a few
// hundred rounds each; same rotation criterion, same sweep order, same results as kb_small_eig.
__global__ __launch_bounds__(256) void kb_jacobi32(int n, const double* __restrict__ A, int lda, double* __restrict__ lam,
                                                   double* __restrict__ T, int ldt, int gram_like) {
  __shared__ Jacobi32Lds L;
  __shared__ double red[4];
  const int t = threadIdx.x;
  double dmax = 0.0;
  for (int idx = t; idx < n * n; idx += 256) {
    const int r = idx / n, c = idx - r * n;
    const double v = 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]);
    L.As[r * J32_LD + c] = v;
    L.Vt[r * J32_LD + c] = r == c ? 1.0 : 0.0;
    if (r == c) dmax = fmax(dmax, fabs(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
  if ((t & 63) == 0) red[t >> 6] = dmax;
  __syncthreads();
  dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  jacobi32_run<256>(n, L, gram_like, dmax);
  if (t < n) lam[t] = L.ev[L.perm[t]];
  for (int idx = t; idx < n * n; idx += 256) {
    const int r = idx / n, c = idx - r * n;
    T[size_t(r) * ldt + c] = L.Vt[L.perm[r] * J32_LD + c];
  }
}

// Rank-revealing whitening transform by PIVOTED CHOLESKY: A = X X^T (n x n Gram matrix, symmetrised on entry),
// P A P^T = L L^T with diagonal pivoting, stopped at the first pivot below rel_tol x the largest one (rank r);
// T = [L_r^-1  0] P, so that the first r rows of T X are orthonormal and the rest are zero.  The rows come out graded
// (row k is orthogonal to the k - 1 stronger ones) -- what the power steps of the range finder need -- and the whole
// thing is n dependent steps of one column each: tens of microseconds, where the eigen-decomposition of an
// ill-conditioned Gram matrix takes ten Jacobi sweeps of n - 1 rounds (1-2 ms).  Like every Gram-based orthonormalisation
// it resolves directions down to ~1e-8 of the strongest per round; callers run two rounds where that matters.
// lam: the squared pivots (descending), zeros behind the rank.  One workgroup, matrix in LDS (n <= SE_LDS_MAX).
__global__ __launch_bounds__(256) void kb_pivchol_whiten(int n, const double* __restrict__ A, int lda, double* __restrict__ lam,
                                                         double* __restrict__ T, int ldt, double rel_tol) {
  extern __shared__ __align__(16) double sm[];
  const int t = threadIdx.x, ld = n | 1;
  double* As = sm;                       // working matrix, lower triangle used
  double* Li = As + size_t(n) * ld;      // L^-1 (r x r, lower)
  double* red = Li + size_t(n) * ld;     // [4] values
  int* redi = reinterpret_cast<int*>(red + 4);  // [4] indices
  int* perm = redi + 4;                  // perm[k] = original index of pivot k
  __shared__ int s_rank, s_piv;
  __shared__ double s_first;
  for (int idx = t; idx < n * n; idx += 256) {
    const int r = idx / n, c = idx % n;
    As[r * ld + c] = 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]);
    Li[r * ld + c] = 0.0;
  }
  for (int i = t; i < n; i += 256) perm[i] = i;
  if (t == 0) s_rank = n;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    // pivot: largest remaining diagonal entry (first one on ties)
    double best = -1e300;
    int at = k;
    for (int j = k + t; j < n; j += 256) {
      const double d = As[j * ld + j];
      if (d > best) { best = d; at = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_down(best, o, 64);
      const int oa = __shfl_down(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((t & 63) == 0) { red[t >> 6] = best; redi[t >> 6] = at; }
    __syncthreads();
    if (t == 0) {
      double bb = red[0];
      int ba = redi[0];
      for (int w = 1; w < 4; ++w)
        if (red[w] > bb || (red[w] == bb && redi[w] < ba)) { bb = red[w]; ba = redi[w]; }
      if (k == 0) s_first = bb;
      s_piv = (bb > rel_tol * s_first && bb > 0.0) ? ba : -1;
      if (s_piv < 0 && s_rank == n) s_rank = k;
    }
    __syncthreads();
    const int pv = s_piv;
    if (pv < 0) break;
    if (pv != k) {  // symmetric swap of rows / columns k and pv (full storage: both triangles kept consistent)
      for (int j = t; j < n; j += 256) {
        const double x = As[k * ld + j];
        As[k * ld + j] = As[pv * ld + j];
        As[pv * ld + j] = x;
      }
      __syncthreads();
      for (int i = t; i < n; i += 256) {
        const double x = As[i * ld + k];
        As[i * ld + k] = As[i * ld + pv];
        As[i * ld + pv] = x;
      }
      if (t == 0) { const int x = perm[k]; perm[k] = perm[pv]; perm[pv] = x; }
      __syncthreads();
    }
    const double lkk = sqrt(As[k * ld + k]);
    __syncthreads();
    for (int i = k + t; i < n; i += 256) As[i * ld + k] = i == k ? lkk : As[i * ld + k] / lkk;   // column k of L
    __syncthreads();
    const int m = n - k - 1;  // trailing update, full square (symmetric storage)
    for (int idx = t; idx < m * m; idx += 256) {
      const int i = k + 1 + idx / m, j = k + 1 + idx % m;
      As[i * ld + j] -= As[i * ld + k] * As[j * ld + k];
    }
    __syncthreads();
  }
  const int r = s_rank;
  // L^-1 by forward substitution, one column per thread: L x = e_j
  for (int j = t; j < r; j += 256) {
    for (int i = j; i < r; ++i) {
      double sacc = i == j ? 1.0 : 0.0;
      for (int q = j; q < i; ++q) sacc -= As[i * ld + q] * Li[q * ld + j];
      Li[i * ld + j] = sacc / As[i * ld + i];
    }
  }
  __syncthreads();
  // T[i, perm[q]] = Li[i, q] (q <= i < r), zero elsewhere
  for (int idx = t; idx < n * n; idx += 256) T[size_t(idx / n) * ldt + idx % n] = 0.0;
  __syncthreads();
  for (int idx = t; idx < r * r; idx += 256) {
    const int i = idx / r, q = idx % r;
    if (q <= i) T[size_t(i) * ldt + perm[q]] = Li[i * ld + q];
  }
  for (int i = t; i < n; i += 256) lam[i] = i < r ? As[i * ld + i] * As[i * ld + i] : 0.0;
}


// The same factorisation for n <= 32 (the row blocks of the POD's sketch passes, six calls per pass): ONE barrier per step.
// No row / column is moved -- a bit mask of the indices still in play replaces the symmetric swap, the factor's column k is
// stored by ORIGINAL row index (Lc[i][k]) -- every wave finds the pivot for itself (an LDS atomic max over keys that
// carry the index: no shuffle chain, no barrier), the trailing update and the factor's
// column are one phase (the update reads row / column pv only, which it does not write), and L^-1 is one forward substitution
// per lane on REGISTERS (fully unrolled; the generic kernel's column-per-thread loop reads back what it has just stored to
// LDS, 500 dependent round trips for the first column).  Same pivots (largest remaining diagonal entry, first on ties), same
// stopping rule, same outputs as kb_pivchol_whiten up to the rounding of x / sqrt(d) against x * (1 / sqrt(d)).
__global__ __launch_bounds__(256) void kb_pivchol_whiten32(int n, const double* __restrict__ A, int lda, double* __restrict__ lam,
                                                           double* __restrict__ T, int ldt, double rel_tol) {
  constexpr int LD = 33;
  __shared__ double As[32 * LD], Lc[32 * LD], dpiv[32];
  __shared__ unsigned long long smax[4];
  __shared__ int perm[32];
  const int t = threadIdx.x;
  for (int idx = t; idx < 32 * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    As[r * LD + c] = (r < n && c < n) ? 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]) : 0.0;
    Lc[r * LD + c] = 0.0;
  }
  __syncthreads();
  unsigned active = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  double first = 0.0;
  int rank = n;
  const int w = t >> 6, lane = t & 63;
  const int j = t & 31, i0 = t >> 5;   // the four entries of a thread: (i0 + 8 u, j) -- the same for every step (nothing moves)
  const unsigned mj = 1u << j;
  for (int k = 0; k < n; ++k) {
    // pivot = largest remaining diagonal entry: one LDS atomic max per lane on a key = the entry's bits with the low five
    // replaced by 31 - index (positive doubles order like their bit patterns; entries that agree to 2^-47 count as ties and
    // the lower index wins, like the first-on-ties rule of the generic kernel).  Each wave does this for itself.
    if (lane == 0) smax[w] = 0ull;
    __builtin_amdgcn_wave_barrier();
    if (lane < 32 && ((active >> lane) & 1u)) {
      const double d = As[lane * LD + lane];
      if (d > 0.0) atomicMax(&smax[w], (static_cast<unsigned long long>(__double_as_longlong(d)) & ~31ull) | unsigned(31 - lane));
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long best = *const_cast<volatile unsigned long long*>(&smax[w]);
    const int pv = best ? 31 - int(best & 31ull) : -1;
    const double dk = pv >= 0 ? As[pv * LD + pv] : 0.0;
    if (k == 0) first = dk;
    if (!(pv >= 0 && dk > rel_tol * first && dk > 0.0)) { rank = k; break; }   // (uniform: every thread sees the same values)
    const double lkk = sqrt(dk), rs = 1.0 / lkk;
    active &= ~(1u << pv);
    const double lj = As[j * LD + pv] * rs;
    if (active & mj) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 8 * u;
        if ((active >> i) & 1u) As[i * LD + j] -= (As[i * LD + pv] * rs) * lj;
      }
    }
    if (t < 32) Lc[t * LD + k] = t == pv ? lkk : ((active & mj) ? lj : 0.0);
    if (t == 0) { perm[k] = pv; dpiv[k] = lkk; }
    __syncthreads();
  }
  if (t == 0) {   // the indices that never became a pivot complete the permutation (their columns of T are zero)
    int at = rank;
    for (int j = 0; j < n; ++j)
      if ((active >> j) & 1u) perm[at++] = j;
  }
  __syncthreads();
  if (t < 32) {
    // column t of L^-1 in pivot order: x_i = (delta_it - sum_{q < i} L[i][q] x_q) / L[i][i]  (x_q = 0 for q < t by itself)
    double x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int pi = i < n ? perm[i] : 0;
      double s = i == t ? 1.0 : 0.0;
#pragma unroll
      for (int q = 0; q < i; ++q) s -= Lc[pi * LD + q] * x[q];
      x[i] = (i < rank && t < rank) ? s / dpiv[i] : 0.0;
    }
    // T[i, perm[t]] = Li[i, t]
    if (t < n) {
      const int col = perm[t];
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (i < n) T[size_t(i) * ldt + col] = x[i];
      lam[t] = t < rank ? dpiv[t] * dpiv[t] : 0.0;
    }
  }
}

int romb_pivchol_whiten(rom_ctx* ctx, int n, const double* A, int lda, double* lam, double* T, int ldt, double rel_tol) {
  if (n <= 0) return ROM_OK;
  if (n <= 32) {
    ROM_PROF(ctx, "pivchol_whiten", 1.0 * n * n * n, 16.0 * n * n);
    kb_pivchol_whiten32<<<1, 256, 0, ctx->stream>>>(n, A, lda, lam, T, ldt, rel_tol);
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  const int ld = n | 1;
  const size_t lds = 2 * size_t(n) * ld * sizeof(double) + 4 * sizeof(double) + (4 + size_t(n)) * sizeof(int) + 16;
  if (lds > 64 * 1024 && !ctx->lds_optin_pivchol) {
    ROM_HIP(hipSetDevice(ctx->device));
    ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kb_pivchol_whiten), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    ctx->lds_optin_pivchol = true;
  }
  {
    ROM_PROF(ctx, "pivchol_whiten", 1.0 * n * n * n, 16.0 * n * n);
    kb_pivchol_whiten<<<1, 256, lds, ctx->stream>>>(n, A, lda, lam, T, ldt, rel_tol);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// =====================================================================================================================
// Cyclic Jacobi of a symmetric matrix of order 97 ... SE_MAX over the WHOLE chip: one launch per round
// =====================================================================================================================
// kb_small_eig keeps a matrix beyond 96 rows in an L2-resident workspace and still runs it on ONE workgroup: 0.13 s at
// n = 217, 6-8 s at n = 900-1000 (the whole-matrix diagonalisation that rom_pod falls back to for a flat spectrum).  The
// rotations of a round act on disjoint index pairs, so a round is a grid: thread (k, l) of the first half * half threads owns
// the 2 x 2 block rows (p_k, q_k) x columns (p_l, q_l) and computes the rotations of its two pairs itself from the pivot
// blocks (jacobi_rotation of rom_small_dense.h: a pure function, the same bits in every thread -- no phase that computes
// rotations for the others, hence no synchronisation inside a round), the other half * ne threads rotate the eigenvector rows;
// everything is read from the buffers the previous round wrote and written to the other pair (every entry is written every
// round).  Same criterion (relative, with the noise levels of a Gram matrix rotated along), same round-robin order as
// jacobi32_run.  n = 1000: 999 launches of a few microseconds per sweep.
__global__ __launch_bounds__(256) void kb_jgrid_round(int ne, int half, int r, const double* __restrict__ Ac, double* __restrict__ An,
                                                      const double* __restrict__ Vc, double* __restrict__ Vn,
                                                      const double* __restrict__ nuc, double* __restrict__ nun,
                                                      const double* __restrict__ par, int* __restrict__ any) {
  const long long gid = blockIdx.x * 256LL + threadIdx.x;
  const int nm1 = ne - 1;
  const double tol2 = par[1], floor_abs = par[2];
  auto pair_of = [&](int k, int& p, int& q) {
    p = r;
    q = nm1;
    if (k) {
      p = r + k;
      if (p >= nm1) p -= nm1;
      q = r - k;
      if (q < 0) q += nm1;
      if (p > q) { const int x = p; p = q; q = x; }
    }
  };
  if (gid < (long long)half * half) {
    const int k = int(gid / half), l = int(gid - (long long)k * half);
    int pk, qk, pl, ql;
    pair_of(k, pk, qk);
    pair_of(l, pl, ql);
    const size_t rk = size_t(pk) * ne, sk = size_t(qk) * ne, rl = size_t(pl) * ne;
    const double kpp = Ac[rk + pk], kqq = Ac[sk + qk], kpq = Ac[rk + qk];
    const double lpp = Ac[rl + pl], lqq = Ac[size_t(ql) * ne + ql], lpq = Ac[rl + ql];
    const double b00 = Ac[rk + pl], b01 = Ac[rk + ql], b10 = Ac[sk + pl], b11 = Ac[sk + ql];
    double ck, s_k, cl, sl, nkpo, nkqo, nlpo, nlqo;
    const bool rot = jacobi_rotation(kpp, kqq, kpq, nuc[pk], nuc[qk], tol2, floor_abs, ck, s_k, nkpo, nkqo);
    (void)jacobi_rotation(lpp, lqq, lpq, nuc[pl], nuc[ql], tol2, floor_abs, cl, sl, nlpo, nlqo);
    const double r00 = ck * b00 - s_k * b10, r01 = ck * b01 - s_k * b11;
    const double r10 = s_k * b00 + ck * b10, r11 = s_k * b01 + ck * b11;
    An[rk + pl] = cl * r00 - sl * r01;
    An[rk + ql] = sl * r00 + cl * r01;
    An[sk + pl] = cl * r10 - sl * r11;
    An[sk + ql] = sl * r10 + cl * r11;
    if (l == 0) {
      nun[pk] = nkpo;
      nun[qk] = nkqo;
      if (rot) *any = 1;
    }
    return;
  }
  const long long idx = gid - (long long)half * half;
  if (idx >= (long long)half * ne) return;
  const int k = int(idx / ne), j = int(idx - (long long)k * ne);
  int pk, qk;
  pair_of(k, pk, qk);
  const size_t rk = size_t(pk) * ne, sk = size_t(qk) * ne;
  double ck, s_k, a, b;
  (void)jacobi_rotation(Ac[rk + pk], Ac[sk + qk], Ac[rk + qk], nuc[pk], nuc[qk], tol2, floor_abs, ck, s_k, a, b);
  const double vp = Vc[rk + j], vq = Vc[sk + j];
  Vn[rk + j] = ck * vp - s_k * vq;
  Vn[sk + j] = s_k * vp + ck * vq;
}

// A0 <- symmetrised copy of A with zero padding to the even order ne, V0 <- identity
__global__ void kb_jgrid_init(int n, int ne, const double* __restrict__ A, int lda, double* __restrict__ A0, double* __restrict__ V0) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= (long long)ne * ne) return;
  const int r = int(idx / ne), c = int(idx - (long long)r * ne);
  A0[idx] = (r < n && c < n) ? 0.5 * (A[size_t(r) * lda + c] + A[size_t(c) * lda + r]) : 0.0;
  V0[idx] = r == c ? 1.0 : 0.0;
}

// par[0] = max |a_ii|, par[1] = tol^2, par[2] = smallest entry that is rotated; nu^2 (noise levels) of the rows
__global__ __launch_bounds__(256) void kb_jgrid_params(int n, int ne, const double* __restrict__ A0, int gram_like, double* __restrict__ par,
                                                       double* __restrict__ nu) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  double dmax = 0.0;
  for (int i = t; i < n; i += 256) dmax = fmax(dmax, fabs(A0[size_t(i) * ne + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
  if ((t & 63) == 0) red[t >> 6] = dmax;
  __syncthreads();
  dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  if (t == 0) {
    const double tol = double(n > 8 ? n : 8) * 1.1e-16;
    par[0] = dmax;
    par[1] = tol * tol;
    par[2] = fmax(1e-300, 1e-40 * dmax);
  }
  for (int i = t; i < ne; i += 256) nu[i] = i < n ? (gram_like ? fabs(A0[size_t(i) * ne + i]) : dmax) : 0.0;
}

__global__ void kb_jgrid_diag(int n, int ne, const double* __restrict__ Ac, double* __restrict__ d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = Ac[size_t(i) * ne + i];
}

// T[i, :] = scale[i] V[perm[i], :], lam[i] = d[perm[i]]
__global__ void kb_jgrid_gather(int n, int ne, const double* __restrict__ Vc, const double* __restrict__ d, const int* __restrict__ perm,
                                const double* __restrict__ scale, double* __restrict__ T, int ldt, double* __restrict__ lam) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= (long long)n * n) return;
  const int i = int(idx / n), c = int(idx - (long long)i * n);
  const int src = perm[i];
  T[size_t(i) * ldt + c] = scale[i] * Vc[size_t(src) * ne + c];
  if (c == 0) lam[i] = d[src];
}

// (mode / rel_tol as kb_small_eig: SE_EIG rows = eigenvectors, SE_WHITEN rows / sqrt(lam) -- zero rows below rel_tol x the
// largest eigenvalue --, SE_LOWDIN the symmetric inverse square root over the same eigenpairs)
static int jacobi_grid(rom_ctx* ctx, int n, const double* A, int lda, double* lam, double* T, int ldt, int mode, double rel_tol,
                       bool gram_like) {
  const int ne = n + (n & 1), half = ne / 2;
  const size_t nn = size_t(ne) * ne;
  Tmp A0, A1, V0, V1, nu0, nu1, par, diag, anyb, permb;
  ROM_TRY(A0.get(ctx, nn));
  ROM_TRY(A1.get(ctx, nn));
  ROM_TRY(V0.get(ctx, nn));
  ROM_TRY(V1.get(ctx, nn));
  ROM_TRY(nu0.get(ctx, ne));
  ROM_TRY(nu1.get(ctx, ne));
  ROM_TRY(par.get(ctx, 4));
  ROM_TRY(diag.get(ctx, n));
  ROM_TRY(anyb.get(ctx, 1));
  ROM_TRY(permb.get(ctx, (size_t(n) + 1) / 2 + 1));
  int* d_any = reinterpret_cast<int*>(anyb.p());
  int* d_perm = reinterpret_cast<int*>(permb.p());
  ROM_PROF(ctx, "jacobi_grid", 30.0 * n * double(n) * n, 16.0 * double(n) * n);
  kb_jgrid_init<<<unsigned((nn + 255) / 256), 256, 0, ctx->stream>>>(n, ne, A, lda, A0, V0);
  kb_jgrid_params<<<1, 256, 0, ctx->stream>>>(n, ne, A0, gram_like ? 1 : 0, par, nu0);
  ROM_HIP(hipGetLastError());
  double *Ac = A0, *An = A1, *Vc = V0, *Vn = V1, *nuc = nu0, *nun = nu1;
  const long long threads = (long long)half * half + (long long)half * ne;
  const unsigned grid = unsigned((threads + 255) / 256);
  for (int sweep = 0; sweep < 60; ++sweep) {
    ROM_HIP(hipMemsetAsync(d_any, 0, sizeof(int), ctx->stream));
    for (int r = 0; r < ne - 1; ++r) {
      kb_jgrid_round<<<grid, 256, 0, ctx->stream>>>(ne, half, r, Ac, An, Vc, Vn, nuc, nun, par, d_any);
      std::swap(Ac, An);
      std::swap(Vc, Vn);
      std::swap(nuc, nun);
    }
    ROM_HIP(hipGetLastError());
    int any = 0;
    ROM_HIP(hipMemcpyAsync(&any, d_any, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
    if (!any) break;
  }
  kb_jgrid_diag<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(n, ne, Ac, diag);
  ROM_HIP(hipGetLastError());
  std::vector<double> d(n);
  ROM_TRY(download(ctx, diag, d.data(), n));
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return d[a] > d[b]; });   // (descending, first index first on ties)
  ROM_HIP(hipMemcpyAsync(d_perm, perm.data(), size_t(n) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  const double lmax = d[perm[0]];
  std::vector<double> scale(n, 1.0);
  if (mode != SE_EIG)
    for (int i = 0; i < n; ++i) {
      const double l = d[perm[i]];
      const bool keep = l > rel_tol * lmax && l > 0.0;
      scale[i] = !keep ? 0.0 : mode == SE_WHITEN ? 1.0 / std::sqrt(l) : 1.0 / std::sqrt(std::sqrt(l));
    }
  Tmp scb;
  ROM_TRY(scb.get(ctx, n));
  ROM_HIP(hipMemcpyAsync(scb.p(), scale.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const unsigned gg = unsigned((size_t(n) * n + 255) / 256);
  if (mode != SE_LOWDIN) {
    kb_jgrid_gather<<<gg, 256, 0, ctx->stream>>>(n, ne, Vc, diag, d_perm, scb, T, ldt, lam);
    ROM_HIP(hipGetLastError());
  } else {
    // sum_i q_i q_i^T / sqrt(lam_i) = S^T S with S = rows q_i^T lam_i^(-1/4): gathered into one ping-pong buffer, transposed
    // into the other, one small product
    double* S = An;
    double* St = Vn;
    kb_jgrid_gather<<<gg, 256, 0, ctx->stream>>>(n, ne, Vc, diag, d_perm, scb, S, n, lam);
    kb_transpose<<<gg, 256, 0, ctx->stream>>>(St, n, S, n, n, n);
    ROM_HIP(hipGetLastError());
    ROM_TRY(rom_launch_gemm_nt(ctx, n, n, n, 1.0, St, n, St, n, 0.0, T, ldt, "gemm_nt"));
  }
  ROM_HIP(hipStreamSynchronize(ctx->stream));   // (perm / scale are host memory of this frame)
  return ROM_OK;
}

int romb_small_eig(rom_ctx* ctx, int n, const double* A, int lda, double* lam, double* T, int ldt, int mode, double rel_tol,
                     bool gram_like) {
  if (n <= 0) return ROM_OK;
  ROM_CHECK(n <= SE_GRID_MAX, "small symmetric eigenproblem: n = %d beyond %d", n, SE_GRID_MAX);
  if (n > SE_LDS_MAX) return jacobi_grid(ctx, n, A, lda, lam, T, ldt, mode, rel_tol, gram_like);   // (one launch per round, the whole chip)
  const int ld = n | 1, half = (n + (n & 1)) / 2;
  const size_t vec = (2 * size_t(half) + 2 * size_t(n) + 8) * sizeof(double) + (2 * size_t(half) + n + 2) * sizeof(int);
  size_t lds = vec + 16 + 2 * size_t(n) * ld * sizeof(double);   // (n <= SE_LDS_MAX here: matrix + eigenvector rows in LDS)
  double* ns_ws = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, 2 * size_t(n) * n, &ns_ws));   // Newton-Schulz fast path: next iterates
  if (lds > 64 * 1024 && !ctx->lds_optin_small_eig) {
    ROM_HIP(hipSetDevice(ctx->device));
    ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kb_small_eig<512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->lds_optin_small_eig = true;
  }
  {
    static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-shape names in the profile records
    char nm[48];
    detail ? snprintf(nm, sizeof nm, "small_eig_n%d_mode%d_%s", n, mode, gram_like ? "gram" : "sym") : snprintf(nm, sizeof nm, "small_eig");
    ROM_PROF(ctx, nm, 30.0 * n * n * n, 16.0 * n * n);
    // n <= 32: ONE wave -- no barrier between the phases of a round and the item map divided out once: 5x the rounds per
    // microsecond of the 512-thread form on the same rotations (same results)
    if (n <= 32 && mode == SE_EIG) kb_jacobi32<<<1, 256, 0, ctx->stream>>>(n, A, lda, lam, T, ldt, gram_like ? 1 : 0);
    else kb_small_eig<512, false><<<1, 512, lds, ctx->stream>>>(n, A, lda, lam, T, ldt, mode, rel_tol, gram_like ? 1 : 0, nullptr, ns_ws);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// test / diagnostic entry: eigen-decomposition of a small symmetric matrix given on the host
extern "C" int rom_small_eig_host(rom_ctx* ctx, int n, const double* A_host, int mode, double rel_tol, int gram_like,
                                  double* lam_host, double* T_host) {
  ROM_CHECK(ctx && A_host && lam_host && T_host && n >= 1 && n <= SE_MAX, "rom_small_eig_host: bad arguments");
  ROM_CHECK(mode >= 0 && mode <= 3, "rom_small_eig_host: mode must be 0, 1, 2 or 3");
  ROM_CHECK(mode != 3 || n <= SE_LDS_MAX, "rom_small_eig_host: the pivoted-Cholesky whitening takes n <= %d", SE_LDS_MAX);
  Tmp A, lam, T;
  ROM_TRY(A.get(ctx, size_t(n) * n));
  ROM_TRY(lam.get(ctx, n));
  ROM_TRY(T.get(ctx, size_t(n) * n));
  ROM_HIP(hipMemcpyAsync(A.p(), A_host, size_t(n) * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (mode == 3) ROM_TRY(romb_pivchol_whiten(ctx, n, A, n, lam, T, n, rel_tol));
  else ROM_TRY(romb_small_eig(ctx, n, A, n, lam, T, n, mode, rel_tol, gram_like != 0));
  ROM_TRY(download(ctx, lam, lam_host, n));
  return download(ctx, T, T_host, size_t(n) * n);
}

// =====================================================================================================================
// orthonormalisation helpers (rows of a (b, dim) block)
// =====================================================================================================================
// X (b x dim) <- T X with T from the b x b Gram matrix X X^T: SE_WHITEN (rank revealing: dependent directions become
// zero rows) or SE_LOWDIN (rows stay individually close to what they were).  `rounds` repetitions (the second one
// removes what the squared condition number of the first Gram matrix left).  Y is a scratch block of the same size; the
// result ends in X.
int romb_gram_transform(rom_ctx* ctx, double* X, double* Y, int b, int64_t dim, int mode, double rel_tol, int rounds) {
  if (b <= 0) return ROM_OK;
  Tmp G, lam, T;
  ROM_TRY(G.get(ctx, size_t(b) * b));
  ROM_TRY(lam.get(ctx, b));
  ROM_TRY(T.get(ctx, size_t(b) * b));
  for (int r = 0; r < rounds; ++r) {
    ROM_TRY(rom_launch_gram(ctx, b, dim, X, dim, G, b));
    if (mode == SE_WHITEN && b <= SE_LDS_MAX) ROM_TRY(romb_pivchol_whiten(ctx, b, G, b, lam, T, b, r == 0 ? rel_tol : 1e-8));
    else ROM_TRY(romb_small_eig(ctx, b, G, b, lam, T, b, mode, r == 0 ? rel_tol : 1e-8));
    ROM_TRY(rom_launch_gemm_nn(ctx, b, dim, b, 1.0, T, b, X, dim, 0.0, Y, dim));
    ROM_HIP(hipMemcpyAsync(X, Y, size_t(b) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return ROM_OK;
}

// rows V[found : found + take] <- orthonormal and orthogonal to the orthonormal rows V[0 : found]: block Gram-Schmidt
// against the old rows (twice), symmetric orthonormalisation among the new ones
int romb_orthonormalize_against(rom_ctx* ctx, double* V, int found, int take, int64_t dim) {
  if (take <= 0) return ROM_OK;
  double* Vn = V + size_t(found) * dim;
  if (found > 0) {
    Tmp C;
    ROM_TRY(C.get(ctx, size_t(take) * found));
    for (int r = 0; r < 2; ++r) {
      ROM_TRY(rom_launch_gemm_nt(ctx, take, found, dim, 1.0, Vn, dim, V, dim, 0.0, C, found, "gemm_nt"));
      ROM_TRY(rom_launch_gemm_nn(ctx, take, dim, found, -1.0, C, found, V, dim, 1.0, Vn, dim));
    }
  }
  Tmp Y;
  ROM_TRY(Y.get(ctx, size_t(take) * dim));
  // (the new rows are orthonormal to ~1e-6 at worst -- lifted Gram modes -- and the symmetric orthonormalisation is second
  // order in that defect: one round)
  return romb_gram_transform(ctx, Vn, Y, take, dim, SE_LOWDIN, 1e-30, 1);
}

// kb_cgs_finish: v <- v / ||v||_2 (norm squared given on the device), zero row if the norm underflows
__global__ void kb_scale_by_inv_norm(double* __restrict__ v, long long dim, const double* __restrict__ nrm2) {
  const double n2 = *nrm2;
  const double a = (n2 > 0.0 && sqrt(n2) > 1e-300) ? 1.0 / sqrt(n2) : 0.0;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < dim; j += (long long)gridDim.x * blockDim.x) v[j] *= a;
}

// orthonormalize_base (src/lib/ReducedBasis.py:18-21): rows of X -> Euclidean-orthonormal rows spanning the same NESTED
// subspaces (thin QR of X^T up to the sign of each row), by re-orthogonalised classical Gram-Schmidt; every step is a
// device operation, the norms never leave the device.
extern "C" int rom_orthonormalize_rows(rom_ctx* ctx, rom_buf* X, int64_t x_row0, int n, int64_t dim, rom_buf* Q, int64_t q_row0) {
  ROM_CHECK(ctx && X && Q, "rom_orthonormalize_rows: null argument");
  ROM_CHECK(n >= 0 && dim >= 1 && x_row0 >= 0 && q_row0 >= 0, "rom_orthonormalize_rows: bad sizes");
  ROM_CHECK(size_t(x_row0 + n) * dim <= X->n && size_t(q_row0 + n) * dim <= Q->n, "rom_orthonormalize_rows: rows out of range");
  if (n == 0) return ROM_OK;
  double* q = Q->p + q_row0 * dim;
  if (q != X->p + x_row0 * dim)
    ROM_HIP(hipMemcpyAsync(q, X->p + x_row0 * dim, size_t(n) * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  Tmp h, nrm;
  ROM_TRY(h.get(ctx, n));
  ROM_TRY(nrm.get(ctx, 1));
  const unsigned grid = unsigned(std::min<int64_t>((dim + 255) / 256, 512));
  for (int j = 0; j < n; ++j) {
    double* v = q + size_t(j) * dim;
    for (int r = 0; r < 2 && j > 0; ++r) {  // "twice is enough"
      ROM_TRY(rom_launch_gemm_nt(ctx, j, 1, dim, 1.0, q, dim, v, dim, 0.0, h, 1, "gemm_nt"));   // h = Q[:j] v
      ROM_TRY(rom_launch_gemm_nn(ctx, 1, dim, j, -1.0, h, j, q, dim, 1.0, v, dim));              // v -= h^T Q[:j]
    }
    ROM_TRY(rom_launch_l2norm(ctx, v, 1, dim, nrm, false));
    kb_scale_by_inv_norm<<<grid, 256, 0, ctx->stream>>>(v, dim, nrm);
    ROM_HIP(hipGetLastError());
  }
  return ROM_OK;
}

// nearly orthonormal rows -> orthonormal rows, each as close as possible to what it was (symmetric / Loewdin
// orthonormalisation: V <- (V V^T)^(-1/2) V, two rounds): the clean-up of modes that were expanded from another basis
extern "C" int rom_symmetric_orthonormalize(rom_ctx* ctx, rom_buf* V, int64_t v_row0, int n, int64_t dim) {
  ROM_CHECK(ctx && V, "rom_symmetric_orthonormalize: null argument");
  ROM_CHECK(n >= 0 && n <= SE_MAX && dim >= 1 && v_row0 >= 0 && size_t(v_row0 + n) * dim <= V->n, "rom_symmetric_orthonormalize: bad sizes");
  if (n == 0) return ROM_OK;
  Tmp Y;
  ROM_TRY(Y.get(ctx, size_t(n) * dim));
  return romb_gram_transform(ctx, V->p + v_row0 * dim, Y, n, dim, SE_LOWDIN, 1e-30, 2);
}

// rows V[found : found + rest] <- pseudo-random directions (seeded: deterministic) made orthonormal and orthogonal to the
// orthonormal rows V[0 : found]: how a mode request beyond what the data determine is completed (LAPACK and scikit-learn
// return SOME orthonormal directions there too)
extern "C" int rom_complete_orthonormal(rom_ctx* ctx, rom_buf* V, int64_t v_row0, int found, int rest, int64_t dim) {
  ROM_CHECK(ctx && V, "rom_complete_orthonormal: null argument");
  ROM_CHECK(found >= 0 && rest >= 0 && dim >= 1 && v_row0 >= 0 && size_t(v_row0 + found + rest) * dim <= V->n,
            "rom_complete_orthonormal: bad sizes");
  ROM_CHECK(int64_t(found) + rest <= dim && rest <= SE_MAX, "rom_complete_orthonormal: more rows than the space has dimensions");
  if (rest == 0) return ROM_OK;
  double* v = V->p + v_row0 * dim;
  ROM_TRY(romb_fill_random(ctx, v + size_t(found) * dim, size_t(rest) * dim, 0xc0de0000ull + unsigned(found), false));
  return romb_orthonormalize_against(ctx, v, found, rest, dim);
}

// =====================================================================================================================
// projectors
// =====================================================================================================================
// project_solutions (src/lib/SolutionsManagers.py:108-139): OUT_i = c_i C with (C A_1 C^T) c_i = C A_1 u_i
extern "C" int rom_project_h10(rom_fem* f, rom_buf* U, int64_t u_row0, int M, rom_buf* C, int64_t c_row0, int n,
                               rom_buf* OUT, int64_t out_row0) {
  ROM_CHECK(f && U && OUT && (C || n == 0), "rom_project_h10: null argument");
  ROM_CHECK(M >= 0 && n >= 0 && u_row0 >= 0 && c_row0 >= 0 && out_row0 >= 0, "rom_project_h10: negative size");
  ROM_CHECK(n <= 4096, "rom_project_h10: basis of %d vectors (at most 4096)", n);
  const int64_t dim = f->dim;
  ROM_CHECK(size_t(u_row0 + M) * dim <= U->n && size_t(out_row0 + M) * dim <= OUT->n && (n == 0 || size_t(c_row0 + n) * dim <= C->n),
            "rom_project_h10: rows out of range");
  rom_ctx* ctx = f->ctx;
  if (M == 0) return ROM_OK;
  double* out = OUT->p + out_row0 * dim;
  if (n == 0) return fill(ctx, out, size_t(M) * dim, 0.0);  // (:109-111)
  const double* u = U->p + u_row0 * dim;
  const double* c = C->p + c_row0 * dim;
  Tmp AC, G, R, ones, coef;
  ROM_TRY(AC.get(ctx, size_t(n) * dim));
  ROM_TRY(G.get(ctx, size_t(n) * n));
  ROM_TRY(R.get(ctx, size_t(M) * n));
  ROM_TRY(ones.get(ctx, M));
  ROM_TRY(coef.get(ctx, size_t(M) * n));
  ROM_TRY(rom_launch_stencil_apply(f, nullptr, c, n, AC));                                           // A_1 C^T (:123)
  ROM_TRY(rom_launch_gemm_nt(ctx, n, n, dim, 1.0, AC, dim, c, dim, 0.0, G, n, "gemm_nt"));            // C A_1 C^T (:136)
  ROM_TRY(rom_launch_gemm_nt(ctx, M, n, dim, 1.0, u, dim, AC, dim, 0.0, R, n, "gemm_nt"));            // rhs (:113-124)
  ROM_TRY(fill(ctx, ones, M, 1.0));
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  ROM_TRY(rom_launch_reduced_solve(ctx, n, n, 1, M, G, ones, R, 1, coef));                            // (:135-138)
  ROM_TRY(rom_launch_gemm_nn(ctx, M, dim, n, 1.0, coef, n, c, dim, 0.0, out, dim));                   // (:139)
  return read_status(ctx, "rom_project_h10");
}

// reduced tensor Ahat[b] = C A_b C^T (k, n, n) and b_hat = C B_total of generate_fm_solutions (:93-103)
static int reduced_tensor(rom_fem* f, const double* c, int n, double* Ahat, double* bhat) {
  rom_ctx* ctx = f->ctx;
  const int k = f->nrb * f->ncb;
  const int64_t dim = f->dim;
  Tmp AC, onehot, Bt;
  ROM_TRY(AC.get(ctx, size_t(n) * dim));
  ROM_TRY(onehot.get(ctx, size_t(k) * k));
  ROM_TRY(Bt.get(ctx, dim));
  std::vector<double> eye(size_t(k) * k, 0.0);
  for (int b = 0; b < k; ++b) eye[size_t(b) * k + b] = 1.0;
  ROM_HIP(hipMemcpyAsync(onehot.p(), eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));  // (the host vector goes out of scope)
  for (int b = 0; b < k; ++b) {
    ROM_TRY(rom_launch_stencil_apply(f, onehot.p() + size_t(b) * k, c, n, AC));
    ROM_TRY(rom_launch_gemm_nt(ctx, n, n, dim, 1.0, AC, dim, c, dim, 0.0, Ahat + size_t(b) * n * n, n, "gemm_nt"));
  }
  ROM_TRY(fill(ctx, Bt, dim, 1.0 / (double(f->N) * f->N)));  // B_total = h^2 (:177-185)
  return rom_launch_gemm_nt(ctx, n, 1, dim, 1.0, c, dim, Bt, dim, 0.0, bhat, 1, "gemm_nt");
}

// generate_fm_solutions (:88-106): OUT_m = c_m C with (sum_b a[m,b] C A_b C^T) c_m = C B_total
extern "C" int rom_galerkin_rom(rom_fem* f, rom_buf* a, int M, rom_buf* C, int64_t c_row0, int n, rom_buf* OUT, int64_t out_row0) {
  ROM_CHECK(f && a && OUT && (C || n == 0), "rom_galerkin_rom: null argument");
  ROM_CHECK(M >= 0 && n >= 0 && c_row0 >= 0 && out_row0 >= 0, "rom_galerkin_rom: negative size");
  ROM_CHECK(n <= 4096, "rom_galerkin_rom: basis of %d vectors (at most 4096)", n);
  const int64_t dim = f->dim;
  const int k = f->nrb * f->ncb;
  ROM_CHECK(size_t(M) * k <= a->n && size_t(out_row0 + M) * dim <= OUT->n && (n == 0 || size_t(c_row0 + n) * dim <= C->n),
            "rom_galerkin_rom: buffers too small");
  rom_ctx* ctx = f->ctx;
  if (M == 0) return ROM_OK;
  double* out = OUT->p + out_row0 * dim;
  if (n == 0) return fill(ctx, out, size_t(M) * dim, 0.0);  // (:89-91)
  const double* c = C->p + c_row0 * dim;
  Tmp Ahat, bhat, coef;
  ROM_TRY(Ahat.get(ctx, size_t(k) * n * n));
  ROM_TRY(bhat.get(ctx, n));
  ROM_TRY(coef.get(ctx, size_t(M) * n));
  ROM_TRY(reduced_tensor(f, c, n, Ahat, bhat));
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  ROM_TRY(rom_launch_reduced_solve(ctx, n, n, k, M, Ahat, a->p, bhat, 0, coef));   // (:104-105)
  ROM_TRY(rom_launch_gemm_nn(ctx, M, dim, n, 1.0, coef, n, c, dim, 0.0, out, dim));  // (:106)
  return read_status(ctx, "rom_galerkin_rom");
}

// =====================================================================================================================
// strong greedy (src/lib/ReducedBasis.py:112-139)
// =====================================================================================================================
// The reference recomputes, in every iteration, the approximation of all M training snapshots in the current basis
// (H^1_0 projection or Galerkin ROM) and the H^1_0 norm of the differences.  Both depend on the SPAN of the basis only,
// so the span is carried as an A_1-orthonormal basis W (w_j^T A_1 w_i = delta_ij) and the projection residuals
//     R_m = u_m - sum_j p_mj w_j ,   p_mj = <u_m, w_j>_{A_1}
// are kept in HBM and UPDATED by one vector per iteration (modified Gram-Schmidt over the training block), ONE pass over
// the block per iteration.  With w_j = R_pick / ||R_pick||_A (the residual of the pick is already orthogonal to W; one
// re-orthogonalisation) and z_j = A_1 w_j, pass j
//     applies the PREVIOUS update          R_m <- R_m - p_{m,j-1} w_{j-1}        (read + write)
//     takes the exact norm of the result   N_m = ||R_m||_A^2                      (edge form, no cancellation)
//     and the new coefficient              p_mj = R_m . z_j
// so that the residual norm after update j is  ||R_m - p_mj w_j||_A^2 = N_m - p_mj^2  (w_j is A_1-orthogonal to R_m's
// complement): ONE step of Pythagoras from a norm that is exact -- its cancellation error is eps N_m, i.e. a relative
// error eps (||R_before|| / ||R_after||)^2 of the new norm, harmless unless a single basis vector takes a residual
// down by more than four orders of magnitude (rows that happens to -- the pick itself, duplicates of it -- are not the
// next maximum), and it does not accumulate: the next pass measures N_m afresh.
// The Galerkin approximation g_m = sum_j c_mj w_j differs from the projection inside span W only, so
//     ||u_m - g_m||_A^2 = ||R_m||_A^2 + sum_j (p_mj - c_mj)^2      (both terms >= 0)
// with c_m from the reduced systems in the W basis, whose tensor W A_b W^T grows by one row per iteration.
// The reference's dense contractions are O(n M dim) per iteration; this is one read and one write of the block.

// err[m] = sqrt(nrm2[m]); rel = err / h1; first maximum -> picks[it], maxerr[it]; one workgroup
__global__ __launch_bounds__(1024) void kb_greedy_select(int M, const double* __restrict__ err2, const double* __restrict__ extra2,
                                                         const double* __restrict__ h1, int it, int* __restrict__ picks,
                                                         double* __restrict__ maxerr) {
  __shared__ double bv[1024];
  __shared__ int bi[1024];
  double best = -1.0;
  int at = 0;
  for (int m = threadIdx.x; m < M; m += 1024) {
    const double e2 = err2[m] + (extra2 ? extra2[m] : 0.0);
    const double rel = sqrt(e2) / h1[m];
    if (rel > best) { best = rel; at = m; }  // ascending m per thread: keeps the first maximum
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = at;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) {
      const double o = bv[threadIdx.x + s];
      const int oi = bi[threadIdx.x + s];
      if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = o; bi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    picks[it] = bi[0];
    maxerr[it] = bv[0];
  }
}

// w <- (Rs[pick] - pc w_prev) / ||.||_A, pc = p_prev[pick] (the one update the block in HBM does not hold yet);
// a pick whose residual is at roundoff of its snapshot (a duplicate) gives w = 0
__global__ void kb_take_pick(double* __restrict__ w, const double* __restrict__ Rs, const double* __restrict__ w_prev,
                             const double* __restrict__ p_prev, long long dim, const int* __restrict__ picks, int it,
                             const double* __restrict__ err2, const double* __restrict__ norm0sq, int* __restrict__ degenerate) {
  const int p = picks[it];
  const double e2 = err2[p];
  const bool dead = !(e2 > 1e-26 * norm0sq[p]) || !(e2 > 0.0);
  const double a = dead ? 0.0 : 1.0 / sqrt(e2);
  if (blockIdx.x == 0 && threadIdx.x == 0) degenerate[it] = dead ? 1 : 0;
  const double* r = Rs + (long long)p * dim;
  const double pc = w_prev ? p_prev[p] : 0.0;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < dim; j += (long long)gridDim.x * blockDim.x)
    w[j] = a * (w_prev ? r[j] - pc * w_prev[j] : r[j]);
}

// w <- w / sqrt(nrm2) unless the vector is dead
__global__ void kb_renormalise(double* __restrict__ w, long long dim, const double* __restrict__ nrm2,
                               const int* __restrict__ degenerate, int it) {
  const double n2 = *nrm2;
  const double a = (degenerate[it] || !(n2 > 0.0)) ? 0.0 : 1.0 / sqrt(n2);
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < dim; j += (long long)gridDim.x * blockDim.x) w[j] *= a;
}

// The pass of one greedy iteration over the training block (see above): d = Rs_m - p_prev[m] w_prev (written to Rout when
// there is a previous update), partial sums of ||d||_A^2 in edge form and of d . z.  Layout of k_h10_partial
// (rom_ops.hip): a thread owns a mesh column and walks down a slab of rows, every entry is read once, the halo row above
// the slab a second time; the east neighbour comes from the next lane, across waves through LDS, across workgroups by a
// load of the workgroup's last thread.  The update goes to a second buffer, not in place: halo entries belong to other
// workgroups.  partial: [2][M][nblk] (norms, dots).
// GU_SYS training rows per workgroup share the loads of w_prev and z (L2 resident, but two of the three loads per entry
// when every row fetches its own).  Measured at C4 (1024 x 262 144, ms per pass; tools/dev/ab_greedy.sh): GU_SYS x
// GU_UNROLL = 1 x 2..5: 1.15-1.17, 1 x 8: 1.59, 2 x 4: 1.46, 4 x 2: 1.58, 4 x 4: 2.28 (187 VGPRs) -- the pass lives on
// occupancy (waves in flight hide the HBM latency), not on fewer L2 reads, so the default shares nothing.
#ifndef GU_SYS_
#define GU_SYS_ 1
#endif
#ifndef GU_UNROLL_
#define GU_UNROLL_ 2
#endif
constexpr int GU_ROWS = 32, GU_UNROLL = GU_UNROLL_, GU_SYS = GU_SYS_;
template <bool PREV>
__global__ __launch_bounds__(256) void kb_greedy_pass(StencilGeom g, const double* __restrict__ Rs, double* __restrict__ Rout,
                                                      const double* __restrict__ w_prev, const double* __restrict__ p_prev,
                                                      const double* __restrict__ z, double* __restrict__ partial, int nblk, int M) {
  __shared__ double red[2][GU_SYS][4];
  __shared__ double west[2][4][GU_UNROLL][GU_SYS];  // first column of every wave, per row of the chunk (double buffered)
  const int m0 = blockIdx.z * GU_SYS;
  const double* u[GU_SYS];
  double* o[GU_SYS];
  double pm[GU_SYS];
  bool live[GU_SYS];
#pragma unroll
  for (int k = 0; k < GU_SYS; ++k) {
    const int mk = min(m0 + k, M - 1);  // (the last group repeats its last row; the repeats are neither stored nor reported)
    live[k] = m0 + k < M;
    u[k] = Rs + mk * g.dim;
    o[k] = PREV ? Rout + mk * g.dim : nullptr;
    pm[k] = PREV ? p_prev[mk] : 0.0;
  }
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * GU_ROWS, r1 = min(g.nr, r0 + GU_ROWS);
  const bool in = c < g.nc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[GU_SYS], dt[GU_SYS], north[GU_SYS];
  {
    const long long i = (long long)(r0 - 1) * g.nc + c;
    const bool ok = in && r0 > 0;
    const double wn = (PREV && ok) ? w_prev[i] : 0.0;
#pragma unroll
    for (int k = 0; k < GU_SYS; ++k) {
      s[k] = dt[k] = 0.0;
      north[k] = ok ? (PREV ? u[k][i] - pm[k] * wn : u[k][i]) : 0.0;
    }
  }
  int buf = 0;
  for (int rb = r0; rb < r1; rb += GU_UNROLL, buf ^= 1) {
    double x[GU_SYS][GU_UNROLL], xe[GU_SYS][GU_UNROLL], zz[GU_UNROLL];
#pragma unroll
    for (int q = 0; q < GU_UNROLL; ++q) {
      const int r = rb + q;
      const long long i = (long long)r * g.nc + c;
      const bool ok = in && r < r1;
      const bool edge = threadIdx.x == 255 && c + 1 < g.nc && r < r1;
      const double wv = (PREV && ok) ? w_prev[i] : 0.0;
      const double we = (PREV && edge) ? w_prev[i + 1] : 0.0;
      zz[q] = ok ? z[i] : 0.0;
#pragma unroll
      for (int k = 0; k < GU_SYS; ++k) {
        x[k][q] = ok ? (PREV ? u[k][i] - pm[k] * wv : u[k][i]) : 0.0;
        xe[k][q] = edge ? (PREV ? u[k][i + 1] - pm[k] * we : u[k][i + 1]) : 0.0;
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < GU_UNROLL; ++q)
#pragma unroll
        for (int k = 0; k < GU_SYS; ++k) west[buf][wave][q][k] = x[k][q];
    }
    // (uniform trip count: r0, r1 depend on the block only; the other buffer is read one barrier later.  The barrier
    // waits for the LDS writes only: __syncthreads() would also drain every global load in flight -- whatever the compiler
    // has hoisted from the next chunk -- once per four rows)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < GU_UNROLL; ++q) {
      const int r = rb + q;
      const bool ok = in && r < r1;
#pragma unroll
      for (int k = 0; k < GU_SYS; ++k) {
        double east = __shfl_down(x[k][q], 1, 64);
        if (lane == 63) east = wave < 3 ? west[buf][wave + 1][q][k] : xe[k][q];
        if (ok) {
          if (PREV && live[k]) __builtin_nontemporal_store(x[k][q], &o[k][(long long)r * g.nc + c]);  // (read by the next pass only)
          if (c + 1 >= g.nc) east = 0.0;
          const double dh = x[k][q] - east, dv = x[k][q] - north[k];
          s[k] += dh * dh + dv * dv;
          if (c == 0) s[k] += x[k][q] * x[k][q];
          if (r == g.nr - 1) s[k] += x[k][q] * x[k][q];
          dt[k] += x[k][q] * zz[q];
          north[k] = x[k][q];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < GU_SYS; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s[k] += __shfl_down(s[k], off, 64);
      dt[k] += __shfl_down(dt[k], off, 64);
    }
    if (lane == 0) {
      red[0][k][wave] = s[k];
      red[1][k][wave] = dt[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * GU_SYS) {
    const int which = threadIdx.x / GU_SYS, k = threadIdx.x % GU_SYS;
    if (m0 + k < M) {
      const long long at_ = (which * (long long)M + m0 + k) * nblk + blockIdx.y * gridDim.x + blockIdx.x;
      partial[at_] = red[which][k][0] + red[which][k][1] + red[which][k][2] + red[which][k][3];
    }
  }
}

// N_m = sum of the norm partials, p_m = sum of the dot partials, err2_m = max(N_m - p_m^2, 0)
__global__ __launch_bounds__(256) void kb_greedy_finish(const double* __restrict__ partial, int nblk, int M,
                                                        double* __restrict__ p_new, double* __restrict__ err2) {
  __shared__ double red[2][4];
  double s = 0.0, d = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    s += partial[blockIdx.x * (long long)nblk + i];
    d += partial[(M + blockIdx.x) * (long long)nblk + i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    d += __shfl_down(d, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double N = red[0][0] + red[0][1] + red[0][2] + red[0][3], pp = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    p_new[blockIdx.x] = pp;
    err2[blockIdx.x] = fmax(N - pp * pp, 0.0);
  }
}

// row / column j of the reduced tensor (k, ld, ld) from col[i, b] = w_i^T A_b w_j (i <= j); a dead vector gets a unit
// diagonal and zero couplings, which keeps every reduced matrix positive definite and its coefficient at 0
__global__ void kb_grow_ahat(double* __restrict__ Ahat, int k, int ld, int j, const double* __restrict__ col,
                             const int* __restrict__ degenerate, int it) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (j + 1) * k) return;
  const int i = idx / k, b = idx % k;
  const bool dead = degenerate[it] != 0;
  const double v = dead ? (i == j ? 1.0 : 0.0) : col[idx];
  Ahat[(size_t(b) * ld + j) * ld + i] = v;
  Ahat[(size_t(b) * ld + i) * ld + j] = v;
}

// extra2[m] = sum_{j < n} (P[j, m] - c[m, j])^2   (P: row j = projection coefficients of basis vector j; c: packed n)
__global__ void kb_galerkin_gap(int M, int n, const double* __restrict__ P, const double* __restrict__ c,
                                double* __restrict__ extra2) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  for (int j = 0; j < n; ++j) {
    const double d = P[size_t(j) * M + m] - c[size_t(m) * n + j];
    s += d * d;
  }
  extra2[m] = s;
}

__global__ void kb_ints_to_doubles(const int* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = double(src[i]);
}

// mode 0: greedy for the H^1_0 projection error; 1: for the Galerkin error.  h1norm: the normalisation the caller passes
// as solutions2train_h1norm (M doubles on the host).  picks_out / max_err_out: n entries.
extern "C" int rom_greedy(rom_fem* f, rom_buf* U, int64_t u_row0, int M, rom_buf* a, const double* h1norm_host, int mode,
                          int n, int64_t* picks_out, double* max_err_out) {
  ROM_CHECK(f && U && h1norm_host && (picks_out || n == 0) && (max_err_out || n == 0), "rom_greedy: null argument");
  ROM_CHECK(mode == 0 || mode == 1, "rom_greedy: mode must be 0 (H10 projection) or 1 (Galerkin)");
  ROM_CHECK(mode == 0 || a, "rom_greedy: the Galerkin greedy needs the training parameters");
  ROM_CHECK(M >= 1 && M <= 65535 && n >= 0 && u_row0 >= 0, "rom_greedy: bad sizes (1 <= M <= 65535)");
  ROM_CHECK(n <= 2048, "rom_greedy: at most 2048 basis vectors");
  const int64_t dim = f->dim;
  const int k = f->nrb * f->ncb;
  ROM_CHECK(size_t(u_row0 + M) * dim <= U->n && (!a || size_t(M) * k <= a->n), "rom_greedy: buffers too small");
  if (n == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  const StencilGeom g = rom_make_geom(f->nrb, f->ncb, f->N);
  const double* u = U->p + u_row0 * dim;
  const int nb = std::max(n - 1, 1);  // basis vectors ever built
  Tmp Ra, Rb, W, AW, P, err2, norm0, h1, ipick, idead, maxerr, t, nrm, part, Ahat, bhat, cg, extra, ZB, col, onehot, Bt;
  ROM_TRY(Ra.get(ctx, size_t(M) * dim));
  ROM_TRY(Rb.get(ctx, size_t(M) * dim));
  ROM_TRY(W.get(ctx, size_t(nb) * dim));
  ROM_TRY(AW.get(ctx, size_t(nb) * dim));
  ROM_TRY(P.get(ctx, size_t(nb) * M));   // row j: p_mj = <R_m, w_j>_A, the projection coefficients
  ROM_TRY(err2.get(ctx, M));
  ROM_TRY(norm0.get(ctx, M));
  ROM_TRY(h1.get(ctx, M));
  ROM_TRY(ipick.get(ctx, n));   // n ints (picks) in a block of n doubles
  ROM_TRY(idead.get(ctx, n));   // n ints (1: the vector built from this pick is dead)
  ROM_TRY(maxerr.get(ctx, n));
  ROM_TRY(t.get(ctx, nb));
  ROM_TRY(nrm.get(ctx, 1));
  const dim3 ugrid((g.nc + 255) / 256, (g.nr + GU_ROWS - 1) / GU_ROWS, (M + GU_SYS - 1) / GU_SYS);
  const int nblk = int(ugrid.x * ugrid.y);
  ROM_TRY(part.get(ctx, 2 * size_t(M) * nblk));
  int* d_picks = reinterpret_cast<int*>(ipick.p());
  int* d_dead = reinterpret_cast<int*>(idead.p());
  double* d_maxerr = maxerr.p();
  if (mode == 1) {
    ROM_TRY(Ahat.get(ctx, size_t(k) * nb * nb));
    ROM_TRY(bhat.get(ctx, nb));
    ROM_TRY(cg.get(ctx, size_t(M) * nb));
    ROM_TRY(extra.get(ctx, M));
    ROM_TRY(ZB.get(ctx, size_t(k) * dim));
    ROM_TRY(col.get(ctx, size_t(nb) * k));
    ROM_TRY(onehot.get(ctx, size_t(k) * k));
    ROM_TRY(Bt.get(ctx, dim));
    ROM_TRY(fill(ctx, Ahat, size_t(k) * nb * nb, 0.0));
    ROM_TRY(fill(ctx, Bt, dim, 1.0 / (double(f->N) * f->N)));
    std::vector<double> eye(size_t(k) * k, 0.0);
    for (int b = 0; b < k; ++b) eye[size_t(b) * k + b] = 1.0;
    ROM_HIP(hipMemcpyAsync(onehot.p(), eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  ROM_HIP(hipMemcpyAsync(h1.p(), h1norm_host, size_t(M) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));  // the caller's array is free again
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  ROM_HIP(hipMemsetAsync(ipick.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemsetAsync(idead.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  ROM_HIP(hipMemsetAsync(maxerr.p(), 0, size_t(n) * sizeof(double), ctx->stream));
  // empty basis (:120-129 with basis (0, 0)): the approximation is zero, the error of snapshot m is ||u_m||_A / h1_m --
  // the same kernel as rom_h10norm, so a caller that passes sm.H10norm(U) gets exactly 1.0 everywhere and index 0
  ROM_TRY(rom_launch_h10norm(f, u, nullptr, M, norm0, false));
  ROM_HIP(hipMemcpyAsync(err2.p(), norm0.p(), size_t(M) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  kb_greedy_select<<<1, 1024, 0, ctx->stream>>>(M, err2, nullptr, h1, 0, d_picks, d_maxerr);
  ROM_HIP(hipGetLastError());
  const double* Rs = u;  // the block in HBM: all updates but the latest applied (the snapshots themselves are never written)
  double* bufs[2] = {Ra.p(), Rb.p()};
  const double *w_prev = nullptr, *p_prev = nullptr;
  const unsigned vgrid = unsigned(std::min<int64_t>((dim + 255) / 256, 512));
  for (int it = 1; it < n; ++it) {
    const int j = it - 1;  // index of the basis vector built from pick it - 1
    double* wj = W.p() + size_t(j) * dim;
    double* zj = AW.p() + size_t(j) * dim;
    double* pj = P.p() + size_t(j) * M;
    kb_take_pick<<<vgrid, 256, 0, ctx->stream>>>(wj, Rs, w_prev, p_prev, dim, d_picks, it - 1, err2, norm0, d_dead);
    ROM_HIP(hipGetLastError());
    if (j > 0) {  // one re-orthogonalisation against W in the A_1 inner product, then renormalise
      ROM_TRY(rom_launch_rowdot(ctx, AW, j, dim, wj, t));                                    // t_i = <w_i, w>_A
      ROM_TRY(rom_launch_gemm_nn(ctx, 1, dim, j, -1.0, t, j, W, dim, 1.0, wj, dim));         // w -= t^T W
      ROM_TRY(rom_launch_h10norm(f, wj, nullptr, 1, nrm, false));
      kb_renormalise<<<vgrid, 256, 0, ctx->stream>>>(wj, dim, nrm, d_dead, it - 1);
      ROM_HIP(hipGetLastError());
    }
    ROM_TRY(rom_launch_stencil_apply(f, nullptr, wj, 1, zj));                                // z = A_1 w
    {
      ROM_PROF(ctx, w_prev ? "greedy_pass" : "greedy_pass_first", 16.0 * double(M) * dim, (w_prev ? 16.0 : 8.0) * double(M) * dim);
      if (w_prev) {
        double* Rnext = bufs[it & 1];
        kb_greedy_pass<true><<<ugrid, 256, 0, ctx->stream>>>(g, Rs, Rnext, w_prev, p_prev, zj, part, nblk, M);
        Rs = Rnext;
      } else {
        kb_greedy_pass<false><<<ugrid, 256, 0, ctx->stream>>>(g, Rs, nullptr, nullptr, nullptr, zj, part, nblk, M);
      }
      kb_greedy_finish<<<M, 256, 0, ctx->stream>>>(part, nblk, M, pj, err2);
    }
    ROM_HIP(hipGetLastError());
    w_prev = wj;
    p_prev = pj;
    if (mode == 1) {
      ROM_TRY(rom_launch_stencil_apply_blocks(f, onehot, wj, ZB));                                       // A_b w_j, all b
      ROM_TRY(rom_launch_gemm_nt(ctx, j + 1, k, dim, 1.0, W, dim, ZB, dim, 0.0, col, k, "gemm_nt"));      // col[i, b] = w_i . A_b w_j
      kb_grow_ahat<<<unsigned(((j + 1) * k + 255) / 256), 256, 0, ctx->stream>>>(Ahat, k, nb, j, col, d_dead, it - 1);
      ROM_HIP(hipGetLastError());
      ROM_TRY(rom_launch_rowdot(ctx, wj, 1, dim, Bt, bhat.p() + j));                                      // w_j . B_total
      ROM_TRY(rom_launch_reduced_solve(ctx, j + 1, nb, k, M, Ahat, a->p, bhat, 0, cg));
      kb_galerkin_gap<<<unsigned((M + 255) / 256), 256, 0, ctx->stream>>>(M, j + 1, P, cg, extra);
      ROM_HIP(hipGetLastError());
    }
    kb_greedy_select<<<1, 1024, 0, ctx->stream>>>(M, err2, mode == 1 ? extra.p() : nullptr, h1, it, d_picks, d_maxerr);
    ROM_HIP(hipGetLastError());
  }
  // results: picks as doubles through one download
  Tmp outd;
  ROM_TRY(outd.get(ctx, 2 * size_t(n)));
  kb_ints_to_doubles<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(d_picks, outd, n);
  ROM_HIP(hipMemcpyAsync(outd.p() + n, d_maxerr, size_t(n) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  std::vector<double> host(2 * size_t(n));
  ROM_TRY(download(ctx, outd, host.data(), host.size()));
  for (int i = 0; i < n; ++i) {
    picks_out[i] = int64_t(host[i]);
    max_err_out[i] = host[n + i];
  }
  return read_status(ctx, "rom_greedy");
}
