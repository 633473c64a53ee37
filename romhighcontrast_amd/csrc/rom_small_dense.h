// Small dense device routines shared by the basis-stage kernels (rom_basis.hip, rom_pod.hip): the Jacobi eigensolver of
// matrices of order <= 32 as a __device__ function on an LDS-resident matrix, so that a kernel that has just produced a
// small Gram matrix (the one-workgroup Rayleigh-Ritz loop of rom_pod.hip) diagonalises it in the same launch.
#pragma once
#include <hip/hip_runtime.h>

// 1 / sqrt(x) to full precision from the hardware estimate (two Newton steps): the cosine of a rotation must satisfy
// c^2 (1 + t^2) = 1 to rounding, or the accumulated eigenvector rows drift from orthonormality
__device__ inline double se_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

constexpr int J32_LD = 33;   // row stride of the LDS matrices (odd: column walks are conflict free)
struct Jacobi32Lds {
  double As[32 * J32_LD], Vt[32 * J32_LD];
  double2 csv[16];
  int2 pq[16];
  double nu2[32], ev[32];
  int perm[32];
  int any;
};

// Cyclic Jacobi for a symmetric matrix of order n <= 32 that sits in L.As (both triangles; L.Vt receives the eigenvector
// rows), NT threads (64: one wave, no barrier instruction at all; 256: the items of a round are one 2 x 2 block and two
// eigenvector entries per lane).  A branch-free round: the round-robin pairs of a round are at most 16, so its 2 x 2
// blocks (<= 256) and eigenvector items (<= 512) are dealt to the lanes by a map that never changes (divided out once);
// every item loads its rotation (c, s) and its index pair unconditionally -- (1, 0) is an exact identity, so pairs that
// do not rotate need no branch -- and all loads of a phase are issued before its first store.  A rotation is applied
// when |a_pq| > tol sqrt(|a_pp a_qq|) -- the relative criterion under which graded matrices keep their small eigenvalues
// (Demmel-Veselic) -- AND |a_pq| > tol nu_p nu_q, the rounding noise of the entry (gram_like: nu_i^2 = |a_ii| at the
// start, rotated along; else max |a_ii|).  On return: eigenvalues on the diagonal of L.As, descending order in L.perm
// (L.ev[L.perm[i]] is the i-th largest), every thread past a barrier.  dmax = max |a_ii| (the caller has it from the load).
// Returns (uniformly) whether any rotation was applied: false = the matrix was diagonal by the criterion on entry.
template <int NT>
__device__ inline bool jacobi32_run(int n, Jacobi32Lds& L, int gram_like, double dmax) {
  constexpr int LD = J32_LD;
  constexpr int BI = 256 / NT, VI = 512 / NT;   // 2 x 2 blocks / eigenvector items of a lane
  double* As = L.As;
  double* Vt = L.Vt;
  const int t = threadIdx.x, ne = n + (n & 1), half = ne / 2;
  if (t < n) L.nu2[t] = gram_like ? fabs(As[t * LD + t]) : dmax;
  int bk[BI], bl[BI], vk[VI], vj[VI];
  bool bon[BI], von[VI];
#pragma unroll
  for (int u = 0; u < BI; ++u) {
    const int idx = u * NT + t;
    bon[u] = idx < half * half;
    bk[u] = bon[u] ? idx / half : 0;
    bl[u] = bon[u] ? idx - bk[u] * half : 0;
  }
#pragma unroll
  for (int u = 0; u < VI; ++u) {
    const int idx = u * NT + t;
    von[u] = idx < half * n;
    vk[u] = von[u] ? idx / n : 0;
    vj[u] = von[u] ? idx - vk[u] * n : 0;
  }
  const double tol = double(n > 8 ? n : 8) * 1.1e-16, tol2 = tol * tol, floor_abs = fmax(1e-300, 1e-40 * dmax);
  __syncthreads();
  bool ever = false;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int r = 0; r < ne - 1; ++r) {
      if (t < half) {
        int p = ne - 1, q = r;
        if (t != 0) {
          p = r + t;
          if (p >= ne - 1) p -= ne - 1;
          q = r - t;
          if (q < 0) q += ne - 1;
        }
        if (p > q) { const int x = p; p = q; q = x; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double app = As[p * LD + p], aqq = As[q * LD + q], apq = As[p * LD + q];
          const double np2 = L.nu2[p], nq2 = L.nu2[q], apq2 = apq * apq;
          if (apq2 > tol2 * fabs(app * aqq) && fabs(apq) > floor_abs && apq2 > tol2 * np2 * nq2) {
            // t = tan(phi) = sign(a) b / (|a| + sqrt(a^2 + b^2)), a = a_qq - a_pp, b = 2 a_pq; c = 1 / sqrt(1 + t^2), s = t c
            const double a = aqq - app, bb = 2.0 * apq;
            const double tt = (a >= 0 ? bb : -bb) / (fabs(a) + sqrt(a * a + bb * bb));
            c = se_rsqrt(1.0 + tt * tt);
            s = tt * c;
            L.nu2[p] = c * c * np2 + s * s * nq2;
            L.nu2[q] = s * s * np2 + c * c * nq2;
            rotated = true;
          }
        } else {
          q = -1;   // p is paired with the dummy index of an odd n: no rotation
        }
        L.csv[t] = make_double2(c, s);
        L.pq[t] = make_int2(p, q);
      }
      __syncthreads();
      {
        // A <- J^T A J as disjoint 2 x 2 blocks (k, l): rows (p_k, q_k) x columns (p_l, q_l) become R_k B R_l^T
        double o00[BI], o01[BI], o10[BI], o11[BI];
        int a00[BI], a01[BI], a10[BI], a11[BI];
        bool w01[BI], w10[BI], w11[BI];
#pragma unroll
        for (int u = 0; u < BI; ++u) {
          const double2 rk = L.csv[bk[u]], rl = L.csv[bl[u]];
          const int2 ik = L.pq[bk[u]], il = L.pq[bl[u]];
          const bool vk_ = ik.y >= 0, vl_ = il.y >= 0;
          const int pk = ik.x, qk = vk_ ? ik.y : ik.x, pl = il.x, ql = vl_ ? il.y : il.x;
          a00[u] = pk * LD + pl;
          a01[u] = pk * LD + ql;
          a10[u] = qk * LD + pl;
          a11[u] = qk * LD + ql;
          const double b00 = As[a00[u]], b01 = vl_ ? As[a01[u]] : 0.0;
          const double b10 = vk_ ? As[a10[u]] : 0.0, b11 = (vk_ && vl_) ? As[a11[u]] : 0.0;
          const double ck = rk.x, sk = rk.y, cl = rl.x, sl = rl.y;
          const double r00 = ck * b00 - sk * b10, r01 = ck * b01 - sk * b11;
          const double r10 = sk * b00 + ck * b10, r11 = sk * b01 + ck * b11;
          o00[u] = cl * r00 - sl * r01;
          o01[u] = sl * r00 + cl * r01;
          o10[u] = cl * r10 - sl * r11;
          o11[u] = sl * r10 + cl * r11;
          w01[u] = bon[u] && vl_;
          w10[u] = bon[u] && vk_;
          w11[u] = bon[u] && vk_ && vl_;
        }
        // eigenvector rows: Vt <- J^T Vt
        double op[VI], oq[VI];
        int ap[VI], aq[VI];
        bool wq[VI];
#pragma unroll
        for (int u = 0; u < VI; ++u) {
          const double2 rk = L.csv[vk[u]];
          const int2 ik = L.pq[vk[u]];
          const bool vq = ik.y >= 0;
          ap[u] = ik.x * LD + vj[u];
          aq[u] = (vq ? ik.y : ik.x) * LD + vj[u];
          const double vp = Vt[ap[u]], vqv = Vt[aq[u]];
          op[u] = rk.x * vp - rk.y * vqv;
          oq[u] = rk.y * vp + rk.x * vqv;
          wq[u] = von[u] && vq;
        }
#pragma unroll
        for (int u = 0; u < BI; ++u) {
          if (bon[u]) As[a00[u]] = o00[u];
          if (w01[u]) As[a01[u]] = o01[u];
          if (w10[u]) As[a10[u]] = o10[u];
          if (w11[u]) As[a11[u]] = o11[u];
        }
#pragma unroll
        for (int u = 0; u < VI; ++u) {
          if (von[u]) Vt[ap[u]] = op[u];   // (a pair with the dummy index: (c, s) = (1, 0), the row keeps its values)
          if (wq[u]) Vt[aq[u]] = oq[u];
        }
      }
      __syncthreads();
    }
    if (NT == 64) {
      if (!__any(rotated)) break;
      ever = true;
    } else {
      if (t < 64) {
        const int any = __any(rotated);
        if (t == 0) L.any = any;
      }
      __syncthreads();
      const int any = L.any;
      __syncthreads();
      if (!any) break;
      ever = true;
    }
  }
  if (t < n) L.ev[t] = As[t * LD + t];
  __syncthreads();
  if (t < n) {
    int rank = 0;
    const double v = L.ev[t];
    for (int j = 0; j < n; ++j) rank += (L.ev[j] > v || (L.ev[j] == v && j < t)) ? 1 : 0;
    L.perm[rank] = t;
  }
  __syncthreads();
  return ever;
}
