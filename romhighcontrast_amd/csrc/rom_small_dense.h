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
  double As[32 * J32_LD], Vt[32 * J32_LD];   // the matrix in / the eigenvector rows out (buffer 0 of the rounds)
  double A1[32 * J32_LD], V1[32 * J32_LD];   // buffer 1
  double nu2[2][32], ev[32];
  int perm[32];
  int any;
};

// One Jacobi rotation from the 2 x 2 pivot block (a_pp, a_pq; a_pq, a_qq) and the noise levels of its rows: applied when
// |a_pq| > tol sqrt(|a_pp a_qq|) -- the relative criterion under which graded matrices keep their small eigenvalues
// (Demmel-Veselic) -- AND |a_pq| > tol nu_p nu_q, the rounding noise of the entry.  (c, s) = (1, 0) otherwise: an exact
// identity, so callers apply it without a branch.  No division and no square root: with a = a_qq - a_pp, b = 2 a_pq
// (scaled by a power of two so that a^2 + b^2 neither overflows nor underflows), r = 1 / sqrt(a^2 + b^2):
//   cos 2phi = |a| r,  c = cos phi = sqrt((1 + cos 2phi) / 2),  s = sin phi = sign(a) b r / (2 c)
// -- the angle of the classical formula t = sign(a) b / (|a| + sqrt(a^2 + b^2)), |phi| <= pi / 4, with s from b itself (no
// cancellation for small angles) and c^2 + s^2 = 1 to rounding.  A pure function of its arguments: every thread that
// evaluates the rotation of a pair gets the same bits.
__device__ inline bool jacobi_rotation(double app, double aqq, double apq, double np2, double nq2, double tol2, double floor_abs,
                                       double& c, double& s, double& np2o, double& nq2o) {
  c = 1.0;
  s = 0.0;
  np2o = np2;
  nq2o = nq2;
  const double apq2 = apq * apq;
  if (!(apq2 > tol2 * fabs(app * aqq) && fabs(apq) > floor_abs && apq2 > tol2 * np2 * nq2)) return false;
  double a = aqq - app, b = 2.0 * apq;
  int e;
  (void)frexp(fmax(fabs(a), fabs(b)), &e);
  a = ldexp(a, -e);
  b = ldexp(b, -e);
  const double r = se_rsqrt(a * a + b * b);
  const double c2 = 0.5 + 0.5 * (fabs(a) * r);
  const double rc = se_rsqrt(c2);
  c = c2 * rc;
  s = (a >= 0.0 ? b : -b) * r * (0.5 * rc);
  np2o = c * c * np2 + s * s * nq2;
  nq2o = s * s * np2 + c * c * nq2;
  return true;
}

// Cyclic Jacobi for a symmetric matrix of order n <= 32 that sits in L.As (both triangles; L.Vt receives the eigenvector
// rows), 256 threads, ONE barrier per round.  The round-robin pairs of a round are at most 16; thread (k, l) owns the
// 2 x 2 block rows (p_k, q_k) x columns (p_l, q_l) of the matrix and the entries of the eigenvector rows (p_k, q_k) in
// columns 2 l, 2 l + 1.  It computes the rotations of its two pairs ITSELF from the pivot blocks (jacobi_rotation: the
// same bits in every thread) -- no lane computes rotations for the others, so there is no second phase behind a barrier and
// no dependent LDS round trip: all loads of a round are issued at once, from the buffer the previous round wrote, and the
// results go to the other buffer (every entry is written every round: the pairs cover all indices; orders below 32 / odd
// orders are padded with zero rows and columns, which never rotate).  gram_like: nu_i^2 = |a_ii| at the start (the
// rounding noise of a Gram matrix of explicit rows), rotated along; else max |a_ii|.  On return: eigenvalues on the
// diagonal of L.As, eigenvector rows in L.Vt, descending order in L.perm (L.ev[L.perm[i]] is the i-th largest), every
// thread past a barrier.  dmax = max |a_ii| (the caller has it from the load).  Returns (uniformly) whether any rotation
// was applied: false = the matrix was diagonal by the criterion on entry.
template <int NT>
__device__ inline bool jacobi32_run(int n, Jacobi32Lds& L, int gram_like, double dmax) {
  static_assert(NT == 256, "jacobi32_run: 256 threads");
  constexpr int LD = J32_LD;
  const int t = threadIdx.x, ne = n + (n & 1), half = ne / 2, nm1 = ne - 1;
  for (int idx = t; idx < 32 * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    if (r >= n || c >= n) {
      L.As[r * LD + c] = 0.0;
      L.Vt[r * LD + c] = r == c ? 1.0 : 0.0;
    }
  }
  if (t < 32) L.nu2[0][t] = t < n ? (gram_like ? fabs(L.As[t * LD + t]) : dmax) : 0.0;
  const bool on = t < half * half;
  const int bk = on ? t / half : 0, bl = on ? t - bk * half : 0;
  const double tol = double(n > 8 ? n : 8) * 1.1e-16, tol2 = tol * tol, floor_abs = fmax(1e-300, 1e-40 * dmax);
  __syncthreads();
  int cur = 0;
  bool ever = false;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int r = 0; r < nm1; ++r) {
      const double* A = cur ? L.A1 : L.As;
      const double* V = cur ? L.V1 : L.Vt;
      double* An = cur ? L.As : L.A1;
      double* Vn = cur ? L.Vt : L.V1;
      if (on) {
        // the pairs of this round (round robin: index ne - 1 stays, the others move around it)
        int pk = r, qk = nm1, pl = r, ql = nm1;
        if (bk) {
          pk = r + bk;
          if (pk >= nm1) pk -= nm1;
          qk = r - bk;
          if (qk < 0) qk += nm1;
          if (pk > qk) { const int x = pk; pk = qk; qk = x; }
        }
        if (bl) {
          pl = r + bl;
          if (pl >= nm1) pl -= nm1;
          ql = r - bl;
          if (ql < 0) ql += nm1;
          if (pl > ql) { const int x = pl; pl = ql; ql = x; }
        }
        const double kpp = A[pk * LD + pk], kqq = A[qk * LD + qk], kpq = A[pk * LD + qk];
        const double lpp = A[pl * LD + pl], lqq = A[ql * LD + ql], lpq = A[pl * LD + ql];
        const double nkp = L.nu2[cur][pk], nkq = L.nu2[cur][qk], nlp = L.nu2[cur][pl], nlq = L.nu2[cur][ql];
        const double b00 = A[pk * LD + pl], b01 = A[pk * LD + ql], b10 = A[qk * LD + pl], b11 = A[qk * LD + ql];
        const int j0 = 2 * bl;
        const double vp0 = V[pk * LD + j0], vp1 = V[pk * LD + j0 + 1], vq0 = V[qk * LD + j0], vq1 = V[qk * LD + j0 + 1];
        double ck, sk, cl, sl, nkpo, nkqo, nlpo, nlqo;
        const bool rk = jacobi_rotation(kpp, kqq, kpq, nkp, nkq, tol2, floor_abs, ck, sk, nkpo, nkqo);
        (void)jacobi_rotation(lpp, lqq, lpq, nlp, nlq, tol2, floor_abs, cl, sl, nlpo, nlqo);
        rotated = rotated || rk;
        // A <- J^T A J on the block: R_k B R_l^T
        const double r00 = ck * b00 - sk * b10, r01 = ck * b01 - sk * b11;
        const double r10 = sk * b00 + ck * b10, r11 = sk * b01 + ck * b11;
        An[pk * LD + pl] = cl * r00 - sl * r01;
        An[pk * LD + ql] = sl * r00 + cl * r01;
        An[qk * LD + pl] = cl * r10 - sl * r11;
        An[qk * LD + ql] = sl * r10 + cl * r11;
        // eigenvector rows: Vt <- J^T Vt
        Vn[pk * LD + j0] = ck * vp0 - sk * vq0;
        Vn[pk * LD + j0 + 1] = ck * vp1 - sk * vq1;
        Vn[qk * LD + j0] = sk * vp0 + ck * vq0;
        Vn[qk * LD + j0 + 1] = sk * vp1 + ck * vq1;
        if (bl == 0) {
          L.nu2[cur ^ 1][pk] = nkpo;
          L.nu2[cur ^ 1][qk] = nkqo;
        }
      }
      cur ^= 1;
      __syncthreads();
    }
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
  if (cur) {   // an odd number of rounds: the results sit in buffer 1
    for (int idx = t; idx < 32 * 32; idx += 256) {
      const int r = idx >> 5, c = idx & 31;
      L.As[r * LD + c] = L.A1[r * LD + c];
      L.Vt[r * LD + c] = L.V1[r * LD + c];
    }
    __syncthreads();
  }
  if (t < n) L.ev[t] = L.As[t * LD + t];
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
