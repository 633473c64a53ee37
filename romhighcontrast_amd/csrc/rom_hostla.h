// Small dense linear algebra in long double for the setup tables of rom_fem_create (host only).
// Everything here runs once per FE space on matrices of at most (N-1) x a few (N-1); the tables are
// rounded to fp64 when they are uploaded.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <thread>
#include <vector>

namespace hostla {

typedef long double ld;

struct Mat {
  int r = 0, c = 0;
  std::vector<ld> v;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), v(size_t(r_) * c_, 0.0L) {}
  ld& operator()(int i, int j) { return v[size_t(i) * c + j]; }
  ld operator()(int i, int j) const { return v[size_t(i) * c + j]; }
  ld* row(int i) { return v.data() + size_t(i) * c; }
  const ld* row(int i) const { return v.data() + size_t(i) * c; }
};

inline Mat transpose(const Mat& A) {
  Mat B(A.c, A.r);
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < A.c; ++j) B(j, i) = A(i, j);
  return B;
}

// C = A * B
inline Mat mul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; ++i) {
    ld* ci = C.row(i);
    for (int k = 0; k < A.c; ++k) {
      const ld a = A(i, k);
      if (a == 0.0L) continue;
      const ld* bk = B.row(k);
      for (int j = 0; j < B.c; ++j) ci[j] += a * bk[j];
    }
  }
  return C;
}

// C = A * B^T
inline Mat mul_nt(const Mat& A, const Mat& B) {
  Mat C(A.r, B.r);
  for (int i = 0; i < A.r; ++i) {
    const ld* ai = A.row(i);
    for (int j = 0; j < B.r; ++j) {
      const ld* bj = B.row(j);
      ld s = 0;
      for (int k = 0; k < A.c; ++k) s += ai[k] * bj[k];
      C(i, j) = s;
    }
  }
  return C;
}

// C = A^T * B
inline Mat mul_tn(const Mat& A, const Mat& B) { return mul(transpose(A), B); }

// The same products with the rows of C spread over host threads, for the few n^3 products that stand alone on the
// critical path of rom_fem_create (x87 long double: ~0.3 GFLOP/s per core).  Every row is computed by the same
// sequential loop as above, so the result does not depend on the number of threads.
template <class F>
inline void rows_on_threads(int n, F fn) {
  const int T = std::max(1, std::min<int>(std::min(std::thread::hardware_concurrency(), 16u), n / 8));
  const int per = (n + T - 1) / T;
  std::vector<std::thread> pool;
  for (int t = 1; t < T; ++t) {
    const int i0 = t * per, i1 = std::min(n, i0 + per);
    if (i0 < i1) pool.emplace_back([i0, i1, &fn] { fn(i0, i1); });
  }
  fn(0, std::min(n, per));
  for (auto& th : pool) th.join();
}
inline Mat mul_par(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  rows_on_threads(A.r, [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      ld* ci = C.row(i);
      for (int k = 0; k < A.c; ++k) {
        const ld a = A(i, k);
        if (a == 0.0L) continue;
        const ld* bk = B.row(k);
        for (int j = 0; j < B.c; ++j) ci[j] += a * bk[j];
      }
    }
  });
  return C;
}
inline Mat mul_nt_par(const Mat& A, const Mat& B) {
  Mat C(A.r, B.r);
  rows_on_threads(A.r, [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      const ld* ai = A.row(i);
      for (int j = 0; j < B.r; ++j) {
        const ld* bj = B.row(j);
        ld s = 0;
        for (int k = 0; k < A.c; ++k) s += ai[k] * bj[k];
        C(i, j) = s;
      }
    }
  });
  return C;
}

inline std::vector<ld> matvec(const Mat& A, const std::vector<ld>& x) {
  std::vector<ld> y(A.r, 0.0L);
  for (int i = 0; i < A.r; ++i) {
    const ld* ai = A.row(i);
    ld s = 0;
    for (int k = 0; k < A.c; ++k) s += ai[k] * x[k];
    y[i] = s;
  }
  return y;
}

// inverse of a symmetric positive definite matrix (Cholesky); returns false if a pivot is not positive
inline bool spd_inverse(const Mat& A, Mat& inv) {
  const int n = A.r;
  Mat L(n, n);
  for (int j = 0; j < n; ++j) {
    ld d = A(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0L)) return false;
    const ld ljj = sqrtl(d);
    L(j, j) = ljj;
    for (int i = j + 1; i < n; ++i) {
      ld s = A(i, j);
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s / ljj;
    }
  }
  Mat Li(n, n);  // L^-1, lower triangular
  for (int c = 0; c < n; ++c)
    for (int i = c; i < n; ++i) {
      ld s = i == c ? 1.0L : 0.0L;
      for (int k = c; k < i; ++k) s -= L(i, k) * Li(k, c);
      Li(i, c) = s / L(i, i);
    }
  inv = Mat(n, n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      ld s = 0;
      for (int k = i; k < n; ++k) s += Li(k, i) * Li(k, j);
      inv(i, j) = inv(j, i) = s;
    }
  return true;
}

// Orthonormal basis W (n x rank) of the numerical column space of C (n x m): Householder QR with
// column pivoting, stopped when the largest remaining column norm drops below tol * (the first pivot).
// `pivots` (optional): the norm of the pivot column of every accepted step (the bases are nested: the first r columns
// of W span the r strongest directions).
inline Mat range_basis(const Mat& C0, ld tol, std::vector<ld>* pivots = nullptr) {
  // (worked on the transpose: the columns the reflections sweep are then contiguous -- same sums in the same order)
  Mat C = transpose(C0);
  const int n = C0.r, m = C0.c;
  const int kmax = std::min(n, m);
  std::vector<std::vector<ld>> refl;  // Householder vectors (unit normalised so that H = I - 2 v v^T)
  ld first = 0;
  int rank = 0;
  for (int k = 0; k < kmax; ++k) {
    int piv = -1;
    ld best = 0;
    for (int j = k; j < m; ++j) {
      const ld* cj = C.row(j);
      ld s = 0;
      for (int i = k; i < n; ++i) s += cj[i] * cj[i];
      if (piv < 0 || s > best) { piv = j; best = s; }
    }
    best = sqrtl(best);
    if (k == 0) first = best;
    if (!(best > tol * first) || best == 0.0L) break;
    if (pivots) pivots->push_back(best);
    if (piv != k) std::swap_ranges(C.row(k), C.row(k) + n, C.row(piv));
    std::vector<ld> v(n, 0.0L);
    const ld alpha = C(k, k) > 0 ? -best : best;
    ld vn = 0;
    for (int i = k; i < n; ++i) {
      v[i] = C(k, i) - (i == k ? alpha : 0.0L);
      vn += v[i] * v[i];
    }
    vn = sqrtl(vn);
    if (vn == 0.0L) break;
    for (int i = k; i < n; ++i) v[i] /= vn;
    for (int j = k; j < m; ++j) {
      ld* cj = C.row(j);
      ld s = 0;
      for (int i = k; i < n; ++i) s += v[i] * cj[i];
      s *= 2;
      for (int i = k; i < n; ++i) cj[i] -= s * v[i];
    }
    refl.push_back(v);
    ++rank;
  }
  // W = H_0 H_1 ... H_{rank-1} [I_rank; 0]   (built transposed as well)
  Mat Wt(rank, n);
  for (int j = 0; j < rank; ++j) Wt(j, j) = 1.0L;
  for (int k = rank - 1; k >= 0; --k) {
    const std::vector<ld>& v = refl[k];
    for (int j = 0; j < rank; ++j) {
      ld* wj = Wt.row(j);
      ld s = 0;
      for (int i = k; i < n; ++i) s += v[i] * wj[i];
      s *= 2;
      for (int i = k; i < n; ++i) wj[i] -= s * v[i];
    }
  }
  return transpose(Wt);
}

inline Mat identity(int n) {
  Mat I(n, n);
  for (int i = 0; i < n; ++i) I(i, i) = 1.0L;
  return I;
}

}  // namespace hostla
