// Standalone timing of the wave-level 64x64 Cholesky / inverse kernels (dev tool): where does the time go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
constexpr int LDC = 66;
__device__ inline double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline double rsqrt_newton(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * fma(-hd * y, y, 1.5);
  y = y * fma(-hd * y, y, 1.5);
  return y;
}
template <int MODE>  // 0: load+store only, 1: full elimination, 2: elimination without LDS broadcast wait
__global__ __launch_bounds__(64) void k_potrf(double* L) {
  __shared__ __align__(16) double Ls[64 * LDC];
  __shared__ __align__(16) double lv[2][64];
  const int m = blockIdx.x, lane = threadIdx.x;
  double* Lt = L + size_t(m) * 4096;
  for (int i = 0; i < 64; ++i) Ls[i * LDC + lane] = Lt[i * 64 + lane];
  __syncthreads();
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; c += 2) {
    double2 v = *reinterpret_cast<const double2*>(&Ls[lane * LDC + c]);
    a[c] = v.x; a[c + 1] = v.y;
  }
  if (MODE >= 1) {
#pragma unroll
    for (int jj = 0; jj < 64; ++jj) {
      const double dj = readlane_f64(a[jj], jj);
      const double rs = rsqrt_newton(dj);
      const double l = a[jj] * rs;
      a[jj] = l;
      if (jj < 63) {
        double* bv = lv[jj & 1];
        bv[lane] = l;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 64; ++c)
          if (c > jj) a[c] -= l * bv[c];
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 64; ++c) Ls[lane * LDC + c] = c <= lane ? a[c] : 0.0;
  __syncthreads();
  for (int i = 0; i < 64; ++i) Lt[i * 64 + lane] = Ls[i * LDC + lane];
}
// 4-column blocked variant: one LDS round trip per 4 columns
__global__ __launch_bounds__(64) void k_potrf_b4(double* L) {
  __shared__ __align__(16) double Ls[64 * LDC];
  __shared__ __align__(16) double lv[2][4][64];
  const int m = blockIdx.x, lane = threadIdx.x;
  double* Lt = L + size_t(m) * 4096;
  for (int i = 0; i < 64; ++i) Ls[i * LDC + lane] = Lt[i * 64 + lane];
  __syncthreads();
  double a[64];
#pragma unroll
  for (int c = 0; c < 64; c += 2) {
    double2 v = *reinterpret_cast<const double2*>(&Ls[lane * LDC + c]);
    a[c] = v.x; a[c + 1] = v.y;
  }
#pragma unroll
  for (int jb = 0; jb < 64; jb += 4) {
    // factor the 4 columns of the panel (each lane on its own row; the 4x4 diagonal block via readlane)
    double l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = a[jb + k];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < k) v -= l[q] * readlane_f64(l[q], jb + k);  // L[lane][jb+q] * L[jb+k][jb+q]
      const double dj = readlane_f64(v, jb + k);
      l[k] = v * rsqrt_newton(dj);
      a[jb + k] = l[k];
    }
    if (jb < 60) {
      double (*bv)[64] = lv[(jb >> 2) & 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[k][lane] = l[k];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 64; ++c)
        if (c >= jb + 4) a[c] -= l[0] * bv[0][c] + l[1] * bv[1][c] + l[2] * bv[2][c] + l[3] * bv[3][c];
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 64; ++c) Ls[lane * LDC + c] = c <= lane ? a[c] : 0.0;
  __syncthreads();
  for (int i = 0; i < 64; ++i) Lt[i * 64 + lane] = Ls[i * LDC + lane];
}
int main() {
  const int M = 1024;
  std::vector<double> h(size_t(M) * 4096);
  for (int m = 0; m < M; ++m)
    for (int r = 0; r < 64; ++r)
      for (int c = 0; c < 64; ++c) h[size_t(m) * 4096 + r * 64 + c] = (r == c ? 70.0 : 0.0) + 1.0 / (1 + abs(r - c)) + 1e-3 * m;
  double *d, *d0;
  hipMalloc(&d, h.size() * 8); hipMalloc(&d0, h.size() * 8);
  hipMemcpy(d0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemcpy(d, d0, h.size() * 8, hipMemcpyDeviceToDevice);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<double> out(4096);
    hipMemcpy(out.data(), d + 4096 * 7, 4096 * 8, hipMemcpyDeviceToHost);
    printf("%-28s %.1f us   L[7][63][63]=%.12f L[7][10][3]=%.12f\n", name, best * 1e3, out[63 * 64 + 63], out[10 * 64 + 3]);
  };
  run("copy only", [&] { k_potrf<0><<<M, 64>>>(d); });
  run("potrf (current)", [&] { k_potrf<1><<<M, 64>>>(d); });
  run("potrf blocked x4", [&] { k_potrf_b4<<<M, 64>>>(d); });
  return 0;
}
