// Write-bandwidth probe: how fast can 1024 snapshot rows of 255*255 doubles (528 MB) be written?
//  (a) plain contiguous fill, (b) the tile pattern of k_extend (64 rows x 4 x 16 doubles per workgroup)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(double* p, size_t n, double v) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// grid (patches = 32 x 8, system tiles, blocks 2x2); thread t: rows (t>>4)&3 + 4g..., like the MFMA D layout
__global__ __launch_bounds__(256) void k_tile(double* U, int M, int N, int nc, long long dim, double v) {
  const int n1 = N - 1, npj = (n1 + 15) / 16;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int pi = blockIdx.x / npj, pj = blockIdx.x % npj;
  const int i0 = 4 * pi, j0 = 16 * pj;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  for (int i = 0; i < 2; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + wr * 32 + i * 16 + (lane >> 4) + 4 * g;
      if (m >= M) continue;
      for (int jb = 0; jb < 2; ++jb) {
        const int c = wc * 32 + jb * 16 + (lane & 15);
        const int ii = i0 + (c >> 4), jj = j0 + (c & 15);
        if (ii >= n1 || jj >= n1) continue;
        U[(long long)m * dim + (long long)(p * N + ii) * nc + (q * N + jj)] = v;
      }
    }
}

// (g) as k_tile but the 64 vertices of a tile are one mesh row x 64 consecutive columns (MFMA D layout stores)
__global__ __launch_bounds__(256) void k_tile_row(double* U, int M, int N, int nc, long long dim, double v) {
  const int n1 = N - 1, npj = (n1 + 63) / 64;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int ii = blockIdx.x / npj, j0 = 64 * (blockIdx.x % npj);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  for (int i = 0; i < 2; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 64 + wr * 32 + i * 16 + (lane >> 4) + 4 * g;
      if (m >= M) continue;
      for (int jb = 0; jb < 2; ++jb) {
        const int jj = j0 + wc * 32 + jb * 16 + (lane & 15);
        if (jj >= n1) continue;
        U[(long long)m * dim + (long long)(p * N + ii) * nc + (q * N + jj)] = v;
      }
    }
}

typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
// (i) k_extend128's pattern: 128 systems x one mesh row (127 vertices); D layout, lane pairs own 2 adjacent vertices
__global__ __launch_bounds__(256) void k_tile128(double* U, int M, int N, int nc, long long dim, double v) {
  const int n1 = N - 1;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int iv = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  for (int i = 0; i < 4; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 128 + wr * 64 + i * 16 + kq + 4 * g;
      if (m >= M) continue;
      for (int hp = 0; hp < 2; ++hp) {
        const int jj = wc * 64 + (2 * hp + odd) * 16 + fr - odd;
        double* dst = U + (long long)m * dim + (long long)(p * N + iv) * nc + (q * N + jj);
        if (jj + 1 < n1) *reinterpret_cast<double2_u*>(dst) = double2_u{v, v};
        else if (jj < n1) dst[0] = v;
      }
    }
}
// (i') as (i) with non-temporal stores
__global__ __launch_bounds__(256) void k_tile128_nt(double* U, int M, int N, int nc, long long dim, double v) {
  const int n1 = N - 1;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int iv = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  for (int i = 0; i < 4; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 128 + wr * 64 + i * 16 + kq + 4 * g;
      if (m >= M) continue;
      for (int hp = 0; hp < 2; ++hp) {
        const int jj = wc * 64 + (2 * hp + odd) * 16 + fr - odd;
        double* dst = U + (long long)m * dim + (long long)(p * N + iv) * nc + (q * N + jj);
        if (jj + 1 < n1) __builtin_nontemporal_store(double2_u{v, v}, reinterpret_cast<double2_u*>(dst));
        else if (jj < n1) __builtin_nontemporal_store(v, dst);
      }
    }
}
// (j) same outputs, but every wave-instruction writes one system's whole 127-vertex run (16 bytes per lane)
__global__ __launch_bounds__(256) void k_rowrun128(double* U, int M, int N, int nc, long long dim, double v) {
  const int n1 = N - 1;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int iv = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = 0; r < 32; ++r) {
    const int m = blockIdx.y * 128 + w * 32 + r;
    if (m >= M) continue;
    const int jj = 2 * lane;
    double* dst = U + (long long)m * dim + (long long)(p * N + iv) * nc + (q * N + jj);
    if (jj + 1 < n1) *reinterpret_cast<double2_u*>(dst) = double2_u{v, v};
    else if (jj < n1) dst[0] = v;
  }
}

// (c) grid (mesh rows 127 x segs, system tiles, blocks): each wave writes `run` consecutive doubles of one system row
__global__ __launch_bounds__(256) void k_runs(double* U, int M, int N, int nc, long long dim, double v, int run) {
  const int n1 = N - 1, nseg = (n1 + run - 1) / run;
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int ii = blockIdx.x / nseg, seg = blockIdx.x % nseg;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = 0; r < 16; ++r) {
    const int m = blockIdx.y * 64 + w * 16 + r;
    if (m >= M) continue;
    for (int c = lane; c < run; c += 64) {
      const int jj = seg * run + c;
      if (jj >= n1) continue;
      U[(long long)m * dim + (long long)(p * N + ii) * nc + (q * N + jj)] = v;
    }
  }
}

// (e) grid (mesh-row groups of R, system tiles): each wave writes R full mesh rows (R * nc contiguous doubles)
// of one system at a time, 16 systems per wave
__global__ __launch_bounds__(256) void k_fullrows(double* U, int M, int nc, long long dim, double v, int R) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long start = (long long)blockIdx.x * R * nc;
  const long long len = min((long long)R * nc, dim - start);
  for (int r = 0; r < 16; ++r) {
    const int m = blockIdx.y * 64 + w * 16 + r;
    if (m >= M) continue;
    double* p = U + (long long)m * dim + start;
    for (long long c = lane; c < len; c += 64) p[c] = v;
  }
}

int main() {
  const int M = 1024, N = 128, nc = 2 * N - 1;
  const long long dim = (long long)nc * nc;
  const size_t n = size_t(M) * dim;
  double* U;
  CK(hipMalloc(&U, n * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CK(hipEventRecord(e0));
    k_fill<<<256 * 32, 256>>>(U, n, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("contiguous fill : %.3f ms  %.2f TB/s\n", ms, n * 8.0 / ms * 1e-9);
    CK(hipEventRecord(e0));
    k_tile<<<dim3(32 * 8, M / 64, 4), 256>>>(U, M, N, nc, dim, 2.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_extend pattern: %.3f ms  %.2f TB/s (of %.0f MB)\n", ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9, 4.0 * 127 * 127 * M * 8e-6);
    CK(hipEventRecord(e0));
    k_tile_row<<<dim3(127 * 2, M / 64, 4), 256>>>(U, M, N, nc, dim, 5.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1x64 tile, D layout: %.3f ms  %.2f TB/s\n", ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9);
    CK(hipEventRecord(e0));
    k_tile128<<<dim3(127, M / 128, 4), 256>>>(U, M, N, nc, dim, 6.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1x128 tile, D layout, 16 B/lane: %.3f ms  %.2f TB/s\n", ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9);
    CK(hipEventRecord(e0));
    k_tile128_nt<<<dim3(127, M / 128, 4), 256>>>(U, M, N, nc, dim, 6.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  ... non-temporal stores      : %.3f ms  %.2f TB/s\n", ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9);
    CK(hipEventRecord(e0));
    k_rowrun128<<<dim3(127, M / 128, 4), 256>>>(U, M, N, nc, dim, 7.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1x128 tile, wave per run, 16 B/lane: %.3f ms  %.2f TB/s\n", ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9);
    for (int R : {1, 2, 4, 8}) {
      CK(hipEventRecord(e0));
      k_fullrows<<<dim3((nc + R - 1) / R, M / 64), 256>>>(U, M, nc, dim, 4.0, R);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("full rows R=%d   : %.3f ms  %.2f TB/s\n", R, ms, n * 8.0 / ms * 1e-9);
    }
    for (int run : {64, 128}) {
      const int nseg = (127 + run - 1) / run;
      CK(hipEventRecord(e0));
      k_runs<<<dim3(127 * nseg, M / 64, 4), 256>>>(U, M, N, nc, dim, 3.0, run);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("runs of %3d     : %.3f ms  %.2f TB/s\n", run, ms, 4.0 * 127 * 127 * M * 8.0 / ms * 1e-9);
    }
  }
  return 0;
}
