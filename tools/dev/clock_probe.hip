// Does a one-workgroup kernel run at the clock a full grid runs at?  A chain of N dependent fp64 FMAs per wave, timed with
// events, for grids of 1 ... 4096 workgroups of one wave; and the same with an LDS round trip + barrier per step (the shape
// of the small dense kernels).  build: hipcc -O3 --offload-arch=gfx950 tools/dev/clock_probe.hip -o tools/dev/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(double* out, int n) {
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9;
  for (int i = 0; i < n; ++i) a = __builtin_fma(a, 1.0000001, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_lds_steps(double* out, int n) {
  __shared__ double s[256 * 2];
  const int t = threadIdx.x;
  s[t] = t;
  __syncthreads();
  double a = 0.0;
  int cur = 0;
  for (int i = 0; i < n; ++i) {
    a = s[cur * 256 + ((t + 1) & 255)] * 1.0000001 + a;
    s[(cur ^ 1) * 256 + t] = a;
    cur ^= 1;
    __syncthreads();
  }
  out[blockIdx.x * 256 + t] = a;
}
int main() {
  double* out;
  hipMalloc(&out, 4096 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 1 << 20;
  for (int rep = 0; rep < 2; ++rep)
    for (int g : {1, 4, 256, 4096}) {
      hipEventRecord(e0);
      k_chain<<<g, 64>>>(out, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("chain   grid %5d: %8.3f ms  -> %6.2f ns per dependent FMA\n", g, ms, ms * 1e6 / n);
    }
  const int m = 1 << 16;
  for (int rep = 0; rep < 2; ++rep)
    for (int g : {1, 256, 2048}) {
      hipEventRecord(e0);
      k_lds_steps<<<g, 256>>>(out, m);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("lds+bar grid %5d: %8.3f ms  -> %6.1f ns per step (LDS read, fma, LDS write, barrier)\n", g, ms, ms * 1e6 / m);
    }
  return 0;
}
