// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on gfx950 (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d;
  hipMalloc(&d, 256 * 8 * 256 * 8 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  for (int wg_per_cu : {1, 2, 4}) {
    int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      k_mfma<4><<<grid, 256>>>(d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = double(grid) * 4 * iters * 4 * 2048.0;
      if (rep) printf("mfma_f64 16x16x4, 4 acc/wave, %d WG/CU: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, fl / ms * 1e-9, ms);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k_mfma<1><<<1024, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 1024.0 * 4 * iters * 1 * 2048.0;
    if (rep) printf("mfma_f64 16x16x4, 1 acc/wave (dependent chain), 4 WG/CU: %.2f TFLOP/s\n", fl / ms * 1e-9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k_fma<<<2048, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2048.0 * 256 * iters * 16 * 2.0;
    if (rep) printf("v_fma_f64 8 WG/CU: %.2f TFLOP/s\n", fl / ms * 1e-9);
  }
  return 0;
}
