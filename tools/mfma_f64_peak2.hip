// fp64 MFMA issue-rate probe: accumulators per wave x waves per SIMD, with the in-kernel clock.
// FLAWED (round 2): hipcc keeps the accumulators of this loop in AGPRs and copies all of them to VGPRs and back in
// every iteration (128 v_accvgpr moves per 8 MFMAs), so its 49 TFLOP/s is not the rate of the instruction.
// tools/mfma_store_overlap.hip (16 accumulators in VGPRs, clean loop) measures 71-73 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, unsigned long long* clk) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC>
void run(double* d, unsigned long long* dc, int wg_per_cu) {
  int iters = 20000 / NACC * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<NACC><<<256 * wg_per_cu, 256>>>(d, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  double fl = 256.0 * wg_per_cu * 4 * iters * NACC * 2048.0;
  double ghz = double(c[0]) / double(c[1]) * 0.1;  // memrealtime ticks at 100 MHz
  double tf = fl / ms * 1e-9;
  double interval = 2048.0 * 1024.0 * ghz * 1e9 / (tf * 1e12);  // chip rate -> issue interval per SIMD at that clock
  printf("acc/wave %2d, waves/SIMD %d: %6.2f TFLOP/s, in-kernel clock %.2f GHz -> one MFMA per %.0f cycles per SIMD\n", NACC,
         wg_per_cu, tf, ghz, interval);
}
int main() {
  double* d; unsigned long long* dc;
  hipMalloc(&d, 256 * 8 * 256 * 8); hipMalloc(&dc, 16);
  for (int w : {1, 2, 4, 8}) { run<1>(d, dc, w); run<4>(d, dc, w); run<8>(d, dc, w); }
  return 0;
}
