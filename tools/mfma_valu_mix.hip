// Microbenchmark: do the fp64 MFMA pipe and the fp64 vector pipe of gfx950 run concurrently?
// Each wave loops over {NM independent v_mfma_f64_16x16x4_f64, NV independent v_fma_f64}; no memory traffic.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_mix.hip -o tools/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters) {
  d4 acc[NM > 0 ? NM : 1];
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = d4{0, 0, 0, 0};
  double x[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  const double c = 1.0000001, d = 1e-9;
  for (int it = 0; it < iters; ++it) {
    // interleave: after every MFMA a share of the vector FMAs
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
      if (NM > 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j % (NM > 0 ? NM : 1) == i) x[j] = fma(x[j], c, d);
    }
  }
  double s = 0;
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV>
void run(double* d, int wg_per_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * wg_per_cu;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_mix<NM, NV><<<grid, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep) best = ms < best ? ms : best;
  }
  const double waves = double(grid) * 4;
  const double fm = waves * iters * NM * 2048.0, fv = waves * iters * NV * 128.0;
  printf("MFMA/iter %2d  FMA/iter %3d  waves/SIMD %d: %7.3f ms  MFMA %5.1f + vector %5.1f = %5.1f TFLOP/s\n", NM, NV, wg_per_cu, best,
         fm / best * 1e-9, fv / best * 1e-9, (fm + fv) / best * 1e-9);
}

int main() {
  double* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
  for (int w : {2, 4}) {
    run<8, 0>(d, w);
    run<0, 32>(d, w);
    run<8, 32>(d, w);
    run<8, 64>(d, w);
    run<8, 128>(d, w);
    run<8, 192>(d, w);
  }
  return 0;
}
