// Power probe (dev tool): runs ONE of three instruction mixes for a few seconds so that rocm-smi can be read beside it
// (tools/dev/gpu_power_probe.sh): 0 = fp64 MFMAs only (register loop, 16 accumulators per wave, 2 waves per SIMD),
// 1 = stores only (each wave 1 KB contiguous per instruction, 528 MB per pass), 2 = both in one instruction stream (a store
// behind every eighth MFMA).  Prints the rates it reached.
// build: hipcc -O3 --offload-arch=gfx950 tools/power_modes.hip -o tools/power_modes ; run: tools/power_modes MODE SECONDS
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
struct alignas(16) d2 { double x, y; };
__global__ __launch_bounds__(512) void k_mix(double* U, double* sink, int mode, int iters, long long per_wave_doubles) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * (lane + 3 * w);
  char* p = reinterpret_cast<char*>(U + (size_t(blockIdx.x) * 8 + w) * per_wave_doubles) + lane * 16;
  const long long nst = per_wave_doubles * 8 / 1024;  // store instructions that fit this wave's slab
  long long done = 0;
  for (int it = 0; it < iters; ++it) {
    if (mode != 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      a += 1e-12; b -= 1e-12;
    }
    if (mode != 0) {
      for (int x = 0; x < 2; ++x) {  // two stores per 16 MFMAs: 528 MB per 8.19 GFLOP-equivalent
        *reinterpret_cast<d2*>(p + (done % nst) * 1024) = d2{a + it, b};
        ++done;
      }
    }
  }
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  const int nwg = 512, iters = 2000;
  const long long per_wave = 16384;  // doubles: 128 KB per wave, 512 MB in all
  double *U, *sink;
  hipMalloc(&U, size_t(nwg) * 8 * per_wave * 8);
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_mix<<<nwg, 512>>>(U, sink, mode, 10, per_wave);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  double ms_sum = 0; int n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k_mix<<<nwg, 512>>>(U, sink, mode, iters, per_wave);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms_sum += ms; n += 10;
  }
  const double ms = ms_sum / n;
  const double flop = mode != 1 ? double(nwg) * 8 * iters * 16 * 2048.0 : 0.0, bytes = mode != 0 ? double(nwg) * 8 * iters * 2 * 1024.0 : 0.0;
  printf("mode %d: %.4f ms per launch, %.1f TFLOP/s fp64 MFMA, %.2f TB/s stores\n", mode, ms, flop / ms * 1e-9, bytes / ms * 1e-9);
  return 0;
}
