// Microbenchmark: the extension product U[m][v] = sum_k Gt[k][v] * y[m][k] on the VECTOR fp64 pipe
// (v_fma_f64 with the y operand in SGPRs through the scalar cache) instead of v_mfma_f64_16x16x4_f64:
// tools/mfma_f64_peak* measure 67 TFLOP/s for v_fma_f64 against 49 TFLOP/s for the fp64 MFMA on gfx950 (round 2: that 49 was an artefact of the probe, the MFMA sustains 71-73).
// Lanes own VPL adjacent vertices, J systems per wave in registers; no LDS, no barriers.
// YT: y stored k-major (yT[k][m]) so that one s_load_dwordx16 fetches 8 systems of one k.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_gemm.hip -o tools/valu_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VPL, int J, int WAVES, bool YT>
__global__ __launch_bounds__(64 * WAVES) void k_valu(const double* __restrict__ Gt, const double* __restrict__ y,
                                                      double* __restrict__ U, int V, int Vp, int K, int ldy, int M,
                                                      int do_store) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int v = blockIdx.x * (64 * VPL) + VPL * lane;
  const int m0 = (blockIdx.y * WAVES + w) * J;
  double acc[J][VPL];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int p = 0; p < VPL; ++p) acc[j][p] = 0.0;
  const double* gp = Gt + v;
  constexpr int KS = YT ? 2 : 4;
  for (int k0 = 0; k0 < K; k0 += KS) {
    double a[KS][VPL];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int p = 0; p < VPL; p += 2) {
        const double2 t = *reinterpret_cast<const double2*>(gp + size_t(k0 + kk) * Vp + p);
        a[kk][p] = t.x;
        a[kk][p + 1] = t.y;
      }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double b = YT ? y[size_t(k0 + kk) * M + m0 + j] : y[size_t(m0 + j) * ldy + k0 + kk];  // wave-uniform -> s_load
#pragma unroll
        for (int p = 0; p < VPL; ++p) acc[j][p] = fma(a[kk][p], b, acc[j][p]);
      }
  }
  if (do_store) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int m = m0 + j;
#pragma unroll
      for (int p = 0; p < VPL; p += 2)
        if (m < M && v + p + 1 < V) *reinterpret_cast<double2*>(U + size_t(m) * V + v + p) = double2{acc[j][p], acc[j][p + 1]};
    }
  } else if (acc[0][0] == 123.456) {
    U[0] = acc[J - 1][VPL - 1];
  }
}

// software-pipelined variant (k-major y only): operands of step k+1 are requested before the FMAs of step k
template <int VPL, int J, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_valu_pipe(const double* __restrict__ Gt, const double* __restrict__ y,
                                                           double* __restrict__ U, int V, int Vp, int K, int ldy, int M,
                                                           int do_store) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int v = blockIdx.x * (64 * VPL) + VPL * lane;
  const int m0 = (blockIdx.y * WAVES + w) * J;
  double acc[J][VPL];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int p = 0; p < VPL; ++p) acc[j][p] = 0.0;
  const double* gp = Gt + v;
  const double* yp = y + m0;
  double a0[VPL], a1[VPL], b0[J], b1[J];
  auto load = [&](int k, double* a, double* b) {
#pragma unroll
    for (int p = 0; p < VPL; p += 2) {
      const double2 t = *reinterpret_cast<const double2*>(gp + size_t(k) * Vp + p);
      a[p] = t.x;
      a[p + 1] = t.y;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) b[j] = yp[size_t(k) * M + j];
  };
  auto mac = [&](const double* a, const double* b) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int p = 0; p < VPL; ++p) acc[j][p] = fma(a[p], b[j], acc[j][p]);
  };
  load(0, a0, b0);
  for (int k0 = 0; k0 < K; k0 += 2) {
    load(k0 + 1, a1, b1);
    mac(a0, b0);
    load(k0 + 2 < K ? k0 + 2 : k0, a0, b0);
    mac(a1, b1);
  }
  if (do_store) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int m = m0 + j;
#pragma unroll
      for (int p = 0; p < VPL; p += 2)
        if (m < M && v + p + 1 < V) *reinterpret_cast<double2*>(U + size_t(m) * V + v + p) = double2{acc[j][p], acc[j][p + 1]};
    }
  } else if (acc[0][0] == 123.456) {
    U[0] = acc[J - 1][VPL - 1];
  }
}

template <int VPL, int J, int WAVES>
void run_pipe(const double* Gt, const double* yT, double* U, int V, int K, int ldy, int M) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int Vp = (V + 255) / 256 * 256;
  dim3 grid(Vp / (64 * VPL), M / (J * WAVES));
  for (int st = 1; st >= 0; --st) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(e0);
      k_valu_pipe<VPL, J, WAVES><<<grid, 64 * WAVES>>>(Gt, yT, U, V, Vp, K, ldy, M, st);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) best = ms < best ? ms : best;
    }
    printf("pipelined VPL=%d J=%2d waves/WG=%d K=%d stores=%d: %.3f ms -> %.1f TFLOP/s, %.2f TB/s written\n", VPL, J, WAVES, K,
           st, best, 2.0 * V * K * M / best * 1e-9, st ? 8.0 * V * M / best * 1e-9 : 0.0);
  }
}

template <int VPL, int J, int WAVES, bool YT>
void run(const double* Gt, const double* y, const double* yT, double* U, int V, int K, int ldy, int M) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int Vp = (V + 255) / 256 * 256;
  dim3 grid(Vp / (64 * VPL), M / (J * WAVES));
  for (int st = 1; st >= 0; --st) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(e0);
      k_valu<VPL, J, WAVES, YT><<<grid, 64 * WAVES>>>(Gt, YT ? yT : y, U, V, Vp, K, ldy, M, st);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) best = ms < best ? ms : best;
    }
    printf("VPL=%d J=%2d waves/WG=%d yT=%d K=%d stores=%d: %.3f ms -> %.1f TFLOP/s, %.2f TB/s written\n", VPL, J, WAVES, int(YT), K,
           st, best, 2.0 * V * K * M / best * 1e-9, st ? 8.0 * V * M / best * 1e-9 : 0.0);
  }
}

int main(int argc, char** argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 64516, K = argc > 2 ? atoi(argv[2]) : 64, M = 1024, ldy = 784;
  const int Vp = (V + 255) / 256 * 256;
  std::vector<double> hG(size_t(K) * Vp), hy(size_t(M) * ldy), hyT(size_t(K) * M);
  srand(1);
  for (auto& x : hG) x = rand() / double(RAND_MAX) - 0.5;
  for (auto& x : hy) x = rand() / double(RAND_MAX) - 0.5;
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) hyT[size_t(k) * M + m] = hy[size_t(m) * ldy + k];
  double *Gt, *y, *yT, *U;
  (void)hipMalloc(&Gt, hG.size() * 8);
  (void)hipMalloc(&y, hy.size() * 8);
  (void)hipMalloc(&yT, hyT.size() * 8);
  (void)hipMalloc(&U, size_t(M) * V * 8 + 4096);
  (void)hipMemcpy(Gt, hG.data(), hG.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(yT, hyT.data(), hyT.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> hU(size_t(M) * V);
  auto check = [&]() {
    (void)hipMemcpy(hU.data(), U, hU.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 2000; ++t) {
      int m = rand() % M, v = rand() % (V - 1);
      double s = 0;
      for (int k = 0; k < K; ++k) s = fma(hG[size_t(k) * Vp + v], hy[size_t(m) * ldy + k], s);
      double e = fabs(s - hU[size_t(m) * V + v]);
      worst = e > worst ? e : worst;
    }
    printf("  max abs diff vs host on 2000 samples: %.3e\n", worst);
    (void)hipMemset(U, 0, size_t(M) * V * 8);
  };
  run_pipe<2, 16, 4>(Gt, yT, U, V, K, ldy, M); check();
  run_pipe<2, 24, 4>(Gt, yT, U, V, K, ldy, 960); 
  run_pipe<4, 16, 4>(Gt, yT, U, V, K, ldy, M); check();
  run_pipe<4, 8, 4>(Gt, yT, U, V, K, ldy, M); check();
  run_pipe<2, 16, 2>(Gt, yT, U, V, K, ldy, M);
  run<2, 16, 4, false>(Gt, y, yT, U, V, K, ldy, M); check();
  run<2, 16, 4, true>(Gt, y, yT, U, V, K, ldy, M); check();
  run<2, 32, 4, true>(Gt, y, yT, U, V, K, ldy, M); check();
  run<4, 16, 4, true>(Gt, y, yT, U, V, K, ldy, M); check();
  run<4, 8, 4, true>(Gt, y, yT, U, V, K, ldy, M); check();
  run<2, 16, 2, true>(Gt, y, yT, U, V, K, ldy, M);
  run<2, 16, 1, true>(Gt, y, yT, U, V, K, ldy, M);
  run<2, 16, 4, true>(Gt, y, yT, U, V, 4, ldy, M);   // (almost) store only
  run<4, 8, 4, true>(Gt, y, yT, U, V, 4, ldy, M);
  return 0;
}
