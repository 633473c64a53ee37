// Do fp64 MFMA work and a store stream overlap on this part, and at which level do they couple?  (dev probe, round 2)
// The extension kernel multiplies 8.5 GFLOP (padded) and writes 528 MB per C2 step; every structure tried so far ends
// at 0.23-0.25 ms, about the SUM of its MFMA time (0.12 ms at the 72 TFLOP/s measured here) and its store time.  This probe runs the two ingredients without any data dependence between them:
//   mfma      every CU: register-only v_mfma_f64_16x16x4_f64 loop (16 independent accumulators)
//   store     every CU: 16-byte-per-lane stores, a wave instruction writes 1 KB contiguous, consecutive instructions
//             consecutive KBs
//   same      every wave does both, 2 stores per 16 MFMAs
//   waves     8 waves per CU: waves 0-3 multiply, waves 4-7 store (one of each per SIMD)
//   cus       CUs with even CU_ID multiply twice their share, CUs with odd CU_ID store twice their share
//   same x2   8 waves per CU (two per SIMD), each half of `same`: 16 MFMAs, 2 stores (x2': 8 MFMAs, 1 store)
//   x2' + LDS  same x2' with the MFMA operands read from LDS every k-step; `LDS only`: without the stores; `exec`: the
//             store issued with an exec mask from SGPRs and an empty twin, as k_extend_p does
//   x2' + loads  same x2' with a 16-byte-per-lane global load (2 MB window, L2) per 8 MFMAs, used 8 k-steps later;
//             `loads only`: without the stores; `+ DMA`: the loads as global_load_lds_dwordx4
//   full A-D  x2' + DMA stepwise towards k_extend_p: 2 DMA loads per 8 MFMAs; + operands read from the DMA's LDS region;
//             + s_barrier every 32 MFMAs; + s_waitcnt vmcnt(6) before it; E: every DMA load gathers 16 rows x 64 B (27 KB apart);
//             F: ... out of a 32 MB table
//   mfma/2    only the even CUs multiply (twice their share), the odd ones exit      } the two halves of `cus`
//   store/2   only the odd CUs store (twice their share), the even ones exit         } run alone
// One workgroup per CU (grid 256, 128 KB of LDS requested so that no two share a CU).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_store_overlap.hip -o tools/mfma_store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

enum { MFMA = 0, STORE, SAME, WAVES, CUS, MFMA_HALF, STORE_HALF, SAME2, SAME2X, SAME2L, SAME2LN, SAME2E, SAME2G, SAME2GN, SAME2D, STORE_8TH, FULL_A, FULL_B, FULL_C, FULL_D, FULL_E, FULL_F, WAVES_PS, WAVES_PM, WAVES_SW, WAVES_SWP, WAVES_NV, WAVES_NVP };

__device__ int g_data;  // 0: operands within 1e-6 of 1 (few mantissa bits toggle), 1: full mantissas
__device__ inline double operand(int which) {
  const unsigned h = (threadIdx.x * 2654435761u + which * 40503u + blockIdx.x * 97u) ^ 0x9e3779b9u;
  return g_data ? 0.5 + double(h) * (1.0 / 4294967296.0) * 0.999 : (which ? 1.0 - threadIdx.x * 1e-9 : 1.0 + threadIdx.x * 1e-9);
}
__device__ inline void mfma_loop(int iters, double* sink) {
  const double a = operand(0), b = operand(1);
  d4_t acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  if (threadIdx.x == 0 && blockIdx.x == 7) sink[1] = double(__builtin_readcyclecounter() - t0) / (16.0 * iters);  // counter ticks per MFMA
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.678) sink[0] = s;
}

// `n` wave-instructions of 1 KB, region of this wave: base + wave_slot * n KB
__device__ inline void store_loop(double* base, long long wave_slot, int n) {
  double* p = base + wave_slot * (long long)n * 128 + (threadIdx.x & 63) * 2;
  for (int i = 0; i < n; ++i) *reinterpret_cast<double2_u*>(p + (long long)i * 128) = double2_u{1.0 * i, 2.0};
}

__global__ __launch_bounds__(512) void k_probe(double* U, double* sink, unsigned* cnt, int mode, int iters, int nstore) {
  extern __shared__ double lds[];
  if (threadIdx.x == 9999) lds[0] = 0.0;
  const int w = threadIdx.x >> 6;
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const int cu_odd = (hwid >> 8) & 1;
  if (threadIdx.x == 0 && cnt) atomicAdd(cnt + cu_odd, 1u);
  const long long slot4 = (long long)blockIdx.x * 4 + (w & 3);
  switch (mode) {
    case MFMA:
      if (w < 4) mfma_loop(iters, sink);
      break;
    case STORE:
      if (w < 4) store_loop(U, slot4, nstore);
      break;
    case SAME:
      if (w < 4) {
        const double a = operand(0), b = operand(1);
        d4_t acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
        double* p = U + slot4 * (long long)nstore * 128 + (threadIdx.x & 63) * 2;
        const int per = (nstore + iters - 1) / iters;
        int done = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
          for (int x = 0; x < per && done < nstore; ++x, ++done)
            *reinterpret_cast<double2_u*>(p + (long long)done * 128) = double2_u{1.0 * it, 2.0};
        }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (s == 12345.678) sink[0] = s;
      }
      break;
    case SAME2:   // 8 waves (two per SIMD), each half the iterations of `same`
    case SAME2X: {  // ... with one store per 8 MFMAs instead of two per 16
      const double a = operand(0), b = operand(1);
      d4_t acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
      const int it2 = iters / 2, ns2 = nstore / 2;
      double* p = U + ((long long)blockIdx.x * 8 + w) * (long long)ns2 * 128 + (threadIdx.x & 63) * 2;
      int done = 0;
      for (int it = 0; it < it2; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        if (mode == SAME2X && done < ns2) { *reinterpret_cast<double2_u*>(p + (long long)done * 128) = double2_u{1.0 * it, 2.0}; ++done; }
#pragma unroll
        for (int j = 8; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        for (int x = 0; x < (mode == SAME2X ? 1 : 2) && done < ns2; ++x, ++done)
          *reinterpret_cast<double2_u*>(p + (long long)done * 128) = double2_u{1.0 * it, 2.0};
      }
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      if (s == 12345.678) sink[0] = s;
      break;
    }
    case SAME2L:    // `same x2'` with the operands of every MFMA k-step read from LDS (6 ds_read_b64 per 8 MFMAs)
    case SAME2LN:   // ... the same without the stores
    case SAME2E: {  // `same x2'` with the store issued the way k_extend_p does (exec mask from an SGPR pair + an empty twin)
      d4_t acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
      for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = operand(i & 1);
      __syncthreads();
      const int it2 = iters, ns2 = nstore / 2;  // (8 MFMAs per iteration: `iters` iterations = half of `same`)
      double* p = U + ((long long)blockIdx.x * 8 + w) * (long long)ns2 * 128 + (threadIdx.x & 63) * 2;
      const double* la = lds + (threadIdx.x & 63) * 9 + w * 1024;
      int done = 0;
      double a[4] = {operand(0), operand(1), operand(0), operand(1)}, b[2] = {operand(1), operand(0)};
      for (int it = 0; it < it2; ++it) {
        if (mode != SAME2E) {
          const double* q = la + (it & 7) * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = q[i * 640];
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = q[4096 + j * 640];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
        if (mode == SAME2L && (it & 1) && done < ns2) { *reinterpret_cast<double2_u*>(p + (long long)done * 128) = double2_u{1.0 * it, 2.0}; ++done; }
        if (mode == SAME2E && (it & 1) && done < ns2) {
          const unsigned long long m16 = __builtin_amdgcn_ballot_w64(threadIdx.x < 100000u), m8 = 0ull;
          unsigned long long sv;
          const double2_u pr = double2_u{1.0 * it, 2.0};
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %3, %4, off\n\t"
                       "s_mov_b64 exec, %2\n\tglobal_store_dwordx2 %3, %5, off\n\ts_mov_b64 exec, %0"
                       : "=&s"(sv) : "s"(m16), "s"(m8), "v"(p + (long long)done * 128), "v"(pr), "v"(pr.x));
          ++done;
        }
      }
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      if (s == 12345.678) sink[0] = s;
      break;
    }
    case SAME2G:    // `same x2'` + one 16-byte-per-lane global load (L2 resident) per 8 MFMAs, used 8 iterations later
    case SAME2GN:   // ... without the stores
    case SAME2D: {  // `same x2'` + one LDS-DMA load (global_load_lds_dwordx4) per 8 MFMAs
      d4_t acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
      const int it2 = iters, ns2 = nstore / 2;
      double* p = U + ((long long)blockIdx.x * 8 + w) * (long long)ns2 * 128 + (threadIdx.x & 63) * 2;
      const double* src = U + (1ll << 27) + ((blockIdx.x & 31) * 8 + w) * 8192 + (threadIdx.x & 63) * 2;  // 2 MB window
      const unsigned ldsb = unsigned(size_t((__attribute__((address_space(3))) char*)lds)) + w * 8192;
      double2_u ring[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) ring[r] = double2_u{0.0, 0.0};
      int done = 0;
      double a = operand(0), b = operand(1);
      for (int it = 0; it < it2; it += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          a += ring[r].x * 1e-300;  // (use of the value loaded 8 iterations ago)
          if (mode == SAME2D) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(ldsb + (r & 7) * 1024)), "v"(src + ((it + r) & 63) * 128) : "memory");
          } else {
            ring[r] = *reinterpret_cast<const double2_u*>(src + ((it + r) & 63) * 128);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
          if (mode != SAME2GN && (r & 1) && done < ns2) { *reinterpret_cast<double2_u*>(p + (long long)done * 128) = double2_u{1.0 * it, 2.0}; ++done; }
        }
      }
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      if (s == 12345.678) sink[0] = s;
      if (mode == SAME2D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      break;
    }
    case WAVES:
      if (w < 4) mfma_loop(iters, sink);
      else store_loop(U, slot4, nstore);
      break;
    case WAVES_PS:  // ... the storing waves at raised priority
      if (w < 4) mfma_loop(iters, sink);
      else { __builtin_amdgcn_s_setprio(3); store_loop(U, slot4, nstore); }
      break;
    case WAVES_PM:  // ... the multiplying waves at raised priority
      if (w < 4) { __builtin_amdgcn_s_setprio(3); mfma_loop(iters, sink); }
      else store_loop(U, slot4, nstore);
      break;
    case WAVES_SW:   // roles swapped: waves 0-3 (the older ones) store
    case WAVES_SWP:  // ... and the multiplying waves at raised priority
      if (w >= 4) { if (mode == WAVES_SWP) __builtin_amdgcn_s_setprio(3); mfma_loop(iters, sink); }
      else store_loop(U, slot4, nstore);
      break;
    case WAVES_NV:   // stores with immediate offsets: one address update per four stores (little VALU work)
    case WAVES_NVP:  // ... at raised priority
      if (w < 4) mfma_loop(iters, sink);
      else {
        if (mode == WAVES_NVP) __builtin_amdgcn_s_setprio(3);
        double* p = U + slot4 * (long long)nstore * 128 + (threadIdx.x & 63) * 2;
        const double2_u v = double2_u{1.0, 2.0};
        for (int i = 0; i < nstore; i += 4, p += 512)
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %1, off offset:1024\n\t"
                       "global_store_dwordx4 %0, %1, off offset:2048\n\tglobal_store_dwordx4 %0, %1, off offset:3072" ::"v"(p), "v"(v) : "memory");
      }
      break;
    case CUS:
      if (w < 4) {
        if (cu_odd) store_loop(U, slot4, 2 * nstore);
        else mfma_loop(2 * iters, sink);
      }
      break;
    case MFMA_HALF:
      if (w < 4 && !cu_odd) mfma_loop(2 * iters, sink);
      break;
    case STORE_HALF:
      if (w < 4 && cu_odd) store_loop(U, slot4, 2 * nstore);
      break;
    case FULL_A:    // x2' + DMA, stepwise towards the structure of k_extend_p: A = 2 DMA loads per 8 MFMAs
    case FULL_B:    // B = A + the MFMA operands read from the LDS region the DMA writes (6 ds_read_b64 per 8 MFMAs)
    case FULL_C:    // C = B + s_barrier every 32 MFMAs (all 8 waves)
    case FULL_D:    // D = C + s_waitcnt vmcnt(6) before the barrier
    case FULL_E:    // E = D with every DMA load gathering 16 rows x 64 bytes (row stride 27 KB) instead of 1 KB contiguous
    case FULL_F: {  // F = E with the rows spread over a 32 MB table (L2 misses) instead of a 2 MB window
      d4_t acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
      for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = operand(i & 1);
      __syncthreads();
      const int it2 = iters, ns2 = nstore / 2;
      double* p = U + ((long long)blockIdx.x * 8 + w) * (long long)ns2 * 128 + (threadIdx.x & 63) * 2;
      const double* src = U + (1ll << 27) + ((blockIdx.x & 31) * 8 + w) * 8192 + (threadIdx.x & 63) * 2;
      const bool gather = mode == FULL_E || mode == FULL_F;
      // gather: lane -> row (lane >> 2), 16-byte unit (lane & 3) of a 64-byte segment; rows 3424 doubles apart
      const double* gsrc = U + (1ll << 27) + (long long)((threadIdx.x & 63) >> 2) * 3424 + (threadIdx.x & 3) * 2 +
                           (mode == FULL_F ? (long long)((blockIdx.x * 8 + w) & 255) * 16384 : (long long)(w & 3) * 8);
      const unsigned ldsb = unsigned(size_t((__attribute__((address_space(3))) char*)lds)) + w * 8192;
      const double* la = lds + (threadIdx.x & 63) * 9 + w * 1024;
      int done = 0;
      double a[4] = {operand(0), operand(1), operand(0), operand(1)}, b[2] = {operand(1), operand(0)};
      for (int it = 0; it < it2; it += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double* s0 = gather ? gsrc + ((it + r) & 63) * 8 + (mode == FULL_F ? (long long)((it >> 2) & 63) * 65536 : 0) : src + ((it + r) & 63) * 128;
          const double* s1 = gather ? s0 + 16 * 3424 : src + ((it + r + 32) & 63) * 128;
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(ldsb + (2 * r) * 1024)), "v"(s0) : "memory");
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(ldsb + (2 * r + 1) * 1024)), "v"(s1) : "memory");
          if (mode != FULL_A) {
            const double* q = la + ((it + r) & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = q[i * 640];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = q[4096 + j * 640];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
          if ((r & 1) && done < ns2) {
            const unsigned long long m16 = ~0ull;
            unsigned long long sv;
            const double2_u pr = double2_u{1.0 * it, 2.0};
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %2, %3, off\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(m16), "v"(p + (long long)done * 128), "v"(pr));
            ++done;
          }
        }
        if (mode == FULL_C) asm volatile("s_barrier" ::: "memory");
        if (mode >= FULL_D) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
      }
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      if (s == 12345.678) sink[0] = s;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      break;
    }
    case STORE_8TH:  // one CU in eight stores (its own share only): what ONE CU can push when the memory side is idle
      if (w < 4 && ((hwid >> 8) & 7) == 0) store_loop(U, slot4, nstore);
      break;
  }
}

int main(int argc, char** argv) {
  const int ncu = 256;
  const int data = argc > 1 ? atoi(argv[1]) : 0, mult = argc > 2 ? atoi(argv[2]) : 1;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_data), &data, sizeof(int)));
  const int iters = 253 * mult;   // x 16 MFMAs x 2048 flop x 1024 SIMDs = 8.49 GFLOP
  const int nstore = 504 * mult;  // x 1 KB x 1024 waves = 528 MB
  double *U, *sink;
  CK(hipMalloc(&U, (size_t(1) << 30) + size_t(ncu) * 4 * 2 * nstore * 1024 + (4 << 20)));
  CK(hipMemset(U, 0, (size_t(1) << 30) + (4 << 20)));
  CK(hipMalloc(&sink, 64));
  unsigned* cnt;
  CK(hipMalloc(&cnt, 8));
  CK(hipMemset(cnt, 0, 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"mfma", "store", "same", "waves", "cus", "mfma/2", "store/2", "same x2", "same x2'", "x2' + LDS", "x2' LDS only", "x2' exec", "x2' + loads", "x2' loads only", "x2' + DMA", "store/8", "full A", "full B", "full C", "full D", "full E", "full F", "waves, store prio", "waves, mfma prio", "waves swapped", "swapped, mfma prio", "waves, imm stores", "imm stores, prio"};
  const double gflop = double(iters) * 16 * 2048 * 4 * ncu * 1e-9, mb = double(nstore) * 1024 * 4 * ncu * 1e-6;
  printf("operands: %s; ", data ? "full mantissas" : "within 1e-6 of 1");
  printf("%.2f GFLOP of fp64 MFMA, %.1f MB of stores per launch, one 256/512-thread workgroup on each of %d CUs\n", gflop, mb, ncu);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 28; ++mode) {
      float best = 1e30f;
      for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0));
        k_probe<<<ncu, 512, 128 * 1024>>>(U, sink, rep == 0 && mode == 0 && t == 0 ? cnt : nullptr, mode, iters, nstore);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      if (rep == 2 && mode == STORE_8TH)
        printf("%-8s %.4f ms   (one CU in eight writes its 2.06 MB: %.1f bytes per cycle and CU at 2.4 GHz)\n", names[mode], best, mb / ncu * 1e6 / (best * 1e-3 * 2.4e9));
      else if (rep == 2)
        printf("%-20s %.4f ms   (%5.1f TFLOP/s %s, %5.2f TB/s %s)\n", names[mode], best, gflop / best,
               mode == STORE || mode == STORE_HALF ? "-" : "mfma", mb / best * 1e-3, mode == MFMA || mode == MFMA_HALF ? "-" : "stores");
    }
  CK(hipMemset(sink, 0, 64));
  CK(hipEventRecord(e0));
  k_probe<<<ncu, 512, 128 * 1024>>>(U, sink, nullptr, MFMA, iters, nstore);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  {
    float ms;
    double hs[2];
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hs, sink, 16, hipMemcpyDeviceToHost));
    printf("mfma alone: %.2f ticks of the cycle counter (s_memtime) per MFMA in one wave; the launch took %.1f us = %.1f ns per MFMA of a wave\n",
           hs[1], ms * 1e3, ms * 1e6 / (16.0 * iters));
  }
  unsigned h[2];
  CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
  printf("workgroups on CUs with even / odd CU_ID: %u / %u\n", h[0], h[1]);
  return 0;
}
