// Store-only probes, round 4: which WRITE PATTERN into the reference layout (snapshot rows of 255 x 255 doubles) lets the
// 528 MB of a C2 step (1024 systems x 4 blocks x 127 x 127 doubles) reach memory fastest?  Round 2 measured the pattern of
// k_extend128 (4 systems x 256 B per wave instruction, pieces starting at arbitrary multiples of 8 B) at 3.4-3.6 TB/s
// against 5.5-6.5 TB/s for line-aligned kilobytes; the extension kernel (0.205 ms) is longer than its stores alone
// (0.15 ms) by little.  Candidates (all write exactly the bytes of the real kernel unless noted):
//   B0  k_extend128's pattern, 8 waves (2 x 4 wave tiles of 64 systems x 32 vertices)
//   P1  one wave instruction = the whole run of ONE system (127 doubles, 16 B per lane at 8-byte alignment)
//   P2  the same with every lane's 16 B aligned to 16 B (63 pairs + one single element in its own masked instruction)
//   P3  P2, workgroup order: system group = id % 8 (one per XCD), neighbours in the snapshot row adjacent in time
//   P4  full mesh rows (255 doubles incl. the interface vertex) per workgroup, aligned lanes, order as P3
//   P5  persistent shape: a workgroup owns R consecutive full mesh rows of 128 systems and writes LINE-ALIGNED kilobytes
//       (what an LDS-staged epilogue with a carried partial line would emit); only the ends of its stretch are partial
//   P6  2 systems x 512 B per instruction (register transposition of a 32 x 64 wave tile), 8-byte alignment
//   P7  P5 with one system per wave instruction pair interleaved the way tiles finish (piece-major), 64 systems per workgroup
// build: hipcc -O3 --offload-arch=gfx950 tools/ext_store_patterns2.hip -o tools/ext_store_patterns2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
typedef double double2_a __attribute__((ext_vector_type(2), aligned(16)));

constexpr int N = 128, n1 = 127, NC = 255;
constexpr long long LD = 255ll * 255;

__device__ inline void st16u(double* p, double a, double b) { *reinterpret_cast<double2_u*>(p) = double2_u{a, b}; }
__device__ inline void st16a(double* p, double a, double b) { *reinterpret_cast<double2_a*>(p) = double2_a{a, b}; }

// B0: k_extend128's epilogue pattern
__global__ __launch_bounds__(512) void k_b0(double* U, int M) {
  const int b = blockIdx.z, p = b / 2, q = b % 2, iv = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3;
  const int fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  const int t = wc * 32 + (odd ? 16 : 0) + fr - odd;
  for (int i = 0; i < 4; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 128 + wr * 64 + i * 16 + kq + 4 * g;
      if (m >= M) continue;
      double* dst = U + m * LD + (long long)(p * N + iv) * NC + (q * N + t);
      if (t + 1 < n1) st16u(dst, 3.0, 4.0);
      else if (t < n1) dst[0] = 3.0;
    }
}


// Round 5 ("a workgroup that owns two adjacent mesh rows of a block").  The per-instruction pattern of
// k_extend128 (B0) unchanged; what changes is WHICH stores leave one CU together:
//   B2r  one 1024-thread workgroup = two ADJACENT MESH ROWS (iv, iv + 1) of one block, waves 0-7 / 8-15 (the literal reading:
//        in the reference layout those are two separate 1016-B runs 2040 B apart, not one 2032-B run)
//   B2q  one 1024-thread workgroup = the same mesh row of the two blocks q = 0, 1 side by side: the two halves of ONE snapshot-row
//        line run (127 | interface vertex | 127), the reading under which the partial lines between the halves meet in one L2
//   B2t  a 64-system x 256-vertex tile (rows iv, iv + 1 of one block), eight waves of 32 systems x 64 vertices
__global__ __launch_bounds__(1024) void k_b2(double* U, int M, int mode) {
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 7, half = threadIdx.x >> 9, wr = w >> 2, wc = w & 3;
  int p, q, iv;
  if (mode == 0) { const int b = blockIdx.z; p = b / 2; q = b % 2; iv = 2 * blockIdx.x + half; }   // B2r
  else { p = blockIdx.z; q = half; iv = blockIdx.x; }                                              // B2q
  if (iv >= n1) return;
  const int fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  const int t = wc * 32 + (odd ? 16 : 0) + fr - odd;
  for (int i = 0; i < 4; ++i)
    for (int g = 0; g < 4; ++g) {
      const int m = blockIdx.y * 128 + wr * 64 + i * 16 + kq + 4 * g;
      if (m >= M) continue;
      double* dst = U + m * LD + (long long)(p * N + iv) * NC + (q * N + t);
      if (t + 1 < n1) st16u(dst, 3.0, 4.0);
      else if (t < n1) dst[0] = 3.0;
    }
}
__global__ __launch_bounds__(512) void k_b2t(double* U, int M) {
  const int b = blockIdx.z, p = b / 2, q = b % 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3;   // 2 x 4 wave tiles of 32 systems x 64 vertices
  const int fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  for (int j = 0; j < 2; ++j) {
    const int tv = wc * 64 + j * 32 + (odd ? 16 : 0) + fr - odd;   // 0 .. 255: vertex of the two-row tile
    const int iv = 2 * blockIdx.x + tv / 128, t = tv % 128;
    if (iv >= n1) continue;
    for (int i = 0; i < 2; ++i)
      for (int g = 0; g < 4; ++g) {
        const int m = blockIdx.y * 64 + wr * 32 + i * 16 + kq + 4 * g;
        if (m >= M) continue;
        double* dst = U + m * LD + (long long)(p * N + iv) * NC + (q * N + t);
        if (t + 1 < n1) st16u(dst, 3.0, 4.0);
        else if (t < n1) dst[0] = 3.0;
      }
  }
}

// decode a workgroup id: ord 0 = grid (row, sysgroup, block) as launched; ord 1 = sysgroup fastest (one per XCD at 8
// groups), then block column q, then row, then block row p
__device__ inline void decode(int ord, int nsg, int& iv, int& sg, int& p, int& q) {
  if (ord == 0) {
    iv = blockIdx.x; sg = blockIdx.y; p = blockIdx.z / 2; q = blockIdx.z % 2;
  } else {
    int id = blockIdx.x;
    sg = id % nsg; id /= nsg;
    q = id % 2; id /= 2;
    iv = id % n1; p = id / n1;
  }
}

// P1 / P2 / P3: one instruction = one system's run of a block's mesh row
template <bool ALIGNED>
__global__ __launch_bounds__(512) void k_run(double* U, int M, int ord, int nsg) {
  int iv, sg, p, q;
  decode(ord, nsg, iv, sg, p, q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long off = (long long)(p * N + iv) * NC + q * N;
  for (int s = 0; s < 16; ++s) {
    const int m = sg * 128 + w * 16 + s;
    if (m >= M) continue;
    const long long a = m * LD + off;  // first double of the run
    if (!ALIGNED) {
      double* dst = U + a + 2 * lane;
      if (lane < 63) st16u(dst, 1.0, 2.0);
      else dst[0] = 1.0;
    } else {
      const int par = int(a & 1);
      double* dst = U + (a - par) + 2 * lane;
      if (par == 0) {
        if (lane < 63) st16a(dst, 1.0, 2.0);
        if (lane == 63) dst[0] = 1.0;
      } else {
        if (lane > 0) st16a(dst, 1.0, 2.0);
        if (lane == 0) dst[1] = 1.0;
      }
    }
  }
}

// P6: 2 systems x 512 B per instruction (8-byte alignment), workgroup = 128 systems x one block mesh row
__global__ __launch_bounds__(512) void k_half(double* U, int M, int ord, int nsg) {
  int iv, sg, p, q;
  decode(ord, nsg, iv, sg, p, q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int l32 = lane & 31, hs = lane >> 5;
  const long long off = (long long)(p * N + iv) * NC + q * N;
  for (int s = 0; s < 16; ++s) {
    const int m = sg * 128 + wr * 32 + 2 * s + hs;
    if (m >= M) continue;
    const int t = wc * 64 + 2 * l32;
    double* dst = U + m * LD + off + t;
    if (t + 1 < n1) st16u(dst, 1.0, 2.0);
    else if (t < n1) dst[0] = 1.0;
  }
}

// P4: full mesh row (255 doubles) per system, aligned lanes; grid: ord 0 (row, sysgroup, p), ord 1: sysgroup fastest
__global__ __launch_bounds__(512) void k_fullrow(double* U, int M, int ord, int nsg) {
  int iv, sg, p;
  if (ord == 0) { iv = blockIdx.x; sg = blockIdx.y; p = blockIdx.z; }
  else { int id = blockIdx.x; sg = id % nsg; id /= nsg; iv = id % n1; p = id / n1; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long off = (long long)(p * N + iv) * NC;
  for (int s = 0; s < 16; ++s) {
    const int m = sg * 128 + w * 16 + s;
    if (m >= M) continue;
    const long long a = m * LD + off, e = a + NC;
    const long long a0 = a & ~1ll;
    for (int k = 0; k < 3; ++k) {
      const long long x = a0 + 2 * (lane + 64 * k);
      if (x >= a && x + 1 < e) st16a(U + x, 1.0, 2.0);
      else if (x + 1 == a) U[a] = 1.0;
      else if (x + 1 == e) U[x] = 1.0;
    }
  }
}

// P5 / P7: persistent shape.  Workgroup = SYS systems x R consecutive full mesh rows of block row p; every instruction
// writes a line-aligned kilobyte of ONE system (the two ends of the stretch are partial); pieces outermost.
template <int SYS>
__global__ __launch_bounds__(512) void k_lines(double* U, int M, int R, int nsg, int nchunk) {
  int id = blockIdx.x;
  const int sg = id % nsg; id /= nsg;
  const int ck = id % nchunk, p = id / nchunk;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = ck * R, r1 = min(n1, r0 + R);
  if (r0 >= r1) return;
  constexpr int SPW = SYS / 8;  // systems per wave
  const long long off0 = (long long)(p * N + r0) * NC, len = (long long)(r1 - r0) * NC;
  const int npiece = int((len + 127 + 127) / 128);
  for (int k = 0; k < npiece; ++k)
    for (int s = 0; s < SPW; ++s) {
      const int m = sg * SYS + w * SPW + s;
      if (m >= M) continue;
      const long long a = m * LD + off0, e = a + len;
      const long long x = (a & ~15ll) + 128ll * k + 2 * lane;
      if (x >= a && x + 1 < e) st16a(U + x, 1.0, 2.0);
      else if (x + 1 == a) U[a] = 1.0;
      else if (x + 1 == e) U[x] = 1.0;
    }
}

int main(int argc, char** argv) {
  const int M = 1024;
  double* U;
  CK(hipMalloc(&U, size_t(M) * LD * 8 + 65536));
  CK(hipMemset(U, 0, size_t(M) * LD * 8 + 65536));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 4.0 * n1 * n1 * M * 8.0;
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  const char* only = argc > 2 ? argv[2] : "";
  auto want = [&](const char* tag) { return only[0] == 0 || strstr(only, tag) != nullptr; };
#define TIME(NAME, TAG, ...)                                                                   \
  if (want(TAG)) {                                                                             \
    float best = 1e9f, ms;                                                                     \
    for (int it = 0; it < 5; ++it) {                                                           \
      CK(hipEventRecord(e0));                                                                  \
      __VA_ARGS__;                                                                             \
      CK(hipEventRecord(e1));                                                                  \
      CK(hipEventSynchronize(e1));                                                             \
      CK(hipEventElapsedTime(&ms, e0, e1));                                                    \
      best = ms < best ? ms : best;                                                            \
    }                                                                                          \
    CK(hipGetLastError());                                                                     \
    printf("%-4s %-86s %.4f ms  %.2f TB/s\n", TAG, NAME, best, bytes / best * 1e-9);           \
  }
  for (int rep = 0; rep < reps; ++rep) {
    TIME("k_extend128's pattern: 4 systems x 256 B per instruction, 8-byte alignment", "B0", (k_b0<<<dim3(n1, M / 128, 4), 512>>>(U, M)));
    TIME("B0's instructions, 1024-thread workgroup = two adjacent mesh rows of one block", "B2r", (k_b2<<<dim3((n1 + 1) / 2, M / 128, 4), 1024>>>(U, M, 0)));
    TIME("B0's instructions, 1024-thread workgroup = one mesh row of both blocks q = 0, 1", "B2q", (k_b2<<<dim3(n1, M / 128, 2), 1024>>>(U, M, 1)));
    TIME("64 systems x 256 vertices (two mesh rows) per 512-thread workgroup", "B2t", (k_b2t<<<dim3((n1 + 1) / 2, M / 64, 4), 512>>>(U, M)));
    TIME("one system's 1016-B run per instruction, 8-byte alignment", "P1", (k_run<false><<<dim3(n1, M / 128, 4), 512>>>(U, M, 0, 8)));
    TIME("one system's run per instruction, 16-byte aligned lanes", "P2", (k_run<true><<<dim3(n1, M / 128, 4), 512>>>(U, M, 0, 8)));
    TIME("P1 + order: system group = id % 8, row neighbours adjacent", "P1o", (k_run<false><<<n1 * 8 * 4, 512>>>(U, M, 1, 8)));
    TIME("P2 + order: system group = id % 8, row neighbours adjacent", "P3", (k_run<true><<<n1 * 8 * 4, 512>>>(U, M, 1, 8)));
    TIME("2 systems x 512 B per instruction, 8-byte alignment", "P6", (k_half<<<dim3(n1, M / 128, 4), 512>>>(U, M, 0, 8)));
    TIME("P6 + order", "P6o", (k_half<<<n1 * 8 * 4, 512>>>(U, M, 1, 8)));
    TIME("full mesh rows (255 doubles), aligned lanes, grid (row, group, p)", "P4", (k_fullrow<<<dim3(n1, M / 128, 2), 512>>>(U, M, 0, 8)));
    TIME("full mesh rows, aligned lanes, system group = id % 8", "P4o", (k_fullrow<<<n1 * 8 * 2, 512>>>(U, M, 1, 8)));
    TIME("persistent: 128 systems x 8 full rows, line-aligned KB per instruction", "P5", (k_lines<128><<<8 * 16 * 2, 512>>>(U, M, 8, 8, 16)));
    TIME("persistent: 128 systems x 4 full rows (2 workgroups per CU)", "P5b", (k_lines<128><<<8 * 32 * 2, 512>>>(U, M, 4, 8, 32)));
    TIME("persistent: 64 systems x 16 full rows, line-aligned KB per instruction", "P7", (k_lines<64><<<16 * 8 * 2, 512>>>(U, M, 16, 16, 8)));
    TIME("persistent: 64 systems x 8 full rows (2 workgroups per CU)", "P7b", (k_lines<64><<<16 * 16 * 2, 512>>>(U, M, 8, 16, 16)));
    TIME("persistent: 64 systems x 1 full row (finest)", "P7c", (k_lines<64><<<16 * 127 * 2, 512>>>(U, M, 1, 16, 127)));
    TIME("persistent: 128 systems x 1 full row", "P5c", (k_lines<128><<<8 * 127 * 2, 512>>>(U, M, 1, 8, 127)));
  }
  return 0;
}
