// Store-only probes of the extension kernels' write patterns (dev tool, round 2): how fast do the 528 MB of a C2 step
// (1024 systems x 4 blocks x 127 x 127 doubles) reach memory when written
//   (a) the way k_extend_s writes them (a wave owns 16 systems and walks along the block in 64-vertex tiles, MFMA D
//       layout, 16 bytes per lane: 4 systems x 256 bytes per instruction),
//   (b) the way k_extend128 does (128 systems x one mesh row per workgroup),
// into the reference layout (rows of 255 x 255 doubles: a block's run of a mesh row starts at an arbitrary multiple of 8
// bytes) and into a PADDED layout (mesh rows of 256 doubles, row bases 128-byte aligned: every 16-vertex group is
// exactly one 128-byte line)?
// build: hipcc -O3 --offload-arch=gfx950 tools/ext_store_patterns.hip -o tools/ext_store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

struct Lay { long long ld; int rs; };  // snapshot-row stride, mesh-row stride (doubles)

// (a) streaming pattern.  unit u -> (block z fastest, system group sg, chunk c); XCD-chunked like k_extend_s
__global__ __launch_bounds__(256) void k_stream(double* U, Lay L, int M, int N, int nz, int nsg, int ntile, int tpc, int per_xcd,
                                                int total, int spin, int tile_w) {
  const int n1 = N - 1;
  const int Lid = blockIdx.x;
  const int u = (Lid & 7) * per_xcd + (Lid >> 3);
  if ((Lid >> 3) >= per_xcd || u >= total) return;
  const int z = u % nz, sg = (u / nz) % nsg, ck = u / (nz * nsg);
  const int p = z / 2, q = z % 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, kq = lane >> 4, odd = lane & 1;
  const int nct = (n1 + tile_w - 1) / tile_w;
  const int t0 = ck * tpc, t1 = min(ntile, t0 + tpc);
  for (int tile = t0; tile < t1; ++tile) {
    const int iv = tile / nct, jc0 = tile_w * (tile % nct);
    for (int hp = 0; hp < tile_w / 32; ++hp) {
      const int jcol = jc0 + (2 * hp + odd) * 16 + fr - odd;
      const long long off = (long long)(p * N + iv) * L.rs + q * N + jcol;
      for (int g = 0; g < 4; ++g) {
        const int m = sg * 64 + w * 16 + kq + 4 * g;
        if (m >= M) continue;
        double* dst = U + m * L.ld + off;
        if (jcol + 1 < n1 || (L.rs == 256 && jcol + 1 < 128)) *reinterpret_cast<double2_u*>(dst) = double2_u{1.0 * tile, 2.0};
        else if (jcol < n1) dst[0] = 1.0;
      }
    }
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(127);  // stand-in for the MFMAs of a tile
  }
}

// (b) k_extend128's pattern: workgroup = 128 systems x one mesh row of one block
__global__ __launch_bounds__(256) void k_t128(double* U, Lay L, int M, int N) {
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
        double* dst = U + m * L.ld + (long long)(p * N + iv) * L.rs + (q * N + jj);
        if (jj + 1 < n1 || L.rs == 256) *reinterpret_cast<double2_u*>(dst) = double2_u{3.0, 4.0};
        else if (jj < n1) dst[0] = 3.0;
      }
    }
}

int main() {
  const int M = 1024, N = 128, n1 = N - 1;
  const Lay ref{255ll * 255, 255}, pad{255ll * 256, 256};
  double* U;
  CK(hipMalloc(&U, size_t(M) * pad.ld * 8 + 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 4.0 * n1 * n1 * M * 8.0;
  auto report = [&](const char* name, float ms) { printf("%-58s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms * 1e-9); };
  for (int rep = 0; rep < 2; ++rep) {
    for (int padded = 0; padded < 2; ++padded) {
      const Lay L = padded ? pad : ref;
      float ms;
      for (int tw : {64, 128}) {
        for (int occ : {3, 2}) {
          const int nz = 4, nsg = M / 64, nct = (n1 + tw - 1) / tw, ntile = n1 * nct;
          int nchunk = occ * 256 / (nz * nsg);
          const int tpc = (ntile + nchunk - 1) / nchunk;
          nchunk = (ntile + tpc - 1) / tpc;
          const int total = nchunk * nz * nsg, per = (total + 7) / 8;
          CK(hipEventRecord(e0));
          k_stream<<<8 * per, 256>>>(U, L, M, N, nz, nsg, ntile, tpc, per, total, 0, tw);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
          char nm[128];
          snprintf(nm, 128, "stream %d-vertex tiles, %d WG/CU, %s layout", tw, occ, padded ? "padded" : "reference");
          report(nm, ms);
        }
      }
      CK(hipEventRecord(e0));
      k_t128<<<dim3(n1, M / 128, 4), 256>>>(U, L, M, N);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      report(padded ? "128 x 128 workgroup tiles, padded layout" : "128 x 128 workgroup tiles, reference layout", ms);
    }
  }
  return 0;
}
