// Dense fp64 MFMA contractions, stencil application, norms and the batched reduced solves.
//
// Reference counterparts: H10norm / l2norm (src/lib/SolutionsManagers.py:56-62), the
// `C A_pq C^T`, `C A_1 U^T`, `c_i . basis` einsums of generate_fm_solutions (:93-106) and
// project_solutions (:113-139), and the reduced `galerkin` solves (:104-105, :135-138).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "rom_mma.h"
#include "rom_ops.h"

// ============================================================================================
// generic NT GEMM:  C[m,n] = alpha * sum_k A[m,k] B[n,k] + beta C      (64x64 tiles, split-K)
// ============================================================================================
// bounds-checked load of 4 consecutive k of one row; ALIGNED => 16-B vector loads are legal
template <bool ALIGNED>
__device__ inline void load4_row(const double* __restrict__ row, long long k, long long kend, double v[4]) {
  if (row == nullptr || k >= kend) {
    v[0] = v[1] = v[2] = v[3] = 0.0;
    return;
  }
  if (ALIGNED && k + 4 <= kend) {
    load4_aligned(row + k, v);
    return;
  }
#pragma unroll
  for (int x = 0; x < 4; ++x) v[x] = (k + x < kend) ? row[k + x] : 0.0;
}

// grid (ceil(n/64), ceil(m/64), splits).  Each z-slice handles K range [z*kper, min(K,(z+1)*kper)).
// splits == 1: writes alpha*acc + beta*C directly; else writes the raw partial tile to `part`
// ([z][m][n], ld n) for the deterministic reduction kernel below.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_gemm_nt(long long m, long long n, long long K, long long kper,
                                                 double alpha, const double* __restrict__ A, long long lda,
                                                 const double* __restrict__ B, long long ldb, double beta,
                                                 double* __restrict__ C, long long ldc, double* __restrict__ part,
                                                 int lower_only) {
  __shared__ __align__(16) double stage[STAGE_TOTAL];
  // Gram (lower_only): 1-D grid over the tiles on/below the diagonal, so that consecutive block ids
  // (dealt round-robin to the 8 XCDs) all carry work; the strict upper triangle comes from the mirror pass
  long long ty = blockIdx.y, tx = blockIdx.x;
  if (lower_only) {
    const long long t = blockIdx.x;
    ty = (long long)((sqrt(8.0 * double(t) + 1.0) - 1.0) * 0.5);
    while (ty * (ty + 1) / 2 > t) --ty;
    while ((ty + 1) * (ty + 2) / 2 <= t) ++ty;
    tx = t - ty * (ty + 1) / 2;
  }
  const WavePos wp;
  const long long r0 = ty * 64LL, c0 = tx * 64LL;
  const long long kbeg = blockIdx.z * kper, kend = std::min<long long>(K, kbeg + kper);
  Acc acc;
  acc_zero(acc);
  const int nch = int((kend - kbeg + BK - 1) / BK);
  if (ALIGNED) {  // rows 16-byte aligned: each thread streams 32 contiguous bytes of one row
    const int srow = stage_row(), sseg = stage_seg();
    const double* Ar = (r0 + srow < m) ? A + (r0 + srow) * lda : nullptr;
    const double* Br = (c0 + srow < n) ? B + (c0 + srow) * ldb : nullptr;
    gemm_loop(
        nch, [&](int ch, double* v) { load4_row<ALIGNED>(Ar, kbeg + ch * BK + sseg, kend, v); },
        [&](int ch, double* v) { load4_row<ALIGNED>(Br, kbeg + ch * BK + sseg, kend, v); }, acc, stage, wp);
  } else {  // odd leading dimension: lane -> k, a wave-instruction covers 4 rows x 128 contiguous bytes
    const long long kk = kmajor_k();
    const double* Ar[4];
    const double* Br[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      Ar[x] = (r0 + kmajor_row(x) < m) ? A + (r0 + kmajor_row(x)) * lda : nullptr;
      Br[x] = (c0 + kmajor_row(x) < n) ? B + (c0 + kmajor_row(x)) * ldb : nullptr;
    }
    gemm_loop_kmajor(
        nch,
        [&](int ch, double* v) {
          const long long k = kbeg + ch * BK + kk;
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] = (Ar[x] && k < kend) ? Ar[x][k] : 0.0;
        },
        [&](int ch, double* v) {
          const long long k = kbeg + ch * BK + kk;
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] = (Br[x] && k < kend) ? Br[x][k] : 0.0;
        },
        acc, stage, wp);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      long long r = r0 + acc_row(wp, i, g);
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        long long c = c0 + acc_col(wp, j);
        if (c >= n) continue;
        double v = acc.c[i][j][g];
        if (part) {
          part[(blockIdx.z * m + r) * n + c] = v;
        } else {
          double* p = C + r * ldc + c;
          *p = beta == 0.0 ? alpha * v : alpha * v + beta * *p;
        }
      }
    }
}

__global__ void k_mirror_lower(long long n, double* __restrict__ C, long long ldc) {
  long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  long long r = idx / n, c = idx % n;
  if (c > r) C[r * ldc + c] = C[c * ldc + r];
}

// (lower_only == 2: the partials are those of the TRANSPOSED product, C[c, r] receives element (r, c))
__global__ void k_splitk_reduce(long long m, long long n, int splits, double alpha, const double* __restrict__ part,
                                double beta, double* __restrict__ C, long long ldc, int lower_only) {
  long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= m * n) return;
  if (lower_only == 1 && (idx % n) / 64 > (idx / n) / 64) return;
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += part[z * m * n + idx];
  long long r = idx / n, c = idx % n;
  double* p = lower_only == 2 ? C + c * ldc + r : C + r * ldc + c;
  *p = beta == 0.0 ? alpha * s : alpha * s + beta * *p;
}

// the same reduction for outputs of a few thousand to tens of thousands of entries with dozens of splits (the b x M
// coefficient rows of a thin product over a snapshot-long K): 64 consecutive entries x 4 groups of splits per workgroup --
// every wave reads 512 contiguous bytes per split (the one-wave-per-entry form below reads a 64-byte sector for 8 of its
// bytes: 21 us for 25 MB of partials), four times the threads of the entry-per-thread form; fixed order: deterministic
__global__ __launch_bounds__(256) void k_splitk_reduce_quad(long long m, long long n, int splits, double alpha,
                                                            const double* __restrict__ part, double beta,
                                                            double* __restrict__ C, long long ldc, int lower_only) {
  __shared__ double red[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long idx = blockIdx.x * 64LL + e;
  const bool in = idx < m * n;
  double s = 0.0;
  if (in)
    for (int z = g; z < splits; z += 4) s += part[z * m * n + idx];
  red[g][e] = s;
  __syncthreads();
  if (g != 0 || !in) return;
  if (lower_only == 1 && (idx % n) / 64 > (idx / n) / 64) return;
  s = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
  const long long r = idx / n, c = idx % n;
  double* p = lower_only == 2 ? C + c * ldc + r : C + r * ldc + c;
  *p = beta == 0.0 ? alpha * s : alpha * s + beta * *p;
}

// the same reduction with one WAVE per output element (the splits are spread over its lanes): thin outputs with hundreds
// of splits -- m x 1 dot products over a snapshot-long K -- are latency bound with one thread walking all the partials
__global__ __launch_bounds__(256) void k_splitk_reduce_wave(long long m, long long n, int splits, double alpha,
                                                            const double* __restrict__ part, double beta,
                                                            double* __restrict__ C, long long ldc, int lower_only) {
  const long long idx = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (idx >= m * n) return;
  if (lower_only == 1 && (idx % n) / 64 > (idx / n) / 64) return;
  const int lane = threadIdx.x & 63;
  // fixed order: lane l sums splits l, l + 64, ...; then the lanes are folded pairwise -- deterministic
  double s = 0.0;
  for (int z = lane; z < splits; z += 64) s += part[z * m * n + idx];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) {
    const long long r = idx / n, c = idx % n;
    double* p = lower_only == 2 ? C + c * ldc + r : C + r * ldc + c;
    *p = beta == 0.0 ? alpha * s : alpha * s + beta * *p;
  }
}

// A/B switch (read once): the 64 x 64 register-staged engine for the thin shapes too
static bool no_thin_gemm() {
  static const bool off = getenv("ROMHC_NO_THIN_GEMM") != nullptr;
  return off;
}
template <int MI>
__global__ void k_gemm_nt_thin(long long m, long long n, long long K, long long kper, const double* __restrict__ A, long long lda,
                               const double* __restrict__ B, long long ldb, double* __restrict__ part);  // (defined below)
// thin A (m <= 64) against a long K: k_gemm_nt_thin over 128-row tiles of B x split-K, then the deterministic reduction
// transposed != 0: (A, m, lda) is the THIN operand of C^T = A B^T, i.e. the caller's product is C (n x m) = B A^T
static int launch_gemm_nt_thin(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                               const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const char* prof_name,
                               int transposed = 0) {
  const long long tiles = (n + 127) / 128;
  // one full round of the 768 resident workgroups (3 per CU), at least 512 columns of K per workgroup
  long long splits = std::max<long long>(1, std::min<long long>((768 + tiles - 1) / tiles, k / 512));
  long long kper = ((k + splits - 1) / splits + BK - 1) / BK * BK;
  splits = (k + kper - 1) / kper;
  double* part = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(splits) * m * n, &part));
  const dim3 grid{unsigned(tiles), 1u, unsigned(splits)};
  {
    static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;
    char nm[64];
    detail ? snprintf(nm, sizeof nm, "%s_thin_%lldx%lldx%lld_s%lld", prof_name, (long long)m, (long long)n, (long long)k, splits)
           : snprintf(nm, sizeof nm, "%s", prof_name);
    ROM_PROF(ctx, nm, 2.0 * m * n * k, 8.0 * (double(m) * k + double(n) * k + double(splits) * m * n));
    switch ((m + 15) / 16) {
      case 1: k_gemm_nt_thin<1><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, A, lda, B, ldb, part); break;
      case 2: k_gemm_nt_thin<2><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, A, lda, B, ldb, part); break;
      case 3: k_gemm_nt_thin<3><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, A, lda, B, ldb, part); break;
      default: k_gemm_nt_thin<4><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, A, lda, B, ldb, part); break;
    }
  }
  ROM_HIP(hipGetLastError());
  {
    ROM_PROF(ctx, "splitk_reduce", double(splits) * m * n, 8.0 * double(splits + 1) * m * n);
    if (splits >= 8 && m * n >= 4096 && m * n <= (1 << 20))
      k_splitk_reduce_quad<<<unsigned((m * n + 63) / 64), 256, 0, ctx->stream>>>(m, n, int(splits), alpha, part, beta, C, ldc,
                                                                                transposed ? 2 : 0);
    else if (splits >= 16 && m * n <= 65536)
      k_splitk_reduce_wave<<<unsigned((m * n + 3) / 4), 256, 0, ctx->stream>>>(m, n, int(splits), alpha, part, beta, C, ldc,
                                                                               transposed ? 2 : 0);
    else
      k_splitk_reduce<<<unsigned((m * n + 255) / 256), 256, 0, ctx->stream>>>(m, n, int(splits), alpha, part, beta, C, ldc,
                                                                              transposed ? 2 : 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

int rom_launch_gemm_nt(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const char* prof_name) {
  return rom_launch_gemm_nt_ex(ctx, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, prof_name, 0);
}

// lower_only != 0: the product is symmetric (A == B); only tiles on/below the diagonal are computed
// and the strict upper triangle is filled by a mirror pass (half the MFMA work of the Gram matrix).
int rom_launch_gemm_nt_ex(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                          const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const char* prof_name,
                          int lower_only) {
  if (m <= 0 || n <= 0) return ROM_OK;
  if (!lower_only && m <= 64 && n >= 256 && k >= 2048 && size_t(lda) * 8 * 64 < (size_t(1) << 32) &&
      size_t(ldb) * 8 * 128 < (size_t(1) << 32) && !no_thin_gemm())
    return launch_gemm_nt_thin(ctx, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, prof_name);
  if (!lower_only && n <= 64 && m >= 256 && k >= 2048 && size_t(ldb) * 8 * 64 < (size_t(1) << 32) &&
      size_t(lda) * 8 * 128 < (size_t(1) << 32) && !no_thin_gemm())  // thin B: the transposed product on the same kernel
    return launch_gemm_nt_thin(ctx, n, m, k, alpha, B, ldb, A, lda, beta, C, ldc, prof_name, 1);
  const long long nt = (m + 63) / 64;
  const long long tiles = lower_only ? nt * (nt + 1) / 2 : nt * ((n + 63) / 64);  // tiles that do work
  int splits = 1;
  if (k >= 1024 && tiles < 512) {
    // (a handful of output tiles: the launch is one latency chain per workgroup -- 128 columns of K each instead of 512)
    const long long kmin = tiles <= 4 ? 128 : 512;
    splits = int(std::min<long long>((768 + tiles - 1) / tiles, (k + kmin - 1) / kmin));
    splits = std::max(splits, 1);
  }
  long long kper = ((k + splits - 1) / splits + BK - 1) / BK * BK;
  if (kper <= 0) kper = BK;
  splits = int((k + kper - 1) / kper);
  if (splits < 1) splits = 1;
  double* part = nullptr;
  if (splits > 1) ROM_TRY(rom_ctx_scratch(ctx, size_t(splits) * m * n, &part));
  const bool aligned = (lda % 2 == 0) && (ldb % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (kper % 4 == 0);
  dim3 grid(unsigned((n + 63) / 64), unsigned((m + 63) / 64), unsigned(splits));
  if (lower_only) grid = dim3(unsigned(tiles), 1, unsigned(splits));
  {
    static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-shape names in the profile records
    char nm[64];
    detail ? snprintf(nm, sizeof nm, "%s_%lldx%lldx%lld_s%d", prof_name, (long long)m, (long long)n, (long long)k, splits)
           : snprintf(nm, sizeof nm, "%s", prof_name);
    ROM_PROF(ctx, nm, (lower_only ? 1.0 : 2.0) * m * n * k + (lower_only ? 64.0 * n * k : 0.0),
             8.0 * (double(m) * k + double(n) * k + double(m) * n));
    if (aligned)
      k_gemm_nt<true><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, alpha, A, lda, B, ldb, beta, C, ldc, part, lower_only);
    else
      k_gemm_nt<false><<<grid, 256, 0, ctx->stream>>>(m, n, k, kper, alpha, A, lda, B, ldb, beta, C, ldc, part, lower_only);
  }
  ROM_HIP(hipGetLastError());
  if (splits > 1) {
    ROM_PROF(ctx, "splitk_reduce", double(splits) * m * n, 8.0 * double(splits + 1) * m * n);
    if (splits >= 8 && m * n >= 4096 && m * n <= (1 << 20))
      k_splitk_reduce_quad<<<unsigned((m * n + 63) / 64), 256, 0, ctx->stream>>>(m, n, splits, alpha, part, beta, C, ldc, lower_only);
    else if (splits >= 16 && m * n <= 65536)
      k_splitk_reduce_wave<<<unsigned((m * n + 3) / 4), 256, 0, ctx->stream>>>(m, n, splits, alpha, part, beta, C, ldc, lower_only);
    else
      k_splitk_reduce<<<unsigned((m * n + 255) / 256), 256, 0, ctx->stream>>>(m, n, splits, alpha, part, beta, C, ldc,
                                                                             lower_only);
    ROM_HIP(hipGetLastError());
  }
  if (lower_only) {
    ROM_PROF(ctx, "mirror_lower", 0, 8.0 * n * n);
    k_mirror_lower<<<unsigned((n * n + 255) / 256), 256, 0, ctx->stream>>>(n, C, ldc);
    ROM_HIP(hipGetLastError());
  }
  return ROM_OK;
}

extern "C" int rom_gemm_nt(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, rom_buf* A, size_t a_off,
                           int64_t lda, rom_buf* B, size_t b_off, int64_t ldb, double beta, rom_buf* C,
                           size_t c_off, int64_t ldc) {
  ROM_CHECK(ctx && A && B && C, "rom_gemm_nt: null argument");
  ROM_CHECK(m >= 0 && n >= 0 && k >= 0 && lda >= k && ldb >= k && ldc >= n, "rom_gemm_nt: bad dimensions");
  if (m == 0 || n == 0) return ROM_OK;
  ROM_CHECK(a_off + size_t(m - 1) * lda + k <= A->n, "rom_gemm_nt: A out of range");
  ROM_CHECK(b_off + size_t(n - 1) * ldb + k <= B->n, "rom_gemm_nt: B out of range");
  ROM_CHECK(c_off + size_t(m - 1) * ldc + n <= C->n, "rom_gemm_nt: C out of range");
  return rom_launch_gemm_nt(ctx, m, n, k, alpha, A->p + a_off, lda, B->p + b_off, ldb, beta, C->p + c_off, ldc,
                            "gemm_nt");
}

// ============================================================================================
// Gram matrix with 128 x 128 workgroup tiles (each wave a 64 x 64 quadrant = 4 x 4 MFMA accumulators):
// twice the arithmetic intensity of the 64 x 64 engine (16 flop per operand byte), which is what the
// snapshot Gram needs once the block no longer fits in the Infinity Cache (C5: 34 GB).
// 1-D grid over the lower tiles (XCD-balanced) x split-K; partial tiles go to `part`, reduced + mirrored after.
// ============================================================================================
// Main loop (round 2): the K chunks (16 wide) of both operands go from global memory straight into LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write; 8 instructions per wave and chunk: 8 rows x 128 bytes
// each), two 32 KB slots, chunk ch + 1 in flight under the 64 MFMAs per wave of chunk ch; a row's eight 16-byte units
// are stored at position u ^ ((row >> 1) & 7) (the DMA writes rows back to back: no padding possible), which makes the
// MFMA fragment reads conflict free; the fragments of k-step j + 1 are read before the MFMAs of step j.  Rows behind
// the matrix are clamped to its last row (their products are not stored).  The tail of K (at most 15 columns of the
// last split) goes through registers as before.
constexpr int G128_SLOT = 2 * 128 * 128;  // bytes: 128 rows of A, 128 rows of B, 16 doubles each

__global__ __launch_bounds__(256, 2) void k_gram128(long long m, long long K, long long kper, const double* __restrict__ A,
                                                    long long lda, double* __restrict__ part) {
  __shared__ __align__(16) char lds[2 * G128_SLOT];  // 65,536 B: two workgroups per CU
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds));
  const long long t = blockIdx.x;
  long long ty = (long long)((sqrt(8.0 * double(t) + 1.0) - 1.0) * 0.5);
  while (ty * (ty + 1) / 2 > t) --ty;
  while ((ty + 1) * (ty + 2) / 2 <= t) ++ty;
  const long long tx = t - ty * (ty + 1) / 2;
  const long long r0 = ty * 128, c0 = tx * 128;
  const long long kbeg = blockIdx.z * kper, kend = std::min<long long>(K, kbeg + kper);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = w >> 1, wc = w & 1;
  const int fr = lane & 15, kq = lane >> 4;
  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  const int nfull = int((kend - kbeg) / BK), rem = int((kend - kbeg) % BK);

  // ---- fragment addressing: lane (fr, kq) reads row fr (+ 16 i) at k = 4 kki + kq: unit (2 kki + (kq >> 1)) ^ (fr >> 1)
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kki = 0; kki < 4; ++kki) {
    const unsigned uo = unsigned((((2 * kki) ^ (kq >> 1) ^ (fr >> 1)) << 4) + (kq & 1) * 8);
    fa[kki] = unsigned((wr * 64 + fr) * 128) + uo;
    fb[kki] = unsigned(16384 + (wc * 64 + fr) * 128) + uo;
  }
#define G_FRAGS(SLOT_, KKI_, AF_, BF_)                                                                  \
  do {                                                                                                  \
    const char* pa_ = lds + (SLOT_) * G128_SLOT + fa[KKI_];                                             \
    const char* pb_ = lds + (SLOT_) * G128_SLOT + fb[KKI_];                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                  \
      AF_[i_] = *reinterpret_cast<const double*>(pa_ + i_ * 2048);                                      \
      BF_[i_] = *reinterpret_cast<const double*>(pb_ + i_ * 2048);                                      \
    }                                                                                                   \
  } while (0)
  // ---- DMA addressing: a wave fetches rows 32 w .. 32 w + 31 of both operands, 8 rows per instruction (q = 0..3):
  // lane -> row 32 w + 8 q + (lane >> 3), stored unit lane & 7 = logical unit (lane & 7) ^ ((4 (q & 1) + (lane >> 4)) & 7)
  unsigned voA[4], voB[4];  // lane offsets (bytes) behind the scalar bases A + (r0 | c0) * lda + k
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = 32 * w + 8 * q + (lane >> 3);
    const unsigned u16 = unsigned((((lane & 7) ^ ((4 * (q & 1) + (lane >> 4)) & 7))) * 16);
    const long long ra = std::min<long long>(rl, m - 1 - r0), rb = std::min<long long>(rl, m - 1 - c0);
    voA[q] = unsigned(ra * lda * 8) + u16;  // (127 * lda * 8 < 2^32: checked on the host)
    voB[q] = unsigned(rb * lda * 8) + u16;
  }
  const char* sA = reinterpret_cast<const char*>(A + r0 * lda + kbeg);
  const char* sB = reinterpret_cast<const char*>(A + c0 * lda + kbeg);
#define G_DMA(LDS_, BASE_, VOFF_)                                                                                   \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),    \
               "v"(VOFF_), "s"(BASE_)                                                                               \
               : "memory", "m0")
#define G_ISSUE(SLOT_)                                                                                              \
  do {                                                                                                              \
    const unsigned sl_ = lds0 + unsigned(SLOT_) * G128_SLOT + unsigned(w) * 4096;                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) G_DMA(sl_ + q_ * 1024, sA, voA[q_]);                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) G_DMA(sl_ + 16384 + q_ * 1024, sB, voB[q_]);                   \
    sA += BK * 8;                                                                                                   \
    sB += BK * 8;                                                                                                   \
  } while (0)
  if (nfull > 0) {
    G_ISSUE(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    double af[2][4], bf[2][4];
    G_FRAGS(0, 0, af[0], bf[0]);
    for (int ch = 0; ch < nfull; ++ch) {
      const int slot = ch & 1;
      if (ch + 1 < nfull) G_ISSUE(slot ^ 1);  // (everybody left that slot at the barrier behind chunk ch - 1)
#pragma unroll
      for (int kki = 0; kki < 4; ++kki) {
        const int pb = kki & 1;
        if (kki < 3) G_FRAGS(slot, kki + 1, af[pb ^ 1], bf[pb ^ 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][i], bf[pb][j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // chunk ch + 1 is in LDS for everybody
      if (ch + 1 < nfull) G_FRAGS(slot ^ 1, 0, af[0], bf[0]);
    }
  }
#undef G_ISSUE
#undef G_DMA
#undef G_FRAGS
  if (rem > 0) {
    // the last columns of K, zero padded to one chunk: lane -> k (t & 15), thread t handles rows (t >> 4) + 16 x
    double* st = reinterpret_cast<double*>(lds);  // [A 128 x LDK | B 128 x LDK] doubles = 36,864 B
    const int sk = threadIdx.x & 15, sr0 = threadIdx.x >> 4;
    const long long k = kbeg + (long long)nfull * BK + sk;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const long long ra = r0 + sr0 + 16 * x, rb = c0 + sr0 + 16 * x;
      st[(sr0 + 16 * x) * LDK + sk] = (ra < m && k < kend) ? A[ra * lda + k] : 0.0;
      st[128 * LDK + (sr0 + 16 * x) * LDK + sk] = (rb < m && k < kend) ? A[rb * lda + k] : 0.0;
    }
    __syncthreads();
    const double* pa = st + (wr * 64 + fr) * LDK + kq;
    const double* pb = st + 128 * LDK + (wc * 64 + fr) * LDK + kq;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = pa[i * 16 * LDK + kk];
        bf[i] = pb[i * 16 * LDK + kk];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // raw partial tile: part[z][m][m]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long long r = r0 + wr * 64 + i * 16 + (lane >> 4) + 4 * g;
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long c = c0 + wc * 64 + j * 16 + (lane & 15);
        if (c < m) part[(blockIdx.z * m + r) * m + c] = acc[i][j][g];
      }
    }
}

// C = sum_z part[z] on the lower 128-tiles, mirrored into the strict upper triangle
__global__ void k_gram128_finish(long long m, int splits, const double* __restrict__ part, double* __restrict__ C,
                                 long long ldc) {
  long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= m * m) return;
  long long r = idx / m, c = idx % m;
  if (c > r) return;  // the upper triangle is written by the mirror image below the diagonal
  double s = 0.0;
  for (int z = 0; z < splits; ++z) s += part[z * m * m + idx];
  C[r * ldc + c] = s;
  if (c != r) C[c * ldc + r] = s;
}

static int launch_gram128(rom_ctx* ctx, int64_t m, int64_t k, const double* A, int64_t lda, double* C, int64_t ldc) {
  const long long nt = (m + 127) / 128, tiles = nt * (nt + 1) / 2;
  // split K so that the grid fills whole rounds of the 512 resident workgroups (2 per CU) with little tail
  const long long smax = std::max<long long>(1, std::min<long long>(64, (k + 2047) / 2048));
  int splits = 1;
  double best = 1e30;
  for (long long sp = 1; sp <= smax; ++sp) {
    const double rounds = std::ceil(double(tiles * sp) / 512.0);
    const double cost = rounds / double(sp) + 0.002 * sp;  // time ~ rounds * (K / sp); mild penalty on partial traffic
    if (cost < best) { best = cost; splits = int(sp); }
  }
  long long kper = ((k + splits - 1) / splits + BK - 1) / BK * BK;
  splits = int((k + kper - 1) / kper);
  double* part = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(splits) * m * m, &part));
  {
    ROM_PROF(ctx, "gram128", double(tiles) * 2.0 * 128 * 128 * k, 8.0 * (2.0 * tiles * 128 * double(k) + double(m) * m));
    k_gram128<<<dim3(unsigned(tiles), 1, unsigned(splits)), 256, 0, ctx->stream>>>(m, k, kper, A, lda, part);
  }
  ROM_HIP(hipGetLastError());
  {
    ROM_PROF(ctx, "gram128_finish", double(splits) * m * m, 8.0 * double(splits + 1) * m * m);
    k_gram128_finish<<<unsigned((m * m + 255) / 256), 256, 0, ctx->stream>>>(m, splits, part, C, ldc);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

int rom_launch_gram(rom_ctx* ctx, int64_t m, int64_t k, const double* A, int64_t lda, double* C, int64_t ldc) {
  if (m <= 0) return ROM_OK;
  if (m >= 512 && k >= 4096 && size_t(lda) * 8 * 128 < (size_t(1) << 32)) return launch_gram128(ctx, m, k, A, lda, C, ldc);
  return rom_launch_gemm_nt_ex(ctx, m, m, k, 1.0, A, lda, A, lda, 0.0, C, ldc, "gram", 1);
}

extern "C" int rom_gram(rom_ctx* ctx, int64_t m, int64_t k, rom_buf* A, size_t a_off, int64_t lda, rom_buf* C,
                        size_t c_off, int64_t ldc) {
  ROM_CHECK(ctx && A && C, "rom_gram: null argument");
  ROM_CHECK(m >= 0 && k >= 0 && lda >= k && ldc >= m, "rom_gram: bad dimensions");
  if (m == 0) return ROM_OK;
  ROM_CHECK(a_off + size_t(m - 1) * lda + k <= A->n, "rom_gram: A out of range");
  ROM_CHECK(c_off + size_t(m - 1) * ldc + m <= C->n, "rom_gram: C out of range");
  return rom_launch_gram(ctx, m, k, A->p + a_off, lda, C->p + c_off, ldc);
}

// ============================================================================================
// THIN contractions against a snapshot block (round 3): one operand has at most 64 rows (sketches, modes, basis
// vectors), the other is the (M, dim) block -- HBM bound: the block should stream through ONCE at the rate of a copy.
// The 64 x 64 register-staged engine above reaches 2.4-3.0 TB/s on these shapes; the two kernels below take their
// operand chunks by LDS-DMA like k_gram128 (no staging registers, 128-row / 128-column workgroup tiles: twice the bytes
// in flight per instruction issued) and give a wave only the 16-row blocks of the thin operand that carry data
// (MI = ceil(m / 16): the fp64 MFMA rate would bound a 64-row padded product at ~4.3 TB/s).
// ============================================================================================
#define T_DMA(LDS_, BASE_, VOFF_)                                                                                   \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),    \
               "v"(VOFF_), "s"(BASE_)                                                                               \
               : "memory", "m0")

// ---- NT: part[z][m][n] = A[m, Kz] B[n, Kz]^T, m <= 64, K snapshot-long (split over z) --------------------------------
// Workgroup: all rows of A x 128 rows of B; wave w: B rows 32 w .. 32 w + 31, accumulators [MI][2].  Slot = {A: 64 rows x
// 128 B | B: 128 rows x 128 B}, a row's eight 16-byte units at position u ^ ((row >> 1) & 7) (k_gram128's layout); two
// slots, chunk ch + 1 in flight under the MFMAs of chunk ch.  Rows behind the operands are clamped (not stored).
constexpr int TN_A = 64 * 128;              // bytes
constexpr int TN_SLOT = TN_A + 128 * 128;   // 24,576 B; two slots: three workgroups per CU
template <int MI>
__global__ __launch_bounds__(256) void k_gemm_nt_thin(long long m, long long n, long long K, long long kper,
                                                      const double* __restrict__ A, long long lda,
                                                      const double* __restrict__ B, long long ldb, double* __restrict__ part) {
  __shared__ __align__(16) char lds[2 * TN_SLOT];
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds));
  const long long c0 = blockIdx.x * 128LL;
  const long long kbeg = blockIdx.z * kper, kend = std::min<long long>(K, kbeg + kper);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  d4_t acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  const int nfull = int((kend - kbeg) / BK), rem = int((kend - kbeg) % BK);
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kki = 0; kki < 4; ++kki) {
    const unsigned uo = unsigned((((2 * kki) ^ (kq >> 1) ^ (fr >> 1)) << 4) + (kq & 1) * 8);
    fa[kki] = unsigned(fr * 128) + uo;
    fb[kki] = unsigned(TN_A + (w * 32 + fr) * 128) + uo;
  }
#define TN_FRAGS(SLOT_, KKI_, AF_, BF_)                                                                    \
  do {                                                                                                     \
    const char* pa_ = lds + (SLOT_) * TN_SLOT + fa[KKI_];                                                  \
    const char* pb_ = lds + (SLOT_) * TN_SLOT + fb[KKI_];                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) AF_[i_] = *reinterpret_cast<const double*>(pa_ + i_ * 2048); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) BF_[j_] = *reinterpret_cast<const double*>(pb_ + j_ * 2048);  \
  } while (0)
  // DMA: 8 rows per instruction; wave w fetches A rows 16 w + 8 q (q < 2) and B rows 32 w + 8 q (q < 4)
  unsigned voA[2], voB[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rl = 16 * w + 8 * q + (lane >> 3);
    const long long ra = std::min<long long>(rl, m - 1);
    voA[q] = unsigned(ra * lda * 8) + unsigned(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = 32 * w + 8 * q + (lane >> 3);
    const long long rb = std::min<long long>(rl, n - 1 - c0);
    voB[q] = unsigned(rb * ldb * 8) + unsigned(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
  }
  const char* sA = reinterpret_cast<const char*>(A + kbeg);
  const char* sB = reinterpret_cast<const char*>(B + c0 * ldb + kbeg);
#define TN_ISSUE(SLOT_)                                                                                      \
  do {                                                                                                       \
    const unsigned sl_ = lds0 + unsigned(SLOT_) * TN_SLOT;                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) T_DMA(sl_ + unsigned(16 * w + 8 * q_) * 128u, sA, voA[q_]);        \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) T_DMA(sl_ + TN_A + unsigned(32 * w + 8 * q_) * 128u, sB, voB[q_]); \
    sA += BK * 8;                                                                                            \
    sB += BK * 8;                                                                                            \
  } while (0)
  if (nfull > 0) {
    TN_ISSUE(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    double af[2][MI], bf[2][2];
    TN_FRAGS(0, 0, af[0], bf[0]);
    for (int ch = 0; ch < nfull; ++ch) {
      const int slot = ch & 1;
      if (ch + 1 < nfull) TN_ISSUE(slot ^ 1);  // (everybody left that slot at the barrier behind chunk ch - 1)
#pragma unroll
      for (int kki = 0; kki < 4; ++kki) {
        const int pb = kki & 1;
        if (kki < 3) TN_FRAGS(slot, kki + 1, af[pb ^ 1], bf[pb ^ 1]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][i], bf[pb][j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // chunk ch + 1 is in LDS for everybody
      if (ch + 1 < nfull) TN_FRAGS(slot ^ 1, 0, af[0], bf[0]);
    }
  }
#undef TN_ISSUE
#undef TN_FRAGS
  if (rem > 0) {
    // the last columns of K (< 16), zero padded to one chunk, through registers: [A 64 x LDK | B 128 x LDK] doubles
    double* st = reinterpret_cast<double*>(lds);
    const int sk = threadIdx.x & 15, sr0 = threadIdx.x >> 4;
    const long long k = kbeg + (long long)nfull * BK + sk;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const long long ra = sr0 + 16 * x;
      st[ra * LDK + sk] = (ra < m && k < kend) ? A[ra * lda + k] : 0.0;
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const long long rb = c0 + sr0 + 16 * x;
      st[64 * LDK + (sr0 + 16 * x) * LDK + sk] = (rb < n && k < kend) ? B[rb * ldb + k] : 0.0;
    }
    __syncthreads();
    const double* pa = st + fr * LDK + kq;
    const double* pb = st + 64 * LDK + (w * 32 + fr) * LDK + kq;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[MI], bf[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = pa[i * 16 * LDK + kk];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = pb[j * 16 * LDK + kk];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long long r = i * 16 + (lane >> 4) + 4 * g;
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long long c = c0 + w * 32 + j * 16 + (lane & 15);
        if (c < n) part[(blockIdx.z * m + r) * n + c] = acc[i][j][g];
      }
    }
}

// ---- NN: C[m, n0 .. n0 + 128) = alpha A[m, K] B[K, n] + beta C, m <= 64, n snapshot-long (>= 128).  The last tile is
// shifted left to end at column n (no column of B beyond n is touched) and stores only the columns no other tile owns ----
// Slot = {A: 64 rows x 16 k (the layout above) | B: 16 k-rows x 128 columns, a k-row = ONE DMA instruction (1 KB), rows
// 1152 B apart so that the four k-rows a fragment read touches fall into different bank halves}.  Wave w: columns
// 32 w .. 32 w + 31, accumulators [MI][2].  The last k-rows (K % 16) go through registers, zero padded.
constexpr int TNN_BROW = 1024 + 128;               // bytes between the k-rows of the B part
constexpr int TNN_SLOT = TN_A + 16 * TNN_BROW;     // 26,624 B; two slots: three workgroups per CU
template <int MI>
__global__ __launch_bounds__(256) void k_gemm_nn_thin(long long m, long long n, long long K, double alpha,
                                                      const double* __restrict__ A, long long lda, const double* __restrict__ B,
                                                      long long ldb, double beta, double* __restrict__ C, long long ldc) {
  __shared__ __align__(16) char lds[2 * TNN_SLOT];
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds));
  const long long c_own = blockIdx.x * 128LL;                  // first column this workgroup stores
  const long long c0 = std::min<long long>(c_own, n - 128);    // first column of its tile
  // (grid.y > 1: A has more than 64 rows and a short K -- the lift  coefficients x basis : row tile blockIdx.y)
  const long long r0 = blockIdx.y * 64LL;
  A += r0 * lda;
  C += r0 * ldc;
  m = std::min<long long>(64, m - r0);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  d4_t acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  const int nfull = int(K / BK), rem = int(K % BK);
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kki = 0; kki < 4; ++kki) {
    fa[kki] = unsigned(fr * 128) + unsigned((((2 * kki) ^ (kq >> 1) ^ (fr >> 1)) << 4) + (kq & 1) * 8);
    fb[kki] = unsigned(TN_A + (4 * kki + kq) * TNN_BROW + (w * 32 + fr) * 8);
  }
#define TNN_FRAGS(SLOT_, KKI_, AF_, BF_)                                                                   \
  do {                                                                                                     \
    const char* pa_ = lds + (SLOT_) * TNN_SLOT + fa[KKI_];                                                 \
    const char* pb_ = lds + (SLOT_) * TNN_SLOT + fb[KKI_];                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) AF_[i_] = *reinterpret_cast<const double*>(pa_ + i_ * 2048); \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) BF_[j_] = *reinterpret_cast<const double*>(pb_ + j_ * 128);   \
  } while (0)
  unsigned voA[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rl = 16 * w + 8 * q + (lane >> 3);
    const long long ra = std::min<long long>(rl, m - 1);
    voA[q] = unsigned(ra * lda * 8) + unsigned(((lane & 7) ^ ((rl >> 1) & 7)) * 16);
  }
  const unsigned voB = unsigned(lane) * 16u;  // lane -> columns 2 lane, 2 lane + 1 of the tile
  const char* sA = reinterpret_cast<const char*>(A);
  const char* sB = reinterpret_cast<const char*>(B + c0);
  const size_t brow = size_t(ldb) * 8;
#define TNN_ISSUE(SLOT_)                                                                                     \
  do {                                                                                                       \
    const unsigned sl_ = lds0 + unsigned(SLOT_) * TNN_SLOT;                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) T_DMA(sl_ + unsigned(16 * w + 8 * q_) * 128u, sA, voA[q_]);  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                         \
        T_DMA(sl_ + TN_A + unsigned(4 * w + q_) * unsigned(TNN_BROW), sB + size_t(4 * w + q_) * brow, voB);  \
    sA += BK * 8;                                                                                            \
    sB += BK * brow;                                                                                         \
  } while (0)
  if (nfull > 0) {
    TNN_ISSUE(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    double af[2][MI], bf[2][2];
    TNN_FRAGS(0, 0, af[0], bf[0]);
    for (int ch = 0; ch < nfull; ++ch) {
      const int slot = ch & 1;
      if (ch + 1 < nfull) TNN_ISSUE(slot ^ 1);
#pragma unroll
      for (int kki = 0; kki < 4; ++kki) {
        const int pb = kki & 1;
        if (kki < 3) TNN_FRAGS(slot, kki + 1, af[pb ^ 1], bf[pb ^ 1]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb][i], bf[pb][j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (ch + 1 < nfull) TNN_FRAGS(slot ^ 1, 0, af[0], bf[0]);
    }
  }
#undef TNN_ISSUE
#undef TNN_FRAGS
  if (rem > 0) {
    // [A 64 x LDK doubles | B 16 x 144 doubles]: the last k-rows, zero padded to one chunk
    double* st = reinterpret_cast<double*>(lds);
    double* stb = st + 64 * LDK;
    const long long k0 = (long long)nfull * BK;
    {
      const int sk = threadIdx.x & 15, sr0 = threadIdx.x >> 4;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const long long ra = sr0 + 16 * x;
        st[ra * LDK + sk] = (ra < m && sk < rem) ? A[ra * lda + k0 + sk] : 0.0;
      }
      const int cc = threadIdx.x & 127, kr0 = threadIdx.x >> 7;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int kr = kr0 + 2 * x;
        stb[kr * 144 + cc] = kr < rem ? B[(k0 + kr) * ldb + c0 + cc] : 0.0;
      }
    }
    __syncthreads();
    const double* pa = st + fr * LDK + kq;
    const double* pb = stb + kq * 144 + w * 32 + fr;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double af[MI], bf[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = pa[i * 16 * LDK + kk];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = pb[kk * 144 + j * 16];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long long r = i * 16 + (lane >> 4) + 4 * g;
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long long c = c0 + w * 32 + j * 16 + (lane & 15);
        if (c < c_own) continue;
        double* p = C + r * ldc + c;
        *p = beta == 0.0 ? alpha * acc[i][j][g] : alpha * acc[i][j][g] + beta * *p;
      }
    }
}
#undef T_DMA

// ============================================================================================
// NN GEMM: C[m,n] = alpha * sum_k A[m,k] B[k,n] + beta C   (k small: the lift  c . basis)
// ============================================================================================
// grid.z > 1: split-K -- slice z takes the K range [z kper, (z + 1) kper) and writes its raw partial tile to `part`
// ([z][m][n], ld n) for the deterministic reduction kernel (a short, wide product with a long K: a handful of tiles)
__global__ __launch_bounds__(256) void k_gemm_nn(long long m, long long n, long long K, double alpha,
                                                 const double* __restrict__ A, long long lda,
                                                 const double* __restrict__ B, long long ldb, double beta,
                                                 double* __restrict__ C, long long ldc, long long kper,
                                                 double* __restrict__ part) {
  __shared__ __align__(16) double stage[2 * STAGE_DOUBLES];
  double* sA = stage;
  double* sB = stage + STAGE_DOUBLES;
  const WavePos wp;
  const long long r0 = blockIdx.y * 64LL, c0 = blockIdx.x * 64LL;
  const int t = threadIdx.x;
  const int srow = stage_row(), sseg = stage_seg();
  const double* Ar = (r0 + srow < m) ? A + (r0 + srow) * lda : nullptr;
  // B staging: lane -> column (t & 63), thread t handles k rows (t >> 6) + 4 x: every wave-instruction
  // reads 512 contiguous bytes of one row of B
  const int bc = t & 63, bk0 = t >> 6;
  Acc acc;
  acc_zero(acc);
  const long long kbeg = part ? blockIdx.z * kper : 0;
  if (part) K = min(K, kbeg + kper);
  for (long long k0 = kbeg; k0 < K; k0 += BK) {
    double va[4], vb[4];
    load4_row<false>(Ar, k0 + sseg, K, va);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const long long kk = k0 + bk0 + 4 * x, c = c0 + bc;
      vb[x] = (kk < K && c < n) ? B[kk * ldb + c] : 0.0;
    }
    __syncthreads();  // previous chunk fully consumed
    stage_store(sA, va);
#pragma unroll
    for (int x = 0; x < 4; ++x) sB[bc * LDK + bk0 + 4 * x] = vb[x];
    __syncthreads();
    mma_chunk(sA, sB, acc, wp);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      long long r = r0 + acc_row(wp, i, g);
      if (r >= m) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        long long c = c0 + acc_col(wp, j);
        if (c >= n) continue;
        double v = acc.c[i][j][g];
        if (part) {
          part[(blockIdx.z * m + r) * n + c] = v;
          continue;
        }
        double* p = C + r * ldc + c;
        *p = beta == 0.0 ? alpha * v : alpha * v + beta * *p;
      }
    }
}

int rom_launch_gemm_nn(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  if (m <= 0 || n <= 0) return ROM_OK;
  static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-shape names in the profile records
  if ((m <= 64 ? k >= 128 : (k >= 16 && k <= 256)) && n >= 1024 && size_t(lda) * 8 * 64 < (size_t(1) << 32) &&
      size_t(ldb) * 8 * 16 + 1024 < (size_t(1) << 32) && (m + 63) / 64 <= 65535 && !no_thin_gemm()) {
    // thin A against the rows of a snapshot-wide B; or a tall A with a short K (the lift: output bound), in row tiles of 64
    char nm[64];
    detail ? snprintf(nm, sizeof nm, "gemm_nn_thin_%lldx%lldx%lld", (long long)m, (long long)n, (long long)k) : snprintf(nm, sizeof nm, "gemm_nn");
    ROM_PROF(ctx, nm, 2.0 * m * n * k, 8.0 * (double(m) * k + double(n) * k + double(m) * n));
    const dim3 grid{unsigned((n + 127) / 128), unsigned((m + 63) / 64)};
    switch (m > 64 ? 4 : (m + 15) / 16) {
      case 1: k_gemm_nn_thin<1><<<grid, 256, 0, ctx->stream>>>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); break;
      case 2: k_gemm_nn_thin<2><<<grid, 256, 0, ctx->stream>>>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); break;
      case 3: k_gemm_nn_thin<3><<<grid, 256, 0, ctx->stream>>>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); break;
      default: k_gemm_nn_thin<4><<<grid, 256, 0, ctx->stream>>>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); break;
    }
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  dim3 grid(unsigned((n + 63) / 64), unsigned((m + 63) / 64));
  // a handful of output tiles under a long K (the sketches of a FACTORED block: 24 x 272 x 1024): split-K, 64 rows of B each
  const long long tiles = (long long)grid.x * grid.y;
  long long splits = 1, kper = k;
  if (tiles <= 32 && k >= 256) {
    splits = std::min<long long>((256 + tiles - 1) / tiles, k / 64);
    kper = ((k + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (k + kper - 1) / kper;
  }
  double* part = nullptr;
  if (splits > 1) {
    ROM_TRY(rom_ctx_scratch(ctx, size_t(splits) * m * n, &part));
    grid.z = unsigned(splits);
  }
  {
    char nm[64];
    detail ? snprintf(nm, sizeof nm, "gemm_nn_%lldx%lldx%lld", (long long)m, (long long)n, (long long)k) : snprintf(nm, sizeof nm, "gemm_nn");
    ROM_PROF(ctx, nm, 2.0 * m * n * k, 8.0 * (double(m) * k + double(n) * k + double(m) * n));
    k_gemm_nn<<<grid, 256, 0, ctx->stream>>>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, kper, part);
  }
  ROM_HIP(hipGetLastError());
  if (splits > 1) {
    ROM_PROF(ctx, "splitk_reduce", double(splits) * m * n, 8.0 * double(splits + 1) * m * n);
    k_splitk_reduce<<<unsigned((m * n + 255) / 256), 256, 0, ctx->stream>>>(m, n, int(splits), alpha, part, beta, C, ldc, 0);
    ROM_HIP(hipGetLastError());
  }
  return ROM_OK;
}

extern "C" int rom_gemm_nn(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, rom_buf* A, size_t a_off,
                           int64_t lda, rom_buf* B, size_t b_off, int64_t ldb, double beta, rom_buf* C,
                           size_t c_off, int64_t ldc) {
  ROM_CHECK(ctx && A && B && C, "rom_gemm_nn: null argument");
  ROM_CHECK(m >= 0 && n >= 0 && k >= 0 && lda >= k && ldb >= n && ldc >= n, "rom_gemm_nn: bad dimensions");
  if (m == 0 || n == 0) return ROM_OK;
  ROM_CHECK(a_off + size_t(m - 1) * lda + k <= A->n, "rom_gemm_nn: A out of range");
  ROM_CHECK(k == 0 || b_off + size_t(k - 1) * ldb + n <= B->n, "rom_gemm_nn: B out of range");
  ROM_CHECK(c_off + size_t(m - 1) * ldc + n <= C->n, "rom_gemm_nn: C out of range");
  return rom_launch_gemm_nn(ctx, m, n, k, alpha, A->p + a_off, lda, B->p + b_off, ldb, beta, C->p + c_off, ldc);
}

// ============================================================================================
// stencil application and norms
// ============================================================================================
// Y = A(coef) X for the 5-point operator with block coefficients (kappa of the four cells around a vertex: SURVEY 8a-1).
// Every entry is loaded ONCE: a thread owns a mesh column and walks down a slab of rows with the entry above, the entry
// itself and the entry below in registers; the east / west neighbours come from the next / previous lane (the wave's edge
// lanes load theirs: same cache lines).  (The first version read five entries per output and moved 3x the bytes through
// the fabric -- FETCH_SIZE x 2 = 1.62 GB for a 533 MB block, profiles/r02_hbm_kernels_c2_pmc_traffic.json -- at
// 2.3 TB/s; this one runs at 4.3-4.8 TB/s.)
constexpr int SA_ROWS = 32;    // rows per slab (+ one halo row above and below: 6 % extra reads)
constexpr int SA_UNROLL = 8;   // loads in flight per thread
template <bool UNIT>
__global__ __launch_bounds__(256) void k_stencil_apply_cols(StencilGeom g, const double* __restrict__ a_one, int coef_stride,
                                                            const double* __restrict__ X, long long x_stride,
                                                            double* __restrict__ Y, int slab0) {
  // vector z of the launch: input row X + z x_stride (0: the same row for every z), coefficients a_one + z coef_stride
  __shared__ double am[64];
  if (!UNIT)
    for (int i = threadIdx.x; i < g.kblk; i += blockDim.x) am[i] = a_one[blockIdx.z * (long long)coef_stride + i];
  __syncthreads();
  const double* x = X + blockIdx.z * x_stride;
  double* y = Y + blockIdx.z * g.dim;
  const int c = blockIdx.x * 256 + threadIdx.x;  // 0-based interior column
  const int r0 = (slab0 + blockIdx.y) * SA_ROWS, r1 = min(g.nr, r0 + SA_ROWS);   // (slab0: a launch over a band of mesh rows)
  const bool in = c < g.nc;
  const int lane = threadIdx.x & 63;
  const int N = g.N;
  // block columns of the cells left / right of vertex column c + 1 (1-based): (c) / N and (c + 1) / N
  const int qw = c / N, qe = (c + 1) / N;
  auto at = [&](int r, int cc) -> double { return x[(long long)r * g.nc + cc]; };
  double xn = (in && r0 > 0) ? at(r0 - 1, c) : 0.0;  // entry above the slab (boundary: 0)
  double xc = (in && r0 < r1) ? at(r0, c) : 0.0;
  for (int rb = r0; rb < r1; rb += SA_UNROLL) {
    double xs[SA_UNROLL], xe_edge[SA_UNROLL], xw_edge[SA_UNROLL];
#pragma unroll
    for (int q = 0; q < SA_UNROLL; ++q) {
      const int r = rb + q;
      xs[q] = (in && r < r1 && r + 1 < g.nr) ? at(r + 1, c) : 0.0;
      xe_edge[q] = (lane == 63 && r < r1 && c + 1 < g.nc) ? at(r, c + 1) : 0.0;
      xw_edge[q] = (lane == 0 && r < r1 && in && c > 0) ? at(r, c - 1) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < SA_UNROLL; ++q) {
      const int r = rb + q;
      double xe = __shfl_down(xc, 1, 64), xw = __shfl_up(xc, 1, 64);
      if (lane == 63) xe = xe_edge[q];
      if (lane == 0) xw = xw_edge[q];
      if (in && r < r1) {
        double k00 = 1.0, k01 = 1.0, k10 = 1.0, k11 = 1.0;
        if (!UNIT) {
          const int pn = r / N, ps = (r + 1) / N;  // block rows of the cells above / below vertex row r + 1 (1-based)
          k00 = am[pn * g.ncb + qw];
          k01 = am[pn * g.ncb + qe];
          k10 = am[ps * g.ncb + qw];
          k11 = am[ps * g.ncb + qe];
        }
        double v = (((k00 + k01) + k10) + k11) * xc;
        if (c + 1 < g.nc) v += (-(k11 + k01) / 2) * xe;
        if (c > 0) v += (-(k10 + k00) / 2) * xw;
        if (r + 1 < g.nr) v += (-(k11 + k10) / 2) * xs[q];
        if (r > 0) v += (-(k01 + k00) / 2) * xn;
        y[(long long)r * g.nc + c] = v;
      }
      xn = xc;
      xc = xs[q];
    }
  }
}

__device__ inline double block_reduce_sum(double v) {
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial[k][blk] = this block's share of d^T A_1 d,  d = u - v (or u), in edge form: with zero boundary values
//   d^T A_1 d = sum over horizontal mesh edges (d_w - d_e)^2 + sum over vertical mesh edges (d_n - d_s)^2
// (A_1 = the unit-coefficient 5-point operator; the same number as d . (A_1 d) up to rounding, all terms >= 0).
// A thread owns one mesh column and walks down a slab of rows: every entry is loaded exactly once (plus the row
// above the slab), the east neighbour comes from the next lane, the north neighbour from the previous iteration.
constexpr int H10_ROWS = 32;   // rows per slab
constexpr int H10_UNROLL = 8;  // loads in flight per thread
template <bool DIFF>
__global__ __launch_bounds__(256) void k_h10_partial(StencilGeom g, const double* __restrict__ U,
                                                     const double* __restrict__ V, double* __restrict__ partial,
                                                     int nblk) {
  const double* u = U + blockIdx.z * g.dim;
  const double* v = DIFF ? V + blockIdx.z * g.dim : nullptr;
  const int c = blockIdx.x * 256 + threadIdx.x;  // 0-based interior column
  const int r0 = blockIdx.y * H10_ROWS, r1 = min(g.nr, r0 + H10_ROWS);
  const bool in = c < g.nc;
  const int lane = threadIdx.x & 63;
  auto at = [&](int r, int cc) -> double {
    const long long i = (long long)r * g.nc + cc;
    return DIFF ? u[i] - v[i] : u[i];
  };
  double s = 0.0;
  double north = (in && r0 > 0) ? at(r0 - 1, c) : 0.0;  // value above the slab (boundary: 0)
  for (int rb = r0; rb < r1; rb += H10_UNROLL) {
    double x[H10_UNROLL], xl[H10_UNROLL];
#pragma unroll
    for (int q = 0; q < H10_UNROLL; ++q) {
      const int r = rb + q;
      x[q] = (in && r < r1) ? at(r, c) : 0.0;
      // the wave's last lane has its east neighbour in another wave: it loads it itself (same cache line)
      xl[q] = (lane == 63 && c + 1 < g.nc && r < r1) ? at(r, c + 1) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < H10_UNROLL; ++q) {
      const int r = rb + q;
      double east = __shfl_down(x[q], 1, 64);
      if (lane == 63) east = xl[q];
      if (in && r < r1) {
        if (c + 1 >= g.nc) east = 0.0;  // boundary
        const double dh = x[q] - east, dv = x[q] - north;
        s += dh * dh + dv * dv;
        if (c == 0) s += x[q] * x[q];              // edge to the west boundary
        if (r == g.nr - 1) s += x[q] * x[q];       // edge to the south boundary
        north = x[q];
      }
    }
  }
  s = block_reduce_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.z * (long long)nblk + blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_sq_partial(long long dim, const double* __restrict__ U,
                                                    double* __restrict__ partial, int nblk, int per_thread) {
  const double* u = U + blockIdx.y * dim;
  double s = 0.0;
  long long base = blockIdx.x * (long long)(256 * per_thread);
  for (int it = 0; it < per_thread; ++it) {
    long long idx = base + it * 256 + threadIdx.x;
    if (idx < dim) s += u[idx] * u[idx];
  }
  s = block_reduce_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.y * (long long)nblk + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_finish_norm(const double* __restrict__ partial, int nblk,
                                                     double* __restrict__ out, int take_sqrt) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[blockIdx.x * (long long)nblk + i];
  s = block_reduce_sum(s);
  if (threadIdx.x == 0) out[blockIdx.x] = take_sqrt ? sqrt(s) : s;
}

// partial[k][blk] = this block's share of U_k . z
__global__ __launch_bounds__(256) void k_rowdot_partial(long long dim, const double* __restrict__ U,
                                                        const double* __restrict__ z, double* __restrict__ partial,
                                                        int nblk, int per_thread) {
  const double* u = U + blockIdx.y * dim;
  double s = 0.0;
  long long base = blockIdx.x * (long long)(256 * per_thread);
  for (int it = 0; it < per_thread; ++it) {
    long long idx = base + it * 256 + threadIdx.x;
    if (idx < dim) s += u[idx] * z[idx];
  }
  s = block_reduce_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.y * (long long)nblk + blockIdx.x] = s;
}

StencilGeom rom_make_geom(int nrb, int ncb, int N) {
  StencilGeom g;
  g.N = N;
  g.ncb = ncb;
  g.kblk = nrb * ncb;
  g.nr = nrb * N - 1;
  g.nc = ncb * N - 1;
  g.dim = (long long)g.nr * g.nc;
  return g;
}

int rom_launch_stencil_apply(rom_fem* f, const double* d_coef, const double* X, int K, double* Y) {
  if (K <= 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  StencilGeom g = rom_make_geom(f->nrb, f->ncb, f->N);
  ROM_CHECK(K <= 65535, "stencil apply: at most 65535 vectors per call");
  {
    ROM_PROF(ctx, "stencil_apply", 14.0 * g.dim * K, 16.0 * g.dim * K);
    const dim3 grid((g.nc + 255) / 256, (g.nr + SA_ROWS - 1) / SA_ROWS, K);
    if (!d_coef) k_stencil_apply_cols<true><<<grid, 256, 0, ctx->stream>>>(g, nullptr, 0, X, g.dim, Y, 0);
    else k_stencil_apply_cols<false><<<grid, 256, 0, ctx->stream>>>(g, d_coef, 0, X, g.dim, Y, 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// The same on the mesh rows [row_lo, row_hi] only (0-based interior rows; whole slabs of SA_ROWS that cover them): entries of Y
// outside the band are NOT written.  For operators that vanish outside a band -- the one-hot block operators A_b.
int rom_launch_stencil_apply_band(rom_fem* f, const double* d_coef, const double* X, int K, double* Y, int row_lo, int row_hi) {
  if (K <= 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  StencilGeom g = rom_make_geom(f->nrb, f->ncb, f->N);
  ROM_CHECK(K <= 65535 && d_coef && row_lo >= 0 && row_hi >= row_lo && row_hi < g.nr, "stencil apply on a band: bad arguments");
  const int s0 = row_lo / SA_ROWS, s1 = row_hi / SA_ROWS;
  {
    ROM_PROF(ctx, "stencil_apply", 14.0 * double(s1 - s0 + 1) * SA_ROWS * g.nc * K, 16.0 * double(s1 - s0 + 1) * SA_ROWS * g.nc * K);
    const dim3 grid((g.nc + 255) / 256, unsigned(s1 - s0 + 1), K);
    k_stencil_apply_cols<false><<<grid, 256, 0, ctx->stream>>>(g, d_coef, 0, X, g.dim, Y, s0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// Y[b, :] = A_b x for every block b (one-hot coefficients, `d_onehot`: kblk x kblk identity on the device): the k
// stencil applications that grow the reduced tensor by one basis vector, in one launch
int rom_launch_stencil_apply_blocks(rom_fem* f, const double* d_onehot, const double* x, double* Y) {
  rom_ctx* ctx = f->ctx;
  StencilGeom g = rom_make_geom(f->nrb, f->ncb, f->N);
  {
    ROM_PROF(ctx, "stencil_apply_blocks", 14.0 * g.dim * g.kblk, 16.0 * g.dim * g.kblk);
    const dim3 grid((g.nc + 255) / 256, (g.nr + SA_ROWS - 1) / SA_ROWS, g.kblk);
    k_stencil_apply_cols<false><<<grid, 256, 0, ctx->stream>>>(g, d_onehot, g.kblk, x, 0, Y, 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_stencil_apply(rom_fem* f, const double* a_one_host, int unit, rom_buf* X, int64_t x_row0,
                                 int K, rom_buf* Y, int64_t y_row0) {
  ROM_CHECK(f && X && Y, "rom_stencil_apply: null argument");
  ROM_CHECK(unit || a_one_host, "rom_stencil_apply: coefficient required when unit == 0");
  ROM_CHECK(K >= 0 && x_row0 >= 0 && y_row0 >= 0, "rom_stencil_apply: negative size");
  ROM_CHECK(size_t(x_row0 + K) * f->dim <= X->n && size_t(y_row0 + K) * f->dim <= Y->n,
            "rom_stencil_apply: rows out of range");
  if (K == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  double* d_a = nullptr;
  if (!unit) {
    ROM_TRY(rom_ctx_scratch(ctx, 64, &d_a));
    ROM_HIP(hipMemcpyAsync(d_a, a_one_host, f->nrb * f->ncb * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  ROM_TRY(rom_launch_stencil_apply(f, d_a, X->p + x_row0 * f->dim, K, Y->p + y_row0 * f->dim));
  if (!unit) ROM_HIP(hipStreamSynchronize(ctx->stream));  // scratch may be re-used by the next call
  return ROM_OK;
}

int rom_launch_h10norm(rom_fem* f, const double* U, const double* V, int K, double* d_out, bool take_sqrt) {
  if (K <= 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  StencilGeom g = rom_make_geom(f->nrb, f->ncb, f->N);
  const dim3 grid((g.nc + 255) / 256, (g.nr + H10_ROWS - 1) / H10_ROWS, K);
  const int nblk = int(grid.x * grid.y);
  ROM_CHECK(K <= 65535, "H10 norm: at most 65535 vectors per call");
  double* scratch = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(K) * nblk, &scratch));
  {
    ROM_PROF(ctx, "h10norm", 10.0 * g.dim * K, (V ? 16.0 : 8.0) * g.dim * K);
    if (V) k_h10_partial<true><<<grid, 256, 0, ctx->stream>>>(g, U, V, scratch, nblk);
    else k_h10_partial<false><<<grid, 256, 0, ctx->stream>>>(g, U, nullptr, scratch, nblk);
    k_finish_norm<<<K, 256, 0, ctx->stream>>>(scratch, nblk, d_out, take_sqrt ? 1 : 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_h10norm(rom_fem* f, rom_buf* U, int64_t u_row0, rom_buf* V, int64_t v_row0, int K,
                           double* out_host) {
  ROM_CHECK(f && U && (out_host || K == 0), "rom_h10norm: null argument");
  ROM_CHECK(K >= 0 && u_row0 >= 0 && v_row0 >= 0, "rom_h10norm: negative size");
  ROM_CHECK(size_t(u_row0 + K) * f->dim <= U->n, "rom_h10norm: U rows out of range");
  ROM_CHECK(!V || size_t(v_row0 + K) * f->dim <= V->n, "rom_h10norm: V rows out of range");
  if (K == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  rom_buf* out = nullptr;
  ROM_TRY(rom_buf_alloc(ctx, size_t(K), &out));
  int st = rom_launch_h10norm(f, U->p + u_row0 * f->dim, V ? V->p + v_row0 * f->dim : nullptr, K, out->p, true);
  if (st == ROM_OK) st = rom_buf_download(out, 0, out_host, size_t(K));
  rom_buf_free(out);
  return st;
}

int rom_launch_l2norm(rom_ctx* ctx, const double* U, int K, int64_t dim, double* d_out, bool take_sqrt) {
  if (K <= 0) return ROM_OK;
  const int per_thread = 8;
  const int nblk = int(std::max<int64_t>(1, (dim + 256 * per_thread - 1) / (256 * per_thread)));
  ROM_CHECK(K <= 65535, "l2 norm: at most 65535 vectors per call");
  double* scratch = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(K) * nblk, &scratch));
  {
    ROM_PROF(ctx, "l2norm", 2.0 * dim * K, 8.0 * dim * K);
    k_sq_partial<<<dim3(nblk, K), 256, 0, ctx->stream>>>(dim, U, scratch, nblk, per_thread);
    k_finish_norm<<<K, 256, 0, ctx->stream>>>(scratch, nblk, d_out, take_sqrt ? 1 : 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

int rom_launch_rowdot(rom_ctx* ctx, const double* U, int K, int64_t dim, const double* z, double* d_out) {
  if (K <= 0) return ROM_OK;
  const int per_thread = 8;
  const int nblk = int(std::max<int64_t>(1, (dim + 256 * per_thread - 1) / (256 * per_thread)));
  ROM_CHECK(K <= 65535, "row dot: at most 65535 vectors per call");
  double* scratch = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(K) * nblk, &scratch));
  {
    ROM_PROF(ctx, "rowdot", 2.0 * dim * K, 8.0 * dim * K);
    k_rowdot_partial<<<dim3(nblk, K), 256, 0, ctx->stream>>>(dim, U, z, scratch, nblk, per_thread);
    k_finish_norm<<<K, 256, 0, ctx->stream>>>(scratch, nblk, d_out, 0);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_l2norm(rom_ctx* ctx, rom_buf* U, int64_t row0, int K, int64_t dim, double* out_host) {
  ROM_CHECK(ctx && U && (out_host || K == 0), "rom_l2norm: null argument");
  ROM_CHECK(K >= 0 && row0 >= 0 && dim >= 0, "rom_l2norm: negative size");
  ROM_CHECK(size_t(row0 + K) * dim <= U->n, "rom_l2norm: rows out of range");
  if (K == 0) return ROM_OK;
  rom_buf* out = nullptr;
  ROM_TRY(rom_buf_alloc(ctx, size_t(K), &out));
  int st = rom_launch_l2norm(ctx, U->p + row0 * dim, K, dim, out->p, true);
  if (st == ROM_OK) st = rom_buf_download(out, 0, out_host, size_t(K));
  rom_buf_free(out);
  return st;
}

// ============================================================================================
// batched reduced SPD solves: one workgroup per system.  The n x n matrix lives in LDS while it fits (n <= 140 with
// the 160 KB of a gfx950 CU opted in; n <= 88 inside the default 64 KB), otherwise in a per-system slab of global
// memory (L2 resident: 200 x 200 doubles = 320 KB per system) -- the reference's galerkin() takes any n
// (src/lib/SolutionsManagers.py:17-40).  Same elimination order either way.
// ============================================================================================
constexpr int REDUCED_LDS_DEFAULT = 64 * 1024, REDUCED_LDS_MAX = 160 * 1024;

__global__ __launch_bounds__(256) void k_reduced_solve(int n, int ldA, int kb, const double* __restrict__ Ahat,
                                                       const double* __restrict__ w, const double* __restrict__ rhs,
                                                       int rhs_per_system, double* __restrict__ cout, int* status,
                                                       double* gws) {
  extern __shared__ __align__(16) double sm[];  // [n*(n+1) matrix, if it lives here] + n rhs + n diag
  const int ld = n + 1;
  const int m = blockIdx.x, t = threadIdx.x;
  double* Am = gws ? gws + size_t(m) * n * ld : sm;
  double* b = gws ? sm : sm + size_t(n) * ld;
  double* sd = b + n;
  // A = sum_b w[m,b] Ahat[b]   (the reference's einsum('pqij,pq->ij') on the reduced tensor)
  for (int idx = t; idx < n * n; idx += 256) {
    int r = idx / n, c = idx % n;
    double s = 0.0;
    for (int q = 0; q < kb; ++q) s += w[size_t(m) * kb + q] * Ahat[(size_t(q) * ldA + r) * ldA + c];
    Am[r * ld + c] = s;
  }
  for (int i = t; i < n; i += 256) b[i] = rhs_per_system ? rhs[size_t(m) * n + i] : rhs[i];
  __syncthreads();
  // right-looking elimination (LDL^T form), lower triangle
  for (int j = 0; j < n - 1; ++j) {
    double invd = 1.0 / Am[j * ld + j];
    for (int idx = t; idx < (n - j - 1) * (n - j - 1); idx += 256) {
      int r = j + 1 + idx / (n - j - 1), c = j + 1 + idx % (n - j - 1);
      if (c <= r) Am[r * ld + c] -= Am[r * ld + j] * invd * Am[c * ld + j];
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 256) {
    double d = Am[i * ld + i];
    if (!(d > 0.0)) atomicOr(status, 1);
    sd[i] = sqrt(d);
  }
  __syncthreads();
  // one wave finishes: L = Am / sd ; forward + backward substitution (n small)
  if (t < 64) {
    // forward: y_r = (b_r - sum_{k<r} L_rk y_k) / L_rr, column sweep so lanes work on rows r > k
    for (int k = 0; k < n; ++k) {
      double yk = b[k] / sd[k];
      __builtin_amdgcn_wave_barrier();
      if (t == 0) b[k] = yk;
      for (int r = k + 1 + t; r < n; r += 64) b[r] -= (Am[r * ld + k] / sd[k]) * yk;
      __builtin_amdgcn_wave_barrier();
    }
    // backward: x_k = (y_k - sum_{r>k} L_rk x_r) / L_kk
    for (int k = n - 1; k >= 0; --k) {
      double xk = b[k] / sd[k];
      __builtin_amdgcn_wave_barrier();
      if (t == 0) b[k] = xk;
      for (int c = t; c < k; c += 64) b[c] -= (Am[k * ld + c] / sd[c]) * xk;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int i = t; i < n; i += 256) cout[size_t(m) * n + i] = b[i];
}

int rom_launch_reduced_solve(rom_ctx* ctx, int n, int ldA, int kb, int M, const double* Ahat, const double* w,
                             const double* rhs, int rhs_per_system, double* c_out) {
  if (M <= 0) return ROM_OK;
  const size_t mat = size_t(n) * (n + 1) * sizeof(double), vecs = 2 * size_t(n) * sizeof(double);
  const bool in_lds = mat + vecs <= size_t(REDUCED_LDS_MAX);
  const size_t lds = in_lds ? mat + vecs : vecs;
  if (lds > size_t(REDUCED_LDS_DEFAULT)) {
    if (!ctx->lds_optin_reduced_solve) {  // once per device: the attribute belongs to the kernel as loaded on THIS device
      ROM_HIP(hipSetDevice(ctx->device));
      ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_reduced_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  REDUCED_LDS_MAX));
      ctx->lds_optin_reduced_solve = true;
    }
  }
  // matrices beyond the LDS: slabs of the scratch block, as many systems per launch as 1 GiB of it holds
  const int per_launch = in_lds ? M : int(std::max<size_t>(1, std::min<size_t>(size_t(M), (size_t(1) << 30) / mat)));
  double* gws = nullptr;
  if (!in_lds) ROM_TRY(rom_ctx_scratch(ctx, size_t(per_launch) * n * (n + 1), &gws));
  for (int m0 = 0; m0 < M; m0 += per_launch) {
    const int Mc = std::min(per_launch, M - m0);
    ROM_PROF(ctx, "reduced_solve", Mc * (double(n) * n * n / 3 + 2.0 * kb * n * n), 8.0 * Mc * (kb + 2.0 * n));
    k_reduced_solve<<<Mc, 256, lds, ctx->stream>>>(n, ldA, kb, Ahat, w + size_t(m0) * kb,
                                                   rhs + (rhs_per_system ? size_t(m0) * n : 0), rhs_per_system,
                                                   c_out + size_t(m0) * n, ctx->d_status, gws);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_reduced_solve_batch(rom_ctx* ctx, int n, int kb, int M, rom_buf* Ahat, rom_buf* w, rom_buf* rhs,
                                       int rhs_per_system, rom_buf* c_out) {
  ROM_CHECK(ctx && Ahat && w && rhs && c_out, "rom_reduced_solve_batch: null argument");
  ROM_CHECK(n >= 1 && n <= 4096, "rom_reduced_solve_batch: reduced dimension %d outside [1, 4096]", n);
  ROM_CHECK(kb >= 1 && M >= 0, "rom_reduced_solve_batch: bad sizes");
  ROM_CHECK(Ahat->n >= size_t(kb) * n * n && w->n >= size_t(M) * kb && c_out->n >= size_t(M) * n &&
                rhs->n >= (rhs_per_system ? size_t(M) * n : size_t(n)),
            "rom_reduced_solve_batch: buffer too small");
  if (M == 0) return ROM_OK;
  ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  ROM_TRY(rom_launch_reduced_solve(ctx, n, n, kb, M, Ahat->p, w->p, rhs->p, rhs_per_system, c_out->p));
  int status = 0;
  ROM_HIP(hipMemcpyAsync(&status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (status) {
    rom_set_error("rom_reduced_solve_batch: reduced matrix not positive definite");
    return ROM_ERR_NOT_SPD;
  }
  return ROM_OK;
}

// ============================================================================================
// small helpers for the basis builders: scaling, mean-centring, P1 point evaluation
// ============================================================================================
__global__ void k_scale(double* p, size_t n, double alpha) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] *= alpha;
}

extern "C" int rom_buf_scale(rom_buf* b, size_t off, size_t n, double alpha) {
  ROM_CHECK(b, "rom_buf_scale: null buffer");
  ROM_CHECK(off + n <= b->n, "rom_buf_scale: range exceeds buffer");
  if (n == 0) return ROM_OK;
  int grid = int(std::min<size_t>((n + 255) / 256, 4096));
  k_scale<<<grid, 256, 0, b->ctx->stream>>>(b->p + off, n, alpha);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// column means of the (M, dim) snapshot block, subtracted in place (PCA centring,
// sklearn PCA.fit called at src/lib/ReducedBasis.py:196); mean[dim] is kept for the caller.
// Two passes over row slabs so that the chip is full for any M x dim: partial column sums per slab
// (fixed order: deterministic), then every slab subtracts the mean it recomputes from the partials.
constexpr int CENTER_SLABS = 16;
__global__ void k_center_partial(const double* __restrict__ X, int M, long long dim, double* __restrict__ part) {
  const long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= dim) return;
  const int per = (M + CENTER_SLABS - 1) / CENTER_SLABS;
  const int m0 = blockIdx.y * per, m1 = min(M, m0 + per);
  double s = 0.0;
  for (int m = m0; m < m1; ++m) s += X[m * dim + j];
  part[blockIdx.y * dim + j] = s;
}

__global__ void k_center_apply(double* __restrict__ X, int M, long long dim, const double* __restrict__ part,
                               double* __restrict__ mean) {
  const long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= dim) return;
  double s = 0.0;
  for (int q = 0; q < CENTER_SLABS; ++q) s += part[q * dim + j];
  s /= double(M);
  if (blockIdx.y == 0) mean[j] = s;
  const int per = (M + CENTER_SLABS - 1) / CENTER_SLABS;
  const int m0 = blockIdx.y * per, m1 = min(M, m0 + per);
  for (int m = m0; m < m1; ++m) X[m * dim + j] -= s;
}

// X[m, :] -= row[:] for every m (the second half of the centring when the column means are already known)
__global__ void k_subtract_row(double* __restrict__ X, int M, long long dim, const double* __restrict__ row) {
  const long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (j >= dim) return;
  const double s = row[j];
  const int per = (M + CENTER_SLABS - 1) / CENTER_SLABS;
  const int m0 = blockIdx.y * per, m1 = min(M, m0 + per);
  for (int m = m0; m < m1; ++m) X[m * dim + j] -= s;
}

int rom_launch_subtract_row(rom_ctx* ctx, double* X, int M, int64_t dim, const double* d_row) {
  const dim3 grid(unsigned((dim + 255) / 256), CENTER_SLABS);
  {
    ROM_PROF(ctx, "center_rows", 1.0 * M * dim, 16.0 * M * dim);
    k_subtract_row<<<grid, 256, 0, ctx->stream>>>(X, M, dim, d_row);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

int rom_launch_center_rows(rom_ctx* ctx, double* X, int M, int64_t dim, double* d_mean) {
  double* part = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(CENTER_SLABS) * dim, &part));
  const dim3 grid(unsigned((dim + 255) / 256), CENTER_SLABS);
  {
    ROM_PROF(ctx, "center_rows", 2.0 * M * dim, 24.0 * M * dim);
    k_center_partial<<<grid, 256, 0, ctx->stream>>>(X, M, dim, part);
    k_center_apply<<<grid, 256, 0, ctx->stream>>>(X, M, dim, part, d_mean);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_center_rows(rom_ctx* ctx, rom_buf* X, int64_t row0, int M, int64_t dim, rom_buf* mean) {
  ROM_CHECK(ctx && X && mean, "rom_center_rows: null argument");
  ROM_CHECK(M >= 1 && dim >= 1 && row0 >= 0, "rom_center_rows: bad sizes");
  ROM_CHECK(size_t(row0 + M) * dim <= X->n && size_t(dim) <= mean->n, "rom_center_rows: buffers too small");
  return rom_launch_center_rows(ctx, X->p + row0 * dim, M, dim, mean->p);
}

// X[row0 + i, :] *= factors[i]  (i < rows; `factors` on the host): the 1/sigma scaling of the lifted POD modes
__global__ void k_rows_scale(double* __restrict__ X, long long dim, const double* __restrict__ fac) {
  const double a = fac[blockIdx.y];
  double* row = X + blockIdx.y * dim;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < dim; j += (long long)gridDim.x * blockDim.x)
    row[j] *= a;
}

int rom_launch_rows_scale(rom_ctx* ctx, double* X, int rows, int64_t dim, const double* d_fac) {
  if (rows <= 0) return ROM_OK;
  k_rows_scale<<<dim3(unsigned(std::min<int64_t>((dim + 255) / 256, 64)), rows), 256, 0, ctx->stream>>>(X, dim, d_fac);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_rows_scale(rom_ctx* ctx, rom_buf* X, int64_t row0, int rows, int64_t dim, const double* factors_host) {
  ROM_CHECK(ctx && X && (factors_host || rows == 0), "rom_rows_scale: null argument");
  ROM_CHECK(rows >= 0 && dim >= 1 && row0 >= 0 && size_t(row0 + rows) * dim <= X->n, "rom_rows_scale: bad sizes");
  if (rows == 0) return ROM_OK;
  double* fac = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(rows), &fac));
  ROM_HIP(hipMemcpyAsync(fac, factors_host, size_t(rows) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));  // the host array may be reused by the caller
  return rom_launch_rows_scale(ctx, X->p + row0 * dim, rows, dim, fac);
}

// sklearn's svd_flip(u_based_decision=False) (PCA call at src/lib/ReducedBasis.py:196): every row is multiplied by
// the sign of its entry of largest magnitude (first one on ties, like np.argmax).  Two launches over (segment, row): the
// per-segment maxima (value, first position), then every segment reduces the SIGN_SEGS candidates of its row in segment
// order and flips its part -- a row is spread over the chip instead of being one workgroup's 254-iteration chain.
constexpr int SIGN_SEGS = 16;
__global__ __launch_bounds__(256) void k_rows_sign_partial(const double* __restrict__ X, long long dim, double* __restrict__ pv,
                                                           long long* __restrict__ pi) {
  __shared__ double bv[4];
  __shared__ long long bi[4];
  const double* row = X + blockIdx.y * dim;
  const long long per = (dim + SIGN_SEGS - 1) / SIGN_SEGS, j0 = blockIdx.x * per, j1 = min(dim, j0 + per);
  double best = -1.0;
  long long at = j0;
  for (long long j = j0 + threadIdx.x; j < j1; j += 256) {
    const double v = fabs(row[j]);
    if (v > best) { best = v; at = j; }  // ascending j per thread: keeps the first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_down(best, o, 64);
    const long long oa = __shfl_down(at, o, 64);
    if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = at; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < at)) { best = bv[w]; at = bi[w]; }
    pv[blockIdx.y * SIGN_SEGS + blockIdx.x] = best;
    pi[blockIdx.y * SIGN_SEGS + blockIdx.x] = at;
  }
}

__global__ __launch_bounds__(256) void k_rows_sign_apply(double* __restrict__ X, long long dim, const double* __restrict__ pv,
                                                         const long long* __restrict__ pi, const double* __restrict__ pivot_vals) {
  double* row = X + blockIdx.y * dim;
  double best = -1.0;
  int seg = 0;
  for (int q = 0; q < SIGN_SEGS; ++q) {  // segments ascend with j: '>' keeps the first maximum
    const double v = pv[blockIdx.y * SIGN_SEGS + q];
    if (v > best) { best = v; seg = q; }
  }
  if (!(pivot_vals[blockIdx.y * SIGN_SEGS + seg] < 0.0)) return;
  const long long per = (dim + SIGN_SEGS - 1) / SIGN_SEGS, j0 = blockIdx.x * per, j1 = min(dim, j0 + per);
  for (long long j = j0 + threadIdx.x; j < j1; j += 256) row[j] = -row[j];
}

// signed value at each segment's candidate position (read before any segment flips)
__global__ void k_rows_sign_pivots(const double* __restrict__ X, long long dim, const long long* __restrict__ pi,
                                   double* __restrict__ pivot_vals, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) pivot_vals[i] = X[(i / SIGN_SEGS) * dim + pi[i]];
}

int rom_launch_rows_sign_flip(rom_ctx* ctx, double* X, int rows, int64_t dim) {
  if (rows <= 0) return ROM_OK;
  double* ws = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(3) * rows * SIGN_SEGS, &ws));
  double* pv = ws;
  long long* pi = reinterpret_cast<long long*>(ws + size_t(rows) * SIGN_SEGS);
  double* piv = ws + size_t(2) * rows * SIGN_SEGS;
  const dim3 grid(SIGN_SEGS, unsigned(rows));
  k_rows_sign_partial<<<grid, 256, 0, ctx->stream>>>(X, dim, pv, pi);
  k_rows_sign_pivots<<<unsigned((rows * SIGN_SEGS + 255) / 256), 256, 0, ctx->stream>>>(X, dim, pi, piv, rows * SIGN_SEGS);
  k_rows_sign_apply<<<grid, 256, 0, ctx->stream>>>(X, dim, pv, pi, piv);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_rows_sign_flip(rom_ctx* ctx, rom_buf* X, int64_t row0, int rows, int64_t dim) {
  ROM_CHECK(ctx && X, "rom_rows_sign_flip: null argument");
  ROM_CHECK(rows >= 0 && dim >= 1 && row0 >= 0 && size_t(row0 + rows) * dim <= X->n, "rom_rows_sign_flip: bad sizes");
  return rom_launch_rows_sign_flip(ctx, X->p + row0 * dim, rows, dim);
}

// evaluate_solutions (src/lib/SolutionsManagers.py:221-244): P1 interpolation on the SW-NE split
// cells.  Cell indices / local coordinates are computed by the host shim (searchsorted on the two
// 1-D grids); the kernel gathers the three vertex values per (solution, point).
__global__ void k_eval_points(int nr, int nc, long long dim, const double* __restrict__ U, int K, int npts,
                              const int* __restrict__ ix, const int* __restrict__ iy,
                              const double* __restrict__ tx, const double* __restrict__ ty,
                              double* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (p >= npts || k >= K) return;
  const double* u = U + k * dim;
  // vertex grid with Dirichlet ring: V[y][x], y = 0..nr+1, x = 0..nc+1 ; inner vertex (y,x) -> u[(y-1)*nc + x-1]
  auto val = [&](int y, int x) -> double {
    return (y >= 1 && y <= nr && x >= 1 && x <= nc) ? u[(long long)(y - 1) * nc + (x - 1)] : 0.0;
  };
  int x0 = ix[p], y0 = iy[p];
  double qx = tx[p], qy = ty[p];
  double v;
  if (qx + qy < 1)
    v = (1 - qx - qy) * val(y0, x0) + qx * val(y0, x0 + 1) + qy * val(y0 + 1, x0);
  else
    v = (qx + qy - 1) * val(y0 + 1, x0 + 1) + (1 - qx) * val(y0 + 1, x0) + (1 - qy) * val(y0, x0 + 1);
  out[(size_t)k * npts + p] = v;
}

extern "C" int rom_evaluate_points(rom_fem* f, rom_buf* U, int64_t row0, int K, int npts, const int* ix_host,
                                   const int* iy_host, const double* tx_host, const double* ty_host,
                                   double* out_host) {
  ROM_CHECK(f && U && (npts == 0 || (ix_host && iy_host && tx_host && ty_host)) && (out_host || K * npts == 0),
            "rom_evaluate_points: null argument");
  ROM_CHECK(K >= 0 && npts >= 0 && row0 >= 0, "rom_evaluate_points: negative size");
  ROM_CHECK(size_t(row0 + K) * f->dim <= U->n, "rom_evaluate_points: rows out of range");
  if (K == 0 || npts == 0) return ROM_OK;
  for (int p = 0; p < npts; ++p)
    ROM_CHECK(ix_host[p] >= 0 && ix_host[p] <= f->nc && iy_host[p] >= 0 && iy_host[p] <= f->nr,
              "rom_evaluate_points: point %d outside the domain", p);
  rom_ctx* ctx = f->ctx;
  // one carve-up of the context's scratch block (the experiment loop calls this once per basis size: no
  // hipMalloc / hipFree per call): [ix | iy] as ints, [tx | ty], out
  const size_t n_idx = (2 * size_t(npts) * sizeof(int) + sizeof(double) - 1) / sizeof(double);
  double* scratch = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, n_idx + 2 * size_t(npts) + size_t(K) * npts, &scratch));
  int* d_i = reinterpret_cast<int*>(scratch);
  double* d_t = scratch + n_idx;
  double* d_out = d_t + 2 * size_t(npts);
  ROM_HIP(hipMemcpyAsync(d_i, ix_host, npts * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipMemcpyAsync(d_i + npts, iy_host, npts * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipMemcpyAsync(d_t, tx_host, npts * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ROM_HIP(hipMemcpyAsync(d_t + npts, ty_host, npts * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ROM_PROF(ctx, "eval_points", 8.0 * K * npts, 32.0 * K * npts);
    k_eval_points<<<dim3((npts + 255) / 256, K), 256, 0, ctx->stream>>>(f->nr, f->nc, f->dim, U->p + row0 * f->dim,
                                                                       K, npts, d_i, d_i + npts, d_t, d_t + npts, d_out);
  }
  ROM_HIP(hipGetLastError());
  ROM_HIP(hipMemcpyAsync(out_host, d_out, size_t(K) * npts * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));  // the caller's host arrays are free again; so is the scratch block
  return ROM_OK;
}
