// k_extend_res: the harmonic extension of blocks whose sides are all compressed AND whose K = sum(rank + 1) fits eight
// 8-wide segments (K <= 64: every block of a 2 x 2 geometry, config C2 / C3) -- persistent workgroups with the systems'
// operand RESIDENT in LDS.
//
// Why (DESIGN.md section 5, round 2 measurements): at K = 64 a 128 x 128 tile of k_extend128 lives 56 k cycles of
// which the matrix pipe needs 16 k -- prologue (first operands 5-6 k cycles away), an epilogue of 32 stores per wave
// that nothing overlaps (stores overlap with MFMAs only inside ONE wave's instruction stream), and a k loop whose four
// chunks each end in `vmcnt(0) + barrier`.  The persistent kernel k_extend_p interleaved stores and MFMAs in one wave
// but spent 98 instructions per 8 MFMAs on a segment cursor, slot arithmetic and store masks.
//
// Here a workgroup owns (128 systems, one block, a contiguous range of 64-vertex tiles of that block's mesh rows):
//   * A = the systems' coefficient blocks [8 segments][128 systems][8]  (64 KB)  is fetched ONCE (LDS-DMA) and stays;
//   * B = the table rows of a tile            [8 segments][ 64 vertices][8]  (32 KB)  is double buffered: the tile's
//     whole K is resident, so there is no chunk loop, no cursor -- ONE counted wait + barrier per tile;
//   * a wave owns 32 systems x 32 vertices = 2 x 2 accumulators in TWO sets: while it multiplies tile t it stores
//     tile t - 1, one 1 KB store per 8 MFMAs (per segment), in the same instruction stream;
//   * vmcnt is counted: at the end of a tile the wave waits for everything but its last 8 memory operations -- the
//     stores it has just issued stay in flight, the operand loads of the next tile (older) have landed.
// 129 KB of LDS: one workgroup of eight waves per CU (two per SIMD); the grid is sized to one workgroup per CU.
// Per 8 MFMAs a wave issues 8 ds_read_b64 (4 per k-step), the store with its ~8 instructions of epilogue arithmetic,
// and its share of 4 DMA instructions per tile: ~28 instructions.
// Rows are bit-identical to k_extend128's: same products in the same order per accumulator (K ascending), same epilogue.
#include "rom_fem_dev.h"

namespace {

__device__ inline const char* res_uniform(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const char*>((unsigned long long)hi << 32 | lo);
}
__device__ inline unsigned long long res_uniform(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return (unsigned long long)hi << 32 | lo;
}

constexpr int RES_A_BYTES = 8 * 128 * 64;   // 65,536
constexpr int RES_B_BYTES = 8 * 64 * 64;    // 32,768 per slot

}  // namespace

size_t rom_extend_res_lds_bytes() { return size_t(RES_A_BYTES) + 2 * RES_B_BYTES + 128 * sizeof(double); }

#define RES_DMA(LDS_, BASE_, VOFF_)                                                                                \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),   \
               "v"(VOFF_), "s"(res_uniform(BASE_))                                                                 \
               : "memory", "m0")

// grid (nsplit, ceil(Mc / 128), blocks of this launch); 512 threads; dynamic LDS rom_extend_res_lds_bytes()
__global__ __launch_bounds__(512, 1) void k_extend_res(FemDev f, X128Args xa, int Mc, double* __restrict__ U, long long row0,
                                                         int nsplit, int with_expand) {
  extern __shared__ __align__(16) char lds_bytes[];
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds_bytes));
  double* const scs = reinterpret_cast<double*>(lds_bytes + RES_A_BYTES + 2 * RES_B_BYTES);
  const int n1 = f.n1, N = f.N;
  const int nct = (n1 + 63) / 64;                 // 64-vertex tiles per mesh row
  const int ntile = n1 * nct;
  const int bz = blockIdx.z;
  const int b = xa.blocks[bz];
  const int p = b / f.ncb, q = b % f.ncb;
  const BlockSide sd = xa.sides[bz];              // by value: one batch of scalar loads
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = w >> 1, wc = w & 1;
  const int fr = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 128;
  const int t_lo = int((long long)blockIdx.x * ntile / nsplit), t_hi = int((long long)(blockIdx.x + 1) * ntile / nsplit);
  const bool has_tiles = t_lo < t_hi;
  // ---- the K segments of this block: side by side, ceil((rank + 1) / 8) segments of 8 each (as in k_extend128).
  // (plain scalars and select chains: indexing the by-value descriptor with a run-time side puts it into scratch memory)
  const int md0 = sd.s[0].mode, md1 = sd.s[1].mode, md2 = sd.s[2].mode, md3 = sd.s[3].mode;
  const int cnt0 = md0 == 2 ? min(2 * sd.s[0].nch, (sd.s[0].r + 1 + 7) / 8) : 0;
  const int cnt1 = md1 == 2 ? min(2 * sd.s[1].nch, (sd.s[1].r + 1 + 7) / 8) : 0;
  const int cnt2 = md2 == 2 ? min(2 * sd.s[2].nch, (sd.s[2].r + 1 + 7) / 8) : 0;
  const int cnt3 = md3 == 2 ? min(2 * sd.s[3].nch, (sd.s[3].r + 1 + 7) / 8) : 0;
  const int off_0 = sd.s[0].off, off_1 = sd.s[1].off, off_2 = sd.s[2].off, off_3 = sd.s[3].off;
  const int gs_0 = sd.s[0].gseg, gs_1 = sd.s[1].gseg, gs_2 = sd.s[2].gseg, gs_3 = sd.s[3].gseg;
  const int pre0 = cnt0, pre1 = pre0 + cnt1, pre2 = pre1 + cnt2;
  const int nseg = min(pre2 + cnt3, 8);           // (<= 8: checked on the host)
  const unsigned ybytes_row = unsigned(f.nGp) * 8u;
  const char* const ybase = reinterpret_cast<const char*>(f.y) + size_t(min(m0, Mc - 1)) * ybytes_row;
  const char* const gsbase = reinterpret_cast<const char*>(f.Gs);
  const size_t seg_stride = size_t(n1) * n1 * 64;  // bytes between the K segments of a table
  // segment e -> (side, position in the side): all uniform
  auto seg_side = [&](int e) -> int { return (e >= pre0 ? 1 : 0) + (e >= pre1 ? 1 : 0) + (e >= pre2 ? 1 : 0); };
  auto seg_q = [&](int e, int s) -> int { return e - (s == 0 ? 0 : s == 1 ? pre0 : s == 2 ? pre1 : pre2); };
  auto side_off = [&](int s) -> int { return s == 0 ? off_0 : s == 1 ? off_1 : s == 2 ? off_2 : off_3; };
  auto side_gseg = [&](int s) -> int { return s == 0 ? gs_0 : s == 1 ? gs_1 : s == 2 ? gs_2 : gs_3; };
  const unsigned du16 = unsigned(((lane & 3) ^ ((lane >> 4) & 3)) * 16);  // logical 16-byte unit this lane fetches (swizzle)

  if (has_tiles) {
  // ---- h^2 / a_b of the workgroup's systems
  if (threadIdx.x < 128) {
    const int m = m0 + threadIdx.x;
    scs[threadIdx.x] = m < Mc ? f.y[size_t(m) * f.nGp + f.sblk0 + b] : 0.0;
  }
  // ---- A: wave w fetches the 16 systems 16 w + (lane >> 2) of every segment (all addresses first, then the loads)
  {
    const int rl = 16 * w + (lane >> 2);
    const unsigned voA = unsigned(max(0, min(rl, Mc - 1 - m0))) * ybytes_row + du16;
    const char* pAe[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int s = seg_side(e);
      pAe[e] = res_uniform(ybase + size_t(side_off(s)) * 8 + size_t(seg_q(e, s)) * 64);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < nseg) RES_DMA(lds0 + unsigned(e) * 8192u + unsigned(w) * 1024u, pAe[e], voA);
  }
  // ---- B: wave w fetches the 16 vertices 16 (w & 3) + (lane >> 2) of the tile for the segments (w >> 2) + 2 k
  const int rg = w & 3;
  const char* pBs[4];
  int bci[4], bcj[4], bk0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = min((w >> 2) + 2 * k, 7), s = seg_side(e);
    pBs[k] = res_uniform(gsbase + size_t(side_gseg(s)) * 8 + size_t(seg_q(e, s)) * seg_stride);
    // row of the side's segment-major table = ci i + cj j + k0 (i, j the 1-based interior indices): k_extend128
    bci[k] = s == 1 ? -n1 : n1;
    bcj[k] = s == 3 ? -1 : 1;
    bk0[k] = s == 1 ? n1 * n1 - 1 : s == 3 ? 0 : -n1 - 1;
  }
  auto issue_B = [&](int slot, int iv, int h) {
    const int jv = min(64 * h + 1 + 16 * rg + (lane >> 2), n1);
    const unsigned dst = lds0 + unsigned(RES_A_BYTES) + unsigned(slot) * unsigned(RES_B_BYTES) + unsigned(rg) * 1024u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = (w >> 2) + 2 * k;
      if (e < nseg) {
        const unsigned voB = unsigned(bci[k] * iv + bcj[k] * jv + bk0[k]) * 64u + du16;
        RES_DMA(dst + unsigned(e) * 4096u, pBs[k], voB);
      }
    }
  };
  // ---- fragment addressing: lane (fr, kq) reads row fr (+ 16 i) at k = 4 half + kq: unit (2 half + (kq >> 1)) ^ ((row >> 2) & 3)
  const unsigned fx = unsigned((kq >> 1) ^ (fr >> 2));
  const unsigned fa0 = unsigned(fr * 64) + (fx << 4) + unsigned(kq & 1) * 8u;
  const unsigned fa1 = unsigned(fr * 64) + ((fx ^ 2u) << 4) + unsigned(kq & 1) * 8u;
  const char* const fragA = lds_bytes + wr * 2048;                    // + segment * 8192 + i * 1024
  const char* const fragB0 = lds_bytes + RES_A_BYTES + wc * 2048;      // + slot * 32768 + segment * 4096 + j * 1024
#define RES_FRAGS(SLOT_, E_, HALF_, AF_, BF_)                                                                       \
  do {                                                                                                             \
    const unsigned fo_ = (HALF_) ? fa1 : fa0;                                                                      \
    const char* pa_ = fragA + (E_) * 8192 + fo_;                                                                   \
    const char* pb_ = fragB0 + (SLOT_) * RES_B_BYTES + (E_) * 4096 + fo_;                                          \
    AF_[0] = *reinterpret_cast<const double*>(pa_);                                                                \
    AF_[1] = *reinterpret_cast<const double*>(pa_ + 1024);                                                         \
    BF_[0] = *reinterpret_cast<const double*>(pb_);                                                                \
    BF_[1] = *reinterpret_cast<const double*>(pb_ + 1024);                                                         \
  } while (0)

  // ---- store side: the lane's rows (systems) and, per tile, its two adjacent vertices
  const bool odd = lane & 1;
  const int tl = wc * 32 + (odd ? 16 : 0) + fr - (odd ? 1 : 0);       // first of the lane's two vertices, tile-local
  const int to0 = wc * 32 + fr, to1 = to0 + 16;                        // the lane's own accumulator columns
  char* const rowp = reinterpret_cast<char*>(U) + size_t(row0 + m0 + wr * 32 + kq) * size_t(f.dim) * 8;
  const long long step = 4ll * f.dim * 8;                             // four systems down
  const bool full = m0 + 128 <= Mc;
  const int mrow = m0 + wr * 32 + kq;

  int iv = t_lo / nct + 1, h = t_lo % nct;                            // mesh row (1-based), column tile
  issue_B(0, iv, h);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // A, B(first tile), scs
  double sc8[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) sc8[st] = scs[wr * 32 + 4 * st + kq];

  d4_t accp[2][2];                    // finished tile (being stored)
  double wp0 = 0.0, wp1 = 0.0;        // W at the lane's columns, finished tile
  char* sp = nullptr;                 // where the lane's next store of the finished tile goes
  unsigned long long k16p = 0, k8p = 0;
  bool have_prev = false;
  int slot = 0;

  // one store step of the finished tile: rows 4 st + kq (+ wr * 32), the lane's two adjacent vertices
#define RES_STORE_STEP(ST_)                                                                                        \
  do {                                                                                                             \
    const int i_ = (ST_) >> 2, g_ = (ST_)&3;                                                                       \
    const double x0_ = accp[i_][0][g_] + sc8[ST_] * wp0, x1_ = accp[i_][1][g_] + sc8[ST_] * wp1;                   \
    const double got_ = lane_swap1(odd ? x0_ : x1_);                                                               \
    const double2_u pr_ = double2_u{odd ? got_ : x0_, odd ? x1_ : got_};                                           \
    unsigned long long in_ = ~0ull;                                                                                \
    if (!full) in_ = __builtin_amdgcn_ballot_w64(mrow + 4 * (ST_) < Mc);                                           \
    const unsigned long long m16_ = res_uniform(k16p & in_), m8_ = res_uniform(k8p & in_);                         \
    unsigned long long sv_;                                                                                        \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %2, %3, off\n\ts_mov_b64 exec, %0" \
                 : "=&s"(sv_)                                                                                      \
                 : "s"(m16_), "v"(sp), "v"(pr_)                                                                    \
                 : "memory");                                                                                      \
    if (m8_ != 0)                                                                                                  \
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx2 %2, %3, off\n\ts_mov_b64 exec, %0" \
                   : "=&s"(sv_)                                                                                    \
                   : "s"(m8_), "v"(sp), "v"(pr_.x)                                                                 \
                   : "memory");                                                                                    \
    sp += step;                                                                                                    \
  } while (0)

  for (int t = t_lo; t < t_hi; ++t) {
    // next tile's coordinates and operands: requested before anything else of this tile
    int iv2 = iv, h2 = h + 1;
    if (h2 == nct) { h2 = 0; ++iv2; }
    const bool more = t + 1 < t_hi;
    if (more) issue_B(slot ^ 1, iv2, h2);
    // W at the lane's own columns of THIS tile (used when the tile is stored, one tile later): plain loads would make
    // the compiler wait for them -- and, one counter in order, for the operand loads just issued -- at their first use
    double wc0, wc1;
    {
      const int j0c = 64 * h + 1 + to0, j1c = 64 * h + 1 + to1;
      const double* a0 = f.W + size_t(iv - 1) * n1 + (min(j0c, n1) - 1);
      const double* a1 = f.W + size_t(iv - 1) * n1 + (min(j1c, n1) - 1);
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(wc0) : "v"(a0) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(wc1) : "v"(a1) : "memory");
    }
    d4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    double af[2][2], bf[2][2];
    RES_FRAGS(slot, 0, 0, af[0], bf[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e < nseg) {
        RES_FRAGS(slot, e, 1, af[1], bf[1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        if (e + 1 < nseg) RES_FRAGS(slot, (e + 1 < 8 ? e + 1 : 7), 0, af[0], bf[0]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
      }
      if (have_prev) RES_STORE_STEP(e);
    }
    // Everything but the last 8 memory operations of this wave has completed: the operand loads of the next tile and
    // this tile's W values (both older than the >= 8 stores above).  Where fewer than 8 stores were issued -- the first
    // tile; a system group with missing systems, whose stores under an empty EXEC mask are not counted -- vmcnt(0).
    if (have_prev && full) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" : "+v"(wc0), "+v"(wc1)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : "+v"(wc0), "+v"(wc1)::"memory");
    // this tile becomes the finished one
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];
    {
      const int jcol = 64 * h + 1 + tl;  // 1-based
      const long long off0 = (long long)(p * N + iv - 1) * f.nc + q * N - 1 + jcol;
      k16p = __builtin_amdgcn_ballot_w64(jcol + 1 <= n1);
      k8p = __builtin_amdgcn_ballot_w64(jcol <= n1 && jcol + 1 > n1);
      sp = rowp + off0 * 8;
      wp0 = (64 * h + 1 + to0) <= n1 ? wc0 : 0.0;
      wp1 = (64 * h + 1 + to1) <= n1 ? wc1 : 0.0;
    }
    have_prev = true;
    slot ^= 1;
    iv = iv2;
    h = h2;
  }
  // the last tile
#pragma unroll
  for (int e = 0; e < 8; ++e) RES_STORE_STEP(e);
#undef RES_STORE_STEP
#undef RES_FRAGS
  }  // has_tiles
  // The (small) expansion of the edge values -- it writes the interface entries of the snapshot rows and the nodal
  // blocks of the interface vector, nothing the extension reads -- rides at the tail of this launch: its items are dealt
  // to the workgroups (at most one each on a full chip), four waves each, staging area = the B slots.
  if (with_expand) {
    static_assert(STAGE_TOTAL * sizeof(double) <= 2 * RES_B_BYTES, "expansion staging must fit");
    __syncthreads();                 // every wave has read its last fragments
    if (threadIdx.x >= 256) return;  // (the expansion is written for four waves)
    const int nx = f.n1p / 64, ny = (Mc + 63) / 64, nitem = nx * ny * (f.nexp + 1);
    const int nwg = int(gridDim.x * gridDim.y * gridDim.z);
    for (int item = int(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); item < nitem; item += nwg)
      expand_tile(f, Mc, U, row0, reinterpret_cast<double*>(lds_bytes + RES_A_BYTES), item % nx, (item / nx) % ny, item / (nx * ny));
  }
}
#undef RES_DMA
