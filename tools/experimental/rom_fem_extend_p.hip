// libromhc -- harmonic extension into the blocks by PERSISTENT workgroups, two per CU (round 2; gfx950 only).
//
// Same product as k_extend128 (rom_fem_kernels.hip: U_I,b[m, v] = (h^2 / a_b) W[v] + sum_sides sum_k c_s[m, k] Tab_s[v][k],
// K = sum(rank + 1) ~ 64 in 8-wide segments), same sums in the same order: identical rows.  What is different is how the
// three streams of the kernel -- operand loads, MFMAs, 528 MB of stores per C2 step -- are kept running at once:
//
//   * what the stamps of k_extend128 show (tools/gpu_stamps.py): a workgroup lives entry -> loads -> k loop -> stores ->
//     exit -> successor's entry, about 62,000 cycles of which the matrix pipe works 16,400; two per CU overlap these
//     lives only pairwise.  And a store stream issued by OTHER waves does not overlap with a SIMD's MFMAs
//     (tools/mfma_store_overlap.hip: waves that only store next to waves that only multiply take the SUM of the two
//     times; one wave that issues a store after every eight MFMAs gets 61 TFLOP/s and 3.8 TB/s at once).
//   * here 2 x 256 workgroups of 4 waves walk over the tiles (64 systems x 128 vertices, x fastest) in contiguous
//     ranges.  The K chunks (16 wide) of ALL the tiles of a workgroup form one stream through a ring of three LDS slots,
//     filled by LDS-DMA loads (global_load_lds_dwordx4: no staging registers, no ds_write): one barrier per chunk, in
//     its middle, publishes chunk g + 1 and frees the slot of chunk g - 1, into which chunk g + 2 is requested at once.
//     The two workgroups of a CU are independent: the address arithmetic of one runs under the MFMAs of the other.
//   * a wave owns 64 systems x 32 vertices (4 x 2 accumulators).  When a tile is done the accumulators move to a second
//     register set and are stored DURING the next tile: after every MFMA k-step one 16-byte-per-lane store (4 systems x
//     32 vertices) -- 16 per wave and tile.
//   * vmcnt retires in order, loads and stores alike: a wait for a load is also a wait for every older store.  The
//     compiler cannot count across the predicated stores and falls back to vmcnt(0), which drains the store stream at
//     every chunk; so every vector-memory instruction of the loop is inline assembly issued UNCONDITIONALLY (lanes and
//     whole instructions that have nothing to do run with an empty exec mask or fetch a page of zeros): every wave
//     issues 7 loads behind the barrier of a chunk and KS stores after each of its 4 k-steps, and the one wait of a
//     chunk is s_waitcnt vmcnt(4 KS): chunk g + 1's loads and everything older have landed, the stores of the last
//     four k-steps stay in flight.
//   * operand fragments of k-step j + 1 are read from LDS before the MFMAs of step j are issued; h^2 / a_b and W travel
//     with the chunks (seventh load of a wave) into four-deep LDS buffers.
//
// LDS slot: {A k 0..7 | A k 8..15 | B k 0..7 | B k 8..15}, a row (A: system, B: vertex) of a half = 64 bytes = four
// 16-byte units stored at position u ^ ((row >> 2) & 3).  A DMA instruction writes 64 lanes x 16 bytes contiguously = 16
// rows of ONE half, i.e. of one block side: its addresses are a scalar base plus a 32-bit lane offset; and the MFMA
// fragment reads (16 rows x 2 k per half wave) hit 32 different bank pairs.
#include <hip/hip_runtime.h>

#include "rom_fem_dev.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

// a pointer that is the same in every lane, in scalar registers whatever the compiler thinks of it
__device__ inline const char* xp_uniform(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const char*>((unsigned long long)hi << 32 | lo);
}

#ifdef ROMHC_STAMPS
// per-chunk cycle stamps of workgroup 3, waves 0 and 3: [chunk][6] = start, before / after the barrier, loads issued,
// end, (tile << 8 | chunk of the tile)
__device__ unsigned long long g_stamps_p[2 * 1024 * 6];
extern "C" int rom_debug_stamps_p(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_p), size_t(n) * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int rom_debug_stamps_p_clear() {
  static unsigned long long zeros[2 * 1024 * 6];
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_p), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
#define XP_STAMP(i)                                                                                            \
  do {                                                                                                         \
    if (blockIdx.x == 3 && (tid == 0 || tid == 192) && g < 1024) {                                             \
      g_stamps_p[((tid ? 1024 : 0) + g) * 6 + (i)] = __builtin_readcyclecounter();                             \
      if ((i) == 0) g_stamps_p[((tid ? 1024 : 0) + g) * 6 + 5] = (unsigned long long)(t - lo) << 8 | cc;       \
    }                                                                                                          \
  } while (0)
#else
#define XP_STAMP(i)
#endif

// DBG (probes, tools/gpu_ext_time.py; results are WRONG with bits 1 and 4): bit 0 = every store runs with an empty
// exec mask, bit 1 = no epilogue at all, bit 4 = no operand loads
template <bool FLAT, int DBG>
__global__ __launch_bounds__(256, 2) void k_extend_p(FemDev f, X128Args xa, const double* __restrict__ a, int Mc,
                                                     double* __restrict__ U, long long row0, int nz) {
  extern __shared__ __align__(16) char xp_lds[];
  // KS = FLAT ? 3 : 2 store instructions per k-step: 16-byte pair, single first, (single second)
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)xp_lds));
  const int n1 = f.n1, N = f.N;
  const int nct = (n1 + 127) / 128;
  const int nvert = n1 * n1;
  const int ntile = FLAT ? (nvert + 127) / 128 : n1 * nct;
  const int mt = (Mc + 63) / 64;
  const long long T = (long long)ntile * mt * nz;
  const int lo = int(T * blockIdx.x / gridDim.x), hi = int(T * (blockIdx.x + 1) / gridDim.x);
  if (lo >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const bool odd = lane & 1;
  const char* const ybytes = reinterpret_cast<const char*>(f.y);
  const char* const gbytes = reinterpret_cast<const char*>(f.G);
  const char* const wbytes = reinterpret_cast<const char*>(f.W);
  char* const ubytes = reinterpret_cast<char*>(U);
  const char* const zbase = reinterpret_cast<const char*>(f.W + size_t(n1) * n1);  // XP_ZERO_PAGE doubles of zeros
  const size_t row_bytes = size_t(f.dim) * 8;
  const unsigned ybytes_row = unsigned(f.nGp) * 8u;

  // ---- fragment addressing: lane (fr, kq) reads row fr (+ 16 i) at k = 4 kki + kq: half kki >> 1, unit
  // (2 (kki & 1) + (kq >> 1)) ^ (fr >> 2), byte (kq & 1) * 8
  const unsigned fx = unsigned((kq >> 1) ^ (fr >> 2));
  const unsigned fa0 = unsigned(fr * 64) + (fx << 4) + unsigned(kq & 1) * 8u;
  const unsigned fa1 = unsigned(fr * 64) + ((fx ^ 2u) << 4) + unsigned(kq & 1) * 8u;
#define XP_FRAGS(SLOT_, KKI_, AF_, BF_)                                                                            \
  do {                                                                                                             \
    const char* pf_ = xp_lds + (SLOT_) * XP_SLOT_BYTES + (((KKI_)&1) ? fa1 : fa0);                                 \
    const char* pg_ = pf_ + 8192 + w * 2048;                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                               \
        AF_[i_] = *reinterpret_cast<const double*>(pf_ + ((KKI_) >> 1) * 4096 + i_ * 1024);                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                               \
        BF_[j_] = *reinterpret_cast<const double*>(pg_ + ((KKI_) >> 1) * 8192 + j_ * 1024);                        \
  } while (0)

  // ---- DMA lane constants: an instruction fetches one half (4 units) of 16 rows: lane -> row (lane >> 2) of the
  // group, stored unit lane & 3 = logical unit (lane & 3) ^ ((lane >> 4) & 3)
  const int drow = lane >> 2;
  const unsigned du16 = unsigned(((lane & 3) ^ ((lane >> 4) & 3)) * 16);

  // ---- load cursor: a walk over the 8-wide segments of the tiles (l_*: uniform; lv_*: per lane).  Rows that do not
  // exist (systems >= Mc, vertices behind the end of a mesh row / of the block) are CLAMPED to the last one that does:
  // their products are never stored, and the loads need no per-lane validity.  Within a side the two running pointers
  // advance by 64 bytes per segment; the arithmetic of a new side / a new tile is paid when it starts.
  int l_tile = lo, l_left = 0, l_segs = 0, l_side = -1, l_pad = 0;  // segments left in the side / in the tile, zero segments behind
  int l_c0 = 0, l_c1 = 0, l_c2 = 0, l_c3 = 0;                        // segments per side
  int l_a0 = 0, l_a1 = 0, l_a2 = 0, l_a3 = 0, l_g0 = 0, l_g1 = 0, l_g2 = 0, l_g3 = 0, l_n0 = 0, l_n1 = 0, l_n2 = 0, l_n3 = 0;
  int l_bx = lo % ntile, l_by = (lo / ntile) % mt, l_bz = lo / (ntile * mt);
  const char* l_pA = zbase;  // (uniform parts) segment of the side's coefficient block, rows of the tile's first system
  const char* l_pB = zbase;  // segment of the side's table
  const char* l_tA = ybytes;  // ybytes + first system of this wave's 16 rows
  int lv_i[2] = {1, 1}, lv_j[2] = {1, 1};  // vertex (i, j), 1-based, of the lane's row in the wave's two B row groups
  unsigned lv_a = 0, lv_b[2] = {0, 0};     // lane offsets (bytes) behind l_pA / l_pB
  const unsigned lv_z = unsigned(lane) * 16u;
  const char* l_ex = zbase;  // the wave's seventh load: h^2 / a_b (waves 0, 1), W (wave 2), nothing (wave 3)
  unsigned lv_ex = lv_z, l_exlds = XP_DUMP_OFF;
  bool l_first = true, l_done = false;  // the cursor is on the first chunk of its tile / behind the last tile
  // the next side that has segments: its pointers and the lanes' table rows
#define XP_NEXT_SIDE()                                                                                           \
  do {                                                                                                           \
    do {                                                                                                         \
      ++l_side;                                                                                                  \
      l_left = l_side == 0 ? l_c0 : l_side == 1 ? l_c1 : l_side == 2 ? l_c2 : l_c3;                              \
    } while (l_left == 0 && l_side < 3);                                                                         \
    const int ao_ = l_side == 0 ? l_a0 : l_side == 1 ? l_a1 : l_side == 2 ? l_a2 : l_a3;                         \
    const int go_ = l_side == 0 ? l_g0 : l_side == 1 ? l_g1 : l_side == 2 ? l_g2 : l_g3;                         \
    const int nb_ = l_side == 0 ? l_n0 : l_side == 1 ? l_n1 : l_side == 2 ? l_n2 : l_n3; /* bytes per table row */ \
    /* row of the side's table = ci i + cj j + c0 (h0_row) */                                                    \
    const int cm_ = (l_side & 1) ? -n1 : n1, k0_ = (l_side & 1) ? (N - 1) * n1 - 1 : -n1 - 1;                    \
    const int ci_ = (l_side & 2) ? 1 : cm_, cj_ = (l_side & 2) ? cm_ : 1;                                        \
    l_pA = l_tA + size_t(ao_) * 8;                                                                               \
    l_pB = gbytes + size_t(go_) * 8;                                                                             \
    lv_b[0] = unsigned(ci_ * lv_i[0] + cj_ * lv_j[0] + k0_) * unsigned(nb_) + du16;                              \
    lv_b[1] = unsigned(ci_ * lv_i[1] + cj_ * lv_j[1] + k0_) * unsigned(nb_) + du16;                              \
  } while (0)
#define XP_SETUP_LOAD()                                                                                          \
  do {                                                                                                           \
    const BlockSide sd_ = xa.sides[l_bz];                                                                        \
    const int b_ = xa.blocks[l_bz];                                                                              \
    const int vt0_ = 128 * l_bx;                                                                                 \
    const int iv_ = l_bx / nct + 1, jv0_ = 128 * (l_bx % nct) + 1;                                               \
    l_c0 = sd_.s[0].mode == 2 ? min(2 * sd_.s[0].nch, (sd_.s[0].r + 1 + 7) / 8) : 0;                             \
    l_c1 = sd_.s[1].mode == 2 ? min(2 * sd_.s[1].nch, (sd_.s[1].r + 1 + 7) / 8) : 0;                             \
    l_c2 = sd_.s[2].mode == 2 ? min(2 * sd_.s[2].nch, (sd_.s[2].r + 1 + 7) / 8) : 0;                             \
    l_c3 = sd_.s[3].mode == 2 ? min(2 * sd_.s[3].nch, (sd_.s[3].r + 1 + 7) / 8) : 0;                             \
    l_segs = l_c0 + l_c1 + l_c2 + l_c3;                                                                          \
    l_pad = l_segs == 0 ? 2 : (l_segs & 1); /* zero segments that fill the tile's last chunk */                  \
    l_a0 = sd_.s[0].off; l_a1 = sd_.s[1].off; l_a2 = sd_.s[2].off; l_a3 = sd_.s[3].off;                          \
    l_g0 = sd_.s[0].gtab; l_g1 = sd_.s[1].gtab; l_g2 = sd_.s[2].gtab; l_g3 = sd_.s[3].gtab;                      \
    l_n0 = sd_.s[0].nch * BK * 8; l_n1 = sd_.s[1].nch * BK * 8; l_n2 = sd_.s[2].nch * BK * 8; l_n3 = sd_.s[3].nch * BK * 8; \
    const int m0_ = l_by * 64 + 16 * w;                                                                          \
    l_tA = ybytes + size_t(min(m0_, Mc - 1)) * ybytes_row;                                                       \
    lv_a = unsigned(max(0, min(drow, Mc - 1 - m0_))) * ybytes_row + du16;                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                           \
      const int r_ = 32 * w + 16 * q_ + drow;                                                                    \
      if (FLAT) {                                                                                                \
        const int v_ = min(vt0_ + r_, nvert - 1);                                                                \
        lv_i[q_] = v_ / n1 + 1;                                                                                  \
        lv_j[q_] = v_ % n1 + 1;                                                                                  \
      } else {                                                                                                   \
        lv_i[q_] = iv_;                                                                                          \
        lv_j[q_] = min(jv0_ + r_, n1);                                                                           \
      }                                                                                                          \
    }                                                                                                            \
    const int par_ = l_tile & 3;                                                                                 \
    if (w < 2) { /* doubles 32 w .. 32 w + 31 of the tile's h^2 / a_b, one dword per lane */                     \
      const int mb_ = l_by * 64 + w * 32;                                                                        \
      l_ex = ybytes + (size_t(min(mb_, Mc - 1)) * f.nGp + f.sblk0 + b_) * 8;                                     \
      lv_ex = unsigned(max(0, min(lane >> 1, Mc - 1 - mb_))) * ybytes_row + unsigned(lane & 1) * 4u;             \
      l_exlds = XP_SC_OFF + par_ * 512 + w * 256;                                                                \
    } else if (w == 2) { /* W at the tile's 128 vertices, two per lane (what lies behind a row's end is not used) */ \
      const int v0_ = FLAT ? vt0_ : (iv_ - 1) * n1 + (jv0_ - 1);                                                 \
      l_ex = wbytes + size_t(v0_) * 8;                                                                           \
      l_exlds = XP_W_OFF + par_ * 1024;                                                                          \
    }                                                                                                            \
    l_side = -1;                                                                                                 \
    l_left = 0;                                                                                                  \
    if (l_segs > 0) XP_NEXT_SIDE();                                                                              \
  } while (0)
#define XP_DMA16(LDS_, BASE_, VOFF_)                                                                             \
  if (!(DBG & 16)) /* probe: WRONG results, no operand loads */                                                  \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)), \
               "v"(VOFF_), "s"(xp_uniform(BASE_))                                                                \
               : "memory")
  // the 3 loads of the segment under the cursor into half H_ of slot SLOT_, then the cursor moves on
#define XP_ISSUE_HALF(SLOT_, H_)                                                                                 \
  do {                                                                                                           \
    const unsigned sbase_ = lds0 + unsigned(SLOT_) * XP_SLOT_BYTES;                                              \
    if (l_segs > 0) {                                                                                            \
      XP_DMA16(sbase_ + (H_) * 4096 + w * 1024, l_pA, lv_a);                                                     \
      XP_DMA16(sbase_ + 8192 + (H_) * 8192 + (2 * w) * 1024, l_pB, lv_b[0]);                                     \
      XP_DMA16(sbase_ + 8192 + (H_) * 8192 + (2 * w + 1) * 1024, l_pB, lv_b[1]);                                 \
      l_pA += 64;                                                                                                \
      l_pB += 64;                                                                                                \
      --l_segs;                                                                                                  \
      if (--l_left == 0 && l_segs > 0) XP_NEXT_SIDE();                                                           \
    } else if (!l_done) { /* zeros: the odd half of a tile's last chunk */                                        \
      XP_DMA16(sbase_ + (H_) * 4096 + w * 1024, zbase, lv_z);                                                    \
      XP_DMA16(sbase_ + 8192 + (H_) * 8192 + (2 * w) * 1024, zbase, lv_z);                                       \
      XP_DMA16(sbase_ + 8192 + (H_) * 8192 + (2 * w + 1) * 1024, zbase, lv_z);                                   \
      --l_pad;                                                                                                   \
    }                                                                                                            \
  } while (0)
  // the 7 loads of the chunk under the cursor into slot SLOT_.  (Behind the last tile the cursor delivers zeros into
  // slots nobody reads -- the number of loads per chunk must not change.)
#define XP_ISSUE_TAIL()                                                                                          \
  do {                                                                                                           \
    if (!l_first) {} else if (DBG & 16) {} else if (w < 2) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds0 + l_exlds)), "v"(lv_ex), "s"(xp_uniform(l_ex)) : "memory"); \
    else XP_DMA16(lds0 + l_exlds, l_ex, lv_ex);                                                                  \
    l_first = false;                                                                                             \
    if (l_segs == 0 && l_pad <= 0) { /* the tile's last chunk is on its way */                                   \
      if (l_tile + 1 < hi) {                                                                                     \
        ++l_tile;                                                                                                \
        if (++l_bx == ntile) {                                                                                   \
          l_bx = 0;                                                                                              \
          if (++l_by == mt) { l_by = 0; ++l_bz; }                                                                \
        }                                                                                                        \
        XP_SETUP_LOAD();                                                                                         \
        l_first = true;                                                                                          \
      } else { /* nothing left */                                                                                \
        l_pad = 1 << 30;                                                                                         \
        l_done = true;                                                                                           \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#define XP_ISSUE(SLOT_)                                                                                          \
  do {                                                                                                           \
    XP_ISSUE_HALF(SLOT_, 0);                                                                                     \
    XP_ISSUE_HALF(SLOT_, 1);                                                                                     \
    XP_ISSUE_TAIL();                                                                                             \
  } while (0)

  d4_t acc[4][2], res[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = res[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
  // epilogue state of the tile being multiplied (c_*) and of the one being stored (p_*)
  unsigned c_v0 = 0, c_v1 = 0;  // lane: (kq * dim + position in the row) * 8
  bool c_k16 = false, c_k8 = false, c_k8b = false;  // 16-byte pair / single first / single second
  int c_mbase = 0, c_par = 0, p_mbase = 0, p_par = 0;
  double p_w0 = 0.0, p_w1 = 0.0;
  unsigned long long p_m16 = 0, p_m8 = 0, p_m8b = 0;  // the three store instructions' lanes (before the row limit)
  bool p_full = true;                                  // all 64 systems of the stored tile exist
  char* p_sp = ubytes;                                 // lane: where the NEXT store of the stored tile goes
  long long p_d1 = 0;                                  // lane (FLAT): second vertex - first vertex, bytes
  const long long rb16 = (long long)(16 * row_bytes), rbm44 = -(long long)(44 * row_bytes);
  bool pending = false;  // `res` / p_* hold a tile that is not stored yet
  // (!FLAT) a mesh row ends in a lane's FIRST vertex only when n1 is odd, and only in the wave that holds the row's end:
  // the other waves issue one store instruction per k-step instead of two
  const bool tail_wave = FLAT || ((n1 & 1) && w == (((n1 - 1) & 127) >> 5));
  // The KS store instructions of (16-system group II_, row G_ of the lanes' four), in the order II_ = 0..3 fastest:
  // the lanes' pointer walks + 16 rows three times, then - 44.  II_, G_ static; ON_ uniform.
#define XP_STORE(II_, G_, ON_)                                                                                    \
  do {                                                                                                            \
    if (DBG & 2) break; /* probe: no epilogue at all (the accumulators are never reset, one dummy store at the end) */ \
    const double sc_ = *reinterpret_cast<const double*>(xp_lds + XP_SC_OFF + p_par * 512 + ((II_) * 16 + 4 * (G_) + kq) * 8); \
    const double x0_ = res[II_][0][G_] + sc_ * p_w0, x1_ = res[II_][1][G_] + sc_ * p_w1;                          \
    /* even lanes: (own x0, neighbour's x0); odd lanes: (neighbour's x1, own x1) */                                \
    const double got_ = lane_swap1(odd ? x0_ : x1_);                                                              \
    const double2_u pr_ = double2_u{odd ? got_ : x0_, odd ? x1_ : got_};                                          \
    unsigned long long m16_ = (ON_) && !(DBG & 1) ? p_m16 : 0ull, m8_ = (ON_) && !(DBG & 1) ? p_m8 : 0ull;        \
    unsigned long long m8b_ = (ON_) && !(DBG & 1) ? p_m8b : 0ull;                                                 \
    if (!p_full) {                                                                                                \
      const unsigned long long in_ = __builtin_amdgcn_ballot_w64(p_mbase + (II_) * 16 + 4 * (G_) + kq < Mc);      \
      m16_ &= in_; m8_ &= in_; m8b_ &= in_;                                                                       \
    }                                                                                                             \
    unsigned long long sv_;                                                                                       \
    if (FLAT) {                                                                                                   \
      asm volatile(                                                                                               \
          "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %4, %6, off\n\t"                      \
          "s_mov_b64 exec, %2\n\tglobal_store_dwordx2 %4, %7, off\n\t"                                            \
          "s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %5, %8, off\n\ts_mov_b64 exec, %0"                          \
          : "=&s"(sv_)                                                                                            \
          : "s"(m16_), "s"(m8_), "s"(m8b_), "v"(p_sp), "v"(p_sp + p_d1), "v"(pr_), "v"(pr_.x), "v"(pr_.y));       \
    } else if (tail_wave) {                                                                                       \
      asm volatile(                                                                                               \
          "s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %3, %4, off\n\t"                      \
          "s_mov_b64 exec, %2\n\tglobal_store_dwordx2 %3, %5, off\n\ts_mov_b64 exec, %0"                          \
          : "=&s"(sv_)                                                                                            \
          : "s"(m16_), "s"(m8_), "v"(p_sp), "v"(pr_), "v"(pr_.x));                                                \
    } else {                                                                                                      \
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx4 %2, %3, off\n\ts_mov_b64 exec, %0" \
                   : "=&s"(sv_)                                                                                   \
                   : "s"(m16_), "v"(p_sp), "v"(pr_));                                                             \
    }                                                                                                             \
    p_sp += (II_) == 3 ? rbm44 : rb16;                                                                            \
  } while (0)
  // one chunk: U_ = its number within the tile mod 4 (static: which of the stored tile's rows it carries), ST_ uniform
#define XP_CHUNK(U_, ST_)                                                                                          \
  do {                                                                                                             \
    const int slot = g % XP_SLOTS, nslot = (g + 1) % XP_SLOTS, fslot = (g + 2) % XP_SLOTS;                         \
    XP_STAMP(0);                                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                             \
      const int pb_ = kk & 1;                                                                                      \
      if (kk == 2) {                                                                                               \
        XP_STAMP(1);                                                                                               \
        /* chunk g + 1 (requested in the second half of the previous chunk) has landed when at most the stores of  \
           the last 3 k-steps are still in flight; behind the barrier it is in LDS for everybody, and nobody reads \
           chunk g - 1 any more */                                                                                 \
        if (FLAT) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");                               \
        else if (tail_wave) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");                          \
        else asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");                                         \
        XP_STAMP(2);                                                                                               \
      }                                                                                                            \
      if (kk < 3) XP_FRAGS(slot, kk + 1, af[pb_ ^ 1], bf[pb_ ^ 1]);                                                \
      else XP_FRAGS(nslot, 0, af[pb_ ^ 1], bf[pb_ ^ 1]);                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pb_][i], bf[pb_][j], acc[i][j], 0, 0, 0);        \
      /* chunk g + 2 is requested behind the MFMAs of the two k-steps after the barrier (3 + 4 loads) */           \
      if (kk == 2) XP_ISSUE_HALF(fslot, 0);                                                                        \
      if (kk == 3) {                                                                                               \
        XP_ISSUE_HALF(fslot, 1);                                                                                   \
        XP_ISSUE_TAIL();                                                                                           \
        XP_STAMP(4);                                                                                               \
      }                                                                                                            \
      /* one of the stored tile's sixteen store groups per k-step: rows 4 U_ + kq of 16-system group kk */         \
      if (kk == 0) XP_STORE(0, U_, ST_);                                                                           \
      if (kk == 1) XP_STORE(1, U_, ST_);                                                                           \
      if (kk == 2) XP_STORE(2, U_, ST_);                                                                           \
      if (kk == 3) XP_STORE(3, U_, ST_);                                                                           \
    }                                                                                                              \
    ++g;                                                                                                           \
  } while (0)

  // ---- prologue: chunks 0 and 1 requested and landed, chunk 0's first fragments read
  XP_SETUP_LOAD();
  XP_ISSUE(0);
  XP_ISSUE(1);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  double af[2][4], bf[2][2];
  XP_FRAGS(0, 0, af[0], bf[0]);
  int g = 0;  // chunks done: chunk g lives in slot g % XP_SLOTS
  int c_bx = lo % ntile, c_by = (lo / ntile) % mt, c_bz = lo / (ntile * mt);
  for (int t = lo; t < hi; ++t) {
    // ---- tile begin: what its epilogue (one tile later) will need
    int tot;
    {
      const int b = xa.blocks[c_bz];
      const int p = b / f.ncb, q = b % f.ncb;
      const BlockSide sd = xa.sides[c_bz];
      int nseg = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) nseg += sd.s[s].mode == 2 ? min(2 * sd.s[s].nch, (sd.s[s].r + 1 + 7) / 8) : 0;
      tot = max(1, (nseg + 1) / 2);
      const int iv = c_bx / nct + 1, jv0 = 128 * (c_bx % nct) + 1, vt0 = 128 * c_bx;
      const int tl = w * 32 + (odd ? 16 : 0) + fr - (odd ? 1 : 0);  // first of the lane's two vertices, tile-local
      long long off0, off1;
      bool ok0, ok1;
      if (FLAT) {
        const int v = vt0 + tl, i0 = v / n1, j0 = v - i0 * n1;
        off0 = (long long)(p * N + i0) * f.nc + q * N + j0;
        off1 = j0 + 1 < n1 ? off0 + 1 : (long long)(p * N + i0 + 1) * f.nc + q * N;
        ok0 = v < nvert;
        ok1 = v + 1 < nvert;
      } else {
        const int jcol = jv0 + tl;  // 1-based
        off0 = (long long)(p * N + iv - 1) * f.nc + q * N - 1 + jcol;
        off1 = off0 + 1;
        ok0 = jcol <= n1;
        ok1 = jcol + 1 <= n1;
      }
      const bool straddle = FLAT && off1 != off0 + 1;  // the pair lies in two mesh rows
      c_k16 = ok1 && !straddle;
      c_k8 = ok0 && (!ok1 || straddle);
      c_k8b = ok1 && straddle;
      c_v0 = unsigned((size_t(kq) * size_t(f.dim) + size_t(off0)) * 8);  // (the host checks that this fits 32 bits)
      c_v1 = unsigned((size_t(kq) * size_t(f.dim) + size_t(off1)) * 8);
      c_mbase = c_by * 64;
      c_par = t & 3;
    }
    for (int c4 = 0; c4 < tot; c4 += 4) {
      const bool st = pending && c4 == 0;
      const int cc = c4;
      (void)cc;
      XP_CHUNK(0, st);
      if (c4 + 1 < tot) XP_CHUNK(1, st);
      if (c4 + 2 < tot) XP_CHUNK(2, st);
      if (c4 + 3 < tot) XP_CHUNK(3, st);
    }
    // ---- tile end
    if (pending && tot < 4) {  // (fewer than four chunks: the rest of the previous tile's stores)
      if (tot < 2) { XP_STORE(0, 1, true); XP_STORE(1, 1, true); XP_STORE(2, 1, true); XP_STORE(3, 1, true); }
      if (tot < 3) { XP_STORE(0, 2, true); XP_STORE(1, 2, true); XP_STORE(2, 2, true); XP_STORE(3, 2, true); }
      XP_STORE(0, 3, true); XP_STORE(1, 3, true); XP_STORE(2, 3, true); XP_STORE(3, 3, true);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        res[i][j] = acc[i][j];
        if (!(DBG & 2)) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
      }
    p_mbase = c_mbase;
    p_par = c_par;
    p_full = c_mbase + 64 <= Mc;
    p_m16 = __builtin_amdgcn_ballot_w64(c_k16);
    p_m8 = __builtin_amdgcn_ballot_w64(c_k8);
    p_m8b = __builtin_amdgcn_ballot_w64(c_k8b);
    p_sp = ubytes + size_t(row0 + c_mbase) * row_bytes + c_v0;
    p_d1 = (long long)c_v1 - (long long)c_v0;
    {  // W at this lane's accumulator columns (it came with the tile's chunks)
      const char* wb = xp_lds + XP_W_OFF + c_par * 1024 + (w * 32 + fr) * 8;
      p_w0 = *reinterpret_cast<const double*>(wb);
      p_w1 = *reinterpret_cast<const double*>(wb + 128);
    }
    pending = true;
    if (++c_bx == ntile) {
      c_bx = 0;
      if (++c_by == mt) { c_by = 0; ++c_bz; }
    }
  }
  // the last tile's stores
  XP_STORE(0, 0, true); XP_STORE(1, 0, true); XP_STORE(2, 0, true); XP_STORE(3, 0, true);
  XP_STORE(0, 1, true); XP_STORE(1, 1, true); XP_STORE(2, 1, true); XP_STORE(3, 1, true);
  XP_STORE(0, 2, true); XP_STORE(1, 2, true); XP_STORE(2, 2, true); XP_STORE(3, 2, true);
  XP_STORE(0, 3, true); XP_STORE(1, 3, true); XP_STORE(2, 3, true); XP_STORE(3, 3, true);
  if (DBG & 2) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sum += res[i][j][0] + res[i][j][1] + res[i][j][2] + res[i][j][3];
    if (sum == 1.2345e-300) U[0] = sum;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last loads still target this workgroup's LDS)
#undef XP_CHUNK
#undef XP_STORE
#undef XP_ISSUE
#undef XP_ISSUE_TAIL
#undef XP_ISSUE_HALF
#undef XP_DMA16
#undef XP_SETUP_LOAD
#undef XP_NEXT_SIDE
#undef XP_FRAGS
}
template __global__ void k_extend_p<false, 0>(FemDev, X128Args, const double*, int, double*, long long, int);
template __global__ void k_extend_p<true, 0>(FemDev, X128Args, const double*, int, double*, long long, int);
#ifdef ROMHC_XP_PROBES  // (make EXTRA=-DROMHC_XP_PROBES; ROMHC_EXT_P=2 / 3 / 17)
template __global__ void k_extend_p<false, 1>(FemDev, X128Args, const double*, int, double*, long long, int);
template __global__ void k_extend_p<false, 2>(FemDev, X128Args, const double*, int, double*, long long, int);
template __global__ void k_extend_p<false, 16>(FemDev, X128Args, const double*, int, double*, long long, int);
#endif
