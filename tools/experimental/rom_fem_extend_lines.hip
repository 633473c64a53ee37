// k_extend_lines: the harmonic extension written in WHOLE CACHE LINES (round 4).
//
// Why.  k_extend128 stores its 128 x 128 tile straight from the MFMA accumulator layout: one wave instruction = 4 systems x
// 256 bytes, and a block's run of a mesh row (n1 doubles inside a row of nc) starts at an arbitrary multiple of 8 bytes, so
// every piece touches three 128-byte lines and every run begins and ends in a line that somebody else completes later.  Store
// only, that pattern reaches 2.9 TB/s on the part (0.18 ms for the 528 MB of a 256 x 256 / 2 x 2 / 1024-system step; the whole
// kernel takes 0.205) -- line-aligned kilobytes with the partial line carried over to the next piece: 4.7-4.9 TB/s
// (tools/ext_store_patterns2.hip, profiles/r04_store_patterns_line_aligned.txt).  The kernel is bound by HOW it stores.
//
// What.  A workgroup owns 64 systems x R consecutive FULL mesh rows of one block row p: for every system a contiguous stretch
// of the snapshot row (src/lib/SolutionsManagers.py:64-68 fixes that layout: row-major inner vertices).  It walks the stretch
// in 128-slot tiles -- the n1 interior vertices of block (p, q) in mesh row i followed by the interface vertex to block
// (p, q + 1): N = n1 + 1 slots per block, whole tiles when N is a multiple of 128 -- and per tile
//   * runs the same MFMA k loop as k_extend128 (same operands in the same order: the rows are bit-identical), operand
//     chunks by LDS-DMA into a ring of three slots, chunk c + 2 in flight under chunk c;
//   * parks the finished 64 x 128 tile in an LDS staging area whose rows are [16 carried doubles | 128 of the tile];
//   * and, while the NEXT tile multiplies, writes every system's row out as ONE line-aligned kilobyte (the doubles left
//     over from the previous tile + as many of this one as complete lines), moving the remainder (< 16 doubles) to the
//     carry slots.  Only the two ends of a stretch are partial lines.
// The interface vertices between the blocks of a row are the values of the vertical edges (u_f = P z + p0 / s): the
// workgroup computes the 64 x R values it needs itself, once, with the very MFMA chain of expand_tile (same bits), so that
// the run of a system stays contiguous across blocks; the expansion keeps those values out of the snapshot rows (skipv).
//
// Roles.  Eight waves multiply (2 x 4 wave tiles of 32 systems x 32 slots, two accumulator sets: tile t + 1 starts while
// tile t is parked).  Waves 0-3 also issue every operand load, waves 4-7 every store: vmcnt retires loads and stores of a
// wave in order, so a wave that did both would wait for its stores whenever it waits for an operand.  One barrier per
// 16-wide K chunk, placed before the last of its four k-steps (the fragments of that step are already in registers: no
// bubble at the chunk boundary); loaders wait with a counted vmcnt (the youngest chunk stays in flight).
// grid = (nrb x row chunks x system groups), 512 threads, one workgroup per CU (~157 KB of LDS).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "rom_fem_dev.h"

namespace {

constexpr int XL_S = 64;                                  // systems per workgroup
constexpr int XL_SR = 144;                                // staging row (doubles): 16 carried + 128 of the tile
constexpr int XL_SLOT = 2 * 4096 + 2 * 8192;              // {A k 0..7 | A k 8..15 | B k 0..7 | B k 8..15}: 24,576 B
constexpr int XL_NSLOT = 3;
constexpr int XL_STAGE_OFF = XL_NSLOT * XL_SLOT;          // 73,728
constexpr int XL_W_OFF = XL_STAGE_OFF + XL_S * XL_SR * 8;  // 147,456
constexpr int XL_W_BYTES = 3 * 1024;                      // particular solution at the slots of three tiles in flight
constexpr int XL_SMALL_OFF = XL_W_OFF + XL_W_BYTES;       // h^2/a_b | interface-vertex values | K segment lists
constexpr int XL_MAXSEG = 32;
constexpr int XL_SEG_BYTES = 16 + XL_MAXSEG * 16;         // per block: {segments, chunks, -, -} + {aoff, bseg, side, -} each

__device__ inline const char* xl_uniform(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const char*>((unsigned long long)hi << 32 | lo);
}
__device__ inline int xl_u(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef double double2_a16 __attribute__((ext_vector_type(2), aligned(16)));

}  // namespace

size_t rom_extend_lines_lds_bytes(int ncb, int R) {
  return size_t(XL_SMALL_OFF) + size_t(ncb) * XL_S * 8 + size_t(ncb - 1) * XL_S * R * 8 + size_t(ncb) * XL_SEG_BYTES;
}
int rom_extend_lines_max_rows(int ncb, size_t lds_limit) {
  if (ncb <= 1) return 16;
  const size_t fixed = rom_extend_lines_lds_bytes(ncb, 0);
  if (fixed >= lds_limit) return 0;
  return int(std::min<size_t>(16, (lds_limit - fixed) / (size_t(ncb - 1) * XL_S * 8)));
}

#define XL_DMA(LDS_, BASE_, VOFF_)                                                                                 \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),   \
               "v"(VOFF_), "s"(xl_uniform(BASE_))                                                                  \
               : "memory", "m0")
#define XL_DMA1(LDS_, BASE_, VOFF_)                                                                                \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(LDS_)),     \
               "v"(VOFF_), "s"(xl_uniform(BASE_))                                                                  \
               : "memory", "m0")

__global__ __launch_bounds__(512, 1) void k_extend_lines(FemDev f, XLArgs g, int Mc, double* __restrict__ U, long long row0) {
  extern __shared__ __align__(16) char lds[];
  const unsigned lds0 = unsigned(size_t((__attribute__((address_space(3))) char*)lds));
  const int n1 = f.n1, N = f.N, ncb = f.ncb, R = g.R;
  const int lane = threadIdx.x & 63, w = xl_u(threadIdx.x >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  int ck, sg, p;
  {
    int id = blockIdx.x;
    if (g.order == 0) { ck = id % g.nchunk; id /= g.nchunk; sg = id % g.nsg; p = id / g.nsg; }
    else { sg = id % g.nsg; id /= g.nsg; ck = id % g.nchunk; p = id / g.nchunk; }
  }
  const int m0 = sg * XL_S;
  const int r0 = ck * R, r1 = min(n1, r0 + R);  // block-local mesh rows (0-based)
  const int ntq = N / 128;                      // tiles per block run
  const int nt = max(0, r1 - r0) * ncb * ntq;
  double* const staging = reinterpret_cast<double*>(lds + XL_STAGE_OFF);
  const double* const wring = reinterpret_cast<const double*>(lds + XL_W_OFF);
  double* const scs = reinterpret_cast<double*>(lds + XL_SMALL_OFF);  // [ncb][64]
  double* const stash = scs + ncb * XL_S;                             // [ncb - 1][64][R]
  char* const segs = reinterpret_cast<char*>(stash + (ncb - 1) * XL_S * R);

  if (nt > 0) {
    // ---------------------------------------------------------------- prologue
    if (threadIdx.x < unsigned(ncb * XL_MAXSEG)) {  // K segment lists of the blocks of this block row
      const int q = threadIdx.x >> 5, e = threadIdx.x & 31;
      const BlockSide* sd = f.sides + (p * ncb + q);
      int nseg = 0;
      int4 rec = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // (same walk as k_extend128: side by side, ceil((rank + 1) / 8) segments each)
        const int cnt = sd->s[s].mode == 2 ? min(2 * sd->s[s].nch, (sd->s[s].r + 1 + 7) / 8) : 0;
        const int qq = e - nseg;
        if (qq >= 0 && qq < cnt) rec = int4{sd->s[s].off + 8 * qq, sd->s[s].gseg + qq * (n1 * n1 * 8), s, 0};
        nseg += cnt;
      }
      *reinterpret_cast<int4*>(segs + q * XL_SEG_BYTES + 16 + e * 16) = rec;
      if (e == 0) *reinterpret_cast<int4*>(segs + q * XL_SEG_BYTES) = int4{nseg, (nseg + 1) / 2, 0, 0};
    }
    if (threadIdx.x < unsigned(ncb * XL_S)) {  // h^2 / a_b of the systems, per block
      const int q = threadIdx.x >> 6, s = threadIdx.x & 63, m = m0 + s;
      scs[q * XL_S + s] = m < Mc ? f.y[size_t(m) * f.nGp + f.sblk0 + (p * ncb + q)] : 0.0;
    }
    // values of the vertical edges at the rows of this stretch: the MFMA chain of expand_tile (k ascending, then + p0 / s)
    for (int u = w; u < (ncb - 1) * 4; u += 8) {
      const int qv = u >> 2, i4 = u & 3;
      const ExpEdge ee = f.exp[f.vexp[p * (ncb - 1) + qv]];
      const int msys = min(m0 + 16 * i4 + fr, Mc - 1);
      const int node = min(r0 + fr, n1 - 1);
      const double* pa = f.y + size_t(msys) * f.nGp + ee.zpos + kq;
      const double* pb = f.P + (size_t(ee.ptab) * f.n1p + node) * f.n1p + kq;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < ee.nch * BK; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], pb[k], acc, 0, 0, 0);
      const double p0n = f.vec[ee.p0off + node];
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) {
        const int sys = 16 * i4 + 4 * g2 + kq, m = m0 + sys;
        const double inv = m < Mc ? f.y[size_t(m) * f.nGp + ee.spos] : 0.0;
        const double v = __builtin_fma(p0n, inv, acc[g2]);  // (fused, as in expand_tile)
        if (fr < R) stash[(qv * XL_S + sys) * R + fr] = v;
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- per-wave constants
    // (everything below is macros over plain locals: closures of nested lambdas ended up in scratch memory, and scratch
    // loads count in vmcnt)
    const int wr = w >> 2, wc = w & 3;
    const unsigned fx = unsigned((kq >> 1) ^ (fr >> 2));
    const unsigned fa0 = unsigned(fr * 64) + (fx << 4) + unsigned(kq & 1) * 8u;
    const unsigned fa1 = unsigned(fr * 64) + ((fx ^ 2u) << 4) + unsigned(kq & 1) * 8u;
    // fragments of k-step KKI of the chunk in the slot at byte offset SB
#define XL_FRAGS(SB_, KKI_, AF_, BF_)                                                                              \
  if (!(g.dbg & 32)) do {                                                                                          \
    const char* pf_ = lds + (SB_) + (((KKI_)&1) ? fa1 : fa0);                                                      \
    const char* pa_ = pf_ + ((KKI_) >> 1) * 4096 + wr * 2048;                                                      \
    const char* pb_ = pf_ + 8192 + ((KKI_) >> 1) * 8192 + wc * 2048;                                               \
    AF_[0] = *reinterpret_cast<const double*>(pa_);                                                                \
    AF_[1] = *reinterpret_cast<const double*>(pa_ + 1024);                                                         \
    BF_[0] = *reinterpret_cast<const double*>(pb_);                                                                \
    BF_[1] = *reinterpret_cast<const double*>(pb_ + 1024);                                                         \
  } while (0)
#define XL_MFMA4(ACC_, AF_, BF_)                                                                                   \
  do {                                                                                                             \
    ACC_[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF_[0], BF_[0], ACC_[0][0], 0, 0, 0);                        \
    ACC_[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF_[0], BF_[1], ACC_[0][1], 0, 0, 0);                        \
    ACC_[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF_[1], BF_[0], ACC_[1][0], 0, 0, 0);                        \
    ACC_[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(AF_[1], BF_[1], ACC_[1][1], 0, 0, 0);                        \
  } while (0)

    // lane q holds the number of K segments of block q (read with readlane: no LDS round trip at a tile switch)
    const int segcnt_v = lane < ncb ? *reinterpret_cast<const int*>(segs + lane * XL_SEG_BYTES) : 0;
    // total number of chunks of this stretch
    int ctotal = 0;
    for (int q = 0; q < ncb; ++q) ctotal += (__builtin_amdgcn_readlane(segcnt_v, q) + 1) / 2;
    ctotal *= (r1 - r0) * ntq;

    // ---------------------------------------------------------------- loader state (waves 0-3)
    const bool loader = w < 4;
    const int lw = w & 3;
    const unsigned du16 = unsigned(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    const unsigned ybytes_row = unsigned(f.nGp) * 8u;
    const unsigned voA = unsigned(max(0, min(16 * lw + (lane >> 2), Mc - 1 - m0))) * ybytes_row + du16;
    const char* const ybase = reinterpret_cast<const char*>(f.y) + size_t(m0) * ybytes_row;
    const char* const gsbase = reinterpret_cast<const char*>(f.Gs);
    const char* const zbase = reinterpret_cast<const char*>(f.W + size_t(n1) * n1);  // EXT_ZERO_PAGE doubles of zeros
    const char* const wbase = reinterpret_cast<const char*>(f.W);
    const unsigned voZ = unsigned(lane) * 16u;
    int L_tile = 0, L_r = r0, L_q = 0, L_h = 0, L_e = 0, L_nseg = 0, L_first = 1, L_w3 = 0;
    int L_issued = 0, L_lastsz = 0;
    int4 N_rec0 = {0, 0, 0, 0}, N_rec1 = {0, 0, 0, 0};  // segment records of the NEXT batch's two halves (fetched a chunk ahead)
    int N_v0 = 0, N_v1 = 0;
#define XL_PREFETCH_RECS()                                                                                         \
  do {                                                                                                             \
    N_v0 = L_tile < nt && L_e < L_nseg;                                                                            \
    N_v1 = L_tile < nt && L_e + 1 < L_nseg;                                                                        \
    if (N_v0) N_rec0 = *reinterpret_cast<const int4*>(segs + L_q * XL_SEG_BYTES + 16 + L_e * 16);                  \
    if (N_v1) N_rec1 = *reinterpret_cast<const int4*>(segs + L_q * XL_SEG_BYTES + 16 + (L_e + 1) * 16);            \
  } while (0)
    unsigned vo02_0 = 0, vo02_1 = 0, vo1_0 = 0, vo1_1 = 0, vo3_0 = 0, vo3_1 = 0;  // lane offsets into a table, by side kind
    // lane offsets of the loader's tile.  Row of a side's segment-major table (k_repack_table), i / j the 1-based interior
    // indices: side 0, 2: n1 (i - 1) + (j - 1); 1: n1 (n1 - i) + (j - 1); 3: n1 (i - 1) + (n1 - j)
#define XL_LOADER_TILE()                                                                                           \
  do {                                                                                                             \
    L_nseg = __builtin_amdgcn_readlane(segcnt_v, L_q);                                                             \
    const int vi_ = L_r + 1;                                                                                       \
    const int vj0_ = min(128 * L_h + 1 + 32 * lw + (lane >> 2), n1), vj1_ = min(128 * L_h + 1 + 32 * lw + 16 + (lane >> 2), n1); \
    vo02_0 = unsigned(n1 * (vi_ - 1) + (vj0_ - 1)) * 64u + du16;                                                   \
    vo02_1 = unsigned(n1 * (vi_ - 1) + (vj1_ - 1)) * 64u + du16;                                                   \
    vo1_0 = unsigned(n1 * (n1 - vi_) + (vj0_ - 1)) * 64u + du16;                                                   \
    vo1_1 = unsigned(n1 * (n1 - vi_) + (vj1_ - 1)) * 64u + du16;                                                   \
    vo3_0 = unsigned(n1 * (vi_ - 1) + (n1 - vj0_)) * 64u + du16;                                                   \
    vo3_1 = unsigned(n1 * (vi_ - 1) + (n1 - vj1_)) * 64u + du16;                                                   \
  } while (0)
    // one half (8-wide K segment) of the next chunk into the slot at byte offset SB
#define XL_ISSUE_HALF(SB_, H_, VALID_, REC_)                                                                       \
  do {                                                                                                             \
    const unsigned dA_ = lds0 + (SB_) + unsigned(H_) * 4096u + unsigned(lw) * 1024u;                               \
    const unsigned dB_ = lds0 + (SB_) + 8192u + unsigned(H_) * 8192u + unsigned(lw) * 2048u;                       \
    if (VALID_) {                                                                                                  \
      const int aoff_ = xl_u(REC_.x), bseg_ = xl_u(REC_.y), side_ = xl_u(REC_.z);                                  \
      const char* pA_ = ybase + size_t(aoff_) * 8;                                                                 \
      const char* pB_ = gsbase + size_t(bseg_) * 8;                                                                \
      const unsigned b0_ = side_ == 1 ? vo1_0 : side_ == 3 ? vo3_0 : vo02_0;                                       \
      const unsigned b1_ = side_ == 1 ? vo1_1 : side_ == 3 ? vo3_1 : vo02_1;                                       \
      XL_DMA(dA_, pA_, voA);                                                                                       \
      XL_DMA(dB_, pB_, b0_);                                                                                       \
      XL_DMA(dB_ + 1024u, pB_, b1_);                                                                               \
      ++L_e;                                                                                                       \
    } else { /* zeros: the odd half of a block's last chunk */                                                     \
      XL_DMA(dA_, zbase, voZ);                                                                                     \
      XL_DMA(dB_, zbase, voZ);                                                                                     \
      XL_DMA(dB_ + 1024u, zbase, voZ);                                                                             \
    }                                                                                                              \
  } while (0)
    // the loads of the next chunk of the stretch (if there is one) into the slot at byte offset SB
#define XL_ISSUE_BATCH(SB_)                                                                                        \
  do {                                                                                                             \
    if (L_tile < nt && !((g.dbg & 8) && L_issued >= 3)) {                                                                                             \
      XL_ISSUE_HALF(SB_, 0, N_v0, N_rec0);                                                                         \
      XL_ISSUE_HALF(SB_, 1, N_v1, N_rec1);                                                                         \
      L_lastsz = 6;                                                                                                \
      if (L_first) { /* the particular solution at the tile's slots: 32 doubles per loader wave */                 \
        const char* src_ = wbase + (size_t(L_r) * n1 + 128 * L_h + 32 * lw) * 8;                                   \
        XL_DMA1(lds0 + unsigned(XL_W_OFF) + unsigned(L_w3) * 1024u + unsigned(lw) * 256u, src_, unsigned(lane) * 4u); \
        L_first = 0;                                                                                               \
        L_lastsz = 7;                                                                                              \
      }                                                                                                            \
      ++L_issued;                                                                                                  \
      if (L_e >= L_nseg) { /* next tile */                                                                         \
        ++L_tile;                                                                                                  \
        if (++L_h == ntq) { L_h = 0; if (++L_q == ncb) { L_q = 0; ++L_r; } }                                       \
        L_e = 0;                                                                                                   \
        L_first = 1;                                                                                               \
        L_w3 = L_w3 == 2 ? 0 : L_w3 + 1;                                                                           \
        if (L_tile < nt) XL_LOADER_TILE();                                                                         \
      }                                                                                                            \
      XL_PREFETCH_RECS();                                                                                          \
    }                                                                                                              \
  } while (0)

    // ---------------------------------------------------------------- storer state (waves 4-7)
    const bool storer = w >= 4;
    const int sw = w & 3;
    const unsigned long long ubase_d = reinterpret_cast<unsigned long long>(U) >> 3;
    int F_active = 0, F_next = 16, F_gpos = 0, F_nvalid = 128, F_first = 0;
    const int gs0 = (p * N + r0) * f.nc;  // first position of the stretch in a snapshot row
    // Write out the staged rows, two systems per opportunity, LDS reads one opportunity ahead of the stores that use them
    // (a wave issues in order: waiting for its own ds_read in front of a store would hold up its MFMAs as well).
    // LOAD: the pair of doubles lane L sends to the line-aligned window of the row, and the double it moves to the carry
    int F_pend = 0;
    double2_u F_v0 = {0.0, 0.0}, F_v1 = {0.0, 0.0};
    double F_t0 = 0.0, F_t1 = 0.0;
    int F_s0 = 0, F_s1 = 0, F_c0 = 0, F_c1 = 0, F_ok0 = 0, F_ok1 = 0;
#define XL_FLUSH_LOAD(V_, T_, S_, C_, OK_, U_)                                                                     \
  do {                                                                                                             \
    S_ = 16 * sw + (U_);                                                                                           \
    OK_ = m0 + S_ < Mc;                                                                                            \
    if (OK_) {                                                                                                     \
      const long long rowd_ = (row0 + m0 + S_) * f.dim;                                                            \
      C_ = int((ubase_d + (unsigned long long)rowd_ + (unsigned long long)F_gpos) & 15ull);                        \
      const int totd_ = C_ + F_nvalid, nl_ = totd_ >> 4, rem_ = totd_ & 15;                                        \
      const double* srow_ = staging + S_ * XL_SR + 16 - C_;                                                        \
      if (lane < 8 * nl_) V_ = *reinterpret_cast<const double2_u*>(srow_ + 2 * lane);                              \
      if (lane < rem_) T_ = srow_[16 * nl_ + lane];                                                                \
    }                                                                                                              \
  } while (0)
#define XL_FLUSH_STORE(V_, T_, S_, C_, OK_)                                                                        \
  do {                                                                                                             \
    if (OK_) {                                                                                                     \
      const long long rowd_ = (row0 + m0 + S_) * f.dim;                                                            \
      const int totd_ = C_ + F_nvalid, nl_ = totd_ >> 4, rem_ = totd_ & 15;                                        \
      double* const dst_ = U + rowd_ + F_gpos - C_;                                                                \
      if (g.dbg & 1) {                                                                                             \
      } else if (!F_first) {                                                                                       \
        if (lane < 8 * nl_) *reinterpret_cast<double2_a16*>(dst_ + 2 * lane) = double2_a16{V_.x, V_.y};            \
      } else if (lane < 8 * nl_) { /* first tile of the stretch: what lies before it belongs to somebody else */   \
        if (2 * lane >= C_) *reinterpret_cast<double2_a16*>(dst_ + 2 * lane) = double2_a16{V_.x, V_.y};            \
        else if (2 * lane + 1 == C_) dst_[C_] = V_.y;                                                              \
      }                                                                                                            \
      if (lane < rem_) staging[S_ * XL_SR + 16 - rem_ + lane] = T_;                                                \
    }                                                                                                              \
  } while (0)
#define XL_FLUSH_OPP()                                                                                             \
  do {                                                                                                             \
    if (storer && F_active && !(g.dbg & 2)) {                                                                      \
      if (F_pend) {                                                                                                \
        XL_FLUSH_STORE(F_v0, F_t0, F_s0, F_c0, F_ok0);                                                             \
        XL_FLUSH_STORE(F_v1, F_t1, F_s1, F_c1, F_ok1);                                                             \
        F_pend = 0;                                                                                                \
      }                                                                                                            \
      if (F_next < 16) {                                                                                           \
        XL_FLUSH_LOAD(F_v0, F_t0, F_s0, F_c0, F_ok0, F_next);                                                      \
        XL_FLUSH_LOAD(F_v1, F_t1, F_s1, F_c1, F_ok1, F_next + 1);                                                  \
        F_next += 2;                                                                                               \
        F_pend = 1;                                                                                                \
      }                                                                                                            \
    }                                                                                                              \
  } while (0)

    // ---------------------------------------------------------------- multiply
    int T_r = r0, T_q = 0, T_h = 0, T_w3 = 0;   // the tile being multiplied
    int P_r = 0, P_q = 0, P_h = 0, P_w3 = 0;   // the tile before it (parked at the start of the current one)
    int cglob = 0, slot = 0;                    // chunk counter of the stretch, its slot (cglob % 3)
    double af0[2], bf0[2], af1[2], bf1[2];
    // park a finished tile: + (h^2 / a_b) W, interface vertex from the stash
#define XL_PARK(ACC_)                                                                                              \
  do {                                                                                                             \
    /* every LDS read first (the compiler keeps reads behind the staging writes they might alias): one wait */     \
    const double* wl_ = wring + P_w3 * 128 + wc * 32 + fr;                                                         \
    const double w0_ = wl_[0], w1_ = wl_[16];                                                                      \
    const bool iface_ = P_q < ncb - 1 && P_h == ntq - 1 && wc == 3;                                                \
    double sc_[8], sv_[8];                                                                                         \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                                             \
      const int sys_ = wr * 32 + 16 * (e_ >> 2) + 4 * (e_ & 3) + kq;                                               \
      sc_[e_] = scs[P_q * XL_S + sys_];                                                                            \
      sv_[e_] = iface_ ? stash[(P_q * XL_S + sys_) * R + (P_r - r0)] : 0.0;                                        \
    }                                                                                                              \
    if (!(g.dbg & 4)) {                                                                                            \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                                             \
      const int i_ = e_ >> 2, g2_ = e_ & 3;                                                                        \
      const int sys_ = wr * 32 + 16 * i_ + 4 * g2_ + kq;                                                           \
      const double x0_ = __builtin_fma(sc_[e_], w0_, ACC_[i_][0][g2_]); /* (fused, as k_extend128's) */            \
      double x1_ = __builtin_fma(sc_[e_], w1_, ACC_[i_][1][g2_]);                                                  \
      if (iface_ && fr == 15) x1_ = sv_[e_];                                                                       \
      double* d_ = staging + sys_ * XL_SR + 16 + wc * 32 + fr;                                                     \
      d_[0] = x0_;                                                                                                 \
      d_[16] = x1_;                                                                                                \
    }                                                                                                              \
    }                                                                                                              \
    F_gpos = (p * N + P_r) * f.nc + P_q * N + 128 * P_h;                                                           \
    F_nvalid = min(128, (P_q < ncb - 1 ? N : n1) - 128 * P_h);                                                     \
    F_first = F_gpos == gs0;                                                                                       \
    F_next = 0;                                                                                                    \
    F_active = 0;                                                                                                  \
  } while (0)
    // one tile: CUR_ collects it, PRV_ (the tile before) is parked at its start and written out under its k loop
#define XL_TILE(CUR_, PRV_, HAS_PREV_)                                                                             \
  do {                                                                                                             \
    const bool hasp_ = (HAS_PREV_);                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)              \
        CUR_[i_][j_] = d4_t{0.0, 0.0, 0.0, 0.0};                                                                   \
    const int tot_ = (__builtin_amdgcn_readlane(segcnt_v, T_q) + 1) / 2;                                           \
    for (int ch_ = 0; ch_ < tot_; ++ch_) {                                                                         \
      const unsigned sb_ = unsigned(slot) * XL_SLOT;                                                               \
      XL_FRAGS(sb_, 1, af1, bf1);                                                                                  \
      XL_MFMA4(CUR_, af0, bf0);                                                                                    \
      if (ch_ == 0) {                                                                                              \
        if (hasp_) {                                                                                               \
          XL_PARK(PRV_);                                                                                           \
        }                                                                                                          \
      } else {                                                                                                     \
        XL_FLUSH_OPP();                                                                                            \
      }                                                                                                            \
      XL_FRAGS(sb_, 2, af0, bf0);                                                                                  \
      XL_MFMA4(CUR_, af1, bf1);                                                                                    \
      XL_FLUSH_OPP();                                                                                              \
      XL_FRAGS(sb_, 3, af1, bf1);                                                                                  \
      XL_MFMA4(CUR_, af0, bf0);                                                                                    \
      XL_FLUSH_OPP();                                                                                              \
      /* chunk cglob + 1 has landed for everybody; nobody reads the slot of chunk cglob any more */                \
      if (loader) {                                                                                                \
        if (L_issued == cglob + 3 && L_lastsz == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");               \
        else if (L_issued == cglob + 3 && L_lastsz == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
      }                                                                                                            \
      if (g.dbg & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                       \
      if (loader) XL_ISSUE_BATCH(sb_);                                                                             \
      if (ch_ == 0) F_active = hasp_ ? 1 : 0;                                                                      \
      if (ch_ == tot_ - 1) F_active = 0;                                                                           \
      const int nslot_ = slot == 2 ? 0 : slot + 1;                                                                 \
      if (cglob + 1 < ctotal) XL_FRAGS(unsigned(nslot_) * XL_SLOT, 0, af0, bf0);                                   \
      XL_MFMA4(CUR_, af1, bf1);                                                                                    \
      XL_FLUSH_OPP();                                                                                              \
      slot = nslot_;                                                                                               \
      ++cglob;                                                                                                     \
    }                                                                                                              \
    P_r = T_r; P_q = T_q; P_h = T_h; P_w3 = T_w3;                                                                  \
    if (++T_h == ntq) { T_h = 0; if (++T_q == ncb) { T_q = 0; ++T_r; } }                                           \
    T_w3 = T_w3 == 2 ? 0 : T_w3 + 1;                                                                               \
  } while (0)

    if (loader) {
      XL_LOADER_TILE();
      XL_PREFETCH_RECS();
      XL_ISSUE_BATCH(0u);
      XL_ISSUE_BATCH(unsigned(XL_SLOT));
      XL_ISSUE_BATCH(2u * XL_SLOT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    XL_FRAGS(0u, 0, af0, bf0);
    d4_t accA[2][2], accB[2][2];
    for (int t = 0; t < ((g.dbg & 64) ? min(nt, (g.dbg >> 8)) : nt); t += 2) {
      XL_TILE(accA, accB, t > 0);
      if (t + 1 < nt) XL_TILE(accB, accA, true);
    }
    // ---------------------------------------------------------------- drain: the last tile, then the carried remainder
    if (nt & 1) XL_PARK(accA);
    else XL_PARK(accB);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (storer) {
      const int end = F_gpos + F_nvalid;  // one past the last position of the stretch
      for (int u = 0; u < 16; ++u) {
        XL_FLUSH_LOAD(F_v0, F_t0, F_s0, F_c0, F_ok0, u);
        XL_FLUSH_STORE(F_v0, F_t0, F_s0, F_c0, F_ok0);
        const int s = 16 * sw + u, m = m0 + s;
        if (m >= Mc) continue;
        const long long rowd = (row0 + m) * f.dim;
        const int rem = int((ubase_d + (unsigned long long)rowd + (unsigned long long)end) & 15ull);
        const double* cr = staging + s * XL_SR + 16 - rem;
        double* dst = U + rowd + end - rem;
        if (2 * lane + 1 < rem) {
          const double2_u v = *reinterpret_cast<const double2_u*>(cr + 2 * lane);
          *reinterpret_cast<double2_a16*>(dst + 2 * lane) = double2_a16{v.x, v.y};
        } else if (2 * lane + 1 == rem) {
          dst[2 * lane] = cr[2 * lane];
        }
      }
    }
#undef XL_TILE
#undef XL_PARK
#undef XL_FLUSH_OPP
#undef XL_FLUSH_STORE
#undef XL_FLUSH_LOAD
#undef XL_PREFETCH_RECS
#undef XL_ISSUE_BATCH
#undef XL_ISSUE_HALF
#undef XL_LOADER_TILE
#undef XL_FRAGS
#undef XL_MFMA4
  }
  // The (small) expansion of the edge values -- the interface entries of the snapshot rows that are not inside a block row's
  // mesh rows, and the nodal blocks of the interface vector -- rides at the tail: its items are dealt to the workgroups
  // (at most one each on a full chip), four waves each, staging area = the operand slots.
  if (g.with_expand) {
    static_assert(STAGE_TOTAL * sizeof(double) <= XL_NSLOT * XL_SLOT, "expansion staging must fit");
    __syncthreads();
    if (threadIdx.x >= 256) return;  // (the expansion is written for four waves)
    const int nx = f.n1p / 64, ny = (Mc + 63) / 64, nitem = nx * ny * (f.nexp + 1);
    for (int item = int(blockIdx.x); item < nitem; item += int(gridDim.x))
      expand_tile(f, Mc, U, row0, reinterpret_cast<double*>(lds), item % nx, (item / nx) % ny, item / (nx * ny), 1);
  }
}
#undef XL_DMA
#undef XL_DMA1

// host side: can this launch take the whole-line kernel, and with which shape
int rom_launch_extend_lines(rom_fem* f, const FemDev& d, int Mc, double* U, long long row, hipStream_t st, bool with_expand) {
  rom_ctx* ctx = f->ctx;
  const size_t lds_limit = 160 * 1024;
  int R = rom_extend_lines_max_rows(f->ncb, lds_limit);
  if (f->sw_lines_rows > 0) R = std::min(R, f->sw_lines_rows);
  ROM_CHECK(R >= 1, "k_extend_lines: no room for the interface-vertex values in LDS");
  const int nsg = (Mc + XL_S - 1) / XL_S;
  if (f->sw_lines_rows <= 0) {
    // rows per stretch: as many as the LDS allows while every CU still gets a workgroup (a stretch ends in partial lines,
    // so long stretches are better; one round of equal workgroups has no tail)
    const int want = std::max(1, ctx->n_cu > 0 ? ctx->n_cu : 256);
    while (R > 4 && (long long)f->nrb * nsg * ((f->n1 + R - 1) / R) < want) --R;
  }
  const int nchunk = (f->n1 + R - 1) / R;
  XLArgs g;
  g.R = R; g.nchunk = nchunk; g.nsg = nsg; g.order = f->sw_lines_order >= 0 ? f->sw_lines_order : 0;
  g.with_expand = with_expand ? 1 : 0;
  g.dbg = f->sw_lines_dbg;  // timing probes: 1 no global stores, 2 no write-out, 4 no parking, 8 no operand loads
  const size_t bytes = rom_extend_lines_lds_bytes(f->ncb, R);
  if (!f->lds_optin_lines) {
    ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_extend_lines), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_limit)));
    f->lds_optin_lines = true;
  }
  k_extend_lines<<<unsigned(f->nrb * nchunk * nsg), 512, bytes, st>>>(d, g, Mc, U, row);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}
