// Internal declarations shared by the translation units of libromhc.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "romhc.h"

// ---- error plumbing ------------------------------------------------------------------------
void rom_set_error(const char* fmt, ...);

#define ROM_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      rom_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ROM_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

#define ROM_CHECK(cond, ...)      \
  do {                            \
    if (!(cond)) {                \
      rom_set_error(__VA_ARGS__); \
      return ROM_ERR_INVALID;     \
    }                             \
  } while (0)

#define ROM_TRY(call)          \
  do {                         \
    int _s = (call);           \
    if (_s != ROM_OK) return _s; \
  } while (0)

// ---- tile geometry of the interface Cholesky ---------------------------------------------------
constexpr int TB = 64;   // tile edge (doubles)
constexpr int BK = 16;   // K chunk staged through LDS
constexpr int LDK = 18;  // LDS row stride of a staged [64][BK] chunk (conflict-free ds_read_b64)
constexpr int LDC = 66;  // LDS row stride of a full [64][64] tile

struct ProfRec {
  hipEvent_t e0, e1;
  int name_id;
  hipStream_t st;
};

struct rom_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // sub-batch streams of rom_solve_batch
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int n_streams = 1;  // ROMHC_STREAMS=2..4 splits a sweep into concurrent sub-batches (no gain measured at M=1024)
  hipStream_t prof_stream = nullptr;  // stream the next ROM_PROF bracket records on (null: `stream`)
  size_t ws_limit = size_t(24) << 30;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // profiling
  bool profile = false;
  std::vector<std::string> prof_names;
  std::vector<double> prof_flops, prof_bytes;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> event_pool;
  // caching allocator of rom_buf blocks (rounded size -> free device pointers)
  std::map<size_t, std::vector<double*>> free_blocks;
  size_t cached_bytes = 0, cache_limit = size_t(64) << 30;
  // device status word (not-SPD flag) + scratch for reductions
  int* d_status = nullptr;
  double* d_scratch = nullptr;
  size_t scratch_doubles = 0;
  // RCCL
  void* comm = nullptr;
  int rank = 0, nranks = 1;
  hipStream_t comm_stream = nullptr;  // collectives overlapped with compute (rom_comm_allgather_async)
  hipEvent_t ev_comm = nullptr, ev_slot[2] = {nullptr, nullptr};
  bool slot_used[2] = {false, false};
};

struct rom_buf {
  rom_ctx* ctx;
  double* p;
  size_t n;
};

int rom_ctx_scratch(rom_ctx* ctx, size_t n_doubles, double** out);

// profiling bracket around one kernel launch
struct ProfScope {
  rom_ctx* ctx;
  int idx = -1;
  ProfScope(rom_ctx* c, const char* name, double flops, double bytes);
  ~ProfScope();
};
#define ROM_PROF(ctx, name, flops, bytes) ProfScope _prof_scope_##__LINE__(ctx, name, flops, bytes)

// ---- assembly descriptors (host-built, device-read) --------------------------------------------
struct TileTerm {
  int blk;   // block index p*ncb+q whose coefficient scales the term
  int tmat;  // sr*4+sc : which Dirichlet-to-Neumann table
  int r0;    // row offset inside the table (= local node offset of the tile rows on its edge)
  int c0;    // col offset inside the table
};

// Contribution of one pre-eliminated edge e (closed-form elimination, D_e = (a_e0 + a_e1) K) to a tile:
//   - c_row c_col / (a_e0 + a_e1) * R[table][r0 + r][c0 + c],   c = a[blk] on edge-node rows/cols,
//   (a_e0 + a_e1)/2 on cross-point slots (R = X_f K^-1 X_f'^T, parameter independent).
struct PreTerm {
  int table;       // index of the R table (n1p x n1p doubles each)
  int r0, c0;      // offset of the tile inside the table
  int brow, bcol;  // blocks scaling edge-node rows / cols (-1: the edge shares no block with e)
  int e0, e1;      // the two blocks of the eliminated edge
};

struct TileDesc {
  int ti, tj;        // tile coordinates in the (permuted) interface ordering, ti >= tj
  int nterms;        // 0..2 Schur terms  - a[blk] * T[tmat][r0+r][c0+c]
  TileTerm term[2];
  int npre;          // 0..4 pre-eliminated-edge terms
  PreTerm pre[4];
  int ndc;           // cols [0, ndc) are unknowns (as ndr for rows)
  int same_edge;     // 1: rows and cols lie on the same edge -> tridiagonal A_GammaGamma part
  int hv;            // 0 horizontal edge (blocks up/down), 1 vertical (left/right)
  int b0, b1;        // the two blocks of that edge: (up, dn) or (lf, rt)
  int lr0, lc0;      // local node index (on the edge) of tile row 0 / col 0
  int nvr, nvc;      // number of edge nodes among the tile rows / cols
  int ndr;           // rows [0, ndr) are unknowns (edge nodes + cross slots); beyond: identity padding
  int x0, x1;        // range in the extras list
  int diag;          // 1 if ti == tj
};

struct TileExtra {  // entries touching cross points
  int r, c;         // position inside the tile
  int kind;         // 0: -(a[b0]+a[b1])/2 ; 1: ((a[b0]+a[b1])+a[b2])+a[b3]
  int b[4];
};

struct BlockSide {  // per block, per side: where its interface values live (or -1)
  int off[4];
};

// rhs correction of one tile of active unknowns by one pre-eliminated edge:
//   y[tile*64 + r] += c_row / (a_e0 + a_e1) * q[qoff + lr0 + r]
struct RhsTerm {
  int tile, lr0, nvr, ndr;  // tile row, its local offset on the edge, edge-node rows, defined rows
  int qoff;                 // offset of the q vector (n1p doubles) in the vector table
  int brow, e0, e1;
};

// back substitution of one pre-eliminated edge:  x_e = (w_e + sum_f B_fe (c_f . x_f)) / (a_e0 + a_e1)
struct PreNb {
  int fpos;    // position of the neighbour's n1p block in the interface vector
  int blk;     // block scaling its edge nodes (-1: none)
  int nused;   // cross-point slots behind the edge nodes (scaled by (a_e0+a_e1)/2)
  int bt;      // index of the B^T table (n1p x n1p): row = node of e, col = position on the neighbour
};
struct PreEdge {
  int pos;     // position of e's n1p block in the interface vector
  int e0, e1;
  int woff;    // offset of w_e = K^-1 g_e in the vector table
  int nnb;
  PreNb nb[8];
};

struct rom_fem {
  rom_ctx* ctx;
  int nrb, ncb, N, n1, n1p, tpe;  // n1 = N-1, n1p = padded to TB multiple, tpe tiles per edge
  int nr, nc;
  int64_t dim;
  int nG;      // real interface unknowns
  int nGp;     // stride of the interface vectors = T*TB + (pre-eliminated edges)*n1p
  int nGa;     // active (factorised) part = T*TB
  int npre;    // edges eliminated in closed form
  int T;       // tiles per dimension
  int nslots;  // nonzero lower tiles
  // device tables
  double* d_A0 = nullptr;    // (n1*n1) x n1p : Q[j,mode] rho_mode(i) (harmonic extension from side i=0, sine basis)
  double* d_Qp = nullptr;    // n1p x n1p sine matrix
  int* d_kmax = nullptr;     // [N+1]
  int* d_epos = nullptr;     // [n_edges]
  int n_edges = 0;
  double ext_flops = 0;
  double* d_yhat = nullptr;  // [ws_M][nGp]
  double* d_Tm = nullptr;    // 16 x n1p x n1p
  double* d_W = nullptr;     // n1*n1 : L^{-1} 1
  double* d_g = nullptr;     // nGp : parameter independent interface rhs
  TileDesc* d_desc = nullptr;
  TileExtra* d_extra = nullptr;
  int* d_slot_of = nullptr;  // T*T -> slot or -1 (lower)
  int* d_kptr = nullptr;     // nslots+1
  int* d_kpair = nullptr;    // 2*entries (slotA, slotB)
  int* d_colptr = nullptr;   // T+1 : rows below the diagonal in column j
  int* d_colrow = nullptr;   // slots of those tiles (and their tile row in d_colti)
  int* d_colti = nullptr;
  double* d_R = nullptr;       // R / B^T tables of the pre-eliminated edges (n1p*n1p each)
  double* d_vec = nullptr;     // q / w vectors of the pre-eliminated edges (n1p each)
  RhsTerm* d_rhs = nullptr;
  PreEdge* d_pre = nullptr;
  int nrhs = 0;
  BlockSide* d_sides = nullptr;  // nrb*ncb
  int* d_vmap = nullptr;         // nG real interface unknowns: padded position -> global dof (or -1), size nGp
  // host copies
  std::vector<TileDesc> desc;
  std::vector<int> slot_of, kptr, kpair, colptr, colrow, colti, diag_slot;
  std::vector<BlockSide> sides;
  std::vector<double> g_host;
  // work accounting
  double flops_solve = 0, bytes_solve = 0;
  // factor workspace (grown on demand)
  double* d_L = nullptr;
  double* d_invL = nullptr;
  double* d_y = nullptr;
  int ws_M = 0;
};

// kernels / launchers implemented in the .hip files
int rom_launch_gemm_nt(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                       int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                       const char* prof_name);
int rom_launch_gemm_nt_ex(rom_ctx* ctx, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                          int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                          const char* prof_name, int lower_only);
