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
  int n_cu = 0;  // compute units of the device (grid of the persistent kernels)
  hipStream_t stream = nullptr;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // sub-batch streams of rom_solve_batch
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  // A sweep can be split into n_streams concurrent sub-batches on separate streams (ROMHC_STREAMS=1..4).  0 = the default:
  // one stream (round 5: no gain left from two at C4 / C5, -1 % at C2).
  int n_streams = 0;
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
  hipEvent_t ev_comm = nullptr, ev_slot[2] = {nullptr, nullptr};  // ev_comm: fork / join-all; ev_slot: end of a slot's collective
  bool slot_used[2] = {false, false};
  // the device ranges the last collective of each slot reads and writes, and whether the compute stream has been
  // ordered behind that collective since (rom_comm_wait_slot / rom_comm_wait): rom_buf_free consults them
  const double* slot_lo[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // send, receive, packed scratch
  const double* slot_hi[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  bool slot_joined[2] = {true, true};
  // kernels that need more than 64 KB of dynamic LDS must opt in once per DEVICE (a context is one device)
  bool lds_optin_reduced_solve = false, lds_optin_small_eig = false, lds_optin_pivchol = false;
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
// The factorised ("reduced") interface vector is a packed sequence of groups: the compressed unknowns
// z_f = W_f^T u_f of every active edge f, each followed by the cross points it hosts.  A tile of the
// reduced matrix is a linear combination of parameter-independent 64x64 tables:
//   tile = sum_t coef_t(a) * pool[tab_t],   table t zero outside its rectangle [r_lo,r_hi) x [c_lo,c_hi)
struct GenTerm {
  int tab;                         // index of the 64x64 table in the pool
  short r_lo, r_hi, c_lo, c_hi;    // bounding rectangle inside the tile
  int kind;                        // coefficient formula, see term_coef() in rom_fem_kernels.hip
  int b[4];                        // block indices it uses
};

struct TileDesc {
  int ti, tj;   // tile coordinates in the reduced ordering, ti >= tj
  int t0, t1;   // range in the term list
  int ndr;      // rows [0, ndr) are unknowns; beyond: identity padding (last tile only)
  int diag;     // 1 if ti == tj
};

// How one side of a block enters the harmonic extension (k_extend)
struct ExtSide {
  int mode;    // 0: domain boundary; 1: sine modes of the edge values; 2: compressed edge [z_f, 1/s_f]
  int off;     // mode 1: nodal block of the edge (in yhat); mode 2: its [z_f, 1/s_f] block (in y)
  int nch;     // mode 2: K chunks = (rank + 1) rounded up to BK, / BK
  int r;       // mode 2: rank
  int gtab;    // mode 2: offset (doubles) of the (n1*n1) x (nch*BK) table H_0 [P_f, p0_f]
  int gseg;    // mode 2: offset (doubles) of the segment-major copy of that table in Gs (k_extend128)
  int b0, b1;  // mode 2: blocks of the edge (s_f = a_b0 + a_b1)
};
// coefficient block of one closed-form edge on the single-tile path (k_solve1): see the dense product there
struct DenseGroup {
  int cpos, r, b0, b1;
  int nv;                                // parts scaled by a[vblk] / (a[vu0] + a[vu1]): the neighbours' p0
  int voff[4], vblk[4], vu0[4], vu1[4];
};

struct BlockSide {
  ExtSide s[4];
};

// rhs of the reduced system:  y[pos + i] += coef * vec[voff + i], i < len
//   kind 0: coef = a[b0] / (a[e0] + a[e1])   (edge group rows, pre-eliminated edge (e0|e1))
//   kind 1: coef = 1/2                        (cross point row)
struct RhsTerm {
  int pos, len, voff, kind, b0, e0, e1;
};

// back substitution of one pre-eliminated edge:  x_e = (w_e + sum_u B_ue^T (c_u . x_u)) / (a_e0 + a_e1)
struct PreNb {
  int fpos;    // position of the neighbour's nodal block (or of the cross block) in the interface vector
  int blk;     // block scaling an edge neighbour (c_u = a[blk]); -1: the cross block (c_u = (a_e0+a_e1)/2)
  int nch;     // K chunks (of BK) to run over
  int bt;      // index of the B^T table (n1p x n1p): row = node of e, col = position in the neighbour block
};
struct PreEdge {
  int pos;     // position of e's nodal block in the interface vector
  int e0, e1;
  int woff;    // offset of w_e = K^-1 g_e in the vector table
  int nnb;
  PreNb nb[8];
};

// nodal values of an active edge from its compressed unknowns:  u_f = P z_f + p0 / (a_b0 + a_b1)
struct ExpEdge {
  int zpos, nch;  // position of z_f in the reduced vector, K chunks (rank rounded up to BK)
  int npos;       // position of the nodal block
  int ptab;       // index of the P table (n1p x n1p, row = node, col = compressed index)
  int p0off;      // offset of p0 in the vector table
  int spos;       // position of 1/(a_b0 + a_b1) in the scalar block of the interface vector
};

// coefficient block [z, 1/s, 0...] of an edge that enters the extension in compressed form (k_coef)
struct CoefTerm {
  int src, len;  // source values in the interface vector (reduced unknowns of a neighbour, or one cross point)
  int blk;       // >= 0: neighbouring edge, weight a[blk]; -1: cross point, weight s_e / 2
  int moff;      // (len x r) matrix in the coefficient-matrix table
  int voff;      // r-vector scaled by 1/(a[u0] + a[u1]) (the neighbour's p0 part), or -1
  int u0, u1;
};
struct CoefGroup {
  int kind;      // 0: active edge, z copied from the reduced vector at zpos; 1: closed-form edge
  int cpos, r, w;  // position and padded width of the block, rank
  int zpos;
  int b0, b1;    // blocks of the edge (s = a_b0 + a_b1)
  int nterm;
  CoefTerm t[8];
};

struct rom_fem {
  rom_ctx* ctx;
  int nrb, ncb, N, n1, n1p;  // n1 = N-1, n1p = n1 padded to a TB multiple
  int nr, nc;
  int64_t dim;
  int nG;      // real interface unknowns
  int nGp;     // stride of the interface vectors = nGa + (edges)*n1p + cross block
  int nGa;     // reduced (factorised) part = T*TB
  int nred;    // reduced unknowns (<= nGa)
  int npre;    // edges eliminated in closed form
  int nexp;    // active edges
  int ncross, xb0;  // cross points and the position of their block in the interface vectors
  int T;       // tiles per dimension
  int nslots;  // nonzero lower tiles
  // device tables
  double* d_G = nullptr;     // extension tables of the compressed edges
  double* d_Gs = nullptr;    // the same tables, segment-major and with the vertices of a mesh row adjacent (k_extend128)
  double* d_A0 = nullptr;    // (n1*n1) x n1p : Q[j,mode] rho_mode(i) (harmonic extension from side i=0, sine basis)
  double* d_Qp = nullptr;    // n1p x n1p sine matrix
  int* d_kmax = nullptr;     // [N+1]
  int* d_epos = nullptr;     // [n_edges]
  int n_edges = 0;
  double ext_flops = 0;
  double* d_yhat = nullptr;  // [ws_M][nGp]
  double* d_dots = nullptr;  // [ws_M][ncf * 8]: k_coef's dot products, only where they do not fit the LDS (large geometries)
  double* d_W = nullptr;     // n1*n1 : L^{-1} 1
  double* d_g = nullptr;     // nGa : parameter independent part of the reduced rhs
  TileDesc* d_desc = nullptr;
  GenTerm* d_terms = nullptr;
  double* d_pool = nullptr;  // 64x64 tables of the tile terms
  int* d_pairs = nullptr;    // (term, block) pairs of the single-tile assembly
  int* d_alist = nullptr;    // tile assembly as a stream: pieces per tile slot and wave (rom_fem_dev.h)
  int* d_aoff = nullptr;
  int npairs = 0;
  double* d_pool_acc = nullptr;  // k_solve1: the pairs' table pieces in accumulator layout, in the order its four waves walk them
  int* d_wmeta = nullptr;        // ... and their metas; wave w walks wp0[w] .. wp0[w + 1] - 1 (rom_fem_dev.h)
  int wp0[5] = {};
  int* d_s1_items = nullptr;   // k_solve1: flat records of the dense items (24 ints) and of the coefficient items (4 ints), rom_fem_setup.hip
  int* d_s1_citems = nullptr;
  int* d_kptr = nullptr;     // nslots+1
  int* d_kpair = nullptr;    // 2*entries (slotA, slotB)
  int* d_colptr = nullptr;   // T+1 : rows below the diagonal in column j
  int* d_colrow = nullptr;   // slots of those tiles (and their tile row in d_colti)
  int* d_colti = nullptr;
  double* d_Bt = nullptr;      // B^T tables of the pre-eliminated edges (n1p*n1p each)
  double* d_P = nullptr;       // expansion tables of the active edges (n1p*n1p each)
  double* d_vec = nullptr;     // vectors: w_e, p0_f, rhs corrections
  RhsTerm* d_rhs = nullptr;
  PreEdge* d_pre = nullptr;
  ExpEdge* d_exp = nullptr;
  int* d_xred = nullptr;       // reduced position of every cross point
  int* d_scb = nullptr;        // scalar block descriptors
  int spos0 = 0, nsc = 0, n_all_edges = 0;
  CoefGroup* d_groups = nullptr;
  double* d_cm = nullptr;
  int* d_item_group = nullptr;
  int* d_item_k = nullptr;
  int ncoef = 0;               // entries of all coefficient blocks
  int* d_item_cf = nullptr;    // k_coef: closed-form index of an entry, tasks {group, k, term, slot} of its dot products (rom_fem_dev.h)
  int* d_ctask = nullptr;
  int nctask = 0, ncf = 0;
  bool lds_optin_coef = false;  // k_coef was given more than the default dynamic LDS on this device
  bool fused1 = false;         // the reduced matrix is one tile: whole solve in k_solve1
  // A/B switches of the kernel sequencing, read from the environment ONCE per FE space (rom_fem_create): ROMHC_NO_FUSED,
  // ROMHC_EXT_FLAT (-1: automatic), ROMHC_NO_EXT128, ROMHC_NO_FOLD_EXPAND
  bool sw_no_fused = false, sw_no_ext128 = false, sw_no_fold = false;
  int sw_x128_sys_fast = -1;   // -1: by table size
  size_t gs_bytes = 0;         // bytes of the segment-major extension tables (k_extend128's B operand)
   // k_extend128: system group fastest in the workgroup order (tables out of cache, not HBM)
  int sw_ext_flat = -1;
  bool sw_coef_global = false;    // (A/B build) k_coef's dot products in HBM although they fit the LDS
  bool sw_no_tile_pairs = false;  // ROMHC_NO_TILE_PAIRS: tile Cholesky with one system per workgroup
  bool sw_no_tile_stream = false; // ROMHC_NO_TILE_STREAM: tile assembly in registers (s_tile_load) instead of the stream into LDS
  DenseGroup* d_dgroups = nullptr;
  int* d_dweight = nullptr;    // per (dense group, source position): block of the weight, -1 cross point, -2 none
  int* d_ditem_group = nullptr;
  int* d_ditem_k = nullptr;
  double* d_dmat = nullptr;    // TB x ndi
  int ndg = 0, ndi = 0;
  int nrhs = 0;
  BlockSide* d_sides = nullptr;  // nrb*ncb
  int* d_lr_blocks = nullptr;    // blocks whose sides are all in compressed form (k_extend_lr)
  std::vector<int> lr_blocks_host;  // the same list on the host (k_extend128 gets its descriptors as kernel arguments)
  int* d_gen_blocks = nullptr;   // the others (k_extend)
  int n_lr_blocks = 0, n_gen_blocks = 0, lr_nch = 0;
  int* d_vmap = nullptr;         // interface position -> global dof (or -1), size nGp
  int* d_scat = nullptr;         // positions k_scatter_interface copies to the snapshot rows
  int nscat = 0;
  // host copies
  std::vector<TileDesc> desc;
  std::vector<int> slot_of, kptr, kpair, colptr, colrow, colti, diag_slot;
  std::vector<BlockSide> sides;
  std::vector<int> ranks;        // compressed size of every active edge (elimination order)
  // work accounting
  double flops_solve = 0, bytes_solve = 0;
  void* fmap = nullptr;          // rom_factored_map: geometry of the snapshots in interface-vector coordinates (rom_factored.hip)
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

void rom_factored_map_free(void* map);  // (rom_factored.hip)
// pack of interface vectors into their compact form on stream `st` (rom_fem_solve.hip)
int rom_launch_pack_reduced(rom_fem* f, const double* Y, double* Yc, int M, hipStream_t st);
