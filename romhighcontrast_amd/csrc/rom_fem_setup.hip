// rom_fem_create: geometry of the interface, closed-form edge elimination, low-rank compression of the edges,
// symbolic tile Cholesky of the reduced matrix and every parameter-independent table, built on the host in
// long double (algorithm: see the header of rom_fem_kernels.hip and DESIGN.md section 3).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <set>
#include <thread>

#include <chrono>
#include <tuple>

#include "rom_fem_dev.h"
#include "rom_hostla.h"

// forcing switches of the A/B build (see the end of rom_fem_create); the product build reads none of them
#ifdef ROMHC_AB
static const char* ab_env(const char* name) { return getenv(name); }
#else
static const char* ab_env(const char*) { return nullptr; }
#endif

FemDev make_dev(const rom_fem* f) {
  FemDev d;
  d.nrb = f->nrb; d.ncb = f->ncb; d.N = f->N; d.n1 = f->n1; d.n1p = f->n1p; d.nr = f->nr; d.nc = f->nc;
  d.nGp = f->nGp; d.nGa = f->nGa; d.npre = f->npre; d.nrhs = f->nrhs; d.nexp = f->nexp; d.ncross = f->ncross;
  d.xb0 = f->xb0; d.pool = f->d_pool; d.alist = f->d_alist; d.aoff = f->d_aoff; d.tile_stream = f->sw_no_tile_stream ? 0 : 1; d.pairs = f->d_pairs; d.npairs = f->npairs; d.pool_acc = f->d_pool_acc; d.wmeta = f->d_wmeta; d.s1_items = f->d_s1_items; d.s1_citems = f->d_s1_citems; for (int i = 0; i < 5; ++i) d.wp0[i] = f->wp0[i]; d.s1_t0 = d.s1_nterm = d.s1_ndr = 0; if (f->fused1 && !f->desc.empty()) { d.s1_t0 = f->desc[0].t0; d.s1_nterm = f->desc[0].t1 - f->desc[0].t0; d.s1_ndr = f->desc[0].ndr; } d.terms = f->d_terms; d.Bt = f->d_Bt; d.P = f->d_P; d.vec = f->d_vec;
  d.rhs = f->d_rhs; d.pre = f->d_pre; d.exp = f->d_exp; d.xred = f->d_xred; d.scb = f->d_scb; d.spos0 = f->spos0; d.nsc = f->nsc;
  d.sblk0 = f->spos0 + f->n_all_edges; d.groups = f->d_groups; d.cm = f->d_cm; d.item_group = f->d_item_group;
  d.item_k = f->d_item_k; d.item_cf = f->d_item_cf; d.ctask = f->d_ctask; d.nctask = f->nctask; d.ncf = f->ncf; d.ncoef = f->ncoef; d.dgroups = f->d_dgroups; d.dweight = f->d_dweight;
  d.ditem_group = f->d_ditem_group; d.ditem_k = f->d_ditem_k; d.dmat = f->d_dmat; d.ndg = f->ndg; d.ndi = f->ndi; d.T = f->T; d.nslots = f->nslots;
  d.kblk = f->nrb * f->ncb; d.dim = f->dim;
  d.G = f->d_G; d.Gs = f->d_Gs; d.A0 = f->d_A0; d.Qp = f->d_Qp; d.kmax = f->d_kmax; d.epos = f->d_epos; d.yhat = f->d_yhat; d.W = f->d_W;
  d.g = f->d_g; d.desc = f->d_desc;
  d.kptr = f->d_kptr; d.kpair = f->d_kpair; d.colptr = f->d_colptr; d.colrow = f->d_colrow;
  d.colti = f->d_colti; d.sides = f->d_sides; d.lr_blocks = f->d_lr_blocks; d.n_lr = f->n_lr_blocks; d.gen_blocks = f->d_gen_blocks; d.vmap = f->d_vmap; d.scat = f->d_scat; d.nscat = f->nscat; d.L = f->d_L; d.invL = f->d_invL;
  d.y = f->d_y; d.status = f->ctx->d_status; d.gdots = f->d_dots;
  return d;
}

// ============================================================================================
// host: geometry, compression, symbolic tile Cholesky, tables
// ============================================================================================
namespace {

using hostla::ld;
using hostla::Mat;

struct Edge {
  int hv, p, q;  // hv 0: horizontal (r = pN, c in block column q); 1: vertical (c = qN, r in block row p)
  int b0, b1;    // up/dn or lf/rt block indices
};

// one parameter-independent block of the reduced matrix: coef(kind, b) * tab at (rpos, cpos)
struct Small {
  int rpos, cpos;
  Mat tab;
  int kind;
  std::array<int, 4> b;
};

// compressed representation of an active edge (shared by all edges with the same surroundings)
struct Comp {
  int r = 0;
  Mat W;               // n1 x r   orthonormal basis of the coupling range
  Mat Kt;              // r x r    (W^T K^-1 W)^-1
  Mat P;               // n1 x r   K^-1 W Kt
  std::vector<ld> gt;  // r        Kt W^T K^-1 g_f
  std::vector<ld> p0;  // n1       (K^-1 - P W^T K^-1) g_f
  Mat KiW;             // n1 x r   K^-1 W          (closed-form edges: u_e = (KiW c_e + wK) / s_e)
  std::vector<ld> wK;  // n1       K^-1 g_f
};

// run fn(0) ... fn(n-1) on up to hardware_concurrency host threads (the long-double table products are
// independent of each other and dominate rom_fem_create)
template <class F>
void parallel_for(size_t n, F fn) {
  const unsigned nthr = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), unsigned(n)));
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t i = next++; i < n; i = next++) fn(i);
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}

// Closed-form tables of one unit block (N x N cells, Dirichlet 5-point Laplacian L) in long double, from the sine
// eigenbasis:  Q[j][m] = sqrt(2/N) sin(pi j m / N),  lam_m = 2 - 2 cos(pi m / N),
// rho_m(i) = sinh((N-i) phi_m) / sinh(N phi_m) with cosh phi_m = 1 + lam_m / 2.
struct UnitBlock {
  int N, n1;
  Mat Q, rho;                  // rho(m, i), i = 0..N
  std::vector<ld> lam, kappa;  // kappa_m = 1 + lam_m/2 - rho_m(1): K = tridiag(-1/2, 2, -1/2) - T_same = Q diag(kappa) Q
  Mat Wl;                      // L^-1 1
  Mat Kmat, Kinv;
  std::vector<ld> gE[2];       // interface rhs of a horizontal / vertical edge: h^2 (1 + W on the two adjacent lines)
  std::vector<double> rho_d, Wd;

  UnitBlock(int N_, bool with_edges) : N(N_), n1(N_ - 1), Q(n1, n1), rho(n1, N_ + 1), lam(n1), kappa(n1), Wl(n1, n1) {
    const ld PI = acosl(-1.0L);
    for (int j = 1; j <= n1; ++j) {
      lam[j - 1] = 2.0L - 2.0L * cosl(PI * j / N);
      for (int m = 1; m <= n1; ++m) Q(j - 1, m - 1) = sqrtl(2.0L / N) * sinl(PI * j * m / (ld)N);
    }
    for (int m = 0; m < n1; ++m) {
      const ld phi = acoshl(1.0L + lam[m] / 2.0L);
      const ld den = -expm1l(-2.0L * N * phi);  // 1 - exp(-2 N phi)
      for (int i = 0; i <= N; ++i) rho(m, i) = expl(-phi * i) * (-expm1l(-2.0L * (N - i) * phi)) / den;
      kappa[m] = 1.0L + lam[m] / 2.0L - rho(m, 1);
    }
    rho_d.resize(rho.v.size());
    for (size_t i = 0; i < rho.v.size(); ++i) rho_d[i] = double(rho.v[i]);
    {  // W = L^{-1} 1 = Q (s s^T / (lam_l + lam_m)) Q
      std::vector<ld> sv(n1, 0.0L);
      Mat Z(n1, n1);
      for (int m = 0; m < n1; ++m)
        for (int j = 0; j < n1; ++j) sv[m] += Q(j, m);
      for (int l = 0; l < n1; ++l)
        for (int m = 0; m < n1; ++m) Z(l, m) = sv[l] * sv[m] / (lam[l] + lam[m]);
      Zs = Z;
    }
    Kmat = Mat(n1, n1);
    Kinv = Mat(n1, n1);
    Mat QK(n1, n1), QKi(n1, n1);
    for (int j = 0; j < n1; ++j)
      for (int m = 0; m < n1; ++m) {
        QK(j, m) = Q(j, m) * kappa[m];
        QKi(j, m) = Q(j, m) / kappa[m];
      }
    // (four n1^3 products, one after the other with their rows spread over the host threads: two of them depend on
    // each other, so one thread per product would leave the critical path at two products)
    Wl = hostla::mul_par(Q, hostla::mul_nt_par(Zs, Q));
    if (with_edges) {
      Kmat = hostla::mul_nt_par(QK, Q);
      Kinv = hostla::mul_nt_par(QKi, Q);
    }
    Wd.resize(size_t(n1) * n1);
    for (size_t i = 0; i < Wd.size(); ++i) Wd[i] = double(Wl.v[i]);
    const ld h2 = 1.0L / ((ld)N * (ld)N);
    for (int hv = 0; hv < 2; ++hv) {
      gE[hv].resize(n1);
      for (int t = 0; t < n1; ++t)
        gE[hv][t] = h2 * (1.0L + (hv == 0 ? Wl(N - 2, t) + Wl(0, t) : Wl(t, N - 2) + Wl(t, 0)));
    }
  }

  // build the tables `ids` (and, if with_tk, their products with K^-1) on several threads
  void prepare(const std::vector<int>& ids, bool with_tk) {
    std::vector<int> todo;
    for (int id : ids)
      if (!(with_tk ? haveTK[id] : haveT[id]) && std::find(todo.begin(), todo.end(), id) == todo.end()) todo.push_back(id);
    parallel_for(todo.size(), [&](size_t i) { build(todo[i], with_tk); });
  }

  // Dirichlet-to-Neumann table T[sr*4+sc][t][k] = H_sc[interior vertex next to node t of side sr][k] (first use builds it)
  const Mat& Tm(int id) {
    if (!haveT[id]) build(id, false);
    return Tm_[id];
  }
  const Mat& Tm_ready(int id) const { return Tm_[id]; }  // (read-only access for the worker threads: built before)
  // ... and its product with K^-1
  const Mat& TK(int id) {
    if (!haveTK[id]) build(id, true);
    return TK_[id];
  }

 private:
  void build(int id, bool with_tk) {  // (distinct ids may be built concurrently)
    if (!haveT[id]) {
      const int sr = id >> 2, sc = id & 3;
      Mat V(n1, n1);
      for (int t = 0; t < n1; ++t) {
        int i, j;
        switch (sr) {
          case 0: i = 1; j = t + 1; break;
          case 1: i = N - 1; j = t + 1; break;
          case 2: i = t + 1; j = 1; break;
          default: i = t + 1; j = N - 1; break;
        }
        const int hr = h0_row(sc, i, j, N, n1);
        const int ii = hr / n1 + 1, jj = hr % n1 + 1;
        for (int m = 0; m < n1; ++m) V(t, m) = Q(jj - 1, m) * rho(m, ii);
      }
      Tm_[id] = hostla::mul_nt(V, Q);
      haveT[id] = 1;
    }
    if (with_tk && !haveTK[id]) {
      TK_[id] = hostla::mul(Tm_[id], Kinv);
      haveTK[id] = 1;
    }
  }
  Mat Zs;
  std::array<Mat, 16> Tm_, TK_;
  std::array<char, 16> haveT{}, haveTK{};
};

// Compressed form of an edge whose couplings act through the tables `tabs` (ids sr*4+sc) and, if x0 / x1, through
// its first / last node (cross points): W = orthonormal basis of the union of their ranges.
// Step 1: the nested basis down to `keep` of the first pivot, with the pivot norms (the caller picks the rank).
void compress_basis(const UnitBlock& ub, const std::vector<int>& tabs, bool x0, bool x1, ld keep, Mat& Wb, std::vector<ld>& pivots) {
  const int n1 = ub.n1;
  const int ntab = int(tabs.size());
  Mat C(n1, ntab * n1 + 2);
  for (int t = 0; t < ntab; ++t) {
    const Mat& Tt = ub.Tm_ready(tabs[t]);
    ld mx = 0;
    for (ld v : Tt.v) mx = std::max(mx, fabsl(v));
    if (mx == 0.0L) mx = 1.0L;
    for (int i = 0; i < n1; ++i)
      for (int k = 0; k < n1; ++k) C(i, t * n1 + k) = Tt(i, k) / mx;
  }
  if (x0) C(0, ntab * n1) = 1.0L;
  if (x1) C(n1 - 1, ntab * n1 + 1) = 1.0L;
  Wb = hostla::range_basis(C, keep, &pivots);
}
// Step 2: everything that follows from the first `r` columns of that basis (r >= n1 or !compress: nodal unknowns,
// W = I).  False if the compressed self block is not SPD.
bool compress_finish(const UnitBlock& ub, const Mat& Wfull, int r, int hv, bool compress, Comp& cp) {
  const int n1 = ub.n1;
  if (!compress || r >= n1) {  // nothing to gain: nodal unknowns
    cp.r = n1;
    cp.W = hostla::identity(n1);
    cp.Kt = ub.Kmat;
    cp.P = hostla::identity(n1);
    cp.gt = ub.gE[hv];
    cp.p0.assign(n1, 0.0L);
    cp.KiW = ub.Kinv;
    cp.wK = hostla::matvec(ub.Kinv, ub.gE[hv]);
    return true;
  }
  Mat Wb(n1, r);
  for (int i = 0; i < n1; ++i)
    for (int k = 0; k < r; ++k) Wb(i, k) = Wfull(i, k);
  cp.r = r;
  cp.W = Wb;
  Mat KiW = hostla::mul(ub.Kinv, Wb);
  Mat G = hostla::mul_tn(Wb, KiW);
  for (int i = 0; i < G.r; ++i)
    for (int j = 0; j < i; ++j) G(i, j) = G(j, i) = (G(i, j) + G(j, i)) / 2;
  if (!hostla::spd_inverse(G, cp.Kt)) return false;
  cp.P = hostla::mul(KiW, cp.Kt);
  std::vector<ld> v = hostla::matvec(ub.Kinv, ub.gE[hv]);
  std::vector<ld> wv = hostla::matvec(hostla::transpose(Wb), v);
  cp.gt = hostla::matvec(cp.Kt, wv);
  std::vector<ld> pw = hostla::matvec(cp.P, wv);
  cp.p0.resize(n1);
  for (int i = 0; i < n1; ++i) cp.p0[i] = v[i] - pw[i];
  cp.KiW = KiW;
  cp.wK = v;
  return true;
}

struct TermAcc {
  std::array<int, 5> key;
  std::vector<double> tab;
  int r_lo, r_hi, c_lo, c_hi;
};

// Table uploads go through the context's (non-blocking) compute stream, the stream every reader runs on, and wait for
// it: the host vector may be destroyed on return, and no ordering is left to a null-stream copy being host-synchronous.
template <class Tp>
int upload(hipStream_t st, Tp** dptr, const std::vector<Tp>& h) {
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(Tp);
  ROM_HIP(hipMalloc(dptr, bytes));
  if (!h.empty()) {
    ROM_HIP(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice, st));
    ROM_HIP(hipStreamSynchronize(st));
  }
  return ROM_OK;
}

// n1p x n1p fp64 table (zero padded) from a long double matrix of at most that size
void put_table(std::vector<double>& pool, size_t idx, int n1p, const Mat& A, bool transposed) {
  double* dst = pool.data() + idx * size_t(n1p) * n1p;
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < A.c; ++j) {
      if (transposed) dst[size_t(j) * n1p + i] = double(A(i, j));
      else dst[size_t(i) * n1p + j] = double(A(i, j));
    }
}

}  // namespace

extern "C" int rom_fem_destroy(rom_fem* f) {
  if (!f) return ROM_OK;
  hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->d_A0, f->d_G, f->d_Gs, f->d_Qp, f->d_kmax, f->d_epos, f->d_yhat, f->d_dots, f->d_W, f->d_g, f->d_desc, f->d_terms, f->d_pool,
                  f->d_kptr, f->d_kpair, f->d_colptr, f->d_colrow, f->d_colti, f->d_sides, f->d_vmap, f->d_L,
                  f->d_invL, f->d_y, f->d_Bt, f->d_P, f->d_vec, f->d_rhs, f->d_pre, f->d_exp, f->d_xred, f->d_groups, f->d_cm,
                  f->d_item_group, f->d_item_k, f->d_item_cf, f->d_ctask, f->d_pairs, f->d_alist, f->d_aoff, f->d_pool_acc, f->d_wmeta, f->d_s1_items, f->d_s1_citems, f->d_lr_blocks, f->d_gen_blocks, f->d_scat, f->d_dgroups, f->d_dweight, f->d_ditem_group,
                  f->d_ditem_k, f->d_dmat, f->d_scb};
  for (void* p : ptrs)
    if (p) hipFree(p);
  rom_factored_map_free(f->fmap);
  delete f;
  return ROM_OK;
}

// ROMHC_VERBOSE: wall time of the phases of rom_fem_create
struct PhaseTimer {
  bool on = getenv("ROMHC_VERBOSE") != nullptr;
  const char* name = nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void next(const char* n) {
    const auto t1 = std::chrono::steady_clock::now();
    if (on && name) fprintf(stderr, "romhc:   [%7.3f s] %s\n", std::chrono::duration<double>(t1 - t0).count(), name);
    name = n;
    t0 = t1;
  }
};
#define ROMHC_PHASE(N_) phase_timer.next(N_)

extern "C" int rom_fem_create(rom_ctx* ctx, int nrb, int ncb, int N, rom_fem** out) {
  ROM_CHECK(ctx && out, "rom_fem_create: null argument");
  ROM_CHECK(nrb >= 1 && ncb >= 1 && N >= 2, "rom_fem_create: need nrb,ncb >= 1 and N >= 2 (got %d,%d,%d)", nrb, ncb, N);
  ROM_CHECK(nrb * ncb <= 64, "rom_fem_create: at most 64 blocks supported (got %d)", nrb * ncb);
  ROM_HIP(hipSetDevice(ctx->device));
  PhaseTimer phase_timer;
  ROMHC_PHASE("geometry");
  rom_fem* f = new rom_fem();
  f->ctx = ctx;
  f->nrb = nrb; f->ncb = ncb; f->N = N;
  const int n1 = N - 1;
  f->n1 = n1;
  f->n1p = (n1 + TB - 1) / TB * TB;
  f->nr = nrb * N - 1;
  f->nc = ncb * N - 1;
  f->dim = int64_t(f->nr) * f->nc;
  const int n1p = f->n1p;

  ROMHC_PHASE("edges, crosses");
  // ---- edges, crosses ------------------------------------------------------------------------
  std::vector<Edge> edges;
  std::map<std::pair<int, int>, int> hid, vid, xid;
  for (int p = 1; p < nrb; ++p)
    for (int q = 0; q < ncb; ++q) {
      hid[{p, q}] = int(edges.size());
      edges.push_back({0, p, q, (p - 1) * ncb + q, p * ncb + q});
    }
  for (int q = 1; q < ncb; ++q)
    for (int p = 0; p < nrb; ++p) {
      vid[{p, q}] = int(edges.size());
      edges.push_back({1, p, q, p * ncb + (q - 1), p * ncb + q});
    }
  std::vector<std::pair<int, int>> crosses;
  for (int p = 1; p < nrb; ++p)
    for (int q = 1; q < ncb; ++q) {
      xid[{p, q}] = int(crosses.size());
      crosses.push_back({p, q});
    }
  const int E = int(edges.size());
  const int ncross = int(crosses.size());
  f->nG = E * n1 + ncross;
  f->ncross = ncross;
  // block -> side -> edge id   (sides: 0 top, 1 bottom, 2 left, 3 right)
  std::vector<std::array<int, 4>> bside(nrb * ncb);
  for (int p = 0; p < nrb; ++p)
    for (int q = 0; q < ncb; ++q) {
      auto& s = bside[p * ncb + q];
      s[0] = p >= 1 ? hid[{p, q}] : -1;
      s[1] = p + 1 < nrb ? hid[{p + 1, q}] : -1;
      s[2] = q >= 1 ? vid[{p, q}] : -1;
      s[3] = q + 1 < ncb ? vid[{p, q + 1}] : -1;
    }
  auto side_of = [&](int blk, int e) {
    for (int s = 0; s < 4; ++s)
      if (bside[blk][s] == e) return s;
    return -1;
  };

  ROMHC_PHASE("cross <-> edge-end couplings");
  // ---- cross <-> edge-end couplings ----------------------------------------------------------------
  struct XCpl { int cross, edge, node; };  // node: 0-based local node on the edge
  std::vector<XCpl> xc;
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    if (ed.hv == 0) {
      if (ed.q >= 1) xc.push_back({xid[{ed.p, ed.q}], e, 0});
      if (ed.q + 1 < ncb) xc.push_back({xid[{ed.p, ed.q + 1}], e, n1 - 1});
    } else {
      if (ed.p >= 1) xc.push_back({xid[{ed.p, ed.q}], e, 0});
      if (ed.p + 1 < nrb) xc.push_back({xid[{ed.p + 1, ed.q}], e, n1 - 1});
    }
  }
  // block adjacency of the edges
  std::vector<std::set<int>> adj(E);
  for (auto& s : bside)
    for (int x = 0; x < 4; ++x)
      for (int y = 0; y < 4; ++y)
        if (x != y && s[x] >= 0 && s[y] >= 0) adj[s[x]].insert(s[y]);
  auto shared_block = [&](int e1, int e2) {  // the one block two distinct edges can share, or -1
    for (int b1 : {edges[e1].b0, edges[e1].b1})
      for (int b2 : {edges[e2].b0, edges[e2].b1})
        if (b1 == b2) return b1;
    return -1;
  };

  ROMHC_PHASE("edges eliminated in closed form: a maximal set n");
  // ---- edges eliminated in closed form: a maximal set no two of which touch the same block (greedy);
  //      their self-interaction is (a_b0 + a_b1) K with K parameter independent and they do not couple
  //      to each other ------------------------------------------------------------------------------------
  std::vector<char> is_pre(E, 0);
  if (!getenv("ROMHC_NO_PREELIM")) {
    std::vector<char> busy(nrb * ncb, 0);
    for (int e = 0; e < E; ++e)
      if (!busy[edges[e].b0] && !busy[edges[e].b1]) { is_pre[e] = 1; busy[edges[e].b0] = busy[edges[e].b1] = 1; }
  }
  std::vector<int> pre_list, pre_index(E, -1);
  for (int e = 0; e < E; ++e)
    if (is_pre[e]) { pre_index[e] = int(pre_list.size()); pre_list.push_back(e); }
  const int npre = int(pre_list.size());
  const int nact = E - npre;

  ROMHC_PHASE("elimination order of the active edges (greedy mi");
  // ---- elimination order of the active edges (greedy minimum degree on the graph: shared block, or
  //      common neighbour of a closed-form edge) ----------------------------------------------------------
  std::vector<int> order, ord_of(E, -1);
  {
    std::vector<std::set<int>> g(E);
    for (int e = 0; e < E; ++e)
      if (!is_pre[e])
        for (int x : adj[e])
          if (!is_pre[x]) g[e].insert(x);
    for (int e : pre_list)
      for (int x : adj[e])
        for (int y : adj[e])
          if (x != y && !is_pre[x] && !is_pre[y]) g[x].insert(y);
    std::vector<char> done(E, 0);
    for (int step = 0; step < nact; ++step) {
      int best = -1;
      size_t bd = 0;
      for (int e = 0; e < E; ++e) {
        if (done[e] || is_pre[e]) continue;
        if (best < 0 || g[e].size() < bd) { best = e; bd = g[e].size(); }
      }
      done[best] = 1;
      ord_of[best] = int(order.size());
      order.push_back(best);
      std::vector<int> nb(g[best].begin(), g[best].end());
      for (int x : nb) {
        g[x].erase(best);
        for (int y : nb)
          if (x != y) g[x].insert(y);
      }
    }
  }

  ROMHC_PHASE("unit-block tables in long double, compression of");
  // ---- unit-block tables in long double, compression of the edges --------------------------------------------
  UnitBlock ub(N, E > 0);
  const Mat& Q = ub.Q;
  const Mat& Kinv = ub.Kinv;
  const std::vector<ld>* gE = ub.gE;
  const std::vector<double>& rho_d = ub.rho_d;
  const std::vector<double>& Wd = ub.Wd;
  const double h2 = 1.0 / (double(N) * double(N));
  auto Tm = [&](int id) -> const Mat& { return ub.Tm(id); };
  auto TK = [&](int id) -> const Mat& { return ub.TK(id); };
  const bool compress = !getenv("ROMHC_NO_COMPRESS");
  // Numerical rank of an edge's coupling tables.  Directions weaker than `ctol` = 1e-14 of the strongest are dropped
  // where that removes work: the tables are rounded to fp64 on upload, and against a basis kept down to 1e-17 the
  // snapshots move by <= 1.2e-14 relative over seven geometries and contrasts up to 1e8 (1e-13: 9e-14; the distance to the
  // reference SuperLU solve, 1e-13..1e-12, does not change in its first three digits) while the reduced system shrinks from 347
  // to 301 unknowns at C4 (6 -> 5 tile columns) and from 801 to 697 at C5 (13 -> 11): -19 % per step
  // (profiles/r02_compress_tolerance.txt).  Where it removes nothing -- the weaker directions fit into the padding of the
  // extension's 8-wide K segments and add no tile column to the reduced matrix (C2) -- they are kept, down to `ckeep`.
  ld ctol = 1e-14L;
  if (const char* s = getenv("ROMHC_COMPRESS_TOL")) ctol = (ld)atof(s);
  const ld ckeep = std::min<ld>(ctol, 1e-17L);
  std::map<std::vector<int>, int> sig_id;  // edges with the same surroundings share one compressed form
  std::vector<std::vector<int>> sigs;
  std::vector<int> comp_of(E, -1);
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    std::vector<int> sig{ed.hv};
    for (int blk : {ed.b0, ed.b1}) {
      const int sf = side_of(blk, e);
      for (int s2 = 0; s2 < 4; ++s2)
        if (s2 != sf && bside[blk][s2] >= 0) sig.push_back(sf * 4 + s2);
    }
    bool x0 = false, x1 = false;
    for (auto& c : xc)
      if (c.edge == e) (c.node == 0 ? x0 : x1) = true;
    sig.push_back(100 + (x0 ? 1 : 0) + (x1 ? 2 : 0));
    auto it = sig_id.find(sig);
    if (it == sig_id.end()) {
      it = sig_id.emplace(sig, int(sigs.size())).first;
      sigs.push_back(sig);
    }
    comp_of[e] = it->second;
  }
  // the edge types are independent: one host thread each (the tables they read are built first)
  std::vector<Comp> comps(sigs.size());
  {
    std::vector<int> ids, ids_tk;
    for (auto& sig : sigs)
      for (size_t t = 1; t + 1 < sig.size(); ++t) ids.push_back(sig[t]);
    for (int e : pre_list)  // the closed-form edges also need T K^-1 towards their neighbours
      for (int u : adj[e]) {
        const int blk = shared_block(e, u);
        ids_tk.push_back(side_of(blk, u) * 4 + side_of(blk, e));
        ids.push_back(side_of(blk, e) * 4 + side_of(blk, u));
      }
    ub.prepare(ids, false);
    ub.prepare(ids_tk, true);
    std::vector<char> ok(sigs.size(), 1);
    std::vector<Mat> Wfull(sigs.size());
    std::vector<std::vector<ld>> pivots(sigs.size());
    if (compress)
      parallel_for(sigs.size(), [&](size_t c) {
        const std::vector<int>& sig = sigs[c];
        const int flags = sig.back() - 100;
        compress_basis(ub, std::vector<int>(sig.begin() + 1, sig.end() - 1), flags & 1, flags & 2, ckeep, Wfull[c], pivots[c]);
      });
    // the ranks: at `ctol`, or -- if that costs neither a K segment nor a tile column -- up to the end of the last segment
    std::vector<int> r_drop(sigs.size(), n1), r_use(sigs.size(), n1);
    if (compress) {
      std::vector<int> r_fill(sigs.size(), n1);
      for (size_t c = 0; c < sigs.size(); ++c) {
        int r = 0;
        while (r < int(pivots[c].size()) && pivots[c][r] > ctol * pivots[c][0]) ++r;
        r_drop[c] = r;
        r_fill[c] = std::min<int>(int(pivots[c].size()), (r + 1 + 7) / 8 * 8 - 1);
      }
      auto tiles = [&](const std::vector<int>& rr) {
        long long nred_ = ncross;
        for (int e : order) nred_ += std::min(rr[comp_of[e]], n1);
        return (nred_ + TB - 1) / TB;
      };
      r_use = tiles(r_fill) <= tiles(r_drop) ? r_fill : r_drop;
    }
    parallel_for(sigs.size(), [&](size_t c) {
      ok[c] = compress_finish(ub, Wfull[c], r_use[c], sigs[c][0], compress, comps[c]);
    });
    for (char o : ok)
      if (!o) { rom_set_error("internal: compressed edge block not positive definite"); return ROM_ERR_INVALID; }
  }

  // kmax[d]: sine modes with rho_mode(d) >= 1e-18 (rounded up to the K chunk): what the extension needs at
  // distance d from a side.  A compressed edge enters the extension through its reduced unknowns instead
  // when rank + 1 (padded) is not much above the average mode count.
  std::vector<int> kmax(N + 1, n1p);
  double kavg = 0;
  for (int dd = 1; dd <= N; ++dd) {
    int last = -1;
    for (int m = 0; m < n1; ++m)
      if (rho_d[size_t(m) * (N + 1) + std::min(dd, N)] >= 1e-18) last = m;
    kmax[dd] = std::min(n1p, std::max(BK, (last + 1 + BK - 1) / BK * BK));
    if (dd <= n1) kavg += kmax[dd] / double(std::max(n1, 1));
  }
  kmax[0] = n1p;
  std::vector<int> rp(comps.size(), 0);
  std::vector<char> use_lr(comps.size(), 0);
  for (size_t c = 0; c < comps.size(); ++c) {
    rp[c] = (comps[c].r + 1 + BK - 1) / BK * BK;
    // (up to a quarter more K than the truncated modes is still a gain: flat K, wide tiles, no edge transforms)
    use_lr[c] = comps[c].r < n1 && rp[c] <= 1.25 * kavg && n1 > 0 && !getenv("ROMHC_NO_LOWRANK_EXT");
  }

  ROMHC_PHASE("layout of the reduced vector: edge groups in eli");
  // ---- layout of the reduced vector: edge groups in elimination order, every cross point right behind
  //      the adjacent active edge that is eliminated last ------------------------------------------------------
  std::vector<int> zpos(E, -1), rk(E, 0), xred(ncross, -1), xhost(ncross, -1);
  for (int x = 0; x < ncross; ++x)
    for (auto& c : xc)
      if (c.cross == x && !is_pre[c.edge] && (xhost[x] < 0 || ord_of[c.edge] > ord_of[xhost[x]])) xhost[x] = c.edge;
  int nred = 0;
  {
    // one tile in total: cross points first, so that their couplings are table ROWS of the upper triangle
    // (the single-tile assembly reads row segments; a cross behind its edges would cost one 8-byte read per
    // edge row instead)
    int total = ncross;
    for (int e : order) total += comps[comp_of[e]].r;
    if (total <= TB)
      for (int x = 0; x < ncross; ++x) xred[x] = nred++;
  }
  for (int e : order) {
    zpos[e] = nred;
    rk[e] = comps[comp_of[e]].r;
    nred += rk[e];
    f->ranks.push_back(rk[e]);
    for (int x = 0; x < ncross; ++x)
      if (xhost[x] == e && xred[x] < 0) xred[x] = nred++;
  }
  for (int x = 0; x < ncross; ++x)
    if (xred[x] < 0) xred[x] = nred++;
  const int T = (nred + TB - 1) / TB;
  f->nred = nred;
  f->T = T;
  f->nGa = T * TB;
  // nodal layout behind the reduced part: one n1p block per edge, then the cross block
  std::vector<int> npos(E, -1);
  for (int e = 0; e < E; ++e) npos[e] = f->nGa + e * n1p;
  f->xb0 = f->nGa + E * n1p;
  f->nGp = f->xb0 + (ncross > 0 ? (ncross + TB - 1) / TB * TB : 0);
  std::vector<int> cpos(E, -1);  // [z_f, 1/s_f] blocks of the edges that enter the extension in compressed form
  for (int e : order)
    if (use_lr[comp_of[e]]) {
      cpos[e] = f->nGp;
      f->nGp += rp[comp_of[e]];
    }
  for (int e : pre_list)  // closed-form edges kept in compressed form: [c_e / s_e, 1/s_e]
    if (use_lr[comp_of[e]]) {
      cpos[e] = f->nGp;
      f->nGp += rp[comp_of[e]];
    }
  // scalar block: 1/(a_p + a_q) of every edge, then h^2/a_b of every block -- with it the expansion stage is a
  // LINEAR map of the interface vector (it never reads the parameters)
  f->spos0 = f->nGp;
  f->n_all_edges = E;
  f->nsc = E + nrb * ncb;
  f->nGp += (f->nsc + BK - 1) / BK * BK;

  // The r x n1 x n1 long-double products W_c^T T (and W_c^T T K^-1) of the next two phases depend only on (edge type c,
  // table id): the edges of a regular grid ask for the same few again and again (1.2 s of the 1.7 s of a 4x4 / N=256
  // setup went into recomputing them one after the other).  Collect the distinct ones, compute them on all host
  // threads, look them up in the loops.
  std::map<std::tuple<int, int, int>, Mat> wt_cache;  // (edge type, table id, 0: T | 1: T K^-1) -> W^T table
  {
    std::vector<std::tuple<int, int, int>> keys;
    for (int e : order)
      for (int e2 : adj[e]) {
        if (is_pre[e2] || e2 <= e) continue;
        const int blk = shared_block(e, e2);
        keys.emplace_back(comp_of[e], side_of(blk, e) * 4 + side_of(blk, e2), 0);
      }
    for (int i = 0; i < npre; ++i) {
      const int e = pre_list[i];
      for (int u : adj[e]) {
        const int blk = shared_block(e, u);
        const int id = side_of(blk, u) * 4 + side_of(blk, e);
        keys.emplace_back(comp_of[u], id, 0);
        keys.emplace_back(comp_of[u], id, 1);
        if (cpos[e] >= 0) keys.emplace_back(comp_of[e], side_of(blk, e) * 4 + side_of(blk, u), 0);
      }
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    // the tables themselves are built lazily: make sure every one that is needed exists (distinct ids concurrently)
    std::vector<std::pair<int, int>> tabs;
    for (const auto& [c, id, which] : keys) tabs.emplace_back(id, which);
    std::sort(tabs.begin(), tabs.end());
    tabs.erase(std::unique(tabs.begin(), tabs.end()), tabs.end());
    std::vector<std::pair<int, int>> tab_ids;  // one entry per id (with K^-1 if any key wants it): a build covers both
    for (const auto& t : tabs) {
      if (!tab_ids.empty() && tab_ids.back().first == t.first) tab_ids.back().second |= t.second;
      else tab_ids.push_back(t);
    }
    parallel_for(tab_ids.size(), [&](size_t k) {
      if (tab_ids[k].second) TK(tab_ids[k].first);
      Tm(tab_ids[k].first);
    });
    std::vector<Mat> vals(keys.size());
    parallel_for(keys.size(), [&](size_t k) {
      const auto& [c, id, which] = keys[k];
      vals[k] = hostla::mul_tn(comps[c].W, which ? TK(id) : Tm(id));
    });
    for (size_t k = 0; k < keys.size(); ++k) wt_cache.emplace(keys[k], std::move(vals[k]));
  }
  ROMHC_PHASE("blocks of the reduced matrix");
  // ---- blocks of the reduced matrix --------------------------------------------------------------------
  std::vector<Small> smalls;
  auto add_small = [&](int rpos, int cpos, const Mat& tab, int kind, std::array<int, 4> b) {
    smalls.push_back(Small{rpos, cpos, tab, kind, b});
    if (rpos != cpos) smalls.push_back(Small{cpos, rpos, hostla::transpose(tab), kind, b});
  };
  for (int e : order) {
    const Edge& ed = edges[e];
    const Comp& ce = comps[comp_of[e]];
    add_small(zpos[e], zpos[e], ce.Kt, 1, {ed.b0, ed.b1, 0, 0});
    for (int e2 : adj[e]) {
      if (is_pre[e2] || e2 <= e) continue;
      const int blk = shared_block(e, e2);
      const int id = side_of(blk, e) * 4 + side_of(blk, e2);
      const Comp& c2 = comps[comp_of[e2]];
      add_small(zpos[e], zpos[e2], hostla::mul(wt_cache.at({comp_of[e], id, 0}), c2.W), 0, {blk, 0, 0, 0});
    }
  }
  for (auto& c : xc) {
    if (is_pre[c.edge]) continue;  // folded into the closed-form tables
    const Comp& ce = comps[comp_of[c.edge]];
    Mat row(1, ce.r);
    for (int k = 0; k < ce.r; ++k) row(0, k) = ce.W(c.node, k);
    add_small(xred[c.cross], zpos[c.edge], row, 2, {edges[c.edge].b1, edges[c.edge].b0, 0, 0});
  }
  for (int x = 0; x < ncross; ++x) {
    const int p = crosses[x].first, q = crosses[x].second;
    Mat one(1, 1);
    one(0, 0) = 1.0L;
    add_small(xred[x], xred[x], one, 3, {(p - 1) * ncb + (q - 1), (p - 1) * ncb + q, p * ncb + (q - 1), p * ncb + q});
  }

  ROMHC_PHASE("closed-form edges: neighbours, reduced-matrix bl");
  // ---- closed-form edges: neighbours, reduced-matrix blocks, rhs terms, back substitution ---------------------------
  std::vector<double> vecs;  // vector table
  auto push_vec = [&](const std::vector<ld>& v, int padded) {
    const int off = int(vecs.size());
    for (ld x : v) vecs.push_back(double(x));
    for (int i = int(v.size()); i < padded; ++i) vecs.push_back(0.0);
    return off;
  };
  std::vector<RhsTerm> rhs_terms;
  std::vector<PreEdge> pre_edges;   // closed-form edges recovered node by node (k_back_pre)
  std::vector<CoefGroup> groups;    // coefficient blocks built by k_coef
  std::vector<double> cm;           // matrices of the closed-form edges kept in compressed form
  std::map<int, int> bt_of_id;               // T table id -> B^T table index
  std::vector<std::pair<int, Mat>> bt_extra;  // cross-block tables (index, n1 x ncross)
  int nbt = 0;
  double pre_flops = 0;
  // the long-double products of every closed-form edge first, one host thread per edge (they read shared tables only);
  // the bookkeeping below, which appends to the shared lists in a fixed order, then just picks them up
  struct Ent { int pos, len, blk; Mat X, Y; };  // X: len x n1 coupling to e (without its weight), Y = X K^-1
  struct PreWork {
    std::vector<ld> we;              // K^-1 g_e
    std::vector<Ent> ents;           // neighbours (order of adj[e]), then the cross points on e (order of xc)
    std::vector<Mat> Mt;             // per neighbour (compressed e only): (W_e^T T P_u)^T
    std::vector<std::vector<ld>> mtv;  //                                    W_e^T T p0_u
    Mat Btx;
    bool has_x = false;
    std::vector<std::vector<ld>> rhsv;  // per entity: Y g_e
    std::vector<Mat> R;                 // Y_u X_v^T for v <= u, in the order of the loops below
  };
  std::vector<PreWork> prework(npre);
  parallel_for(size_t(npre), [&](size_t i) {
    const int e = pre_list[i];
    const Edge& pe = edges[e];
    const bool lr_e = cpos[e] >= 0;
    PreWork& w = prework[i];
    w.we = hostla::matvec(Kinv, gE[pe.hv]);
    for (int u : adj[e]) {
      const int blk = shared_block(e, u);
      const int id = side_of(blk, u) * 4 + side_of(blk, e);
      const Comp& cu = comps[comp_of[u]];
      w.ents.push_back(Ent{zpos[u], cu.r, blk, wt_cache.at({comp_of[u], id, 0}), wt_cache.at({comp_of[u], id, 1})});
      if (lr_e) {
        const Mat& WT = wt_cache.at({comp_of[e], side_of(blk, e) * 4 + side_of(blk, u), 0});  // r_e x n1
        w.Mt.push_back(hostla::transpose(hostla::mul(WT, cu.P)));                            // r_u x r_e
        w.mtv.push_back(hostla::matvec(WT, cu.p0));
      }
    }
    w.Btx = Mat(n1, std::max(ncross, 1));
    for (auto& c : xc) {
      if (c.edge != e) continue;
      Ent en{xred[c.cross], 1, -1, Mat(1, n1), Mat(1, n1)};
      en.X(0, c.node) = 1.0L;
      for (int k = 0; k < n1; ++k) {
        en.Y(0, k) = Kinv(c.node, k);
        w.Btx(k, c.cross) = Kinv(k, c.node);
      }
      w.ents.push_back(std::move(en));
      w.has_x = true;
    }
    for (size_t u = 0; u < w.ents.size(); ++u) {
      w.rhsv.push_back(hostla::matvec(w.ents[u].Y, gE[pe.hv]));
      for (size_t v = 0; v <= u; ++v) {
        Mat R = hostla::mul_nt(w.ents[u].Y, w.ents[v].X);
        if (u == v)
          for (int a = 0; a < R.r; ++a)
            for (int b = 0; b < a; ++b) R(a, b) = R(b, a) = (R(a, b) + R(b, a)) / 2;
        w.R.push_back(std::move(R));
      }
    }
  });
  for (int i = 0; i < npre; ++i) {
    const int e = pre_list[i];
    const Edge& pe = edges[e];
    const bool lr_e = cpos[e] >= 0;
    PreEdge P;
    memset(&P, 0, sizeof(P));
    CoefGroup cg;
    memset(&cg, 0, sizeof(cg));
    const Comp& cpe = comps[comp_of[e]];
    cg.kind = 1; cg.cpos = cpos[e]; cg.r = cpe.r; cg.w = rp[comp_of[e]]; cg.b0 = pe.b0; cg.b1 = pe.b1;
    auto add_cterm = [&](int src, int blk, const Mat& Mt, int voff, int u0, int u1) {  // Mt: len x r_e
      if (cg.nterm >= 8) return false;
      cg.t[cg.nterm++] = CoefTerm{src, Mt.r, blk, int(cm.size()), voff, u0, u1};
      for (ld v : Mt.v) cm.push_back(double(v));
      return true;
    };
    P.pos = npos[e];
    P.e0 = pe.b0;
    P.e1 = pe.b1;
    const PreWork& pw = prework[i];
    P.woff = push_vec(pw.we, n1p);
    const std::vector<Ent>& ents = pw.ents;
    size_t nbi = 0;  // neighbour counter (index into pw.Mt / pw.mtv)
    for (int u : adj[e]) {
      const int blk = shared_block(e, u);
      const int id = side_of(blk, u) * 4 + side_of(blk, e);
      const Comp& cu = comps[comp_of[u]];
      if (lr_e) {
        // c_e += a_blk * (W_e^T T^(e,u) P_u z_u + W_e^T T^(e,u) p0_u / s_u)
        const Mat& Mt = pw.Mt[nbi];
        const int voff = push_vec(pw.mtv[nbi], cpe.r);
        ++nbi;
        if (!add_cterm(zpos[u], blk, Mt, voff, edges[u].b0, edges[u].b1)) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
        pre_flops += 2.0 * cpe.r * double(cu.r);
      } else {
        if (!bt_of_id.count(id)) bt_of_id[id] = nbt++;
        if (P.nnb >= 8) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
        P.nb[P.nnb++] = PreNb{npos[u], blk, n1p / BK, bt_of_id[id]};
        pre_flops += 2.0 * n1p * double(n1p);
      }
    }
    const Mat& Btx = pw.Btx;
    const bool has_x = pw.has_x;
    for (auto& c : xc) {
      if (c.edge != e) continue;
      if (lr_e) {  // c_e += (s_e / 2) W_e[node, :]^T u_x
        Mat Mt(1, cpe.r);
        for (int k = 0; k < cpe.r; ++k) Mt(0, k) = cpe.W(c.node, k);
        if (!add_cterm(xred[c.cross], -1, Mt, -1, 0, 0)) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
      }
    }
    if (lr_e) groups.push_back(cg);
    if (has_x && !lr_e) {
      if (P.nnb >= 8) { rom_set_error("internal: more than 8 neighbours of an eliminated edge"); return ROM_ERR_INVALID; }
      P.nb[P.nnb++] = PreNb{f->xb0, -1, (ncross + BK - 1) / BK, nbt};
      bt_extra.push_back({nbt++, Btx});
      pre_flops += 2.0 * n1p * double((ncross + BK - 1) / BK * BK);
    }
    size_t ri = 0;
    for (size_t u = 0; u < ents.size(); ++u) {
      const Ent& eu = ents[u];
      rhs_terms.push_back(RhsTerm{eu.pos, eu.len, push_vec(pw.rhsv[u], eu.len), eu.blk >= 0 ? 0 : 1,
                                  std::max(eu.blk, 0), pe.b0, pe.b1});
      for (size_t v = 0; v <= u; ++v) {
        const Ent& ev = ents[v];
        const Mat& R = pw.R[ri++];
        if (eu.blk >= 0 && ev.blk >= 0)
          add_small(eu.pos, ev.pos, R, 4, {std::min(eu.blk, ev.blk), std::max(eu.blk, ev.blk), pe.b0, pe.b1});
        else if (eu.blk >= 0 || ev.blk >= 0)
          add_small(eu.pos, ev.pos, R, 5, {std::max(eu.blk, ev.blk), 0, 0, 0});
        else
          add_small(eu.pos, ev.pos, R, 6, {0, 0, pe.b0, pe.b1});
      }
    }
    if (!lr_e) pre_edges.push_back(P);
  }
  f->npre = int(pre_edges.size());

  ROMHC_PHASE("tile mask + symbolic fill");
  // ---- tile mask + symbolic fill --------------------------------------------------------------------
  std::vector<char> mask(size_t(T) * T, 0);
  auto M_ = [&](int i, int j) -> char& { return mask[size_t(i) * T + j]; };
  for (int t = 0; t < T; ++t) M_(t, t) = 1;
  for (const Small& s : smalls)
    for (int tr = s.rpos / TB; tr <= (s.rpos + s.tab.r - 1) / TB; ++tr)
      for (int tc = s.cpos / TB; tc <= (s.cpos + s.tab.c - 1) / TB; ++tc) M_(tr, tc) = M_(tc, tr) = 1;
  for (int k = 0; k < T; ++k)
    for (int i = k + 1; i < T; ++i)
      if (M_(i, k))
        for (int j = k + 1; j <= i; ++j)
          if (M_(j, k)) M_(i, j) = M_(j, i) = 1;

  f->slot_of.assign(size_t(T) * T, -1);
  std::vector<std::pair<int, int>> slots;
  f->colptr.assign(T + 1, 0);
  f->diag_slot.assign(T, -1);
  for (int j = 0; j < T; ++j) {
    f->diag_slot[j] = int(slots.size());
    f->slot_of[size_t(j) * T + j] = int(slots.size());
    slots.push_back({j, j});
    for (int i = j + 1; i < T; ++i)
      if (M_(i, j)) {
        f->slot_of[size_t(i) * T + j] = int(slots.size());
        f->colrow.push_back(int(slots.size()));
        f->colti.push_back(i);
        slots.push_back({i, j});
      }
    f->colptr[j + 1] = int(f->colrow.size());
  }
  f->nslots = int(slots.size());
  f->kptr.assign(f->nslots + 1, 0);
  double flops = 0;
  for (int s = 0; s < f->nslots; ++s) {
    int i = slots[s].first, j = slots[s].second;
    for (int k = 0; k < j; ++k)
      if (M_(i, k) && M_(j, k)) {
        f->kpair.push_back(f->slot_of[size_t(i) * T + k]);
        f->kpair.push_back(f->slot_of[size_t(j) * T + k]);
        flops += 2.0 * TB * TB * TB;
      }
    f->kptr[s + 1] = int(f->kpair.size() / 2);
    flops += (i == j) ? TB * double(TB) * TB / 3.0 : 2.0 * TB * TB * TB;  // potrf | trsm-as-gemm
  }

  ROMHC_PHASE("distribute the blocks over the tiles: one 64x64 ");
  // ---- distribute the blocks over the tiles: one 64x64 table per (tile, coefficient formula) ---------------------
  std::vector<std::vector<TermAcc>> slot_terms(f->nslots);
  for (const Small& s : smalls)
    for (int tr = s.rpos / TB; tr <= (s.rpos + s.tab.r - 1) / TB; ++tr)
      for (int tc = s.cpos / TB; tc <= (s.cpos + s.tab.c - 1) / TB && tc <= tr; ++tc) {
        const int slot = f->slot_of[size_t(tr) * T + tc];
        if (slot < 0) { rom_set_error("internal: reduced-matrix block outside the tile mask"); return ROM_ERR_INVALID; }
        const std::array<int, 5> key{s.kind, s.b[0], s.b[1], s.b[2], s.b[3]};
        TermAcc* ta = nullptr;
        for (auto& cand : slot_terms[slot])
          if (cand.key == key) ta = &cand;
        if (!ta) {
          slot_terms[slot].push_back(TermAcc{key, std::vector<double>(4096, 0.0), TB, 0, TB, 0});
          ta = &slot_terms[slot].back();
        }
        const int i0 = std::max(s.rpos, tr * TB), i1 = std::min(s.rpos + s.tab.r, (tr + 1) * TB);
        const int j0 = std::max(s.cpos, tc * TB), j1 = std::min(s.cpos + s.tab.c, (tc + 1) * TB);
        for (int i = i0; i < i1; ++i)
          for (int j = j0; j < j1; ++j)
            ta->tab[size_t(i - tr * TB) * TB + (j - tc * TB)] += double(s.tab(i - s.rpos, j - s.cpos));
        ta->r_lo = std::min(ta->r_lo, i0 - tr * TB);
        ta->r_hi = std::max(ta->r_hi, i1 - tr * TB);
        ta->c_lo = std::min(ta->c_lo, j0 - tc * TB);
        ta->c_hi = std::max(ta->c_hi, j1 - tc * TB);
      }
  std::vector<GenTerm> terms;
  std::vector<double> pool;
  f->desc.resize(f->nslots);
  for (int s = 0; s < f->nslots; ++s) {
    TileDesc d;
    memset(&d, 0, sizeof(d));
    d.ti = slots[s].first;
    d.tj = slots[s].second;
    d.diag = d.ti == d.tj;
    d.ndr = std::max(0, std::min(TB, nred - d.ti * TB));
    d.t0 = int(terms.size());
    for (auto& ta : slot_terms[s]) {
      GenTerm g;
      g.tab = int(pool.size() / 4096);
      g.r_lo = short(ta.r_lo); g.r_hi = short(ta.r_hi); g.c_lo = short(ta.c_lo); g.c_hi = short(ta.c_hi);
      g.kind = ta.key[0];
      for (int q = 0; q < 4; ++q) g.b[q] = ta.key[1 + q];
      terms.push_back(g);
      pool.insert(pool.end(), ta.tab.begin(), ta.tab.end());
    }
    d.t1 = int(terms.size());
    f->desc[s] = d;
  }
  slot_terms.clear();
  smalls.clear();
  // The assembly of a tile as a stream of kilobytes (s_tile_to_lds): wave w of a workgroup owns rows 16 w .. 16 w + 15 as
  // 2 x 4 positions of 8 rows x 16 columns; per tile slot and wave, the (position x, term) pieces that meet the term's
  // rectangle, sorted by position, the terms of a position in their order; padded to whole rings with no-ops, 128 no-ops behind the end.
  std::vector<int> alist, aoff;
  {
    const int RING = 8;
    for (size_t sl = 0; sl < f->desc.size(); ++sl) {
      const TileDesc& d = f->desc[sl];
      for (int w = 0; w < 4; ++w) {
        aoff.push_back(int(alist.size() / 2));
        if (d.t1 - d.t0 > 128) continue;  // (such a tile takes the register path)
        for (int x = 0; x < 8; ++x) {  // position x = 4 pr + cs: rows 16 w + 8 pr .. + 7, columns 16 cs .. + 15
          const int r0 = 16 * w + 8 * (x >> 2), c0 = 16 * (x & 3);
          const size_t first = alist.size();
          for (int t = d.t0; t < d.t1; ++t) {
            const GenTerm& g = terms[t];
            if (!(r0 + 8 > g.r_lo && r0 < g.r_hi && c0 + 16 > g.c_lo && c0 < g.c_hi)) continue;
            alist.push_back(int(size_t(g.tab) * 4096 + size_t(r0) * TB + c0));
            alist.push_back(x | (t - d.t0) << 8);
          }
          if (alist.size() > first) alist.back() |= 1 << 16;  // last piece of this position
        }
        while ((alist.size() / 2 - size_t(aoff.back())) % RING) { alist.push_back(0); alist.push_back(1 << 17); }
      }
    }
    aoff.push_back(int(alist.size() / 2));
    for (int i = 0; i < 128; ++i) { alist.push_back(0); alist.push_back(1 << 17); }
    if (getenv("ROMHC_VERBOSE")) {
      size_t real = 0;
      for (size_t i = 1; i < alist.size(); i += 2) real += (alist[i] >> 17) ? 0 : 1;
      fprintf(stderr, "romhc:   tile assembly as a stream: %zu KB per system and sweep in pieces of 8 rows x 16 columns\n", real);
    }
  }
  if (getenv("ROMHC_VERBOSE")) {  // what the assembly of the tiles reads: 128-byte strips (one thread-row x 16 columns) that meet a term's rectangle
    double strips = 0, tiles_diag = 0, tiles_sub = 0, nterm = 0;
    for (const TileDesc& d : f->desc) {
      (d.ti == d.tj ? tiles_diag : tiles_sub) += 1;
      for (int t = d.t0; t < d.t1; ++t) {
        const GenTerm& g = terms[t];
        nterm += 1;
        for (int r = g.r_lo; r < g.r_hi; ++r)
          for (int c0 = 0; c0 < 64; c0 += 16)
            if (c0 < g.c_hi && c0 + 16 > g.c_lo) strips += 1;
      }
    }
    fprintf(stderr, "romhc:   tile assembly: %.0f diagonal + %.0f sub-diagonal tiles, %.0f terms, %.1f KB of table strips per system and sweep\n",
            tiles_diag, tiles_sub, nterm, strips * 128 / 1024);
  }
  // the whole reduced solve in one wave (k_solve1) if the reduced matrix is a single tile; its assembly walks
  // the (term, 16x16 block) pairs whose rectangle and block intersect
  f->fused1 = T == 1 && f->desc[0].t1 - f->desc[0].t0 < COEF_MAX && f->nGa == TB;
  std::vector<int> pairs;
  if (f->fused1) {
    const TileDesc& d0 = f->desc[0];
    int q = 0;
    for (int ib = 0; ib < 4; ++ib)
      for (int jb = 0; jb <= ib; ++jb, ++q) {  // sorted by block: the kernel keeps a block's sum in registers
        const size_t first = pairs.size();
        for (int t = d0.t0; t < d0.t1; ++t) {
          const GenTerm& g = terms[t];
          bool any = false;  // (skip blocks where the table is all zero inside the rectangle, too)
          for (int r = std::max<int>(g.r_lo, 16 * ib); r < std::min<int>(g.r_hi, 16 * ib + 16) && !any; ++r)
            for (int c = std::max<int>(g.c_lo, 16 * jb); c < std::min<int>(g.c_hi, 16 * jb + 16); ++c)
              if (pool[size_t(g.tab) * 4096 + size_t(r) * TB + c] != 0.0) { any = true; break; }
          if (!any) continue;
          pairs.push_back(int(size_t(g.tab) * 4096 + size_t(16 * ib) * TB + 16 * jb));
          pairs.push_back((t - d0.t0) | (q << 8));
        }
        if (pairs.size() > first) pairs.back() |= 1 << 16;  // last pair of this block
      }
  }
  // k_solve1 runs four systems per workgroup and deals the BLOCKS to its four waves (largest first, to the wave with the
  // fewest pairs so far); wave w walks pairs wp0[w] .. wp0[w + 1] - 1 of `wmeta`, whose table pieces lie in the same order in
  // `pool_acc`: piece = the 16 x 16 block of the pair's table in the accumulator layout of its consumer,
  // [g pair h][lane][e] = table(16 ib + 4 (2 h + e) + (lane >> 4), 16 jb + (lane & 15))
  std::vector<double> pool_acc;
  std::vector<int> wmeta;
  if (f->fused1) {
    const int npr = int(pairs.size() / 2);
    std::vector<std::vector<int>> of_block(10), of_wave(4);
    for (int i = 0; i < npr; ++i) of_block[(pairs[2 * i + 1] >> 8) & 0xff].push_back(i);
    std::vector<int> order(10), load(4, 0);
    for (int q = 0; q < 10; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return of_block[x].size() > of_block[y].size(); });
    std::vector<std::vector<int>> blocks_of(4);
    for (int q : order) {
      const int w = int(std::min_element(load.begin(), load.end()) - load.begin());
      blocks_of[w].push_back(q);
      load[w] += int(of_block[q].size());
    }
    for (int w = 0; w < 4; ++w) {
      std::sort(blocks_of[w].begin(), blocks_of[w].end());
      f->wp0[w] = int(wmeta.size());
      for (int q : blocks_of[w])
        for (int i : of_block[q]) {
          wmeta.push_back(pairs[2 * i + 1]);
          const size_t o = pool_acc.size();
          pool_acc.resize(o + 256);
          for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 2; ++e)
                pool_acc[o + h * 128 + lane * 2 + e] = pool[size_t(pairs[2 * i]) + size_t(4 * (2 * h + e) + (lane >> 4)) * TB + (lane & 15)];
        }
      while ((int(wmeta.size()) - f->wp0[w]) % PAIR_RING) {  // the walk goes PAIR_RING pairs at a time: no-ops (weight 0) on zero pieces
        wmeta.push_back(COEF_MAX - 1);
        pool_acc.resize(pool_acc.size() + 256, 0.0);
      }
    }
    f->wp0[4] = int(wmeta.size());
    wmeta.resize(wmeta.size() + 128, COEF_MAX - 1);        // no-ops (the walk reads its metas 64 at a time, one group ahead)
    pool_acc.resize(pool_acc.size() + 2 * PAIR_RING * 256, 0.0);  // the fetches that run ahead of the walk
  }
  f->npairs = int((pairs.size() / 2 + 63) / 64 * 64);
  // no-op padding: term slot COEF_MAX - 1 is never a real term (its weight is 0), block 0 of the first table
  while (int(pairs.size() / 2) < f->npairs + 64) { pairs.push_back(0); pairs.push_back(COEF_MAX - 1); }

  ROMHC_PHASE("block sides, vmap, parameter-independent part of");
  // ---- block sides, vmap, parameter-independent part of the reduced rhs --------------------------------------------
  std::vector<int> vmap(std::max(f->nGp, 1), -1);
  for (int e = 0; e < E; ++e) {
    const Edge& ed = edges[e];
    for (int t = 0; t < n1; ++t) {
      int r, c;  // 1-based inner vertex coordinates
      if (ed.hv == 0) { r = ed.p * N; c = ed.q * N + t + 1; }
      else { r = ed.p * N + t + 1; c = ed.q * N; }
      vmap[npos[e] + t] = (r - 1) * f->nc + (c - 1);
    }
  }
  for (int x = 0; x < ncross; ++x) {
    int r = crosses[x].first * N, c = crosses[x].second * N;
    vmap[f->xb0 + x] = (r - 1) * f->nc + (c - 1);
  }
  std::vector<double> g_red(std::max(f->nGa, 1), 0.0);
  for (int e : order) {
    const Comp& ce = comps[comp_of[e]];
    for (int k = 0; k < ce.r; ++k) g_red[zpos[e] + k] = double(ce.gt[k]);
  }
  for (int x = 0; x < ncross; ++x) g_red[xred[x]] = h2;

  ROMHC_PHASE("expansion tables of the active edges, back subst");
  // ---- expansion tables of the active edges, back substitution tables of the closed-form ones ------------------------
  const size_t tsz = size_t(n1p) * n1p;
  // table variants of a compressed-edge type: 0 = active edge (P, p0), 1 = closed-form edge (K^-1 W, K^-1 g)
  std::map<std::pair<int, int>, int> ptab_of, p0_of;
  std::vector<std::pair<int, int>> ptab_list;
  auto variant = [&](int c, int v) {
    if (!ptab_of.count({c, v})) {
      ptab_of[{c, v}] = int(ptab_list.size());
      ptab_list.push_back({c, v});
      p0_of[{c, v}] = push_vec(v == 0 ? comps[c].p0 : comps[c].wK, n1p);
    }
    return ptab_of[{c, v}];
  };
  std::vector<ExpEdge> exps;
  for (int e : order) {
    const int c = comp_of[e], pt = variant(c, 0);
    exps.push_back(ExpEdge{zpos[e], (rk[e] + BK - 1) / BK, npos[e], pt, p0_of[{c, 0}], f->spos0 + e});
    if (cpos[e] >= 0) {
      CoefGroup cg;
      memset(&cg, 0, sizeof(cg));
      cg.kind = 0; cg.cpos = cpos[e]; cg.r = rk[e]; cg.w = rp[c]; cg.b0 = edges[e].b0; cg.b1 = edges[e].b1; cg.zpos = zpos[e];
      groups.push_back(cg);
    }
  }
  for (int e : pre_list)
    if (cpos[e] >= 0) {
      const int c = comp_of[e], pt = variant(c, 1);
      exps.push_back(ExpEdge{cpos[e], (comps[c].r + BK - 1) / BK, npos[e], pt, p0_of[{c, 1}], f->spos0 + e});
    }
  f->nexp = int(exps.size());
  std::vector<int> item_group, item_k;
  for (size_t g = 0; g < groups.size(); ++g)
    for (int k = 0; k < groups[g].w; ++k) { item_group.push_back(int(g)); item_k.push_back(k); }
  f->ncoef = int(item_group.size());
  // k_coef spreads the dot products of the closed-form blocks over its workgroup: tasks (group, entry k, term t) ordered by
  // (group, term, k) -- neighbouring threads read neighbouring entries of a matrix row -- as flat records (rom_fem_dev.h)
  std::vector<int> item_cf(item_group.size(), -1), ctask;
  {
    std::vector<int> first_cf(groups.size(), -1);
    int ncf = 0;
    for (size_t it = 0; it < item_group.size(); ++it) {
      const CoefGroup& cg = groups[item_group[it]];
      if (cg.kind == 1 && item_k[it] < cg.r) {
        if (first_cf[item_group[it]] < 0) first_cf[item_group[it]] = ncf;
        item_cf[it] = ncf++;
      }
    }
    for (size_t g = 0; g < groups.size(); ++g)
      if (groups[g].kind == 1)
        for (int t = 0; t < groups[g].nterm; ++t)
          for (int k = 0; k < groups[g].r; ++k) {
            const CoefTerm& ct = groups[g].t[t];
            const int rec[8] = {ct.moff + k, ct.src, ct.len, groups[g].r, ct.voff >= 0 ? ct.voff + k : -1, ct.u0, ct.u1, (first_cf[g] + k) * 8 + t};
            ctask.insert(ctask.end(), rec, rec + 8);
          }
    f->ncf = ncf;
    f->nctask = int(ctask.size() / 8);
  }
  {
    // single-tile path: the coefficient blocks of the closed-form edges as one dense product (k_solve1)
    std::vector<DenseGroup> dgroups;
    std::vector<int> dweight, ditem_group, ditem_k;
    std::vector<std::pair<int, int>> dsrc;  // (group index in `groups`, first item)
    for (size_t g = 0; g < groups.size(); ++g)
      if (groups[g].kind == 1) {
        DenseGroup dg;
        memset(&dg, 0, sizeof(dg));
        dg.cpos = groups[g].cpos; dg.r = groups[g].r; dg.b0 = groups[g].b0; dg.b1 = groups[g].b1;
        dsrc.push_back({int(g), int(ditem_group.size())});
        for (int k = 0; k < groups[g].r; ++k) { ditem_group.push_back(int(dgroups.size())); ditem_k.push_back(k); }
        dgroups.push_back(dg);
      }
    const int ndi = int(ditem_group.size());
    std::vector<double> dmat(size_t(TB) * std::max(ndi, 1) + 128, 0.0);  // (+ 1 KB: k_solve1 copies it to LDS in whole kilobytes)
    dweight.assign(dgroups.size() * TB, -2);
    bool ok = f->fused1 && int(dgroups.size()) <= DENSE_GROUPS_MAX && rhs_terms.size() <= 64 && nrb * ncb <= 64 && ndi <= 64;  // (k_solve1: a lane per rhs term / per block coefficient)
    for (size_t dgi = 0; dgi < dgroups.size() && ok; ++dgi) {
      const CoefGroup& cg = groups[dsrc[dgi].first];
      for (int t = 0; t < cg.nterm && ok; ++t) {
        const CoefTerm& ct = cg.t[t];
        for (int j = 0; j < ct.len; ++j) {
          if (ct.src + j >= TB) { ok = false; break; }
          dweight[dgi * TB + ct.src + j] = ct.blk >= 0 ? ct.blk : -1;
          for (int k = 0; k < cg.r; ++k) dmat[size_t(ct.src + j) * ndi + dsrc[dgi].second + k] = cm[ct.moff + size_t(j) * cg.r + k];
        }
        if (ct.voff >= 0) {
          DenseGroup& dg = dgroups[dgi];
          if (dg.nv >= 4) { ok = false; break; }
          dg.voff[dg.nv] = ct.voff; dg.vblk[dg.nv] = ct.blk; dg.vu0[dg.nv] = ct.u0; dg.vu1[dg.nv] = ct.u1;
          ++dg.nv;
        }
      }
    }
    if (!ok) f->fused1 = false;
    f->ndg = f->fused1 ? int(dgroups.size()) : 0;
    f->ndi = f->fused1 ? ndi : 0;
    ROM_TRY(upload(ctx->stream, &f->d_dgroups, dgroups));
    ROM_TRY(upload(ctx->stream, &f->d_dweight, dweight));
    // k_solve1 reads one FLAT record per item and lane (every level of indirection is a memory round trip a lone wave waits
    // out): dense item = {group, position, nv, b0, b1, voff[4] + k, vblk[4], vu0[4], vu1[4], pad} (24 ints);
    // coefficient item = {position | code << 28, source, b0, b1}, code 0 = the dense product's, 1 = 1 / (a_b0 + a_b1),
    // 2 = copy of the solution, 3 = zero
    std::vector<int> s1_items(size_t(std::max(ndi, 1)) * 24, 0), s1_citems(size_t(std::max(f->ncoef, 1)) * 4, 0);
    if (f->fused1) {
      for (int it = 0; it < ndi; ++it) {
        const DenseGroup& dg = dgroups[ditem_group[it]];
        int* r = &s1_items[size_t(it) * 24];
        r[0] = ditem_group[it]; r[1] = dg.cpos + ditem_k[it]; r[2] = dg.nv; r[3] = dg.b0; r[4] = dg.b1;
        for (int v = 0; v < 4; ++v) {
          r[5 + v] = v < dg.nv ? dg.voff[v] + ditem_k[it] : 0;
          r[9 + v] = v < dg.nv ? dg.vblk[v] : 0; r[13 + v] = v < dg.nv ? dg.vu0[v] : 0; r[17 + v] = v < dg.nv ? dg.vu1[v] : 0;
        }
      }
      for (int it = 0; it < f->ncoef; ++it) {
        const CoefGroup& cg = groups[item_group[it]];
        const int k = item_k[it];
        const int code = cg.kind == 1 && k < cg.r ? 0 : k == cg.r ? 1 : k < cg.r ? 2 : 3;
        int* r = &s1_citems[size_t(it) * 4];
        r[0] = (cg.cpos + k) | code << 28; r[1] = cg.zpos + k; r[2] = cg.b0; r[3] = cg.b1;
      }
    }
    ROM_TRY(upload(ctx->stream, &f->d_s1_items, s1_items));
    ROM_TRY(upload(ctx->stream, &f->d_s1_citems, s1_citems));
    ROM_TRY(upload(ctx->stream, &f->d_ditem_group, ditem_group));
    ROM_TRY(upload(ctx->stream, &f->d_ditem_k, ditem_k));
    ROM_TRY(upload(ctx->stream, &f->d_dmat, dmat));
  }
  {
    std::vector<double> Ptab(std::max<size_t>(ptab_list.size() * tsz, 1), 0.0);
    for (size_t t = 0; t < ptab_list.size(); ++t)
      put_table(Ptab, t, n1p, ptab_list[t].second == 0 ? comps[ptab_list[t].first].P : comps[ptab_list[t].first].KiW, false);
    ROM_TRY(upload(ctx->stream, &f->d_P, Ptab));
    std::vector<double> Bt(std::max<size_t>(size_t(nbt) * tsz, 1), 0.0);
    for (auto& kv : bt_of_id) put_table(Bt, kv.second, n1p, TK(kv.first), true);  // (T K^-1)^T: row = node of e
    for (auto& kv : bt_extra) put_table(Bt, kv.first, n1p, kv.second, false);
    ROM_TRY(upload(ctx->stream, &f->d_Bt, Bt));
  }

  ROMHC_PHASE("device tables of the harmonic extension");
  // ---- device tables of the harmonic extension ---------------------------------------------------------------
  std::vector<double> Qp(size_t(n1p) * n1p, 0.0);
  for (int j = 0; j < n1; ++j)
    for (int m = 0; m < n1; ++m) Qp[size_t(j) * n1p + m] = double(Q(j, m));
  double* d_rho = nullptr;
  ROM_TRY(upload(ctx->stream, &f->d_Qp, Qp));
  ROM_TRY(upload(ctx->stream, &d_rho, rho_d));
  const size_t hrows = size_t(n1) * n1;
  ROM_HIP(hipMalloc(&f->d_A0, std::max<size_t>(hrows * n1p, 1) * sizeof(double)));
  if (hrows > 0) {
    size_t total = hrows * n1p;
    k_build_A0<<<unsigned((total + 255) / 256), 256, 0, ctx->stream>>>(f->d_A0, f->d_Qp, d_rho, n1, n1p, N);
    ROM_HIP(hipGetLastError());
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  hipFree(d_rho);
  {
    ROM_TRY(upload(ctx->stream, &f->d_kmax, kmax));

    // Representation of every block side in the extension.  A compressed edge enters through its reduced
    // unknowns when that is cheaper than the distance-truncated sine modes: table G_c = H_0 [P_c, p0_c]
    // = A0 (Q [P_c, p0_c]), one (n1*n1) x rp_c table per compressed-edge type.
    // one table per (compressed-edge type, variant) that a block side actually uses
    std::map<std::pair<int, int>, long long> goff;
    long long gtotal = 0;
    for (int e = 0; e < E; ++e)
      if (cpos[e] >= 0 && !goff.count({comp_of[e], int(is_pre[e])})) {
        goff[{comp_of[e], int(is_pre[e])}] = gtotal;
        gtotal += (long long)hrows * rp[comp_of[e]];
      }
    ROM_CHECK(gtotal < (1ll << 31), "rom_fem_create: extension tables too large");
    ROM_HIP(hipMalloc(&f->d_G, std::max<size_t>(size_t(gtotal), 1) * sizeof(double)));
    // the sine coefficients Q^T [P_c, p0_c] of every table: long-double products, one host thread per table
    std::vector<std::pair<std::pair<int, int>, long long>> gkeys(goff.begin(), goff.end());
    std::vector<std::vector<double>> Bhs(gkeys.size());
    parallel_for(gkeys.size(), [&](size_t gi) {
      const int c = gkeys[gi].first.first;
      const Comp& cp = comps[c];
      const Mat& Pm = gkeys[gi].first.second == 0 ? cp.P : cp.KiW;
      const std::vector<ld>& pv = gkeys[gi].first.second == 0 ? cp.p0 : cp.wK;
      Mat Bm = hostla::mul_tn(Pm, Q);  // r x n1
      std::vector<double>& Bh = Bhs[gi];
      Bh.assign(size_t(rp[c]) * n1p, 0.0);
      for (int k = 0; k < cp.r; ++k)
        for (int m = 0; m < n1; ++m) Bh[size_t(k) * n1p + m] = double(Bm(k, m));
      for (int m = 0; m < n1; ++m) {
        ld sacc = 0;
        for (int t = 0; t < n1; ++t) sacc += pv[t] * Q(t, m);
        Bh[size_t(cp.r) * n1p + m] = double(sacc);
      }
    });
    for (size_t gi = 0; gi < gkeys.size(); ++gi) {
      const auto& kv = gkeys[gi];
      const int c = kv.first.first;
      const std::vector<double>& Bh = Bhs[gi];
      double* d_B = nullptr;
      ROM_TRY(upload(ctx->stream, &d_B, Bh));
      ROM_TRY(rom_launch_gemm_nt(ctx, int64_t(hrows), rp[c], n1p, 1.0, f->d_A0, n1p, d_B, n1p, 0.0, f->d_G + kv.second,
                                 rp[c], "setup_gemm_G"));
      ROM_HIP(hipStreamSynchronize(ctx->stream));
      hipFree(d_B);
    }
    f->sides.resize(nrb * ncb);
    std::map<std::tuple<int, int, int>, long long> gsoff;  // (type, variant, orientation) -> offset in Gs
    long long gstotal = 0;
    std::vector<char> need_tr(E, 0);
    double fl = 0;
    const int npj = (n1 + 15) / 16, npi = (n1 + 3) / 4;
    for (int b = 0; b < nrb * ncb; ++b)
      for (int sdx = 0; sdx < 4; ++sdx) {
        ExtSide& es = f->sides[b].s[sdx];
        memset(&es, 0, sizeof(es));
        const int e = bside[b][sdx];
        if (e < 0) continue;
        const int c = comp_of[e];
        if (cpos[e] >= 0) {
          es = ExtSide{2, cpos[e], rp[c] / BK, comps[c].r, int(goff[{c, int(is_pre[e])}]), 0, edges[e].b0, edges[e].b1};
          // segment-major copy for k_extend128: rows ordered for this side's orientation (one per table and orientation)
          {
            const int orient = sdx >= 2 ? 1 : 0;
            const auto key = std::make_tuple(c, int(is_pre[e]), orient);
            if (!gsoff.count(key)) {
              gsoff[key] = gstotal;
              gstotal += (long long)((comps[c].r + 1 + 7) / 8) * (long long)hrows * 8;
            }
            es.gseg = int(gsoff[key]);
          }
          fl += 2.0 * double(n1) * n1 * (comps[c].r + 1);  // algorithmic: no padding of K or of the vertex tiles
        } else {
          es.mode = 1;
          es.off = npos[e];
          need_tr[e] = 1;
          for (int pi = 0; pi < npi; ++pi)
            for (int pj = 0; pj < npj; ++pj) {
              int i0 = 4 * pi + 1, j0 = 16 * pj + 1, i1 = std::min(i0 + 3, n1), j1 = std::min(j0 + 15, n1);
              int dist[4] = {i0, N - i1, j0, N - j1};
              fl += 2.0 * 64 * kmax[dist[sdx]];
            }
        }
      }
    std::vector<int> lr_blocks, gen_blocks;
    f->lr_nch = 0;
    for (int b = 0; b < nrb * ncb; ++b) {
      int nlr = 0, nother = 0, nch = 0;
      for (int sdx = 0; sdx < 4; ++sdx) {
        const ExtSide& es = f->sides[b].s[sdx];
        if (es.mode == 2) { ++nlr; nch += es.nch; }
        else if (es.mode != 0) ++nother;
      }
      if (nlr > 0 && nother == 0 && !ab_env("ROMHC_NO_EXT_LR")) {
        lr_blocks.push_back(b);
        f->lr_nch = std::max(f->lr_nch, nch);
      } else {
        gen_blocks.push_back(b);
      }
    }
    f->n_lr_blocks = int(lr_blocks.size());
    f->n_gen_blocks = int(gen_blocks.size());
    ROM_TRY(upload(ctx->stream, &f->d_lr_blocks, lr_blocks));
    f->lr_blocks_host = lr_blocks;
    ROM_TRY(upload(ctx->stream, &f->d_gen_blocks, gen_blocks));
    std::vector<int> eposv;
    for (int e = 0; e < E; ++e)
      if (need_tr[e]) eposv.push_back(npos[e]);
    f->n_edges = int(eposv.size());
    if (eposv.empty()) eposv.push_back(0);
    ROM_TRY(upload(ctx->stream, &f->d_epos, eposv));
    // flops of the extension, per system (for the work accounting)
    f->ext_flops = fl + 2.0 * f->n_edges * double(n1p) * n1p;
    // the segment-major copies k_extend128 reads
    ROM_CHECK(gstotal < (1ll << 31), "rom_fem_create: extension tables too large");
    ROM_HIP(hipMalloc(&f->d_Gs, std::max<size_t>(size_t(gstotal), 1) * sizeof(double)));
    f->gs_bytes = size_t(gstotal) * sizeof(double);
    for (auto& kv : gsoff) {
      const int c = std::get<0>(kv.first), variant = std::get<1>(kv.first), orient = std::get<2>(kv.first);
      const int nseg = (comps[c].r + 1 + 7) / 8;
      const size_t total = size_t(nseg) * hrows * 8;
      if (total == 0) continue;
      k_repack_table<<<unsigned((total + 255) / 256), 256, 0, ctx->stream>>>(f->d_G + goff[{c, variant}], rp[c], nseg, n1, orient,
                                                                              f->d_Gs + kv.second);
      ROM_HIP(hipGetLastError());
    }
    ROM_HIP(hipStreamSynchronize(ctx->stream));
  }
  {
    std::vector<double> Wz(Wd);  // + a page of zeros: where k_extend128 points the lanes that have nothing to load
    Wz.resize(Wd.size() + EXT_ZERO_PAGE, 0.0);
    ROM_TRY(upload(ctx->stream, &f->d_W, Wz));
  }
  ROM_TRY(upload(ctx->stream, &f->d_g, g_red));
  ROM_TRY(upload(ctx->stream, &f->d_vec, vecs));
  ROM_TRY(upload(ctx->stream, &f->d_pool, pool));
  ROM_TRY(upload(ctx->stream, &f->d_terms, terms));
  ROM_TRY(upload(ctx->stream, &f->d_pairs, pairs));
  ROM_TRY(upload(ctx->stream, &f->d_alist, alist));
  ROM_TRY(upload(ctx->stream, &f->d_aoff, aoff));
  ROM_TRY(upload(ctx->stream, &f->d_pool_acc, pool_acc));
  ROM_TRY(upload(ctx->stream, &f->d_wmeta, wmeta));
  if (f->fused1)  // (the attribute belongs to the kernel as loaded on this device; setting it again is harmless)
    ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve1), hipFuncAttributeMaxDynamicSharedMemorySize, S1_LDS_BYTES));
  f->nrhs = int(rhs_terms.size());
  ROM_TRY(upload(ctx->stream, &f->d_rhs, rhs_terms));
  ROM_TRY(upload(ctx->stream, &f->d_pre, pre_edges));
  ROM_TRY(upload(ctx->stream, &f->d_exp, exps));
  ROM_TRY(upload(ctx->stream, &f->d_groups, groups));
  ROM_TRY(upload(ctx->stream, &f->d_cm, cm));
  ROM_TRY(upload(ctx->stream, &f->d_item_group, item_group));
  ROM_TRY(upload(ctx->stream, &f->d_item_k, item_k));
  ROM_TRY(upload(ctx->stream, &f->d_item_cf, item_cf));
  ROM_TRY(upload(ctx->stream, &f->d_ctask, ctask));
  ROM_TRY(upload(ctx->stream, &f->d_xred, xred));
  {
    std::vector<int> scb;  // (b0, b1) per scalar: an edge's two blocks, or (block, -1)
    for (int e = 0; e < E; ++e) { scb.push_back(edges[e].b0); scb.push_back(edges[e].b1); }
    for (int b = 0; b < nrb * ncb; ++b) { scb.push_back(b); scb.push_back(-1); }
    ROM_TRY(upload(ctx->stream, &f->d_scb, scb));
  }
  ROM_TRY(upload(ctx->stream, &f->d_desc, f->desc));
  ROM_TRY(upload(ctx->stream, &f->d_kptr, f->kptr));
  ROM_TRY(upload(ctx->stream, &f->d_kpair, f->kpair));
  ROM_TRY(upload(ctx->stream, &f->d_colptr, f->colptr));
  ROM_TRY(upload(ctx->stream, &f->d_colrow, f->colrow));
  ROM_TRY(upload(ctx->stream, &f->d_colti, f->colti));
  ROM_TRY(upload(ctx->stream, &f->d_sides, f->sides));
  ROM_TRY(upload(ctx->stream, &f->d_vmap, vmap));
  {
    std::vector<char> expanded(E, 0);
    for (int e : order) expanded[e] = 1;
    for (int e : pre_list)
      if (cpos[e] >= 0) expanded[e] = 1;
    std::vector<int> scat;
    for (int e = 0; e < E; ++e)
      if (!expanded[e])
        for (int t = 0; t < n1; ++t) scat.push_back(npos[e] + t);
    for (int x = 0; x < ncross; ++x) scat.push_back(f->xb0 + x);
    f->nscat = int(scat.size());
    ROM_TRY(upload(ctx->stream, &f->d_scat, scat));
  }

  f->sw_no_fused = getenv("ROMHC_NO_FUSED") != nullptr;
  // k_extend128's workgroup order: system group fastest once the extension tables outgrow what the caches keep next to
  // the store stream (measured: C5, 4 x 4 / N = 256, tables 100+ MB: fetch 21.7 -> 10.6 GB per launch of 2048 systems, kernel -2...-6 %;
  // C4, 3 x 3 / N = 171: no gain; C2, 16 MB of tables: 5 % slower) -- ROMHC_X128_SYS_FAST = 0 / 1 / 2 overrides in the A/B build
  f->sw_no_ext128 = getenv("ROMHC_NO_EXT128") != nullptr;
#ifdef ROMHC_AB
  // The A/B build (libromhc_ab.so, `make ab`; only tests/ab_variants.py loads it) can FORCE choices that the product makes by
  // geometry -- tilings, workgroup orders, one system per workgroup, tiles assembled in registers -- to check that the forms
  // the product uses on different geometries give the same bits on one.  The product build does not read these.
  f->sw_x128_sys_fast = ab_env("ROMHC_X128_SYS_FAST") ? atoi(ab_env("ROMHC_X128_SYS_FAST")) : -1;
  f->sw_no_fold = ab_env("ROMHC_NO_FOLD_EXPAND") != nullptr;
  f->sw_coef_global = ab_env("ROMHC_COEF_GLOBAL") != nullptr;
  f->sw_no_tile_pairs = ab_env("ROMHC_NO_TILE_PAIRS") != nullptr;
  f->sw_no_tile_stream = ab_env("ROMHC_NO_TILE_STREAM") != nullptr;
  f->sw_ext_flat = ab_env("ROMHC_EXT_FLAT") ? (atoi(ab_env("ROMHC_EXT_FLAT")) != 0 ? 1 : 0) : -1;
#endif
  ROMHC_PHASE("end");
  if (getenv("ROMHC_VERBOSE")) {
    fprintf(stderr, "romhc: %dx%d blocks N=%d: %d edges (%d closed-form, %d of them compressed), reduced size %d -> %d tiles, "
                    "%d slots, %zu terms, kavg %.1f\n", nrb, ncb, N, E, int(pre_list.size()), int(pre_list.size()) - f->npre, nred, T,
            f->nslots, terms.size(), kavg);
    for (size_t c = 0; c < comps.size(); ++c) {
      int cnt = 0;
      for (int e = 0; e < E; ++e) cnt += comp_of[e] == int(c);
      fprintf(stderr, "romhc:   edge type %zu: rank %d (padded %d), %d edges, extension %s\n", c, comps[c].r, rp[c], cnt,
              use_lr[c] ? "from the reduced unknowns" : "sine modes");
    }
    fprintf(stderr, "romhc:   blocks extended by the 128-tile kernel: %d, general kernel: %d\n", f->n_lr_blocks, f->n_gen_blocks);
    {  // how sparse the term tables are: non-zero 16 x 16 blocks, bounding rectangles
      size_t nzb = 0, rect = 0, rows16 = 0;
      for (size_t t = 0; t < terms.size(); ++t) {
        const double* tb = pool.data() + size_t(terms[t].tab) * 4096;
        rect += size_t(terms[t].r_hi - terms[t].r_lo) * size_t(terms[t].c_hi - terms[t].c_lo);
        for (int ib = 0; ib < 4; ++ib)
          for (int jb = 0; jb < 4; ++jb) {
            bool nz = false;
            for (int i = 0; i < 16 && !nz; ++i)
              for (int j = 0; j < 16; ++j)
                if (tb[(16 * ib + i) * 64 + 16 * jb + j] != 0.0) { nz = true; break; }
            nzb += nz;
          }
        for (int r = 0; r < 64; ++r)
          for (int sg = 0; sg < 4; ++sg) {
            bool nz = false;
            for (int j = 0; j < 16; ++j) nz = nz || tb[r * 64 + 16 * sg + j] != 0.0;
            rows16 += nz;
          }
      }
      fprintf(stderr, "romhc:   term tables: %zu tables, %zu non-zero 16x16 blocks of %zu (%.2f), bounding rectangles cover %.2f, non-zero 1x16 strips %.2f\n",
              terms.size(), nzb, terms.size() * 16, double(nzb) / (terms.size() * 16), double(rect) / (terms.size() * 4096.0),
              double(rows16) / (terms.size() * 256.0));
    }
  }

  ROMHC_PHASE("work accounting of this algorithm, per snapshot ");
  // ---- work accounting of this algorithm, per snapshot solve ------------------------------------------------
  double exp_flops = 0;
  for (int e : order) exp_flops += 2.0 * n1p * double((rk[e] + BK - 1) / BK * BK);
  const double back_flops = 2.0 * 4096.0 * (f->nslots + T);
  f->flops_solve = flops + f->ext_flops + back_flops + pre_flops + exp_flops;
  // HBM bytes: factor tiles written once + read once by the back substitution, inverse tiles w+r,
  // the snapshot row written once, the coefficients read.
  f->bytes_solve = 8.0 * (2.0 * 4096.0 * f->nslots + 2.0 * 4096.0 * T + double(f->dim) + nrb * ncb);
  *out = f;
  return ROM_OK;
}

