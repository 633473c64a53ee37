// Entry points of the sweep: rom_solve_batch and its two stages, workspace management, kernel sequencing
// (replaces generate_solutions, src/lib/SolutionsManagers.py:64-68, of the reference).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "rom_fem_dev.h"

extern "C" int rom_fem_dims(rom_fem* f, int* nr, int* nc, int64_t* dim, int* n_interface, int* n_tiles) {
  ROM_CHECK(f, "null fem");
  if (nr) *nr = f->nr;
  if (nc) *nc = f->nc;
  if (dim) *dim = f->dim;
  if (n_interface) *n_interface = f->nG;
  if (n_tiles) *n_tiles = f->nslots;
  return ROM_OK;
}

extern "C" int rom_fem_load_vector_host(rom_fem* f, double* B) {
  ROM_CHECK(f && B, "bad arguments");
  // (:177-185) every inner vertex collects area/6 + area/3 + area/3 + area/6 in this order
  double area = (1.0 / f->N) * (1.0 / f->N);
  double v = 0.0;
  v += area / 6;
  v += area / 3;
  v += area / 3;
  v += area / 6;
  for (int64_t i = 0; i < f->dim; ++i) B[i] = v;
  return ROM_OK;
}

extern "C" int rom_solve_work(rom_fem* f, double* flops_own, double* bytes_own, double* flops_banded,
                              double* bytes_banded) {
  ROM_CHECK(f, "null fem");
  if (flops_own) *flops_own = f->flops_solve;
  if (bytes_own) *bytes_own = f->bytes_solve;
  double b = std::min(f->nr, f->nc), dim = double(f->dim);
  double nnzL = dim * (b + 1) - b * (b + 1) / 2;
  if (bytes_banded) *bytes_banded = 8.0 * (3 * nnzL + 5 * dim);
  if (flops_banded) *flops_banded = dim * b * b + 4 * dim * b;
  return ROM_OK;
}

extern "C" int rom_assemble_batch(rom_fem* f, rom_buf* a, int M, rom_buf* diag, rom_buf* east, rom_buf* north) {
  ROM_CHECK(f && a && diag && east && north, "rom_assemble_batch: null argument");
  ROM_CHECK(M >= 0, "rom_assemble_batch: negative M");
  const int kblk = f->nrb * f->ncb;
  ROM_CHECK(a->n >= size_t(M) * kblk, "rom_assemble_batch: `a` holds %zu doubles, need %zu", a->n, size_t(M) * kblk);
  ROM_CHECK(diag->n >= size_t(M) * f->dim && east->n >= size_t(M) * f->nr * (f->nc - 1) &&
                north->n >= size_t(M) * (f->nr - 1) * f->nc,
            "rom_assemble_batch: output buffers too small");
  if (M == 0) return ROM_OK;
  FemDev d = make_dev(f);
  dim3 grid(unsigned((f->dim + 255) / 256), M);
  {
    ROM_PROF(f->ctx, "assemble_stencil", 7.0 * f->dim * M, 24.0 * f->dim * M);
    k_assemble_stencil<<<grid, 256, 0, f->ctx->stream>>>(d, a->p, M, diag->p, east->p, north->p);
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}


static int ensure_workspace(rom_fem* f, int Mc) {
  if (f->ws_M >= Mc) return ROM_OK;
  ROM_HIP(hipStreamSynchronize(f->ctx->stream));
  if (f->d_L) hipFree(f->d_L);
  if (f->d_invL) hipFree(f->d_invL);
  if (f->d_y) hipFree(f->d_y);
  if (f->d_yhat) hipFree(f->d_yhat);
  if (f->d_dots) hipFree(f->d_dots);
  f->d_L = f->d_invL = f->d_y = f->d_yhat = f->d_dots = nullptr;
  f->ws_M = 0;
  const size_t nL = std::max<size_t>(size_t(Mc) * f->nslots * 4096, 1) * sizeof(double);
  const size_t nI = std::max<size_t>(size_t(Mc) * f->T * 4096, 1) * sizeof(double);
  const size_t nY = std::max<size_t>(size_t(Mc) * f->nGp, 1) * sizeof(double);
  ROM_HIP(hipMalloc(&f->d_L, nL));
  ROM_HIP(hipMalloc(&f->d_invL, nI));
  ROM_HIP(hipMalloc(&f->d_y, nY));
  ROM_HIP(hipMalloc(&f->d_yhat, nY));
  // (k_coef keeps its dot products in LDS up to 156 KB; a geometry with more closed-form entries than that gets them in HBM)
  if (f->ncf > 0 && (f->sw_coef_global || size_t(std::max(f->nGa, 1)) * 8 + size_t(f->ncf) * 64 > 156 * 1024)) ROM_HIP(hipMalloc(&f->d_dots, size_t(Mc) * f->ncf * 8 * sizeof(double)));
  // Zeroed once: padding slots of d_y are read (against zero table entries) before anything writes them, and a
  // fresh hipMalloc block may hold another process's NaN bits.  The fill goes on the context's compute stream:
  // that stream and the sub-batch streams forked from it are hipStreamNonBlocking, so a null-stream hipMemset
  // is NOT ordered before the kernels that read the buffer.
  rom_ctx* ctx = f->ctx;
  ROM_HIP(hipMemsetAsync(f->d_y, 0, nY, ctx->stream));
  ROM_HIP(hipMemsetAsync(f->d_yhat, 0, nY, ctx->stream));
  if (getenv("ROMHC_POISON_WS")) {  // debugging aid: NaN patterns in everything a kernel must write before it reads
    ROM_HIP(hipMemsetAsync(f->d_L, 0xFF, nL, ctx->stream));
    ROM_HIP(hipMemsetAsync(f->d_invL, 0xFF, nI, ctx->stream));
    ROM_HIP(hipMemsetAsync(f->d_yhat, 0xFF, nY, ctx->stream));
  }
  f->ws_M = Mc;
  return ROM_OK;
}

// enqueue every kernel of one sub-batch (Mc systems, workspace pointers already offset) on `st`
// `stages`: 1 = reduced solve (interface vector: reduced unknowns, cross points, coefficient blocks),
//           2 = expansion of the interface vector into snapshot rows, 3 = both
static int enqueue_solve(rom_fem* f, const FemDev& d, const double* am, int Mc, double* U, long long row,
                         hipStream_t st, size_t lds_back, int stages) {
  rom_ctx* ctx = f->ctx;
  const int kblk = f->nrb * f->ncb;
  static const bool detail = getenv("ROMHC_PROF_DETAIL") != nullptr;  // per-column kernel names
  char nm[4][48];
  const bool fused1 = f->fused1 && !f->sw_no_fused;  // the whole reduced solve in one wave-per-system kernel
  if (f->nGp > 0 && fused1 && (stages & 1)) {
    ROM_PROF(ctx, "solve1", Mc * (262144 / 3.0 + 3 * 4096.0), Mc * 8.0 * 4096 * 3);
    k_solve1<<<(Mc + 3) / 4, 512, S1_LDS_BYTES, st>>>(d, am, Mc);  // four systems per workgroup, two waves each
  }
  if (f->nGp > 0 && !fused1 && (stages & 1)) {
    {
      ROM_PROF(ctx, "rhs", 0, 8.0 * Mc * f->nGa);
      k_rhs<<<Mc, 256, 0, st>>>(d, am);
    }
    for (int j = 0; j < f->T; ++j) {
      {
        int slot = f->diag_slot[j];
        double nk = f->kptr[slot + 1] - f->kptr[slot];
        const char* base[4] = {"diag_update", "diag_factor", "", "factor_panel"};
        for (int q = 0; q < 4; ++q) detail ? snprintf(nm[q], 48, "%s_j%02d", base[q], j) : snprintf(nm[q], 48, "%s", base[q]);
        {
          ROM_PROF(ctx, nm[0], Mc * nk * 2.0 * 262144, Mc * 8.0 * 4096 * (1 + 2 * nk));
          // (two systems per workgroup share one pass over the term tables: -20 % at 1024 systems, nothing at 4096)
          if (f->sw_no_tile_pairs || Mc > 2048) k_diag_update<1><<<Mc, 256, 0, st>>>(d, am, slot, Mc);
          else k_diag_update<2><<<(Mc + 1) / 2, 256, 0, st>>>(d, am, slot, Mc);
        }
        {
          ROM_PROF(ctx, nm[1], Mc * (2 * 262144 / 3.0 + 4096.0), Mc * 8.0 * 4096 * 3);
          k_diag_factor<<<Mc, 64, 0, st>>>(d, slot, j);
        }
      }
      int nrows = f->colptr[j + 1] - f->colptr[j];
      if (nrows > 0) {
        double nk = 0;
        for (int e = f->colptr[j]; e < f->colptr[j + 1]; ++e) nk += f->kptr[f->colrow[e] + 1] - f->kptr[f->colrow[e]];
        ROM_PROF(ctx, nm[3], Mc * (nk + nrows) * 2.0 * 262144, Mc * 8.0 * 4096 * (2 * nk + 2 * nrows));
        k_factor_panel<1><<<nrows * Mc, 256, 0, st>>>(d, am, j, Mc);
      }
    }
    if (f->T > 0) {
      ROM_PROF(ctx, "backsolve", Mc * 2.0 * 4096 * (f->nslots + f->T), Mc * 8.0 * 4096 * (f->nslots + f->T));
      k_backsolve<<<Mc, 256, lds_back, st>>>(d);
    }
    {
      ROM_PROF(ctx, "coef", 0, 8.0 * Mc * f->ncoef);
      // the nGa reduced unknowns + 8 dot products per closed-form entry (those in HBM where they outgrow the LDS: d.gdots)
      const size_t lds_coef = d.gdots ? lds_back : lds_back + size_t(f->ncf) * 64;
      if (lds_coef > 48 * 1024 && !f->lds_optin_coef) {  // (the attribute belongs to the kernel as loaded on this device)
        ROM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_coef), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        f->lds_optin_coef = true;
      }
      k_coef<<<Mc, 1024, lds_coef, st>>>(d, am);
    }
  }
  if (!(stages & 2)) {
    ROM_HIP(hipGetLastError());
    return ROM_OK;
  }
  // the copy of the unexpanded interface values rides in an extra grid slice of k_expand when nothing later in
  // the sequence produces them (no edge recovered node by node) ...
  const bool fold_scatter = f->nexp > 0 && f->npre == 0 && f->nscat > 0;
  // ... and the whole expansion rides in the extension launch when that is one k_extend128 over all blocks
  // 128 x 128 tiles (k_extend128) when there are systems to fill them and they pad no worse than 64-vertex tiles
  // of one mesh row; their 128 vertices are one mesh row, or consecutive vertices of the block when rows pad badly
  const int t_row = f->n1 * ((f->n1 + 127) / 128), t_flat = (f->n1 * f->n1 + 127) / 128;
  const bool flat = f->sw_ext_flat >= 0 ? f->sw_ext_flat != 0 : 100 * t_flat < 97 * t_row;
  const int t128 = flat ? t_flat : t_row;
  // (k_extend128 addresses its table rows and interface-vector rows with 32-bit lane offsets)
  const bool fits32 = size_t(f->n1) * f->n1 * 64 * BK * 8 < (size_t(1) << 32) && size_t(128) * f->nGp * 8 < (size_t(1) << 32);
  const bool wide = Mc >= 128 && fits32 && 100 * 128 * t128 <= 102 * 64 * f->n1 * ((f->n1 + 63) / 64) && !f->sw_no_ext128;
  const bool fold_expand = f->nexp > 0 && f->npre == 0 && f->n_edges == 0 && f->n_gen_blocks == 0 && f->n_lr_blocks > 0 &&
                           wide && !f->sw_no_fold;
  if (f->nGp > 0) {
    if (f->nexp > 0 && !fold_expand) {
      ROM_PROF(ctx, "expand", Mc * 2.0 * f->n1p * 32.0 * f->nexp, 8.0 * Mc * f->n1p * f->nexp);
      k_expand<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->nexp + (fold_scatter ? 1 : 0)), 256, 0, st>>>(d, am, Mc, U, row);
    }
    if (f->npre > 0) {
      ROM_PROF(ctx, "back_pre", Mc * 2.0 * f->n1p * double(f->n1p) * 3.0 * f->npre, 8.0 * Mc * f->n1p * 4.0 * f->npre);
      k_back_pre<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->npre), 256, 0, st>>>(d, am, Mc);
    }
  }
  {
    const int nij = f->n1 * f->n1;
    if (nij > 0) {
      if (f->n_edges > 0) {
        ROM_PROF(ctx, "edge_transform", Mc * 2.0 * f->n_edges * double(f->n1p) * f->n1p, 16.0 * Mc * f->n_edges * f->n1p);
        k_edge_transform<<<dim3(f->n1p / 64, (Mc + 63) / 64, f->n_edges), 256, 0, st>>>(d, Mc);
      }
      const int npatch = ((f->n1 + 3) / 4) * ((f->n1 + 15) / 16);
      const double fl_ext = (f->ext_flops - 2.0 * f->n_edges * double(f->n1p) * f->n1p) * Mc / kblk;  // per block (average)
      if (f->n_gen_blocks > 0) {
        dim3 grid(npatch, (Mc + 63) / 64, f->n_gen_blocks);
        ROM_PROF(ctx, "extend", fl_ext * f->n_gen_blocks, 8.0 * Mc * double(f->n_gen_blocks) * nij);
        k_extend<<<grid, 256, 0, st>>>(d, am, Mc, U, row, d.gen_blocks, 4);
      }
      if (f->n_lr_blocks > 0) {
        ROM_PROF(ctx, "extend_lr", fl_ext * f->n_lr_blocks, 8.0 * Mc * double(f->n_lr_blocks) * nij);
        if (wide) {
          const int mt = (Mc + 127) / 128;
          const int items = (f->n1p / 64) * ((Mc + 63) / 64) * (f->nexp + 1);  // workgroups of the folded expansion
          // block descriptors travel as kernel arguments, X128_BLOCKS per launch; the folded expansion rides in the first
          for (int z0 = 0; z0 < f->n_lr_blocks; z0 += X128_BLOCKS) {
            const int nz = std::min(X128_BLOCKS, f->n_lr_blocks - z0);
            X128Args xa;
            memset(&xa, 0, sizeof(xa));
            for (int z = 0; z < nz; ++z) {
              xa.blocks[z] = f->lr_blocks_host[z0 + z];
              xa.sides[z] = f->sides[xa.blocks[z]];
            }
            const int extra = fold_expand && z0 == 0 ? (items + mt * nz - 1) / (mt * nz) : 0;
            const int order = f->sw_x128_sys_fast >= 0 ? f->sw_x128_sys_fast : (f->gs_bytes >= (size_t(64) << 20) && mt >= 16 ? 1 : 0);
            const int sys_fast = order == 1 ? mt : order == 2 ? -mt : 0;
            const int nx = sys_fast < 0 ? (t128 + extra + 7) / 8 * 8 : t128 + extra;
            dim3 grid(sys_fast ? nx * mt : nx, sys_fast ? 1 : mt, nz);
            if (flat) k_extend128<true><<<grid, 512, 0, st>>>(d, xa, am, Mc, U, row, extra, sys_fast);
            else k_extend128<false><<<grid, 512, 0, st>>>(d, xa, am, Mc, U, row, extra, sys_fast);
          }
        } else {
          dim3 grid(f->n1 * ((f->n1 + 63) / 64), (Mc + 63) / 64, f->n_lr_blocks);
          k_extend<<<grid, 256, 0, st>>>(d, am, Mc, U, row, d.lr_blocks, 6);
        }
      }
    }
    if (f->nscat > 0 && !(f->nexp > 0 && f->npre == 0)) {
      ROM_PROF(ctx, "scatter_interface", 0, 16.0 * Mc * f->nscat);
      k_scatter_interface<<<dim3((f->nscat + 255) / 256, Mc), 256, 0, st>>>(d, Mc, U, row);
    }
  }
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

// enqueue the sweep; `check` = also wait for it and report a non-positive pivot.  `Y` (optional): the caller's
// interface vectors (rows y_row0 .. of stride nGp) instead of the internal workspace; `stages` as in enqueue_solve.
static int solve_batch_impl(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0, bool check, rom_buf* Y = nullptr,
                            int64_t y_row0 = 0, int stages = 3) {
  ROM_CHECK(f && a && (U || !(stages & 2)), "rom_solve_batch: null argument");
  ROM_CHECK(M >= 0 && row0 >= 0 && y_row0 >= 0, "rom_solve_batch: negative M or row offset");
  const int kblk = f->nrb * f->ncb;
  ROM_CHECK(a->n >= size_t(M) * kblk, "rom_solve_batch: `a` holds %zu doubles, need %zu", a->n, size_t(M) * kblk);
  if (stages & 2)
    ROM_CHECK(U->n >= size_t(row0 + M) * f->dim, "rom_solve_batch: U holds %zu doubles, need %zu", U->n,
              size_t(row0 + M) * f->dim);
  if (Y)
    ROM_CHECK(Y->n >= size_t(y_row0 + M) * f->nGp, "rom_solve_batch: the interface-vector buffer holds %zu doubles, need %zu",
              Y->n, size_t(y_row0 + M) * f->nGp);
  if (M == 0) return ROM_OK;
  rom_ctx* ctx = f->ctx;
  ROM_HIP(hipSetDevice(ctx->device));
  // chunk the sweep so that the factor workspace respects the budget
  size_t per_sys = (size_t(f->nslots) * 4096 + size_t(f->T) * 4096 + 2 * size_t(f->nGp)) * sizeof(double);
  int Mc_max = int(std::max<size_t>(1, std::min<size_t>(size_t(M), ctx->ws_limit / std::max<size_t>(per_sys, 1))));
  if (f->ws_M > 0 && f->ws_M < Mc_max && f->ws_M >= 256) Mc_max = f->ws_M;  // reuse what we have
  ROM_TRY(ensure_workspace(f, Mc_max));
  const size_t lds_back = size_t(std::max(f->nGa, 1)) * sizeof(double);
  ROM_CHECK(lds_back <= 60 * 1024, "rom_solve_batch: interface too large for the LDS-resident back substitution");
  // Sub-batches run on separate HIP streams: the wave-per-system diagonal kernels are latency bound
  // (one wave per SIMD), the MFMA kernels of another sub-batch fill the chip meanwhile.
  // (under per-kernel profiling the sweep stays on one stream: an event bracket then times its kernel alone)
  // (default: ONE stream.  Rounds 1-3 ran the tile Cholesky as two concurrent sub-batches, +1.4 % at C4 / +2.8 % at C5 then; with
  // round 4's kernels -- four workgroups per CU in the update and panel kernels -- one stream is 0.5-1 % FASTER at both
  // (profiles/r05_tile_cholesky_probes.txt).  ROMHC_STREAMS=2..4 still splits the sweep.)
  const int want = ctx->profile ? 1 : ctx->n_streams > 0 ? ctx->n_streams : 1;
  const int nsub = std::max(1, std::min(want, (M + 255) / 256));
  if (nsub > 1) {  // (no marker packet in the compute queue when the sweep stays on one stream: it costs a kernel-to-kernel bubble)
    ROM_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int s = 1; s < nsub; ++s) ROM_HIP(hipStreamWaitEvent(ctx->aux[s - 1], ctx->ev_fork, 0));
  }
  for (int m0 = 0; m0 < M; m0 += Mc_max) {
    const int Mchunk = std::min(Mc_max, M - m0);
    const int per = ((Mchunk + nsub - 1) / nsub + 63) / 64 * 64;
    for (int s = 0; s < nsub; ++s) {
      const int off = s * per;
      if (off >= Mchunk) break;
      const int Mc = std::min(per, Mchunk - off);
      hipStream_t st = s == 0 ? ctx->stream : ctx->aux[s - 1];
      ctx->prof_stream = st;
      FemDev d = make_dev(f);
      d.L += size_t(off) * f->nslots * 4096;
      d.invL += size_t(off) * f->T * 4096;
      if (Y) d.y = Y->p + size_t(y_row0 + m0 + off) * f->nGp;
      else d.y += size_t(off) * f->nGp;
      d.yhat += size_t(off) * f->nGp;
      if (d.gdots) d.gdots += size_t(off) * f->ncf * 8;
      ROM_TRY(enqueue_solve(f, d, a->p + size_t(m0 + off) * kblk, Mc, U ? U->p : nullptr, (long long)(row0 + m0 + off), st,
                            lds_back, stages));
    }
    ctx->prof_stream = nullptr;
    if (m0 + Mc_max < M) {  // the workspace is reused by the next chunk: join first
      for (int s = 1; s < nsub; ++s) {
        ROM_HIP(hipEventRecord(ctx->ev_join[s - 1], ctx->aux[s - 1]));
        ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[s - 1], 0));
      }
      if (nsub > 1) {
        ROM_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int s = 1; s < nsub; ++s) ROM_HIP(hipStreamWaitEvent(ctx->aux[s - 1], ctx->ev_fork, 0));
      }
    }
  }
  for (int s = 1; s < nsub; ++s) {
    ROM_HIP(hipEventRecord(ctx->ev_join[s - 1], ctx->aux[s - 1]));
    ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[s - 1], 0));
  }
  return check ? rom_solve_status(ctx) : ROM_OK;
}

extern "C" int rom_solve_batch(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0) {
  return solve_batch_impl(f, a, M, U, row0, true);
}

// Same without the host round trip: the sweep is only enqueued on the compute stream; a non-positive pivot
// is remembered on the device until rom_solve_status() is asked.
extern "C" int rom_solve_batch_async(rom_fem* f, rom_buf* a, int M, rom_buf* U, int64_t row0) {
  return solve_batch_impl(f, a, M, U, row0, false);
}

extern "C" int rom_fem_expansion_is_linear(rom_fem* f, int* flag) {
  ROM_CHECK(f && flag, "rom_fem_expansion_is_linear: null argument");
  *flag = f->npre == 0 ? 1 : 0;  // no edge is recovered node by node (k_back_pre weights its inputs with the parameters)
  return ROM_OK;
}

// [nodal_begin, nodal_end): the part of an interface vector that the expansion WRITES (nodal edge values) and never
// reads; everything else is its input
extern "C" int rom_fem_reduced_layout(rom_fem* f, int64_t* nodal_begin, int64_t* nodal_end) {
  ROM_CHECK(f && nodal_begin && nodal_end, "rom_fem_reduced_layout: null argument");
  *nodal_begin = f->nGa;
  *nodal_end = f->xb0;
  return ROM_OK;
}

// ---- compact form of the interface vectors: what has to TRAVEL between ranks -------------------------------------------
// The nodal part [nodal_begin, nodal_end) is an output of the expansion, so an exchange only needs the other
// rom_fem_compact_stride() entries of a vector (272 of 784 at 256 x 256 / 2 x 2).  pack: Yc[c_row0 + m, :] = the entries
// of Y[y_row0 + m, :] outside the nodal part, in order; unpack: the inverse (nodal part set to zero).  Both only enqueue.
__global__ void k_pack_reduced(const double* __restrict__ Y, long long ystride, int nb, int ne, double* __restrict__ Yc, int kc) {
  const double* y = Y + blockIdx.x * ystride;
  double* yc = Yc + blockIdx.x * (long long)kc;
  for (int c = threadIdx.x; c < kc; c += blockDim.x) yc[c] = y[c < nb ? c : c + (ne - nb)];
}
__global__ void k_unpack_reduced(const double* __restrict__ Yc, int kc, int nb, int ne, double* __restrict__ Y, long long ystride) {
  double* y = Y + blockIdx.x * ystride;
  const double* yc = Yc + blockIdx.x * (long long)kc;
  for (int c = threadIdx.x; c < int(ystride); c += blockDim.x) y[c] = c < nb ? yc[c] : (c < ne ? 0.0 : yc[c - (ne - nb)]);
}

extern "C" int rom_fem_compact_stride(rom_fem* f, int64_t* stride) {
  ROM_CHECK(f && stride, "rom_fem_compact_stride: null argument");
  *stride = f->nGp - (f->xb0 - f->nGa);
  return ROM_OK;
}

// (internal: the pack on a given stream -- rom_comm_allgather_packed_async runs it on the communication stream)
int rom_launch_pack_reduced(rom_fem* f, const double* Y, double* Yc, int M, hipStream_t st) {
  const int kc = f->nGp - (f->xb0 - f->nGa);
  if (M <= 0 || kc <= 0) return ROM_OK;
  k_pack_reduced<<<M, 256, 0, st>>>(Y, f->nGp, f->nGa, f->xb0, Yc, kc);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_fem_pack_reduced_async(rom_fem* f, rom_buf* Y, int64_t y_row0, int M, rom_buf* Yc, int64_t c_row0) {
  ROM_CHECK(f && Y && Yc, "rom_fem_pack_reduced_async: null argument");
  const int kc = f->nGp - (f->xb0 - f->nGa);
  ROM_CHECK(M >= 0 && y_row0 >= 0 && c_row0 >= 0 && Y->n >= size_t(y_row0 + M) * f->nGp && Yc->n >= size_t(c_row0 + M) * kc,
            "rom_fem_pack_reduced_async: buffers too small");
  if (M == 0 || kc == 0) return ROM_OK;
  ROM_HIP(hipSetDevice(f->ctx->device));
  k_pack_reduced<<<M, 256, 0, f->ctx->stream>>>(Y->p + size_t(y_row0) * f->nGp, f->nGp, f->nGa, f->xb0, Yc->p + size_t(c_row0) * kc, kc);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_fem_unpack_reduced_async(rom_fem* f, rom_buf* Yc, int64_t c_row0, int M, rom_buf* Y, int64_t y_row0) {
  ROM_CHECK(f && Y && Yc, "rom_fem_unpack_reduced_async: null argument");
  const int kc = f->nGp - (f->xb0 - f->nGa);
  ROM_CHECK(M >= 0 && y_row0 >= 0 && c_row0 >= 0 && Y->n >= size_t(y_row0 + M) * f->nGp && Yc->n >= size_t(c_row0 + M) * kc,
            "rom_fem_unpack_reduced_async: buffers too small");
  if (M == 0 || f->nGp == 0) return ROM_OK;
  ROM_HIP(hipSetDevice(f->ctx->device));
  k_unpack_reduced<<<M, 256, 0, f->ctx->stream>>>(Yc->p + size_t(c_row0) * kc, kc, f->nGa, f->xb0, Y->p + size_t(y_row0) * f->nGp, f->nGp);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_fem_reduced_stride(rom_fem* f, int64_t* stride) {
  ROM_CHECK(f && stride, "rom_fem_reduced_stride: null argument");
  *stride = f->nGp;
  return ROM_OK;
}

// Stage 1 only: Y[y_row0 + m, :] = interface vector of system m (reduced unknowns, cross points, the coefficient
// blocks the extension reads): everything that depends on the solve, 1/85 of a snapshot row at 256x256 / 2x2.
extern "C" int rom_solve_reduced_async(rom_fem* f, rom_buf* a, int M, rom_buf* Y, int64_t y_row0) {
  ROM_CHECK(f && Y, "rom_solve_reduced_async: null argument");
  ROM_CHECK(M >= 0 && y_row0 >= 0 && Y->n >= size_t(y_row0 + M) * f->nGp, "rom_solve_reduced_async: Y too small");
  // padding slots are read against zero table entries: they must hold finite numbers (k_solve1 zeroes its vector itself)
  if (M > 0 && f->nGp > 0 && !(f->fused1 && !f->sw_no_fused))
    ROM_HIP(hipMemsetAsync(Y->p + size_t(y_row0) * f->nGp, 0, size_t(M) * f->nGp * sizeof(double), f->ctx->stream));
  return solve_batch_impl(f, a, M, nullptr, 0, false, Y, y_row0, 1);
}

// Stage 2 only: snapshot rows U[row0 + m, :] from the interface vectors Y[y_row0 + m, :] (of this or any other
// rank: the expansion is deterministic, so every rank reproduces the owner's rows bit for bit).
extern "C" int rom_expand_batch_async(rom_fem* f, rom_buf* a, int M, rom_buf* Y, int64_t y_row0, rom_buf* U, int64_t row0) {
  ROM_CHECK(f && Y && U, "rom_expand_batch_async: null argument");
  return solve_batch_impl(f, a, M, U, row0, false, Y, y_row0, 2);
}

extern "C" int rom_solve_status(rom_ctx* ctx) {
  ROM_CHECK(ctx, "rom_solve_status: null context");
  int status = 0;
  ROM_HIP(hipMemcpyAsync(&status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  if (status != 0) {
    ROM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    rom_set_error("rom_solve_batch: interface matrix not positive definite (non-positive pivot); "
                  "all block coefficients must be > 0");
    return ROM_ERR_NOT_SPD;
  }
  return ROM_OK;
}

