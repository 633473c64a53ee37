#include <mutex>
// Context, device buffers, error strings, HIP-event stopwatch and per-kernel profiling.
#include "romhc_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

static thread_local char g_err[1024] = "";

void rom_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rom_last_error(void) { return g_err; }
extern "C" int rom_version(void) { return 100; }

extern "C" int rom_device_count(int* n) {
  ROM_CHECK(n, "rom_device_count: null output");
  ROM_HIP(hipGetDeviceCount(n));
  return ROM_OK;
}

extern "C" int rom_init(int device, rom_ctx** out) {
  ROM_CHECK(out, "rom_init: null output");
  int n = 0;
  ROM_HIP(hipGetDeviceCount(&n));
  ROM_CHECK(device >= 0 && device < n, "rom_init: device %d out of range (have %d)", device, n);
  ROM_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  ROM_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    rom_set_error("rom_init: device %d is %s; libromhc is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
    return ROM_ERR_INVALID;
  }
  rom_ctx* c = new rom_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  ROM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (int i = 0; i < 3; ++i) {
    ROM_HIP(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
    ROM_HIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
  }
  ROM_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  if (const char* e = getenv("ROMHC_STREAMS")) c->n_streams = std::max(1, std::min(4, atoi(e)));
  ROM_HIP(hipEventCreate(&c->t0));
  ROM_HIP(hipEventCreate(&c->t1));
  ROM_HIP(hipMalloc(&c->d_status, sizeof(int)));
  ROM_HIP(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
  *out = c;
  return ROM_OK;
}

extern "C" int rom_shutdown(rom_ctx* c) {
  if (!c) return ROM_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  rom_comm_destroy(c);
  for (auto e : c->event_pool) hipEventDestroy(e);
  for (auto& r : c->prof_recs) {
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  for (auto& kv : c->free_blocks)
    for (double* p : kv.second) hipFree(p);
  c->free_blocks.clear();
  if (c->d_status) hipFree(c->d_status);
  if (c->d_scratch) hipFree(c->d_scratch);
  hipEventDestroy(c->t0);
  hipEventDestroy(c->t1);
  for (int i = 0; i < 3; ++i) {
    if (c->aux[i]) hipStreamDestroy(c->aux[i]);
    if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  hipStreamDestroy(c->stream);
  delete c;
  return ROM_OK;
}

extern "C" int rom_synchronize(rom_ctx* c) {
  ROM_CHECK(c, "null context");
  ROM_HIP(hipStreamSynchronize(c->stream));
  return ROM_OK;
}

extern "C" int rom_set_workspace_limit(rom_ctx* c, size_t bytes) {
  ROM_CHECK(c, "null context");
  c->ws_limit = bytes;
  return ROM_OK;
}

extern "C" int rom_device_name(rom_ctx* c, char* out, size_t cap) {
  ROM_CHECK(c && out && cap > 0, "bad arguments");
  hipDeviceProp_t prop;
  ROM_HIP(hipGetDeviceProperties(&prop, c->device));
  snprintf(out, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return ROM_OK;
}

int rom_ctx_scratch(rom_ctx* c, size_t n, double** out) {
  if (c->scratch_doubles < n) {
    ROM_HIP(hipStreamSynchronize(c->stream));
    if (c->d_scratch) ROM_HIP(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_doubles = 0;
    ROM_HIP(hipMalloc(&c->d_scratch, n * sizeof(double)));
    c->scratch_doubles = n;
  }
  if (getenv("ROMHC_POISON_WS")) ROM_HIP(hipMemsetAsync(c->d_scratch, 0xFF, n * sizeof(double), c->stream));
  *out = c->d_scratch;
  return ROM_OK;
}

// ---- stopwatch ---------------------------------------------------------------------------------
extern "C" int rom_timer_start(rom_ctx* c) {
  ROM_CHECK(c, "null context");
  ROM_HIP(hipEventRecord(c->t0, c->stream));
  return ROM_OK;
}

extern "C" int rom_timer_stop(rom_ctx* c, double* ms) {
  ROM_CHECK(c && ms, "bad arguments");
  ROM_HIP(hipEventRecord(c->t1, c->stream));
  ROM_HIP(hipEventSynchronize(c->t1));
  float f = 0;
  ROM_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
  *ms = f;
  return ROM_OK;
}

// ---- debugging aid: NaN patterns in the LDS of every CU before a kernel starts ---------------------
// (ROMHC_POISON_LDS=1: a kernel that reads LDS it has not written picks up whatever the previous workgroup on
// that CU left there -- harmless zeros-times-finite most of the time, NaN once in a while.  160 KB per workgroup:
// one workgroup per CU at a time, 4 x the CU count so that every CU gets at least one.)
__global__ __launch_bounds__(256) void k_poison_lds() {
  extern __shared__ unsigned long long poison[];
  for (int i = threadIdx.x; i < 160 * 1024 / 8; i += 256) poison[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();
  if (poison[(threadIdx.x * 977) % (160 * 1024 / 8)] == 0) __builtin_trap();  // keep the stores
}

static void poison_lds(hipStream_t st) {
  static bool ready = false;
  if (!ready) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    ready = true;
  }
  k_poison_lds<<<1024, 256, 160 * 1024, st>>>();
  const hipError_t e = hipGetLastError();
  static bool told = false;
  if (!told) {
    fprintf(stderr, "romhc: ROMHC_POISON_LDS: poison kernel launch: %s\n", hipGetErrorString(e));
    told = true;
  }
}

// ---- per-kernel profiling ----------------------------------------------------------------------
ProfScope::ProfScope(rom_ctx* c, const char* name, double flops, double bytes) : ctx(c) {
  if (getenv("ROMHC_POISON_LDS")) poison_lds(c->prof_stream ? c->prof_stream : c->stream);
  if (!c->profile) return;
  int id = -1;
  for (size_t i = 0; i < c->prof_names.size(); ++i)
    if (c->prof_names[i] == name) id = int(i);
  if (id < 0) {
    id = int(c->prof_names.size());
    c->prof_names.push_back(name);
    c->prof_flops.push_back(0);
    c->prof_bytes.push_back(0);
  }
  c->prof_flops[id] += flops;
  c->prof_bytes[id] += bytes;
  ProfRec r;
  r.name_id = id;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  r.st = c->prof_stream ? c->prof_stream : c->stream;
  hipEventRecord(r.e0, r.st);
  c->prof_recs.push_back(r);
  idx = int(c->prof_recs.size()) - 1;
}

ProfScope::~ProfScope() {
  if (idx >= 0) hipEventRecord(ctx->prof_recs[idx].e1, ctx->prof_recs[idx].st);
}

extern "C" int rom_profile_enable(rom_ctx* c, int on) {
  ROM_CHECK(c, "null context");
  c->profile = on != 0;
  return ROM_OK;
}

extern "C" int rom_profile_reset(rom_ctx* c) {
  ROM_CHECK(c, "null context");
  ROM_HIP(hipStreamSynchronize(c->stream));
  for (auto& r : c->prof_recs) {
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  c->prof_recs.clear();
  c->prof_names.clear();
  c->prof_flops.clear();
  c->prof_bytes.clear();
  return ROM_OK;
}

extern "C" int rom_profile_count(rom_ctx* c, int* n) {
  ROM_CHECK(c && n, "bad arguments");
  *n = int(c->prof_names.size());
  return ROM_OK;
}

extern "C" int rom_profile_query(rom_ctx* c, int idx, char* name, size_t cap, double* total_ms,
                                 long* launches, double* flops, double* bytes) {
  ROM_CHECK(c && idx >= 0 && idx < int(c->prof_names.size()), "rom_profile_query: bad index");
  ROM_HIP(hipStreamSynchronize(c->stream));
  double tot = 0;
  long cnt = 0;
  for (auto& r : c->prof_recs) {
    if (r.name_id != idx) continue;
    float f = 0;
    ROM_HIP(hipEventElapsedTime(&f, r.e0, r.e1));
    tot += f;
    ++cnt;
  }
  if (name && cap) snprintf(name, cap, "%s", c->prof_names[idx].c_str());
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  if (flops) *flops = c->prof_flops[idx];
  if (bytes) *bytes = c->prof_bytes[idx];
  return ROM_OK;
}

// ---- buffers -------------------------------------------------------------------------------------
// Device buffers come from a small caching allocator: the host-side drivers (greedy, POD) create and
// drop many temporaries per iteration, and hipMalloc / hipFree are slow and synchronise the device.
// A freed block goes back to a per-size free list and is handed to the next request of the same
// rounded size.  Reuse rule: the next user of a recycled block enqueues on the context's in-order compute
// stream, so the block is safe once everything that touched it is ordered before the compute stream's tail:
//   * kernels of the library run on the compute stream itself, or on the sub-batch streams of a sweep, which
//     rom_solve_batch joins back into the compute stream before it returns;
//   * collectives run on the communication stream and are NOT joined by themselves: rom_buf_free therefore
//     makes the compute stream wait for the end of an outstanding collective (its slot's event, no host wait)
//     when -- and only when -- the freed block is one that collective reads or writes.
static size_t round_bytes(size_t bytes) {
  if (bytes < 4096) return 4096;
  if (bytes < (size_t(1) << 20)) return (bytes + 4095) / 4096 * 4096;
  return (bytes + (size_t(1) << 20) - 1) >> 20 << 20;  // 1 MiB granules above 1 MiB
}

extern "C" int rom_buf_alloc(rom_ctx* c, size_t n, rom_buf** out) {
  ROM_CHECK(c && out, "bad arguments");
  ROM_HIP(hipSetDevice(c->device));
  const size_t bytes = round_bytes((n ? n : 1) * sizeof(double));
  rom_buf* b = new rom_buf{c, nullptr, n};
  auto it = c->free_blocks.find(bytes);
  if (it != c->free_blocks.end() && !it->second.empty()) {
    b->p = it->second.back();
    it->second.pop_back();
    c->cached_bytes -= bytes;
  } else {
    hipError_t e = hipMalloc(&b->p, bytes);
    if (e != hipSuccess && c->cached_bytes > 0) {  // give the cache back and retry once
      hipStreamSynchronize(c->stream);
      for (auto& kv : c->free_blocks)
        for (double* p : kv.second) hipFree(p);
      c->free_blocks.clear();
      c->cached_bytes = 0;
      e = hipMalloc(&b->p, bytes);
    }
    if (e != hipSuccess) {
      delete b;
      rom_set_error("rom_buf_alloc: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
      return ROM_ERR_NOMEM;
    }
  }
  if (getenv("ROMHC_POISON_WS"))  // debugging aid: a fresh buffer holds NaN patterns, never stale numbers
    ROM_HIP(hipMemsetAsync(b->p, 0xFF, bytes, c->stream));
  *out = b;
  return ROM_OK;
}

extern "C" int rom_buf_free(rom_buf* b) {
  if (!b) return ROM_OK;
  rom_ctx* c = b->ctx;
  const size_t bytes = round_bytes((b->n ? b->n : 1) * sizeof(double));
  if (c->comm_stream) {
    // Only a block that an outstanding collective reads or writes needs the wait (temporaries of the basis stage that
    // are dropped during a step must not serialise the compute stream behind the all-gather it is meant to overlap):
    // the compute stream is ordered behind the END of that slot's collective, through the slot's own event.
    const double* lo = b->p;
    const double* hi = reinterpret_cast<const double*>(reinterpret_cast<const char*>(b->p) + bytes);
    for (int s = 0; s < 2; ++s) {
      if (!c->slot_used[s] || c->slot_joined[s]) continue;
      bool touches = false;
      for (int k = 0; k < 3; ++k) touches = touches || (c->slot_lo[s][k] < hi && lo < c->slot_hi[s][k]);
      if (!touches) continue;
      // (an error here must neither leak the block nor return it to the cache unordered: fall back to the host waiting for
      // the whole communication stream)
      if (hipSetDevice(c->device) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_slot[s], 0) != hipSuccess)
        hipStreamSynchronize(c->comm_stream);
      c->slot_joined[s] = true;
    }
  }
  if (c->cached_bytes + bytes <= c->cache_limit) {
    c->free_blocks[bytes].push_back(b->p);
    c->cached_bytes += bytes;
  } else {
    hipStreamSynchronize(c->stream);
    hipFree(b->p);
  }
  delete b;
  return ROM_OK;
}

// ---- page-locked host arrays for large downloads ---------------------------------------------------
// The reference API hands NumPy rows back to its caller (generate_solutions: (M, dim) doubles, 533 MB at C2), so the
// device -> host copy is part of every such call; into pageable memory it runs at 11-25 GB/s, into page-locked
// memory at the PCIe rate (48 GB/s measured).  Pinning is slow (page by page: 110 ms for 533 MB), so freed blocks
// are kept in a process-wide pool (finalisers of the NumPy arrays may run on any thread, after their context is
// gone: no rom_ctx here) and the caller decides when a new block is worth pinning (`pooled_only`).
namespace {
std::mutex g_host_mu;
std::map<size_t, std::vector<void*>> g_host_free;   // rounded bytes -> blocks
std::map<void*, size_t> g_host_live;                 // block -> rounded bytes
size_t g_host_cached = 0;
const size_t g_host_cache_limit = size_t(8) << 30;
}  // namespace

extern "C" int rom_host_alloc(size_t n, int pooled_only, double** out) {
  ROM_CHECK(out, "rom_host_alloc: null argument");
  *out = nullptr;
  const size_t bytes = round_bytes((n ? n : 1) * sizeof(double));
  std::lock_guard<std::mutex> lk(g_host_mu);
  void* p = nullptr;
  auto it = g_host_free.find(bytes);
  if (it != g_host_free.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
    g_host_cached -= bytes;
  } else if (pooled_only) {
    return ROM_OK;  // nothing of this size in the pool: *out stays null
  } else {
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      rom_set_error("rom_host_alloc: hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
      return ROM_ERR_NOMEM;
    }
  }
  g_host_live[p] = bytes;
  *out = static_cast<double*>(p);
  return ROM_OK;
}

extern "C" int rom_host_free(double* p) {
  if (!p) return ROM_OK;
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_host_live.find(p);
  if (it == g_host_live.end()) {
    rom_set_error("rom_host_free: not a block of rom_host_alloc");
    return ROM_ERR_INVALID;
  }
  const size_t bytes = it->second;
  g_host_live.erase(it);
  if (g_host_cached + bytes <= g_host_cache_limit) {
    g_host_free[bytes].push_back(p);
    g_host_cached += bytes;
  } else {
    (void)hipHostFree(p);
  }
  return ROM_OK;
}

extern "C" int rom_buf_size(rom_buf* b, size_t* n) {
  ROM_CHECK(b && n, "bad arguments");
  *n = b->n;
  return ROM_OK;
}

extern "C" int rom_buf_upload(rom_buf* b, size_t off, const double* host, size_t n) {
  ROM_CHECK(b && (host || n == 0), "bad arguments");
  ROM_CHECK(off + n <= b->n, "rom_buf_upload: range [%zu,%zu) exceeds buffer of %zu", off, off + n, b->n);
  if (n == 0) return ROM_OK;
  ROM_HIP(hipMemcpyAsync(b->p + off, host, n * sizeof(double), hipMemcpyHostToDevice, b->ctx->stream));
  ROM_HIP(hipStreamSynchronize(b->ctx->stream));  // host array may be reused by the caller
  return ROM_OK;
}

extern "C" int rom_buf_download(rom_buf* b, size_t off, double* host, size_t n) {
  ROM_CHECK(b && (host || n == 0), "bad arguments");
  ROM_CHECK(off + n <= b->n, "rom_buf_download: range [%zu,%zu) exceeds buffer of %zu", off, off + n, b->n);
  if (n == 0) return ROM_OK;
  ROM_HIP(hipMemcpyAsync(host, b->p + off, n * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
  ROM_HIP(hipStreamSynchronize(b->ctx->stream));
  return ROM_OK;
}

__global__ void k_fill(double* p, size_t n, double v) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

extern "C" int rom_buf_fill(rom_buf* b, size_t off, size_t n, double value) {
  ROM_CHECK(b, "bad arguments");
  ROM_CHECK(off + n <= b->n, "rom_buf_fill: range exceeds buffer");
  if (n == 0) return ROM_OK;
  if (value == 0.0) {
    ROM_HIP(hipMemsetAsync(b->p + off, 0, n * sizeof(double), b->ctx->stream));
    return ROM_OK;
  }
  int grid = int(std::min<size_t>((n + 255) / 256, 2048));
  k_fill<<<grid, 256, 0, b->ctx->stream>>>(b->p + off, n, value);
  ROM_HIP(hipGetLastError());
  return ROM_OK;
}

extern "C" int rom_buf_copy(rom_buf* dst, size_t dst_off, rom_buf* src, size_t src_off, size_t n) {
  ROM_CHECK(dst && src, "bad arguments");
  ROM_CHECK(dst_off + n <= dst->n && src_off + n <= src->n, "rom_buf_copy: range exceeds buffer");
  if (n == 0) return ROM_OK;
  ROM_HIP(hipMemcpyAsync(dst->p + dst_off, src->p + src_off, n * sizeof(double), hipMemcpyDeviceToDevice,
                         dst->ctx->stream));
  return ROM_OK;
}

// flag <- 1 if any of the n doubles differs in its bits (NaNs with equal bits count as equal: this compares storage)
__global__ void k_buf_differ(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, size_t n, int* __restrict__ flag) {
  bool d = false;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) d = d || a[i] != b[i];
  if (d) *flag = 1;
}

extern "C" int rom_buf_equal(rom_buf* A, size_t a_off, rom_buf* B, size_t b_off, size_t n, int* equal_host) {
  ROM_CHECK(A && B && equal_host, "rom_buf_equal: null argument");
  ROM_CHECK(A->ctx == B->ctx, "rom_buf_equal: buffers of different contexts");
  ROM_CHECK(a_off + n <= A->n && b_off + n <= B->n, "rom_buf_equal: range exceeds buffer");
  *equal_host = 1;
  if (n == 0) return ROM_OK;
  rom_ctx* ctx = A->ctx;
  int* d_flag = nullptr;
  ROM_HIP(hipMalloc(&d_flag, sizeof(int)));
  ROM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
  k_buf_differ<<<unsigned(std::min<size_t>((n + 255) / 256, 8192)), 256, 0, ctx->stream>>>(
      reinterpret_cast<const unsigned long long*>(A->p + a_off), reinterpret_cast<const unsigned long long*>(B->p + b_off), n, d_flag);
  int differ = 0;
  hipError_t e = hipMemcpyAsync(&differ, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_flag);
  ROM_HIP(e);
  *equal_host = differ ? 0 : 1;
  return ROM_OK;
}

extern "C" int rom_buf_gather_rows(rom_buf* dst, rom_buf* src, const int64_t* rows, int n_rows, size_t dim) {
  ROM_CHECK(dst && src && (rows || n_rows == 0), "bad arguments");
  ROM_CHECK(size_t(n_rows) * dim <= dst->n, "rom_buf_gather_rows: destination too small");
  for (int i = 0; i < n_rows; ++i) {
    ROM_CHECK(rows[i] >= 0 && size_t(rows[i] + 1) * dim <= src->n, "rom_buf_gather_rows: row %lld out of range",
              (long long)rows[i]);
    ROM_HIP(hipMemcpyAsync(dst->p + size_t(i) * dim, src->p + size_t(rows[i]) * dim, dim * sizeof(double),
                           hipMemcpyDeviceToDevice, dst->ctx->stream));
  }
  return ROM_OK;
}
