// RCCL all-gather of the snapshot block over xGMI.
//
// The reference has no communication backend (its only parallelism is a process-pool `map`,
// src/lib/SolutionsManagers.py:51); the sweep shards contiguously over the GPUs of a node and the
// (M/G, dim) fp64 shards are exchanged once with ncclAllGather before the basis stage.
// librccl.so.1 is dlopen()ed on first use so that single-GPU runs carry no RCCL dependency.
#include <algorithm>
#include <dlfcn.h>

#include <cstring>

#include "romhc_internal.h"

namespace {

// minimal slice of the RCCL ABI (rccl.h): opaque comm, 128-byte unique id, enums
typedef struct { char internal[128]; } nccl_uid_t;
typedef void* nccl_comm_t;
enum { NCCL_FLOAT64 = 8 };            // ncclDouble
enum { NCCL_SUM = 0, NCCL_MAX = 2 };  // ncclSum, ncclMax

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_uid_t*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return ROM_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    rom_set_error("cannot dlopen librccl: %s", dlerror());
    return ROM_ERR_COMM;
  }
#define SYM(field, name)                                               \
  *(void**)(&g_rccl.field) = dlsym(h, name);                           \
  if (!g_rccl.field) {                                                 \
    rom_set_error("librccl lacks symbol %s", name);                    \
    return ROM_ERR_COMM;                                               \
  }
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.h = h;
  return ROM_OK;
}

#define ROM_NCCL(call)                                                                   \
  do {                                                                                   \
    int _r = (call);                                                                     \
    if (_r != 0) {                                                                       \
      rom_set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return ROM_ERR_COMM;                                                               \
    }                                                                                    \
  } while (0)

}  // namespace

extern "C" int rom_comm_unique_id(char* id_out, size_t cap) {
  ROM_CHECK(id_out && cap >= sizeof(nccl_uid_t), "rom_comm_unique_id: need a %zu-byte buffer", sizeof(nccl_uid_t));
  ROM_TRY(load_rccl());
  nccl_uid_t id;
  ROM_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return ROM_OK;
}

extern "C" int rom_comm_init(rom_ctx* ctx, const char* id, size_t id_len, int rank, int nranks) {
  ROM_CHECK(ctx && id && id_len >= sizeof(nccl_uid_t), "rom_comm_init: bad arguments");
  ROM_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "rom_comm_init: rank %d / %d", rank, nranks);
  ROM_CHECK(!ctx->comm, "rom_comm_init: communicator already initialised");
  ROM_TRY(load_rccl());
  ROM_HIP(hipSetDevice(ctx->device));
  nccl_uid_t uid;
  memcpy(&uid, id, sizeof(uid));
  nccl_comm_t comm = nullptr;
  ROM_NCCL(g_rccl.CommInitRank(&comm, nranks, uid, rank));
  ctx->comm = comm;
  ctx->rank = rank;
  ctx->nranks = nranks;
  return ROM_OK;
}

extern "C" int rom_comm_destroy(rom_ctx* ctx) {
  if (!ctx || !ctx->comm) return ROM_OK;
  hipStreamSynchronize(ctx->stream);
  if (ctx->comm_stream) {
    hipStreamSynchronize(ctx->comm_stream);
    hipStreamDestroy(ctx->comm_stream);
    hipEventDestroy(ctx->ev_comm);
    for (int i = 0; i < 2; ++i) {
      hipEventDestroy(ctx->ev_slot[i]);
      ctx->ev_slot[i] = nullptr;
      ctx->slot_used[i] = false;
      ctx->slot_joined[i] = true;
    }
    ctx->comm_stream = nullptr;
    ctx->ev_comm = nullptr;
  }
  g_rccl.CommDestroy((nccl_comm_t)ctx->comm);
  ctx->comm = nullptr;
  ctx->rank = 0;
  ctx->nranks = 1;
  return ROM_OK;
}

extern "C" int rom_comm_allgather(rom_ctx* ctx, rom_buf* send, size_t send_off, rom_buf* recv, size_t recv_off,
                                  size_t count) {
  ROM_CHECK(ctx && send && recv, "rom_comm_allgather: null argument");
  ROM_CHECK(ctx->comm, "rom_comm_allgather: communicator not initialised (rom_comm_init)");
  ROM_CHECK(send_off + count <= send->n, "rom_comm_allgather: send range out of bounds");
  ROM_CHECK(recv_off + count * size_t(ctx->nranks) <= recv->n, "rom_comm_allgather: recv range out of bounds");
  {
    ROM_PROF(ctx, "rccl_allgather", 0, 8.0 * count * ctx->nranks);
    ROM_NCCL(g_rccl.AllGather(send->p + send_off, recv->p + recv_off, count, NCCL_FLOAT64, (nccl_comm_t)ctx->comm,
                              ctx->stream));
  }
  return ROM_OK;
}

// Ranges a slot's outstanding collective(s) touch.  A slot that is issued again before the compute stream has been ordered
// behind its previous collective keeps the UNION of the old and the new ranges: the slot's event is recorded again behind
// the new collective (the communication stream runs them in order), so one wait covers both, and rom_buf_free must go on
// seeing the buffers of the earlier one.
static void slot_ranges(rom_ctx* ctx, int slot, const double* const lo[3], const double* const hi[3]) {
  const bool merge = ctx->slot_used[slot] && !ctx->slot_joined[slot];
  for (int k = 0; k < 3; ++k) {
    if (merge && ctx->slot_lo[slot][k] != ctx->slot_hi[slot][k] && lo[k] != hi[k]) {
      ctx->slot_lo[slot][k] = std::min(ctx->slot_lo[slot][k], lo[k]);
      ctx->slot_hi[slot][k] = std::max(ctx->slot_hi[slot][k], hi[k]);
    } else if (!merge || lo[k] != hi[k]) {
      ctx->slot_lo[slot][k] = lo[k];
      ctx->slot_hi[slot][k] = hi[k];
    }
  }
  ctx->slot_used[slot] = true;
  ctx->slot_joined[slot] = false;
}

// Overlapped form: the collective runs on the context's communication stream, ordered after everything
// enqueued so far on the compute stream; the compute stream is NOT blocked, so the next sweep step can
// run while the shards travel over xGMI.  rom_comm_wait() makes the compute stream (and the host, if
// `host_sync`) wait for all outstanding collectives.
extern "C" int rom_comm_allgather_async(rom_ctx* ctx, rom_buf* send, size_t send_off, rom_buf* recv, size_t recv_off,
                                        size_t count, int slot) {
  ROM_CHECK(slot >= 0 && slot < 2, "rom_comm_allgather_async: slot must be 0 or 1");
  ROM_CHECK(ctx && send && recv, "rom_comm_allgather_async: null argument");
  ROM_CHECK(ctx->comm, "rom_comm_allgather_async: communicator not initialised (rom_comm_init)");
  ROM_CHECK(send_off + count <= send->n, "rom_comm_allgather_async: send range out of bounds");
  ROM_CHECK(recv_off + count * size_t(ctx->nranks) <= recv->n, "rom_comm_allgather_async: recv range out of bounds");
  if (!ctx->comm_stream) {
    ROM_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    ROM_HIP(hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) ROM_HIP(hipEventCreateWithFlags(&ctx->ev_slot[i], hipEventDisableTiming));
  }
  ROM_HIP(hipEventRecord(ctx->ev_comm, ctx->stream));
  ROM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm, 0));
  ROM_NCCL(g_rccl.AllGather(send->p + send_off, recv->p + recv_off, count, NCCL_FLOAT64, (nccl_comm_t)ctx->comm,
                            ctx->comm_stream));
  ROM_HIP(hipEventRecord(ctx->ev_slot[slot], ctx->comm_stream));
  {
    const double* lo[3] = {send->p + send_off, recv->p + recv_off, nullptr};
    const double* hi[3] = {send->p + send_off + count, recv->p + recv_off + count * size_t(ctx->nranks), nullptr};
    slot_ranges(ctx, slot, lo, hi);
  }
  return ROM_OK;
}

// The step of a sharded sweep in one call: the interface vectors Y[y_row0 .. + M) of the own shard are packed into their
// compact form (rom_fem_compact_stride() doubles each) and all-gathered -- BOTH on the communication stream, ordered after
// everything enqueued so far on the compute stream, which goes straight on to the expansion (a pack kernel between the
// reduced solves and the extension costs the compute stream 8 us per step at C2).  `send`: scratch for the packed shard
// (M x compact stride), `recv`: nranks x M x compact stride from recv_off.  Slot semantics as rom_comm_allgather_async.
extern "C" int rom_comm_allgather_packed_async(rom_fem* f, rom_buf* Y, int64_t y_row0, int M, rom_buf* send, rom_buf* recv,
                                               size_t recv_off, int slot) {
  ROM_CHECK(f && Y && send && recv, "rom_comm_allgather_packed_async: null argument");
  rom_ctx* ctx = f->ctx;
  ROM_CHECK(slot >= 0 && slot < 2, "rom_comm_allgather_packed_async: slot must be 0 or 1");
  ROM_CHECK(ctx->comm, "rom_comm_allgather_packed_async: communicator not initialised (rom_comm_init)");
  const size_t kc = size_t(f->nGp - (f->xb0 - f->nGa));
  const size_t count = size_t(M) * kc;
  ROM_CHECK(M >= 0 && y_row0 >= 0 && Y->n >= size_t(y_row0 + M) * f->nGp && send->n >= count &&
                recv_off + count * size_t(ctx->nranks) <= recv->n,
            "rom_comm_allgather_packed_async: buffers too small");
  if (!ctx->comm_stream) {
    ROM_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    ROM_HIP(hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) ROM_HIP(hipEventCreateWithFlags(&ctx->ev_slot[i], hipEventDisableTiming));
  }
  ROM_HIP(hipEventRecord(ctx->ev_comm, ctx->stream));
  ROM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm, 0));
  ROM_TRY(rom_launch_pack_reduced(f, Y->p + size_t(y_row0) * f->nGp, send->p, M, ctx->comm_stream));
  if (count)
    ROM_NCCL(g_rccl.AllGather(send->p, recv->p + recv_off, count, NCCL_FLOAT64, (nccl_comm_t)ctx->comm, ctx->comm_stream));
  ROM_HIP(hipEventRecord(ctx->ev_slot[slot], ctx->comm_stream));
  {  // (the slot's buffers: rom_buf_free waits for the slot when one of them is freed)
    const double* lo[3] = {Y->p + size_t(y_row0) * f->nGp, recv->p + recv_off, send->p};
    const double* hi[3] = {Y->p + size_t(y_row0 + M) * f->nGp, recv->p + recv_off + count * size_t(ctx->nranks), send->p + count};
    slot_ranges(ctx, slot, lo, hi);
  }
  return ROM_OK;
}

// compute stream waits until the collective last issued with this slot has finished (its send buffer
// may then be overwritten by the next sweep step)
extern "C" int rom_comm_wait_slot(rom_ctx* ctx, int slot) {
  ROM_CHECK(ctx && slot >= 0 && slot < 2, "rom_comm_wait_slot: bad arguments");
  if (!ctx->comm_stream || !ctx->slot_used[slot] || ctx->slot_joined[slot]) return ROM_OK;
  // In the steady state of a sweep the collective of two steps ago finished long before the host gets here: then nothing
  // needs to be enqueued -- a wait packet in the compute queue costs a kernel-to-kernel bubble even when its event is done.
  const hipError_t q = hipEventQuery(ctx->ev_slot[slot]);
  if (q == hipErrorNotReady) {
    (void)hipGetLastError();  // (not an error: the collective is still in flight)
    ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_slot[slot], 0));
  } else {
    ROM_HIP(q);
  }
  ctx->slot_joined[slot] = true;
  return ROM_OK;
}

extern "C" int rom_comm_wait(rom_ctx* ctx, int host_sync) {
  ROM_CHECK(ctx, "rom_comm_wait: null context");
  if (!ctx->comm_stream) return ROM_OK;
  ROM_HIP(hipEventRecord(ctx->ev_comm, ctx->comm_stream));
  ROM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_comm, 0));
  ctx->slot_joined[0] = ctx->slot_joined[1] = true;
  if (host_sync) ROM_HIP(hipStreamSynchronize(ctx->comm_stream));
  return ROM_OK;
}

extern "C" int rom_comm_allreduce_host(rom_ctx* ctx, double* vals, int n, int op) {
  ROM_CHECK(ctx && vals && n >= 0 && n <= 1024, "rom_comm_allreduce_host: bad arguments");
  ROM_CHECK(ctx->comm, "rom_comm_allreduce_host: communicator not initialised (rom_comm_init)");
  if (n == 0) return ROM_OK;
  double* d = nullptr;
  ROM_TRY(rom_ctx_scratch(ctx, size_t(n), &d));
  ROM_HIP(hipMemcpyAsync(d, vals, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int r = g_rccl.AllReduce(d, d, n, NCCL_FLOAT64, op == 1 ? NCCL_MAX : NCCL_SUM, (nccl_comm_t)ctx->comm, ctx->stream);
  if (r != 0) {
    rom_set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    return ROM_ERR_COMM;
  }
  ROM_HIP(hipMemcpyAsync(vals, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ROM_HIP(hipStreamSynchronize(ctx->stream));
  return ROM_OK;
}
