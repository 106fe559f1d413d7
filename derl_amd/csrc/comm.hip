// The one exchange step of the path behind the C-ABI: an RCCL communicator
// owned by the library -- one process per GPU, ranks joined by a 128-byte unique id that the host
// side passes around (torch.distributed only bootstraps it) -- and the gradient all-reduce issued
// on the library's own stream, ordered against the caller's stream by events, so that a native
// update (dx_cnn_ppo_epoch) can start the reduction of one half of the flat gradient buffer
// while the backward of the other half is still running.  derl has no distributed code; the step
// this sits inside is derl/alg/common.py:66-78 (Trainer.step: backward -> [all-reduce] -> clip ->
// optimizer step).
//
// RCCL is resolved with dlopen at dx_comm_init: the library has no link-time dependency on it
// (it loads, and every other entry point works, on a box without RCCL), and inside a torch
// process the copy torch already loaded is the one used (same HIP runtime).
#include "common.hpp"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace dx {
namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 0, device = -1;
  hipStream_t stream = nullptr;  // the gradient all-reduces run here
  hipEvent_t ready = nullptr, done = nullptr;
  bool pending = false;          // an all-reduce was issued since the last dx_allreduce_wait
  long long issued = 0;          // all-reduces issued through the library (tests, bench self-check)
  long long bytes = 0;
};

Rccl g_rccl;
Comm g_comm;
std::mutex g_lock;

int load_rccl() {
  if (g_rccl.handle) return DX_OK;
  // the copy already in the process first (inside torch: torch/lib/librccl.so, soname librccl.so.1)
  const char *names[] = {"librccl.so.1", "librccl.so"};
  void *h = nullptr;
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(DX_ENOSUP, "dx_comm: cannot load RCCL (librccl.so.1): %s", dlerror());
  Rccl r;
  r.handle = h;
#define DX_SYM(field, name)                                             \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));       \
  if (!r.field) return fail(DX_ENOSUP, "dx_comm: RCCL has no symbol %s", name)
  DX_SYM(GetUniqueId, "ncclGetUniqueId");
  DX_SYM(CommInitRank, "ncclCommInitRank");
  DX_SYM(CommDestroy, "ncclCommDestroy");
  DX_SYM(AllReduce, "ncclAllReduce");
  DX_SYM(Broadcast, "ncclBroadcast");
  DX_SYM(GetErrorString, "ncclGetErrorString");
#undef DX_SYM
  g_rccl = r;
  return DX_OK;
}

#define DX_NCCL(expr)                                                                     \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess)                                                                \
      return ::dx::fail(DX_EHIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), \
                        __FILE__, __LINE__);                                              \
  } while (0)

int require_comm(const char *who) {
  if (g_comm.comm == nullptr) return fail(DX_EINVAL, "%s: no communicator (call dx_comm_init first)", who);
  int dev = -1;
  DX_HIP(hipGetDevice(&dev));
  if (dev != g_comm.device)
    return fail(DX_EINVAL, "%s: communicator lives on device %d, current device is %d", who, g_comm.device, dev);
  return DX_OK;
}

}  // namespace

// what the native update calls between the two halves of the backward (cnn_update.hip)
bool comm_active() { return g_comm.comm != nullptr; }

int comm_allreduce_async(float *buf, long long count, hipStream_t stream) {
  if (int rc = require_comm("dx_allreduce_grads")) return rc;
  // the reduction starts when everything enqueued on `stream` so far has finished ...
  DX_HIP(hipEventRecord(g_comm.ready, stream));
  DX_HIP(hipStreamWaitEvent(g_comm.stream, g_comm.ready, 0));
  DX_NCCL(g_rccl.AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat32, ncclSum, g_comm.comm, g_comm.stream));
  // ... and `stream` does NOT wait for it until dx_allreduce_wait
  DX_HIP(hipEventRecord(g_comm.done, g_comm.stream));
  g_comm.pending = true;
  ++g_comm.issued;
  g_comm.bytes += 4 * count;
  return DX_OK;
}

int comm_wait(hipStream_t stream) {
  if (!g_comm.pending) return DX_OK;
  DX_HIP(hipStreamWaitEvent(stream, g_comm.done, 0));  // in-order comm stream: the last event covers all
  g_comm.pending = false;
  return DX_OK;
}

}  // namespace dx

using namespace dx;

extern "C" {

int dx_comm_unique_id(void *id_out_host) {
  DX_TRACE("dx_comm_unique_id");
  DX_REQUIRE(id_out_host != nullptr, "dx_comm_unique_id: null output");
  std::lock_guard<std::mutex> guard(g_lock);
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == DX_COMM_ID_BYTES, "unique id size");
  DX_NCCL(g_rccl.GetUniqueId(static_cast<ncclUniqueId *>(id_out_host)));
  return DX_OK;
}

int dx_comm_init(const void *unique_id_host, int rank, int world) {
  DX_TRACE("dx_comm_init");
  DX_REQUIRE(unique_id_host != nullptr && world >= 1 && rank >= 0 && rank < world,
             "dx_comm_init: bad arguments (rank %d of %d)", rank, world);
  std::lock_guard<std::mutex> guard(g_lock);
  DX_REQUIRE(g_comm.comm == nullptr, "dx_comm_init: a communicator already exists (dx_comm_destroy first)");
  if (int rc = load_rccl()) return rc;
  Comm c;
  c.rank = rank;
  c.world = world;
  DX_HIP(hipGetDevice(&c.device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id_host, sizeof(id));
  DX_NCCL(g_rccl.CommInitRank(&c.comm, world, id, rank));
  int lo = 0, hi = 0;  // the reduction must not queue behind the conv kernels it overlaps
  const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
  const bool ok = (prio ? hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi)
                        : hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking)) == hipSuccess &&
                  hipEventCreateWithFlags(&c.ready, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&c.done, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    g_rccl.CommDestroy(c.comm);
    return fail(DX_EHIP, "dx_comm_init: cannot create the communication stream / events");
  }
  g_comm = c;
  return DX_OK;
}

int dx_comm_info(int *rank, int *world, long long *allreduces_issued, long long *allreduce_bytes) {
  std::lock_guard<std::mutex> guard(g_lock);
  if (rank) *rank = g_comm.comm ? g_comm.rank : 0;
  if (world) *world = g_comm.comm ? g_comm.world : 0;
  if (allreduces_issued) *allreduces_issued = g_comm.issued;
  if (allreduce_bytes) *allreduce_bytes = g_comm.bytes;
  return DX_OK;
}

int dx_comm_destroy(void) {
  DX_TRACE("dx_comm_destroy");
  std::lock_guard<std::mutex> guard(g_lock);
  if (g_comm.comm == nullptr) return DX_OK;
  (void)hipStreamSynchronize(g_comm.stream);
  const ncclResult_t r = g_rccl.CommDestroy(g_comm.comm);
  (void)hipEventDestroy(g_comm.ready);
  (void)hipEventDestroy(g_comm.done);
  (void)hipStreamDestroy(g_comm.stream);
  g_comm = Comm();
  if (r != ncclSuccess) return fail(DX_EHIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
  return DX_OK;
}

int dx_allreduce_grads(float *flat_grads, long long count, void *stream) {
  DX_TRACE("dx_allreduce_grads");
  DX_REQUIRE(flat_grads != nullptr && count >= 1, "dx_allreduce_grads: bad arguments");
  return comm_allreduce_async(flat_grads, count, as_stream(stream));
}

int dx_allreduce_wait(void *stream) {
  DX_TRACE("dx_allreduce_wait");
  if (g_comm.comm == nullptr) return DX_OK;
  return comm_wait(as_stream(stream));
}

int dx_allreduce_sum_f64(double *buf, long long count, void *stream) {
  DX_TRACE("dx_allreduce_sum_f64");
  DX_REQUIRE(buf != nullptr && count >= 1, "dx_allreduce_sum_f64: bad arguments");
  if (int rc = require_comm("dx_allreduce_sum_f64")) return rc;
  DX_NCCL(g_rccl.AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat64, ncclSum, g_comm.comm, as_stream(stream)));
  ++g_comm.issued;
  g_comm.bytes += 8 * count;
  return DX_OK;
}

int dx_comm_broadcast_f32(float *buf, long long count, int root, void *stream) {
  DX_TRACE("dx_comm_broadcast_f32");
  DX_REQUIRE(buf != nullptr && count >= 1, "dx_comm_broadcast_f32: bad arguments");
  if (int rc = require_comm("dx_comm_broadcast_f32")) return rc;
  DX_REQUIRE(root >= 0 && root < g_comm.world, "dx_comm_broadcast_f32: root %d of %d ranks", root, g_comm.world);
  DX_NCCL(g_rccl.Broadcast(buf, buf, static_cast<size_t>(count), ncclFloat32, root, g_comm.comm, as_stream(stream)));
  return DX_OK;
}

}  // extern "C"
