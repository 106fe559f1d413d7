// The one exchange step of the path behind the C-ABI: an RCCL communicator
// owned by the library -- one process per GPU, ranks joined by a 128-byte unique id that the host
// side passes around (torch.distributed only bootstraps it) -- and the gradient all-reduce issued
// on the library's own stream, ordered against the caller's stream by events, so that a native
// update (dx_cnn_ppo_epoch) can start the reduction of one half of the flat gradient buffer
// while the backward of the other half is still running.  derl has no distributed code; the step
// this sits inside is derl/alg/common.py:66-78 (Trainer.step: backward -> [all-reduce] -> clip ->
// optimizer step).
//
// ONE communicator is only ever driven from ONE stream: every collective -- the gradient
// all-reduces, the float64 statistics all-reduce, the parameter broadcast -- is enqueued on the
// library's stream behind a `ready` event of the caller's stream; the in-stream ones make the
// caller's stream wait for their `done` event at once, the gradient all-reduces at
// dx_allreduce_wait.
//
// RCCL is resolved with dlopen at dx_comm_init: the library has no link-time dependency on it
// (it loads, and every other entry point works, on a box without RCCL), and inside a torch
// process the copy torch already loaded is the one used (same HIP runtime).
#include "common.hpp"
#include <cstdlib>
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
  void *h = nullptr;
  // DERL_AMD_RCCL_LIBRARY=<path>: this RCCL build and no other (a site's own build; also how the
  // two-rank bootstrap test makes ONE rank really unable to load it)
  const char *forced = getenv("DERL_AMD_RCCL_LIBRARY");
  if (forced && *forced) {
    h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(DX_ENOSUP, "dx_comm: cannot load RCCL from DERL_AMD_RCCL_LIBRARY=%s: %s", forced, dlerror());
  }
  // the copy already in the process first (inside torch: torch/lib/librccl.so, soname librccl.so.1)
  const char *names[] = {"librccl.so.1", "librccl.so"};
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

#if DX_DIAG
// Ordering probe (diag flavour only; it computes WRONG results on purpose).
// DX_COMM_TEST_HOOK="<delay_us>:<scale>": behind every gradient all-reduce the library's stream
// first idles for delay_us and then multiplies the reduced buffer by `scale`.  At ONE rank an
// in-place all-reduce is the identity, so the plain one-rank run cannot see a reduction that
// starts before the backward has written its part of the buffer (that part would be overwritten:
// unscaled) or a norm / optimizer step that does not wait for the reduction (it would read the
// buffer before the delayed scaling).  With the hook the update equals a step on `scale` x the
// gradient only if both orderings hold: tests/dist_worker.py rccl_ordering.
__global__ void comm_test_hook_kernel(float *buf, long long count, float scale, long long delay_ticks) {
  const long long start = wall_clock64();  // constant 100 MHz counter: the spin ends on every wave
  while (wall_clock64() - start < delay_ticks) __builtin_amdgcn_s_sleep(32);
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    buf[i] *= scale;
}

int comm_test_hook(float *buf, long long count) {
  static const char *spec = getenv("DX_COMM_TEST_HOOK");
  if (spec == nullptr || *spec == 0) return DX_OK;
  char *end = nullptr;
  const double delay_us = strtod(spec, &end);
  const double scale = (end && *end == ':') ? strtod(end + 1, nullptr) : 1.0;
  const long long ticks = static_cast<long long>(delay_us < 0 ? 0 : (delay_us > 50000 ? 50000 : delay_us)) * 100;
  hipLaunchKernelGGL(comm_test_hook_kernel, dim3(256), dim3(256), 0, g_comm.stream, buf, count,
                     static_cast<float>(scale), ticks);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
#endif

// a collective of the communicator's stream that the caller's stream waits for at once
template <class Issue>
int comm_in_stream(const char *who, hipStream_t stream, Issue issue) {
  if (int rc = require_comm(who)) return rc;
  DX_HIP(hipEventRecord(g_comm.ready, stream));
  DX_HIP(hipStreamWaitEvent(g_comm.stream, g_comm.ready, 0));
  if (int rc = issue(g_comm.stream)) return rc;
  DX_HIP(hipEventRecord(g_comm.done, g_comm.stream));
  // (in-order communicator stream: `done` also covers the gradient reductions issued before it -- for THIS
  // caller stream.  `pending` stays set: the reductions' owner may be another stream, whose dx_allreduce_wait
  // must still wait; waiting again for the newest `done` event is harmless)
  DX_HIP(hipStreamWaitEvent(stream, g_comm.done, 0));
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
#if DX_DIAG
  if (int rc = comm_test_hook(buf, count)) return rc;
#endif
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

int dx_comm_available(void) {
  DX_TRACE("dx_comm_available");
  std::lock_guard<std::mutex> guard(g_lock);
  DX_REQUIRE(g_comm.comm == nullptr, "dx_comm_available: a communicator already exists (dx_comm_destroy first)");
  if (int rc = load_rccl()) return rc;
  // the device answers: its context comes up (hipFree(nullptr) creates it) and it reports its compute units
  int dev = -1, cus = 0;
  DX_HIP(hipGetDevice(&dev));
  DX_HIP(hipFree(nullptr));
  DX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  DX_REQUIRE(cus > 0, "dx_comm_available: device %d reports %d compute units", dev, cus);
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
  if (int rc = comm_in_stream("dx_allreduce_sum_f64", as_stream(stream), [&](hipStream_t cs) -> int {
        DX_NCCL(g_rccl.AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat64, ncclSum, g_comm.comm, cs));
        return DX_OK;
      }))
    return rc;
  ++g_comm.issued;
  g_comm.bytes += 8 * count;
  return DX_OK;
}

int dx_comm_broadcast_f32(float *buf, long long count, int root, void *stream) {
  DX_TRACE("dx_comm_broadcast_f32");
  DX_REQUIRE(buf != nullptr && count >= 1, "dx_comm_broadcast_f32: bad arguments");
  if (int rc = require_comm("dx_comm_broadcast_f32")) return rc;
  DX_REQUIRE(root >= 0 && root < g_comm.world, "dx_comm_broadcast_f32: root %d of %d ranks", root, g_comm.world);
  return comm_in_stream("dx_comm_broadcast_f32", as_stream(stream), [&](hipStream_t cs) -> int {
    DX_NCCL(g_rccl.Broadcast(buf, buf, static_cast<size_t>(count), ncclFloat32, root, g_comm.comm, cs));
    return DX_OK;
  });
}

}  // extern "C"
