"""Data-parallel plumbing: one process per GPU.

The data path's collectives -- the gradient all-reduce per optimizer step, the
per-rollout all-reduce of the minibatch advantage statistics and the initial parameter broadcast --
run on an RCCL communicator owned by the native library (``dx_comm_init`` / ``dx_allreduce_grads``,
include/derl_amd.h), so that a native update can issue them from inside one C call.
``torch.distributed`` (backend "nccl" = RCCL on ROCm) is the launcher contract and the bootstrap:
it carries the communicator's 128-byte unique id from rank 0 to the others, barriers and the
bench's max-over-ranks -- nothing on the data path.

Backend "gloo" is a REHEARSAL mode for boxes with fewer GPUs than ranks (RCCL refuses two ranks on
one GPU): the same sharding rules with torch.distributed doing the reductions.
``DERL_AMD_NATIVE_COMM=0`` keeps the library's communicator out (torch.distributed's RCCL
collectives from Python, update by update: the round-2 path).
``DERL_AMD_FORCE_COLLECTIVES=1`` makes a single process take the sharded code path end to end
(RCCL accepts a one-rank communicator): what the GPU suite uses to execute the RCCL branch on a
one-GPU box."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib

_native = False  # the library's communicator exists


def is_initialized():
  return dist.is_available() and dist.is_initialized()


def world_size():
  return dist.get_world_size() if is_initialized() else 1


def rank():
  return dist.get_rank() if is_initialized() else 0


def forced():
  return os.environ.get("DERL_AMD_FORCE_COLLECTIVES", "0") not in ("", "0")


def sharded():
  """True when the data path must issue its collectives: more than one rank, or a single rank
  with the collectives forced (tests)."""
  return world_size() > 1 or (forced() and _native)


def native_comm():
  return _native


def init_from_env(backend=None):
  """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (the
  torch.distributed.run contract) if WORLD_SIZE > 1 (or the collectives are forced), binds this
  rank to its GPU and -- on backend nccl -- creates the library's RCCL communicator."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if (world <= 1 and not forced()) or is_initialized():
    return world_size()
  local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
  if backend is None:
    # DERL_AMD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("DERL_AMD_DIST_BACKEND") or (
        "nccl" if torch.cuda.is_available() else "gloo")
  if backend == "nccl":
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29517")
  os.environ.setdefault("RANK", "0")
  os.environ.setdefault("WORLD_SIZE", "1")
  dist.init_process_group(backend=backend)
  if backend == "nccl" and os.environ.get("DERL_AMD_NATIVE_COMM", "1") not in ("", "0"):
    try:
      init_native_comm()
    except _lib.NativeError as error:
      # init_native_comm raises on EVERY rank or on none (its ranks agree through MIN all-reduces
      # before any of them enters ncclCommInitRank and before any leaves): without the library's communicator the whole run continues on
      # torch.distributed's own RCCL collectives, update by update
      print(f"derl_amd: native RCCL communicator unavailable ({error}); using torch.distributed", flush=True)
  return world_size()


def init_native_comm():
  """Creates the library's RCCL communicator over the ranks of the default process group: every
  rank checks that it CAN join (dx_comm_available: RCCL loads, no communicator exists, the device
  answers), rank 0 makes the unique id (dx_comm_unique_id), one torch.distributed broadcast hands
  it out, every rank joins (dx_comm_init) on its current device.

  The ranks AGREE before they act, twice.  (1) ncclCommInitRank is itself a collective over all
  ranks: a rank that skipped it would leave the others blocked inside it.  So no rank enters it
  unless a MIN all-reduce of the per-rank readiness flags says that every rank will -- the
  failures a rank can have on its own (RCCL not loadable, a communicator left over, no device)
  are all caught by dx_comm_available, which has no side effects.  (2) What can still fail is the
  id on rank 0 (it then broadcasts an all-zero id -- a real one never is -- and nobody calls
  dx_comm_init) and ncclCommInitRank itself, which RCCL reports on every rank that is in it; a
  second MIN all-reduce of the per-rank outcomes follows dx_comm_init.  Either every rank returns
  with the communicator up, or every rank tears its own down again and raises NativeError, and no
  two ranks disagree on whether the data path reduces through the library or through
  torch.distributed.  (Not covered: a rank that dies or hangs INSIDE ncclCommInitRank -- RCCL's
  own timeout / abort is what ends that.)"""
  global _native  # pylint: disable=global-statement
  if _native:
    return
  if not is_initialized():
    raise RuntimeError("init_native_comm: torch.distributed is not initialised (the unique id "
                       "travels over it)")
  on_gpu = dist.get_backend() == "nccl"
  device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
  problem = None
  try:
    _lib.call("dx_comm_available")
  except Exception as error:  # pylint: disable=broad-except
    # NativeError (the library said no), but also whatever _lib.load() itself can raise on ONE rank -- OSError from
    # ctypes.CDLL, AttributeError for a stale .so without this symbol: the rank still takes part in the agreement
    # below (its peers would block in it otherwise) and every rank falls back together
    problem = error
  ready = torch.tensor([0 if problem else 1], dtype=torch.int32, device=device)
  dist.all_reduce(ready, op=dist.ReduceOp.MIN)
  if int(ready.item()) == 0:
    if problem is not None and not isinstance(problem, _lib.NativeError):
      raise _lib.NativeError(f"the native library cannot be used on this rank: {type(problem).__name__}: {problem}")
    raise _lib.NativeError(str(problem) if problem is not None else
                           "another rank cannot join an RCCL communicator (dx_comm_available failed there)")
  ident = (ctypes.c_ubyte * 128)()
  if rank() == 0:
    try:
      _lib.call("dx_comm_unique_id", ctypes.byref(ident))
    except _lib.NativeError as error:
      problem = error
      ident = (ctypes.c_ubyte * 128)()  # the all-zero id tells the others
  carrier = torch.tensor(list(ident), dtype=torch.uint8, device=device)
  dist.broadcast(carrier, src=0)
  received = carrier.cpu().tolist()
  joined = False
  if not any(received):
    problem = problem or _lib.NativeError("rank 0 could not create the communicator's unique id")
  else:
    ident = (ctypes.c_ubyte * 128)(*received)
    try:
      _lib.call("dx_comm_init", ctypes.byref(ident), rank(), world_size())
      joined = True
    except _lib.NativeError as error:
      problem = error
  verdict = torch.tensor([1 if joined else 0], dtype=torch.int32, device=device)
  dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
  if int(verdict.item()) == 0:
    if joined:
      try:
        _lib.call("dx_comm_destroy")
      except _lib.NativeError:
        pass
    raise _lib.NativeError(str(problem) if problem is not None else
                           "another rank could not join the communicator")
  _native = True


def destroy():
  """Tears down the library's communicator and the process group (end of a run / a test)."""
  global _native  # pylint: disable=global-statement
  if _native:
    _lib.call("dx_comm_destroy")
    _native = False
  if is_initialized():
    dist.destroy_process_group()


def comm_info():
  """(rank, world, all-reduces issued, bytes reduced) of the library's communicator."""
  r, w = ctypes.c_int(0), ctypes.c_int(0)
  n, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
  _lib.call("dx_comm_info", ctypes.byref(r), ctypes.byref(w), ctypes.byref(n), ctypes.byref(b))
  return r.value, w.value, n.value, b.value


def _native_ok(tensor):
  return _native and isinstance(tensor, torch.Tensor) and tensor.is_cuda and tensor.is_contiguous()


class _NativeHandle:
  """What all_reduce_sum_async returns on the native communicator: ``wait()`` orders the current
  stream after every reduction issued so far (dx_allreduce_wait)."""
  def __init__(self, device):
    self.device = device

  def wait(self):
    _lib.call("dx_allreduce_wait", _lib.stream_ptr(self.device))


def all_reduce_sum(tensor):
  """In-place SUM over the ranks, ordered like a kernel on the current stream."""
  if not sharded():
    return tensor
  if _native_ok(tensor) and tensor.dtype == torch.float64:
    _lib.call("dx_allreduce_sum_f64", _lib.ptr(tensor), tensor.numel(), _lib.stream_ptr(tensor.device))
  elif _native_ok(tensor) and tensor.dtype == torch.float32:
    _lib.call("dx_allreduce_grads", _lib.ptr(tensor), tensor.numel(), _lib.stream_ptr(tensor.device))
    _lib.call("dx_allreduce_wait", _lib.stream_ptr(tensor.device))
  elif world_size() > 1:
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
  return tensor


def all_reduce_sum_async(tensor):
  """Starts the all-reduce on the communicator's stream (ordered after the work already on the
  current stream) and returns a handle; ``handle.wait()`` orders the current stream after it."""
  if not sharded():
    return None
  if _native_ok(tensor) and tensor.dtype == torch.float32:
    _lib.call("dx_allreduce_grads", _lib.ptr(tensor), tensor.numel(), _lib.stream_ptr(tensor.device))
    return _NativeHandle(tensor.device)
  if world_size() > 1:
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)
  return None


def all_reduce_mean_grads(flat_grads):
  """Sum over ranks of the per-shard gradients.  Each rank's loss kernel already scales
  by 1/global_batch, so the SUM is the single-process mean gradient."""
  return all_reduce_sum(flat_grads)


def broadcast_(tensor, src=0):
  if not sharded():
    return tensor
  if _native_ok(tensor) and tensor.dtype == torch.float32:
    _lib.call("dx_comm_broadcast_f32", _lib.ptr(tensor), tensor.numel(), int(src),
              _lib.stream_ptr(tensor.device))
  elif world_size() > 1:
    dist.broadcast(tensor, src=src)
  return tensor


def barrier():
  if world_size() > 1:
    if dist.get_backend() == "nccl":  # name the device: no guess from the rank, no warning
      dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
      dist.barrier()
