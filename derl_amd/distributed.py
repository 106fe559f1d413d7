"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" for CPU rehearsals).  The hot path has exactly one real exchange
step -- the gradient all-reduce per optimizer step -- plus a 3-double
all-reduce for the per-minibatch advantage statistics."""
import os

import torch
import torch.distributed as dist


def is_initialized():
  return dist.is_available() and dist.is_initialized()


def world_size():
  return dist.get_world_size() if is_initialized() else 1


def rank():
  return dist.get_rank() if is_initialized() else 0


def init_from_env(backend=None):
  """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* (the
  torch.distributed.run contract) if WORLD_SIZE > 1; binds this rank to its GPU."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world <= 1 or is_initialized():
    return world_size()
  local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
  if backend is None:
    # DERL_AMD_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("DERL_AMD_DIST_BACKEND") or (
        "nccl" if torch.cuda.is_available() else "gloo")
  if backend == "nccl":
    torch.cuda.set_device(local_rank)
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  dist.init_process_group(backend=backend)
  return world_size()


def all_reduce_sum(tensor):
  if world_size() > 1:
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
  return tensor


def all_reduce_sum_async(tensor):
  """Starts the all-reduce on the communicator's stream (ordered after the work already on the
  current stream) and returns the handle; ``handle.wait()`` orders the current stream after it."""
  if world_size() > 1:
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)
  return None


def all_reduce_mean_grads(flat_grads):
  """Sum over ranks of the per-shard gradients.  Each rank's loss kernel already scales
  by 1/global_batch, so the SUM is the single-process mean gradient."""
  return all_reduce_sum(flat_grads)


def broadcast_(tensor, src=0):
  if world_size() > 1:
    dist.broadcast(tensor, src=src)
  return tensor


def barrier():
  if world_size() > 1:
    if dist.get_backend() == "nccl":  # name the device: no guess from the rank, no warning
      dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
      dist.barrier()
