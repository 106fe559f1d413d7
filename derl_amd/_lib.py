"""ctypes binding of the C-ABI in include/derl_amd.h (libderl_amd.so).

Fails loudly: there is no CPU or PyTorch fallback for any hot-path op.  If the shared
library is missing or a call returns non-zero, a ``NativeError`` carrying
``dx_last_error()`` is raised.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libderl_amd.so")
# DERL_AMD_LIBRARY=diag: the -DDX_DIAG=1 flavour (derl_amd/build.py), for tools/ only -- it carries
# the in-kernel stamps and bisecting switches that are compiled out of the product library
if os.environ.get("DERL_AMD_LIBRARY", "") == "diag":
  LIB_PATH = os.path.join(_PKG, "libderl_amd_diag.so")
elif os.environ.get("DERL_AMD_LIBRARY", "").endswith(".so"):  # another build of the library (tools/: A/B on one box)
  LIB_PATH = os.path.join(_PKG, os.path.basename(os.environ["DERL_AMD_LIBRARY"]))
ABI_VERSION = 6

c_int, c_float, c_void_p, c_char_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p
c_size_t, c_int64, c_uint64, c_double = ctypes.c_size_t, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
c_longlong = ctypes.c_longlong
P = c_void_p

# name -> argtypes; every function returns int (0 = OK) unless listed in _RESTYPES
SIGNATURES = {
    "dx_abi_version": [],
    "dx_last_error": [],
    "dx_reload_env": [],
    "dx_launch_count": [],
    "dx_device_info": [c_int, c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "dx_gae_f32": [P, P, P, P, c_int, c_int, c_float, c_float, P, P, P],
    "dx_adv_stats_f32": [P, c_longlong, P, P],
    "dx_adv_stats_segments_f32": [P, P, c_longlong, c_longlong, P, P],
    "dx_adv_normalize_f32": [P, P, c_longlong, c_float, P, c_int, P],
    "dx_grad_sumsq_f32": [P, c_longlong, P, c_int, P],
    "dx_clip_adam_step_f32": [P, P, P, P, c_longlong, P, c_int, c_double, c_double, c_double,
                              c_double, c_double, c_longlong, P, P],
    "dx_clip_rmsprop_step_f32": [P, P, P, c_longlong, P, c_int, c_double, c_double, c_double,
                                 c_double, P, P],
    "dx_host_compose_permutations": [P, P, c_longlong, c_int, c_int, P],
    "dx_gather_rows": [P, P, P, c_longlong, c_longlong, P],
    "dx_gather_rows_multi": [P, P, P, c_int, P, c_longlong, P],
    "dx_reward_summary_f32": [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P, P],
    "dx_normalize_step_f32": [P, c_int, c_int, P, P, P, P, P, P, c_longlong, c_float, c_float,
                              c_double, c_double, c_int, P, P, P],
    "dx_frame_max_u8": [P, P, P, P, P, c_int, c_longlong, P],
    "dx_frame_queue_u8": [P, P, P, P, P, c_int, c_longlong, c_int, c_int, c_int, P],
    "dx_gray_resize_u8": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "dx_categorical_act_f32": [P, c_int, c_int, P, c_uint64, c_uint64, P, P, P, P],
    "dx_categorical_loss_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float,
                                c_float, c_longlong, P, P, c_int, P, P],
    "dx_synth_atari_step": [P, c_longlong, P, P, c_int, c_uint64, c_uint64, c_float, c_float, P],
    "dx_synth_mujoco_step": [P, P, P, c_int, c_int, c_uint64, c_uint64, c_float, P],
    "dx_mlp_init": [P],
    "dx_mlp_pack": [P, P],
    "dx_mlp_forward": [P, P, c_int, P],
    "dx_mlp_backward": [P, c_int, P],
    "dx_mlp_rollout_synth": [P, P, c_int, c_int, P, P, P, P, P, c_uint64, c_uint64, c_uint64, c_uint64, c_float, P],
    "dx_mlp_ppo_epoch": [P, P, P],
    "dx_mlp_persist_plan": [P, c_int, c_longlong, P, P],
    "dx_mlp_last_route": [],
    "dx_normal_act_f32": [P, P, c_int, c_int, P, c_uint64, c_uint64, P, P, P, P],
    "dx_normal_loss_f32": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float,
                           c_longlong, P, P, P, c_int, P, P],
    "dx_cnn_init": [P],
    "dx_cnn_pack": [P, P],
    "dx_cnn_forward": [P, P, c_int, P, c_int, P],
    "dx_cnn_backward": [P, P, c_int, P, c_int, P],
    "dx_cnn_forward_trunk": [P, P, c_int, P, c_int, P],
    "dx_cnn_heads_loss_f32": [P, P, P, P, P, P, P, c_float, P, c_int, c_int, c_float, c_float, c_float,
                              c_longlong, P, c_int, P, P, P],
    "dx_cnn_backward_part": [P, P, c_int, P, c_int, c_int, P],
    "dx_cnn_stage": [P, c_int, P, c_int, P, c_int, P],
    "dx_cnn_last_route": [c_int],
    "dx_cnn_tail_factored": [P],
    "dx_cnn_fused_heads": [P],
    "dx_cnn_tail_fused": [P],
    "dx_cnn_act": [P, P, c_int, c_int, P, c_uint64, c_uint64, P, P, P, P],
    "dx_cnn_rollout_synth": [P, P, c_int, c_int, P, P, P, P, P, c_uint64, c_uint64, c_uint64,
                             c_uint64, c_float, c_float, P],
    "dx_cnn_ppo_epoch": [P, P, P],
    "dx_comm_available": [],
    "dx_comm_unique_id": [P],
    "dx_comm_init": [P, c_int, c_int],
    "dx_comm_info": [P, P, P, P],
    "dx_comm_destroy": [],
    "dx_allreduce_grads": [P, c_longlong, P],
    "dx_allreduce_wait": [P],
    "dx_allreduce_sum_f64": [P, c_longlong, P],
    "dx_comm_broadcast_f32": [P, c_longlong, c_int, P],
}


class CnnCtx(ctypes.Structure):
  """Mirror of ``dx_cnn_ctx`` (include/derl_amd.h); dx_cnn_init checks the size."""
  _fields_ = (
      [(n, c_int) for n in ("struct_bytes", "in_h", "in_w", "in_c", "num_actions", "max_batch",
                            "h0", "w0", "h1", "w1", "h2", "w2", "flat", "reserved0")]
      + [("off_w", ctypes.c_longlong * 6), ("off_b", ctypes.c_longlong * 6),
         ("param_count", ctypes.c_longlong)]
      + [(n, ctypes.c_longlong) for n in ("pk_c0f", "pk_c1f", "pk_c2f", "pk_fcf", "pk_hdf", "pk_hdb")]
      + [("pk_c1d", ctypes.c_longlong * 4)]
      + [(n, ctypes.c_longlong) for n in ("pk_c2d", "pk_fcd", "pk_hdd", "packed_count", "slab_count",
                                          "y0_count", "y1_count", "y2_count", "hid_count", "head_count",
                                          "hid_slab_count", "pb_c1f", "pb_c2f", "pb_fcf", "pb_c1d",
                                          "pb_c2d", "pb_fcd", "pb_c0f", "pk_wc", "pk_beff", "pk_wcs", "ps_c1f",
                                          "ps_c2f", "ps_wc", "ps_c1d", "ps_c2d")]
      + [(n, c_void_p) for n in ("params", "grads", "packed", "y0", "y1", "y2", "hid", "head",
                                 "dy0", "dy1", "dy2", "dhid", "dhead", "slabs", "hid_slabs")])
_RESTYPES = {"dx_last_error": c_char_p, "dx_launch_count": c_longlong, "dx_cnn_last_route": c_char_p}

_lib = None


class NativeError(RuntimeError):
  """A C-ABI call failed (carries dx_last_error(); ``status`` = the call's return code, one of the
  DX_E* values of include/derl_amd.h, or None when no call was made)."""
  status = None


# return codes of include/derl_amd.h
DX_EINVAL, DX_EHIP, DX_ENOSUP, DX_EWS, DX_ETIMEOUT = -1, -2, -3, -4, -5


def load():
  """Loads libderl_amd.so once.  ``import torch`` must have happened first so that the
  HIP runtime torch ships (soname libamdhip64.so.7) is the one the library binds to."""
  global _lib
  if _lib is not None:
    return _lib
  import torch  # noqa: F401  (loads libamdhip64 before our DT_NEEDED is resolved)
  if not os.path.exists(LIB_PATH):
    raise NativeError(
        f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  derl_amd has no CPU fallback.")
  lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
  for name, argtypes in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the .so is stale
    fn.argtypes = argtypes
    fn.restype = _RESTYPES.get(name, c_int)
  if lib.dx_abi_version() != ABI_VERSION:
    raise NativeError(f"libderl_amd.so ABI {lib.dx_abi_version()} != binding {ABI_VERSION}")
  _lib = lib
  return lib


def last_error():
  msg = load().dx_last_error()
  return msg.decode() if msg else ""


def check(status, what):
  if status != 0:
    error = NativeError(f"{what} failed with status {status}: {last_error()}")
    error.status = status
    raise error


def call(name, *args):
  """Calls a C-ABI function and raises NativeError on a non-zero status."""
  check(getattr(load(), name)(*args), name)


def ptr(tensor):
  """Device (or host) address of a torch tensor as a void*; None -> NULL."""
  if tensor is None:
    return None
  return c_void_p(tensor.data_ptr())


_raw_stream = None


def stream_ptr(device=None):
  """The current torch HIP stream as void* (torch is the stream/memory plumbing)."""
  global _raw_stream  # pylint: disable=global-statement
  import torch
  if _raw_stream is None:
    # the raw getter is ~20x cheaper than building a torch.cuda.Stream object per call
    _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)  # pylint: disable=protected-access
  index = getattr(device, "index", device)
  if _raw_stream:
    if index is None:  # torch.device("cuda"): the current device
      index = torch.cuda.current_device()
    if isinstance(index, int):
      return c_void_p(_raw_stream(index))
  return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class MlpEpoch(ctypes.Structure):
  """Mirror of ``dx_mlp_epoch`` (include/derl_amd.h); dx_mlp_ppo_epoch checks the size."""
  _fields_ = [
      ("struct_bytes", c_int), ("mbsize", c_int), ("samples", c_longlong),
      ("obs", c_void_p), ("actions", c_void_p), ("action_is_f32", c_int), ("mode", c_int),
      ("old_log_prob", c_void_p), ("advantages", c_void_p), ("old_values", c_void_p),
      ("value_targets", c_void_p), ("normalize", c_int), ("norm_eps", c_float),
      ("cliprange", c_float), ("value_loss_coef", c_float), ("entropy_coef", c_float),
      ("global_batch", c_longlong), ("adv_normalized", c_void_p), ("stats", c_void_p),
      ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("sumsq_partials", c_void_p),
      ("npartials", c_int), ("loss_partials_capacity", c_int), ("loss_partials", c_void_p),
      ("max_grad_norm", c_double), ("lr", c_double), ("beta1", c_double), ("beta2", c_double),
      ("adam_eps", c_double), ("first_step", c_longlong), ("grad_norm_out", c_void_p),
      ("loss_out", c_void_p), ("grad_norm_stride", c_int), ("persistent", c_int),
      ("workspace", c_void_p), ("workspace_bytes", c_longlong), ("stats_all", c_void_p),
      ("status_host", c_void_p)]


class CnnEpoch(ctypes.Structure):
  """Mirror of ``dx_cnn_epoch`` (include/derl_amd.h); dx_cnn_ppo_epoch checks the size."""
  _fields_ = [
      ("struct_bytes", c_int), ("mbsize", c_int), ("samples", c_longlong),
      ("obs", c_void_p), ("obs_is_u8", c_int), ("mode", c_int), ("index", c_void_p),
      ("actions", c_void_p), ("old_log_prob", c_void_p), ("advantages", c_void_p),
      ("old_values", c_void_p), ("value_targets", c_void_p), ("normalize", c_int),
      ("norm_eps", c_float), ("stats_ready", c_void_p), ("stats", c_void_p),
      ("adv_normalized", c_void_p), ("cliprange", c_float), ("value_loss_coef", c_float),
      ("entropy_coef", c_float), ("world", c_int), ("allreduce", c_int), ("optimizer", c_int),
      ("npartials", c_int), ("state0", c_void_p), ("state1", c_void_p),
      ("sumsq_partials", c_void_p), ("loss_partials", c_void_p),
      ("loss_partials_capacity", c_int), ("grad_norm_stride", c_int),
      ("mirrors_current", c_int), ("more_epochs", c_int), ("loss_counter", c_void_p),
      ("max_grad_norm", c_double), ("lr", c_double), ("beta1", c_double), ("beta2", c_double),
      ("opt_eps", c_double), ("first_step", c_longlong), ("grad_norm_out", c_void_p),
      ("loss_out", c_void_p)]


class MlpCtx(ctypes.Structure):
  """Mirror of ``dx_mlp_ctx`` (include/derl_amd.h); dx_mlp_init checks the size."""
  _fields_ = (
      [(n, c_int) for n in ("struct_bytes", "obs_dim", "policy_out", "has_logstd", "max_batch",
                            "obs_pad", "reserved0", "reserved1")]
      + [("off_logstd", ctypes.c_longlong), ("off_w", ctypes.c_longlong * 6),
         ("off_b", ctypes.c_longlong * 6), ("param_count", ctypes.c_longlong),
         ("pk_f0", ctypes.c_longlong * 2), ("pk_d2", ctypes.c_longlong * 2),
         ("pk_d1", ctypes.c_longlong * 2)]
      + [(n, ctypes.c_longlong) for n in ("packed_count", "slab_per_net", "slab_count", "x_count",
                                          "h_count", "head_count")]
      + [(n, c_void_p) for n in ("params", "grads", "packed", "xpad")]
      + [("h1", c_void_p * 2), ("h2", c_void_p * 2)]
      + [(n, c_void_p) for n in ("head", "dhead", "da", "db", "slabs")])
