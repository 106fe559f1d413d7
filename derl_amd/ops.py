"""Thin torch-tensor front ends of the C-ABI ops (shape / dtype checks, then one call).

torch supplies device memory and the stream only; all arithmetic happens in
libderl_amd.so.  Every function raises if its inputs are not on a HIP device.
"""
import math

import torch

from . import _lib


def _dev(t, name, dtype=None):
  if not isinstance(t, torch.Tensor) or not t.is_cuda:
    raise ValueError(f"{name} must be a tensor on the GPU (derl_amd has no CPU path)")
  if dtype is not None and t.dtype != dtype:
    raise ValueError(f"{name} must have dtype {dtype}, got {t.dtype}")
  if not t.is_contiguous():
    raise ValueError(f"{name} must be contiguous")
  return t


def gae(rewards, resets, values, last_values, gamma, lambda_, out_advantages=None,
        out_value_targets=None):
  """Device GAE over time-major (T, N) arrays (trajectory_transforms.py:45-65).

  rewards (T,N) f32, resets (T,N) bool/uint8, values (T,N) f32, last_values (N,) f32.
  Returns (advantages (T,N), value_targets (T,N))."""
  _dev(rewards, "rewards", torch.float32)
  _dev(values, "values", torch.float32)
  _dev(last_values, "last_values", torch.float32)
  _dev(resets, "resets")
  if resets.dtype not in (torch.bool, torch.uint8):
    raise ValueError(f"resets must be bool or uint8, got {resets.dtype}")
  if rewards.ndim != 2:
    raise ValueError(f"rewards must be (T, N), got {tuple(rewards.shape)}")
  T, N = rewards.shape
  if tuple(values.shape) != (T, N) or tuple(resets.shape) != (T, N):
    raise ValueError("rewards, resets and values must share the shape (T, N): "
                     f"{tuple(rewards.shape)}, {tuple(resets.shape)}, {tuple(values.shape)}")
  if last_values.numel() != N:
    raise ValueError(f"last_values must have {N} elements, got {tuple(last_values.shape)}")
  adv = out_advantages if out_advantages is not None else torch.empty_like(values)
  vt = out_value_targets if out_value_targets is not None else torch.empty_like(values)
  _dev(adv, "out_advantages", torch.float32)
  _dev(vt, "out_value_targets", torch.float32)
  _lib.call("dx_gae_f32", _lib.ptr(rewards), _lib.ptr(resets), _lib.ptr(values),
            _lib.ptr(last_values), T, N, float(gamma), float(lambda_), _lib.ptr(adv),
            _lib.ptr(vt), _lib.stream_ptr(rewards.device))
  return adv, vt


def adv_normalize(advantages, epsilon=1e-8, out=None, stats=None, stats_ready=False):
  """(a - mean) / (std + eps) on the device (trajectory_transforms.py:89-92).  ``stats`` is a
  float64[3] device tensor {sum, sumsq, count}; pass stats_ready=True to reuse reduced stats."""
  _dev(advantages, "advantages", torch.float32)
  out = torch.empty_like(advantages) if out is None else _dev(out, "out", torch.float32)
  if stats is None:
    stats = torch.empty(3, dtype=torch.float64, device=advantages.device)
  _lib.call("dx_adv_normalize_f32", _lib.ptr(advantages), _lib.ptr(out), advantages.numel(),
            float(epsilon), _lib.ptr(stats), int(stats_ready), _lib.stream_ptr(advantages.device))
  return out


def adv_stats(advantages, stats=None):
  _dev(advantages, "advantages", torch.float32)
  if stats is None:
    stats = torch.empty(3, dtype=torch.float64, device=advantages.device)
  _lib.call("dx_adv_stats_f32", _lib.ptr(advantages), advantages.numel(), _lib.ptr(stats),
            _lib.stream_ptr(advantages.device))
  return stats


def adv_stats_segments(advantages, index, seglen, stats=None):
  """{sum, sumsq, count} of every ``seglen``-long slice of ``advantages[index]`` (the minibatches
  of one epoch; ``index`` int32 or None) -> float64 (ceil(n / seglen), 3)."""
  _dev(advantages, "advantages", torch.float32)
  n = advantages.numel() if index is None else index.numel()
  nseg = -(-n // seglen)
  if index is not None:
    _dev(index, "index", torch.int32)
  if stats is None:
    stats = torch.empty((nseg, 3), dtype=torch.float64, device=advantages.device)
  _lib.call("dx_adv_stats_segments_f32", _lib.ptr(advantages), _lib.ptr(index) if index is not None else None,
            n, seglen, _lib.ptr(stats), _lib.stream_ptr(advantages.device))
  return stats


NORM_PARTIALS = 256


def grad_sumsq(grads, partials=None):
  _dev(grads, "grads", torch.float32)
  if partials is None:
    partials = torch.empty(NORM_PARTIALS, dtype=torch.float64, device=grads.device)
  _lib.call("dx_grad_sumsq_f32", _lib.ptr(grads), grads.numel(), _lib.ptr(partials),
            partials.numel(), _lib.stream_ptr(grads.device))
  return partials


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, partials, max_norm, lr, step,
                   beta1=0.9, beta2=0.999, eps=1e-8, norm_out=None):
  for name, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg),
                  ("exp_avg_sq", exp_avg_sq)):
    _dev(t, name, torch.float32)
  _lib.call("dx_clip_adam_step_f32", _lib.ptr(params), _lib.ptr(grads), _lib.ptr(exp_avg),
            _lib.ptr(exp_avg_sq), params.numel(), _lib.ptr(partials),
            0 if partials is None else partials.numel(),
            float(max_norm) if max_norm is not None else 0.0, float(lr), float(beta1),
            float(beta2), float(eps), int(step), _lib.ptr(norm_out),
            _lib.stream_ptr(params.device))


def clip_rmsprop_step(params, grads, square_avg, partials, max_norm, lr, alpha=0.99, eps=1e-8,
                      norm_out=None):
  for name, t in (("params", params), ("grads", grads), ("square_avg", square_avg)):
    _dev(t, name, torch.float32)
  _lib.call("dx_clip_rmsprop_step_f32", _lib.ptr(params), _lib.ptr(grads), _lib.ptr(square_avg),
            params.numel(), _lib.ptr(partials), 0 if partials is None else partials.numel(),
            float(max_norm) if max_norm is not None else 0.0, float(lr), float(alpha), float(eps),
            _lib.ptr(norm_out), _lib.stream_ptr(params.device))


def gather_rows(src, idx, out=None):
  """out[i] = src[idx[i]] along dim 0 (idx int32 on the device)."""
  _dev(src, "src")
  _dev(idx, "idx", torch.int32)
  n = idx.numel()
  out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device) if out is None else out
  row_bytes = src.element_size()
  for d in src.shape[1:]:
    row_bytes *= d
  _lib.call("dx_gather_rows", _lib.ptr(src), _lib.ptr(idx), _lib.ptr(out), n, row_bytes,
            _lib.stream_ptr(src.device))
  return out


def gather_rows_multi(sources, idx):
  """[src[idx] for src in sources] along dim 0 with one launch per 16 arrays."""
  _dev(idx, "idx", torch.int32)
  n = idx.numel()
  outs = []
  for start in range(0, len(sources), 16):
    group = sources[start:start + 16]
    k = len(group)
    src_p, dst_p, rb = (_lib.c_void_p * k)(), (_lib.c_void_p * k)(), (_lib.c_longlong * k)()
    for i, src in enumerate(group):
      _dev(src, "src")
      out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
      outs.append(out)
      src_p[i], dst_p[i] = src.data_ptr(), out.data_ptr()
      rb[i] = src.element_size() * math.prod(src.shape[1:])
    _lib.call("dx_gather_rows_multi", src_p, dst_p, rb, k, _lib.ptr(idx), n,
              _lib.stream_ptr(idx.device))
  return outs


def categorical_act(head_out, num_actions, uniforms=None, seed=0, counter=0, out=None):
  """Sample / log_prob / value from the padded head output (B, 32) (policies.py:61-80).
  ``out`` = (actions int64 (B,), log_prob f32 (B,), values f32 (B,)) to write in place."""
  _dev(head_out, "head_out", torch.float32)
  B = head_out.shape[0]
  dev = head_out.device
  if out is not None:
    actions, log_prob, values = out
    _dev(actions, "actions", torch.int64)
    _dev(log_prob, "log_prob", torch.float32)
    _dev(values, "values", torch.float32)
    if actions.numel() != B or log_prob.numel() != B or values.numel() != B:
      raise ValueError("categorical_act: output buffers must hold B elements")
  else:
    actions = torch.empty(B, dtype=torch.int64, device=dev)
    log_prob = torch.empty(B, dtype=torch.float32, device=dev)
    values = torch.empty(B, dtype=torch.float32, device=dev)
  if uniforms is not None:
    _dev(uniforms, "uniforms", torch.float32)
  _lib.call("dx_categorical_act_f32", _lib.ptr(head_out), B, int(num_actions), _lib.ptr(uniforms),
            int(seed), int(counter), _lib.ptr(actions), _lib.ptr(log_prob), _lib.ptr(values),
            _lib.stream_ptr(dev))
  return actions, log_prob, values


def categorical_loss(head_out, actions, old_log_prob, advantages, old_values, value_targets,
                     num_actions, mode, cliprange, value_loss_coef, entropy_coef, dhead_out,
                     global_batch=0, partials=None, loss_out=None):
  """Fused PPO (mode 0) / A2C (mode 1) loss + gradient w.r.t. the head output."""
  _dev(head_out, "head_out", torch.float32)
  B = head_out.shape[0]
  dev = head_out.device
  need = 8 * ((B + 7) // 8)
  if partials is None or partials.numel() < need:
    partials = torch.empty(need, dtype=torch.float64, device=dev)
  if loss_out is None:
    loss_out = torch.empty(8, dtype=torch.float32, device=dev)
  _lib.call("dx_categorical_loss_f32", _lib.ptr(head_out), _lib.ptr(actions),
            _lib.ptr(old_log_prob), _lib.ptr(advantages), _lib.ptr(old_values),
            _lib.ptr(value_targets), B, int(num_actions), int(mode),
            -1.0 if cliprange is None else float(cliprange), float(value_loss_coef),
            float(entropy_coef), int(global_batch), _lib.ptr(dhead_out), _lib.ptr(partials),
            partials.numel(), _lib.ptr(loss_out), _lib.stream_ptr(dev))
  return loss_out


def normal_act(head_out, logstd, normals=None, seed=0, counter=0, out=None):
  """Diagonal-Gaussian sample / log_prob / value from the padded head (policies.py:66,76-77)."""
  _dev(head_out, "head_out", torch.float32)
  _dev(logstd, "logstd", torch.float32)
  B, P = head_out.shape[0], logstd.numel()
  dev = head_out.device
  if out is not None:
    actions, log_prob, values = out
  else:
    actions = torch.empty(B, P, dtype=torch.float32, device=dev)
    log_prob = torch.empty(B, dtype=torch.float32, device=dev)
    values = torch.empty(B, dtype=torch.float32, device=dev)
  _lib.call("dx_normal_act_f32", _lib.ptr(head_out), _lib.ptr(logstd), B, P, _lib.ptr(normals),
            int(seed), int(counter), _lib.ptr(actions), _lib.ptr(log_prob), _lib.ptr(values),
            _lib.stream_ptr(dev))
  return actions, log_prob, values


def normal_loss(head_out, logstd, actions, old_log_prob, advantages, old_values, value_targets,
                mode, cliprange, value_loss_coef, entropy_coef, dhead_out, dlogstd_out,
                global_batch=0, partials=None, loss_out=None):
  """Fused PPO / A2C loss + gradients for the diagonal-Gaussian head."""
  _dev(head_out, "head_out", torch.float32)
  _dev(actions, "actions", torch.float32)
  B, P = head_out.shape[0], logstd.numel()
  dev = head_out.device
  need = 40 * ((B + 255) // 256)
  if partials is None or partials.numel() < need:
    partials = torch.empty(need, dtype=torch.float64, device=dev)
  if loss_out is None:
    loss_out = torch.empty(8, dtype=torch.float32, device=dev)
  _lib.call("dx_normal_loss_f32", _lib.ptr(head_out), _lib.ptr(logstd), _lib.ptr(actions),
            _lib.ptr(old_log_prob), _lib.ptr(advantages), _lib.ptr(old_values),
            _lib.ptr(value_targets), B, P, int(mode),
            -1.0 if cliprange is None else float(cliprange), float(value_loss_coef),
            float(entropy_coef), int(global_batch), _lib.ptr(dhead_out), _lib.ptr(dlogstd_out),
            _lib.ptr(partials), partials.numel(), _lib.ptr(loss_out), _lib.stream_ptr(dev))
  return loss_out
