"""Thin torch-tensor front ends of the C-ABI ops (shape / dtype checks, then one call).

torch supplies device memory and the stream only; all arithmetic happens in
libderl_amd.so.  Every function raises if its inputs are not on a HIP device.
"""
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
