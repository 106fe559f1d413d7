"""Device-resident ``Normalize`` wrapper (derl/env/mujoco_wrappers.py:64-124): observation
and return normalisation of a batched env with running statistics, one native call per env step
(``dx_normalize_step_f32``) between the env and the policy forward -- the observations never
leave HBM.  Wraps any env with the device interface (``SyntheticMuJoCoEnv``, ``HostEnvBridge``).

Statistics are float64 on the device like the reference's NumPy state; ``save_wrapper`` /
``restore_wrapper`` use the reference's file names and keys (``<name>-obs-rmv.npz`` /
``<name>-ret-rmv.npz`` with ``mean``, ``var``, ``count``).
"""
import numpy as np
import torch

from .. import _lib


class Normalize:
  """A vectorized wrapper that normalizes the observations and returns of an env."""
  # pylint: disable=too-many-arguments
  def __init__(self, env, obs=True, ret=True, clipobs=10., cliprew=10., gamma=0.99, eps=1e-8,
               stat_eps=1e-4):
    device = getattr(env, "device", None)
    if device is None or getattr(env.unwrapped, "nenvs", None) is None:
      raise TypeError("the device Normalize wraps a batched device-resident env "
                      "(wrap host envs in HostEnvBridge first)")
    self.env = env
    self.device = torch.device(device)
    self.nenvs = env.unwrapped.nenvs
    self.observation_space, self.action_space = env.observation_space, env.action_space
    shape = tuple(self.observation_space.shape)
    if len(shape) != 1:
      raise ValueError(f"Normalize expects vector observations, got shape {shape}")
    self.dim = shape[0]
    f64 = dict(dtype=torch.float64, device=self.device)
    self.obs_stats = None
    if obs:  # {mean[D], var[D], count}
      self.obs_stats = torch.cat([torch.zeros(self.dim, **f64), torch.ones(self.dim, **f64),
                                  torch.full((1,), stat_eps, **f64)])
    self.ret_stats = torch.tensor([0., 1., stat_eps], **f64) if ret else None
    self.ret = torch.zeros(self.nenvs, **f64)
    self.clipob, self.cliprew, self.gamma, self.eps = clipobs, cliprew, gamma, eps
    self._raw_obs = torch.empty((self.nenvs, self.dim), dtype=torch.float32, device=self.device)
    self._raw_rew = torch.empty(self.nenvs, dtype=torch.float32, device=self.device)
    self._workspace = torch.empty(256 * self.dim, **f64)  # partial moments per row block

  rollout_done = None  # the runner's per-rollout hook stops here: inner statistics are per step

  @property
  def unwrapped(self):
    return self.env.unwrapped

  def __getattr__(self, name):
    if name in ("env", "__setstate__"):
      raise AttributeError(name)
    return getattr(self.env, name)

  def _call(self, raw_obs, rewards, resets, out, rewards_out, update=True):
    shape = (self.nenvs, self.dim)
    if out is None:
      out = torch.empty(shape, dtype=torch.float32, device=self.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
      raise ValueError(f"out must be a contiguous float32 tensor of shape {shape}")
    _lib.call("dx_normalize_step_f32", _lib.ptr(raw_obs), self.nenvs, self.dim, _lib.ptr(rewards),
              _lib.ptr(resets), _lib.ptr(self.obs_stats), _lib.ptr(self.ret_stats), _lib.ptr(self.ret),
              _lib.ptr(self._workspace), self._workspace.numel(), float(self.clipob), float(self.cliprew), float(self.gamma), float(self.eps),
              int(update), _lib.ptr(out), _lib.ptr(rewards_out), _lib.stream_ptr(self.device))
    return out

  def observation(self, obs, out=None, update=True):
    """Normalises a batch of raw observations (updating the statistics like the reference)."""
    obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
    return self._call(obs, None, None, out, None, update)

  def reset(self, out=None):
    self.ret.zero_()
    raw = self.env.reset(out=self._raw_obs)
    return self._call(raw, None, None, out, None)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    raw, rewards, resets, infos = self.env.step(actions, out=self._raw_obs, rewards_out=self._raw_rew,
                                                resets_out=resets_out)
    inner_stats = getattr(self.env, "rollout_done", None)
    if inner_stats is not None:  # reward summaries see the RAW rewards (mujoco_wrap order)
      inner_stats(rewards.reshape(1, -1), resets.reshape(1, -1))
    if rewards_out is None:
      rewards_out = torch.empty(self.nenvs, dtype=torch.float32, device=self.device)
    resets_u8 = resets.view(torch.uint8) if resets.dtype == torch.bool else resets
    out = self._call(raw, rewards.contiguous(), resets_u8.contiguous(), out, rewards_out)
    return out, rewards_out, resets, infos

  # ---- persistence with the reference's file layout (:81-97) --------------------------
  @staticmethod
  def _strip(filename):
    return filename[:-3] if filename.endswith("npz") else filename

  def save_wrapper(self, filename):
    filename = self._strip(filename)
    if self.obs_stats is not None:
      host = self.obs_stats.cpu().numpy()
      np.savez(f"{filename}-obs-rmv", mean=host[:self.dim], var=host[self.dim:2 * self.dim],
               count=host[2 * self.dim])
    if self.ret_stats is not None:
      host = self.ret_stats.cpu().numpy()
      np.savez(f"{filename}-ret-rmv", mean=host[0], var=host[1], count=host[2])

  def restore_wrapper(self, filename):
    if self.obs_stats is not None:
      data = np.load(f"{filename}-obs-rmv.npz")
      host = np.concatenate([np.asarray(data["mean"], np.float64).reshape(-1),
                             np.asarray(data["var"], np.float64).reshape(-1),
                             np.asarray(data["count"], np.float64).reshape(1)])
      self.obs_stats.copy_(torch.from_numpy(host))
    if self.ret_stats is not None:
      data = np.load(f"{filename}-ret-rmv.npz")
      host = np.array([data["mean"], data["var"], data["count"]], np.float64)
      self.ret_stats.copy_(torch.from_numpy(host))
