"""Device-resident synthetic batched environments for the headline metric (SURVEY.md 8d:
ALE / MuJoCo are absent, the metric is quoted on synthetic frames).

They follow derl's batched-env contract (derl/env/env_batch.py:35-134): ``nenvs``,
``observation_space`` / ``action_space`` of ONE env, ``reset() -> obs``,
``step(actions) -> (obs, rewards, resets, infos)`` with auto-reset.  Observations, rewards
and resets are produced on the GPU from a counter-based generator and never visit the
host; ``step(..., out=slot)`` writes the next frame batch straight into a rollout-buffer
slot."""
import numpy as np
import torch

from .. import _lib
from .spaces import Box, Discrete


class SyntheticAtariEnv:
  """uint8 (nenvs, 84, 84, 4) frames i.i.d. uniform 0..255; rewards sign(n)*[u<0.1] in
  {-1,0,1}; resets Bernoulli(0.01) (derl/env/make_env.py:132-133 obs contract,
  derl/env/atari_wrappers.py:189-192 reward contract)."""

  host_rng_free = True  # never touches the global np.random stream (see IterateWithMinibatches)

  def __init__(self, nenvs, num_actions=4, seed=0, obs_shape=(84, 84, 4), p_reset=0.01,
               device="cuda", rank=0):
    self.nenvs = int(nenvs)
    self.unwrapped = self
    self.device = torch.device(device)
    self.observation_space = Box(0, 255, obs_shape, np.uint8)
    self.action_space = Discrete(num_actions)
    self.p_reset = p_reset
    self.seed = int(seed) * 1000003 + int(rank)
    self.counter = 0

  def _generate(self, out, rewards=None, resets=None):
    shape = (self.nenvs,) + self.observation_space.shape
    if out is None:
      out = torch.empty(shape, dtype=torch.uint8, device=self.device)
    elif tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_contiguous():
      raise ValueError(f"out must be a contiguous uint8 tensor of shape {shape}")
    _lib.call("dx_synth_atari_step", _lib.ptr(out), out.numel(), _lib.ptr(rewards),
              _lib.ptr(resets), self.nenvs, self.seed, self.counter, 0.1, float(self.p_reset),
              _lib.stream_ptr(self.device))
    self.counter += 1
    return out

  def reset(self, out=None):
    return self._generate(out)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    del actions  # the synthetic dynamics ignore the action
    rewards = rewards_out if rewards_out is not None else torch.empty(
        self.nenvs, dtype=torch.float32, device=self.device)
    resets = resets_out if resets_out is not None else torch.empty(
        self.nenvs, dtype=torch.bool, device=self.device)
    obs = self._generate(out, rewards, resets)
    return obs, rewards, resets, None


class SyntheticMuJoCoEnv:
  """float32 (nenvs, obs_dim) N(0,1) observations clipped to +-10 (the range
  derl/env/mujoco_wrappers.py:64-124 Normalize produces), rewards N(0,1), resets
  Bernoulli(0.001) -- every value a hash of (seed, step counter, env, component)
  (``dx_synth_mujoco_step``), so the native whole-horizon rollout (``dx_mlp_rollout_synth``)
  draws the same numbers as this per-step loop."""

  host_rng_free = True  # never touches the global np.random stream (see IterateWithMinibatches)

  def __init__(self, nenvs, obs_dim=17, act_dim=6, seed=0, p_reset=0.001, device="cuda", rank=0):
    if not 1 <= int(obs_dim) <= 64:
      raise ValueError("obs_dim must be in 1 .. 64")
    self.nenvs = int(nenvs)
    self.unwrapped = self
    self.device = torch.device(device)
    self.observation_space = Box(-10., 10., (obs_dim,), np.float32)
    self.action_space = Box(-1., 1., (act_dim,), np.float32)
    self.p_reset = p_reset
    self.seed = int(seed) * 1000003 + int(rank)
    self.counter = 0

  def _generate(self, out, rewards=None, resets=None):
    shape = (self.nenvs,) + self.observation_space.shape
    if out is None:
      out = torch.empty(shape, dtype=torch.float32, device=self.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
      raise ValueError(f"out must be a contiguous float32 tensor of shape {shape}")
    _lib.call("dx_synth_mujoco_step", _lib.ptr(out), _lib.ptr(rewards), _lib.ptr(resets), self.nenvs,
              shape[1], self.seed, self.counter, float(self.p_reset), _lib.stream_ptr(self.device))
    self.counter += 1
    return out

  def reset(self, out=None):
    return self._generate(out)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    del actions  # the synthetic dynamics ignore the action
    rewards = rewards_out if rewards_out is not None else torch.empty(
        self.nenvs, dtype=torch.float32, device=self.device)
    resets = resets_out if resets_out is not None else torch.empty(
        self.nenvs, dtype=torch.bool, device=self.device)
    obs = self._generate(out, rewards, resets)
    return obs, rewards, resets, None
