"""Host envs feeding the device-resident rollout.

``HostEnvBridge`` wraps any batched host env (``EnvBatch`` / ``ParallelEnvBatch`` / a user's
object with ``nenvs``, ``reset()``, ``step(actions)``) and gives it the interface the
device-resident ``EnvRunner`` drives (``device``, ``reset(out=)``,
``step(actions, out=, rewards_out=, resets_out=)``): the policy forward, GAE and the update
then stay on the GPU exactly as with the synthetic env, and per env step only

  * the action vector comes to the host (one small D2H copy -- the env needs it anyway), and
  * the new frames go to their slot of the rollout buffer in HBM through a PINNED, double
    buffered staging area with an asynchronous copy: the host may fill staging slot k+1
    while the DMA of slot k is still in flight, and nothing is stacked per rollout
    (derl/runners/onpolicy.py:18-27 stacks 2 x 925 MB at C2).

With a ``ParallelEnvBatch`` the workers already write into shared memory, so the staging copy
is the only host-side touch of a frame.
"""
import numpy as np
import torch


class HostEnvBridge:
  """Device-facing adapter of a batched host env."""
  def __init__(self, env, device="cuda"):
    if getattr(env.unwrapped if hasattr(env, "unwrapped") else env, "nenvs", None) is None:
      raise TypeError(f"HostEnvBridge needs a batched env (with nenvs), got {env}")
    self.env = env
    self.device = torch.device(device)
    self.nenvs = env.nenvs
    self.unwrapped = self
    self.observation_space = env.observation_space
    self.action_space = env.action_space
    ospace = env.observation_space
    self._obs_dtype = torch.uint8 if np.dtype(ospace.dtype) == np.uint8 else torch.float32
    shape = (self.nenvs,) + tuple(ospace.shape)
    pin = self.device.type == "cuda"
    self._stage_obs = [torch.empty(shape, dtype=self._obs_dtype, pin_memory=pin) for _ in range(2)]
    self._stage_rew = [torch.empty(self.nenvs, dtype=torch.float32, pin_memory=pin) for _ in range(2)]
    self._stage_done = [torch.empty(self.nenvs, dtype=torch.bool, pin_memory=pin) for _ in range(2)]
    self._events = [None, None]
    self._slot = 0
    self.last_infos = None

  def _next_slot(self):
    self._slot ^= 1
    event = self._events[self._slot]
    if event is not None:
      event.synchronize()  # the DMA that last read this staging slot has finished
    return self._slot

  def _upload(self, slot, obs, out):
    stage = self._stage_obs[slot]
    stage.numpy()[...] = obs  # one host copy (casts float64 observations to float32)
    if out is None:
      out = torch.empty(stage.shape, dtype=stage.dtype, device=self.device)
    elif tuple(out.shape) != tuple(stage.shape) or out.dtype != stage.dtype:
      raise ValueError(f"out must be a {stage.dtype} tensor of shape {tuple(stage.shape)}")
    out.copy_(stage, non_blocking=True)
    return out

  def _record(self, slot):
    if self.device.type == "cuda":
      event = self._events[slot] or torch.cuda.Event()
      event.record(torch.cuda.current_stream(self.device))
      self._events[slot] = event

  def reset(self, out=None):
    slot = self._next_slot()
    reset = getattr(self.env, "reset_shared", self.env.reset)
    out = self._upload(slot, reset(), out)
    self._record(slot)
    return out

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    if isinstance(actions, torch.Tensor):
      actions = actions.detach().cpu().numpy()  # synchronises: the env needs the values
    step = getattr(self.env, "step_shared", self.env.step)
    obs, rewards, dones, infos = step(actions)
    slot = self._next_slot()
    out = self._upload(slot, obs, out)
    self._stage_rew[slot].numpy()[...] = rewards
    self._stage_done[slot].numpy()[...] = dones
    if rewards_out is None:
      rewards_out = torch.empty(self.nenvs, dtype=torch.float32, device=self.device)
    if resets_out is None:
      resets_out = torch.empty(self.nenvs, dtype=torch.bool, device=self.device)
    rewards_out.copy_(self._stage_rew[slot], non_blocking=True)
    resets_out.copy_(self._stage_done[slot], non_blocking=True)
    self._record(slot)
    self.last_infos = infos
    return out, rewards_out, resets_out, infos

  def close(self):
    close = getattr(self.env, "close", None)
    if close is not None:
      close()
