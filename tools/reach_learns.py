"""The Gaussian MLP path as a learner: PPO (MuJoCo preset, fused two-net MLP kernels, diagonal
Gaussian loss incl. logstd) on a device-resident reaching task: the observation holds a target
vector, the reward is -mean((action - target)^2).  usage: python tools/reach_learns.py [iterations]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import derl_amd as derl  # noqa: E402
from derl_amd.env.spaces import Box  # noqa: E402


class ReachEnv:
  """obs (N, 17) float32: columns 0..5 the target in [-1, 1], the rest noise; actions (N, 6)."""
  def __init__(self, nenvs, seed=0, device="cuda"):
    self.nenvs, self.unwrapped, self.device = int(nenvs), self, torch.device(device)
    self.observation_space = Box(-10., 10., (17,), np.float32)
    self.action_space = Box(-1., 1., (6,), np.float32)
    self.generator = torch.Generator(device=self.device)
    self.generator.manual_seed(seed)
    self.obs = None

  def _draw(self, out):
    if out is None:
      out = torch.empty((self.nenvs, 17), dtype=torch.float32, device=self.device)
    out.normal_(generator=self.generator)
    out[:, :6] = torch.rand((self.nenvs, 6), device=self.device, generator=self.generator) * 2 - 1
    self.obs = out
    return out

  def reset(self, out=None):
    return self._draw(out)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    rewards = -((actions.reshape(self.nenvs, 6) - self.obs[:, :6]) ** 2).mean(1)
    resets = torch.zeros(self.nenvs, dtype=torch.bool, device=self.device)
    obs = self._draw(out)
    if rewards_out is not None:
      rewards = rewards_out.copy_(rewards)
    if resets_out is not None:
      resets = resets_out.copy_(resets)
    return obs, rewards, resets, None


def run(iterations=30, nenvs=64, horizon=64, seed=0):
  derl.summary.stop_recording()
  torch.manual_seed(seed)
  np.random.seed(seed)
  env = ReachEnv(nenvs, seed)
  kwargs = derl.PPOFactory.get_kwargs("mujoco")
  kwargs.update(nenvs=nenvs, num_runner_steps=horizon, num_train_steps=nenvs * horizon * iterations)
  alg = derl.PPOFactory(**kwargs).make(env)
  updates = kwargs["num_epochs"] * kwargs["num_minibatches"]
  data, curve = alg.runner.run(), []
  start = time.perf_counter()
  for _ in range(iterations):
    for k in range(updates):
      batch = next(data)
      if k == 0:
        curve.append(float(alg.runner.unwrapped._buffers["rewards"].mean().item()))
      alg.step(batch)
      derl.summary.stop_recording()
  return curve, time.perf_counter() - start


if __name__ == "__main__":
  its = int(sys.argv[1]) if len(sys.argv) > 1 else 30
  curve, seconds = run(its)
  print(json.dumps(dict(iterations=its, seconds=round(seconds, 2), mean_reward=[round(c, 3) for c in curve])))
