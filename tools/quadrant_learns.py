"""The conv path as a learner: PPO (NatureCNN, device-resident runner, frames gathered by index in
the conv loader) on a contextual bandit whose frames show a bright quadrant = the rewarded action.
usage: python tools/quadrant_learns.py [iterations]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import derl_amd as derl  # noqa: E402
from derl_amd.env.spaces import Box, Discrete  # noqa: E402


class QuadrantEnv:
  """Device-resident batched env (the contract of derl_amd/env/synthetic.py): uint8 (N, 84, 84, 4)
  noise frames with one bright 42 x 42 quadrant; reward 1 when the action names that quadrant."""
  def __init__(self, nenvs, seed=0, device="cuda"):
    self.nenvs, self.unwrapped, self.device = int(nenvs), self, torch.device(device)
    self.observation_space = Box(0, 255, (84, 84, 4), np.uint8)
    self.action_space = Discrete(4)
    self.generator = torch.Generator(device=self.device)
    self.generator.manual_seed(seed)
    self.target = None

  def _render(self, out):
    shape = (self.nenvs, 84, 84, 4)
    if out is None:
      out = torch.empty(shape, dtype=torch.uint8, device=self.device)
    self.target = torch.randint(0, 4, (self.nenvs,), device=self.device, generator=self.generator)
    frame = torch.randint(0, 64, shape, device=self.device, generator=self.generator, dtype=torch.int32)
    rows = torch.arange(84, device=self.device)
    top = (self.target // 2)[:, None] == (rows >= 42)[None].long()   # (N, 84): row in the target half
    left = (self.target % 2)[:, None] == (rows >= 42)[None].long()
    frame += 150 * (top[:, :, None] & left[:, None, :])[..., None].int()
    out.copy_(frame.to(torch.uint8))
    return out

  def reset(self, out=None):
    return self._render(out)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    rewards = (actions.reshape(-1) == self.target).float()
    resets = torch.zeros(self.nenvs, dtype=torch.bool, device=self.device)
    obs = self._render(out)
    if rewards_out is not None:
      rewards = rewards_out.copy_(rewards)
    if resets_out is not None:
      resets = resets_out.copy_(resets)
    return obs, rewards, resets, None


def run(iterations=40, nenvs=64, horizon=16, seed=0, lr=1e-3, algorithm="ppo"):
  derl.summary.stop_recording()
  torch.manual_seed(seed)
  np.random.seed(seed)
  env = QuadrantEnv(nenvs, seed)
  factory = derl.PPOFactory if algorithm == "ppo" else derl.A2CFactory
  kwargs = factory.get_kwargs("atari") if algorithm == "ppo" else factory.get_kwargs()
  kwargs.update(nenvs=nenvs, num_runner_steps=horizon, num_train_steps=nenvs * horizon * iterations, lr=lr)
  alg = factory(**kwargs).make(env)
  updates = kwargs["num_epochs"] * kwargs["num_minibatches"] if algorithm == "ppo" else 1
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
  its = int(sys.argv[1]) if len(sys.argv) > 1 else 40
  algo = sys.argv[2] if len(sys.argv) > 2 else "ppo"
  curve, seconds = run(its, algorithm=algo) if algo == "ppo" else run(its, horizon=5, lr=7e-4, algorithm=algo)
  print(json.dumps(dict(algorithm=algo, iterations=its, seconds=round(seconds, 2),
                        mean_reward=[round(float(np.mean(curve[i:i + 10])), 3) for i in range(0, its, max(its // 20, 1))])))
