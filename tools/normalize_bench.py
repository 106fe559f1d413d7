"""Latency of dx_normalize_step_f32 at BASELINE config-3 shapes and a large batch."""
import json, sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd.env import Box, Normalize

dev = torch.device("cuda:0")


class Raw:
  def __init__(self, n, d):
    self.device, self.nenvs, self.unwrapped = dev, n, self
    self.observation_space = Box(-np.inf, np.inf, (d,), np.float32)
    self.action_space = Box(-1., 1., (6,), np.float32)
    self.x = torch.randn(n, d, device=dev) * 3 + 1
    self.r = torch.randn(n, device=dev)
    self.z = torch.rand(n, device=dev) < 0.01

  def reset(self, out=None):
    return out.copy_(self.x)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    return self.x, self.r, self.z, None  # no copies: time the normalisation only


for n, d in ((2048, 17), (2048, 376), (65536, 17)):
  env = Normalize(Raw(n, d))
  out = torch.empty(n, d, device=dev)
  rew = torch.empty(n, device=dev)
  env.reset(out=out)
  for _ in range(5):
    env.step(None, out=out, rewards_out=rew)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
    env.step(None, out=out, rewards_out=rew)
  e1.record(); e1.synchronize()
  us = e0.elapsed_time(e1) * 20
  nbytes = n * d * 4 * 4 + n * 25  # obs read 3x + written once; reward/reset/return traffic
  print(json.dumps(dict(N=n, D=d, us_per_step=round(us, 1), GBps=round(nbytes / us / 1e3, 1))), flush=True)
