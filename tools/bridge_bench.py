"""PPO iteration rate when the frames come from HOST memory through HostEnvBridge (pinned,
double-buffered H2D) instead of being generated on the GPU: the PCIe-inclusive figure.

The host env is a stand-in that hands out uint8 frames from a pre-generated pool (no emulator
cost), so the number isolates the boundary: per env step one D2H of the actions, one host copy of
nenvs x 28 KB into pinned staging and one H2D of the same bytes.

usage: python tools/bridge_bench.py [nenvs] [iterations]
"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import derl_amd as derl
from derl_amd.env import Box, Discrete, HostEnvBridge


class PoolEnv:
  def __init__(self, nenvs, pool=8):
    self.nenvs, self.unwrapped = nenvs, self
    self.observation_space = Box(0, 255, (84, 84, 4), np.uint8)
    self.action_space = Discrete(4)
    rs = np.random.RandomState(0)
    self.pool = rs.randint(0, 256, size=(pool, nenvs, 84, 84, 4)).astype(np.uint8)
    self.rewards = rs.choice([-1.0, 0.0, 1.0], size=(pool, nenvs), p=[0.05, 0.9, 0.05])
    self.dones = rs.rand(pool, nenvs) < 0.01
    self.t = 0

  def reset(self):
    return self.pool[0]

  def step(self, actions):
    del actions
    self.t += 1
    k = self.t % len(self.pool)
    return self.pool[k], self.rewards[k], self.dones[k], None


nenvs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
device = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
env = HostEnvBridge(PoolEnv(nenvs), device)
kwargs = derl.PPOFactory.get_kwargs("atari")
kwargs.update(nenvs=nenvs, num_runner_steps=128, num_train_steps=1e12)
alg = derl.PPOFactory(**kwargs).make(env, nlogs=1e5)
derl.summary.stop_recording()
data_iter = alg.runner.run()
updates = kwargs["num_epochs"] * kwargs["num_minibatches"]


def iteration():
  for _ in range(updates):
    alg.step(next(data_iter))
    derl.summary.stop_recording()


for _ in range(2):
  iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
  iteration()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(json.dumps(dict(nenvs=nenvs, ms_per_iteration=round(dt * 1e3, 2),
                      env_steps_per_s=round(nenvs * 128 / dt, 1),
                      h2d_mb_per_step=round(nenvs * 28224 / 1e6, 2))))
