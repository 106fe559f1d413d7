"""CPU restatement of derl's vectorised Normalize wrapper (TEST INFRASTRUCTURE ONLY: imported
by tests/, never by the product).  Follows derl/env/mujoco_wrappers.py:8-61 (RunningMeanVar,
update_mean_var_count_from_moments) and :64-124 (Normalize.observation / step / reset) in
float64 NumPy, as the reference does.  Pinned by tests/golden/normalize.npz (recorded from the
unmodified reference wrapper by tests/golden/generate_normalize.py)."""
import numpy as np


class RunningMeanVar:
  """mujoco_wrappers.py:8-45: mean 0, var 1, count eps; parallel-variance merge per batch."""
  def __init__(self, eps=1e-4, shape=()):
    self.mean, self.var, self.count = np.zeros(shape), np.ones(shape), eps

  def update(self, batch):
    bmean, bvar, bcount = np.mean(batch, axis=0), np.var(batch, axis=0), batch.shape[0]
    delta, tot = bmean - self.mean, self.count + bcount  # :48-61
    self.var = (self.var * (self.count / tot) + bvar * (bcount / tot)
                + np.square(delta) * (self.count * bcount / tot ** 2))
    self.mean = self.mean + delta * bcount / tot
    self.count = tot


class NormalizeState:
  """The state and per-step arithmetic of Normalize for a batch of nenvs envs."""
  def __init__(self, nenvs, obs_shape, obs=True, ret=True, clipobs=10., cliprew=10., gamma=0.99,
               eps=1e-8):
    self.obs_rmv = RunningMeanVar(shape=obs_shape) if obs else None
    self.ret_rmv = RunningMeanVar(shape=()) if ret else None
    self.clipob, self.cliprew, self.gamma, self.eps = clipobs, cliprew, gamma, eps
    self.ret = np.zeros(nenvs)

  def observation(self, obs):  # :99-110
    if self.obs_rmv is None:
      return obs
    self.obs_rmv.update(obs)
    obs = (obs - self.obs_rmv.mean) / np.sqrt(self.obs_rmv.var + self.eps)
    return np.clip(obs, -self.clipob, self.clipob)

  def step(self, obs, rews, resets):  # :112-121
    self.ret = self.ret * self.gamma + rews
    obs = self.observation(obs)
    if self.ret_rmv is not None:
      self.ret_rmv.update(self.ret)
      rews = np.clip(rews / np.sqrt(self.ret_rmv.var + self.eps), -self.cliprew, self.cliprew)
    self.ret[resets] = 0.
    return obs, rews

  def reset(self, obs):  # :123-126
    self.ret = np.zeros_like(self.ret)
    return self.observation(obs)
