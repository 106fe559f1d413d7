"""Golden vectors of derl's Normalize wrapper (derl/env/mujoco_wrappers.py:64-124): run HERE with
`python -m tests.golden.generate_normalize`; writes tests/golden/normalize.npz.  Inputs are
regenerated from RandomState by `normalize_inputs` (float32-representable values so that the
device kernel sees exactly the same numbers)."""
import os

import numpy as np

from . import _ref_import

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "normalize.npz")
CASES = {"hc": dict(nenvs=64, dim=17, steps=12, seed=3, obs=True, ret=True),
         "wide": dict(nenvs=33, dim=70, steps=6, seed=4, obs=True, ret=True),
         "obs_only": dict(nenvs=16, dim=5, steps=5, seed=5, obs=True, ret=False),
         "ret_only": dict(nenvs=16, dim=5, steps=5, seed=6, obs=False, ret=True)}


def normalize_inputs(nenvs, dim, steps, seed, **_):
  rs = np.random.RandomState(seed)
  scale = rs.uniform(0.1, 30.0, size=dim)
  shift = rs.uniform(-5.0, 5.0, size=dim)
  obs = (rs.standard_normal((steps + 1, nenvs, dim)) * scale + shift).astype(np.float32)
  rewards = (rs.standard_normal((steps, nenvs)) * 3.0 + 0.5).astype(np.float32)
  resets = rs.rand(steps, nenvs) < 0.15
  return obs, rewards, resets


class ReplayEnv:
  """Batched env that replays the generated arrays (float64, as a MuJoCo env would return)."""
  def __init__(self, obs, rewards, resets):
    self.obs, self.rewards, self.resets = obs, rewards, resets
    self.nenvs, self.t = obs.shape[1], 0
    self.unwrapped = self
    self.observation_space = type("S", (), dict(shape=obs.shape[2:], dtype=np.float64))()
    self.action_space = None

  def reset(self):
    self.t = 0
    return self.obs[0].astype(np.float64)

  def step(self, action):
    del action
    t = self.t
    self.t += 1
    return (self.obs[t + 1].astype(np.float64), self.rewards[t].astype(np.float64), self.resets[t], {})


def main():
  _ref_import.import_reference()
  from derl.env.mujoco_wrappers import Normalize  # pylint: disable=import-error
  result = {}
  for name, case in CASES.items():
    obs, rewards, resets = normalize_inputs(**case)
    env = Normalize(ReplayEnv(obs, rewards, resets), obs=case["obs"], ret=case["ret"])
    out_obs, out_rew = [env.reset()], []
    for _ in range(case["steps"]):
      o, r, _, _ = env.step(None)
      out_obs.append(o); out_rew.append(r)
    result[f"{name}.obs"] = np.stack(out_obs)
    result[f"{name}.rewards"] = np.stack(out_rew)
    result[f"{name}.ret"] = env.ret
    if env.obs_rmv is not None:
      result[f"{name}.obs_mean"], result[f"{name}.obs_var"] = env.obs_rmv.mean, env.obs_rmv.var
      result[f"{name}.obs_count"] = np.float64(env.obs_rmv.count)
    if env.ret_rmv is not None:
      result[f"{name}.ret_stats"] = np.array([env.ret_rmv.mean, env.ret_rmv.var, env.ret_rmv.count])
  np.savez(OUT, **result)
  print("wrote", OUT, sorted(result))


if __name__ == "__main__":
  main()
