"""Golden vectors of derl's env batches (derl/env/env_batch.py:35-199) on scripted envs:
run HERE (where /root/reference exists) with `python -m tests.golden.generate_env_batch`;
writes tests/golden/env_batch.npz."""
import os

import numpy as np

from . import _ref_import
from .generate_stub_env import ScriptedEnv, scripted_actions

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_batch.npz")
NSTEPS = 14


def record(batch, nenvs):
  out = {"reset": np.asarray(batch.reset())}
  obs, rews, dones, ts = [], [], [], []
  for step in range(NSTEPS):
    ob, rew, done, infos = batch.step(scripted_actions(step, nenvs))
    obs.append(np.asarray(ob)); rews.append(np.asarray(rew)); dones.append(np.asarray(done))
    ts.append(np.array([info["t"] for info in infos]))
  out.update(obs=np.stack(obs), rewards=np.stack(rews), dones=np.stack(dones), info_t=np.stack(ts))
  return out


def main():
  derl = _ref_import.import_reference()
  from derl.env.env_batch import EnvBatch, ParallelEnvBatch, SingleEnvBatch  # pylint: disable=import-error
  del derl
  lens = [3, 4, 6, 2]
  fns = [lambda i=i, n=n: ScriptedEnv(i, n) for i, n in enumerate(lens)]
  result = {}
  for name, batch, nenvs in (("serial", EnvBatch(fns), 4),
                             ("same", EnvBatch(lambda: ScriptedEnv(7, 5), nenvs=3), 3),
                             ("single", None, 1),
                             ("parallel", ParallelEnvBatch(fns), 4)):
    if name == "single":
      # SingleEnvBatch is a gym.Wrapper: the stub Wrapper forwards attributes like gym's does
      batch = SingleEnvBatch(ScriptedEnv(9, 4))
    for key, val in record(batch, nenvs).items():
      result[f"{name}.{key}"] = val
    if name == "parallel":
      batch.close()
  np.savez(OUT, **result)
  print("wrote", OUT, {k: v.shape for k, v in result.items()})


if __name__ == "__main__":
  main()
