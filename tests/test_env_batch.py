"""Env batches against vectors recorded from derl/env/env_batch.py on scripted
envs (tests/golden/generate_env_batch.py), plus their argument / error contract."""
import os

import numpy as np
import numpy.testing as nt
import pytest

from derl_amd.env import EnvBatch, ParallelEnvBatch, SingleEnvBatch, SpaceBatch
from tests.golden.generate_stub_env import ScriptedEnv, _PlainSpace, scripted_actions

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_batch.npz"))
LENS = [3, 4, 6, 2]


def factories():
  return [lambda i=i, n=n: ScriptedEnv(i, n) for i, n in enumerate(LENS)]


def check(batch, nenvs, tag, step="step", reset="reset"):
  nt.assert_array_equal(np.asarray(getattr(batch, reset)()), GOLD[f"{tag}.reset"])
  for t in range(GOLD[f"{tag}.obs"].shape[0]):
    obs, rew, done, infos = getattr(batch, step)(scripted_actions(t, nenvs))
    nt.assert_array_equal(np.asarray(obs), GOLD[f"{tag}.obs"][t])   # auto-reset observations
    nt.assert_array_equal(np.asarray(rew), GOLD[f"{tag}.rewards"][t])
    nt.assert_array_equal(np.asarray(done), GOLD[f"{tag}.dones"][t])
    nt.assert_array_equal([info["t"] for info in infos], GOLD[f"{tag}.info_t"][t])


def test_env_batch_matches_reference():
  check(EnvBatch(factories()), 4, "serial")
  check(EnvBatch(lambda: ScriptedEnv(7, 5), nenvs=3), 3, "same")


def test_single_env_batch_matches_reference():
  check(SingleEnvBatch(ScriptedEnv(9, 4)), 1, "single")


@pytest.mark.parametrize("shared,start_method", [(False, None), (True, None), (False, "forkserver"),
                                                 (True, "spawn")])
def test_parallel_env_batch_matches_reference(shared, start_method):
  """fork (the default while the process has not touched the GPU) and the fork-free start methods
  a GPU process must use: closures travel to the workers by value (cloudpickle)."""
  batch = ParallelEnvBatch(factories(), start_method=start_method)
  try:
    assert batch.start_method == (start_method or "fork")
    assert batch.nenvs == 4 and batch.observation_space.shape == (3,)
    if shared:
      check(batch, 4, "parallel", step="step_shared", reset="reset_shared")
    else:
      check(batch, 4, "parallel")
  finally:
    batch.close()
  batch.close()  # idempotent


def test_parallel_step_returns_fresh_arrays():
  """step() keeps the reference's contract (callers append observations to lists); only
  step_shared() hands out the double-buffered shared rows."""
  batch = ParallelEnvBatch(factories())
  try:
    batch.reset()
    kept = [batch.step(scripted_actions(t, 4))[0] for t in range(4)]
    for t, obs in enumerate(kept):
      nt.assert_array_equal(obs, GOLD["parallel.obs"][t])
  finally:
    batch.close()


def test_argument_and_action_errors():
  with pytest.raises(ValueError, match="must be a list"):
    EnvBatch(lambda: ScriptedEnv(0, 3))
  with pytest.raises(ValueError, match="must be callable"):
    EnvBatch(factories(), nenvs=4)
  batch = EnvBatch(factories())
  with pytest.raises(ValueError, match="number of actions"):
    batch.step([0, 1])
  with pytest.raises(ValueError, match="render not defined"):
    ParallelEnvBatch.render(batch)


def test_space_batch_checks():
  a, b = _PlainSpace((3,), np.float32), _PlainSpace((4,), np.float32)
  with pytest.raises(ValueError, match="different shapes"):
    SpaceBatch([a, b])
  with pytest.raises(ValueError, match="different data types"):
    SpaceBatch([a, _PlainSpace((3,), np.float64)])

  class Other(_PlainSpace):
    pass
  with pytest.raises(TypeError, match="different types"):
    SpaceBatch([Other((3,), np.float32), a])
  space = SpaceBatch([_PlainSpace((), np.int64, n=5)] * 2)
  assert space.n == 5 and space.shape == () and space.sample().shape == (2,)


def test_worker_start_method_follows_gpu_state(monkeypatch):
  """fork only while the GPU runtime is untouched; forkserver afterwards; env override wins; an
  explicit fork from a GPU process is refused."""
  from derl_amd.env import env_batch
  assert env_batch.worker_start_method() == "fork"  # the CPU suite never initialises the GPU
  monkeypatch.setattr(env_batch, "_gpu_in_use", lambda: True)
  assert env_batch.worker_start_method() == "forkserver"
  with pytest.raises(RuntimeError, match="refusing to fork"):
    ParallelEnvBatch(factories(), start_method="fork")
  monkeypatch.setenv("DERL_AMD_ENV_START_METHOD", "spawn")
  assert env_batch.worker_start_method() == "spawn"
