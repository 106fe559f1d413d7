"""HostEnvBridge: host envs feeding the device-resident rollout give exactly
the interactions the generic (reference-contract) runner builds from the same envs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Space:
  def __init__(self, shape, dtype, n=None):
    self.shape, self.dtype = tuple(shape), np.dtype(dtype)
    if n is not None:
      self.n = n


class FrameEnv:
  """uint8 (84, 84, 4) frames that depend on (tag, episode, t, last action); fixed episode length."""
  def __init__(self, tag, episode_len):
    self.tag, self.episode_len = tag, episode_len
    self.episode, self.t, self.action = -1, 0, 0
    self.observation_space = _Space((84, 84, 4), np.uint8)
    self.action_space = _Space((), np.int64, n=4)

  def _obs(self):
    rs = np.random.RandomState(1000 * self.tag + 37 * self.episode + 5 * self.t + self.action)
    return rs.randint(0, 256, size=(84, 84, 4)).astype(np.uint8)

  def reset(self):
    self.episode += 1
    self.t, self.action = 0, 0
    return self._obs()

  def step(self, action):
    self.t += 1
    self.action = int(action)
    return self._obs(), float(action) - 1.5, self.t == self.episode_len, {}


def _rollouts(kind, batch_cls_name):
  import derl_amd as derl
  from derl_amd.env import EnvBatch, HostEnvBridge, ParallelEnvBatch
  from derl_amd.policies import ActorCriticPolicy
  torch.manual_seed(0)
  model = derl.NatureCNNModel([4, 1], max_batch=64)
  policy = ActorCriticPolicy(model, seed=11)
  fns = [lambda i=i: FrameEnv(i, 3 + i % 4) for i in range(6)]
  host = (ParallelEnvBatch if batch_cls_name == "parallel" else EnvBatch)(fns)
  env = HostEnvBridge(host) if kind == "bridge" else host
  try:
    runner = derl.EnvRunner(env, policy, horizon=5, nsteps=6 * 5 * 2)
    outs = []
    for inter in runner.run():
      rec = {}
      for key in ("observations", "next_observations", "actions", "log_prob", "values", "rewards", "resets"):
        val = inter[key]
        val = val.cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
        rec[key] = val.copy()
      latest = inter["state"]["latest_observations"]
      rec["latest"] = latest.cpu().numpy().copy() if isinstance(latest, torch.Tensor) else np.asarray(latest)
      outs.append(rec)
    assert runner.step_count == 6 * 5 * 2
    return outs
  finally:
    host.close()


@pytest.mark.parametrize("batch_cls_name", ["serial", "parallel"])
def test_bridge_equals_generic_runner(batch_cls_name):
  generic = _rollouts("generic", batch_cls_name)
  bridged = _rollouts("bridge", batch_cls_name)
  assert len(generic) == len(bridged) == 2
  for g, b in zip(generic, bridged):
    for key in g:
      assert g[key].shape == b[key].shape, key
      if g[key].dtype.kind == "f":
        np.testing.assert_allclose(b[key], g[key].astype(b[key].dtype), rtol=0, atol=0, err_msg=key)
      else:
        np.testing.assert_array_equal(b[key], g[key], err_msg=key)
