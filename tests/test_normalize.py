"""Normalize wrapper: the oracle against vectors recorded from the reference
(CPU), and the device kernel against both (GPU)."""
import os

import numpy as np
import numpy.testing as nt
import pytest

from oracle.normalize import NormalizeState
from tests.golden.generate_normalize import CASES, normalize_inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "normalize.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_vectors(name):
  case = CASES[name]
  obs, rewards, resets = normalize_inputs(**case)
  state = NormalizeState(case["nenvs"], (case["dim"],), obs=case["obs"], ret=case["ret"])
  nt.assert_allclose(state.reset(obs[0].astype(np.float64)), GOLD[f"{name}.obs"][0], rtol=1e-13, atol=1e-13)
  for t in range(case["steps"]):
    o, r = state.step(obs[t + 1].astype(np.float64), rewards[t].astype(np.float64), resets[t])
    nt.assert_allclose(o, GOLD[f"{name}.obs"][t + 1], rtol=1e-13, atol=1e-13)
    nt.assert_allclose(r, GOLD[f"{name}.rewards"][t], rtol=1e-13, atol=1e-13)
  nt.assert_allclose(state.ret, GOLD[f"{name}.ret"], rtol=1e-13)
  if case["obs"]:
    nt.assert_allclose(state.obs_rmv.var, GOLD[f"{name}.obs_var"], rtol=1e-13)
    assert state.obs_rmv.count == GOLD[f"{name}.obs_count"]


class _ReplayDeviceEnv:
  """Device-interface env replaying arrays (what SyntheticMuJoCoEnv / HostEnvBridge look like)."""
  def __init__(self, obs, rewards, resets, device):
    import torch
    from derl_amd.env import Box
    self.device, self.nenvs, self.unwrapped, self.t = device, obs.shape[1], self, 0
    self.obs = torch.from_numpy(obs).to(device)
    self.rewards = torch.from_numpy(rewards).to(device)
    self.resets = torch.from_numpy(resets).to(device)
    self.observation_space = Box(-np.inf, np.inf, obs.shape[2:], np.float32)
    self.action_space = Box(-1., 1., (2,), np.float32)

  def reset(self, out=None):
    self.t = 0
    return out.copy_(self.obs[0])

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    t, self.t = self.t, self.t + 1
    out.copy_(self.obs[t + 1])
    rewards_out.copy_(self.rewards[t])
    resets = self.resets[t] if resets_out is None else resets_out.copy_(self.resets[t])
    return out, rewards_out, resets, None


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_normalize_matches_reference_vectors(name, tmp_path):
  import torch
  from derl_amd.env import Normalize
  dev = torch.device("cuda:0")
  case = CASES[name]
  obs, rewards, resets = normalize_inputs(**case)
  env = Normalize(_ReplayDeviceEnv(obs, rewards, resets, dev), obs=case["obs"], ret=case["ret"])
  # float32 outputs of float64 arithmetic: half an ulp of the rounded value
  tol = dict(rtol=2e-7, atol=1e-7)
  nt.assert_allclose(env.reset().cpu().numpy(), GOLD[f"{name}.obs"][0], **tol)
  for t in range(case["steps"]):
    resets_out = torch.empty(case["nenvs"], dtype=torch.bool, device=dev) if t % 2 else None
    o, r, z, _ = env.step(None, resets_out=resets_out)
    nt.assert_allclose(o.cpu().numpy(), GOLD[f"{name}.obs"][t + 1], **tol)
    nt.assert_allclose(r.cpu().numpy(), GOLD[f"{name}.rewards"][t], **tol)
    nt.assert_array_equal(z.cpu().numpy(), resets[t])
  nt.assert_allclose(env.ret.cpu().numpy(), GOLD[f"{name}.ret"], rtol=1e-12, atol=1e-12)
  if case["obs"]:
    stats = env.obs_stats.cpu().numpy()
    nt.assert_allclose(stats[:case["dim"]], GOLD[f"{name}.obs_mean"], rtol=1e-12, atol=1e-12)
    nt.assert_allclose(stats[case["dim"]:-1], GOLD[f"{name}.obs_var"], rtol=1e-12)
    assert stats[-1] == GOLD[f"{name}.obs_count"]
  if case["ret"]:
    nt.assert_allclose(env.ret_stats.cpu().numpy(), GOLD[f"{name}.ret_stats"], rtol=1e-12)
  # persistence with the reference's file names / keys, and frozen statistics
  env.save_wrapper(str(tmp_path / "norm.npz"))
  again = Normalize(_ReplayDeviceEnv(obs, rewards, resets, dev), obs=case["obs"], ret=case["ret"])
  again.restore_wrapper(str(tmp_path / "norm."))
  if case["obs"]:
    assert sorted(np.load(tmp_path / "norm.-obs-rmv.npz").files) == ["count", "mean", "var"]
    assert torch.equal(again.obs_stats, env.obs_stats)
    x = torch.from_numpy(obs[1]).to(dev)
    before = env.obs_stats.clone()
    y = env.observation(x, update=False)
    assert torch.equal(env.obs_stats, before)
    ref = np.clip((obs[1].astype(np.float64) - GOLD[f"{name}.obs_mean"]) /
                  np.sqrt(GOLD[f"{name}.obs_var"] + 1e-8), -10, 10)
    nt.assert_allclose(y.cpu().numpy(), ref, **tol)
  if case["ret"]:
    assert torch.equal(again.ret_stats, env.ret_stats)


@pytest.mark.gpu
def test_make_with_normalize_runs_a_ppo_iteration():
  """derl.env.make(..., normalize=True) -> Normalize(SyntheticMuJoCoEnv) drives the device runner."""
  import torch
  import derl_amd as derl
  env = derl.env.make("HalfCheetah-v3", nenvs=64, seed=1, normalize=True)
  kwargs = derl.PPOFactory.get_kwargs("mujoco")
  kwargs.update(nenvs=64, num_runner_steps=16, num_train_steps=64 * 16 * 2, num_epochs=2, num_minibatches=2)
  alg = derl.PPOFactory(**kwargs).make(env, nlogs=1e9)
  derl.summary.stop_recording()
  count = 0
  for data in alg.runner.run():
    loss = alg.step(data)
    count += 1
  assert count == 2 * 2 * 2 and torch.isfinite(loss).item()
  assert float(env.obs_stats[-1]) > 64 * 16
