"""Reward summaries: oracle, host summarizer and device kernel against the rows the
reference RewardSummarizer hands to summary.add_scalar (tests/golden/summarize.npz)."""
import os

import numpy as np
import numpy.testing as nt
import pytest

from oracle.summarize import RewardSummarizerOracle
from tests.golden.generate_summarize import CASES, TAGS, summarize_inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "summarize.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_rows(name):
  case = CASES[name]
  rewards, resets = summarize_inputs(**case)
  summ = RewardSummarizerOracle(case["N"], case["Q"])
  rows = [r for t in range(case["T"]) for r in [summ.step(rewards[t].astype(np.float64), resets[t])] if r is not None]
  assert len(rows) == len(GOLD[f"{name}.rows"]) > 0
  nt.assert_allclose(np.array(rows), GOLD[f"{name}.rows"], rtol=1e-13)
  nt.assert_allclose(summ.rewards, GOLD[f"{name}.final_rewards"])


def collect(prefix, size):
  from derl_amd import summary
  rows, calls = [], []
  orig = summary.add_scalar

  def capture(tag, value, global_step=None, **kwargs):
    calls.append((tag, float(value), global_step))
  summary.add_scalar = capture
  return calls, orig


def rows_from(calls, size):
  out = []
  for k in range(0, len(calls), 5):
    chunk = dict((tag.split("/")[1], (val, step)) for tag, val, step in calls[k:k + 5])
    out.append([chunk[t][0] for t in TAGS] + [chunk[f"reward_mean_{size}"][0], chunk["total_reward"][1]])
  return np.array(out, np.float64).reshape(-1, 6)


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_summarizer_matches_reference_rows(name):
  from derl_amd import summary
  from derl_amd.env.summarize import RewardSummarizer
  case = CASES[name]
  rewards, resets = summarize_inputs(**case)
  calls, orig = collect("env", case["Q"])
  summary.start_recording()
  try:
    summ = RewardSummarizer(case["N"], "env", running_mean_size=case["Q"])
    for t in range(case["T"]):
      summ.step(rewards[t].astype(np.float64), resets[t])
  finally:
    summary.add_scalar = orig
    summary.stop_recording()
  nt.assert_allclose(rows_from(calls, case["Q"]), GOLD[f"{name}.rows"], rtol=1e-13)
  nt.assert_allclose(summ.rewards, GOLD[f"{name}.final_rewards"])
  nt.assert_allclose(summ.episode_lengths, GOLD[f"{name}.final_lengths"])


def test_summarize_wrapper_uses_real_done():
  from derl_amd import summary
  from derl_amd.env.summarize import Summarize

  class Env:
    nenvs, unwrapped = 2, None

    def step(self, action):
      return np.zeros((2, 1)), np.array([1.0, 2.0]), np.array([True, False]), [{"real_done": False}, {}]

    def reset(self):
      return np.zeros((2, 1))
  env = Env()
  env.unwrapped = env
  wrapped = Summarize.reward_summarizer(env, prefix="x")
  summary.stop_recording()
  wrapped.reset()
  wrapped.step(None)
  assert not wrapped.summarizer.had_ended_episodes.any()  # real_done overrides done for env 0
  nt.assert_array_equal(wrapped.summarizer.rewards, [1.0, 2.0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("chunk", [1, 7, 1000])
def test_device_summarizer_matches_reference_rows(name, chunk):
  """Whole rollouts (or single steps) per native call give the reference's rows."""
  import torch
  from derl_amd import summary
  from derl_amd.env.summarize import DeviceSummarize
  case = CASES[name]
  rewards, resets = summarize_inputs(**case)
  dev = torch.device("cuda:0")

  class Env:
    device, nenvs, observation_space, action_space = dev, case["N"], None, None
  env = Env()
  env.unwrapped = env
  calls, orig = collect("env", case["Q"])
  summary.start_recording()
  try:
    summ = DeviceSummarize(env, "env", running_mean_size=case["Q"], max_rows=max(8, min(chunk, case["T"])))
    r, z = torch.from_numpy(rewards).to(dev), torch.from_numpy(resets).to(dev)
    for t0 in range(0, case["T"], chunk):
      summ.rollout_done(r[t0:t0 + chunk], z[t0:t0 + chunk])
  finally:
    summary.add_scalar = orig
    summary.stop_recording()
  nt.assert_allclose(rows_from(calls, case["Q"]), GOLD[f"{name}.rows"], rtol=1e-12)
  nt.assert_allclose(summ.acc.cpu().numpy(), GOLD[f"{name}.final_rewards"])
  nt.assert_allclose(summ.ep_len.cpu().numpy(), GOLD[f"{name}.final_lengths"])
  assert int(summ.step_count_dev.item()) == case["T"] * case["N"]


@pytest.mark.gpu
def test_make_with_summarize_emits_tags_during_training():
  """derl.env.make(..., summarize=True): the device runner's hook feeds DeviceSummarize and the
  reference's tags appear once every env has finished an episode (fused native rollout kept)."""
  import derl_amd as derl
  from derl_amd import summary
  env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=16, seed=3, summarize=True)
  env.unwrapped.p_reset = 0.2  # short synthetic episodes
  kwargs = derl.PPOFactory.get_kwargs("atari")
  kwargs.update(nenvs=16, num_runner_steps=32, num_train_steps=16 * 32 * 2, num_epochs=1, num_minibatches=2)
  alg = derl.PPOFactory(**kwargs).make(env, nlogs=2)
  summary.last_scalars.clear()
  for data in alg.runner.run():
    alg.step(data)
  summary.stop_recording()
  tags = [t for t in summary.last_scalars if t.startswith("BreakoutNoFrameskip-v4/")]
  assert sorted(t.split("/")[1] for t in tags) == ["episode_length", "max_reward", "min_reward",
                                                   "reward_mean_100", "total_reward"]
  assert int(env.step_count_dev.item()) == 16 * 32 * 2
