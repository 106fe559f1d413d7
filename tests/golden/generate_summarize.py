"""Golden vectors of derl's RewardSummarizer (derl/env/summarize.py:8-52): run HERE with
`python -m tests.golden.generate_summarize`; writes tests/golden/summarize.npz (the rows it
hands to summary.add_scalar, with their global steps)."""
import os

import numpy as np

from . import _ref_import

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "summarize.npz")
CASES = {"small": dict(T=60, N=6, Q=4, p=0.2, seed=1), "atari": dict(T=400, N=64, Q=100, p=0.05, seed=2),
         "wide": dict(T=120, N=1500, Q=3, p=0.3, seed=3)}
TAGS = ["total_reward", "episode_length", "min_reward", "max_reward"]


def summarize_inputs(T, N, p, seed, **_):
  rs = np.random.RandomState(seed)
  rewards = rs.choice([-1.0, 0.0, 1.0, 2.5], size=(T, N)).astype(np.float32)
  resets = rs.rand(T, N) < p
  return rewards, resets


def main():
  derl = _ref_import.import_reference()
  from derl.env.summarize import RewardSummarizer  # pylint: disable=import-error
  import derl.summary as summary  # pylint: disable=import-error
  del derl
  result = {}
  for name, case in CASES.items():
    rewards, resets = summarize_inputs(**case)
    calls = []
    summary.should_record = lambda: True
    summary.add_scalar = lambda tag, val, global_step=None, calls=calls: calls.append((tag, float(val), global_step))
    summ = RewardSummarizer(case["N"], "env", running_mean_size=case["Q"])
    for t in range(case["T"]):
      summ.step(rewards[t].astype(np.float64), resets[t])
    rows = []
    for k in range(0, len(calls), 5):
      chunk = dict((tag.split("/")[1], (val, step)) for tag, val, step in calls[k:k + 5])
      rows.append([chunk[t][0] for t in TAGS] + [chunk[f"reward_mean_{case['Q']}"][0], chunk["total_reward"][1]])
    result[f"{name}.rows"] = np.array(rows, np.float64).reshape(-1, 6)
    result[f"{name}.final_rewards"] = summ.rewards
    result[f"{name}.final_lengths"] = summ.episode_lengths
  np.savez(OUT, **result)
  print("wrote", OUT, {k: v.shape for k, v in result.items()})


if __name__ == "__main__":
  main()
