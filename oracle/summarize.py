"""CPU restatement of derl's RewardSummarizer (TEST INFRASTRUCTURE ONLY).  Follows
derl/env/summarize.py:8-52 with deques, as the reference does; ``step`` returns the summary row
it would write (or None).  Pinned by tests/golden/summarize.npz."""
from collections import deque

import numpy as np


class RewardSummarizerOracle:
  def __init__(self, nenvs, running_mean_size=100):
    self.step_count = 0
    self.had_ended = np.zeros(nenvs, dtype=bool)
    self.rewards = np.zeros(nenvs)
    self.episode_lengths = np.zeros(nenvs)
    self.queues = [deque([], maxlen=running_mean_size) for _ in range(nenvs)]

  def _row(self):  # add_summaries, :25-38
    last = [q[-1] for q in self.queues]
    return np.array([np.mean(last), np.mean(self.episode_lengths), min(last), max(last),
                     np.mean([np.mean(q) for q in self.queues]), self.step_count], np.float64)

  def step(self, rewards, resets, record=True):  # :40-52
    self.rewards += rewards
    self.episode_lengths[~self.had_ended] += 1
    for i in np.nonzero(resets)[0]:
      self.queues[i].append(self.rewards[i])
      self.rewards[i] = 0
      self.had_ended[i] = True
    self.step_count += self.rewards.shape[0]
    if record and np.all(self.had_ended):
      row = self._row()
      self.episode_lengths.fill(0)
      self.had_ended.fill(False)
      return row
    return None
