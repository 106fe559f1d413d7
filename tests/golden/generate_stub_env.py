"""Deterministic env / policy used for the runner-contract golden vectors (same definitions
as in generate.py, importable without the reference)."""
import numpy as np


class CountingEnv:
  """obs[t] = t broadcast, reward = action parity, reset every 5 steps."""
  def __init__(self, nenvs):
    self.nenvs = nenvs
    self.t = 0
    self.unwrapped = self

  def reset(self):
    self.t = 0
    return np.zeros((self.nenvs, 3), np.float32)

  def step(self, actions):
    self.t += 1
    obs = np.full((self.nenvs, 3), self.t, np.float32)
    rew = (np.asarray(actions) % 2).astype(np.float64)
    done = np.full(self.nenvs, self.t % 5 == 0)
    return obs, rew, done, [{} for _ in range(self.nenvs)]


class CountingPolicy:
  def __init__(self):
    self.calls = 0

  def is_recurrent(self):
    return False

  def act(self, inputs, state=None, update_state=True, training=False):
    self.calls += 1
    n = inputs.shape[0]
    return dict(actions=np.arange(n) + self.calls, log_prob=np.full(n, -0.5, np.float32),
                values=np.full((n, 1), float(self.calls), np.float32))
