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


class _PlainSpace:
  """Stand-in for a gym space: only the attributes derl's SpaceBatch reads."""
  def __init__(self, shape, dtype, n=None):
    self.shape, self.dtype = tuple(shape), np.dtype(dtype)
    if n is not None:
      self.n = n

  def sample(self):
    return np.zeros(self.shape, self.dtype)


class ScriptedEnv:
  """Single env with a fixed episode length: obs = (tag, episode, t), reward = 0.5*action + t,
  done when t == episode_len (the env-batch fixtures exercise the auto-reset with it)."""
  def __init__(self, tag, episode_len):
    self.tag, self.episode_len = float(tag), int(episode_len)
    self.episode, self.t = -1, 0
    self.observation_space = _PlainSpace((3,), np.float32)
    self.action_space = _PlainSpace((), np.int64, n=5)

  def _obs(self):
    return np.array([self.tag, self.episode, self.t], np.float32)

  def reset(self):
    self.episode += 1
    self.t = 0
    return self._obs()

  def step(self, action):
    self.t += 1
    reward = 0.5 * float(action) + self.t
    done = self.t == self.episode_len
    return self._obs(), reward, done, {"t": self.t}

  def close(self):
    pass


def scripted_actions(step, nenvs):
  return (np.arange(nenvs) * 3 + step) % 5
