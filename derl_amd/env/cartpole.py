"""Built-in batched CartPole-v1 on the host (NumPy) for BASELINE config 1: the classic
cart-pole dynamics (Barto, Sutton & Anderson 1983; the constants gym's CartPole-v1 uses),
vectorised over envs with auto-reset as derl's EnvBatch does (derl/env/env_batch.py:121-124).
gym is not installed in the build image; this keeps `derl ppo --env-id CartPole-v1` runnable."""
import numpy as np

from .spaces import Box, Discrete


class CartPoleBatch:
  gravity, masscart, masspole, length = 9.8, 1.0, 0.1, 0.5
  force_mag, tau = 10.0, 0.02
  theta_limit, x_limit, max_steps = 12 * 2 * np.pi / 360, 2.4, 500

  def __init__(self, nenvs, seed=0):
    self.nenvs = int(nenvs)
    self.unwrapped = self
    high = np.array([4.8, np.finfo(np.float32).max, 0.42, np.finfo(np.float32).max], np.float32)
    self.observation_space = Box(-high, high, (4,), np.float32)
    self.action_space = Discrete(2)
    self.rng = np.random.RandomState(seed)
    self.state = np.zeros((self.nenvs, 4), np.float64)
    self.steps = np.zeros(self.nenvs, np.int64)
    self.episodes_done = 0  # finished episodes so far (learning-curve checks)

  def _reset_rows(self, rows):
    self.state[rows] = self.rng.uniform(-0.05, 0.05, size=(int(rows.sum()), 4))
    self.steps[rows] = 0

  def reset(self):
    self._reset_rows(np.ones(self.nenvs, bool))
    return self.state.astype(np.float32)

  def step(self, actions):
    actions = np.asarray(actions).reshape(self.nenvs)
    x, x_dot, theta, theta_dot = self.state.T
    force = np.where(actions == 1, self.force_mag, -self.force_mag)
    total_mass = self.masspole + self.masscart
    polemass_length = self.masspole * self.length
    costheta, sintheta = np.cos(theta), np.sin(theta)
    temp = (force + polemass_length * theta_dot ** 2 * sintheta) / total_mass
    thetaacc = (self.gravity * sintheta - costheta * temp) / (
        self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / total_mass))
    xacc = temp - polemass_length * thetaacc * costheta / total_mass
    x = x + self.tau * x_dot
    x_dot = x_dot + self.tau * xacc
    theta = theta + self.tau * theta_dot
    theta_dot = theta_dot + self.tau * thetaacc
    self.state = np.stack([x, x_dot, theta, theta_dot], 1)
    self.steps += 1
    done = ((np.abs(x) > self.x_limit) | (np.abs(theta) > self.theta_limit)
            | (self.steps >= self.max_steps))
    rewards = np.ones(self.nenvs, np.float64)
    if done.any():
      self.episodes_done += int(done.sum())
      self._reset_rows(done)
    return self.state.astype(np.float32), rewards, done, [{} for _ in range(self.nenvs)]
