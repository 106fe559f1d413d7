"""GAE backward recursion, time/batch merge and advantage normalisation (CPU oracle).

Follows derl/runners/trajectory_transforms.py:18-72 (GAE.__call__), :75-81
(MergeTimeBatch) and :84-92 (NormalizeAdvantages).
"""
import numpy as np


def gae_advantages(rewards, resets, values, last_values, gamma=0.99, lambda_=0.95):
  """Returns (advantages, value_targets) exactly as the reference computes them.

  trajectory_transforms.py:45-65.  ``1 - resets`` is an int64 array and gamma a
  Python float, so each step is evaluated in float64 and rounded to float32 when
  stored into the float32 ``gae`` array; this restatement keeps that.

  rewards (T, ...) float, resets (T, ...) bool, values (T, ..., [1]) float32,
  last_values (..., [1]) float32 = value of state["latest_observations"].
  value_targets keeps ``values``' trailing unit dimension (:63-65).
  """
  rewards = np.asarray(rewards)
  resets = np.asarray(resets)
  values_in = np.asarray(values)
  values = values_in
  if not (0 <= values.ndim - rewards.ndim <= 1) or (
      values.ndim == rewards.ndim + 1 and values.shape[-1] != 1):
    raise ValueError("values must match rewards' ndim or carry a trailing 1")  # :35-41
  if values.ndim == rewards.ndim + 1:
    values = values[..., 0]
  last_values = np.asarray(last_values)
  if np.asarray(resets[-1]).ndim < last_values.ndim:  # :51-52
    last_values = last_values[..., 0]
  nsteps = values.shape[0]
  gae = np.zeros_like(values, dtype=np.float32)
  not_reset = 1 - resets.astype(np.int64)
  g64 = float(gamma)
  # :46,53 -- two separate float32 stores in the reference
  gae[-1] = (rewards[-1] - values[-1]).astype(np.float32)
  gae[-1] = (gae[-1] + not_reset[-1] * g64 * last_values).astype(np.float32)
  for i in range(nsteps - 1, 0, -1):  # :56-62
    delta = (rewards[i - 1] + not_reset[i - 1] * g64 * values[i] - values[i - 1])
    # not_reset * gamma * lambda_ is evaluated left to right in the reference
    gae[i - 1] = delta + not_reset[i - 1] * g64 * float(lambda_) * gae[i]
  value_targets = gae + values
  value_targets = value_targets[(...,) + (None,) * (values_in.ndim - value_targets.ndim)]
  return gae, value_targets


def merge_time_batch(array):
  """(T, N, ...) -> (T*N, ...), time-major (trajectory_transforms.py:75-81)."""
  array = np.asarray(array)
  return np.reshape(array, (-1,) + array.shape[2:])


def normalize_advantages(advantages, epsilon=1e-8):
  """(a - mean) / (std + eps), population std, float32 numpy
  (trajectory_transforms.py:89-92)."""
  advantages = np.asarray(advantages)
  return (advantages - advantages.mean()) / (advantages.std() + epsilon)
