"""Trajectory transformations (derl/runners/trajectory_transforms.py) on device tensors.

Each transform is ``callable(trajectory_dict) -> None`` mutating the dict in place, like
the reference; NumPy inputs are uploaded once and stay on the GPU afterwards."""
import numpy as np
import torch

from .. import distributed, ops


def _device_of(policy):
  model = getattr(policy, "model", None)
  device = getattr(model, "device", None) or getattr(getattr(model, "engine", None), "device", None)
  return device or torch.device("cuda")


def to_device(value, device, dtype=None):
  if isinstance(value, np.ndarray):
    value = torch.from_numpy(np.ascontiguousarray(value))
  if not isinstance(value, torch.Tensor):
    value = torch.as_tensor(value)
  return value.to(device=device, dtype=dtype).contiguous()


class GAE:
  """Generalized Advantage Estimator (trajectory_transforms.py:5-72) on the device scan."""
  def __init__(self, policy, gamma=0.99, lambda_=0.95, normalize=None, epsilon=1e-8):
    self.policy = policy
    self.gamma = gamma
    self.lambda_ = lambda_
    self.normalize = normalize
    self.epsilon = epsilon

  def __call__(self, trajectory):
    """Returns (advantages, value_targets) and stores them in the trajectory."""
    if "advantages" in trajectory:
      raise ValueError("trajectory cannot contain 'advantages'")
    if "value_targets" in trajectory:
      raise ValueError("trajectory cannot contain 'value_targets'")
    device = _device_of(self.policy)
    rewards = to_device(trajectory["rewards"], device, torch.float32)
    resets = to_device(trajectory["resets"], device)
    values_in = to_device(trajectory["values"], device, torch.float32)
    values = values_in
    if (not (0 <= values.ndim - rewards.ndim <= 1)
        or values.ndim == rewards.ndim + 1 and values.shape[-1] != 1):
      raise ValueError(
          f"trajectory['values'] of shape {tuple(values_in.shape)} "
          "must have the same number of dimensions as "
          f"trajectory['rewards'] which has shape {tuple(rewards.shape)} "
          "or have last dimension of size 1")
    if values.ndim == rewards.ndim + 1:
      values = values.squeeze(-1)
    if rewards.ndim not in (1, 2):
      raise ValueError(f"rewards must be (T,) or (T, N), got {tuple(rewards.shape)}")
    observation = trajectory["state"]["latest_observations"]
    state = trajectory["state"].get("policy_state", None)
    last_value = self.policy.act(observation, state=state, update_state=False)["values"]
    last_value = to_device(last_value, device, torch.float32).reshape(-1)
    shape = tuple(values.shape)
    T = shape[0]
    N = 1 if rewards.ndim == 1 else shape[1]
    gae, value_targets = ops.gae(rewards.reshape(T, N).contiguous(),
                                 resets.reshape(T, N).contiguous(),
                                 values.reshape(T, N).contiguous(), last_value,
                                 self.gamma, self.lambda_)
    gae = gae.reshape(shape)
    value_targets = value_targets.reshape(shape)
    value_targets = value_targets[(...,) + (None,) * (values_in.ndim - value_targets.ndim)]
    if self.normalize or self.normalize is None and gae.numel() > 1:
      gae = ops.adv_normalize(gae.reshape(-1), self.epsilon).reshape(shape)
    trajectory["rewards"], trajectory["resets"], trajectory["values"] = rewards, resets, values_in
    trajectory["advantages"] = gae
    trajectory["value_targets"] = value_targets
    return gae, value_targets


class MergeTimeBatch:
  """Merges the time and env-batch axes (trajectory_transforms.py:75-81): views, no copy."""
  def __call__(self, trajectory):
    assert trajectory["resets"].ndim == 2, trajectory["resets"].shape
    for key, val in trajectory.items():
      if isinstance(val, (np.ndarray, torch.Tensor)):
        trajectory[key] = val.reshape((-1,) + tuple(val.shape[2:]))


class NormalizeAdvantages:
  """(a - mean) / (std + eps) per minibatch (trajectory_transforms.py:84-92).  When the
  batch is sharded over ranks the statistics are the GLOBAL ones: ``prepare``,
  called by ``IterateWithMinibatches`` once the order of a rollout's minibatches is known,
  sums {sum, sumsq, count} of ALL of them over the ranks with one small all-reduce; a minibatch
  that arrives without prepared statistics costs one 3-double all-reduce of its own."""
  STATE_KEY = "advantage_stats"

  def __init__(self, epsilon=1e-8):
    self.epsilon = epsilon

  def prepare(self, interactions, orders_dev, mbsize):
    """orders_dev: (epochs, samples) int32 composed permutations on the device."""
    advantages = interactions.get("advantages")
    if (not distributed.sharded() or not isinstance(advantages, torch.Tensor)
        or not advantages.is_cuda or advantages.dtype != torch.float32
        or advantages.numel() != orders_dev.shape[1]):
      return None
    flat = advantages.reshape(-1)
    per_epoch = -(-orders_dev.shape[1] // mbsize)
    stats = torch.empty((orders_dev.shape[0], per_epoch, 3), dtype=torch.float64, device=flat.device)
    for epoch in range(orders_dev.shape[0]):
      ops.adv_stats_segments(flat, orders_dev[epoch], mbsize, stats=stats[epoch])
    distributed.all_reduce_sum(stats)

    def extras(epoch, k):
      return {self.STATE_KEY: stats[epoch, k]}

    extras.epoch_stats = lambda epoch: stats[epoch]  # (minibatches, 3): what a native epoch takes
    return extras

  def __call__(self, trajectory):
    state = trajectory.get("state")
    epoch = state.get("epoch") if isinstance(state, dict) else None
    if epoch is not None and epoch[0].normalized is not None and epoch[0].norm_eps == self.epsilon:
      # the trainer enqueued this epoch's updates natively and normalised every minibatch on the way
      context, k = epoch
      lazy = getattr(trajectory, "lazy_set", None)
      if lazy is not None and not trajectory.touched and trajectory.rows is not None:
        first, stop = trajectory.rows  # a LazyMinibatch straight from the iterator: nothing is cut yet
        lazy("advantages", lambda: context.normalized[first:stop])
        return
      advantages = trajectory["advantages"]
      start = k * context.mbsize
      trajectory["advantages"] = context.normalized[start:start + advantages.numel()].reshape(advantages.shape)
      return
    advantages = trajectory["advantages"]
    if not isinstance(advantages, torch.Tensor) or not advantages.is_cuda:
      advantages = to_device(advantages, torch.device("cuda"), torch.float32)
    flat = advantages.reshape(-1)
    first_of_epoch = epoch is not None and epoch[1] == 0 and not epoch[0].consumed
    ready = state.get(self.STATE_KEY) if isinstance(state, dict) else None
    if ready is not None:
      out = ops.adv_normalize(flat, self.epsilon, stats=ready, stats_ready=True)
    elif distributed.sharded():
      stats = ops.adv_stats(flat)
      distributed.all_reduce_sum(stats)
      out = ops.adv_normalize(flat, self.epsilon, stats=stats, stats_ready=True)
    else:
      out = ops.adv_normalize(flat, self.epsilon)
    trajectory["advantages"] = out.reshape(advantages.shape)
    if first_of_epoch and (ready is not None or not distributed.sharded()):
      # opt in to the native epoch (EpochContext): same rule, this epsilon (a sharded run only with
      # the prepared global statistics of every minibatch)
      epoch[0].norm_eps, epoch[0].norm_first = self.epsilon, trajectory["advantages"]


class Take:
  """Keeps data only from specified indices (trajectory_transforms.py:95-103)."""
  def __init__(self, indices, axis=1):
    self.indices = indices
    self.axis = axis

  def __call__(self, trajectory):
    for key, val in trajectory.items():
      if key == "state" or val is None:
        continue
      if isinstance(val, torch.Tensor):
        index = torch.as_tensor(self.indices, device=val.device).long()
        trajectory[key] = val.index_select(self.axis, index.reshape(-1))
      else:
        trajectory[key] = np.take(val, self.indices, axis=self.axis)
