"""Policies with derl's interface (derl/policies.py:11-80)."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import ops


class Policy(ABC):
  """RL policy (derl/policies.py:11-32)."""
  def is_recurrent(self):  # pylint: disable=no-self-use
    return False

  def get_state(self):  # pylint: disable=no-self-use
    return None

  def reset(self):  # pylint: disable=no-self-use
    """Resets the state."""

  @abstractmethod
  def act(self, inputs, state=None, update_state=True, training=False):
    """Returns `dict` of all the outputs of the policy."""


def refuse_mid_epoch(model, what):
  """Between two minibatch steps of an epoch whose updates were all enqueued by one native call
  (Trainer.native_epochs) the parameters are AHEAD of what derl's Trainer.step would show
  (derl/alg/common.py:66-78 applies one update per step): reading the policy then raises instead
  of silently answering from post-epoch parameters."""
  context = getattr(getattr(model, "engine", None), "open_epoch", None)
  if context is not None:
    raise RuntimeError(
        f"{what} between minibatch steps {context.next_k - 1} and {context.next_k} of an epoch whose "
        f"{context.num_minibatches} updates are already applied (Trainer.native_epochs: one native call per "
        "epoch).  Set trainer.native_epochs = False to evaluate the policy between minibatch updates.")


class DeviceCategorical:
  """What ``act(training=True)`` returns under "distribution" for a categorical policy:
  holds the padded head output the fused loss kernel consumes.  ``log_prob`` / ``entropy``
  / ``logits`` are provided for API compatibility (torch elementwise ops, off the hot path;
  the losses never call them)."""
  def __init__(self, head, num_actions):
    self.head = head
    self.num_actions = num_actions

  @property
  def logits(self):
    raw = self.head[:, :self.num_actions]
    return raw - torch.logsumexp(raw, -1, keepdim=True)

  def log_prob(self, actions):
    actions = torch.as_tensor(actions, device=self.head.device).long()
    return self.logits.gather(-1, actions[:, None])[:, 0]

  def entropy(self):
    logp = self.logits
    return -(logp.exp() * logp).sum(-1)


class ActorCriticPolicy(Policy):
  """Actor-critic policy (derl/policies.py:45-80).

  Rollout mode returns ``{"actions", "log_prob", "values"}``: NumPy arrays when the
  observations came from the host (drop-in for CPU envs), device tensors when they are
  already on the GPU (device-resident runner; no host round trip).  Training mode returns
  ``{"distribution", "values"}`` computed on the minibatch.  Sampling uses the
  counter-based generator of the head kernel, keyed by (seed, call counter).
  """
  def __init__(self, model, distribution=None, seed=0):
    if distribution is not None:
      raise NotImplementedError("custom distributions are not supported by the device heads")
    self.model = model
    self.distribution = distribution
    self.seed = int(seed)
    self.act_counter = 0

  def act(self, inputs, state=None, update_state=True, training=False):
    _ = update_state
    if state is not None:
      raise NotImplementedError()
    refuse_mid_epoch(self.model, "policy.act")
    return self.model.policy_act(self, inputs, training)

  def act_into(self, observations, actions_out, log_prob_out, values_out):
    """Rollout act for the device-resident runner: writes straight into the rollout
    buffers' slots for this step (no per-step allocations or copies)."""
    refuse_mid_epoch(self.model, "policy.act_into")
    self.model.policy_act_into(self, observations, actions_out, log_prob_out, values_out)

  def rollout_into(self, env, buffers, horizon):
    """Whole-rollout fast path when model and env support enqueuing all steps from one native
    call (same launches as the per-step loop).  Returns False if unsupported."""
    refuse_mid_epoch(self.model, "policy.rollout_into")
    fused = getattr(self.model, "policy_rollout_into", None)
    return bool(fused and fused(self, env, buffers, horizon))


def sampling_seed(env):
  """Seed of a policy's action-sampling stream for a given env shard: the env's own integer seed
  when it has one (``derl.env.make`` folds the run seed and the data-parallel rank into it), else
  the rank -- so the shards of a data-parallel run never draw the same uniforms for the same
  local env index, and ``--seed`` reaches action sampling."""
  from . import distributed  # pylint: disable=import-outside-toplevel
  seed = getattr(getattr(env, "unwrapped", env), "seed", None)
  if isinstance(seed, bool) or not isinstance(seed, int):
    seed = 0
  return (seed * 8191 + distributed.rank()) & 0x7FFFFFFFFFFFFFFF


def numpy_like_input(observations):
  return isinstance(observations, np.ndarray) or (
      isinstance(observations, torch.Tensor) and not observations.is_cuda)
