"""Proximal Policy Optimization (derl/alg/ppo.py:8-123) on the fused loss kernels."""
import torch

from .common import Alg, Loss, _LossBackward
from .. import distributed, summary
from ..policies import refuse_mid_epoch


class ActorCriticDeviceLoss(Loss):
  """Shared machinery of PPOLoss / A2CLoss: one model forward on the minibatch, one fused
  loss forward+backward launch on the head outputs, and a scalar whose ``.backward()``
  runs the model backward."""
  mode = None

  def __init__(self, policy, name=None):
    super().__init__(model=policy.model, name=name)
    self.policy = policy
    self.last_terms = None  # device float32[8], see dx_categorical_loss_f32 / dx_normal_loss_f32

  def _check(self, trajectory, need_old):
    for key in ("advantages", "value_targets"):
      if key not in trajectory:
        raise ValueError(f"trajectory does not contain '{key}'")
    batch = trajectory["actions"].shape[0]
    log_prob_shape = (batch,)
    advantages = trajectory["advantages"]
    if tuple(advantages.shape) != log_prob_shape:
      raise ValueError("trajectory has mismatched shapes: "
                       f"log_prob.shape={log_prob_shape} "
                       f"advantages.shape={tuple(advantages.shape)}")
    if need_old and tuple(trajectory["log_prob"].shape) != log_prob_shape:
      raise ValueError("trajectory has mismatched shapes: "
                       f"log_prob.shape={log_prob_shape} "
                       f"old_log_prob.shape={tuple(trajectory['log_prob'].shape)}")
    if tuple(trajectory["value_targets"].shape) != (batch, 1):
      raise ValueError("trajectory has mismatched shapes "
                       f"values.shape={(batch, 1)} "
                       f"value_targets.shape={tuple(trajectory['value_targets'].shape)}")

  def _evaluate(self, data, cliprange, value_loss_coef, entropy_coef):
    """Forward + fused loss.  Returns (terms float32[8] on the device, backward closure)."""
    self._check(data, need_old=self.mode == 0)
    refuse_mid_epoch(self.model, "evaluating the loss")
    f32 = torch.float32
    batch = data["actions"].shape[0]
    global_batch = batch * distributed.world_size()
    terms, backward_fn = self.model.loss_forward_backward(
        self.policy, data, self.mode, cliprange, value_loss_coef, entropy_coef, global_batch,
        actions=self.to_device(data["actions"]),
        old_log_prob=self.to_device(data["log_prob"], f32) if self.mode == 0 else None,
        advantages=self.to_device(data["advantages"], f32),
        old_values=self.to_device(data["values"], f32).reshape(-1) if self.mode == 0 else None,
        value_targets=self.to_device(data["value_targets"], f32).reshape(-1))
    self.last_terms = terms
    return terms, backward_fn

  def native_update_arrays(self, data):
    """The device arrays of ONE minibatch as the engine's native update reads them (keys of
    ``EpochContext.shuffled`` plus ``observations`` / ``index``), after the same checks and
    uploads ``__call__`` makes; None when the observations are not a batch of frames."""
    from ..models import GatheredRows  # pylint: disable=import-outside-toplevel
    self._check(data, need_old=self.mode == 0)
    f32 = torch.float32
    observations, index = data["observations"], None
    if isinstance(observations, GatheredRows):
      observations, index = observations.base, observations.index
    observations = self.model.prepare(observations)
    if observations.ndim != 4:
      return None
    actions = self.to_device(data["actions"])
    arrays = dict(observations=observations, index=index,
                  actions=actions if actions.dtype == torch.int64 else actions.long(),
                  advantages=self.to_device(data["advantages"], f32).reshape(-1),
                  value_targets=self.to_device(data["value_targets"], f32).reshape(-1))
    if self.mode == 0:
      arrays["log_prob"] = self.to_device(data["log_prob"], f32).reshape(-1)
      arrays["values"] = self.to_device(data["values"], f32).reshape(-1)
    return arrays

  def _summaries(self, terms):
    tags = dict(policy_loss=1, entropy=2, value_loss=3, advantages=4, value_preds=5,
                value_targets=6, r_squared=7, loss=0)
    for key, i in tags.items():
      summary.add_scalar(f"{self.name}/{key}", terms[i], global_step=self.call_count)

  def _loss_tensor(self, terms, backward_fn):
    return _LossBackward.apply(self.model._anchor, terms[0], backward_fn)


class PPOLoss(ActorCriticDeviceLoss):
  """PPO loss (derl/alg/ppo.py:8-108): clipped-ratio policy loss with entropy bonus,
  clipped value loss; ``loss = policy - entropy_coef*H + value_loss_coef*value``."""
  mode = 0

  def __init__(self, policy, cliprange=0.2, value_loss_coef=0.25, entropy_coef=0.01, name=None):
    super().__init__(policy, name=name)
    self.cliprange = cliprange
    self.value_loss_coef = value_loss_coef
    self.entropy_coef = entropy_coef

  def policy_loss(self, trajectory, act=None):
    """Policy loss including entropy regularisation (ppo.py:24-64); a plain device scalar."""
    del act
    terms, _ = self._evaluate(trajectory, self.cliprange, self.value_loss_coef, self.entropy_coef)
    return terms[1] - self.entropy_coef * terms[2]

  def value_loss(self, trajectory, act=None):
    """Value loss (ppo.py:66-98); a plain device scalar."""
    del act
    terms, _ = self._evaluate(trajectory, self.cliprange, self.value_loss_coef, self.entropy_coef)
    return terms[3]

  def epoch_arguments(self):
    """Hyper-parameters of the fused loss kernels for a native epoch (Trainer.step)."""
    return dict(mode=0, cliprange=self.cliprange, value_loss_coef=self.value_loss_coef,
                entropy_coef=self.entropy_coef)

  def evaluate_native(self, data):
    """(loss scalar on the device, closure running the model backward): what ``__call__``
    wraps into an autograd scalar; Trainer.step calls the closure directly."""
    terms, backward_fn = self._evaluate(data, self.cliprange, self.value_loss_coef,
                                        self.entropy_coef)
    if summary.should_record():
      self._summaries(terms)
    self.call_count += 1
    return terms[0], backward_fn

  def __call__(self, data):
    terms, backward_fn = self._evaluate(data, self.cliprange, self.value_loss_coef,
                                        self.entropy_coef)
    if summary.should_record():
      self._summaries(terms)
    self.call_count += 1
    return self._loss_tensor(terms, backward_fn)


class PPO(Alg):
  """Proximal Policy Optimization algorithm (derl/alg/ppo.py:111-123)."""
  def __init__(self, runner, trainer, cliprange=0.2, value_loss_coef=0.25, entropy_coef=0.01,
               name=None):
    loss_fn = PPOLoss(runner.policy, cliprange=cliprange, value_loss_coef=value_loss_coef,
                      entropy_coef=entropy_coef, name=name)
    super().__init__(runner, trainer, loss_fn, name=name)
