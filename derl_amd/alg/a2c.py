"""Advantage Actor-Critic (derl/alg/a2c.py:7-91) on the fused loss kernels."""
from .common import Alg
from .ppo import ActorCriticDeviceLoss
from .. import summary


class A2CLoss(ActorCriticDeviceLoss):
  """A2C loss (derl/alg/a2c.py:7-79): ``-mean(logp*A) - entropy_coef*H +
  value_loss_coef*mean((v - vt)^2)``."""
  mode = 1

  def __init__(self, policy, value_loss_coef=0.25, entropy_coef=0.01, name=None):
    super().__init__(policy, name=name)
    self.value_loss_coef = value_loss_coef
    self.entropy_coef = entropy_coef

  def policy_loss(self, trajectory, act=None):
    del act
    terms, _ = self._evaluate(trajectory, None, self.value_loss_coef, self.entropy_coef)
    return terms[1] - self.entropy_coef * terms[2]

  def value_loss(self, trajectory, act=None):
    del act
    terms, _ = self._evaluate(trajectory, None, self.value_loss_coef, self.entropy_coef)
    return terms[3]

  def epoch_arguments(self):
    """Hyper-parameters of the fused loss kernels for a native update (Trainer.step)."""
    return dict(mode=1, cliprange=None, value_loss_coef=self.value_loss_coef,
                entropy_coef=self.entropy_coef)

  def evaluate_native(self, data):
    """(loss scalar on the device, model-backward closure); see PPOLoss.evaluate_native."""
    terms, backward_fn = self._evaluate(data, None, self.value_loss_coef, self.entropy_coef)
    if summary.should_record():
      self._summaries(terms)
    self.call_count += 1
    return terms[0], backward_fn

  def __call__(self, data):
    terms, backward_fn = self._evaluate(data, None, self.value_loss_coef, self.entropy_coef)
    if summary.should_record():
      self._summaries(terms)
    self.call_count += 1
    return self._loss_tensor(terms, backward_fn)


class A2C(Alg):
  """Advantage Actor Critic (derl/alg/a2c.py:82-91)."""
  def __init__(self, runner, trainer, value_loss_coef=0.25, entropy_coef=0.01, name=None):
    loss_fn = A2CLoss(runner.policy, value_loss_coef=value_loss_coef,
                      entropy_coef=entropy_coef, name=name)
    super().__init__(runner, trainer, loss_fn, name=name)
