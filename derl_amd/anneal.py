"""Schedules that move a 0-dim tensor as training progresses -- the contract of derl/anneal.py
(``AnnealingVariable``: ``get_tensor`` / ``step`` / ``step_to`` / ``summarize``;
``LinearAnneal(start, nsteps, end=0.)``, ``TorchSched(scheduler)``).

The tensor object is created once and mutated in place, so an optimizer that was given
``schedule.get_tensor()`` as its learning rate follows the schedule.  ``LinearAnneal.step_to`` jumps
straight to the target: the reference walks there one ``step()`` at a time (32,768 iterations per
rollout at the headline config) and only the last assignment survives; the value is the same
float32 rounding of the same float64 expression, so the jump is bit-equal (checked in
tests/test_host_logic.py against the recorded reference values).
"""
import re
from abc import ABC, abstractmethod

import torch

from . import summary

_WORD_STARTS = re.compile(r"(?<=[a-z0-9])(?=[A-Z])|(?<=[A-Z])(?=[A-Z][a-z])")


def camel2snake(string):
  """``LinearAnneal`` -> ``linear_anneal`` (default names of the variables)."""
  return _WORD_STARTS.sub("_", string).lower()


class AnnealingVariable(ABC):
  """A value that depends on how many steps have been taken."""
  def __init__(self, name=None):
    self.name = camel2snake(type(self).__name__) if name is None else name
    self.step_count = 0

  @abstractmethod
  def get_tensor(self):
    """The tensor whose value follows the schedule (always the same object)."""

  @abstractmethod
  def step(self):
    """Advances the schedule by one step."""

  def get_current_value(self):
    return self.get_tensor().clone()

  def _check_forward(self, target):
    if target < self.step_count:
      raise ValueError(f"val={target} cannot be smaller than self.step_count={self.step_count}")

  def step_to(self, val):
    """Advances to step ``val`` (schedules never run backwards)."""
    self._check_forward(val)
    while self.step_count < val:
      self.step()

  def summarize(self, global_step):
    summary.add_scalar(f"anneal/{self.name}", self.get_tensor(), global_step=global_step)


class TorchSched(AnnealingVariable):
  """Follows a ``torch.optim.lr_scheduler`` object (derl/anneal.py:46-62): the tensor holds the
  scheduler's last learning rates, one element per parameter group of the optimizer the scheduler
  drives.  (No factory uses it; the flat optimizers of this package take ``LinearAnneal``'s 0-dim
  tensor.  It is here so that code written against derl's annealing API keeps working.)"""
  def __init__(self, scheduler, name=None):
    super().__init__(name)
    self.scheduler = scheduler
    self.tensor = self._last_rates()

  def _last_rates(self):
    return torch.tensor(self.scheduler.get_last_lr())

  def get_tensor(self):
    return self.tensor

  def step(self):
    self.scheduler.step()
    self.step_count += 1
    self.tensor.data = self._last_rates()
    return self.get_current_value()


class LinearAnneal(AnnealingVariable):
  """``start`` -> ``end`` over ``nsteps`` steps, clamped to that interval afterwards."""
  def __init__(self, start, nsteps, end=0., name=None):
    super().__init__(name)
    self.start, self.end, self.nsteps = start, end, nsteps
    self.tensor = torch.tensor(start)  # host, float32 for a Python float: as in the reference
    self._low, self._high = min(start, end), max(start, end)

  def get_tensor(self):
    return self.tensor

  def _assign(self, step_count):
    progress = step_count / self.nsteps
    value = torch.tensor(self.start + (self.end - self.start) * progress)  # f64 math, f32 tensor
    self.step_count = step_count
    self.tensor.data = torch.clamp(value, self._low, self._high)

  def step(self):
    self._assign(self.step_count + 1)
    return self.get_current_value()

  def step_to(self, val):
    self._check_forward(val)
    if val != self.step_count:
      self._assign(val)
