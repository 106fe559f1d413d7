"""Annealing variables (derl/anneal.py:14-86).  ``LinearAnneal.step_to`` uses the closed
form of the reference's per-step loop -- bit-equal (float32 of the float64 expression,
clamped), without 32,768 Python iterations per rollout."""
import re
from abc import ABC, abstractmethod

import torch

from . import summary


def camel2snake(string):
  sub = re.sub('(.)([A-Z][a-z]+)', r'\1_\2', string)
  return re.sub('([a-z0-9])([A-Z])', r'\1_\2', sub).lower()


class AnnealingVariable(ABC):
  """Variable the value of which changes after each step (anneal.py:14-43)."""
  def __init__(self, name=None):
    self.name = name or camel2snake(self.__class__.__name__)
    self.step_count = 0

  @abstractmethod
  def get_tensor(self):
    """Returns the torch.Tensor that changes after each call to step."""

  def get_current_value(self):
    return self.get_tensor().clone()

  @abstractmethod
  def step(self):
    """Updates the value of the variable."""

  def step_to(self, val):
    if val < self.step_count:
      raise ValueError(f"val={val} cannot be smaller than "
                       f"self.step_count={self.step_count}")
    for _ in range(val - self.step_count):
      self.step()

  def summarize(self, global_step):
    summary.add_scalar(f"anneal/{self.name}", self.get_tensor(), global_step=global_step)


class LinearAnneal(AnnealingVariable):
  """Linearly annealing variable (anneal.py:65-86)."""
  def __init__(self, start, nsteps, end=0., name=None):
    super().__init__(name)
    self.start = start
    self.nsteps = nsteps
    self.end = end
    self.tensor = torch.tensor(self.start)  # 0-dim float32 on the host, like the reference

  def get_tensor(self):
    return self.tensor

  def _value_at(self, step_count):
    step_frac = step_count / self.nsteps
    return torch.clamp(torch.tensor(self.start + (self.end - self.start) * step_frac),
                       min(self.start, self.end), max(self.start, self.end))

  def step(self):
    self.step_count += 1
    self.tensor.data = self._value_at(self.step_count)
    return self.get_current_value()

  def step_to(self, val):
    if val < self.step_count:
      raise ValueError(f"val={val} cannot be smaller than "
                       f"self.step_count={self.step_count}")
    if val > self.step_count:  # the loop's last iteration is all that survives
      self.step_count = val
      self.tensor.data = self._value_at(val)
