"""Minimal observation / action spaces (gym is not a dependency of the hot path).  Real
``gym.spaces`` objects are accepted everywhere these are: only ``shape``, ``dtype`` and
``n`` are read (derl/models.py:281-298)."""
import numpy as np


class Space:
  def __init__(self, shape=None, dtype=None):
    self.shape = None if shape is None else tuple(shape)
    self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
  def __init__(self, low, high, shape=None, dtype=np.float32):
    if shape is None:
      shape = np.shape(low)
    super().__init__(shape, dtype)
    self.low, self.high = low, high


class Discrete(Space):
  def __init__(self, n):
    super().__init__((), np.int64)
    self.n = int(n)


def is_discrete(space):
  return hasattr(space, "n") and not getattr(space, "shape", ())


def is_box(space):
  return not is_discrete(space) and getattr(space, "shape", None) is not None
