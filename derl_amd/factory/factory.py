"""Building algorithms from flat hyper-parameter dictionaries -- the API of
derl/factory/factory.py (``KwargsDict``, ``Factory.make / from_args / from_default_kwargs /
get_kwargs``) on a different mechanism: arguments live in a stack of layers (a ChainMap; every
``override_context`` pushes one) next to a ledger of the keys that have been read, so "was every
argument consumed?" (factory.py:119-126) is a set difference and leaving a context is a pop.
"""
from abc import ABC, abstractmethod
from collections import ChainMap
from contextlib import contextmanager

from ..scripts.parsers import get_defaults_parser


class KwargsDict:
  """Keyword arguments with a record of which ones were looked at."""
  def __init__(self, **kwargs):
    self._layers = ChainMap(dict(kwargs))
    self._read = set()

  # -- views ---------------------------------------------------------------------------
  @property
  def kwargs(self):
    """The arguments visible right now (innermost override wins)."""
    return dict(self._layers)

  @property
  def unused(self):
    return set(self._layers) - self._read

  # -- lookups (each one marks its key as used) ------------------------------------------
  def has_arg(self, key):
    self._read.add(key)
    return key in self._layers

  def get_arg(self, key):
    self._read.add(key)
    return self._layers[key]

  def get_arg_default(self, key, default=None):
    return self.get_arg(key) if key in self._layers else default

  def get_arg_list(self, *keys):
    return list(map(self.get_arg, keys))

  def get_arg_dict(self, *keys, check_exists=True):
    wanted = [key for key in keys if self.has_arg(key)] if check_exists else keys
    return dict(zip(wanted, map(self.get_arg, wanted)))

  # -- scoped overrides ------------------------------------------------------------------
  @contextmanager
  def override_context(self, **kwargs):
    """Arguments that exist (and must be consumed) only inside the ``with`` block."""
    self._layers = self._layers.new_child(dict(kwargs))
    self._read -= set(kwargs)
    try:
      yield
    finally:
      leftover = set(kwargs) - self._read
      if leftover:  # raised with the layer still in place, as the reference leaves its dict
        raise ValueError(f"override_context: the block never read {sorted(leftover)}")
      self._layers = self._layers.parents
      self._read &= set(self._layers)  # keys that only lived in the popped layer are forgotten

  def reset_unused(self):
    self._read.clear()


class Factory(ABC):
  """Builds runner -> trainer -> algorithm from one bag of keyword arguments."""
  def __init__(self, *, ignore_unused=None, **kwargs):
    self.kwargs = KwargsDict(**kwargs)
    self.ignore_unused = frozenset(ignore_unused or ())

  def __getattr__(self, name):  # has_arg / get_arg / override_context ... of the bag
    if name == "kwargs":
      raise AttributeError(name)
    return getattr(self.kwargs, name)

  # -- defaults and parsing ----------------------------------------------------------------
  @staticmethod
  @abstractmethod
  def get_parser_defaults(args_type="atari"):
    """``{"flag-name": default | argparse-kwargs}`` for the env family, or None."""

  @staticmethod
  def make_env_kwargs(env_id):
    del env_id
    return {}

  @classmethod
  def _parse(cls, args_type, argv):
    return vars(get_defaults_parser(cls.get_parser_defaults(args_type)).parse_args(argv))

  @classmethod
  def get_kwargs(cls, args_type="atari"):
    return cls._parse(args_type, [])

  @classmethod
  def from_default_kwargs(cls, args_type="atari", ignore_unused=None, **kwargs):
    return cls(ignore_unused=ignore_unused, **{**cls.get_kwargs(args_type), **kwargs})

  @classmethod
  def from_args(cls, args_type="atari", ignore_unused=None, args=None):
    return cls(ignore_unused=ignore_unused, **cls._parse(args_type, args))

  # -- construction ------------------------------------------------------------------------
  @abstractmethod
  def make_runner(self, env, nlogs=1e5, **kwargs):
    """Creates the (wrapped) runner."""

  @abstractmethod
  def make_trainer(self, runner, **kwargs):
    """Creates the trainer."""

  @abstractmethod
  def make_alg(self, runner, trainer, **kwargs):
    """Creates the algorithm from its runner and trainer."""

  def make(self, env, nlogs=1e5, check_kwargs=True, **kwargs):
    with self.override_context(**kwargs):
      runner = self.make_runner(env, nlogs=nlogs)
      alg = self.make_alg(runner, self.make_trainer(runner))
      stray = self.kwargs.unused - self.ignore_unused
      if check_kwargs and stray:
        raise ValueError(f"these keyword arguments were given but never used: {sorted(stray)}; list "
                         "them in ignore_unused= of the factory or pass check_kwargs=False to make()")
    self.kwargs.reset_unused()
    return alg
