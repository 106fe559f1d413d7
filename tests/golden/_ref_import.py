"""Import the unmodified reference (mknbv/derl at /root/reference) in THIS container only.

Test tooling, not product code: used by ``tests/golden/generate.py`` to produce the
golden vectors committed under ``tests/golden/`` and by the optional cross-check tests
that are skipped wherever ``/root/reference`` does not exist (e.g. on the GPU box).

The reference needs ``gym``, ``atari_py``, ``cv2`` and ``tensorboard`` at import time
(derl/__init__.py:2 -> env/env_batch.py:4, env/make_env.py:8, env/atari_wrappers.py:4,
summary.py:4).  None is installed and none is on the hot path, so inert stand-ins are
registered in ``sys.modules`` before the import.  Nothing from the reference is copied.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, "derl"))


class _Space:
  def __init__(self, shape=None, dtype=None):
    self.shape = None if shape is None else tuple(shape)
    self.dtype = dtype


class _Box(_Space):
  def __init__(self, low, high, shape=None, dtype="float32"):
    import numpy as np
    if shape is None:
      shape = np.shape(low)
    super().__init__(shape, np.dtype(dtype))
    self.low, self.high = low, high


class _Discrete(_Space):
  def __init__(self, n):
    import numpy as np
    super().__init__((), np.dtype("int64"))
    self.n = n


class _Env:
  metadata = {}
  reward_range = (-float("inf"), float("inf"))
  spec = None
  action_space = None
  observation_space = None

  @property
  def unwrapped(self):
    return self


class _Wrapper(_Env):
  def __init__(self, env):
    self.env = env
    self.action_space = getattr(env, "action_space", None)
    self.observation_space = getattr(env, "observation_space", None)

  @property
  def unwrapped(self):
    return self.env.unwrapped

  def __getattr__(self, name):
    return getattr(self.env, name)


def install_stubs():
  """Registers the stand-in third-party modules (idempotent)."""
  if "gym" in sys.modules and getattr(sys.modules["gym"], "_derl_amd_stub", False):
    return
  gym = types.ModuleType("gym")
  gym._derl_amd_stub = True
  gym.Env, gym.Wrapper, gym.Space = _Env, _Wrapper, _Space
  gym.ObservationWrapper = type("ObservationWrapper", (_Wrapper,), {})
  gym.RewardWrapper = type("RewardWrapper", (_Wrapper,), {})
  gym.ActionWrapper = type("ActionWrapper", (_Wrapper,), {})
  gym.make = lambda *a, **k: (_ for _ in ()).throw(
      RuntimeError("gym is a stand-in here; real envs are unavailable"))
  spaces = types.ModuleType("gym.spaces")
  spaces.Box, spaces.Discrete, spaces.Space = _Box, _Discrete, _Space
  gym.spaces = spaces
  envs = types.ModuleType("gym.envs")
  atari = types.ModuleType("gym.envs.atari")
  atari.AtariEnv = type("AtariEnv", (_Env,), {})
  envs.atari = atari
  gym.envs = envs
  atari_py = types.ModuleType("atari_py")
  atari_py.list_games = lambda: ["breakout", "space_invaders", "pong"]
  cv2 = types.ModuleType("cv2")
  cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda flag: None)
  cv2.INTER_AREA = 3
  cv2.COLOR_RGB2GRAY = 7
  tb = types.ModuleType("tensorboard")

  class _SummaryWriter:
    def __init__(self, *a, **k):
      pass

    def add_scalar(self, *a, **k):
      pass

  import torch.utils  # noqa: F401  (parent package must exist first)
  tut = types.ModuleType("torch.utils.tensorboard")
  tut.SummaryWriter = _SummaryWriter
  for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs),
                    ("gym.envs.atari", atari), ("atari_py", atari_py),
                    ("cv2", cv2), ("tensorboard", tb),
                    ("torch.utils.tensorboard", tut)]:
    sys.modules[name] = mod


def import_reference():
  """Returns the imported reference package ``derl`` (from /root/reference)."""
  if not reference_available():
    raise RuntimeError("reference tree not present (expected on the GPU box)")
  sys.dont_write_bytecode = True
  install_stubs()
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  stale = sys.modules.get("derl")
  if stale is not None and not getattr(stale, "__file__", "").startswith(REFERENCE_ROOT):
    # the repo's own `derl` alias package (import name of derl_amd) must not shadow the reference
    for name in [n for n in sys.modules if n == "derl" or n.startswith("derl.")]:
      del sys.modules[name]
  import derl  # pylint: disable=import-error
  assert derl.__file__.startswith(REFERENCE_ROOT), derl.__file__
  import derl.summary as summary
  summary.stop_recording()
  summary.should_record = lambda *a, **k: False
  return derl
