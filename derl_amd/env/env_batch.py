"""Batched host environments with derl's contract (derl/env/env_batch.py:8-199): a batch steps
every env with its own action, resets an env the moment it reports ``done`` (the returned
observation is then the first one of the next episode) and stacks the results.

``ParallelEnvBatch`` is rebuilt for the device-resident rollout: worker processes write their
observations straight into a double-buffered shared-memory block (one row per env) and only the
small things -- reward, done flag, info dict -- travel through the pipes.  ``step`` keeps the
reference's contract (fresh arrays); ``step_shared`` hands out the shared rows without a copy so
that ``HostEnvBridge`` (bridge.py) can DMA them into the rollout buffer in HBM.

Works with any env object that has ``reset()``, ``step(action)``, ``observation_space`` and
``action_space`` (gym is not required).
"""
import multiprocessing as mp
import os
import pickle
import sys
from multiprocessing import shared_memory

import numpy as np

from .spaces import Space


class SpaceBatch(Space):
  """Identical spaces of the members of an env batch (env_batch.py:8-32)."""
  def __init__(self, spaces):
    first = spaces[0]
    for space in spaces:
      if not isinstance(space, type(first)):
        raise TypeError(f"spaces have different types: {type(first)}, {type(space)}")
      if first.shape != space.shape:
        raise ValueError(f"spaces have different shapes: {first.shape}, {space.shape}")
      if first.dtype != space.dtype:
        raise ValueError(f"spaces have different data types: {first.dtype}, {space.dtype}")
    self.spaces = spaces
    super().__init__(shape=first.shape, dtype=first.dtype)

  def sample(self):
    return np.stack([space.sample() for space in self.spaces])

  def __getattr__(self, attr):
    if attr == "spaces":  # not constructed yet (copy / pickle)
      raise AttributeError(attr)
    return getattr(self.spaces[0], attr)


def _make_env_functions(make_env, nenvs):
  """Argument rules of EnvBatch.__init__ (env_batch.py:43-52)."""
  if nenvs is None and not isinstance(make_env, list):
    raise ValueError("When nenvs is None make_env must be a list of callables")
  if nenvs is not None and not callable(make_env):
    raise ValueError("When nenvs is not None make_env must be callable")
  return [make_env] * nenvs if nenvs is not None else make_env


class EnvBatch:
  """Envs stepped one after another in this process (env_batch.py:35-86)."""
  def __init__(self, make_env, nenvs=None):
    self._envs = [fn() for fn in _make_env_functions(make_env, nenvs)]
    self._nenvs = len(self._envs)
    self.observation_space = SpaceBatch([env.observation_space for env in self._envs])
    self.action_space = SpaceBatch([env.action_space for env in self._envs])

  @property
  def unwrapped(self):
    return self

  @property
  def nenvs(self):
    return self._nenvs

  @property
  def envs(self):
    return self._envs

  def _check_actions(self, actions):
    if len(actions) != self.nenvs:
      raise ValueError("number of actions is not equal to number of envs: "
                       f"len(actions) = {len(actions)}, nenvs = {self.nenvs}")

  def step(self, actions):
    self._check_actions(actions)
    obs, rews, resets, infos = [], [], [], []
    for env, action in zip(self.envs, actions):
      ob, rew, done, info = env.step(action)
      if done:
        ob = env.reset()
      obs.append(ob)
      rews.append(rew)
      resets.append(done)
      infos.append(info)
    return np.stack(obs), np.stack(rews), np.stack(resets), infos

  def reset(self):
    return np.stack([env.reset() for env in self.envs])

  def close(self):
    for env in self.envs:
      close = getattr(env, "close", None)
      if close is not None:
        close()


class SingleEnvBatch(EnvBatch):
  """One env presented as a batch of one (env_batch.py:89-111)."""
  def __init__(self, env):  # pylint: disable=super-init-not-called
    self.env = env
    self._envs = [env]
    self._nenvs = 1
    self.observation_space = SpaceBatch([env.observation_space])
    self.action_space = SpaceBatch([env.action_space])

  def step(self, actions):
    self._check_actions(actions)
    ob, rew, done, info = self.env.step(actions[0])
    if done:
      ob = self.env.reset()
    return np.asarray(ob)[None], np.expand_dims(rew, 0), np.expand_dims(done, 0), [info]

  def reset(self):
    return np.asarray(self.env.reset())[None]


class _Shipped:
  """An env factory on its way to a worker started WITHOUT fork: factories are usually closures /
  lambdas, which the standard pickle refuses, so they travel as cloudpickle bytes (by value) when
  cloudpickle is importable and as a plain pickle otherwise (module-level callables only)."""
  def __init__(self, fn):
    self.fn = fn

  def __getstate__(self):
    try:
      import cloudpickle  # pylint: disable=import-outside-toplevel
      return cloudpickle.dumps(self.fn)
    except ImportError:
      return pickle.dumps(self.fn)

  def __setstate__(self, payload):
    self.fn = pickle.loads(payload)  # cloudpickle output is loadable by pickle

  def __call__(self):
    return self.fn()


def _gpu_in_use():
  """True once this process has initialised the GPU runtime.  Every HIP call of this package goes
  through torch-allocated tensors, so torch's own flag covers the native library too."""
  torch = sys.modules.get("torch")
  return bool(torch is not None and torch.cuda.is_initialized())


def worker_start_method():
  """How env workers are created.  ``fork`` (cheap; closures work as they are) ONLY while this
  process has not touched the GPU: a forked child of a GPU process inherits device objects whose
  release -- by a garbage-collection pass, a dropped reference, an unwinding exception -- calls
  into a HIP runtime that does not survive fork (a recorded segfault, round 1).  Afterwards
  ``forkserver``: workers are forked from a clean server process that never saw the GPU and only
  receive their (pickled) factory.  ``DERL_AMD_ENV_START_METHOD`` overrides the choice."""
  forced = os.environ.get("DERL_AMD_ENV_START_METHOD")
  if forced:
    return forced
  return "forkserver" if _gpu_in_use() else "fork"


def _worker(conn, make_env, index):
  """Env process: observations go to shared memory, the rest through the pipe."""
  env = make_env()
  conn.send((env.observation_space, env.action_space))
  name, nenvs, shape, dtype = conn.recv()
  block = shared_memory.SharedMemory(name=name)
  slots = np.ndarray((2, nenvs) + tuple(shape), dtype=dtype, buffer=block.buf)
  try:
    while True:
      cmd, payload, slot = conn.recv()
      if cmd == "step":
        ob, rew, done, info = env.step(payload)
        if done:
          ob = env.reset()
        slots[slot, index] = ob
        conn.send((rew, done, info))
      elif cmd == "reset":
        slots[slot, index] = env.reset()
        conn.send(None)
      elif cmd == "close":
        close = getattr(env, "close", None)
        if close is not None:
          close()
        break
      else:
        raise NotImplementedError(f"Unknown command {cmd}")
  finally:
    del slots
    block.close()
    conn.close()


class ParallelEnvBatch(EnvBatch):
  """One process per env (env_batch.py:137-199); observations through shared memory.
  ``start_method`` None picks ``worker_start_method()``."""
  def __init__(self, make_env, nenvs=None, start_method=None):  # pylint: disable=super-init-not-called
    functions = _make_env_functions(make_env, nenvs)
    self._nenvs = len(functions)
    self.start_method = start_method or worker_start_method()
    if self.start_method == "fork" and _gpu_in_use():
      raise RuntimeError("refusing to fork env workers from a process that has initialised the GPU "
                         "(inherited device objects crash the child); use forkserver or spawn")
    ctx = mp.get_context(self.start_method)
    if self.start_method == "forkserver":
      # the server imports this module once; every worker is then a fork of the server
      ctx.set_forkserver_preload([__name__])
    if self.start_method != "fork":
      functions = [_Shipped(fn) for fn in functions]
    self._conns, self._processes = [], []
    for index, fn in enumerate(functions):
      parent, child = ctx.Pipe()
      proc = ctx.Process(target=_worker, args=(child, fn, index), daemon=True)
      proc.start()
      child.close()
      self._conns.append(parent)
      self._processes.append(proc)
    self._closed = False
    spaces = [conn.recv() for conn in self._conns]
    self.observation_space = SpaceBatch([s[0] for s in spaces])
    self.action_space = SpaceBatch([s[1] for s in spaces])
    shape, dtype = tuple(self.observation_space.shape), np.dtype(self.observation_space.dtype)
    nbytes = max(1, 2 * self._nenvs * int(np.prod(shape, dtype=np.int64)) * dtype.itemsize)
    self._block = shared_memory.SharedMemory(create=True, size=nbytes)
    self._slots = np.ndarray((2, self._nenvs) + shape, dtype=dtype, buffer=self._block.buf)
    self._slot = 0
    for conn in self._conns:
      conn.send((self._block.name, self._nenvs, shape, dtype.str))

  @property
  def envs(self):
    raise AttributeError("the envs of a ParallelEnvBatch live in other processes")

  @property
  def shared_slots(self):
    """The double buffer ``(2, nenvs, *obs_shape)`` (for pinning by HostEnvBridge)."""
    return self._slots

  def step_shared(self, actions):
    """Like ``step`` but returns the shared-memory rows: valid until the step after next."""
    self._check_actions(actions)
    self._slot ^= 1
    for conn, action in zip(self._conns, actions):
      conn.send(("step", action, self._slot))
    rews, dones, infos = zip(*[conn.recv() for conn in self._conns])
    return self._slots[self._slot], np.stack(rews), np.stack(dones), infos

  def step(self, actions):
    obs, rews, dones, infos = self.step_shared(actions)
    return obs.copy(), rews, dones, infos

  def reset_shared(self):
    self._slot ^= 1
    for conn in self._conns:
      conn.send(("reset", None, self._slot))
    for conn in self._conns:
      conn.recv()
    return self._slots[self._slot]

  def reset(self):
    return self.reset_shared().copy()

  def close(self):
    if self._closed:
      return
    self._closed = True
    for conn in self._conns:
      try:
        conn.send(("close", None, 0))
      except (BrokenPipeError, OSError):
        pass
    for proc in self._processes:
      proc.join(timeout=5)
      if proc.is_alive():
        proc.terminate()
    self._slots = None
    self._block.close()
    try:
      self._block.unlink()
    except FileNotFoundError:
      pass

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def render(self):
    raise ValueError(f"render not defined for {self}")
