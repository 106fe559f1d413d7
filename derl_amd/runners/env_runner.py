"""Environment runner (derl/runners/env_runner.py:6-89).

Two data paths behind one interface:
 * device-resident (env exposes ``device`` and writes its step into a given buffer, policy
   exposes ``act_into``): observations, actions, log-probs, values, rewards and resets of
   a whole rollout live in preallocated time-major HBM buffers ``(T[+1], N, ...)``; nothing
   is stacked or copied afterwards and ``next_observations`` is the same storage shifted by
   one step (batched envs auto-reset, env_runner.py:58-65);
 * generic (any env / policy with the reference's contract): per-key Python lists exactly
   as the reference builds them (env_runner.py:42-57).
"""
from abc import ABC, abstractmethod
from collections import defaultdict

import torch


class EnvRunner:
  """Iterable that interacts with an env."""
  def __init__(self, env, policy, horizon, nsteps=None, time_limit=None):
    self.env = env
    self.policy = policy
    self.horizon = horizon
    self.nsteps = int(nsteps)
    if (time_limit is not None
        and getattr(self.env.unwrapped, "nenvs", None) is not None):
      raise TypeError("batched envs are not supported for time_limit "
                      f"not equal to None, got env={self.env}, "
                      f"time_limit={time_limit}")
    self.time_limit = time_limit
    self.step_count = 0
    self.episode_length = 0
    self._buffers = None

  @property
  def nenvs(self):
    return getattr(self.env.unwrapped, "nenvs", None)

  def is_exhausted(self):
    return self.nsteps is not None and self.step_count >= self.nsteps

  def __len__(self):
    return self.nsteps if self.nsteps is not None else self.step_count

  def _device_resident(self):
    return (self.nenvs is not None and getattr(self.env, "device", None) is not None
            and hasattr(self.policy, "act_into"))

  def run(self, obs=None):
    """Interacts with the environment starting from obs for horizon steps."""
    if self._device_resident():
      yield from self._run_device(obs)
    else:
      yield from self._run_generic(obs)

  # ---- device-resident rollout -------------------------------------------------------
  def _allocate(self):
    T, N = self.horizon, self.nenvs
    dev = self.env.device
    ospace, aspace = self.env.observation_space, self.env.action_space
    obs_dtype = torch.uint8 if str(ospace.dtype) == "uint8" else torch.float32
    discrete = hasattr(aspace, "n")
    ashape = () if discrete else tuple(aspace.shape)
    self._buffers = dict(
        obs=torch.empty((T + 1, N) + tuple(ospace.shape), dtype=obs_dtype, device=dev),
        actions=torch.empty((T, N) + ashape, dtype=torch.int64 if discrete else torch.float32,
                            device=dev),
        log_prob=torch.empty((T, N), dtype=torch.float32, device=dev),
        values=torch.empty((T, N, 1), dtype=torch.float32, device=dev),
        rewards=torch.empty((T, N), dtype=torch.float32, device=dev),
        resets=torch.empty((T, N), dtype=torch.bool, device=dev))

  def _run_device(self, obs):
    if self._buffers is None:
      self._allocate()
    buf = self._buffers
    T = self.horizon
    if obs is None:
      self.env.reset(out=buf["obs"][0])
    else:
      buf["obs"][0].copy_(obs)
    fused = getattr(self.policy, "rollout_into", None)
    while not self.is_exhausted():
      if fused is None or not fused(self.env, buf, T):
        for t in range(T):
          self.policy.act_into(buf["obs"][t], buf["actions"][t], buf["log_prob"][t],
                               buf["values"][t])
          self.env.step(buf["actions"][t], out=buf["obs"][t + 1],
                        rewards_out=buf["rewards"][t], resets_out=buf["resets"][t])
      interactions = dict(
          observations=buf["obs"][:T], actions=buf["actions"], log_prob=buf["log_prob"],
          values=buf["values"], rewards=buf["rewards"], resets=buf["resets"],
          infos=None, next_observations=buf["obs"][1:],
          state=dict(latest_observations=buf["obs"][T]))
      self.step_count += T * self.nenvs
      hook = getattr(self.env, "rollout_done", None)
      if hook is not None:  # env-side statistics over the whole rollout (DeviceSummarize)
        hook(buf["rewards"], buf["resets"])
      yield interactions
      buf["obs"][0].copy_(buf["obs"][T])

  # ---- generic rollout (reference contract) -------------------------------------------
  def _start_episode(self):
    self.episode_length = 0
    return self.env.reset()

  def _run_generic(self, obs):
    """Per-key Python lists of length ``horizon``, exactly the dict derl's runner yields
    (env_runner.py:42-57): every key ``policy.act`` returns, plus observations / rewards /
    resets / infos / next_observations and ``state.latest_observations``."""
    single = self.nenvs is None  # an unbatched env must be reset by the runner
    if obs is None:
      obs = self._start_episode()
    while not self.is_exhausted():
      record = defaultdict(list)
      for _ in range(self.horizon):
        decision = self.policy.act(obs)
        if "actions" not in decision:
          raise ValueError("result of policy.act must contain 'actions' "
                           f"but has keys {list(decision.keys())}")
        record["observations"].append(obs)
        for key, val in decision.items():
          record[key].append(val)
        obs, reward, done, info = self.env.step(decision["actions"])
        self.episode_length += 1
        for key, val in (("rewards", reward), ("resets", done), ("infos", info),
                         ("next_observations", obs)):
          record[key].append(val)
        if single and (done or self.episode_length == self.time_limit):
          obs = self._start_episode()
      record["state"] = dict(latest_observations=obs)
      self.step_count += self.horizon * (1 if single else self.nenvs)
      yield dict(record)


class RunnerWrapper(ABC):
  """Wraps an env runner (derl/runners/env_runner.py:72-89)."""
  def __init__(self, runner):
    self.runner = runner
    self.unwrapped = getattr(runner, "unwrapped", runner)

  def __getattr__(self, attr):
    if attr not in {"env", "policy", "horizon", "nsteps", "step_count",
                    "nenvs", "is_exhausted"}:
      raise AttributeError(f"'{self.__class__.__name__}' "
                           f"has no attribute '{attr}'")
    return getattr(self.runner, attr)

  def __len__(self):
    return len(self.runner)

  @abstractmethod
  def run(self, obs=None):
    """Interacts with the environment starting from obs for horizon steps."""
