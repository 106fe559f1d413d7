"""Env reward summaries (derl/env/summarize.py:8-91).

``RewardSummarizer`` / ``Summarize`` keep the reference's contract for HOST envs (per-step
``step(rewards, resets)``, tags ``<prefix>/total_reward``, ``episode_length``, ``min_reward``,
``max_reward``, ``reward_mean_<k>``; the ``real_done`` info key overrides ``done``).
``DeviceSummarize`` wraps a device-resident env: its rewards and resets never leave HBM, so the
statistics of a whole rollout are advanced by ONE native call (``dx_reward_summary_f32``) from the
runner's ``rollout_done`` hook, and only the emitted summary rows (if any) are read back.
"""
import numpy as np
import torch

from .. import _lib, summary

TAGS = ("total_reward", "episode_length", "min_reward", "max_reward")


class RewardSummarizer:
  """Summarizes rewards received from a (batched) host environment."""
  def __init__(self, nenvs, prefix, running_mean_size=100):
    self.prefix = prefix
    self.step_count = 0
    self.size = int(running_mean_size)
    self.had_ended_episodes = np.zeros(nenvs, dtype=bool)
    self.rewards = np.zeros(nenvs)
    self.episode_lengths = np.zeros(nenvs)
    self.queue = np.zeros((nenvs, self.size))  # ring of the last `size` episode rewards
    self.qlen = np.zeros(nenvs, dtype=np.int64)
    self.qpos = np.zeros(nenvs, dtype=np.int64)

  def should_add_summaries(self):
    return summary.should_record() and bool(np.all(self.had_ended_episodes))

  def _push(self, rows):
    self.queue[rows, self.qpos[rows]] = self.rewards[rows]
    self.qpos[rows] = (self.qpos[rows] + 1) % self.size
    self.qlen[rows] = np.minimum(self.qlen[rows] + 1, self.size)
    self.rewards[rows] = 0
    self.had_ended_episodes[rows] = True

  def summaries(self):
    """The five statistics of add_summaries (summarize.py:25-38)."""
    rows = np.arange(self.queue.shape[0])
    last = self.queue[rows, (self.qpos - 1) % self.size]
    valid = np.arange(self.size)[None, :] < self.qlen[:, None]
    means = (self.queue * valid).sum(1) / self.qlen
    out = dict(total_reward=np.mean(last), episode_length=np.mean(self.episode_lengths),
               min_reward=last.min(), max_reward=last.max())
    out[f"reward_mean_{self.size}"] = np.mean(means)
    return out

  def add_summaries(self):
    for key, val in self.summaries().items():
      summary.add_scalar(f"{self.prefix}/{key}", val, global_step=self.step_count)

  def _maybe_summarize(self):
    if self.should_add_summaries():
      self.add_summaries()
      self.episode_lengths.fill(0)
      self.had_ended_episodes.fill(False)

  def step(self, rewards, resets):
    self.rewards += rewards
    self.episode_lengths[~self.had_ended_episodes] += 1
    rows = np.nonzero(np.asarray(resets).reshape(-1))[0]
    if rows.size:
      self._push(rows)
    self.step_count += self.rewards.shape[0]
    self._maybe_summarize()

  def reset(self):
    rows = np.nonzero(self.episode_lengths)[0]
    if rows.size:
      self._push(rows)
    self._maybe_summarize()


class Summarize:
  """Writes env summaries for a host env (summarize.py:66-91)."""
  def __init__(self, env, summarizer):
    self.env = env
    self.summarizer = summarizer
    self.observation_space = getattr(env, "observation_space", None)
    self.action_space = getattr(env, "action_space", None)

  @classmethod
  def reward_summarizer(cls, env, prefix=None, running_mean_size=100):
    nenvs = getattr(env.unwrapped, "nenvs", None) or 1
    prefix = prefix if prefix is not None else env.spec.id
    return cls(env, RewardSummarizer(nenvs, prefix, running_mean_size=running_mean_size))

  @property
  def unwrapped(self):
    return self.env.unwrapped

  def __getattr__(self, name):
    if name == "env":
      raise AttributeError(name)
    return getattr(self.env, name)

  def step(self, action):
    obs, rew, done, info = self.env.step(action)
    infos = [info] if isinstance(info, dict) else info
    dones = [done] if isinstance(done, bool) else done
    resets = np.asarray([(i or {}).get("real_done", dones[k]) for k, i in enumerate(infos)]
                        if infos is not None else dones)
    self.summarizer.step(rew, resets)
    return obs, rew, done, info

  def reset(self, **kwargs):
    self.summarizer.reset()
    return self.env.reset(**kwargs)


class DeviceSummarize:
  """Reward summaries of a device-resident env; statistics advance once per rollout."""
  def __init__(self, env, prefix, running_mean_size=100, max_rows=64):
    device = getattr(env, "device", None)
    if device is None:
      raise TypeError("DeviceSummarize wraps a device-resident env; use Summarize for host envs")
    self.env, self.prefix, self.size = env, prefix, int(running_mean_size)
    self.device = torch.device(device)
    self.nenvs = env.unwrapped.nenvs
    self.observation_space, self.action_space = env.observation_space, env.action_space
    n, dev = self.nenvs, self.device
    f64 = dict(dtype=torch.float64, device=dev)
    self.acc, self.ep_len = torch.zeros(n, **f64), torch.zeros(n, **f64)
    self.ended = torch.zeros(n, dtype=torch.uint8, device=dev)
    self.queue = torch.zeros((n, self.size), **f64)
    self.qlen = torch.zeros(n, dtype=torch.int32, device=dev)
    self.qpos = torch.zeros(n, dtype=torch.int32, device=dev)
    self.step_count_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    self.rows = torch.zeros((max_rows, 6), **f64)
    self.nrows = torch.zeros(1, dtype=torch.int32, device=dev)
    self.max_rows = max_rows

  @property
  def unwrapped(self):
    return self.env.unwrapped

  def __getattr__(self, name):
    if name == "env":
      raise AttributeError(name)
    return getattr(self.env, name)

  def reset(self, *args, **kwargs):
    return self.env.reset(*args, **kwargs)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    """Pass-through: the statistics advance in ``rollout_done``, which the device-resident
    EnvRunner calls once per rollout with the (T, N) reward / reset buffers."""
    return self.env.step(actions, out=out, rewards_out=rewards_out, resets_out=resets_out)

  def rollout_done(self, rewards, resets):
    """Advance the statistics over (T, N) rewards / resets; emits the rows the reference would
    have written step by step (summary.add_scalar with the same tags and global steps)."""
    rewards = rewards.to(torch.float32).contiguous()
    resets = (resets.view(torch.uint8) if resets.dtype == torch.bool else resets).contiguous()
    record = summary.should_record()
    T = rewards.shape[0]
    _lib.call("dx_reward_summary_f32", _lib.ptr(rewards), _lib.ptr(resets), T, self.nenvs, self.size,
              int(record), _lib.ptr(self.acc), _lib.ptr(self.ep_len), _lib.ptr(self.ended),
              _lib.ptr(self.queue), _lib.ptr(self.qlen), _lib.ptr(self.qpos),
              _lib.ptr(self.step_count_dev), _lib.ptr(self.rows), self.max_rows, _lib.ptr(self.nrows),
              _lib.stream_ptr(self.device))
    if record:  # one small read-back per recorded rollout
      count = int(self.nrows.item())
      if count:
        rows = self.rows[:count].cpu().numpy()
        self.nrows.zero_()
        for row in rows:
          for k, tag in enumerate(TAGS):
            summary.add_scalar(f"{self.prefix}/{tag}", row[k], global_step=int(row[5]))
          summary.add_scalar(f"{self.prefix}/reward_mean_{self.size}", row[4], global_step=int(row[5]))
