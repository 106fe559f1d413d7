"""CPU port of one derl PPO iteration (rollout -> GAE -> minibatch updates) -- the
``cpu_baseline`` leg of bench.py and a whole-path cross-check for tests.  TEST
INFRASTRUCTURE: never imported by derl_amd/.

Same dataflow as the reference on its CPU path, restated from its behaviour:
  rollout    per-step Python loop, policy forward + Categorical sample, per-key lists
             (derl/runners/env_runner.py:41-69, derl/policies.py:51-80)
  stacking   list -> np.asarray for every key, incl. next_observations
             (derl/runners/onpolicy.py:20-27)
  GAE        Python loop over T, NumPy over N (derl/runners/trajectory_transforms.py:45-65)
  minibatch  per-epoch in-place permutation of every array, contiguous slices, per-minibatch
             advantage normalisation (derl/runners/onpolicy.py:44-62,74)
  update     forward, PPOLoss, autograd backward, clip_grad_norm_(0.5), LR anneal, Adam
             (derl/alg/ppo.py:100-108, derl/alg/common.py:66-78, derl/factory/ppo.py:74-83)
The environment is the synthetic zero-cost batched env of the survey's measurement: uint8
frames from a 4-deep pre-generated pool, rewards in {-1,0,1}, resets Bernoulli(0.01).
"""
import time

import numpy as np
import torch

from .gae import gae_advantages, merge_time_batch, normalize_advantages
from .models import init_nature_cnn, nature_cnn_forward
from .losses import ppo_loss_terms
from .distributions import categorical_log_prob_entropy
from .optim import linear_anneal


class SyntheticFramePool:
  def __init__(self, nenvs, seed=0, depth=4):
    rs = np.random.RandomState(seed)
    self.pool = rs.randint(0, 256, size=(depth, nenvs, 84, 84, 4)).astype(np.uint8)
    self.rs = rs
    self.nenvs = nenvs
    self.t = 0

  def reset(self):
    return self.pool[0]

  def step(self, actions):
    del actions
    self.t += 1
    u = self.rs.uniform(size=(3, self.nenvs))
    rewards = np.where(u[0] < 0.1, np.where(u[1] < 0.5, -1.0, 1.0), 0.0)
    return self.pool[self.t % len(self.pool)], rewards, u[2] < 0.01, [{}] * self.nenvs


class CpuPPO:
  """Holds torch-CPU parameters + Adam and runs whole iterations."""
  def __init__(self, nenvs=256, nsteps=128, num_actions=4, num_epochs=3, num_minibatches=4,
               gamma=0.99, lambda_=0.95, cliprange=0.1, value_loss_coef=0.25, entropy_coef=0.01,
               max_grad_norm=0.5, lr=2.5e-4, num_train_steps=10e6, eps=1e-5, seed=0, threads=None):
    if threads:
      torch.set_num_threads(threads)
    self.cfg = dict(nenvs=nenvs, nsteps=nsteps, num_epochs=num_epochs,
                    num_minibatches=num_minibatches, gamma=gamma, lambda_=lambda_,
                    cliprange=cliprange, value_loss_coef=value_loss_coef,
                    entropy_coef=entropy_coef, max_grad_norm=max_grad_norm, lr=lr,
                    num_train_steps=num_train_steps)
    self.params = {k: v.clone().requires_grad_(True)
                   for k, v in init_nature_cnn((num_actions, 1), seed=seed).items()}
    self.lr = torch.tensor(lr)
    self.optimizer = torch.optim.Adam(list(self.params.values()), lr=self.lr, eps=eps)
    self.env = SyntheticFramePool(nenvs, seed)
    self.obs = self.env.reset()
    self.step_count = 0
    self.losses = []

  def act(self, obs):
    logits, values = nature_cnn_forward(self.params, obs)  # builds and drops a graph (G3)
    dist = torch.distributions.Categorical(logits=logits)
    actions = dist.sample()
    return (actions.numpy(), dist.log_prob(actions).detach().numpy(), values.detach().numpy())

  def rollout(self):
    c = self.cfg
    inter = {k: [] for k in ("observations", "actions", "log_prob", "values", "rewards", "resets",
                             "next_observations")}
    obs = self.obs
    for _ in range(c["nsteps"]):
      actions, log_prob, values = self.act(obs)
      inter["observations"].append(obs)
      inter["actions"].append(actions)
      inter["log_prob"].append(log_prob)
      inter["values"].append(values)
      new_obs, rew, done, _ = self.env.step(actions)
      inter["rewards"].append(rew)
      inter["resets"].append(done)
      inter["next_observations"].append(new_obs)
      obs = new_obs
    self.obs = obs
    self.step_count += c["nsteps"] * c["nenvs"]
    data = {k: np.asarray(v) for k, v in inter.items()}
    last_values = self.act(obs)[2]
    adv, vt = gae_advantages(data["rewards"], data["resets"], data["values"], last_values,
                             c["gamma"], c["lambda_"])
    data["advantages"], data["value_targets"] = adv, vt
    return {k: merge_time_batch(v) for k, v in data.items()}

  def update(self, data):
    c = self.cfg
    n = data["observations"].shape[0]
    for _ in range(c["num_epochs"]):
      perm = np.random.permutation(n)
      data = {k: v[perm] for k, v in data.items()}
      mbsize = n // c["num_minibatches"]
      for start in range(0, n, mbsize):
        mb = {k: v[start:start + mbsize] for k, v in data.items()}
        mb["advantages"] = normalize_advantages(mb["advantages"])
        logits, values = nature_cnn_forward(self.params, mb["observations"])
        log_prob, entropy, _ = categorical_log_prob_entropy(logits, mb["actions"])
        terms = ppo_loss_terms(log_prob, entropy, values, mb["log_prob"], mb["advantages"],
                               mb["values"], mb["value_targets"], c["cliprange"],
                               c["value_loss_coef"], c["entropy_coef"])
        self.optimizer.zero_grad()
        terms["loss"].backward()
        torch.nn.utils.clip_grad_norm_(list(self.params.values()), c["max_grad_norm"])
        self.lr.data = torch.tensor(float(linear_anneal(c["lr"], c["num_train_steps"],
                                                        self.step_count)))
        self.optimizer.step()
        self.losses.append(float(terms["loss"].detach()))

  def iteration(self):
    self.update(self.rollout())


def available_cores(capped=True):
  """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota
  (a GPU box shows every host core but grants a share of them) and -- ``capped`` -- by
  DERL_AMD_CPU_THREADS (default 16: torch-CPU's conv / GEMM kernels at these shapes stop
  scaling there; the cap is stated in the baseline's ``sample`` string)."""
  import os
  cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
  for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
      with open(path) as f:
        fields = f.read().split()
      if path.endswith("cpu.max"):
        if fields[0] != "max":
          cores = min(cores, max(1, int(int(fields[0]) / int(fields[1]))))
      else:
        quota = int(fields[0])
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
          period = int(f.read())
        if quota > 0:
          cores = min(cores, max(1, quota // period))
    except (OSError, ValueError, IndexError):
      continue
  if not capped:
    return cores
  return min(cores, int(os.environ.get("DERL_AMD_CPU_THREADS", "16")))


def time_cpu_baseline(nenvs=256, nsteps=128, iterations=3, warmup=1, threads=None, budget_s=None,
                      **kwargs):
  """BASELINE.md section 3.1 protocol: same shapes as the GPU run, ``warmup`` untimed iteration(s)
  then ``iterations`` timed ones on ``threads`` host cores (default: every core this process may
  use).  ``budget_s`` bounds the timed part on a slow box: the number of timed iterations is cut
  (never below 1) so that it fits, and the returned ``sample`` string says what was run.
  Returns dict(value env-steps/s, seconds, cores, sample, iterations, warmup)."""
  threads = threads or available_cores()
  granted = available_cores(capped=False)
  cap_note = (f" (capped: the box grants {granted} cores, DERL_AMD_CPU_THREADS lifts the cap)"
              if granted > threads else f" (all {granted} cores granted to this process)")
  ppo = CpuPPO(nenvs=nenvs, nsteps=nsteps, threads=threads, **kwargs)
  warm_seconds = 0.0
  for _ in range(warmup):
    start = time.perf_counter()
    ppo.iteration()
    warm_seconds = time.perf_counter() - start
  if budget_s is not None and warmup and warm_seconds > 0:
    iterations = max(1, min(iterations, int(budget_s / warm_seconds)))
  start = time.perf_counter()
  for _ in range(iterations):
    ppo.iteration()
  seconds = time.perf_counter() - start
  steps = iterations * nenvs * nsteps
  return dict(value=steps / seconds, seconds=seconds, cores=threads, iterations=iterations,
              warmup=warmup,
              sample=f"{warmup} warm-up + {iterations} timed PPO iteration(s) of nenvs={nenvs} x "
                     f"nsteps={nsteps} (3 epochs x 4 minibatches of {nenvs * nsteps // 4}), NatureCNN, "
                     f"synthetic frames, torch-CPU fp32 on {threads} threads{cap_note}")
