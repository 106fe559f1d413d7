"""Throughput of the other BASELINE.json configs on one GPU (parity-test cases, not bench lines):
  c3: PPO HalfCheetah-v3-shaped, nenvs=2048, nsteps=64, MLP Gaussian policy (10 epochs x 32 mb)
  c5: A2C Breakout-shaped, per-GPU shard of nenvs=4096/8 = 512, nsteps=5 (1 update per rollout)
usage: python tools/bench_configs.py [c3|c5|c1] [iters]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import derl_amd as derl  # noqa: E402


def run(name, iters):
  torch.manual_seed(0)
  np.random.seed(0)
  derl.summary.stop_recording()
  if name == "c3":
    env = derl.env.make("HalfCheetah-v3", nenvs=2048, seed=0)
    kw = derl.PPOFactory.get_kwargs("mujoco")
    kw.update(nenvs=2048, num_runner_steps=64, num_train_steps=1e12)
    alg = derl.PPOFactory(**kw).make(env)
    updates, steps_per_iter = kw["num_epochs"] * kw["num_minibatches"], 2048 * 64
  elif name == "c5":
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=512, seed=0)
    kw = derl.A2CFactory.get_kwargs()
    kw.update(nenvs=512, num_train_steps=1e12)
    alg = derl.A2CFactory(**kw).make(env)
    updates, steps_per_iter = 1, 512 * 5
  else:
    env = derl.env.make("CartPole-v1", nenvs=8, seed=0)
    kw = derl.PPOFactory.get_kwargs("atari")
    kw.update(nenvs=8, num_train_steps=1e12)
    alg = derl.PPOFactory(**kw).make(env)
    updates, steps_per_iter = 12, 8 * 128
  it = alg.runner.run()

  def iteration():
    for _ in range(updates):
      alg.step(next(it))
      derl.summary.stop_recording()

  for _ in range(2):
    iteration()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(iters):
    iteration()
  enq = time.perf_counter() - t0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print(json.dumps(dict(config=name, env_steps_per_s=round(iters * steps_per_iter / dt, 1),
                        ms_per_iteration=round(dt / iters * 1e3, 2),
                        host_enqueue_ms=round(enq / iters * 1e3, 2), updates_per_iteration=updates,
                        loss=float(alg.loss_fn.last_terms[0].item()))), flush=True)


if __name__ == "__main__":
  names = [sys.argv[1]] if len(sys.argv) > 1 else ["c3", "c5", "c1"]
  iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  for n in names:
    run(n, iters if n != "c5" else iters * 20)
