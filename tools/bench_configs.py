"""Throughput of the other BASELINE.json configs on one GPU (parity-test cases, not the bench line's
headline; bench.py reports them under ``other_configs``):
  c3: PPO HalfCheetah-v3-shaped, nenvs=2048, nsteps=64, MLP Gaussian policy (10 epochs x 32 mb)
  c5: A2C Breakout-shaped, per-GPU shard of nenvs=4096/8 = 512, nsteps=5 (1 update per rollout)
  c1: PPO CartPole-v1 nenvs=8 (plumbing)
usage: python tools/bench_configs.py [c3|c5|c1] [iters]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import derl_amd as derl  # noqa: E402
from derl_amd import _lib  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
MACS_C0, MACS_C1, MACS_C2 = 3_276_800, 2_654_208, 1_806_336
PEAK_F32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector fp32 FMA peak equals the fp32 matrix peak
PEAK_HBM_GBPS = 8000.0
CNN_FWD_MFLOP = 18.69  # BASELINE.md section 4 (A = 4)


def build(name):
  torch.manual_seed(0)
  np.random.seed(0)
  derl.summary.stop_recording()
  if name == "c3":
    env = derl.env.make("HalfCheetah-v3", nenvs=2048, seed=0)
    kw = derl.PPOFactory.get_kwargs("mujoco")
    kw.update(nenvs=2048, num_runner_steps=64, num_train_steps=1e12)
    alg = derl.PPOFactory(**kw).make(env)
    return alg, kw["num_epochs"] * kw["num_minibatches"], 2048 * 64
  if name == "c5":
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=512, seed=0)
    kw = derl.A2CFactory.get_kwargs()
    kw.update(nenvs=512, num_train_steps=1e12)
    alg = derl.A2CFactory(**kw).make(env)
    return alg, 1, 512 * 5
  env = derl.env.make("CartPole-v1", nenvs=8, seed=0)
  kw = derl.PPOFactory.get_kwargs("atari")
  kw.update(nenvs=8, num_train_steps=1e12)
  alg = derl.PPOFactory(**kw).make(env)
  return alg, 12, 8 * 128


def bounds(name, steps_per_iter, updates, seconds_per_iter):
  """The roofline that applies and the fraction reached (algorithmic work per env step)."""
  if name == "c5":  # NatureCNN; A2C = rollout forward + bootstrap / T + one forward + backward per env step
    mflop = CNN_FWD_MFLOP * (1 + 1 / 5 + 3)
    achieved = mflop * 1e6 * steps_per_iter / seconds_per_iter / 1e12
    # the ceiling: what the kernels EXECUTE, each part at the nameplate peak of the unit it runs on -- the conv layers on
    # bf16 MFMA (conv0: 3 exact bf16 products per fp32 product, forward + weight gradient; conv1 / conv2: 6, forward +
    # data gradient + weight gradient; rollout forwards 1 + 1 / 5 per env step), the factored linear layer + heads as
    # HBM passes over y2 (3136 floats: once per rollout forward, three times per update sample)
    roll = 1 + 1 / 5
    bf16_mflop = 2e-6 * (3.0 * MACS_C0 * (roll + 2.0) + 6.0 * (MACS_C1 + MACS_C2) * (roll + 3.0))
    hbm_bytes = 3136 * 4 * (roll + 3.0)
    floor_s = bf16_mflop / (PEAK_BF16_MFMA_TFLOPS * 1e6) + hbm_bytes / (PEAK_HBM_GBPS * 1e9)
    composite = 1.0 / floor_s  # env steps per second
    rate = steps_per_iter / seconds_per_iter
    return dict(bound="mfma (bf16, executed flops) + hbm (factored tail)", unit="env-steps/s",
                achieved=round(rate, 1), peak=round(composite, 1), frac=round(rate / composite, 4),
                executed_bf16_mflop_per_env_step=round(bf16_mflop, 1),
                hbm_bytes_per_env_step_of_the_factored_tail=round(hbm_bytes, 1),
                algorithmic_mflop_per_env_step=round(mflop, 1), algorithmic_TFLOPs=round(achieved, 2),
                vs_fp32_reference_line=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                note="`frac` = measured env-steps/s over the composite ceiling (every executed part at its unit's nameplate "
                     "peak: bf16 MFMA 2.5 PFLOP/s, HBM 8 TB/s); `vs_fp32_reference_line` prices the ALGORITHMIC fp32 flops "
                     "at the fp32-MFMA peak and may exceed 1 -- it is not a roofline fraction")
  if name == "c3":  # 11,085-parameter MLP: neither roofline is near -- the bound is the dependent-launch chain
    kflop = 21.6 * (1 + 1 / 64 + 10 * 3)  # per env step
    flops = kflop * 1e3 * steps_per_iter / seconds_per_iter / 1e12
    # HBM bytes per update: 4,096 rows x (17 obs + 6 actions + 5 scalars) floats in, 28 B / parameter of Adam
    nbytes = updates * (4096 * 28 * 4 + 28 * 11085) + 17 * 4 * 2 * steps_per_iter
    gbps = nbytes / seconds_per_iter / 1e9
    return dict(bound="latency (dependent launches / grid barriers; both rooflines are < 3 %)",
                valu_TFLOPs=round(flops, 3), valu_frac=round(flops / PEAK_F32_VALU_TFLOPS, 5),
                hbm_GBps=round(gbps, 1), hbm_frac=round(gbps / PEAK_HBM_GBPS, 5),
                algorithmic_kflop_per_env_step=round(kflop, 1))
  return dict(bound="host (8 envs stepped on the CPU)")


def measure(name, iters, warmup=2, budget_s=None):
  """dict(config, env_steps_per_s, ms_per_iteration, launches_per_update, ...) of `iters` iterations
  (cut to fit `budget_s` seconds of timed work, never below 1)."""
  alg, updates, steps_per_iter = build(name)
  it = alg.runner.run()

  def iteration():
    for _ in range(updates):
      alg.step(next(it))
      derl.summary.stop_recording()

  for _ in range(warmup):
    iteration()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  iteration()
  torch.cuda.synchronize()
  one = time.perf_counter() - t0
  if budget_s is not None:
    iters = max(1, min(iters, int(budget_s / max(one, 1e-6))))
  launches0 = _lib.load().dx_launch_count()
  t0 = time.perf_counter()
  for _ in range(iters):
    iteration()
  enq = time.perf_counter() - t0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  launches = _lib.load().dx_launch_count() - launches0
  # what one iteration costs the host when the queue is empty (nothing makes the enqueue calls wait)
  t1 = time.perf_counter()
  iteration()
  unblocked = time.perf_counter() - t1
  torch.cuda.synchronize()
  out = dict(config=name, env_steps_per_s=round(iters * steps_per_iter / dt, 1),
             ms_per_iteration=round(dt / iters * 1e3, 3), iterations=iters,
             host_enqueue_ms=round(enq / iters * 1e3, 3), host_enqueue_ms_unblocked=round(unblocked * 1e3, 3),
             updates_per_iteration=updates,
             library_launches_per_iteration=round(launches / iters, 1),
             loss=float(alg.loss_fn.last_terms[0].item()))
  out["roofline"] = bounds(name, steps_per_iter, updates, dt / iters)
  return out


if __name__ == "__main__":
  names = [sys.argv[1]] if len(sys.argv) > 1 else ["c3", "c5", "c1"]
  iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  for n in names:
    print(json.dumps(measure(n, iters if n != "c5" else iters * 20)), flush=True)
