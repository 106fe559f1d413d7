"""PPO on the built-in CartPole-v1 (BASELINE config 1 hyper-parameters) really learns: mean episode
length per rollout while training.  usage: python tools/cartpole_learns.py [iterations]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import derl_amd as derl  # noqa: E402


def run(iterations=150, seed=0):
  derl.summary.stop_recording()
  torch.manual_seed(seed)
  np.random.seed(seed)
  env = derl.env.make("CartPole-v1", nenvs=8, seed=seed)
  kwargs = derl.PPOFactory.get_kwargs("atari")
  kwargs.update(nenvs=8, num_runner_steps=128, num_train_steps=8 * 128 * iterations)
  alg = derl.PPOFactory(**kwargs).make(env)
  lengths, updates = [], kwargs["num_epochs"] * kwargs["num_minibatches"]
  data = alg.runner.run()
  start = time.perf_counter()
  finished = 0
  for _ in range(iterations):
    for _ in range(updates):
      alg.step(next(data))
      derl.summary.stop_recording()
    # the first next() of an iteration collected the rollout: episodes that ended in it
    now = env.unwrapped.episodes_done
    lengths.append(8 * 128 / max(now - finished, 1))
    finished = now
  return lengths, time.perf_counter() - start


if __name__ == "__main__":
  its = int(sys.argv[1]) if len(sys.argv) > 1 else 150
  lengths, seconds = run(its)
  print(json.dumps(dict(iterations=its, seconds=round(seconds, 2),
                        mean_episode_length_first5=round(float(np.mean(lengths[:5])), 1),
                        mean_episode_length_last10=round(float(np.mean(lengths[-10:])), 1),
                        curve=[round(float(np.mean(lengths[i:i + 10])), 1) for i in range(0, its, 10)])))
