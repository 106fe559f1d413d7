"""Is the host ahead of the GPU when an iteration's update phase begins?  After the first next()
of an iteration (rollout + GAE + first minibatch enqueued) an event is recorded and queried at
once: True = the GPU had already drained its queue (the host is the limiter there).
usage: python3 tools/host_ahead_probe.py [nenvs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import derl_amd as derl  # noqa: E402

nenvs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=nenvs, seed=0)
kwargs = derl.PPOFactory.get_kwargs("atari")
kwargs.update(nenvs=nenvs, num_train_steps=10 ** 9)
alg = derl.PPOFactory(**kwargs).make(env)
it = alg.runner.run()
per_iter = kwargs["num_epochs"] * kwargs["num_minibatches"]
drained, lag = 0, []
for iteration in range(14):
  for k in range(per_iter):
    data = next(it)
    derl.summary.stop_recording()
    if k == 0 and iteration >= 4:
      ev = torch.cuda.Event()
      ev.record()
      done = ev.query()
      drained += int(done)
      t0 = time.perf_counter()
      ev.synchronize()
      lag.append((time.perf_counter() - t0) * 1e3)
    alg.step(data)
torch.cuda.synchronize()
print(f"nenvs {nenvs}: GPU already idle at the start of the update phase in {drained} of 10 iterations; "
      f"otherwise it was behind the host by {min(lag):.2f}-{max(lag):.2f} ms")
