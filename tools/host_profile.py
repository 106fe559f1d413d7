"""cProfile of the host side of PPO iterations (what the interpreter costs between launches).

usage: python tools/host_profile.py [nenvs] [iterations]
"""
import cProfile, io, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import derl_amd as derl

nenvs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
device = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=nenvs, seed=0, device=device)
kwargs = derl.PPOFactory.get_kwargs("atari")
kwargs.update(nenvs=nenvs, num_runner_steps=128, num_train_steps=1e12)
alg = derl.PPOFactory(**kwargs).make(env, nlogs=1e5)
derl.summary.stop_recording()
data_iter = alg.runner.run()
updates = kwargs["num_epochs"] * kwargs["num_minibatches"]


def iteration():
  for _ in range(updates):
    alg.step(next(data_iter))
    derl.summary.stop_recording()


for _ in range(2):
  iteration()
torch.cuda.synchronize()
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
for _ in range(iters):
  iteration()
prof.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"host enqueue {host / iters * 1e3:.2f} ms/iteration, wall {total / iters * 1e3:.2f} ms/iteration")
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(30)
print(out.getvalue())
