"""cProfile of config 3 (PPO, MLP Gaussian policy, nenvs 2048 x 64, 320 updates per rollout)."""
import cProfile, io, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import derl_amd as derl

torch.manual_seed(0); np.random.seed(0)
env = derl.env.make("HalfCheetah-v3", nenvs=2048, seed=0)
kw = derl.PPOFactory.get_kwargs("mujoco")
kw.update(nenvs=2048, num_runner_steps=64, num_train_steps=1e12)
alg = derl.PPOFactory(**kw).make(env)
derl.summary.stop_recording()
it = alg.runner.run()
updates = kw["num_epochs"] * kw["num_minibatches"]


def iteration():
  for _ in range(updates):
    alg.step(next(it))
    derl.summary.stop_recording()


iteration(); iteration()
torch.cuda.synchronize()
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
for _ in range(3):
  iteration()
prof.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host {host / 3 * 1e3:.1f} ms/iteration, wall {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/iteration")
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(40)
print(out.getvalue())
