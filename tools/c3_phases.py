import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import derl_amd as derl
from tools.bench_configs import build
alg, updates, steps = build("c3")
it = alg.runner.run()
def iteration(stamp=None):
  for u in range(updates):
    t0 = time.perf_counter()
    d = next(it)
    t1 = time.perf_counter()
    derl.summary.stop_recording()
    alg.step(d)
    t2 = time.perf_counter()
    if stamp is not None: stamp.append((u, t1 - t0, t2 - t1))
for _ in range(3): iteration()
torch.cuda.synchronize()
st = []
t0 = time.perf_counter(); iteration(st); host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
nx = sum(a for _, a, _ in st); sp = sum(b for _, _, b in st)
first = [ (u,a,b) for u,a,b in st if u % 32 == 0]
print(f"host {host*1e3:.2f} ms wall {wall*1e3:.2f} ms; next() total {nx*1e3:.2f} ms, step() total {sp*1e3:.2f} ms")
print("first minibatch of each epoch (next ms, step ms):", [(round(a*1e3,3), round(b*1e3,3)) for _,a,b in first])
rest = [(a,b) for u,a,b in st if u % 32]
print(f"other minibatches: next {np.mean([a for a,_ in rest])*1e6:.1f} us, step {np.mean([b for _,b in rest])*1e6:.1f} us")
