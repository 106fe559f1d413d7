"""Where the host waits during config 3's epochs: wall time of every C-ABI call and of torch.empty."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import derl_amd as derl
from derl_amd import _lib
from tools.bench_configs import build
alg, updates, steps = build("c3")
it = alg.runner.run()
def iteration():
  for u in range(updates):
    d = next(it); derl.summary.stop_recording(); alg.step(d)
for _ in range(3): iteration()
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig_call = _lib.call
def timed_call(name, *args):
  t0 = time.perf_counter(); r = orig_call(name, *args); dt = time.perf_counter() - t0
  a = acc[name]; a[0] += 1; a[1] += dt; a[2] = max(a[2], dt); return r
_lib.call = timed_call
import derl_amd.ops, derl_amd.mlp_engine
orig_empty = torch.empty
def timed_empty(*a, **k):
  t0 = time.perf_counter(); r = orig_empty(*a, **k); dt = time.perf_counter() - t0
  x = acc["torch.empty"]; x[0] += 1; x[1] += dt; x[2] = max(x[2], dt); return r
torch.empty = timed_empty
t0 = time.perf_counter(); iteration(); host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host {host*1e3:.2f} ms")
for k, (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
  print(f"{k:28s} calls {n:5d} total {tot*1e3:8.3f} ms max {mx*1e3:8.3f} ms")
