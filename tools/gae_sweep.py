"""GAE scan latency at the BASELINE shapes and a bandwidth sweep over N at T=128.

Algorithmic bytes = 17 B per (t, n) element + 4 B per env."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd import ops  # noqa: E402


def time_gae(T, N, iters=50, warm=5):
  dev = torch.device("cuda:0")
  r = torch.randn(T, N, device=dev)
  z = torch.rand(T, N, device=dev) < 0.01
  v = torch.randn(T, N, device=dev)
  lv = torch.randn(N, device=dev)
  adv, vt = torch.empty_like(v), torch.empty_like(v)
  for _ in range(warm):
    ops.gae(r, z, v, lv, 0.99, 0.95, adv, vt)
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  start.record()
  for _ in range(iters):
    ops.gae(r, z, v, lv, 0.99, 0.95, adv, vt)
  end.record()
  torch.cuda.synchronize()
  us = start.elapsed_time(end) * 1e3 / iters
  nbytes = 17 * T * N + 4 * N
  return dict(T=T, N=N, us=round(us, 2), GBps=round(nbytes / us / 1e3, 1),
              frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 3))


if __name__ == "__main__":
  shapes = [(128, 256), (64, 2048), (5, 4096), (5, 512)]
  shapes += [(128, 1 << k) for k in range(8, 22, 2)] + [(128, 1 << 21)]
  for T, N in shapes:
    print(json.dumps(time_gae(T, N)), flush=True)
