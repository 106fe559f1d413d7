"""Runs the GAE scan a few times at one shape (for rocprofv3 --pmc passes)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda:0")
r = torch.randn(T, N, device=dev)
z = torch.rand(T, N, device=dev) < 0.01
v = torch.randn(T, N, device=dev)
lv = torch.randn(N, device=dev)
adv, vt = torch.empty_like(v), torch.empty_like(v)
for _ in range(5):
  ops.gae(r, z, v, lv, 0.99, 0.95, adv, vt)
torch.cuda.synchronize()
print("algorithmic bytes per launch", 17 * T * N + 4 * N)
