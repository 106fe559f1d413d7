"""Times the forward trunk of a training minibatch (dx_cnn_forward_trunk) -- usage: python3 tools/trunk_bench.py [minibatch ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd import _lib  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

dev = torch.device("cuda:0")
for batch in [int(b) for b in sys.argv[1:]] or [8192]:
  eng = CnnEngine(4, max_batch=batch, device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  idx = torch.randperm(batch, device=dev, dtype=torch.int32)
  for _ in range(3):
    eng.forward_trunk(obs, idx)
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    start.record()
    for _ in range(20):
      eng.forward_trunk(obs, idx)
    end.record()
    torch.cuda.synchronize()
    best = min(best, start.elapsed_time(end) / 20)
  routes = [_lib.load().dx_cnn_last_route(i).decode() for i in range(3)]
  print(json.dumps(dict(minibatch=batch, trunk_us=round(best * 1e3, 1), routes=routes)), flush=True)
