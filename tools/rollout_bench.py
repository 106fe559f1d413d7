"""Times the native rollout (dx_cnn_rollout_synth: horizon act + synthetic env steps, one launch per call).
usage: python3 tools/rollout_bench.py [nenvs [horizon]]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

dev = torch.device("cuda:0")
nenvs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 128
eng = CnnEngine(4, max_batch=max(nenvs, 64), device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
buffers = dict(obs=torch.randint(0, 256, (horizon + 1, nenvs, 84, 84, 4), dtype=torch.uint8, device=dev),
               actions=torch.empty(horizon, nenvs, dtype=torch.int64, device=dev),
               log_prob=torch.empty(horizon, nenvs, device=dev), values=torch.empty(horizon, nenvs, device=dev),
               rewards=torch.empty(horizon, nenvs, device=dev),
               resets=torch.empty(horizon, nenvs, dtype=torch.uint8, device=dev))


def run(n):
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  start.record()
  for i in range(n):
    eng.rollout_synth(buffers, horizon, nenvs, 7, i * horizon, 11, i * horizon, 0.05, 0.01)
  end.record()
  torch.cuda.synchronize()
  return start.elapsed_time(end) / n


run(3)
ms = min(run(10) for _ in range(3))
print(json.dumps(dict(nenvs=nenvs, horizon=horizon, rollout_ms=round(ms, 3), step_us=round(ms * 1e3 / horizon, 2),
                      env_steps_per_s=round(nenvs * horizon / ms * 1e3))), flush=True)
