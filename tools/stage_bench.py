"""Runs every network stage a few times at a given batch (for rocprofv3 --pmc passes and quick
A/B timing).  usage: python3 tools/stage_bench.py [batch] [iters] [stage ...]"""
import ctypes
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from derl_amd import _lib  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = [int(s) for s in sys.argv[3:]]
dev = torch.device("cuda:0")
torch.manual_seed(0)
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
pool = max(batch, 1024)
obs = torch.randint(0, 256, (pool, 84, 84, 4), dtype=torch.uint8, device=dev)
idx = torch.randperm(pool, device=dev)[:batch].to(torch.int32)


class M:
  engine = eng


times = bench.time_stages(M, obs, idx, batch, iters=iters)
for stage, name in enumerate(bench.time_stages.names):  # the stages an update really launches (factored tail or layer by layer)
  if only and stage not in only:
    continue
  fl = bench.stage_flops(name, batch, 4)
  print(json.dumps(dict(stage=stage, name=name, us=round(times[name], 1),
                        TFLOPs=round(fl / times[name] / 1e6, 2) if fl else None)), flush=True)
