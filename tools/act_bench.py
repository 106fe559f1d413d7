"""Times the rollout's act step (dx_cnn_act: trunk + heads + sampling) and its forward stages at
rollout batch sizes.  usage: python3 tools/act_bench.py [batch ...]"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

dev = torch.device("cuda:0")
for batch in [int(b) for b in sys.argv[1:]] or [32, 128, 256]:
  eng = CnnEngine(4, max_batch=max(batch, 64), device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  actions = torch.empty(batch, dtype=torch.int64, device=dev)
  log_prob = torch.empty(batch, device=dev)
  values = torch.empty(batch, device=dev)
  for _ in range(20):
    eng.act(obs, actions, log_prob, values)
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  start.record()
  for _ in range(200):
    eng.act(obs, actions, log_prob, values)
  end.record()
  torch.cuda.synchronize()

  class M:
    engine = eng

  stages = bench.time_stages(M, obs, None, batch, iters=20)
  print(json.dumps(dict(batch=batch, act_us=round(start.elapsed_time(end) * 1e3 / 200, 1),
                        stages={k: round(v, 1) for k, v in stages.items() if k.endswith("_fwd")})), flush=True)
