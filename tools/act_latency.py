"""Latency of the rollout step (dx_cnn_act + synthetic env step) per batch size."""
import json, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd.cnn_engine import CnnEngine
dev = torch.device("cuda:0")
for B in (32, 64, 128, 256):
  eng = CnnEngine(4, max_batch=B, device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  obs = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device=dev)
  a = torch.empty(B, dtype=torch.int64, device=dev); l = torch.empty(B, device=dev); v = torch.empty(B, device=dev)
  for _ in range(5):
    eng.act(obs, a, l, v)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50):
    eng.act(obs, a, l, v)
  e1.record(); e1.synchronize()
  print(json.dumps(dict(B=B, act_us=round(e0.elapsed_time(e1) * 20, 1))), flush=True)
