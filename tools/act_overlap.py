"""Two half-batch act chains on two streams vs one full-batch chain (is there idle capacity
between the rollout's dependent launches?)."""
import json, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd.cnn_engine import CnnEngine
dev = torch.device("cuda:0")


def make(B):
  eng = CnnEngine(4, max_batch=B, device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty(); eng.pack()
  obs = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device=dev)
  out = (torch.empty(B, dtype=torch.int64, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
  return eng, obs, out


for B in (32, 64, 128, 256):
  full = make(B)
  halves = [make(B // 2), make(B // 2)]
  streams = [torch.cuda.Stream(), torch.cuda.Stream()]
  def run_full(n):
    for _ in range(n):
      full[0].act(full[1], *full[2])
  def run_halves(n):
    for _ in range(n):
      for (eng, obs, out), st in zip(halves, streams):
        with torch.cuda.stream(st):
          eng.act(obs, *out)
  res = {}
  for name, fn in (("full", run_full), ("two_halves", run_halves)):
    fn(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(100)
    for st in streams:
      torch.cuda.current_stream().wait_stream(st)
    e1.record(); e1.synchronize()
    res[name] = round(e0.elapsed_time(e1) * 10, 1)
  print(json.dumps(dict(B=B, us_per_step=res)), flush=True)
