"""Does bracketing every stage of an update with HIP events change what the stages take?  bench.time_stages times one update's
stages (a) with an event pair per stage and (b) with one event pair around all passes; it reports (a) scaled to (b).
usage: python3 tools/stage_events_probe.py [minibatch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
obs = torch.randint(0, 256, (4 * batch, 84, 84, 4), dtype=torch.uint8, device=dev)
idx = torch.randperm(4 * batch, device=dev)[:batch].to(torch.int32)


class M:
  engine = eng


scaled = bench.time_stages(M, obs, idx, batch, iters=10)
raw = bench.time_stages.bracketed_us
print("bracketed (an event pair per stage): sum %.1f us" % sum(raw.values()), {k: round(v, 1) for k, v in raw.items()})
print("one event pair around all passes: %.1f us per pass -> scale %.4f" % (bench.time_stages.pass_us, bench.time_stages.bracket_scale))
print("stage table:", {k: round(v, 1) for k, v in scaled.items()})
