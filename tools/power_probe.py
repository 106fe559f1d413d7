"""Package power, sclk and temperature (rocm-smi, sampled from a thread) while ONE stage of an update -- or the rollout's
one-launch horizon -- runs back to back for a few seconds: is a stage's duration set by a pipe or by the package power
limit?  Every stage an update really launches (bench.stage_launcher: the same launch closures bench.py times) plus
`rollout` (dx_cnn_rollout_synth, 128 steps x 256 envs).
usage: python3 tools/power_probe.py [batch] [seconds per stage] [out.json]     (one JSON document; rows also on stdout)"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
out_path = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
torch.manual_seed(0)
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
idx = torch.randperm(batch, device=dev).to(torch.int32)


class M:
  engine = eng


names, routes, launch = bench.stage_launcher(M, obs, idx, batch)
for _ in range(2):  # defines every buffer a later stage reads
  for name in names:
    launch(name)
nenvs, horizon = 256, 128
buffers = dict(obs=torch.randint(0, 256, (horizon + 1, nenvs, 84, 84, 4), dtype=torch.uint8, device=dev),
               actions=torch.empty(horizon, nenvs, dtype=torch.int64, device=dev),
               log_prob=torch.empty(horizon, nenvs, device=dev), values=torch.empty(horizon, nenvs, device=dev),
               rewards=torch.empty(horizon, nenvs, device=dev),
               resets=torch.empty(horizon, nenvs, dtype=torch.uint8, device=dev))
samples, stop = [], threading.Event()


def sample():
  while not stop.is_set():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"],
                         capture_output=True, text=True).stdout
    try:
      card = next(iter(json.loads(out).values()))
    except Exception:  # pylint: disable=broad-except
      samples.append(dict(raw=out[:200]))
      time.sleep(0.3)
      continue
    row = {}
    for key, val in card.items():
      low = key.lower()
      if "power" in low and "(w)" in low:
        row["power_w"] = float(val)
      elif low.startswith("sclk"):  # "sclk clock speed:": "(1915Mhz)"; "sclk clock level:": "1" (no MHz: must not overwrite)
        m = re.search(r"(\d+)\s*mhz", str(val).lower())
        if m:
          row["sclk_mhz"] = int(m.group(1))
      elif "junction" in low or "hotspot" in low:
        row["temp_c"] = float(val)
    samples.append(row)
    time.sleep(0.05)


def mean(key, rows):
  vals = [r[key] for r in rows if isinstance(r.get(key), (int, float))]
  return round(sum(vals) / len(vals), 1) if vals else None


def probe(name, fn, per_call):
  del samples[:]
  stop.clear()
  thread = threading.Thread(target=sample)
  thread.start()
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0, n = time.time(), 0
  start.record()
  while time.time() - t0 < seconds:
    for _ in range(per_call):
      fn()
    n += per_call
    torch.cuda.synchronize()
  end.record()
  torch.cuda.synchronize()
  stop.set()
  thread.join()
  us = start.elapsed_time(end) * 1e3 / n
  tail = samples[len(samples) // 2:]  # steady state: the second half
  return dict(stage=name, us=round(us, 1), launches=n, power_w=mean("power_w", tail), power_w_max=max(
      [r["power_w"] for r in tail if isinstance(r.get("power_w"), float)], default=None), sclk_mhz=mean("sclk_mhz", tail),
              temp_c=mean("temp_c", tail), samples=len(tail))


rows = []
idle = probe("idle (no launches)", lambda: time.sleep(0.02), 1)
idle.pop("us")
rows.append(idle)
print(json.dumps(idle), flush=True)
route = routes()
for name in names:
  row = probe(name, lambda n=name: launch(n), 50)
  row["route"] = route.get(name, "")
  ex, peak, how = bench.executed_flops(name, route.get(name, ""), batch, 4)
  if ex:
    row.update(executed_TFLOPs=round(ex / row["us"] / 1e6, 1), peak_TFLOPs=peak, mfma=how,
               pJ_per_executed_flop_incl_everything=round(row["power_w"] * row["us"] * 1e-6 / ex * 1e12, 3) if row["power_w"] else None)
  rows.append(row)
  print(json.dumps(row), flush=True)
count = [0]


def rollout():
  eng.rollout_synth(buffers, horizon, nenvs, 7, count[0] * horizon, 11, count[0] * horizon, 0.05, 0.01)
  count[0] += 1


row = probe("rollout (dx_cnn_rollout_synth, 128 steps x 256 envs)", rollout, 4)
rows.append(row)
print(json.dumps(row), flush=True)
doc = dict(what="rocm-smi package power / sclk / junction temperature, mean over the second half of the samples taken while "
                "one stage runs back to back for the given time (tools/power_probe.py); `us` = HIP-event time per launch "
                "over the whole probe",
           minibatch=batch, seconds_per_stage=seconds, device=torch.cuda.get_device_name(0), rows=rows)
if out_path:
  with open(out_path, "w") as f:
    json.dump(doc, f, indent=1)
