"""Samples rocm-smi (power, sclk, temperature) while one network stage runs in a loop: is the
stage's duration set by the matrix pipe or by the package power limit?
usage: python3 tools/power_probe.py [batch] [seconds per stage] [stage ...]"""
import ctypes
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
from derl_amd import _lib  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
stages = [int(s) for s in sys.argv[3:]] or list(range(len(bench.STAGES) - 1))
dev = torch.device("cuda:0")
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
eng._ensure_backward()
eng.forward(obs)
eng.dhead[:batch * 32].normal_()
eng.backward(obs)
stream = _lib.stream_ptr(dev)
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
      elif low.startswith("sclk"):
        m = re.search(r"(\d+)\s*mhz", str(val).lower())
        row["sclk_mhz"] = int(m.group(1)) if m else val
      elif "junction" in low or "hotspot" in low:
        row["temp_c"] = float(val)
    samples.append(row)
    time.sleep(0.3)


def mean(key, rows):
  vals = [r[key] for r in rows if isinstance(r.get(key), (int, float))]
  return round(sum(vals) / len(vals), 1) if vals else None


for stage in stages:
  del samples[:]
  stop.clear()
  thread = threading.Thread(target=sample)
  thread.start()
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0, n = time.time(), 0
  start.record()
  while time.time() - t0 < seconds:
    for _ in range(50):
      _lib.call("dx_cnn_stage", ctypes.byref(eng.ctx), stage, _lib.ptr(obs), 1, None, batch, stream)
    n += 50
    torch.cuda.synchronize()
  end.record()
  torch.cuda.synchronize()
  stop.set()
  thread.join()
  us = start.elapsed_time(end) * 1e3 / n
  name = bench.STAGES[stage]
  fl = bench.stage_flops(name, batch, 4)
  tail = samples[len(samples) // 2:]  # steady state: second half
  print(json.dumps(dict(stage=name, us=round(us, 1), TFLOPs=round(fl / us / 1e6, 1) if fl else None,
                        power_w=mean("power_w", tail), sclk_mhz=mean("sclk_mhz", tail),
                        temp_c=mean("temp_c", tail), nsamples=len(tail), first=samples[:1])), flush=True)
