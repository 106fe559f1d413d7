"""Sustained fp32 MFMA rate of the GPU with no memory traffic (dx_diag_mfma_f32).
usage: python tools/mfma_peak.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd import _lib  # noqa: E402
from tools import _diag  # noqa: E402

dev = torch.device("cuda:0")
out = torch.zeros(4, device=dev)
stream = _lib.stream_ptr(dev)
for entry in ("dx_diag_mfma_f32", "dx_diag_mfma_f32_chain"):
  for blocks_per_cu in (1, 2, 4):
    blocks, iters = 256 * blocks_per_cu, 20000
    for _ in range(2):
      _diag.call(entry, blocks, iters, _lib.ptr(out), stream)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
      _diag.call(entry, blocks, iters, _lib.ptr(out), stream)
    end.record()
    end.synchronize()
    ms = start.elapsed_time(end) / 5
    flops = blocks * 4 * iters * 4 * 4096.0
    print(json.dumps(dict(kernel=entry, waves_per_simd=blocks_per_cu, ms=round(ms, 3),
                          TFLOPs=round(flops / ms / 1e9, 1))), flush=True)

for mode, rnd in ((0, 1), (1, 1), (1, -1), (2, 1), (3, 1)):  # modes 2 / 3: eight accumulator tiles, scattered / chained  # rnd = -1: pseudo-random operands (power / clock depend on the data)
  for blocks_per_cu in (1, 2, 4):
    blocks, iters = 256 * blocks_per_cu, 4000
    for _ in range(2 if rnd > 0 else 40):
      _diag.call("dx_diag_lds_mfma_f32", blocks, rnd * iters, mode, _lib.ptr(out), stream)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
      _diag.call("dx_diag_lds_mfma_f32", blocks, rnd * iters, mode, _lib.ptr(out), stream)
    end.record()
    end.synchronize()
    ms = start.elapsed_time(end) / 5
    flops = blocks * 4 * iters * 32 * 4096.0
    print(json.dumps(dict(kernel="dx_diag_lds_mfma_f32", mode=mode, random_data=rnd < 0, waves_per_simd=blocks_per_cu,
                          ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1))), flush=True)
