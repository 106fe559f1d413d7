"""Where the NT GEMM K loop loses matrix-pipe time: dx_diag_gemm_loop_f32 at conv1's shape
(5184 tiles of 128 x 64, K = 512) and at a long K.  usage: python tools/gemm_loop.py"""
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
for tiles, ktiles in ((5184, 16), (1024, 98), (1024, 16)):
  A = torch.randn(tiles * 128, 32 * ktiles, device=dev)
  B = torch.randn(64, 32 * ktiles, device=dev)
  for what in (2, 3, 7, 8):
    for _ in range(3):
      _diag.call("dx_diag_gemm_loop_f32", _lib.ptr(A), _lib.ptr(B), tiles, ktiles, what, _lib.ptr(out), stream)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(10):
      _diag.call("dx_diag_gemm_loop_f32", _lib.ptr(A), _lib.ptr(B), tiles, ktiles, what, _lib.ptr(out), stream)
    end.record()
    end.synchronize()
    us = start.elapsed_time(end) * 100
    flops = 2.0 * tiles * 128 * 64 * 32 * ktiles
    print(json.dumps(dict(tiles=tiles, K=32 * ktiles, what=["lds reads + mfma", "+ lds writes, barriers", "+ global loads", "global loads two tiles ahead", "+ global loads, A from 8 cache-resident tiles", "double-buffered LDS BK 32, one barrier", "double-buffered LDS BK 16, one barrier", "LDS-DMA, 3 stages, 128-row tile", "LDS-DMA, 3 stages, 256-row tile"][what],
                          us=round(us, 1), TFLOPs=round(flops / us / 1e6, 1))), flush=True)
  del A, B
