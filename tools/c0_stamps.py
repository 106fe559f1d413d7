"""Reads the phase stamps a DIAGNOSTIC build of conv0_fwd_b16_kernel leaves at the start of y0
(cycles per tile and wave: first barrier, patch store + second barrier, MFMA loop incl. index math,
epilogue issue).  Only meaningful with that build; see DESIGN.md section 3."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from derl_amd import _lib  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

batch = 8192
dev = torch.device("cuda:0")
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
eng._ensure_backward()
for _ in range(5):
  eng.forward(obs)
  eng.backward(obs)
stream = _lib.stream_ptr(dev)
_lib.call("dx_cnn_stage", ctypes.byref(eng.ctx), 0, _lib.ptr(obs), 1, None, batch, stream)
torch.cuda.synchronize()
raw = eng.y0.view(torch.int64)[:512 * 4 * 8].cpu().numpy().reshape(512 * 4, 8)
raw = raw[raw[:, 4] > 0]
tiles = raw[:, 4]
names = ["wait at the first barrier", "patch store + second barrier", "index math + MFMA loop", "epilogue (stores issued)"]
total = raw[:, 5]
print("waves %d, tiles per wave median %d, kernel cycles per wave median %d" % (len(raw), np.median(tiles), np.median(total)))
for i, n in enumerate(names):
  print("%-32s %7.0f cycles per tile (%4.1f %% of the kernel)" % (n, np.median(raw[:, i] / tiles), 100 * np.median(raw[:, i] / total)))
