"""In-kernel timing of the fc wgrad stage loop (DX_FC_DIAG bit 3): cycles per 16-row stage per
wave, the clock the chip holds inside the loop, and how the workgroups' start times spread.
usage: DERL_AMD_LIBRARY=diag DX_FC_DIAG=8 python tools/fc_wgrad_stamps.py   (add 1 / 2 to drop the copies / barrier)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from derl_amd import _lib  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

assert int(os.environ.get("DX_FC_DIAG", "0")) & 8, "set DX_FC_DIAG=8 (+1, +2)"
assert os.environ.get("DERL_AMD_LIBRARY") == "diag", "the switches exist in the diag flavour only: DERL_AMD_LIBRARY=diag"
batch = 8192
dev = torch.device("cuda:0")
eng = CnnEngine(4, max_batch=batch, device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
eng._ensure_backward()
eng.forward(obs)
eng.dhead[:batch * 32].normal_()
stream = _lib.stream_ptr(dev)
for _ in range(30):  # steady clocks: whole backward passes
  eng.backward(obs)
_lib.call("dx_cnn_stage", ctypes.byref(eng.ctx), 7, _lib.ptr(obs), 1, None, batch, stream)
torch.cuda.synchronize()
nwg = 250
raw = eng.slabs.view(torch.int64)
# the linear layer's slabs start at the plan's w_off: find the stamps by their stage count field
arr = raw.cpu().numpy()
idx = np.where((arr[2::4][: (arr.size - 2) // 4] >= 40) & (arr[2::4][: (arr.size - 2) // 4] <= 60))[0]
start = idx[0] * 4
st = arr[start:start + nwg * 8 * 4].reshape(nwg, 8, 4)
cyc, ticks, stages, entry = st[..., 0], st[..., 1], st[..., 2], st[..., 3]
per_stage = cyc / stages
print("cycles per stage per wave: median %.0f  min %.0f  max %.0f  (ideal 2 waves x 64 MFMA x 64 = 8192)"
      % (np.median(per_stage), per_stage.min(), per_stage.max()))
print("in-loop clock: median %.3f GHz" % np.median(cyc / ticks * 0.1))
print("loop duration: median %.1f us, max %.1f us" % (np.median(ticks) / 100, ticks.max() / 100))
e = entry[:, 0] - entry.min()
print("workgroup entry spread: median %.1f us  max %.1f us" % (np.median(e) / 100, e.max() / 100))
print("last workgroup ends at %.1f us after the first entry" % ((entry[:, 0] - entry.min() + ticks.max(axis=1)).max() / 100))
