"""In-kernel shader clock of the conv-stack kernels UNDER SUSTAINED LOAD (MI355X_MICROARCH.md, DVFS give-back item 6): the
diag flavour of the library runs the product kernels until DX_CS_DIAG is set, so each kernel is launched back to back for
a few seconds and THEN once with its stamps on (s_memtime against s_memrealtime around the launch, read by
launch_convstack / launch_convstack_train and printed on stderr).  One JSON line per kernel on stdout.
usage: DERL_AMD_LIBRARY=diag python3 tools/sustained_clock.py [seconds]"""
import json
import os
import re
import subprocess
import sys
import time

os.environ.setdefault("DERL_AMD_LIBRARY", "diag")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch
from derl_amd.cnn_engine import CnnEngine
dev = torch.device("cuda:0")
kind, seconds = sys.argv[1], float(sys.argv[2])
torch.manual_seed(0)
if kind == "train":
  batch = 8192
  eng = CnnEngine(4, max_batch=batch, device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  idx = torch.randperm(batch, device=dev).to(torch.int32)
  run = lambda: eng.forward_trunk(obs, idx)
else:
  nenvs, horizon = 256, 128
  eng = CnnEngine(4, max_batch=nenvs, device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  buffers = dict(obs=torch.randint(0, 256, (horizon + 1, nenvs, 84, 84, 4), dtype=torch.uint8, device=dev),
                 actions=torch.empty(horizon, nenvs, dtype=torch.int64, device=dev),
                 log_prob=torch.empty(horizon, nenvs, device=dev), values=torch.empty(horizon, nenvs, device=dev),
                 rewards=torch.empty(horizon, nenvs, device=dev),
                 resets=torch.empty(horizon, nenvs, dtype=torch.uint8, device=dev))
  count = [0]
  def run():
    eng.rollout_synth(buffers, horizon, nenvs, 7, count[0] * horizon, 11, count[0] * horizon, 0.05, 0.01)
    count[0] += 1
t0 = time.time()
while time.time() - t0 < seconds:
  for _ in range(20):
    run()
  torch.cuda.synchronize()
os.environ["DX_CS_DIAG"] = "0"
os.environ["DX_CS_STEP"] = "5"
run()
torch.cuda.synchronize()
"""


def main():
  seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
  for kind, kernel in (("train", "convstack_train_kernel (minibatch 8192)"), ("rollout", "convstack_roll_kernel (256 envs x 128 steps)")):
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), kind, str(seconds)], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ))
    m = re.findall(r"whole launch: (\d+) cycles per workgroup \((\d+) per (?:image|step)\) in ([\d.]+) us: shader clock (\d+) MHz", out.stderr)
    row = dict(kernel=kernel, after_seconds_of_back_to_back_launches=seconds)
    if m:
      cyc, per, us, mhz = m[-1]
      row.update(cycles_per_workgroup=int(cyc), cycles_per_image_or_step=int(per), launch_us_stamped_flavour=float(us), shader_clock_mhz=int(mhz))
    else:
      row["error"] = out.stderr[-400:]
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
  main()
