"""In-kernel phase stamps of the rollout's conv-stack kernel (diag flavour).
usage: [DX_CS_DIAG=<wave>] [DX_CS_STEP=<t>] python3 tools/cs_stamps.py [batch [horizon]]   (horizon > 1: the native rollout;
horizon 0: the forward of a training minibatch of `batch` images, step = the workgroup's t-th image)"""
import os
import sys

os.environ.setdefault("DERL_AMD_LIBRARY", "diag")
os.environ.setdefault("DX_CS_DIAG", "0")  # the wave whose stamps are reported (0-7)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = CnnEngine(4, max_batch=max(batch, 64), device=dev)
with torch.no_grad():
  eng.params.normal_(0, 0.02)
eng.mark_dirty()
if horizon == 0:
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  idx = torch.randperm(batch, device=dev).to(torch.int32)
  for _ in range(3):
    eng.forward_trunk(obs, idx)
elif horizon == 1:
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  actions = torch.empty(batch, dtype=torch.int64, device=dev)
  log_prob, values = torch.empty(batch, device=dev), torch.empty(batch, device=dev)
  for _ in range(3):
    eng.act(obs, actions, log_prob, values)
else:
  buffers = dict(obs=torch.randint(0, 256, (horizon + 1, batch, 84, 84, 4), dtype=torch.uint8, device=dev),
                 actions=torch.empty(horizon, batch, dtype=torch.int64, device=dev),
                 log_prob=torch.empty(horizon, batch, device=dev), values=torch.empty(horizon, batch, device=dev),
                 rewards=torch.empty(horizon, batch, device=dev),
                 resets=torch.empty(horizon, batch, dtype=torch.uint8, device=dev))
  for i in range(3):
    eng.rollout_synth(buffers, horizon, batch, 7, i * horizon, 11, i * horizon, 0.05, 0.01)
torch.cuda.synchronize()
