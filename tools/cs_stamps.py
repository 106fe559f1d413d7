"""In-kernel phase stamps of the rollout's conv-stack kernel (diag flavour): python3 tools/cs_stamps.py [batch ...]"""
import os
import sys

os.environ["DERL_AMD_LIBRARY"] = "diag"
os.environ.setdefault("DX_CS_DIAG", "0")  # the wave whose stamps are reported (0-7)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from derl_amd.cnn_engine import CnnEngine  # noqa: E402

dev = torch.device("cuda:0")
for batch in [int(b) for b in sys.argv[1:]] or [128, 256]:
  eng = CnnEngine(4, max_batch=max(batch, 64), device=dev)
  with torch.no_grad():
    eng.params.normal_(0, 0.02)
  eng.mark_dirty()
  obs = torch.randint(0, 256, (batch, 84, 84, 4), dtype=torch.uint8, device=dev)
  actions = torch.empty(batch, dtype=torch.int64, device=dev)
  log_prob, values = torch.empty(batch, device=dev), torch.empty(batch, device=dev)
  for _ in range(4):
    eng.act(obs, actions, log_prob, values)
  torch.cuda.synchronize()
