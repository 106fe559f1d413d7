"""Build-container tool (imports the unmodified reference like tests/golden/generate.py): scans
seeds of a golden Trainer.step case for the trajectory whose closest ReLU unit stays farthest
from zero, so that the fixture pins ALL its steps for any float32 summation order
(tests/golden/inputs.py, "a2c_step_cnn_late").

  python tools/scan_relu_margin.py a2c_step_cnn_late 14 1400

The schedule is replaced by an equivalent cheap one (the reference's LinearAnneal walks
`step_count` Python iterations); margins are float32 estimates -- generate.py records the float64
margins of the chosen seed and refuses values below the case's `min_relu_margin`."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
import generate as gen  # noqa: E402  (imports the reference as `gen.derl`)

derl = gen.derl


def margin(model, obs):
  weights = {k: v.detach() for k, v in model.state_dict().items()}
  x = (torch.from_numpy(obs).permute(0, 3, 1, 2).float() / 255).contiguous()
  best = 1.0
  for i, stride in enumerate((4, 2, 1)):
    x = F.conv2d(x, weights[f"base.conv-{i}.weight"], weights[f"base.conv-{i}.bias"], stride=stride)
    best = min(best, float(x.abs().min() / x.abs().max()))
    x = F.relu(x)
  return best


def trajectory_margin(cfg):
  model = gen.load_cnn(cfg["num_actions"], cfg["seed"])
  mb = gi.cnn_minibatch(cfg["batch"], cfg["num_actions"], cfg["seed"] + 50)
  policy = derl.ActorCriticPolicy(model)
  with torch.no_grad():
    act = policy.act(dict(observations=mb["observations"]), training=True)
    new_lp = act["distribution"].log_prob(torch.from_numpy(mb["actions"])).numpy()
    new_v = act["values"].numpy()
  data = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=(new_lp + mb["logp_noise"]).astype(np.float32), advantages=mb["advantages"].copy(),
              values=(new_v + mb["value_noise"]).astype(np.float32),
              value_targets=(new_v + mb["target_noise"]).astype(np.float32))
  lr_now = cfg["lr"] * (1 - cfg["step_count"] / cfg["num_train_steps"])
  span = 4096 / (cfg["num_train_steps"] - cfg["step_count"])  # relative decay of the one LR change
  lr = derl.LinearAnneal(lr_now, 4096 / span, 0., name="lr")
  opt = torch.optim.RMSprop(model.parameters(), lr.get_tensor(), alpha=cfg["optimizer_alpha"],
                            eps=cfg["optimizer_epsilon"])
  trainer = derl.alg.common.Trainer(opt, anneals=[lr], max_grad_norm=cfg["max_grad_norm"])
  alg = derl.A2C(gen.FakeRunner(policy, 0), trainer, value_loss_coef=cfg["value_loss_coef"],
                 entropy_coef=cfg["entropy_coef"])
  margins, losses = [], []
  for step in range(cfg["nsteps"]):
    margins.append(margin(model, mb["observations"]))
    if step == 2:
      alg.runner.step_count += 4096
    losses.append(alg.step(data).item())
  return margins, losses


def main(name, first, last):
  results = []
  for seed in range(int(first), int(last)):
    margins, losses = trajectory_margin(dict(gi.STEP_CASES[name], seed=seed))
    results.append((min(margins), seed, losses))
    print(seed, ["%.2e" % m for m in margins], ["%.3f" % l for l in losses], flush=True)
  results.sort(reverse=True)
  print("best:", results[:5])


if __name__ == "__main__":
  main(*sys.argv[1:4])
