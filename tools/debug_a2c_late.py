"""Per-step diagnosis of a golden Trainer.step trajectory on the GPU: for every step, the engine's
gradient against the float64 oracle evaluated FROM THE ENGINE'S OWN current parameters on the ReLU
branch the engine took, and the post-step parameters against the oracle's optimizer step from the
same start -- separates per-step kernel error from divergence of the trajectory."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle  # noqa: E402
from tests.test_ppo_e2e_gpu import build_case  # noqa: E402
from tests.test_cnn_gpu import engine_relu_masks, mask_disagreement  # noqa: E402
from tests.test_oracle_golden import _check_summary  # noqa: E402


def main(name="a2c_step_cnn_late"):
  cfg, g, names, data, model, alg, lr = build_case(name)
  eng = model.engine
  host = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
  batch = cfg["batch"]
  sq = {k: np.zeros(tuple(v.shape), np.float32) for k, v in model.state_dict().items()}
  for step in range(cfg["nsteps"]):
    if step == 2:
      alg.runner.step_count += 4096
    before = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    loss = alg.step(data).item()
    masks = engine_relu_masks(eng, batch)
    flipped, worst = mask_disagreement(before, host["observations"], masks)
    terms, grads = oracle.a2c_loss_and_grads(before, host, "cnn", cfg["value_loss_coef"], cfg["entropy_coef"],
                                             dtype=torch.float64, relu_masks=masks)
    clipped, norm = oracle.clip_grad_norm([grads[k] for k in names], cfg["max_grad_norm"])
    got = eng.named_views(eng.grads)
    lr_now = float(lr.get_tensor().item())
    print(f"step {step}: loss {loss:.8f} oracle {terms['loss']:.8f} golden {g['losses'][step]:.8f} "
          f"norm {alg.trainer.optimizer.grad_norm.item():.6f} oracle {norm:.6f} flips {flipped} worst {worst:.2e} lr {lr_now:.3e}")
    after = model.state_dict()
    for k, c in zip(names, clipped):
      gd = got[k].cpu().numpy()
      err = np.abs(gd - c)
      p_exp, sq[k] = oracle.rmsprop_step(before[k], c, sq[k], lr_now, cfg["optimizer_alpha"], cfg["optimizer_epsilon"])
      perr = np.abs(after[k].cpu().numpy() - p_exp)
      try:
        _check_summary(after[k].cpu().numpy(), g, f"param{step}.{k}", rtol=1e-5, atol=2e-6)
        gold = "golden ok"
      except AssertionError as e:
        gold = "GOLDEN MISMATCH " + str(e).split("Max absolute difference")[1].split("\n")[0]
      print(f"   {k:28s} |g|max {np.abs(c).max():.3e} grad err max {err.max():.3e} (rel to max {err.max() / np.abs(c).max():.2e}) "
            f"param err vs same-start oracle {perr.max():.3e}  {gold}")


if __name__ == "__main__":
  main(*sys.argv[1:])
