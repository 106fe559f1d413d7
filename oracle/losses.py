"""PPO and A2C losses with gradients (CPU oracle).

Follows derl/alg/ppo.py:24-108 (PPOLoss) and derl/alg/a2c.py:19-79 (A2CLoss); the
closed-form head gradients restate what autograd produces for them (SURVEY.md
Appendix A.1-A.5) and are cross-checked against autograd in the tests.
"""
import numpy as np
import torch

from .distributions import categorical_log_prob_entropy, diag_normal_log_prob_entropy
from .models import mlp_forward, nature_cnn_forward, mujoco_forward


def ppo_loss_terms(log_prob, entropy, values, old_log_prob, advantages, old_values,
                   value_targets, cliprange=0.1, value_loss_coef=0.25, entropy_coef=0.01):
  """Returns dict(policy_loss, entropy, value_loss, loss) as 0-dim torch tensors.

  ppo.py:45-53: ratio = exp(lp - lp_old); max(-ratio*A, -clip(ratio)*A) mean.
  ppo.py:54,64: minus entropy_coef * mean entropy.
  ppo.py:82-89: max((v - vt)^2, (v_old + clip(v - v_old) - vt)^2) mean.
  ppo.py:104: loss = policy + value_loss_coef * value.
  """
  t = torch.as_tensor
  log_prob, entropy, values = t(log_prob), t(entropy), t(values)
  old_log_prob, advantages = t(old_log_prob), t(advantages)
  old_values, value_targets = t(old_values), t(value_targets)
  if log_prob.shape != old_log_prob.shape or log_prob.shape != advantages.shape:
    raise ValueError("trajectory has mismatched shapes")  # ppo.py:36-43
  if values.shape != value_targets.shape:
    raise ValueError("trajectory has mismatched shapes")  # ppo.py:77-80
  ratio = torch.exp(log_prob - old_log_prob)
  pl = -ratio * advantages
  if cliprange is not None:
    pl = torch.max(pl, -torch.clamp(ratio, 1. - cliprange, 1. + cliprange) * advantages)
  policy_loss = pl.mean()
  ent = entropy.mean()
  vl = (values - value_targets) ** 2
  if cliprange is not None:
    vclipped = old_values + torch.clamp(values - old_values, -cliprange, cliprange)
    vl = torch.max(vl, (vclipped - value_targets) ** 2)
  value_loss = vl.mean()
  loss = (policy_loss - entropy_coef * ent) + value_loss_coef * value_loss
  return dict(policy_loss=policy_loss, entropy=ent, value_loss=value_loss, loss=loss)


def a2c_loss_terms(log_prob, entropy, values, advantages, value_targets,
                   value_loss_coef=0.5, entropy_coef=0.01):
  """a2c.py:32-33: -mean(lp*A) - ent_coef*mean(H); :57 mean((v-vt)^2); :74 sum."""
  t = torch.as_tensor
  log_prob, entropy, values = t(log_prob), t(entropy), t(values)
  advantages, value_targets = t(advantages), t(value_targets)
  if log_prob.shape != advantages.shape:
    raise ValueError("trajectory has mismatched shapes")  # a2c.py:26-29
  if values.shape != value_targets.shape:
    raise ValueError("trajectory has mismatched shapes")  # a2c.py:52-55
  policy_loss = -(log_prob * advantages).mean()
  ent = entropy.mean()
  value_loss = ((values - value_targets) ** 2).mean()
  loss = (policy_loss - entropy_coef * ent) + value_loss_coef * value_loss
  return dict(policy_loss=policy_loss, entropy=ent, value_loss=value_loss, loss=loss)


def ppo_head_grads(logits, actions, values, old_log_prob, advantages, old_values,
                   value_targets, cliprange, value_loss_coef, entropy_coef):
  """Closed-form dL/dlogits (B,A) and dL/dvalues (B,) for the categorical PPO loss
  (float64 NumPy; Appendix A.2/A.3)."""
  logits = np.asarray(logits, np.float64)
  B = logits.shape[0]
  m = logits.max(-1, keepdims=True)
  logp = logits - (m + np.log(np.exp(logits - m).sum(-1, keepdims=True)))
  p = np.exp(logp)
  a = np.asarray(actions).astype(np.int64)
  lp = logp[np.arange(B), a]
  H = -(p * logp).sum(-1)
  A = np.asarray(advantages, np.float64)
  ratio = np.exp(lp - np.asarray(old_log_prob, np.float64))
  l1 = -ratio * A
  if cliprange is None:
    active = np.ones(B, bool)
  else:
    l2 = -np.clip(ratio, 1. - cliprange, 1. + cliprange) * A
    inside = (ratio >= 1. - cliprange) & (ratio <= 1. + cliprange)
    active = (l1 > l2) | inside
  dlp = np.where(active, -A * ratio / B, 0.)
  onehot = np.zeros_like(p)
  onehot[np.arange(B), a] = 1.
  dlogits = dlp[:, None] * (onehot - p) + (-entropy_coef / B) * (-p * (logp + H[:, None]))
  v = np.asarray(values, np.float64).reshape(B)
  vt = np.asarray(value_targets, np.float64).reshape(B)
  vo = np.asarray(old_values, np.float64).reshape(B)
  e1 = (v - vt) ** 2
  if cliprange is None:
    vact = np.ones(B, bool)
  else:
    e2 = (vo + np.clip(v - vo, -cliprange, cliprange) - vt) ** 2
    vact = (e1 > e2) | (np.abs(v - vo) <= cliprange)
  dv = np.where(vact, value_loss_coef * 2. * (v - vt) / B, 0.)
  return dlogits, dv


def a2c_head_grads(logits, actions, values, advantages, value_targets,
                   value_loss_coef, entropy_coef):
  """Closed-form head gradients of the A2C loss (Appendix A.5)."""
  logits = np.asarray(logits, np.float64)
  B = logits.shape[0]
  m = logits.max(-1, keepdims=True)
  logp = logits - (m + np.log(np.exp(logits - m).sum(-1, keepdims=True)))
  p = np.exp(logp)
  a = np.asarray(actions).astype(np.int64)
  H = -(p * logp).sum(-1)
  A = np.asarray(advantages, np.float64)
  onehot = np.zeros_like(p)
  onehot[np.arange(B), a] = 1.
  dlogits = (-A / B)[:, None] * (onehot - p) + (-entropy_coef / B) * (-p * (logp + H[:, None]))
  v = np.asarray(values, np.float64).reshape(B)
  vt = np.asarray(value_targets, np.float64).reshape(B)
  dv = value_loss_coef * 2. * (v - vt) / B
  return dlogits, dv


def _leaf_params(params, dtype=torch.float32):
  return {k: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v)
          .detach().to(dtype).clone().requires_grad_(True) for k, v in params.items()}


def _cast_data(data, dtype):
  if dtype == torch.float32:
    return data
  out = dict(data)
  for key in ("log_prob", "advantages", "values", "value_targets"):
    if key in out:
      out[key] = torch.as_tensor(out[key]).to(dtype)
  if torch.as_tensor(out["actions"]).is_floating_point():
    out["actions"] = torch.as_tensor(out["actions"]).to(dtype)
  return out


def _forward_dist(params, data, kind, relu_masks=None):
  if kind == "cnn":
    logits, values = nature_cnn_forward(params, data["observations"], relu_masks)
    log_prob, entropy, _ = categorical_log_prob_entropy(logits, data["actions"])
  elif kind == "mlp_cat":
    # vector observations with Discrete actions (BASELINE config 1; not in the reference, whose
    # make_model cannot build it): the reference's MLP (models.py:224-237) once per output and its
    # categorical distribution math (policies.py:64,76-77)
    weight = params["module_list.0.0.weight"]
    x = torch.as_tensor(np.asarray(data["observations"])).to(weight.dtype)
    logits = mlp_forward(params, "module_list.0", x)
    values = mlp_forward(params, "module_list.1", x)
    log_prob, entropy, _ = categorical_log_prob_entropy(logits, data["actions"])
  else:
    mean, std, values = mujoco_forward(params, data["observations"])
    log_prob, entropy = diag_normal_log_prob_entropy(
        mean, std, torch.as_tensor(data["actions"]).to(mean.dtype))
  return log_prob, entropy, values


def ppo_loss_and_grads(params, data, kind="cnn", cliprange=0.1, value_loss_coef=0.25,
                       entropy_coef=0.01, dtype=torch.float32, relu_masks=None):
  """Full model forward + PPOLoss + autograd backward on CPU (ppo.py:100-108,
  common.py:68-70).  ``data`` holds observations, actions, log_prob, advantages,
  values (B,1), value_targets (B,1).  Returns (terms, grads dict keyed like params).

  ``dtype=torch.float64`` evaluates the same algorithm in double precision: the reference
  for large batches, where some pre-activation inevitably lies within float32 rounding of
  zero and float32 evaluations with different summation orders disagree on its ReLU mask
  (measured: torch-CPU's NCHW and channels-last float32 paths differ on one unit at batch
  130).  ``relu_masks``: see ``nature_cnn_forward``."""
  leaf = _leaf_params(params, dtype)
  data = _cast_data(data, dtype)
  log_prob, entropy, values = _forward_dist(leaf, data, kind, relu_masks)
  terms = ppo_loss_terms(log_prob, entropy, values, data["log_prob"], data["advantages"],
                         data["values"], data["value_targets"], cliprange,
                         value_loss_coef, entropy_coef)
  terms["loss"].backward()
  grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
           for k, v in leaf.items()}
  return {k: float(v.detach()) for k, v in terms.items()}, grads


def a2c_loss_and_grads(params, data, kind="cnn", value_loss_coef=0.5, entropy_coef=0.01,
                       dtype=torch.float32, relu_masks=None):
  """Full model forward + A2CLoss + autograd backward on CPU (a2c.py:68-79)."""
  leaf = _leaf_params(params, dtype)
  data = _cast_data(data, dtype)
  log_prob, entropy, values = _forward_dist(leaf, data, kind, relu_masks)
  terms = a2c_loss_terms(log_prob, entropy, values, data["advantages"],
                         data["value_targets"], value_loss_coef, entropy_coef)
  terms["loss"].backward()
  grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
           for k, v in leaf.items()}
  return {k: float(v.detach()) for k, v in terms.items()}, grads
