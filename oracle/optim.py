"""Gradient clipping, Adam / RMSprop steps and the linear LR anneal (CPU oracle, NumPy).

Follows derl/alg/common.py:56-78 (Trainer.step / preprocess_gradients, which calls
torch.nn.utils.clip_grad_norm_), derl/factory/ppo.py:74-83 (Adam, eps=1e-5, lr tensor),
derl/factory/a2c.py:64-75 (RMSprop alpha=.99) and derl/anneal.py:65-86 (LinearAnneal).
"""
import numpy as np


def clip_grad_norm(grads, max_norm):
  """Returns (clipped grads, total_norm).  torch clip_grad_norm_: n = sqrt(sum ||g||^2),
  g *= clamp(max_norm / (n + 1e-6), max=1) (common.py:59-60)."""
  grads = [np.asarray(g, np.float32) for g in grads]
  total = np.sqrt(np.sum([np.sum(g.astype(np.float64) ** 2) for g in grads]))
  total32 = np.float32(total)
  coef = np.float32(max_norm) / (total32 + np.float32(1e-6))
  coef = np.minimum(coef, np.float32(1.0))
  return [g * coef for g in grads], float(total32)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999,
              eps=1e-5):
  """One torch.optim.Adam update (no weight decay / amsgrad), ``step`` = 1,2,...
  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  Returns new
  (param, exp_avg, exp_avg_sq) as float32 arrays."""
  f = np.float32
  g = np.asarray(grad, f)
  m = (np.asarray(exp_avg, f) * f(beta1) + g * f(1 - beta1)).astype(f)
  v = (np.asarray(exp_avg_sq, f) * f(beta2) + g * g * f(1 - beta2)).astype(f)
  bc1 = 1.0 - beta1 ** step
  bc2 = 1.0 - beta2 ** step
  step_size = f(float(lr) / bc1)
  denom = (np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)).astype(f)
  p = (np.asarray(param, f) - step_size * (m / denom)).astype(f)
  return p, m, v


def rmsprop_step(param, grad, square_avg, lr, alpha=0.99, eps=1e-5):
  """One torch.optim.RMSprop update (no momentum, not centered):
  s = alpha*s + (1-alpha)*g^2; p -= lr * g / (sqrt(s) + eps)."""
  f = np.float32
  g = np.asarray(grad, f)
  s = (np.asarray(square_avg, f) * f(alpha) + g * g * f(1 - alpha)).astype(f)
  p = (np.asarray(param, f) - f(lr) * (g / (np.sqrt(s) + f(eps)))).astype(f)
  return p, s


def linear_anneal(start, nsteps, step_count, end=0.):
  """Closed form of LinearAnneal after ``step_count`` steps (anneal.py:77-86):
  clamp(float32(start + (end - start) * n / nsteps), min, max).  The reference loops n
  times; the last iteration's value is this expression (bit-equal)."""
  if step_count == 0:
    return np.float32(start)
  frac = step_count / nsteps
  val = np.float32(start + (end - start) * frac)
  return np.float32(np.clip(val, np.float32(min(start, end)), np.float32(max(start, end))))
