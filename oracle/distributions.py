"""Categorical / diagonal-Gaussian log-prob, entropy and sampling (CPU oracle).

Restates what torch.distributions computes at the reference's call sites
derl/policies.py:64,66,76-77 and derl/alg/ppo.py:35,54, derl/alg/a2c.py:23,33.
"""
import math
import numpy as np
import torch


def categorical_log_prob_entropy(logits, actions):
  """Categorical(logits): logp = logits - logsumexp; log_prob(a) = logp[a];
  entropy = -sum p*logp.  Returns (log_prob (B,), entropy (B,), logp (B,A))."""
  logits = torch.as_tensor(logits)
  logp = logits - torch.logsumexp(logits, -1, keepdim=True)
  actions = torch.as_tensor(actions).long()
  log_prob = logp.gather(-1, actions[..., None])[..., 0]
  entropy = -(logp.exp() * logp).sum(-1)
  return log_prob, entropy, logp


def diag_normal_log_prob_entropy(mean, std, actions):
  """Independent(Normal(mean, std), 1) (policies.py:40-42): sums over the last dim."""
  mean, std, actions = (torch.as_tensor(a) for a in (mean, std, actions))
  var = std ** 2
  log_prob = (-((actions - mean) ** 2) / (2 * var) - std.log()
              - math.log(math.sqrt(2 * math.pi))).sum(-1)
  entropy = (0.5 + 0.5 * math.log(2 * math.pi) + std.log()).sum(-1)
  return log_prob, entropy


def categorical_sample_from_uniform(logits, uniforms):
  """Inverse-CDF categorical sampling from given U(0,1) draws.

  torch.multinomial's stream is not reproducible across torch versions,
  so the build samples by inverse CDF from a counter-based uniform; this is the CPU
  statement of that rule: action = #{k : cdf_k <= u}, clipped to A-1, with
  p = softmax(logits) accumulated left to right in float32.
  """
  logits = np.asarray(logits, np.float32)
  m = logits.max(-1, keepdims=True)
  e = np.exp(logits - m).astype(np.float32)
  # sequential float32 accumulation, same order as the kernel
  cdf = np.zeros_like(e)
  acc = np.zeros(e.shape[:-1], np.float32)
  for k in range(e.shape[-1]):
    acc = (acc + e[..., k]).astype(np.float32)
    cdf[..., k] = acc
  thresh = (np.asarray(uniforms, np.float32) * acc).astype(np.float32)
  actions = (cdf <= thresh[..., None]).sum(-1)
  return np.minimum(actions, e.shape[-1] - 1).astype(np.int64)
