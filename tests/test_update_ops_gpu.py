"""GPU parity (through the C-ABI) of the HBM-bound update ops: advantage normalisation,
grad-norm clip + Adam / RMSprop, row gather, categorical sampling and the fused PPO / A2C
loss head, against the CPU oracle."""
import numpy as np
import numpy.testing as nt
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("n", [1, 7, 4096, 8192, 100003])
def test_advantage_normalisation(n):
  from derl_amd import ops
  rs = np.random.RandomState(n)
  adv = (rs.standard_normal(n) * 3 + 0.7).astype(np.float32)
  out = ops.adv_normalize(t(adv), 1e-8).cpu().numpy()
  nt.assert_allclose(out, oracle.normalize_advantages(adv), rtol=1e-5, atol=1e-6)
  # reduced statistics supplied by the caller (the sharded-batch path)
  stats = torch.tensor([adv.astype(np.float64).sum(), (adv.astype(np.float64) ** 2).sum(), n],
                       dtype=torch.float64, device=DEV)
  out2 = ops.adv_normalize(t(adv), 1e-8, stats=stats, stats_ready=True).cpu().numpy()
  nt.assert_allclose(out2, out, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n,seglen,indexed", [(1000, 333, True), (8192, 2048, True), (77, 77, False), (5, 2, True)])
def test_advantage_statistics_of_all_minibatches_in_one_launch(n, seglen, indexed):
  """dx_adv_stats_segments_f32: {sum, sumsq, count} of every minibatch slice of a permuted
  epoch == dx_adv_stats_f32 on the gathered minibatch, bit for bit (same summation order), and
  == float64 NumPy sums."""
  from derl_amd import ops
  rs = np.random.RandomState(n)
  a = (rs.randn(n) * 2 + 0.5).astype(np.float32)
  perm = rs.permutation(n).astype(np.int32) if indexed else None
  adv = torch.from_numpy(a).to(DEV)
  index = torch.from_numpy(perm).to(DEV) if indexed else None
  stats = ops.adv_stats_segments(adv, index, seglen).cpu().numpy()
  order = perm if indexed else np.arange(n)
  nseg = -(-n // seglen)
  assert stats.shape == (nseg, 3)
  for k in range(nseg):
    piece = a[order[k * seglen:(k + 1) * seglen]]
    single = ops.adv_stats(torch.from_numpy(piece).to(DEV)).cpu().numpy()
    nt.assert_array_equal(stats[k], single)
    p64 = piece.astype(np.float64)
    nt.assert_allclose(stats[k], [p64.sum(), (p64 ** 2).sum(), piece.size], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n,max_norm", [(11085, 0.5), (1686693, 0.5), (1000, None), (37, 100.0)])
def test_clip_adam_steps(n, max_norm):
  from derl_amd import ops
  rs = np.random.RandomState(n)
  p = rs.standard_normal(n).astype(np.float32)
  m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
  dp, dm, dv = t(p), t(m), t(v)
  norm_out = torch.zeros(1, device=DEV)
  for step in range(1, 4):
    g = (rs.standard_normal(n) * (0.01 * step)).astype(np.float32)
    dg = t(g)
    partials = ops.grad_sumsq(dg)
    ops.clip_adam_step(dp, dg, dm, dv, partials, max_norm, 2.5e-4 / step, step, eps=1e-5,
                       norm_out=norm_out)
    if max_norm is not None:
      (gc,), norm = oracle.clip_grad_norm([g], max_norm)
    else:
      gc, norm = g, float(np.sqrt((g.astype(np.float64) ** 2).sum()))
    p, m, v = oracle.adam_step(p, gc, m, v, step, 2.5e-4 / step, eps=1e-5)
    nt.assert_allclose(norm_out.item(), norm, rtol=1e-6)
    nt.assert_allclose(dg.cpu().numpy(), gc, rtol=1e-6, atol=1e-9)  # clipped in place
    nt.assert_allclose(dp.cpu().numpy(), p, rtol=0, atol=1e-6)  # 1e-6 per step
    nt.assert_allclose(dm.cpu().numpy(), m, rtol=1e-5, atol=1e-9)
    nt.assert_allclose(dv.cpu().numpy(), v, rtol=1e-5, atol=1e-12)


def test_clip_rmsprop_steps():
  from derl_amd import ops
  n = 50001
  rs = np.random.RandomState(4)
  p = rs.standard_normal(n).astype(np.float32)
  s = np.zeros(n, np.float32)
  dp, ds = t(p), t(s)
  for step in range(1, 4):
    g = rs.standard_normal(n).astype(np.float32) * 0.05
    dg = t(g)
    ops.clip_rmsprop_step(dp, dg, ds, ops.grad_sumsq(dg), 0.5, 7e-4, 0.99, 1e-5)
    (gc,), _ = oracle.clip_grad_norm([g], 0.5)
    p, s = oracle.rmsprop_step(p, gc, s, 7e-4, 0.99, 1e-5)
    nt.assert_allclose(dp.cpu().numpy(), p, rtol=0, atol=2e-6)
    nt.assert_allclose(ds.cpu().numpy(), s, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("shape,dtype", [((50, 84, 84, 4), np.uint8), ((1000,), np.float32),
                                         ((300, 1), np.float32), ((77,), np.int64),
                                         ((64, 17), np.float32), ((10, 3), np.uint8)])
def test_gather_rows(shape, dtype):
  from derl_amd import ops
  rs = np.random.RandomState(len(shape))
  src = (rs.uniform(0, 255, size=shape)).astype(dtype)
  idx = rs.permutation(shape[0])[: max(1, shape[0] // 2)].astype(np.int32)
  out = ops.gather_rows(t(src), t(idx)).cpu().numpy()
  nt.assert_array_equal(out, src[idx])  # bit-exact: byte movement


@pytest.mark.parametrize("narrays", [1, 5, 8, 11, 16, 19])
def test_gather_rows_multi(narrays):
  """Several per-sample arrays selected by one index vector in one launch (per 16 arrays)."""
  from derl_amd import ops
  rs = np.random.RandomState(narrays)
  n = 333
  specs = [((n,), np.float32), ((n, 1), np.float32), ((n,), np.int64), ((n, 3), np.uint8),
           ((n,), np.uint8), ((n, 17), np.float32), ((n, 2), np.float64), ((n, 5), np.int32),
           ((n, 1), np.float32), ((n,), np.int64), ((n, 7), np.uint8), ((n, 6), np.float32), ((n,), np.float32),
           ((n, 2), np.int64), ((n, 9), np.uint8), ((n, 4), np.float32), ((n,), np.uint8), ((n, 3), np.float64),
           ((n, 1), np.int32)][:narrays]
  sources = [rs.uniform(0, 255, size=shape).astype(dtype) for shape, dtype in specs]
  idx = rs.randint(0, n, size=200).astype(np.int32)  # repeats allowed
  outs = ops.gather_rows_multi([t(src) for src in sources], t(idx))
  assert len(outs) == narrays
  for src, out in zip(sources, outs):
    nt.assert_array_equal(out.cpu().numpy(), src[idx])  # bit-exact: byte movement


def padded_head(logits, values):
  B, A = logits.shape
  head = np.zeros((B, 32), np.float32)
  head[:, :A] = logits
  head[:, A] = values.reshape(-1)
  return head


@pytest.mark.parametrize("A", [2, 4, 6, 18, 31])
def test_categorical_act_with_given_uniforms(A):
  from derl_amd import ops
  rs = np.random.RandomState(A)
  B = 1001
  logits = (rs.standard_normal((B, A)) * 2).astype(np.float32)
  values = rs.standard_normal(B).astype(np.float32)
  u = rs.uniform(size=B).astype(np.float32)
  u[:4] = [0.0, 0.99999994, 0.5, 1e-12]
  actions, log_prob, vals = ops.categorical_act(t(padded_head(logits, values)), A, t(u))
  expected = oracle.categorical_sample_from_uniform(logits, u)
  got = actions.cpu().numpy()
  # expf on the device vs numpy may differ in the last bit: allow boundary flips only
  assert (got != expected).mean() < 2e-3
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, got)
  nt.assert_allclose(log_prob.cpu().numpy(), lp.numpy(), rtol=1e-5, atol=1e-6)
  nt.assert_array_equal(vals.cpu().numpy(), values)
  assert got.min() >= 0 and got.max() <= A - 1


def test_categorical_act_internal_generator_statistics():
  from derl_amd import ops
  A, B = 5, 1 << 16
  logits = np.tile(np.array([[0.3, -1.0, 1.2, 0.0, -2.5]], np.float32), (B, 1))
  head = t(padded_head(logits, np.zeros(B, np.float32)))
  p = np.exp(logits[0] - logits[0].max()); p /= p.sum()
  counts = np.zeros(A)
  for counter in range(4):
    a, _, _ = ops.categorical_act(head, A, None, seed=123, counter=counter)
    counts += np.bincount(a.cpu().numpy(), minlength=A)
  n = 4 * B
  sigma = np.sqrt(n * p * (1 - p))
  assert np.all(np.abs(counts - n * p) < 5 * sigma), (counts, n * p)
  # different counters / seeds give different draws; same (seed, counter) reproduces
  a0, _, _ = ops.categorical_act(head, A, None, seed=1, counter=0)
  a1, _, _ = ops.categorical_act(head, A, None, seed=1, counter=1)
  a0b, _, _ = ops.categorical_act(head, A, None, seed=1, counter=0)
  assert (a0 != a1).float().mean() > 0.3 and torch.equal(a0, a0b)


@pytest.mark.parametrize("A,B,mode,clip", [(6, 257, 0, 0.1), (4, 8192, 0, 0.2), (18, 100, 0, None),
                                           (6, 40, 1, None), (31, 9, 0, 0.1)])
def test_categorical_loss_forward_backward(A, B, mode, clip):
  from derl_amd import ops
  rs = np.random.RandomState(A * 1000 + B)
  logits = (rs.standard_normal((B, A)) * 2).astype(np.float32)
  values = rs.standard_normal((B, 1)).astype(np.float32)
  actions = rs.randint(0, A, B).astype(np.int64)
  adv = rs.standard_normal(B).astype(np.float32)
  lp, ent, _ = oracle.categorical_log_prob_entropy(logits, actions)
  old_lp = (lp.numpy() + rs.standard_normal(B) * 0.2).astype(np.float32)
  old_v = (values + rs.standard_normal((B, 1)) * 0.3).astype(np.float32)
  targ = (values + rs.standard_normal((B, 1))).astype(np.float32)
  vcoef, ecoef = (0.25, 0.01) if mode == 0 else (0.5, 0.01)
  dhead = torch.full((B, 32), 7.0, device=DEV)
  loss = ops.categorical_loss(t(padded_head(logits, values)), t(actions), t(old_lp), t(adv),
                              t(old_v.reshape(-1)), t(targ.reshape(-1)), A, mode, clip, vcoef, ecoef,
                              dhead).cpu().numpy()
  if mode == 0:
    terms = oracle.ppo_loss_terms(lp, ent, torch.from_numpy(values), old_lp, adv, old_v, targ, clip, vcoef, ecoef)
    dl, dv = oracle.ppo_head_grads(logits, actions, values, old_lp, adv, old_v, targ, clip, vcoef, ecoef)
  else:
    terms = oracle.a2c_loss_terms(lp, ent, torch.from_numpy(values), adv, targ, vcoef, ecoef)
    dl, dv = oracle.a2c_head_grads(logits, actions, values, adv, targ, vcoef, ecoef)
  nt.assert_allclose(loss[0], terms["loss"].item(), rtol=1e-5, atol=1e-6)
  nt.assert_allclose(loss[1], terms["policy_loss"].item(), rtol=1e-5, atol=1e-6)
  nt.assert_allclose(loss[2], terms["entropy"].item(), rtol=1e-5, atol=1e-6)
  nt.assert_allclose(loss[3], terms["value_loss"].item(), rtol=1e-5, atol=1e-6)
  nt.assert_allclose(loss[4], adv.mean(), rtol=1e-4, atol=1e-6)
  d = dhead.cpu().numpy()
  # rows whose ratio / value sits within float rounding of a clip boundary may pick the
  # other branch; everything else must match the closed form
  bad_rows = (np.abs(d[:, :A] - dl) > 1e-6 + 1e-4 * np.abs(dl)).any(1) | (
      np.abs(d[:, A] - dv) > 1e-6 + 1e-4 * np.abs(dv))
  assert bad_rows.sum() <= max(1, B // 2000), bad_rows.sum()
  nt.assert_array_equal(d[:, A + 1:], 0)
