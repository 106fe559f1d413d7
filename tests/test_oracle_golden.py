"""Pins the CPU oracle (oracle/) to the golden vectors produced by the unmodified
reference (tests/golden/generate.py) and to the reference's own fixtures
(tests/golden/upstream/).  CPU only."""
import os

import numpy as np
import numpy.testing as nt
import pytest
import torch

import oracle
from oracle import models as om
import inputs as gi

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
  return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("case", list(gi.GAE_CASES))
def test_gae_matches_reference(case):
  d = gi.gae_inputs(case)
  adv, vt = oracle.gae_advantages(d["rewards"], d["resets"], d["values"],
                                  d["last_values"], d["gamma"], d["lambda_"])
  if d["rewards"].ndim == 2:
    adv, vt = oracle.merge_time_batch(adv), oracle.merge_time_batch(vt)
  with load("gae.npz") as g:
    nt.assert_array_equal(adv, g[f"{case}.advantages"])  # bit-exact: same arithmetic
    nt.assert_array_equal(vt, g[f"{case}.value_targets"])
    assert adv.dtype == np.float32 and vt.dtype == np.float32


@pytest.mark.parametrize("case", list(gi.GAE_CASES))
def test_plain_c_gae_checker_is_pinned_to_the_reference_bit_for_bit(case):
  """oracle/gae.c (the checker of tests/test_gae_gpu.py's full-size cases) against the same reference-generated
  vectors as the NumPy restatement: bit-equal advantages and value targets on all seven cases.  (The reference feeds
  float64 rewards where the golden case has them; the C checker takes the engine's float32 rewards, so a case whose
  rewards are not exactly representable in float32 would differ -- none is.)"""
  import ctypes
  import __graft_entry__
  path = os.path.join(os.path.dirname(G), os.pardir, "oracle", "liboracle_gae.so")
  if not os.path.exists(path):
    __graft_entry__.build_oracle()
  lib = ctypes.CDLL(os.path.abspath(path))
  lib.oracle_gae_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                         ctypes.c_void_p, ctypes.c_void_p]
  d = gi.gae_inputs(case)
  r, z, v = d["rewards"], d["resets"], d["values"][..., 0]
  if r.ndim == 1:
    r, z, v = r[:, None], z[:, None], v[:, None]
  r32 = np.ascontiguousarray(r, np.float32)
  assert np.array_equal(r32.astype(r.dtype), r)  # the float32 hand-over loses nothing
  z8, v32 = np.ascontiguousarray(z, np.uint8), np.ascontiguousarray(v, np.float32)
  lv = np.ascontiguousarray(d["last_values"].reshape(-1), np.float32)
  T, N = r32.shape
  adv, vt = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
  assert lib.oracle_gae_f32(r32.ctypes.data, z8.ctypes.data, v32.ctypes.data, lv.ctypes.data, T, N,
                            float(d["gamma"]), float(d["lambda_"]), adv.ctypes.data, vt.ctypes.data) == 0
  with load("gae.npz") as g:
    nt.assert_array_equal(adv.reshape(-1), g[f"{case}.advantages"].reshape(-1))
    nt.assert_array_equal(vt.reshape(-1), g[f"{case}.value_targets"].reshape(-1))


def test_gae_whole_batch_normalisation():
  d = gi.gae_inputs("ragged")
  adv, _ = oracle.gae_advantages(d["rewards"], d["resets"], d["values"],
                                 d["last_values"], d["gamma"], d["lambda_"])
  with load("gae.npz") as g:
    nt.assert_allclose(oracle.normalize_advantages(adv),
                       g["ragged.normalized_advantages"], rtol=1e-6, atol=1e-6)


def test_gae_shape_errors():
  d = gi.gae_inputs("ragged")
  with pytest.raises(ValueError):
    oracle.gae_advantages(d["rewards"], d["resets"], d["values"][..., None],
                          d["last_values"])


@pytest.mark.parametrize("num_actions,seed", [(4, 21), (6, 22)])
def test_cnn_act_matches_reference(num_actions, seed):
  params = gi.nature_cnn_weights(num_actions, seed)
  obs = gi.frames(32, seed + 100)
  actions = np.random.RandomState(seed).randint(0, num_actions, size=32)
  logits, values = oracle.nature_cnn_forward(params, obs)
  log_prob, entropy, logp = oracle.categorical_log_prob_entropy(logits, actions)
  tag = f"cnn_a{num_actions}"
  with load("act.npz") as g:
    nt.assert_allclose(logits.numpy(), g[f"{tag}.raw_logits"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(logp.numpy(), g[f"{tag}.logits"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(values.numpy(), g[f"{tag}.values"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(log_prob.numpy(), g[f"{tag}.log_prob"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(entropy.numpy(), g[f"{tag}.entropy"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(om.nature_cnn_hidden(params, obs).numpy(), g[f"{tag}.hidden"],
                       rtol=1e-5, atol=1e-6)
    one = oracle.nature_cnn_forward(params, obs[:1])
    nt.assert_allclose(one[0].numpy()[0], g[f"{tag}.unbatched_logits"], rtol=1e-5, atol=1e-6)


def test_mlp_act_matches_reference():
  params = gi.mujoco_weights(17, 6, 23)
  mb = gi.mlp_minibatch(64, 17, 6, 123)
  mean, std, values = oracle.mujoco_forward(params, mb["observations"])
  log_prob, entropy = oracle.diag_normal_log_prob_entropy(mean, std, mb["actions"])
  with load("act.npz") as g:
    nt.assert_allclose(mean.numpy(), g["mlp.mean"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(std.numpy(), g["mlp.std"], rtol=1e-6)
    nt.assert_allclose(values.numpy(), g["mlp.values"], rtol=1e-5, atol=1e-6)
    nt.assert_allclose(log_prob.numpy(), g["mlp.log_prob"], rtol=1e-5, atol=1e-5)
    nt.assert_allclose(entropy.numpy(), g["mlp.entropy"], rtol=1e-6)
    assert list(g["mlp.rollout_keys"]) == ["actions", "log_prob", "values"]


def _check_summary(actual, g, prefix, rtol, atol):
  if f"{prefix}.full" in g.files:
    nt.assert_allclose(np.asarray(actual).reshape(-1), g[f"{prefix}.full"], rtol=rtol, atol=atol)
  else:
    flat = np.asarray(actual, np.float32).reshape(-1)
    nt.assert_allclose(flat[::gi.SAMPLE_STRIDE], g[f"{prefix}.sample"], rtol=rtol, atol=atol)
    nt.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g[f"{prefix}.norm"],
                       rtol=1e-5)
    nt.assert_allclose(flat.astype(np.float64).sum(), g[f"{prefix}.sum"],
                       rtol=1e-4, atol=atol * np.sqrt(flat.size) * 10)


def oracle_step_case(name):
  """Runs the oracle's restatement of NormalizeAdvantages + loss + Trainer.step."""
  cfg = gi.STEP_CASES[name]
  g = load(f"{name}.npz")
  if cfg["kind"] == "cnn":
    params = gi.nature_cnn_weights(cfg["num_actions"], cfg["seed"])
    mb = gi.cnn_minibatch(cfg["batch"], cfg["num_actions"], cfg["seed"] + 50)
  else:
    params = gi.mujoco_weights(cfg["obs_dim"], cfg["act_dim"], cfg["seed"])
    mb = gi.mlp_minibatch(cfg["batch"], cfg["obs_dim"], cfg["act_dim"], cfg["seed"] + 50)
  names = list(g["param_names"])
  data = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=g["data.log_prob"], advantages=mb["advantages"].copy(),
              values=g["data.values"], value_targets=g["data.value_targets"])
  if cfg["alg"] == "ppo":
    data["advantages"] = oracle.normalize_advantages(data["advantages"])
  return cfg, g, params, names, data


@pytest.mark.parametrize("name", list(gi.STEP_CASES))
def test_training_steps_match_reference(name):
  cfg, g, params, names, data = oracle_step_case(name)
  if cfg["alg"] == "ppo":
    nt.assert_allclose(data["advantages"], g["normalized_advantages"], rtol=1e-6, atol=1e-7)
  params = {k: np.asarray(params[k]) for k in names}
  state = {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in params.items()}
  step_count = cfg["step_count"]
  for step in range(cfg["nsteps"]):
    if step == 2:
      step_count += 4096
    if cfg["alg"] == "ppo":
      terms, grads = oracle.ppo_loss_and_grads(
          params, data, cfg["kind"], cfg["cliprange"], cfg["value_loss_coef"],
          cfg["entropy_coef"])
    else:
      terms, grads = oracle.a2c_loss_and_grads(
          params, data, cfg["kind"], cfg["value_loss_coef"], cfg["entropy_coef"])
    # A float32 restatement with another summation order can follow the reference's trajectory only
    # while the reference's own run stays clear of every ReLU boundary: generate.py records the
    # smallest |conv pre-activation| / layer scale of every step (relu_margin.<step>).  Where it
    # was below 3e-6 at or before a step, a unit's side was decided by rounding and (with RMSprop's
    # first normalised steps of ~10 lr per weight) parameters may sit 1e-4 off: the bounds are then
    # the wider ones -- keyed on the fixture's record, not on the case's name.  The tight per-step
    # statement for such cases is the same-start float64 comparison of the GPU suite.
    near_boundary = cfg["kind"] == "cnn" and \
        min(float(g[f"relu_margin.{s}"]) for s in range(step + 1)) < 3e-6

    def tight_else_wide(check):
      """check(loose) with the tight bounds; the wide ones only if the fixture says the reference
      passed a ReLU boundary within rounding at or before this step."""
      try:
        check(False)
      except AssertionError:
        if not near_boundary:
          raise
        check(True)

    tight_else_wide(lambda loose: nt.assert_allclose(terms["loss"], g["losses"][step],
                                                     rtol=2e-3 if loose else 1e-5, atol=1e-5))
    if step == 0:
      nt.assert_allclose(terms["loss"], g["loss0"], rtol=1e-6, atol=1e-6)
      for k in names:
        _check_summary(grads[k], g, f"grad0.{k}", rtol=1e-4, atol=2e-6)
    clipped, norm = oracle.clip_grad_norm([grads[k] for k in names], cfg["max_grad_norm"])
    if step == 0:
      nt.assert_allclose(norm, g["grad_norm0"], rtol=1e-5)
    lr = oracle.linear_anneal(cfg["lr"], cfg["num_train_steps"], step_count)
    nt.assert_equal(lr, g[f"lr.{step}"])
    for k, gk in zip(names, clipped):
      if cfg["alg"] == "ppo":
        params[k], state[k]["m"], state[k]["v"] = oracle.adam_step(
            params[k], gk, state[k]["m"], state[k]["v"], step + 1, lr,
            eps=cfg["optimizer_epsilon"])
      else:
        params[k], state[k]["v"] = oracle.rmsprop_step(
            params[k], gk, state[k]["v"], lr, cfg["optimizer_alpha"],
            cfg["optimizer_epsilon"])
      tight_else_wide(lambda loose, k=k: _check_summary(params[k], g, f"param{step}.{k}", rtol=1e-5,
                                                        atol=2e-4 if loose else 2e-6))


def test_closed_form_head_grads_match_autograd():
  """Appendix A.2/A.3/A.5 closed forms (what the fused HIP loss kernel implements)."""
  rs = np.random.RandomState(5)
  B, A = 257, 6
  logits = rs.standard_normal((B, A)).astype(np.float32) * 2
  values = rs.standard_normal((B, 1)).astype(np.float32)
  actions = rs.randint(0, A, B)
  adv = rs.standard_normal(B).astype(np.float32)
  lt = torch.tensor(logits, requires_grad=True)
  vt_ = torch.tensor(values, requires_grad=True)
  lp, ent, _ = oracle.categorical_log_prob_entropy(lt, actions)
  old_lp = lp.detach().numpy() + rs.standard_normal(B).astype(np.float32) * 0.2
  old_v = values + rs.standard_normal((B, 1)).astype(np.float32) * 0.3
  targ = values + rs.standard_normal((B, 1)).astype(np.float32)
  terms = oracle.ppo_loss_terms(lp, ent, vt_, old_lp, adv, old_v, targ, 0.1, 0.25, 0.01)
  terms["loss"].backward()
  dl, dv = oracle.ppo_head_grads(logits, actions, values, old_lp, adv, old_v, targ,
                                 0.1, 0.25, 0.01)
  nt.assert_allclose(dl, lt.grad.numpy(), rtol=1e-4, atol=1e-8)
  nt.assert_allclose(dv, vt_.grad.numpy()[:, 0], rtol=1e-4, atol=1e-8)
  lt.grad = None
  vt_.grad = None
  lp, ent, _ = oracle.categorical_log_prob_entropy(lt, actions)
  terms = oracle.a2c_loss_terms(lp, ent, vt_, adv, targ, 0.5, 0.01)
  terms["loss"].backward()
  dl, dv = oracle.a2c_head_grads(logits, actions, values, adv, targ, 0.5, 0.01)
  nt.assert_allclose(dl, lt.grad.numpy(), rtol=1e-4, atol=1e-8)
  nt.assert_allclose(dv, vt_.grad.numpy()[:, 0], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("tag,n,epochs,nmb", [("even", 1024, 3, 4), ("remainder", 1030, 2, 4)])
def test_minibatch_order_matches_reference(tag, n, epochs, nmb):
  np.random.seed(1234)
  with load("minibatch_order.npz") as g:
    for i, (_, _, idx) in enumerate(oracle.minibatch_indices(n, epochs, nmb)):
      nt.assert_array_equal(idx, g[f"{tag}.{i}"])
    assert f"{tag}.{i + 1}" not in g.files


def test_linear_anneal_matches_reference():
  with load("anneal.npz") as g:
    for tag, (start, nsteps) in dict(atari=(2.5e-4, 10e6), mujoco=(3e-4, 1e6)).items():
      for count, expected in zip(g[f"{tag}.counts"], g[f"{tag}.values"]):
        nt.assert_equal(oracle.linear_anneal(start, nsteps, int(count)), expected)


# ---- the reference's own fixtures (tests/golden/upstream) ---------------------------
# They depend on torch.manual_seed(0) + torch's CPU init reproducing the upstream
# weights (true for torch 2.10 here); skip with a message where it does not.

def _skip_unless_close(actual, expected, tol, what):
  if not np.allclose(actual, expected, rtol=0, atol=tol * 1e3):
    pytest.skip(f"seed-0 init of this torch build does not reproduce upstream {what}")


def test_upstream_dqn_base_outputs():
  torch.manual_seed(0)
  params = om.init_nature_cnn(output_units=())
  # NatureCNNBase() default init is torch's kaiming-uniform, not orthogonal
  torch.manual_seed(0)
  convs = [torch.nn.Conv2d(4, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2),
           torch.nn.Conv2d(64, 64, 3, 1)]
  linear = torch.nn.Linear(3136, 512)
  for i, c in enumerate(convs):
    params[f"base.conv-{i}.weight"], params[f"base.conv-{i}.bias"] = c.weight.detach(), c.bias.detach()
  params["base.linear.weight"], params["base.linear.bias"] = linear.weight.detach(), linear.bias.detach()
  inputs = torch.rand(32, 84, 84, 4)
  expected = np.load(os.path.join(G, "upstream", "dqn-base-outputs.npy"))
  out = om.nature_cnn_hidden(params, inputs).numpy()
  _skip_unless_close(out, expected, 1e-6, "dqn-base.pt")
  nt.assert_allclose(out, expected, atol=1e-6)  # models_test.py:41-45


def test_upstream_ppo_pybullet_loss_and_grads():
  torch.manual_seed(0)
  params = om.init_mujoco(26, (6, 1))
  with np.load(os.path.join(G, "upstream", "ppo_pybullet_interactions.npz")) as d:
    data = {k: d[k] for k in d.files}
  expected_loss = np.load(os.path.join(G, "upstream", "ppo_pybullet_losses.npy"))[0]
  data["advantages"] = data["advantages"].astype(np.float32)
  terms, grads = oracle.ppo_loss_and_grads(params, data, "mlp", 0.2, 0.25, 0.0)
  _skip_unless_close(terms["loss"], expected_loss, 1e-5, "MuJoCoModel(26,[6,1])")
  nt.assert_allclose(terms["loss"], expected_loss, rtol=1e-5, atol=1e-5)  # ppo_test.py:52-53
  with np.load(os.path.join(G, "upstream", "ppo_pybullet_grads.npz")) as eg:
    for i, k in enumerate(om.mujoco_keys(2)):
      nt.assert_allclose(grads[k], eg[f"grad_{i}"], rtol=1e-5, atol=1e-5)  # ppo_test.py:49-50


def test_upstream_a2c_atari_interactions_and_loss():
  torch.manual_seed(0)
  params = om.init_nature_cnn((6, 1))
  with np.load(os.path.join(G, "upstream", "a2c_atari_interactions.npz")) as d:
    data = {k: d[k] for k in d.files}
  logits, values = oracle.nature_cnn_forward(params, data["observations"])
  _skip_unless_close(values.numpy(), data["values"], 1e-6, "a2c model.pt")
  nt.assert_allclose(values.numpy(), data["values"], rtol=1e-6, atol=1e-6)  # a2c_test.py:19-21
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, data["actions"])
  nt.assert_allclose(lp.numpy(), data["log_prob"], rtol=1e-5, atol=1e-6)
  # GAE (T=5, N=8, lambda=1) with the bootstrap value from latest_observations
  _, last_values = oracle.nature_cnn_forward(params, data["latest_observations"])
  T, N = 5, 8
  adv, vt = oracle.gae_advantages(data["rewards"].reshape(T, N), data["resets"].reshape(T, N),
                                  data["values"].reshape(T, N, 1), last_values.numpy(),
                                  0.99, 1.0)
  nt.assert_allclose(oracle.merge_time_batch(adv), data["advantages"], rtol=1e-5, atol=1e-6)
  nt.assert_allclose(oracle.merge_time_batch(vt), data["value_targets"], rtol=1e-5, atol=1e-6)
  terms, _ = oracle.a2c_loss_and_grads(params, data, "cnn", 0.5, 0.01)
  expected = np.load(os.path.join(G, "upstream", "a2c_atari_losses.npy"))[0]
  nt.assert_allclose(terms["loss"], expected, rtol=1e-5, atol=1e-4)  # a2c_test.py:26-27
