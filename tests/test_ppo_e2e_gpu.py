"""End-to-end GPU checks through derl's API surface: the reference's golden training
trajectories (three Trainer.steps), the device-resident runner's contract, and a short
PPOFactory run."""
import os

import numpy as np
import numpy.testing as nt
import pytest
import torch

import inputs as gi
import oracle

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


class FakeRunner:
  def __init__(self, policy, step_count):
    self.policy, self.step_count = policy, step_count


def build_case(name):
  import derl_amd as derl
  from derl_amd.optim import Adam, RMSprop
  from tests.test_oracle_golden import oracle_step_case
  cfg, g, params, names, data = oracle_step_case(name)
  data["advantages"] = gi.cnn_minibatch(cfg["batch"], cfg["num_actions"], cfg["seed"] + 50)["advantages"]
  model = derl.NatureCNNModel([cfg["num_actions"], 1], max_batch=64)
  model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
  policy = derl.ActorCriticPolicy(model)
  lr = derl.LinearAnneal(cfg["lr"], cfg["num_train_steps"], name="lr")
  if cfg["alg"] == "ppo":
    derl.NormalizeAdvantages()(data)
    optimizer = Adam(model, lr=lr.get_tensor(), eps=cfg["optimizer_epsilon"])
    trainer = derl.Trainer(optimizer, anneals=[lr], max_grad_norm=cfg["max_grad_norm"])
    alg = derl.PPO(FakeRunner(policy, cfg["step_count"]), trainer, cliprange=cfg["cliprange"],
                   value_loss_coef=cfg["value_loss_coef"], entropy_coef=cfg["entropy_coef"])
  else:
    optimizer = RMSprop(model, lr.get_tensor(), alpha=cfg["optimizer_alpha"],
                        eps=cfg["optimizer_epsilon"])
    trainer = derl.Trainer(optimizer, anneals=[lr], max_grad_norm=cfg["max_grad_norm"])
    alg = derl.A2C(FakeRunner(policy, cfg["step_count"]), trainer,
                   value_loss_coef=cfg["value_loss_coef"], entropy_coef=cfg["entropy_coef"])
  return cfg, g, names, data, model, alg, lr


def _golden_step_matches(g, step, loss, lr_value, model):
  """The reference's recorded step: loss to its own test tolerance (alg/a2c_test.py:27 uses 1e-4;
  1e-5 here), learning rate bit-equal, parameters to 2e-6.  Returns the failure text or None."""
  from tests.test_oracle_golden import _check_summary
  try:
    nt.assert_allclose(loss, g["losses"][step], rtol=1e-5, atol=1e-5)
    for k, p in model.named_parameters():
      _check_summary(p.detach().cpu().numpy(), g, f"param{step}.{k}", rtol=1e-5, atol=2e-6)
  except AssertionError as error:
    return str(error)
  nt.assert_equal(np.float32(lr_value), g[f"lr.{step}"])
  return None


@pytest.mark.parametrize("name", ["ppo_step_cnn", "a2c_step_cnn", "a2c_step_cnn_late"])
def test_trainer_steps_match_reference_golden(name):
  """alg/test.py:35-69 style: gradients after loss.backward(), then consecutive alg.step
  losses, learning rates and post-step parameters against the reference's own run -- and, step
  by step, against the float64 oracle started from the engine's OWN parameters on the ReLU branch
  the engine took (loss 1e-5, post-step parameters 2e-6 on every step of every case).

  The two comparisons answer different questions.  The same-start oracle bounds the error of ONE
  step of the kernels, whatever the trajectory does.  The golden run checks the trajectory, and
  it is only reproducible by another summation order while the reference's own float32 run stays
  clear of every ReLU boundary: generate.py records, per step, the smallest |conv pre-activation|
  relative to its layer's scale in the reference's run (``relu_margin.<step>``).  A step may
  leave the golden trajectory ONLY if that margin was below 3e-6 at it or before it (the unit's
  side is then decided by rounding; with RMSprop's first normalised steps of ~10 lr per weight
  that moves parameters by 1e-4, ``a2c_step_cnn``: margins 5e-7 / 3e-7 / 7e-7) -- never because
  of a tolerance chosen by case name."""
  from tests.test_cnn_gpu import engine_relu_masks, mask_disagreement
  from tests.test_oracle_golden import _check_summary
  import derl_amd as derl
  derl.summary.stop_recording()
  cfg, g, names, data, model, alg, lr = build_case(name)
  if cfg["alg"] == "ppo":
    nt.assert_allclose(data["advantages"].cpu().numpy(), g["normalized_advantages"], rtol=1e-5, atol=1e-6)
  assert [k for k, _ in model.named_parameters()] == list(g["param_names"])
  loss = alg.loss(data)
  loss.backward()
  nt.assert_allclose(loss.item(), g["loss0"], rtol=1e-5, atol=1e-5)
  for k, p in model.named_parameters():
    _check_summary(p.grad.cpu().numpy(), g, f"grad0.{k}", rtol=1e-4, atol=1e-5)
  alg.loss_fn.call_count = 0
  host = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in data.items()}
  state = None
  on_golden = True  # the engine's trajectory still IS the reference's
  for step in range(cfg["nsteps"]):
    if step == 2:
      alg.runner.step_count += 4096
    before = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    if state is None:
      state = {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in before.items()}
    loss = alg.step(data).item()
    lr_now = lr.get_tensor().item()
    # ---- one step of the kernels against the float64 oracle from the same start -------------
    masks = engine_relu_masks(model.engine, cfg["batch"])
    flipped, worst = mask_disagreement(before, host["observations"], masks)
    assert worst < 3e-6, f"step {step}: the engine's ReLU masks differ from float64 on a unit {worst:.1e} from zero"
    if cfg["alg"] == "ppo":
      terms, grads = oracle.ppo_loss_and_grads(before, host, "cnn", cfg["cliprange"], cfg["value_loss_coef"],
                                               cfg["entropy_coef"], dtype=torch.float64, relu_masks=masks)
    else:
      terms, grads = oracle.a2c_loss_and_grads(before, host, "cnn", cfg["value_loss_coef"], cfg["entropy_coef"],
                                               dtype=torch.float64, relu_masks=masks)
    nt.assert_allclose(loss, terms["loss"], rtol=1e-5, atol=1e-5, err_msg=f"step {step} (same-start oracle)")
    clipped, norm = oracle.clip_grad_norm([grads[k].astype(np.float32) for k in names], cfg["max_grad_norm"])
    nt.assert_allclose(alg.trainer.optimizer.grad_norm.item(), norm, rtol=1e-5)
    after = model.state_dict()
    for k, c in zip(names, clipped):
      if cfg["alg"] == "ppo":
        expect, state[k]["m"], state[k]["v"] = oracle.adam_step(
            before[k], c, state[k]["m"], state[k]["v"], step + 1, lr_now, eps=cfg["optimizer_epsilon"])
      else:
        expect, state[k]["v"] = oracle.rmsprop_step(before[k], c, state[k]["v"], lr_now,
                                                    cfg["optimizer_alpha"], cfg["optimizer_epsilon"])
      nt.assert_allclose(after[k].cpu().numpy(), expect, rtol=0, atol=2e-6,
                         err_msg=f"step {step} {k} (same-start oracle)")
    # ---- the trajectory against the reference's own run ------------------------------------
    nt.assert_equal(np.float32(lr_now), g[f"lr.{step}"])
    if on_golden:
      failure = _golden_step_matches(g, step, loss, lr_now, model)
      if failure is not None:
        margin = min(float(g[f"relu_margin.{s}"]) for s in range(step + 1))
        assert margin < 3e-6, (f"step {step} left the golden trajectory although the reference's run stayed "
                               f"{margin:.1e} clear of every ReLU boundary:\n{failure}")
        on_golden = False
  if "min_relu_margin" in cfg or cfg["alg"] == "ppo":
    assert on_golden, "a case whose reference run is reproducible must stay on the golden trajectory"
  assert alg.trainer.step_count == cfg["nsteps"]


def test_loss_shape_and_key_errors():
  import derl_amd as derl
  cfg, g, names, data, model, alg, lr = build_case("ppo_step_cnn")
  bad = dict(data)
  del bad["advantages"]
  with pytest.raises(ValueError, match="advantages"):
    alg.loss(bad)
  bad = dict(data, value_targets=data["value_targets"].reshape(-1))
  with pytest.raises(ValueError, match="mismatched shapes"):
    alg.loss(bad)
  bad = dict(data, advantages=np.zeros((cfg["batch"], 1), np.float32))
  with pytest.raises(ValueError, match="mismatched shapes"):
    alg.loss(bad)
  policy = alg.runner.policy
  with pytest.raises(NotImplementedError):
    policy.act(data["observations"], state=object())


def test_policy_act_contract_and_broadcast():
  import derl_amd as derl
  model = derl.NatureCNNModel([6, 1])
  policy = derl.ActorCriticPolicy(model)
  obs = gi.frames(5, 3)
  act = policy.act(obs)  # host input -> NumPy outputs (policies.py:78-80)
  assert list(act.keys()) == ["actions", "log_prob", "values"]
  assert act["actions"].shape == (5,) and act["actions"].dtype == np.int64
  assert act["log_prob"].shape == (5,) and act["log_prob"].dtype == np.float32
  assert act["values"].shape == (5, 1)
  one = policy.act(obs[0])  # unbatched input: batch dimension stripped (models_test.py:76-81)
  assert one["actions"].shape == () and one["values"].shape == (1,)
  dev = policy.act(torch.from_numpy(obs).to(DEV))
  assert dev["actions"].is_cuda
  train = policy.act(dict(observations=obs), training=True)
  assert list(train.keys()) == ["distribution", "values"]
  logits, values = oracle.nature_cnn_forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, obs)
  nt.assert_allclose(train["values"].cpu().numpy(), values.numpy(), rtol=1e-4, atol=1e-5)
  lp, ent, _ = oracle.categorical_log_prob_entropy(logits, act["actions"])
  nt.assert_allclose(train["distribution"].log_prob(act["actions"]).cpu().numpy(), lp.numpy(), rtol=1e-4, atol=1e-5)
  nt.assert_allclose(act["log_prob"], lp.numpy(), rtol=1e-4, atol=1e-5)
  outs = model(obs[0])  # (84,84,4) -> (A,), (1,)
  assert outs[0].shape == (6,) and outs[1].shape == (1,)


def test_model_init_matches_reference_seeded_init():
  import derl_amd as derl
  torch.manual_seed(0)
  model = derl.NatureCNNModel([6, 1])
  ref = oracle.init_nature_cnn((6, 1), seed=0)
  for k, v in model.state_dict().items():
    nt.assert_array_equal(v.cpu().numpy(), ref[k].numpy())
  assert sum(p.numel() for p in model.parameters()) == 1_687_719  # A = 6
  for name, p in model.named_parameters():
    if name.endswith("bias"):
      assert float(p.abs().max()) == 0.0


def test_device_runner_contract_and_ppo_factory_iterations():
  import derl_amd as derl
  derl.summary.stop_recording()
  torch.manual_seed(0)
  np.random.seed(0)
  env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=16, seed=0)
  kwargs = derl.PPOFactory.get_kwargs()
  kwargs.update(nenvs=16, num_runner_steps=8, num_train_steps=16 * 8 * 3)
  alg = derl.PPOFactory(**kwargs).make(env)
  runner = alg.runner
  assert runner.nenvs == 16 and runner.horizon == 8 and len(runner) == 384
  seen = 0
  losses = []
  for data in runner.run():
    assert set(data.keys()) == {"observations", "actions", "log_prob", "values", "rewards", "resets",
                                "infos", "next_observations", "state", "advantages", "value_targets"}
    assert data["actions"].shape == (32,) and data["advantages"].shape == (32,)
    assert data["values"].shape == (32, 1) and data["value_targets"].shape == (32, 1)
    assert isinstance(data["observations"], derl.GatheredRows) and data["observations"].shape == (32, 84, 84, 4)
    adv = data["advantages"].cpu().numpy()
    assert abs(adv.mean()) < 1e-5 and abs(adv.std() - 1) < 1e-3
    losses.append(alg.step(data).item())
    seen += 1
  assert seen == 3 * 3 * 4 and runner.step_count == 384 and runner.is_exhausted()
  assert all(np.isfinite(losses))
  # rewards / resets of the synthetic env honour the contract
  buf = runner.unwrapped._buffers
  assert set(np.unique(buf["rewards"].cpu().numpy())) <= {-1.0, 0.0, 1.0}
  assert buf["resets"].dtype == torch.bool
  # GAE on the runner's buffers equals the oracle's recursion
  T, N = 8, 16
  last = alg.runner.policy.act(buf["obs"][T])["values"].cpu().numpy()
  adv_ref, _ = oracle.gae_advantages(buf["rewards"].cpu().numpy(), buf["resets"].cpu().numpy(),
                                     buf["values"].cpu().numpy(), last, 0.99, 0.95)
  traj = dict(rewards=buf["rewards"], resets=buf["resets"], values=buf["values"],
              state=dict(latest_observations=buf["obs"][T]))
  adv, vt = derl.GAE(alg.runner.policy, normalize=False)(traj)
  nt.assert_allclose(adv.cpu().numpy(), adv_ref, rtol=1e-5, atol=1e-5)
  with pytest.raises(ValueError, match="advantages"):
    derl.GAE(alg.runner.policy)(traj)


def test_minibatch_order_matches_reference():
  """Composed shuffles: same sample order as the reference's in-place shuffling."""
  import derl_amd as derl

  class R:
    env = policy = None
    horizon = nsteps = step_count = 0
    nenvs = 1

    def __init__(self, n):
      self.n = n

    def run(self, obs=None):
      yield dict(observations=torch.arange(self.n, device=DEV)[:, None].float(),
                 index=torch.arange(self.n, device=DEV), state=dict(latest_observations=None))

  with np.load(os.path.join(G, "minibatch_order.npz")) as g:
    for tag, (n, epochs, nmb) in dict(even=(1024, 3, 4), remainder=(1030, 2, 4)).items():
      np.random.seed(1234)
      it = derl.IterateWithMinibatches(R(n), num_epochs=epochs, num_minibatches=nmb)
      count = 0
      for i, mb in enumerate(it.run()):
        nt.assert_array_equal(mb["index"].cpu().numpy(), g[f"{tag}.{i}"])
        count += 1
      assert f"{tag}.{count}" not in g.files


@pytest.mark.parametrize("nenvs", [64, 512])  # 512 = the per-GPU shard of BASELINE config 5
def test_a2c_factory_config5_shard_matches_oracle_update(nenvs):
  """A2C at BASELINE config 5's shape (nenvs 4096 x nsteps 5 over 8 GPUs = 512 x 5 = 2560
  samples per GPU; lambda 1, no minibatching, RMSprop): GAE on the rollout, the first update's
  loss, its clipped gradients and the post-step parameters against the oracle evaluated on the
  same rollout (float64, on the ReLU branch the engine took -- see test_cnn_gpu)."""
  import derl_amd as derl
  from tests.test_cnn_gpu import engine_relu_masks, mask_disagreement, count_ambiguous_relu_units
  derl.summary.stop_recording()
  torch.manual_seed(0)
  env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=nenvs, seed=3)
  kwargs = derl.A2CFactory.get_kwargs()
  kwargs.update(nenvs=nenvs, num_train_steps=nenvs * 5 * 3)
  alg = derl.A2CFactory(**kwargs).make(env)
  engine = alg.model.engine
  weights0 = {k: v.detach().cpu().numpy().copy() for k, v in alg.model.state_dict().items()}
  names = list(weights0)
  losses = []
  batch = nenvs * 5
  for i, data in enumerate(alg.runner.run()):
    assert data["observations"].shape == (batch, 84, 84, 4) and data["advantages"].shape == (batch,)
    if i == 0:
      host = {k: v.cpu().numpy() for k, v in data.items() if isinstance(v, torch.Tensor)}
      buf = alg.runner.unwrapped._buffers
      logits, last = oracle.nature_cnn_forward(weights0, buf["obs"][5].cpu().numpy())
      adv_ref, vt_ref = oracle.gae_advantages(buf["rewards"].cpu().numpy(), buf["resets"].cpu().numpy(),
                                              buf["values"].cpu().numpy(), last.numpy(), 0.99, 1.0)
      nt.assert_allclose(host["advantages"], adv_ref.reshape(-1), rtol=1e-4, atol=1e-4)
      nt.assert_allclose(host["value_targets"], vt_ref.reshape(-1, 1), rtol=1e-4, atol=1e-4)
      step_count = alg.runner.step_count
    losses.append(alg.step(data).item())
    if i == 0:
      masks = engine_relu_masks(engine, batch)
      flipped, worst = mask_disagreement(weights0, host["observations"], masks)
      assert worst < 3e-6 and flipped <= count_ambiguous_relu_units(weights0, host["observations"])
      terms, grads = oracle.a2c_loss_and_grads(weights0, host, "cnn", 0.5, 0.01,
                                               dtype=torch.float64, relu_masks=masks)
      nt.assert_allclose(losses[0], terms["loss"], rtol=1e-5, atol=1e-5)  # a2c_test.py:27 is 1e-4
      clipped, norm = oracle.clip_grad_norm([grads[k] for k in names], 0.5)
      nt.assert_allclose(alg.trainer.optimizer.grad_norm.item(), norm, rtol=1e-5)
      got = engine.named_views(engine.grads)  # the fused step writes the clipped gradient back
      lr = oracle.linear_anneal(7e-4, nenvs * 5 * 3, step_count)
      assert lr == np.float32(alg.trainer.optimizer.current_lr())
      after = alg.model.state_dict()
      for k, g in zip(names, clipped):
        scale = np.abs(g).max()
        nt.assert_allclose(got[k].cpu().numpy(), g, rtol=1e-4, atol=1e-5 * scale + 1e-9, err_msg=k)
        expect, _ = oracle.rmsprop_step(weights0[k], g, np.zeros_like(g), lr, 0.99, 1e-5)
        nt.assert_allclose(after[k].cpu().numpy(), expect, rtol=0, atol=2e-6, err_msg=k)
  assert len(losses) == 3 and np.all(np.isfinite(losses)) and alg.runner.step_count == nenvs * 15
  assert alg.trainer.optimizer.step_count == 3


def test_ppo_factory_config3_first_epoch_matches_oracle():
  """BASELINE config 3 end to end: PPO HalfCheetah-v3 nenvs 2048 x nsteps 64 through PPOFactory
  (mujoco preset: 10 epochs x 32 minibatches of 4096, Gaussian MLP policy, Adam).  The first
  epoch's 32 updates against the oracle stepping the SAME minibatches in the same order from the
  same initial weights: minibatch order = the reference's shuffle of the rollout
  (runners/onpolicy.py:44-62), normalised advantages, every loss, the learning rate and the
  parameters after the 32 steps."""
  import derl_amd as derl
  derl.summary.stop_recording()
  torch.manual_seed(0)
  np.random.seed(7)
  nenvs, horizon, nmb = 2048, 64, 32
  env = derl.env.make("HalfCheetah-v3", nenvs=nenvs, seed=1)
  kwargs = derl.PPOFactory.get_kwargs("mujoco")
  kwargs.update(nenvs=nenvs, num_runner_steps=horizon, num_train_steps=nenvs * horizon * 2)
  assert kwargs["num_epochs"] == 10 and kwargs["num_minibatches"] == nmb
  alg = derl.PPOFactory(**kwargs).make(env)
  names = [k for k, _ in alg.model.named_parameters()]
  params = {k: v.detach().cpu().numpy().astype(np.float32).copy()
            for k, v in alg.model.state_dict().items()}
  state = {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in params.items()}
  np.random.seed(7)
  order = next(iter(oracle.minibatch_indices(nenvs * horizon, 1, 1)))[2]  # epoch 0's permutation
  np.random.seed(7)
  worst = 0.0
  for i, data in enumerate(alg.runner.run()):
    if i == nmb:
      break
    assert data["actions"].shape == (4096, 6) and data["observations"].shape == (4096, 17)
    host = {k: (v.materialize() if isinstance(v, derl.GatheredRows) else v).cpu().numpy()
            for k, v in data.items() if isinstance(v, (torch.Tensor, derl.GatheredRows))}
    if i == 0:
      buf = alg.runner.unwrapped._buffers
      rollout = {k: buf[k].reshape((nenvs * horizon,) + tuple(buf[k].shape[2:])).cpu().numpy()
                 for k in ("actions", "log_prob", "values")}
      raw_adv, _ = oracle.gae_advantages(
          buf["rewards"].cpu().numpy(), buf["resets"].cpu().numpy(), buf["values"].cpu().numpy(),
          alg.runner.policy.act(buf["obs"][horizon])["values"].cpu().numpy(), 0.99, 0.95)
      raw_adv = raw_adv.reshape(-1)
      step_count = alg.runner.step_count
      assert step_count == nenvs * horizon
    idx = order[i * 4096:(i + 1) * 4096]
    for k in ("actions", "log_prob", "values"):  # the reference's shuffle-then-slice
      nt.assert_array_equal(host[k], rollout[k][idx], err_msg=k)
    nt.assert_allclose(host["advantages"], oracle.normalize_advantages(raw_adv[idx]),
                       rtol=1e-4, atol=2e-5)
    loss = alg.step(data).item()
    terms, grads = oracle.ppo_loss_and_grads(params, host, "mlp", 0.2, 0.25, 0.0)
    nt.assert_allclose(loss, terms["loss"], rtol=1e-4, atol=1e-5, err_msg=f"minibatch {i}")
    worst = max(worst, abs(loss - terms["loss"]) / max(abs(terms["loss"]), 1e-3))
    clipped, _ = oracle.clip_grad_norm([grads[k] for k in names], 0.5)
    lr = oracle.linear_anneal(3e-4, nenvs * horizon * 2, step_count)
    assert lr == np.float32(alg.trainer.optimizer.current_lr())
    for k, g in zip(names, clipped):
      params[k], state[k]["m"], state[k]["v"] = oracle.adam_step(
          params[k], g, state[k]["m"], state[k]["v"], i + 1, lr, eps=1e-5)
  assert alg.trainer.step_count == nmb
  # the route the bench runs: ONE persistent launch for the epoch (128 workgroups x 32 minibatches)
  from derl_amd import _lib
  assert alg.model.engine.last_epoch_route == "persistent" and _lib.load().dx_mlp_last_route() == 1, \
      "config 3 left the persistent epoch (csrc/mlp_persist.hip)"
  after = alg.model.state_dict()
  for k in names:  # 32 Adam steps of <= lr each: differences stay at rounding level
    nt.assert_allclose(after[k].cpu().numpy(), params[k], rtol=0, atol=2e-5, err_msg=k)
  print("config 3 first epoch: worst relative loss deviation", worst)


def test_cli_entry_point_runs_ppo(tmp_path):
  """`derl ppo --env-id ... --logdir ...` (scripts/derl:15-34) as a subprocess."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cmd = [sys.executable, os.path.join(root, "derl_amd", "scripts", "derl"), "ppo",
         "--env-id", "BreakoutNoFrameskip-v4", "--logdir", str(tmp_path), "--nenvs", "8",
         "--num-runner-steps", "8", "--num-train-steps", "128", "--nlogs", "2"]
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  args = (tmp_path / "args.txt").read_text()
  assert "env_id: BreakoutNoFrameskip-v4" in args and "num_minibatches: 4" in args
  bad = subprocess.run(cmd[:2] + ["dqn"] + cmd[3:], capture_output=True, text=True, timeout=60)
  assert bad.returncode != 0  # only the on-policy family is in scope


def test_fused_native_rollout_equals_per_step_loop():
  """dx_cnn_rollout_synth enqueues the same launches as the Python per-step loop: identical
  rollout buffers (same policy / env seeds and counters)."""
  import derl_amd as derl
  from derl_amd.policies import ActorCriticPolicy

  def rollout(use_fused):
    torch.manual_seed(0)
    model = derl.NatureCNNModel([4, 1], max_batch=64)
    policy = ActorCriticPolicy(model, seed=5)
    if not use_fused:
      policy.rollout_into = lambda *a, **k: False
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=24, seed=2)
    runner = derl.EnvRunner(env, policy, horizon=6, nsteps=24 * 6 * 2)
    outs = []
    for inter in runner.run():
      outs.append({k: v.clone() for k, v in inter.items() if isinstance(v, torch.Tensor)})
    return outs

  a, b = rollout(True), rollout(False)
  assert len(a) == len(b) == 2
  for x, y in zip(a, b):
    for k in x:
      assert torch.equal(x[k], y[k]), k


def test_fused_native_mlp_rollout_equals_per_step_loop():
  """dx_mlp_rollout_synth (the whole horizon of the Gaussian MLP policy against the MuJoCo-shaped synthetic env in
  ONE launch) writes the rollout buffers of the Python per-step loop (dx_mlp_forward + dx_normal_act_f32 +
  dx_synth_mujoco_step per step) bit for bit -- same policy / env seeds and counters -- over two rollouts, with an
  env count that is not a multiple of the kernel's 8 rows per workgroup."""
  import derl_amd as derl
  from derl_amd.policies import ActorCriticPolicy

  def rollout(use_fused):
    torch.manual_seed(0)
    env = derl.env.make("HalfCheetah-v3", nenvs=133, seed=3)
    model = derl.make_model(env.observation_space, env.action_space, 1)
    policy = ActorCriticPolicy(model, seed=9)
    taken = []
    if not use_fused:
      policy.rollout_into = lambda *a, **k: False
    else:
      inner = policy.rollout_into
      policy.rollout_into = lambda *a, **k: (taken.append(inner(*a, **k)), taken[-1])[1]
    runner = derl.EnvRunner(env, policy, horizon=7, nsteps=133 * 7 * 2)
    outs = []
    for inter in runner.run():
      outs.append({k: v.clone() for k, v in inter.items() if isinstance(v, torch.Tensor)})
    return outs, taken

  (a, taken), (b, _) = rollout(True), rollout(False)
  assert taken == [True, True]
  assert len(a) == len(b) == 2
  for x, y in zip(a, b):
    for k in x:
      assert torch.equal(x[k], y[k]), k
  obs = a[0]["observations"]
  assert float(obs.abs().max()) <= 10.0 and 0.9 < float(obs.std()) < 1.1 and abs(float(obs.mean())) < 0.05
  assert abs(float(a[0]["rewards"].mean())) < 0.2 and 0.8 < float(a[0]["rewards"].std()) < 1.2


def test_ppo_learns_cartpole():
  """The whole path as a learner, not only as arithmetic: PPO with BASELINE config 1's
  hyper-parameters (factory/ppo.py atari preset, nenvs 8 x 128 steps) on the built-in CartPole-v1
  -- rollout, GAE, minibatches, fused loss, backward, clip + Adam with the annealed rate -- raises
  the mean episode length from ~30 steps to several hundred.  60 iterations (61 k env steps)."""
  from tools.cartpole_learns import run
  lengths, _ = run(iterations=60, seed=0)
  first, last = float(np.mean(lengths[:5])), float(np.mean(lengths[-10:]))
  assert first < 60, lengths[:5]
  assert last > 100 and last > 3 * first, (first, last)


def test_ppo_cnn_learns_image_bandit():
  """The conv path as a learner (tools/quadrant_learns.py): device-resident runner, frames gathered
  by index inside the conv loader, fused categorical loss, conv backward, clip + Adam.  The frames
  show a bright quadrant = the rewarded action; the mean reward goes from chance (0.25) to > 0.9
  in 40 iterations of 64 envs x 16 steps."""
  from tools.quadrant_learns import run
  curve, _ = run(iterations=40, nenvs=64, horizon=16, seed=0, lr=1e-3)
  assert np.mean(curve[:2]) < 0.4, curve[:2]
  assert np.mean(curve[-5:]) > 0.9, curve[-5:]


def test_bench_line_contract():
  """bench.py prints ONE JSON line with the driver's keys, the roofline and cpu_baseline objects."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1",
                        "--cpu-nsteps", "4"], capture_output=True, text=True, timeout=600, cwd=root)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
    assert key in d, key
  assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["vs_baseline"] is None
  assert d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
  assert "workload" in d["config"] and "model" not in d["config"]
  roof = d["roofline"]
  for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
    assert key in roof, key
  assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
  assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.2 < roof["frac"] < 1.0
  base = d["cpu_baseline"]
  for key in ("value", "unit", "cores", "kind", "sample"):
    assert key in base, key
  assert base["kind"] == "port" and base["cores"] >= 1 and base["value"] > 0
  assert d["value"] > 10 * base["value"]  # BASELINE.json: >= 10x the reference CPU path
  assert d["gae_roofline"]["asymptote"]["frac"] > 0.55

  def fractions(node, path=""):  # every key named `frac` anywhere in the line is a fraction OF A CEILING: never above 1
    if isinstance(node, dict):
      for key, val in node.items():
        if key == "frac" and isinstance(val, (int, float)):
          yield path + "/frac", val
        else:
          yield from fractions(val, path + "/" + str(key))
    elif isinstance(node, list):
      for i, val in enumerate(node):
        yield from fractions(val, f"{path}[{i}]")

  over = [(where, val) for where, val in fractions(d) if not 0.0 <= val <= 1.0]
  assert not over, over
  assert "action" in d["roofline"]["native_rollout"]["note"]  # the one-launch rollout states what it leans on
  assert isinstance(d["roofline"]["power"], dict)  # read from profiles/ (or says that no probe is committed), never a literal


def test_a2c_cnn_learns_image_bandit():
  """The A2C route (GAE with lambda 1, fused A2C loss, RMSprop with the annealed rate) as a learner
  on the same image bandit: 300 rollouts of 64 envs x 5 steps at lr 5e-5 annealed to zero.  The
  task has a trap -- a policy that never tries one of the four actions scores 0.75 and stays there
  -- and RMSprop's first normalised steps decide whether a run falls into it: at the preset's 7e-4
  every run does, at 1e-4 it depends on float32 summation order (tools/a2c_probe.py: 2 of 5 seeds
  on one kernel route, 0 of 5 on another), at 5e-5 all five seeds reach 1.0 on every route.  Two
  seeds here."""
  from tools.quadrant_learns import run
  for seed in (0, 1):
    curve, _ = run(iterations=300, nenvs=64, horizon=5, seed=seed, lr=5e-5, algorithm="a2c")
    assert np.mean(curve[:5]) < 0.6, curve[:5]
    assert np.mean(curve[-20:]) > 0.9, (seed, curve[-20:])


@pytest.mark.parametrize("kind", ["cnn", "mlp"])
def test_inplace_writes_through_parameters_refresh_packed_weights(kind):
  """A Parameter re-pointed at a view of the flat buffer keeps its OWN version counter, so the
  engines watch every Parameter: writes through model.parameters() by a torch optimizer,
  nn.init, p.copy_() ... must reach the packed mirrors the kernels read (forward == oracle on
  the new weights, every time)."""
  import derl_amd as derl
  torch.manual_seed(0)
  if kind == "cnn":
    model = derl.NatureCNNModel([4, 1], max_batch=16)
    obs = gi.frames(8, 3)
    reference = lambda w: [o.numpy() for o in oracle.nature_cnn_forward(w, obs)]
  else:
    model = derl.MuJoCoModel(17, [6, 1])
    obs = gi.mlp_minibatch(8, 17, 6, 3)["observations"]
    reference = lambda w: [o.numpy() for o in oracle.mujoco_forward(w, obs)[::2]]

  def check(what):
    outs = model(obs)
    first, last = (outs[0], outs[-1]) if kind == "cnn" else (outs[0], outs[2])
    weights = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    expect = reference(weights)
    nt.assert_allclose(first.detach().cpu().numpy(), expect[0], rtol=1e-4, atol=2e-5, err_msg=what)
    nt.assert_allclose(last.detach().cpu().numpy(), expect[1], rtol=1e-4, atol=2e-5, err_msg=what)
    return first.detach().clone()

  before = check("initial")
  sgd = torch.optim.SGD(model.parameters(), lr=0.5)
  for p in model.parameters():
    p.grad.copy_(torch.randn_like(p) * 0.05)
  sgd.step()
  after = check("torch.optim.SGD.step on model.parameters()")
  assert (after - before).abs().max().item() > 1e-3
  with torch.no_grad():
    for p in model.parameters():
      if p.ndim >= 2:
        torch.nn.init.normal_(p, std=0.03)
  again = check("nn.init on the parameters")
  assert (again - after).abs().max().item() > 1e-4
  with torch.no_grad():
    next(iter(model.parameters())).mul_(0.5)
  check("in-place op on one parameter")


@pytest.mark.parametrize("horizon,rollouts", [(5, 2), (128, 1)])  # 128 x 256 = the BASELINE rollout, as bench.py runs it
def test_one_launch_native_rollout_matches_per_step_loop(horizon, rollouts):
  """dx_cnn_rollout_synth against the synthetic device env is ONE persistent launch of the conv-stack kernel (csrc/
  convstack.hip: one workgroup per env walks all `horizon` steps -- frame -> conv stack -> y2 Wc^T -> sample -> next
  frame never leaves the workgroup).  Against the Python per-step loop (one dx_cnn_act + one env launch per step, whole
  batch): observations, rewards and resets bit-identical at every step of every env, samples from the same stream
  positions (a logit that differs in the last bit may flip a sample on a CDF boundary), log-probs and values to float32
  rounding.  The 128-step case is the launch the benchmark times."""
  import derl_amd as derl
  from derl_amd.policies import ActorCriticPolicy

  def rollout(use_fused):
    torch.manual_seed(0)
    model = derl.NatureCNNModel([4, 1], max_batch=256)
    policy = ActorCriticPolicy(model, seed=5)
    if not use_fused:
      policy.rollout_into = lambda *a, **k: False
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=256, seed=2)
    runner = derl.EnvRunner(env, policy, horizon=horizon, nsteps=256 * horizon * rollouts)
    outs = []
    for inter in runner.run():
      outs.append({k: v.clone() for k, v in inter.items() if isinstance(v, torch.Tensor)})
    return outs

  a, b = rollout(True), rollout(False)
  assert len(a) == len(b) == rollouts
  for x, y in zip(a, b):
    assert x["observations"].shape[0] == horizon
    for k in ("observations", "rewards", "resets"):
      assert torch.equal(x[k], y[k]), k
    same = x["actions"] == y["actions"]
    assert (~same).float().mean().item() <= 2e-3
    for k in ("log_prob", "values"):
      if k in x:
        nt.assert_allclose(x[k][same].cpu().numpy(), y[k][same].cpu().numpy(), rtol=1e-4, atol=2e-5)
