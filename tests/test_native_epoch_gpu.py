"""The native-epoch path (Trainer._step_epoch -> dx_mlp_ppo_epoch / dx_cnn_ppo_epoch): every
minibatch update of an epoch enqueued from ONE C call, against the per-update path
(Trainer.native_epochs = False) and against the CPU oracle stepping the same minibatches
(derl/alg/common.py:66-78 inside derl/runners/onpolicy.py:44-62).

The tests count the native calls to prove the path was entered; most switch summary recording off
after each ``next()`` like bench.py does, one records through the native path on purpose."""
import numpy as np
import numpy.testing as nt
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


class DeviceVectorEnv:
  """Device-resident batched env with vector observations and Discrete actions (the contract of
  derl/env/env_batch.py:35-134 as the device runner uses it): counter-seeded torch generator."""
  host_rng_free = True

  def __init__(self, nenvs, obs_dim, num_actions, seed):
    from derl_amd.env.spaces import Box, Discrete
    self.nenvs, self.unwrapped = nenvs, self
    self.device = torch.device("cuda")
    self.observation_space = Box(-10., 10., (obs_dim,), np.float32)
    self.action_space = Discrete(num_actions)
    self.generator = torch.Generator(device=self.device)
    self.generator.manual_seed(seed)
    self.seed = seed

  def reset(self, out=None):
    shape = (self.nenvs,) + self.observation_space.shape
    out = torch.empty(shape, dtype=torch.float32, device=self.device) if out is None else out
    return out.normal_(generator=self.generator)

  def step(self, actions, out=None, rewards_out=None, resets_out=None):
    obs = self.reset(out)
    rewards = (actions.to(torch.float32) - 1.0) * torch.randn(self.nenvs, device=self.device,
                                                            generator=self.generator)
    resets = torch.rand(self.nenvs, device=self.device, generator=self.generator) < 0.05
    return obs, rewards_out.copy_(rewards), resets_out.copy_(resets), None


def make_alg(kind, native, nenvs, horizon, epochs, nmb):
  import derl_amd as derl
  torch.manual_seed(0)
  np.random.seed(11)
  if kind == "gaussian":
    env = derl.env.make("HalfCheetah-v3", nenvs=nenvs, seed=3)
  elif kind == "cnn":
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=nenvs, seed=3)
  elif kind == "cnn18":  # 18 actions: the full Atari set, still on the factored tail
    env = derl.env.make("SeaquestNoFrameskip-v4", nenvs=nenvs, seed=3)
  elif kind == "cnn20":  # beyond it: linear layer and heads layer by layer, the loss as its own launch
    from derl_amd.env.synthetic import SyntheticAtariEnv
    env = SyntheticAtariEnv(nenvs, 20, 3)
  else:
    env = DeviceVectorEnv(nenvs, 11, 5, seed=3)
  kwargs = derl.PPOFactory.get_kwargs("atari" if kind.startswith("cnn") else "mujoco")
  kwargs.update(nenvs=nenvs, num_runner_steps=horizon, num_epochs=epochs, num_minibatches=nmb,
                num_train_steps=nenvs * horizon * 4, entropy_coef=0.01)
  alg = derl.PPOFactory(**kwargs).make(env)
  alg.trainer.native_epochs = native
  calls = []
  inner = alg.trainer.optimizer.native_epoch

  def counted(loss_fn, context, **kwargs):
    calls.append(context.num_minibatches)
    return inner(loss_fn, context, **kwargs)

  alg.trainer.optimizer.native_epoch = counted
  return alg, calls


def run(kind, native, nenvs, horizon, epochs, nmb, rollouts, keep_host=False):
  import derl_amd as derl
  alg, calls = make_alg(kind, native, nenvs, horizon, epochs, nmb)
  per_epoch = -(-(nenvs * horizon) // ((nenvs * horizon) // nmb))
  start = {k: v.detach().clone() for k, v in alg.model.state_dict().items()}
  it = alg.runner.run()
  losses, advantages, terms, host, step_counts = [], [], [], [], []
  for i in range(rollouts * epochs * per_epoch):
    data = next(it)
    derl.summary.stop_recording()
    advantages.append(data["advantages"].clone())
    if i == per_epoch:
      after_first_epoch = alg.model.engine.params.clone()
    if keep_host and i < per_epoch:
      host.append({k: v.cpu().numpy() for k, v in data.items() if isinstance(v, torch.Tensor)})
      step_counts.append(alg.runner.step_count)
    losses.append(alg.step(data).clone())
    terms.append(alg.loss_fn.last_terms.clone())
  opt = alg.trainer.optimizer
  return dict(alg=alg, calls=calls, start=start, losses=torch.stack(losses), advantages=advantages,
              terms=torch.stack(terms), params=alg.model.engine.params.clone(), m=opt.exp_avg.clone(),
              v=opt.exp_avg_sq.clone(), steps=opt.step_count, host=host, step_counts=step_counts,
              per_epoch=per_epoch, after_first_epoch=after_first_epoch)


# 33 envs x 16 steps = 528 samples in 5 minibatches of 105 + a ragged sixth of 3
@pytest.mark.parametrize("kind", ["gaussian", "categorical", "cnn", "cnn18", "cnn20"])
def test_native_epoch_equals_per_update_path_and_oracle(kind):
  nenvs, horizon, epochs, nmb, rollouts = 33, 16, 2, 5, 2
  fast = run(kind, True, nenvs, horizon, epochs, nmb, rollouts, keep_host=True)
  slow = run(kind, False, nenvs, horizon, epochs, nmb, rollouts)
  assert fast["per_epoch"] == 6
  assert fast["calls"] == [6] * (rollouts * epochs), "the native epoch path was not entered"
  assert slow["calls"] == []
  assert fast["steps"] == slow["steps"] == rollouts * epochs * 6
  # same launches on the same buffers in the same order: bit-identical
  assert torch.equal(fast["losses"], slow["losses"])
  assert torch.equal(fast["terms"], slow["terms"])
  for a, b in zip(fast["advantages"], slow["advantages"]):  # incl. context.normalized slices
    assert torch.equal(a, b)
  assert torch.equal(fast["params"], slow["params"])
  assert torch.equal(fast["m"], slow["m"]) and torch.equal(fast["v"], slow["v"])
  if kind.startswith("cnn"):  # the conv path's oracle parity: the golden Trainer.step tests (same native call)
    # up to 18 actions: heads + loss + the heads' backward in one launch; 20: forward + separate loss launch + whole backward
    assert fast["alg"].model.engine.fused_heads() == (kind != "cnn20")
    return
  # the oracle on the first epoch's minibatches (incl. the ragged one), from the same start
  alg = fast["alg"]
  names = [k for k, _ in alg.model.named_parameters()]
  params = {k: v.cpu().numpy().astype(np.float32).copy() for k, v in fast["start"].items()}
  state = {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in params.items()}
  okind = "mlp" if kind == "gaussian" else "mlp_cat"
  for i, host in enumerate(fast["host"]):
    assert host["actions"].shape[0] == (105 if i < 5 else 3)
    terms, grads = oracle.ppo_loss_and_grads(params, host, okind, 0.2, 0.25, 0.01)
    nt.assert_allclose(fast["losses"][i].item(), terms["loss"], rtol=1e-4, atol=1e-5, err_msg=f"minibatch {i}")
    clipped, _ = oracle.clip_grad_norm([grads[k] for k in names], 0.5)
    lr = oracle.linear_anneal(3e-4, nenvs * horizon * 4, fast["step_counts"][i])
    for k, g in zip(names, clipped):
      params[k], state[k]["m"], state[k]["v"] = oracle.adam_step(
          params[k], g, state[k]["m"], state[k]["v"], i + 1, lr, eps=1e-5)
  got = alg.model.engine.named_views(fast["after_first_epoch"])
  for k in names:  # six Adam steps of <= lr each
    nt.assert_allclose(got[k].cpu().numpy(), params[k], rtol=0, atol=5e-6, err_msg=k)


def test_native_epoch_normalises_only_when_the_transform_opted_in():
  """A pipeline WITHOUT NormalizeAdvantages (or with another epsilon) must train on exactly what
  its per-update path sees: the native epoch takes its normalisation from the transform's own
  record in the EpochContext, not from a constant."""
  import derl_amd as derl
  from derl_amd.runners.onpolicy import IterateWithMinibatches, TransformInteractions
  from derl_amd.runners.trajectory_transforms import NormalizeAdvantages

  def rewire(alg, eps):
    # ppo_runner_wrap's chain: TransformInteractions([normalize]) <- IterateWithMinibatches <- ...
    iterate = alg.runner.runner
    assert isinstance(iterate, IterateWithMinibatches)
    if eps is None:
      alg.runner = iterate
    else:
      normalize = NormalizeAdvantages(epsilon=eps)
      iterate.prepare = normalize.prepare
      alg.runner = TransformInteractions(iterate, [normalize])
    return alg

  for eps in (None, 0.25):
    results = []
    for native in (True, False):
      alg, calls = make_alg("gaussian", native, 32, 16, 2, 4)
      rewire(alg, eps)
      it = alg.runner.run()
      seen = []
      for _ in range(8):
        data = next(it)
        derl.summary.stop_recording()
        seen.append(data["advantages"].clone())
        alg.step(data)
      results.append((alg.model.engine.params.clone(), seen, list(calls)))
    assert results[0][2] == [4, 4] and results[1][2] == []
    assert torch.equal(results[0][0], results[1][0]), f"eps={eps}: native epoch trained on other data"
    for a, b in zip(results[0][1], results[1][1]):
      assert torch.equal(a, b)


def test_native_epoch_declines_when_a_transform_edited_the_minibatch():
  """A transform that replaces ``log_prob`` (or the advantages) after NormalizeAdvantages ran makes
  the minibatch differ from the epoch's arrays: the trainer must go update by update."""
  import derl_amd as derl
  from derl_amd.runners.onpolicy import TransformInteractions

  def halve(key):
    def transform(trajectory):
      trajectory[key] = trajectory[key] * 0.5
    return transform

  for key in ("log_prob", "advantages"):
    alg, calls = make_alg("gaussian", True, 32, 16, 1, 4)
    alg.runner = TransformInteractions(alg.runner, [halve(key)])
    it = alg.runner.run()
    for _ in range(4):
      data = next(it)
      derl.summary.stop_recording()
      alg.step(data)
    assert calls == [], f"native epoch ran although a transform replaced '{key}'"
    assert alg.trainer.optimizer.step_count == 4


@pytest.mark.parametrize("kind", ["gaussian", "cnn"])
def test_native_epoch_never_silently_differs_from_one_update_per_step(kind):
  """derl's Trainer.step applies exactly one update (alg/common.py:66-78); the native epoch applies
  all of an epoch's at minibatch 0.  Every way a caller could OBSERVE that difference raises
  RuntimeError naming ``native_epochs = False`` instead of answering from post-epoch parameters:
  reading the policy / the model between two minibatch steps, stepping a minibatch twice or out of
  order, handing a later minibatch of the consumed epoch in edited form (it would be applied a
  second time), and starting the next epoch with minibatches of this one left over.  A run that
  steps every minibatch once, in order, is untouched (and may read the policy between EPOCHS)."""
  import derl_amd as derl
  nenvs, horizon = (32, 16) if kind == "gaussian" else (8, 16)

  def fresh():
    alg, calls = make_alg(kind, True, nenvs, horizon, 2, 4)
    it = alg.runner.run()
    first = next(it)
    derl.summary.stop_recording()
    alg.step(first)
    assert calls == [4] and alg.model.engine.open_epoch is not None
    return alg, it, first

  # (1) reading the policy or the model mid-epoch
  alg, it, first = fresh()
  probe = alg.runner.unwrapped._buffers["obs"][0] if hasattr(alg.runner.unwrapped, "_buffers") else None
  for read in (lambda: alg.model.state_dict(), lambda: alg.loss(first),
               lambda: alg.runner.policy.act(probe)):
    with pytest.raises(RuntimeError, match="native_epochs = False"):
      read()
  for _ in range(3):  # ... and the epoch can still be finished, after which reading is fine again
    alg.step(next(it))
  assert alg.model.engine.open_epoch is None
  alg.model.state_dict()
  alg.runner.policy.act(probe)
  assert alg.trainer.step_count == 4 == alg.trainer.optimizer.step_count
  # (2) the same minibatch twice / a minibatch skipped
  alg, it, first = fresh()
  with pytest.raises(RuntimeError, match="minibatch 0 was stepped where minibatch 1"):
    alg.step(first)
  alg, it, first = fresh()
  next(it)
  with pytest.raises(RuntimeError, match="minibatch 2 was stepped where minibatch 1"):
    alg.step(next(it))
  # (3) a later minibatch of the consumed epoch arrives edited: no second application of its update
  alg, it, first = fresh()
  second = next(it)
  second["log_prob"] = second["log_prob"] * 0.5
  before = alg.trainer.optimizer.step_count
  with pytest.raises(RuntimeError, match="arrived edited"):
    alg.step(second)
  assert alg.trainer.optimizer.step_count == before == 4
  # (4) the next epoch is started with minibatches of this one never stepped
  alg, it, first = fresh()
  for _ in range(3):
    next(it)
  with pytest.raises(RuntimeError, match="never stepped"):
    alg.step(next(it))
  # per-update mode has none of these restrictions (the reference's semantics, update by update)
  alg, calls = make_alg(kind, False, nenvs, horizon, 2, 4)
  it = alg.runner.run()
  first = next(it)
  derl.summary.stop_recording()
  alg.step(first)
  alg.model.state_dict()
  alg.step(first)
  assert calls == [] and alg.trainer.optimizer.step_count == 2


def test_mirrors_after_an_epoch_that_another_epoch_follows():
  """dx_cnn_epoch.more_epochs: the last update of an epoch that another epoch of the same rollout follows re-packs only
  what the training kernels read (the full pack runs once per rollout).  The engine knows: whatever else reads the
  mirrors between the epochs -- here the rollout's act step and the layer-by-layer forward -- packs first and sees the
  stepped parameters, and the epoch that follows trains on from the light mirrors (mirrors_current = 2)."""
  from derl_amd.cnn_engine import CnnEngine
  alg, calls = make_alg("cnn", True, 8, 16, 2, 4)
  import derl_amd as derl
  it = alg.runner.run()
  for _ in range(4):  # the first of the rollout's two epochs
    data = next(it)
    derl.summary.stop_recording()
    alg.step(data)
  eng = alg.model.engine
  assert calls == [4] and eng.open_epoch is None
  assert eng._packed_version is None and eng._train_packed_version == eng._version()
  fresh = CnnEngine(eng.num_actions, max_batch=64, device=eng.device)  # the same parameters, every mirror packed
  fresh.params.copy_(eng.params)
  fresh.mark_dirty()
  obs = torch.randint(0, 256, (8, 84, 84, 4), dtype=torch.uint8, device=eng.device, generator=torch.Generator(eng.device).manual_seed(5))
  uniforms = torch.rand(8, device=eng.device, generator=torch.Generator(eng.device).manual_seed(6))
  outs = []
  for e in (eng, fresh):
    actions = torch.empty(8, dtype=torch.int64, device=eng.device)
    log_prob, values = torch.empty(8, device=eng.device), torch.empty(8, device=eng.device)
    e.act(obs, actions, log_prob, values, uniforms=uniforms)
    outs.append((actions.clone(), log_prob.clone(), values.clone(), e.forward(obs).clone()))
  for a, b in zip(*outs):
    assert torch.equal(a, b)
  assert eng._packed_version == eng._version()  # act packed on demand
  # the second epoch trains on from the light mirrors: the same parameters as a run whose epochs all end with the full pack
  other, _ = make_alg("cnn", True, 8, 16, 2, 4)
  it2 = other.runner.run()
  for k in range(8):
    if k >= 4:
      alg.step(next(it))
    data2 = next(it2)
    derl.summary.stop_recording()
    data2.epoch[0].more_epochs = False  # (read when the epoch's first minibatch is stepped)
    other.step(data2)
  assert torch.equal(alg.model.engine.params, other.model.engine.params)


@pytest.mark.parametrize("switches", ["", "DX_FC_FACTORED=0", "DX_PACK_DIRECT=0"])
def test_back_to_back_epochs_train_on_from_the_light_mirrors(switches):
  """Two native epochs of one rollout with NOTHING in between (no act, no state_dict: the training loop's own shape):
  the second call must reach dx_cnn_ppo_epoch with mirrors_current = 2 -- the C side trusts the flag, a wrong 2 would
  train on stale mirrors without an error -- and end at the same parameters, bit for bit, as a run whose epochs all end
  with the full pack (more_epochs off: mirrors_current 1 on the second call).  128 samples in minibatches of 42: a ragged
  fourth minibatch of 2.  Also with the linear layer + heads layer by layer and with the general packs."""
  import os
  import ctypes
  import derl_amd as derl
  from derl_amd import _lib
  lib = _lib.load()
  names = [item.split("=")[0] for item in switches.split()]
  saved = {n: os.environ.get(n) for n in names}
  real_call = _lib.call
  try:
    for item in switches.split():
      os.environ[item.split("=")[0]] = item.split("=")[1]
    assert lib.dx_reload_env() == 0

    def two_epochs(light):
      alg, calls = make_alg("cnn", True, 8, 16, 2, 3)
      seen = []

      def spying_call(name, *args):
        if name == "dx_cnn_ppo_epoch":
          epoch = args[1]._obj
          seen.append((epoch.mirrors_current, epoch.more_epochs, epoch.mbsize, epoch.samples))
        return real_call(name, *args)

      _lib.call = spying_call
      try:
        it = alg.runner.run()
        for k in range(8):  # 2 epochs x (3 minibatches of 42 + one of 2)
          data = next(it)
          derl.summary.stop_recording()
          if not light:
            data.epoch[0].more_epochs = False  # (read when the epoch's first minibatch is stepped)
          alg.step(data)
      finally:
        _lib.call = real_call
      assert calls == [4, 4] and alg.model.engine.open_epoch is None
      return alg.model.engine.params.clone(), seen

    light_params, light_seen = two_epochs(True)
    full_params, full_seen = two_epochs(False)
    assert [s[:2] for s in light_seen] == [(1, 1), (2, 0)], light_seen  # the rollout's act packed everything; then the light pack
    assert [s[:2] for s in full_seen] == [(1, 0), (1, 0)], full_seen
    assert all(s[2:] == (42, 128) for s in light_seen + full_seen)
    assert torch.equal(light_params, full_params)
  finally:
    _lib.call = real_call
    for n, v in saved.items():
      if v is None:
        os.environ.pop(n, None)
      else:
        os.environ[n] = v
    lib.dx_reload_env()


def _scalars_of_a_run(native):
  """Every scalar the summaries record over one rollout's updates, as (tag, global_step, value)."""
  import derl_amd as derl
  alg, calls = make_alg("cnn", native, 8, 16, 2, 4)
  seen = []
  original = derl.summary.add_scalar

  def capture(tag, value, global_step=None, **kwargs):
    seen.append((tag, global_step, float(value)))

  derl.summary.add_scalar = capture
  try:
    it = alg.runner.run()
    for _ in range(8):
      alg.step(next(it))  # PeriodicSummaries armed recording for this (first) rollout
  finally:
    derl.summary.add_scalar = original
    derl.summary.stop_recording()
  return seen, calls


def test_native_epoch_records_the_same_summaries():
  """A recording rollout (every rollout of a default `derl ppo` run is one) takes the native epoch
  too: the loss terms of every minibatch, its pre-clip gradient norm and the learning rate come out
  with the tags, global steps and values of the per-update path (derl/alg/ppo.py:56-62,90-96,
  derl/alg/common.py:61-64)."""
  fast, calls = _scalars_of_a_run(True)
  slow, none = _scalars_of_a_run(False)
  assert calls == [4, 4] and none == []
  assert sorted(fast) == sorted(slow)
  tags = {t for t, _, _ in fast}
  assert {"ppo/loss", "ppo/policy_loss", "ppo/entropy", "ppo/value_loss", "ppo/grad_norm"} <= tags, tags
  assert sum(1 for t, _, _ in fast if t == "ppo/grad_norm") == 8


@pytest.mark.parametrize("nenvs", [16])
def test_a2c_update_goes_through_the_native_call(nenvs):
  """A2C has no minibatch epochs: each rollout's single update is handed to the same native call
  as a one-minibatch epoch (RMSprop inside dx_cnn_ppo_epoch) -- bit-identical to the per-update
  path over several rollouts."""
  import derl_amd as derl

  def run(native):
    torch.manual_seed(0)
    np.random.seed(5)
    env = derl.env.make("BreakoutNoFrameskip-v4", nenvs=nenvs, seed=2)
    kwargs = derl.A2CFactory.get_kwargs()
    kwargs.update(nenvs=nenvs, num_train_steps=nenvs * 5 * 6)
    alg = derl.A2CFactory(**kwargs).make(env)
    alg.trainer.native_epochs = native
    calls = []
    inner = alg.trainer.optimizer.native_epoch

    def counted(loss_fn, context, **kw):
      calls.append(context.num_minibatches)
      return inner(loss_fn, context, **kw)

    alg.trainer.optimizer.native_epoch = counted
    losses = []
    for data in alg.runner.run():
      derl.summary.stop_recording()
      losses.append(alg.step(data).item())
    return losses, alg.model.engine.params.clone(), alg.trainer.optimizer.square_avg.clone(), calls

  fast, slow = run(True), run(False)
  assert fast[3] == [1] * 6 and slow[3] == []
  assert fast[0] == slow[0] and torch.equal(fast[1], slow[1]) and torch.equal(fast[2], slow[2])


def test_persistent_epoch_matches_launch_per_stage_epoch_and_oracle():
  """The ONE-launch-per-epoch form (csrc/mlp_persist.hip: the model resident in LDS, two grid
  barriers per update) against the launch-per-stage epoch on the same rollouts: other partition of
  the gradient sums, so float32 rounding instead of bit equality -- losses 1e-5, parameters and Adam
  moments 1e-6 after 2 rollouts x 2 epochs x 4 updates (866-row minibatches = 28 row tiles + a ragged
  fourth of 2 rows) -- and against the CPU oracle stepping the first epoch's minibatches."""
  import derl_amd as derl
  nenvs, horizon, epochs, nmb, rollouts = 65, 40, 2, 3, 2

  def go(persistent):
    alg, calls = make_alg("gaussian", True, nenvs, horizon, epochs, nmb)
    alg.model.engine.persistent_epochs = persistent
    start = {k: v.detach().clone() for k, v in alg.model.state_dict().items()}
    it = alg.runner.run()
    losses, host, step_counts, after_first = [], [], [], None
    for i in range(rollouts * epochs * 4):
      data = next(it)
      derl.summary.stop_recording()
      if i == 4:
        after_first = alg.model.engine.params.clone()
      if i < 4:
        host.append({k: v.cpu().numpy() for k, v in data.items() if isinstance(v, torch.Tensor)})
        step_counts.append(alg.runner.step_count)
      losses.append(alg.step(data).clone())
    torch.cuda.synchronize()
    opt = alg.trainer.optimizer
    assert calls == [4] * (rollouts * epochs)
    return dict(alg=alg, start=start, losses=torch.stack(losses).cpu().numpy(), host=host, step_counts=step_counts,
                params=alg.model.engine.params.cpu().numpy(), m=opt.exp_avg.cpu().numpy(),
                v=opt.exp_avg_sq.cpu().numpy(), after_first=after_first,
                used=getattr(alg.model.engine, "_persist_ws", None) is not None)

  fast, slow = go(True), go(False)
  assert fast["used"] and not slow["used"], "the persistent epoch was not taken"
  assert fast["alg"].model.engine.last_epoch_route == "persistent"
  assert slow["alg"].model.engine.last_epoch_route == "per-stage"
  assert np.all(np.isfinite(fast["losses"])), "a grid barrier timed out"
  nt.assert_allclose(fast["losses"], slow["losses"], rtol=1e-5, atol=1e-6)
  nt.assert_allclose(fast["params"], slow["params"], rtol=0, atol=1e-6)
  nt.assert_allclose(fast["m"], slow["m"], rtol=1e-4, atol=1e-8)
  nt.assert_allclose(fast["v"], slow["v"], rtol=1e-4, atol=1e-10)
  for a, b in zip(fast["host"], slow["host"]):  # the same minibatches, the same normalised advantages
    for key in a:
      nt.assert_array_equal(a[key], b[key], err_msg=key)
  alg = fast["alg"]
  names = [k for k, _ in alg.model.named_parameters()]
  params = {k: v.cpu().numpy().astype(np.float32).copy() for k, v in fast["start"].items()}
  state = {k: dict(m=np.zeros_like(v), v=np.zeros_like(v)) for k, v in params.items()}
  for i, host in enumerate(fast["host"]):
    assert host["actions"].shape[0] == (866 if i < 3 else 2)
    terms, grads = oracle.ppo_loss_and_grads(params, host, "mlp", 0.2, 0.25, 0.01)
    nt.assert_allclose(fast["losses"][i], terms["loss"], rtol=1e-4, atol=1e-5, err_msg=f"minibatch {i}")
    clipped, _ = oracle.clip_grad_norm([grads[k] for k in names], 0.5)
    lr = oracle.linear_anneal(3e-4, nenvs * horizon * 4, fast["step_counts"][i])
    for k, g in zip(names, clipped):
      params[k], state[k]["m"], state[k]["v"] = oracle.adam_step(
          params[k], g, state[k]["m"], state[k]["v"], i + 1, lr, eps=1e-5)
  got = alg.model.engine.named_views(fast["after_first"])
  for k in names:
    nt.assert_allclose(got[k].cpu().numpy(), params[k], rtol=0, atol=5e-6, err_msg=k)


_GIVE_UP_CHILD = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {tests!r})
import derl_amd as derl
from derl_amd import _lib
from test_native_epoch_gpu import make_alg
assert _lib.LIB_PATH.endswith("libderl_amd_diag.so")
alg, calls = make_alg("gaussian", True, 65, 40, 2, 3)
engine, opt = alg.model.engine, alg.trainer.optimizer
it = alg.runner.run()
data = next(it)
derl.summary.stop_recording()
before = engine.params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()
loss = alg.step(data)          # enqueues the first epoch as one persistent launch
torch.cuda.synchronize()
assert engine.last_epoch_route == "persistent"
assert not np.isfinite(loss.item()), loss.item()                         # NaN losses ...
assert all(torch.equal(a, b) for a, b in zip(before, (engine.params, opt.exp_avg, opt.exp_avg_sq))), \
    "parameters or moments were stepped by an epoch whose barrier gave up"  # ... and nothing stepped
assert engine._persist_status_np[0] != 0
for _ in range(3):             # the rest of the epoch only hands out the (NaN) results
  alg.step(next(it))
try:
  alg.step(next(it))           # the next epoch: the library refuses, naming the barrier
except _lib.NativeError as error:
  assert "grid barrier" in str(error) and "NOT" in str(error), str(error)
  print("GAVE UP LOUDLY:", error)
else:
  raise AssertionError("the epoch after a give-up was accepted")
engine._persist_status_np = None   # past the engine's own check: the C-ABI itself refuses too
try:
  alg.trainer.optimizer.native_epoch(alg.loss_fn, data["state"]["epoch"][0])
except _lib.NativeError as error:
  assert "dx_mlp_ppo_epoch" in str(error) and "grid barrier" in str(error), str(error)
  print("C-ABI REFUSED:", error)
else:
  raise AssertionError("dx_mlp_ppo_epoch accepted a poisoned status word")
assert torch.equal(before[0], engine.params)
"""


def test_persistent_epoch_gives_up_loudly_and_steps_nothing():
  """The failure path of the persistent epoch's grid barrier, forced: the diag flavour of the
  library with DX_MLP_PERSIST_SPIN_LIMIT=0 makes every workgroup that is not the last to arrive
  give up on its first poll.  The epoch must leave parameters and Adam moments untouched, return
  NaN losses, and the next epoch must raise NativeError naming the barrier -- from the engine's
  host-side check and from dx_mlp_ppo_epoch itself (DX_ETIMEOUT)."""
  import os
  import subprocess
  import sys
  tests = os.path.dirname(os.path.abspath(__file__))
  root = os.path.dirname(tests)
  env = dict(os.environ, DERL_AMD_LIBRARY="diag", DX_MLP_PERSIST_SPIN_LIMIT="0")
  out = subprocess.run([sys.executable, "-c", _GIVE_UP_CHILD.format(root=root, tests=tests)], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  assert "GAVE UP LOUDLY" in out.stdout and "C-ABI REFUSED" in out.stdout, out.stdout[-2000:]
