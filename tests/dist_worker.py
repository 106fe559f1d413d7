"""Worker for the world_size-2 tests (launched by torch.distributed.run with the gloo backend).

mode cpu_math : the sharding rules of SURVEY.md 8e with the CPU oracle as the compute --
                per-rank gradients scaled by 1/global_batch, one SUM all-reduce, global
                advantage statistics from one 3-double all-reduce -- equal the single-process
                result on the concatenated batch.
mode gpu_step : two ranks sharing one GPU run Trainer.step on half a minibatch each through the
                HIP kernels; parameters after the step equal a single-process step on the
                whole minibatch.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
import oracle  # noqa: E402
from derl_amd import distributed  # noqa: E402


def global_normalize(adv_local):
  """NormalizeAdvantages under sharding: stats {sum, sumsq, count} all-reduced."""
  a = adv_local.astype(np.float64)
  stats = torch.tensor([a.sum(), (a ** 2).sum(), a.size], dtype=torch.float64)
  distributed.all_reduce_sum(stats)
  mean = stats[0].item() / stats[2].item()
  var = max(stats[1].item() / stats[2].item() - mean * mean, 0.0)
  return ((adv_local - np.float32(mean)) / (np.float32(np.sqrt(var)) + np.float32(1e-8))).astype(np.float32)


def cpu_math():
  world, rank = distributed.world_size(), distributed.rank()
  assert world == 2
  obs_dim, act_dim, B = 17, 6, 64
  weights = gi.mujoco_weights(obs_dim, act_dim, 5)
  mb = gi.mlp_minibatch(B, obs_dim, act_dim, 9)
  mean, std, vals = oracle.mujoco_forward(weights, mb["observations"])
  lp, _ = oracle.diag_normal_log_prob_entropy(mean, std, mb["actions"])
  full = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=(lp.numpy() + mb["logp_noise"]).astype(np.float32),
              advantages=mb["advantages"], values=(vals.numpy() + mb["value_noise"]).astype(np.float32),
              value_targets=(vals.numpy() + mb["target_noise"]).astype(np.float32))
  sl = slice(rank * B // world, (rank + 1) * B // world)
  shard = {k: v[sl] for k, v in full.items()}
  shard["advantages"] = global_normalize(shard["advantages"])
  ref = dict(full, advantages=oracle.normalize_advantages(full["advantages"]))
  np.testing.assert_allclose(shard["advantages"], ref["advantages"][sl], rtol=1e-5, atol=1e-6)
  _, grads = oracle.ppo_loss_and_grads(weights, shard, "mlp", 0.2, 0.25, 0.01, dtype=torch.float64)
  flat = torch.cat([torch.from_numpy(np.asarray(g, np.float64)).reshape(-1) for g in grads.values()])
  flat *= (B // world) / B  # what the loss kernel's 1/global_batch scaling does
  distributed.all_reduce_mean_grads(flat)
  _, gref = oracle.ppo_loss_and_grads(weights, ref, "mlp", 0.2, 0.25, 0.01, dtype=torch.float64)
  fref = torch.cat([torch.from_numpy(np.asarray(g, np.float64)).reshape(-1) for g in gref.values()])
  # the only difference is float32 rounding of the (globally) normalised advantages
  np.testing.assert_allclose(flat.numpy(), fref.numpy(), rtol=1e-5, atol=1e-7)
  # env sharding arithmetic used by bench.py
  nenvs_total = 256
  assert nenvs_total % world == 0 and sum(nenvs_total // world for _ in range(world)) == nenvs_total
  t = torch.tensor([float(rank + 1)])
  distributed.broadcast_(t, src=0)
  assert t.item() == 1.0
  print(f"rank {rank}: cpu_math OK", flush=True)


def bootstrap_agreement():
  """init_native_comm with the library calls replaced by scripted ones (no GPU, gloo): whatever
  fails on whichever rank, BOTH ranks leave the bootstrap the same way -- with the communicator,
  or with NativeError and their own communicator torn down -- and the next collective of the
  process group still matches up (nobody is left inside the id broadcast)."""
  from derl_amd import _lib
  world, rank = distributed.world_size(), distributed.rank()
  assert world == 2
  real_call = _lib.call
  log = []

  def scripted(fail_id_on=None, fail_init_on=None, unavailable_on=None, stale_library_on=None):
    def call(name, *args):
      log.append(name)
      if name == "dx_comm_available":
        if rank == unavailable_on:
          raise _lib.NativeError("scripted: RCCL cannot be loaded")
        if rank == stale_library_on:  # what _lib.load() raises for a .so from before the symbol existed
          raise AttributeError("scripted: undefined symbol: dx_comm_available")
        return 0
      if name == "dx_comm_unique_id":
        if rank == fail_id_on:
          raise _lib.NativeError("scripted: no unique id")
        ident = args[0]._obj
        for i in range(128):
          ident[i] = (37 * i + 11) % 251 + 1
        return 0
      if name == "dx_comm_init":
        assert bytes(args[0]._obj) == bytes((37 * i + 11) % 251 + 1 for i in range(128))
        assert (args[1], args[2]) == (rank, world)
        if rank == fail_init_on:
          raise _lib.NativeError("scripted: cannot join")
        return 0
      if name == "dx_comm_destroy":
        return 0
      return real_call(name, *args)
    return call

  def attempt(**script):
    del log[:]
    _lib.call = scripted(**script)
    try:
      distributed.init_native_comm()
      outcome = "native"
    except _lib.NativeError as error:
      outcome = "fallback: " + str(error)
    finally:
      _lib.call = real_call
    ones = torch.ones(3)
    torch.distributed.all_reduce(ones)  # would hang / mismatch if a rank were still in the bootstrap
    assert ones.tolist() == [2.0, 2.0, 2.0]
    native = distributed.native_comm()
    distributed._native = False  # pylint: disable=protected-access
    return outcome, native, list(log)

  # rank 1 cannot join: both fall back, rank 0 (which had joined) destroys its communicator
  outcome, native, calls = attempt(fail_init_on=1)
  assert outcome.startswith("fallback") and not native, (rank, outcome)
  assert calls == (["dx_comm_available", "dx_comm_unique_id", "dx_comm_init", "dx_comm_destroy"] if rank == 0
                   else ["dx_comm_available", "dx_comm_init"]), calls
  # rank 0 cannot join: the same, mirrored
  outcome, native, calls = attempt(fail_init_on=0)
  assert outcome.startswith("fallback") and not native, (rank, outcome)
  assert calls == (["dx_comm_available", "dx_comm_unique_id", "dx_comm_init"] if rank == 0
                   else ["dx_comm_available", "dx_comm_init", "dx_comm_destroy"]), calls
  # rank 0 cannot even make the id: it still broadcasts (zeros); nobody calls dx_comm_init
  outcome, native, calls = attempt(fail_id_on=0)
  assert outcome.startswith("fallback") and not native, (rank, outcome)
  assert calls == (["dx_comm_available", "dx_comm_unique_id"] if rank == 0 else ["dx_comm_available"]), calls
  # a rank that cannot join at all (RCCL not loadable ...): NOBODY makes an id or enters dx_comm_init --
  # ncclCommInitRank is a collective, the ready rank would wait in it for the other
  for who in (0, 1):
    outcome, native, calls = attempt(unavailable_on=who)
    assert outcome.startswith("fallback") and not native, (rank, outcome)
    assert calls == ["dx_comm_available"], calls
  # the library itself unusable on one rank (a stale .so: AttributeError, not NativeError): that rank still takes
  # part in the agreement and both fall back
  for who in (0, 1):
    outcome, native, calls = attempt(stale_library_on=who)
    assert outcome.startswith("fallback") and not native, (rank, outcome)
    assert calls == ["dx_comm_available"], calls
    assert ("AttributeError" in outcome) == (rank == who), (rank, outcome)
  # nothing fails: both ranks have the communicator
  outcome, native, calls = attempt()
  assert outcome == "native" and native, (rank, outcome)
  assert "dx_comm_destroy" not in calls
  print(f"rank {rank}: bootstrap_agreement OK", flush=True)


def bootstrap_real_failure():
  """The same agreement with the REAL library calls (gloo; with or without a GPU): rank 1's library is
  pointed at an RCCL that does not exist (DERL_AMD_RCCL_LIBRARY), so its dx_comm_available really fails in
  dlopen.  Both ranks must leave init_native_comm with NativeError, neither may have made an id or entered
  dx_comm_init (a collective the other rank would never join), and the process group's next collective
  still matches up."""
  from derl_amd import _lib
  world, rank = distributed.world_size(), distributed.rank()
  assert world == 2
  if rank == 1:
    os.environ["DERL_AMD_RCCL_LIBRARY"] = "/nonexistent/librccl-for-the-bootstrap-test.so"
  real_call, log = _lib.call, []

  def logging_call(name, *args):
    log.append(name)
    return real_call(name, *args)

  _lib.call = logging_call
  try:
    distributed.init_native_comm()
    outcome = "native"
  except _lib.NativeError as error:
    outcome = "fallback: " + str(error)
  finally:
    _lib.call = real_call
  assert outcome.startswith("fallback") and not distributed.native_comm(), (rank, outcome)
  assert log == ["dx_comm_available"], (rank, log)
  if rank == 1:
    assert "DERL_AMD_RCCL_LIBRARY" in outcome, outcome
    try:
      real_call("dx_comm_available")
      raise AssertionError("rank 1 loaded an RCCL that does not exist")
    except _lib.NativeError as error:
      assert error.status == _lib.DX_ENOSUP, error.status
  ones = torch.ones(3)
  torch.distributed.all_reduce(ones)
  assert ones.tolist() == [2.0, 2.0, 2.0]
  print(f"rank {rank}: bootstrap_real_failure OK", flush=True)


def gpu_step():
  import derl_amd as derl
  from derl_amd.optim import Adam
  world, rank = distributed.world_size(), distributed.rank()
  derl.summary.stop_recording()
  A, B = 4, 32
  weights = gi.nature_cnn_weights(A, 3)
  mb = gi.cnn_minibatch(B, A, 5)
  logits, vals = oracle.nature_cnn_forward(weights, mb["observations"])
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, mb["actions"])
  full = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=(lp.numpy() + mb["logp_noise"]).astype(np.float32),
              advantages=mb["advantages"], values=(vals.numpy() + mb["value_noise"]).astype(np.float32),
              value_targets=(vals.numpy() + mb["target_noise"]).astype(np.float32))

  class Runner:
    step_count = 1000

  def run(data, sharded):
    model = derl.NatureCNNModel([A, 1], max_batch=32)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    policy = derl.ActorCriticPolicy(model)
    runner = Runner()
    runner.policy = policy
    lr = derl.LinearAnneal(2.5e-4, 1e6, name="lr")
    trainer = derl.Trainer(Adam(model, lr=lr.get_tensor(), eps=1e-5), anneals=[lr], max_grad_norm=0.5)
    alg = derl.PPO(runner, trainer, cliprange=0.1, value_loss_coef=0.25, entropy_coef=0.01)
    data = dict(data)
    if sharded:
      derl.NormalizeAdvantages()(data)  # global statistics through the process group
      for _ in range(2):
        alg.step(data)
    else:
      # single-process reference on the whole batch: hide the process group from the step
      saved = distributed.world_size
      distributed.world_size = lambda: 1
      try:
        derl.NormalizeAdvantages()(data)
        for _ in range(2):
          alg.step(data)
      finally:
        distributed.world_size = saved
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}

  sl = slice(rank * B // world, (rank + 1) * B // world)
  sharded = run({k: v[sl] for k, v in full.items()}, True)
  single = run(full, False)
  for k in single:
    np.testing.assert_allclose(sharded[k], single[k], rtol=0, atol=2e-6, err_msg=k)
  print(f"rank {rank}: gpu_step OK", flush=True)


def gpu_minibatch_stats():
  """Global advantage statistics of every minibatch of a rollout from ONE all-reduce
  (NormalizeAdvantages.prepare) == the per-minibatch all-reduce path, bit for bit, and == the
  oracle's normalisation of the concatenated minibatch."""
  import derl_amd as derl
  from derl_amd.runners.onpolicy import IterateWithMinibatches, TransformInteractions
  world, rank = distributed.world_size(), distributed.rank()
  dev = torch.device("cuda", 0)
  S, epochs, nmb = 1000, 3, 3  # 333-sample minibatches + a remainder of one sample
  rng = np.random.RandomState(100 + rank)
  shard = dict(advantages=rng.randn(S).astype(np.float32) * 3 + rank, observations=np.arange(S, dtype=np.float32))

  class OneRollout:
    env = policy = None
    horizon, nsteps, step_count, nenvs = S, S, 0, None
    def is_exhausted(self):
      return False
    def run(self, obs=None):
      yield {k: torch.from_numpy(v).to(dev) for k, v in shard.items()}

  def collect(prepared):
    np.random.seed(7 + rank)
    norm = derl.NormalizeAdvantages()
    calls = [0]
    saved = distributed.all_reduce_sum
    def counting(t):
      calls[0] += 1
      return saved(t)
    distributed.all_reduce_sum = counting
    try:
      it = IterateWithMinibatches(OneRollout(), epochs, nmb, prepare=norm.prepare if prepared else None)
      out = [(mb["advantages"].cpu().numpy(), mb["observations"].cpu().numpy().astype(np.int64))
             for mb in TransformInteractions(it, [norm]).run()]
    finally:
      distributed.all_reduce_sum = saved
    return out, calls[0]

  fast, fast_calls = collect(True)
  slow, slow_calls = collect(False)
  assert len(fast) == len(slow) == epochs * (nmb + 1)
  assert fast_calls == 1 and slow_calls == len(slow), (fast_calls, slow_calls)
  for (a, ia), (b, ib) in zip(fast, slow):
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(a, b)
  # against the oracle on the concatenation of both ranks' minibatches
  for k, (a, idx) in enumerate(fast):
    mine = torch.zeros(world, 400, dtype=torch.float32)
    mine[rank, :idx.size] = torch.from_numpy(shard["advantages"][idx])
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = idx.size
    torch.distributed.all_reduce(mine)
    torch.distributed.all_reduce(sizes)
    whole = np.concatenate([mine[r, :sizes[r]].numpy() for r in range(world)])
    want = oracle.normalize_advantages(whole)
    off = int(sizes[:rank].sum())
    np.testing.assert_allclose(a, want[off:off + idx.size], rtol=2e-6, atol=2e-6, err_msg=f"minibatch {k}")
  print(f"rank {rank}: gpu_minibatch_stats OK", flush=True)


def gpu_learns():
  """Data-parallel PPO as a learner: every rank owns half of the envs of the image bandit
  (tools/quadrant_learns.py); with the gradient all-reduce (overlapped with the backward) and the
  per-rollout statistics all-reduce the replicas stay bit-identical and the mean reward reaches
  > 0.9 as in the single-process run."""
  from tools.quadrant_learns import run
  import derl_amd as derl
  world, rank = distributed.world_size(), distributed.rank()
  kept = {}
  original = derl.PPOFactory.make

  def make(self, env, *args, **kwargs):  # keep a handle on the algorithm the tool builds
    alg = original(self, env, *args, **kwargs)
    distributed.broadcast_(alg.model.engine.params)
    alg.model.engine.mark_dirty()
    kept["alg"] = alg
    return alg

  derl.PPOFactory.make = make
  try:
    curve, _ = run(iterations=40, nenvs=64 // world, horizon=16, seed=rank, lr=1e-3)
  finally:
    derl.PPOFactory.make = original
  reward = torch.tensor([float(np.mean(curve[-5:]))], dtype=torch.float64)
  torch.distributed.all_reduce(reward)
  assert reward.item() / world > 0.9, (curve[-5:], reward.item() / world)
  params = kept["alg"].model.engine.params.detach().double().cpu()
  low, high = params.clone(), params.clone()
  torch.distributed.all_reduce(low, op=torch.distributed.ReduceOp.MIN)
  torch.distributed.all_reduce(high, op=torch.distributed.ReduceOp.MAX)
  assert torch.equal(low, high), "replicas diverged"
  print(f"rank {rank}: gpu_learns OK", flush=True)


def rccl_one_rank():
  """The RCCL branch of the data path, executed: backend nccl, ONE rank (RCCL accepts a one-rank
  communicator), collectives forced (DERL_AMD_FORCE_COLLECTIVES=1).  The library's communicator is
  bootstrapped through torch.distributed (dx_comm_unique_id -> broadcast -> dx_comm_init); two
  Trainer.steps then issue the parameter broadcast, the advantage-statistics all-reduce and the
  two overlapped gradient all-reduces per step through dx_comm_* -- and must equal the
  single-process step bit for bit (a sum over one rank is the identity)."""
  import derl_amd as derl
  from derl_amd.optim import Adam
  assert torch.distributed.get_backend() == "nccl" and distributed.world_size() == 1
  assert distributed.native_comm() and distributed.sharded()
  assert distributed.comm_info()[:2] == (0, 1)
  derl.summary.stop_recording()
  A, B = 4, 32
  weights = gi.nature_cnn_weights(A, 3)
  mb = gi.cnn_minibatch(B, A, 5)
  logits, vals = oracle.nature_cnn_forward(weights, mb["observations"])
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, mb["actions"])
  full = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=(lp.numpy() + mb["logp_noise"]).astype(np.float32),
              advantages=mb["advantages"], values=(vals.numpy() + mb["value_noise"]).astype(np.float32),
              value_targets=(vals.numpy() + mb["target_noise"]).astype(np.float32))

  class Runner:
    step_count = 1000

  def run(with_collectives):
    saved = distributed.forced
    distributed.forced = (lambda: True) if with_collectives else (lambda: False)
    try:
      model = derl.NatureCNNModel([A, 1], max_batch=32)
      model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
      runner = Runner()
      runner.policy = derl.ActorCriticPolicy(model)
      lr = derl.LinearAnneal(2.5e-4, 1e6, name="lr")
      trainer = derl.Trainer(Adam(model, lr=lr.get_tensor(), eps=1e-5), anneals=[lr], max_grad_norm=0.5)
      alg = derl.PPO(runner, trainer, cliprange=0.1, value_loss_coef=0.25, entropy_coef=0.01)
      data = dict(full)
      derl.NormalizeAdvantages()(data)
      losses = [alg.step(data).item() for _ in range(2)]
      torch.cuda.synchronize()
      return losses, model.engine.params.clone()
    finally:
      distributed.forced = saved

  before = distributed.comm_info()
  losses_c, params_c = run(True)
  after = distributed.comm_info()
  # per run: 1 statistics all-reduce (3 doubles) + 2 steps x 2 halves of the gradient buffer
  assert after[2] - before[2] == 5, (before, after)
  assert after[3] - before[3] == 24 + 2 * 4 * params_c.numel(), (before, after)
  losses_s, params_s = run(False)
  assert distributed.comm_info()[2] == after[2], "the single-process run issued a collective"
  assert losses_c == losses_s and torch.equal(params_c, params_s)
  # the in-stream float32 sum and the broadcast
  ones = torch.ones(1000, device="cuda")
  distributed.all_reduce_sum(ones)
  distributed.broadcast_(ones)
  assert float(ones.sum().item()) == 1000.0
  print("rank 0: rccl_one_rank OK", flush=True)


def rccl_ordering():
  """Stream ordering of the two gradient all-reduces inside the native update, observable at ONE
  rank.  Diag flavour of the library with DX_COMM_TEST_HOOK="<delay_us>:2": behind every gradient
  all-reduce the communicator's stream idles for delay_us and then doubles the reduced piece
  (csrc/comm.hip).  The update then equals a single-process step on 2 x the gradient ONLY IF each
  reduction starts after the backward has written its piece (else the backward overwrites the
  doubled values) and the norm / optimizer step waits for both (else it reads the buffer before
  the delayed doubling) -- orderings that a plain one-rank run (identity all-reduce) cannot see."""
  import derl_amd as derl
  from derl_amd import _lib
  from derl_amd.optim import Adam
  assert _lib.LIB_PATH.endswith("libderl_amd_diag.so") and os.environ.get("DX_COMM_TEST_HOOK", "").endswith(":2")
  assert torch.distributed.get_backend() == "nccl" and distributed.native_comm() and distributed.sharded()
  derl.summary.stop_recording()
  A, B = 4, 32
  weights = gi.nature_cnn_weights(A, 3)
  mb = gi.cnn_minibatch(B, A, 5)
  logits, vals = oracle.nature_cnn_forward(weights, mb["observations"])
  lp, _, _ = oracle.categorical_log_prob_entropy(logits, mb["actions"])
  full = dict(observations=mb["observations"], actions=mb["actions"],
              log_prob=(lp.numpy() + mb["logp_noise"]).astype(np.float32),
              advantages=mb["advantages"], values=(vals.numpy() + mb["value_noise"]).astype(np.float32),
              value_targets=(vals.numpy() + mb["target_noise"]).astype(np.float32))

  class Runner:
    step_count = 1000

  def run(collectives, grad_scale):
    saved = distributed.forced
    distributed.forced = (lambda: True) if collectives else (lambda: False)
    try:
      model = derl.NatureCNNModel([A, 1], max_batch=32)
      model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
      runner = Runner()
      runner.policy = derl.ActorCriticPolicy(model)
      lr = derl.LinearAnneal(2.5e-4, 1e6, name="lr")
      optimizer = Adam(model, lr=lr.get_tensor(), eps=1e-5)
      trainer = derl.Trainer(optimizer, anneals=[lr], max_grad_norm=0.5)
      if not collectives:
        trainer.native_epochs = False  # update by update from Python: the gradient can be scaled in between
        plain = optimizer.reduce_and_norm

        def scaled_then_norm():
          model.engine.grads.mul_(grad_scale)
          return plain()
        optimizer.reduce_and_norm = scaled_then_norm
      alg = derl.PPO(runner, trainer, cliprange=0.1, value_loss_coef=0.25, entropy_coef=0.01)
      data = dict(full)
      derl.NormalizeAdvantages()(data)
      losses = [float(alg.step(data).item()) for _ in range(2)]
      torch.cuda.synchronize()
      return losses, model.engine.params.clone()
    finally:
      distributed.forced = saved

  before = distributed.comm_info()[2]
  losses_hook, params_hook = run(True, None)
  assert distributed.comm_info()[2] - before == 5  # 1 statistics + 2 updates x 2 pieces, all through the library
  losses_twice, params_twice = run(False, 2.0)
  losses_once, params_once = run(False, 1.0)
  assert not torch.equal(params_twice, params_once), "the doubled gradient changes nothing: vacuous check"
  assert losses_hook == losses_twice, (losses_hook, losses_twice)
  assert torch.equal(params_hook, params_twice), float((params_hook - params_twice).abs().max())
  print("rank 0: rccl_ordering OK", flush=True)


if __name__ == "__main__":
  mode = sys.argv[1]
  distributed.init_from_env(backend="nccl" if mode.startswith("rccl") else "gloo")
  {"cpu_math": cpu_math, "bootstrap_agreement": bootstrap_agreement, "bootstrap_real_failure": bootstrap_real_failure, "gpu_step": gpu_step, "gpu_minibatch_stats": gpu_minibatch_stats,
   "gpu_learns": gpu_learns, "rccl_one_rank": rccl_one_rank,
   "rccl_ordering": rccl_ordering}[mode]()
  distributed.destroy()
