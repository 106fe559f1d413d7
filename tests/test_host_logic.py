"""CPU tests of host-side logic that needs no GPU: the API mirror's argument handling,
factory bookkeeping, LR schedule, parsers, the dequantisation rule and the fast-division
constants the kernels rely on."""
import numpy as np
import numpy.testing as nt
import pytest


def test_u8_dequant_rule_is_exact_for_every_byte():
  """igemm.hip dequant_u8: q = x*r, e = fma(-q, 255, x), q' = fma(e, r, q) must equal the
  IEEE float32 division x/255 for all 256 byte values (models.py:121: x.float() / 255)."""
  x = np.arange(256, dtype=np.float32)
  r = np.float32(1.0) / np.float32(255.0)
  q = x * r
  e = (x.astype(np.float64) - q.astype(np.float64) * 255.0).astype(np.float32)  # exact fma
  q2 = (q.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(np.float32)
  nt.assert_array_equal(q2, x / np.float32(255))


def test_fastdiv_constants_are_exact():
  """igemm.hpp make_fastdiv: q = umulhi(n, magic) >> shift is exact for n < 2^31."""
  rs = np.random.RandomState(0)
  for d in [2, 3, 7, 9, 10, 20, 49, 81, 100, 400, 1023, 1024, 1025, 6370, 65537]:
    L = int(np.ceil(np.log2(d)))
    magic = (1 << (31 + L)) // d + 1
    assert magic < (1 << 32)
    n = np.concatenate([rs.randint(0, 2 ** 31 - 1, size=20000, dtype=np.int64),
                        np.arange(0, 5 * d), np.array([2 ** 31 - 1, 2 ** 31 - d, 2 ** 31 - d - 1])])
    q = ((n * magic) >> 32) >> (L - 1)
    nt.assert_array_equal(q, n // d)


def test_linear_anneal_closed_form_and_errors():
  import derl_amd as derl
  with np.load("tests/golden/anneal.npz") as g:
    for tag, (start, nsteps) in dict(atari=(2.5e-4, 10e6), mujoco=(3e-4, 1e6)).items():
      lr = derl.LinearAnneal(start, nsteps, name="lr")
      for count, expected in zip(g[f"{tag}.counts"], g[f"{tag}.values"]):
        lr.step_to(int(count))
        nt.assert_equal(np.float32(lr.get_tensor().item()), expected)
      with pytest.raises(ValueError):
        lr.step_to(0)
  # the per-step path agrees with the closed form
  a, b = derl.LinearAnneal(1.0, 10), derl.LinearAnneal(1.0, 10)
  for _ in range(7):
    a.step()
  b.step_to(7)
  assert a.get_tensor().item() == b.get_tensor().item() and a.step_count == b.step_count == 7
  assert derl.LinearAnneal(1.0, 10, name="lr").name == "lr" and derl.LinearAnneal(1.0, 10).name == "linear_anneal"


def test_torch_sched_follows_a_torch_scheduler():
  """derl/anneal.py:46-62: the tensor tracks scheduler.get_last_lr() (one element per parameter
  group), the same tensor object throughout, step_to walks step by step and never backwards."""
  import torch
  import derl_amd as derl
  from derl import TorchSched  # the drop-in import path
  assert TorchSched is derl.TorchSched
  w = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(3))]
  opt = torch.optim.SGD([dict(params=[w[0]], lr=0.5), dict(params=[w[1]], lr=0.25)])
  sched = derl.TorchSched(torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5))
  assert sched.name == "torch_sched" and sched.step_count == 0
  tensor = sched.get_tensor()
  nt.assert_array_equal(tensor.numpy(), np.float32([0.5, 0.25]))
  opt.step()
  value = sched.step()
  assert sched.step_count == 1 and value is not tensor
  sched.step_to(4)
  assert sched.get_tensor() is tensor and sched.step_count == 4
  nt.assert_array_equal(tensor.numpy(), np.float32([0.125, 0.0625]))
  assert [g["lr"] for g in opt.param_groups] == [0.125, 0.0625]
  with pytest.raises(ValueError):
    sched.step_to(3)


def test_factory_kwargs_accounting():
  import derl_amd as derl
  kwargs = derl.PPOFactory.get_kwargs()
  assert kwargs["num_runner_steps"] == 128 and kwargs["cliprange"] == 0.1 and kwargs["nenvs"] == 8
  mj = derl.PPOFactory.get_kwargs("mujoco")
  assert mj["num_epochs"] == 10 and mj["num_minibatches"] == 32 and mj["nenvs"] is None
  a2c = derl.A2CFactory.get_kwargs()
  assert a2c["lambda_"] == 1.0 and a2c["normalize_gae"] is False and a2c["optimizer_alpha"] == 0.99
  kd = derl.KwargsDict(a=1, b=2)
  assert kd.get_arg("a") == 1 and kd.unused == {"b"}
  with pytest.raises(ValueError, match="never read"):
    with kd.override_context(c=3):
      pass
  kd = derl.KwargsDict(a=1, b=2)
  with kd.override_context(c=3):
    assert kd.get_arg("c") == 3
  assert "c" not in kd.kwargs
  kd.reset_unused()
  assert kd.unused == {"a", "b"}


def test_parsers_and_env_ids(tmp_path):
  import derl_amd as derl
  assert derl.env.is_atari_id("BreakoutNoFrameskip-v4") and not derl.env.is_atari_id("CartPole-v1")
  assert derl.env.is_mujoco_id("HalfCheetah-v3") and derl.env.is_mujoco_id("HalfCheetahBulletEnv-v0")
  args = derl.get_args(atari_defaults=derl.PPOFactory.get_parser_defaults("atari"),
                       mujoco_defaults=derl.PPOFactory.get_parser_defaults("mujoco"),
                       args=["--env-id", "BreakoutNoFrameskip-v4", "--logdir", str(tmp_path),
                             "--nenvs", "256", "--lr", "1e-3"])
  assert args.nenvs == 256 and args.lr == 1e-3 and args.num_epochs == 3
  assert (tmp_path / "args.txt").read_text().count("\n") >= 10
  args = derl.get_args(atari_defaults=derl.PPOFactory.get_parser_defaults("atari"),
                       mujoco_defaults=derl.PPOFactory.get_parser_defaults("mujoco"),
                       args=["--env-id", "CartPole-v1", "--logdir", str(tmp_path), "--defaults", "atari"])
  assert args.defaults == "atari" and args.num_minibatches == 4
  with pytest.raises(SystemExit):
    derl.get_args(atari_defaults={}, mujoco_defaults={},
                  args=["--env-id", "CartPole-v1", "--logdir", str(tmp_path)])
  with pytest.raises(ValueError):
    derl.env.make("NoSuchEnv-v0")


def test_cartpole_batch_contract():
  import derl_amd as derl
  env = derl.env.make("CartPole-v1", nenvs=8, seed=1)
  obs = env.reset()
  assert obs.shape == (8, 4) and obs.dtype == np.float32 and env.action_space.n == 2
  total_done = 0
  for t in range(300):
    obs, rew, done, infos = env.step(np.full(8, t % 2))
    assert obs.shape == (8, 4) and rew.shape == (8,) and done.dtype == bool and len(infos) == 8
    total_done += done.sum()
    assert np.all(np.abs(obs[:, 0]) <= 2.5)  # auto-reset keeps states in range
  assert total_done > 0


def test_env_runner_generic_contract_matches_reference():
  """The generic (list-building) runner path against the reference's recorded run
  (tests/golden/runner_contract.npz)."""
  import derl_amd as derl
  from generate_stub_env import CountingEnv, CountingPolicy
  env, policy = CountingEnv(4), CountingPolicy()
  runner = derl.EnvRunner(env, policy, horizon=6, nsteps=48)
  with np.load("tests/golden/runner_contract.npz") as g:
    n = 0
    for i, inter in enumerate(runner.run()):
      assert list(inter.keys()) == list(g[f"{i}.keys"])
      assert runner.step_count == int(g[f"{i}.step_count"])
      for key in ("observations", "actions", "log_prob", "values", "rewards", "resets", "next_observations"):
        nt.assert_array_equal(np.asarray(inter[key]), g[f"{i}.{key}"])
      nt.assert_array_equal(inter["state"]["latest_observations"], g[f"{i}.latest_observations"])
      n += 1
    assert n == int(g["niters"]) and len(runner) == int(g["len"])
  with pytest.raises(TypeError):
    derl.EnvRunner(env, policy, 6, 48, time_limit=10)
  wrapped = derl.TransformInteractions(runner)
  assert wrapped.horizon == 6 and wrapped.nenvs == 4
  with pytest.raises(AttributeError):
    wrapped.no_such_attribute  # pylint: disable=pointless-statement


def test_derl_import_name_is_an_alias_not_a_copy():
  """`import derl` (the reference's import name) binds the SAME module objects as derl_amd."""
  import derl
  import derl_amd
  import derl.env
  import derl.runners.onpolicy as aliased
  import derl_amd.runners.onpolicy as real
  from derl.alg.ppo import PPOLoss
  assert derl.PPOFactory is derl_amd.PPOFactory and PPOLoss is derl_amd.PPOLoss
  assert aliased is real and derl.env is derl_amd.env


def test_package_installs_with_launcher(tmp_path):
  """setup.py (reference: setup.py:11-12): `pip install .` offline into a scratch target gives the
  packages, the native library and an executable `derl` launcher."""
  import os
  import shutil
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  target = tmp_path / "site"
  try:
    out = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-build-isolation",
                          "--no-index", "--quiet", "--target", str(target), root],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  finally:
    for left in ("build", "derl_amd.egg-info"):
      shutil.rmtree(os.path.join(root, left), ignore_errors=True)
  assert (target / "derl_amd" / "libderl_amd.so").exists() and (target / "derl" / "__init__.py").exists()
  env = dict(os.environ, PYTHONPATH=str(target))
  usage = subprocess.run([sys.executable, str(target / "bin" / "derl"), "dqn"], capture_output=True,
                         text=True, cwd=str(tmp_path), env=env, timeout=300)
  assert usage.returncode == 2 and "a2c" in usage.stderr and "ppo" in usage.stderr
  where = subprocess.run([sys.executable, "-c", "import derl; print(derl.__file__)"], capture_output=True,
                         text=True, cwd=str(tmp_path), env=env, timeout=300)
  assert where.stdout.strip().startswith(str(target)), where.stdout + where.stderr


def test_native_permutation_composer_is_numpy_bit_for_bit():
  """dx_host_compose_permutations (csrc/host_permute.hip: MT19937 + NumPy's legacy shuffle restated
  in C so that the draws do not hold the GIL) against `order = order[np.random.permutation(n)]` per
  epoch (derl/runners/onpolicy.py:44-49) from the same generator state: the same orders AND the same
  generator state afterwards, for sizes around the block boundary of the generator and BASELINE's
  shapes."""
  import ctypes
  from derl_amd import _lib
  lib = _lib.load()

  def native(n, epochs):
    name, key, pos, has_gauss, cached = np.random.get_state()
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    position = ctypes.c_int(int(pos))
    out = np.empty((epochs, n), np.int32)
    assert lib.dx_host_compose_permutations(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(position), n,
                                            epochs, 1, out.ctypes.data_as(ctypes.c_void_p)) == 0
    np.random.set_state((name, key, position.value, has_gauss, cached))
    return out

  def reference(n, epochs):
    order, rows = np.arange(n), []
    for _ in range(epochs):
      order = order[np.random.permutation(n)]
      rows.append(order)
    return np.stack(rows)

  for seed, n, epochs in [(0, 1, 2), (1, 2, 3), (5, 1024, 3), (7, 1030, 2), (3, 623, 5), (4, 625, 5),
                          (11, 32768, 3), (13, 131072, 10)]:
    np.random.seed(seed)
    np.random.rand(seed % 7)  # start somewhere inside a block of 624 words
    want = reference(n, epochs)
    tail_want = np.random.randint(0, 1 << 30, 8)
    np.random.seed(seed)
    np.random.rand(seed % 7)
    got = native(n, epochs)
    tail_got = np.random.randint(0, 1 << 30, 8)
    nt.assert_array_equal(got, want, err_msg=f"seed {seed} n {n}")
    nt.assert_array_equal(tail_got, tail_want, err_msg="generator state after the draws")


class _ResidentRunner:
  """A device-resident runner as far as IterateWithMinibatches' prefetch rule is concerned
  (`_device_resident`, `is_exhausted`, `env.host_rng_free`), yielding host arrays."""
  class _Env:
    host_rng_free = True

  def __init__(self, rollouts, samples, between=None):
    self.env, self.unwrapped = self._Env(), self
    self.rollouts, self.samples, self.between = rollouts, samples, between
    self.produced = 0

  def _device_resident(self):
    return True

  def is_exhausted(self):
    return self.produced >= self.rollouts

  def run(self, obs=None):
    while self.produced < self.rollouts:
      if self.between is not None:
        self.between(self.produced)
      self.produced += 1
      base = 1000 * self.produced
      yield {"observations": np.arange(base, base + self.samples, dtype=np.int64)}


def test_permutations_drawn_ahead_keep_the_reference_stream_and_never_touch_a_moved_generator():
  """IterateWithMinibatches draws the NEXT rollout's permutations on a worker thread from a
  SNAPSHOT of np.random (derl/runners/onpolicy.py:44-62 draws them after the rollout): the result
  counts only if the global generator still is at that snapshot when the rollout arrives.  (a) the
  minibatches equal the reference's order for an undisturbed stream, and np.random ends where the
  reference's would; (b) a reseed (or any draw) between two rollouts drops the draw made ahead --
  the old worker can no longer install a stale state behind a new seed (the race that made a
  native-vs-per-update comparison differ once in ~40 runs)."""
  from derl_amd.runners.onpolicy import IterateWithMinibatches

  def reference(rollouts, samples, epochs, nmb, between=None):
    out = []
    for r in range(rollouts):
      if between is not None:
        between(r)
      obs = np.arange(1000 * (r + 1), 1000 * (r + 1) + samples)
      order = np.arange(samples)
      for _ in range(epochs):
        order = order[np.random.permutation(samples)]
        out.extend(obs[order[k:k + samples // nmb]] for k in range(0, samples, samples // nmb))
    return out

  def ours(rollouts, samples, epochs, nmb, between=None):
    it = IterateWithMinibatches(_ResidentRunner(rollouts, samples, between), num_epochs=epochs, num_minibatches=nmb)
    assert it._prefetch_allowed()
    return [np.asarray(mb["observations"]) for mb in it.run()]

  def reseed(r):
    if r == 2:
      np.random.seed(99)  # while the draw for rollout 2 made ahead is pending or done
    if r == 3:
      np.random.rand(5)   # any other consumer of the stream

  for between in (None, reseed):
    np.random.seed(21)
    want = reference(5, 96, 3, 4, between)
    tail_want = np.random.randint(0, 1 << 30, 4)
    np.random.seed(21)
    got = ours(5, 96, 3, 4, between)
    tail_got = np.random.randint(0, 1 << 30, 4)
    assert len(got) == len(want) == 5 * 3 * 4
    for a, b in zip(got, want):
      nt.assert_array_equal(a, b)
    nt.assert_array_equal(tail_got, tail_want)
