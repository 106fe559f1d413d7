"""Generates the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (``python tests/golden/generate.py``): it imports
mknbv/derl from /root/reference through ``_ref_import`` (stand-ins for gym / atari_py /
cv2 / tensorboard) and records what the reference computes on seeded synthetic inputs
(``inputs.py``).  Only outputs (and inputs that are derived from reference outputs) are
stored.  The reference never travels; these .npz files do.

Also copies the reference's own small test fixtures that still pin the path
 -- they are data, not source.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402
from _ref_import import import_reference, REFERENCE_ROOT  # noqa: E402

derl = import_reference()
torch.set_num_threads(8)


class FixedValuePolicy:
  """Stands in for the policy GAE bootstraps from (trajectory_transforms.py:47-50)."""
  def __init__(self, last_values):
    self.last_values = last_values

  def act(self, inputs, state=None, update_state=True, training=False):
    return {"values": self.last_values}


def gen_gae():
  out = {}
  for name in gi.GAE_CASES:
    d = gi.gae_inputs(name)
    traj = dict(rewards=d["rewards"], resets=d["resets"], values=d["values"],
                state=dict(latest_observations=None))
    gae = derl.GAE(FixedValuePolicy(d["last_values"]), gamma=d["gamma"],
                   lambda_=d["lambda_"], normalize=False)
    adv, vt = gae(traj)
    if d["rewards"].ndim == 2:
      derl.MergeTimeBatch()(traj)
    out[f"{name}.advantages"] = traj["advantages"]
    out[f"{name}.value_targets"] = traj["value_targets"]
  # whole-batch normalisation branch (trajectory_transforms.py:67-68), default normalize=None
  d = gi.gae_inputs("ragged")
  traj = dict(rewards=d["rewards"], resets=d["resets"], values=d["values"],
              state=dict(latest_observations=None))
  derl.GAE(FixedValuePolicy(d["last_values"]), gamma=d["gamma"], lambda_=d["lambda_"])(traj)
  out["ragged.normalized_advantages"] = traj["advantages"]
  np.savez_compressed(os.path.join(HERE, "gae.npz"), **out)
  print("gae.npz", {k: v.shape for k, v in out.items()})


def load_cnn(num_actions, seed):
  model = derl.NatureCNNModel([num_actions, 1])
  model.to("cpu")
  weights = gi.nature_cnn_weights(num_actions, seed)
  model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
  return model


def load_mlp(obs_dim, act_dim, seed):
  model = derl.MuJoCoModel(obs_dim, [act_dim, 1])
  model.to("cpu")
  weights = gi.mujoco_weights(obs_dim, act_dim, seed)
  model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
  return model


def gen_act():
  out = {}
  for num_actions, seed in ((4, 21), (6, 22)):
    model = load_cnn(num_actions, seed)
    policy = derl.ActorCriticPolicy(model)
    obs = gi.frames(32, seed + 100)
    actions = np.random.RandomState(seed).randint(0, num_actions, size=32)
    act = policy.act(dict(observations=obs), training=True)
    dist = act["distribution"]
    tag = f"cnn_a{num_actions}"
    out[f"{tag}.logits"] = dist.logits.detach().numpy()  # normalised logits = logp
    raw_logits, values = model(torch.from_numpy(obs))
    out[f"{tag}.raw_logits"] = raw_logits.detach().numpy()
    out[f"{tag}.values"] = act["values"].detach().numpy()
    out[f"{tag}.log_prob"] = dist.log_prob(torch.from_numpy(actions)).detach().numpy()
    out[f"{tag}.entropy"] = dist.entropy().detach().numpy()
    out[f"{tag}.hidden"] = model.base(torch.from_numpy(obs)).detach().numpy()
    # unbatched broadcast path (models.py:141-163): (84,84,4) -> (A,), (1,)
    one_logits, one_value = model(torch.from_numpy(obs[0]))
    out[f"{tag}.unbatched_logits"] = one_logits.detach().numpy()
    out[f"{tag}.unbatched_value"] = one_value.detach().numpy()
  model = load_mlp(17, 6, 23)
  policy = derl.ActorCriticPolicy(model)
  mb = gi.mlp_minibatch(64, 17, 6, 123)
  act = policy.act(dict(observations=mb["observations"]), training=True)
  dist = act["distribution"]
  out["mlp.mean"] = dist.mean.detach().numpy()
  out["mlp.std"] = dist.stddev.detach().numpy()
  out["mlp.values"] = act["values"].detach().numpy()
  out["mlp.log_prob"] = dist.log_prob(torch.from_numpy(mb["actions"])).detach().numpy()
  out["mlp.entropy"] = dist.entropy().detach().numpy()
  # rollout-mode contract (policies.py:76-80): key order and dtypes; float64 obs accepted
  torch.manual_seed(0)
  roll = policy.act(mb["observations"].astype(np.float64))
  out["mlp.rollout_keys"] = np.array(list(roll.keys()))
  out["mlp.rollout_shapes"] = np.array([str((v.shape, str(v.dtype))) for v in roll.values()])
  np.savez_compressed(os.path.join(HERE, "act.npz"), **out)
  print("act.npz", {k: v.shape for k, v in out.items()})


class FakeRunner:
  def __init__(self, policy, step_count):
    self.policy = policy
    self.step_count = step_count


def relu_margin(model, observations):
  """Smallest |conv pre-activation| relative to its layer's largest one (float64 evaluation of
  the reference model's CURRENT weights): how far the closest ReLU unit is from changing side."""
  import torch.nn.functional as F
  weights = {k: v.detach().double() for k, v in model.state_dict().items()}
  x = (torch.from_numpy(observations).permute(0, 3, 1, 2).float() / 255).double().contiguous()
  margin = np.inf
  for i, stride in enumerate((4, 2, 1)):
    x = F.conv2d(x, weights[f"base.conv-{i}.weight"], weights[f"base.conv-{i}.bias"], stride=stride)
    margin = min(margin, float(x.abs().min() / x.abs().max()))
    x = F.relu(x)
  return margin


def gen_steps(only=None):
  for name, cfg in gi.STEP_CASES.items():
    if only is not None and name not in only:
      continue
    out = {}
    if cfg["kind"] == "cnn":
      model = load_cnn(cfg["num_actions"], cfg["seed"])
      mb = gi.cnn_minibatch(cfg["batch"], cfg["num_actions"], cfg["seed"] + 50)
      actions = torch.from_numpy(mb["actions"])
    else:
      model = load_mlp(cfg["obs_dim"], cfg["act_dim"], cfg["seed"])
      mb = gi.mlp_minibatch(cfg["batch"], cfg["obs_dim"], cfg["act_dim"], cfg["seed"] + 50)
      actions = torch.from_numpy(mb["actions"])
    policy = derl.ActorCriticPolicy(model)
    with torch.no_grad():
      act = policy.act(dict(observations=mb["observations"]), training=True)
      new_lp = act["distribution"].log_prob(actions).numpy()
      new_v = act["values"].numpy()
    # "old" rollout quantities placed around the current ones so that both clip
    # branches (ratio and value) are exercised
    data = dict(observations=mb["observations"], actions=mb["actions"],
                log_prob=(new_lp + mb["logp_noise"]).astype(np.float32),
                advantages=mb["advantages"].copy(),
                values=(new_v + mb["value_noise"]).astype(np.float32),
                value_targets=(new_v + mb["target_noise"]).astype(np.float32))
    out["data.log_prob"] = data["log_prob"]
    out["data.values"] = data["values"]
    out["data.value_targets"] = data["value_targets"]
    if cfg["alg"] == "ppo":
      derl.NormalizeAdvantages()(data)
      out["normalized_advantages"] = data["advantages"]
      lr = derl.LinearAnneal(cfg["lr"], cfg["num_train_steps"], name="lr")
      optimizer = torch.optim.Adam(model.parameters(), lr=lr.get_tensor(),
                                   eps=cfg["optimizer_epsilon"])
      trainer = derl.alg.common.Trainer(optimizer, anneals=[lr],
                                        max_grad_norm=cfg["max_grad_norm"])
      alg = derl.PPO(FakeRunner(policy, cfg["step_count"]), trainer,
                     cliprange=cfg["cliprange"], value_loss_coef=cfg["value_loss_coef"],
                     entropy_coef=cfg["entropy_coef"])
    else:
      lr = derl.LinearAnneal(cfg["lr"], cfg["num_train_steps"], 0., name="lr")
      optimizer = torch.optim.RMSprop(model.parameters(), lr.get_tensor(),
                                      alpha=cfg["optimizer_alpha"],
                                      eps=cfg["optimizer_epsilon"])
      trainer = derl.alg.common.Trainer(optimizer, anneals=[lr],
                                        max_grad_norm=cfg["max_grad_norm"])
      alg = derl.A2C(FakeRunner(policy, cfg["step_count"]), trainer,
                     value_loss_coef=cfg["value_loss_coef"],
                     entropy_coef=cfg["entropy_coef"])
    names = [k for k, _ in model.named_parameters()]
    out["param_names"] = np.array(names)
    # unclipped gradients of the first step (alg/test.py:35-52 style)
    loss0 = alg.loss(data)
    model.zero_grad()
    loss0.backward()
    out["loss0"] = np.float32(loss0.item())
    grads = [p.grad.detach().numpy().copy() for p in model.parameters()]
    out["grad_norm0"] = np.float64(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum())
                                               for g in grads)))
    for pname, g in zip(names, grads):
      for key, val in gi.summarize_tensor(g).items():
        out[f"grad0.{pname}.{key}"] = val
    model.zero_grad()
    alg.loss_fn.call_count = 0
    losses = []
    for step in range(cfg["nsteps"]):
      # the runner's env-step counter advances between rollouts only; emulate one
      # rollout boundary between step 1 and 2 so the LR changes once
      if step == 2:
        alg.runner.step_count += 4096
      if cfg["kind"] == "cnn":
        # how close the reference's OWN run comes to a ReLU boundary at this step: below ~3e-6 the
        # side a unit falls on depends on float32 summation order, and from that step on another
        # implementation may only be compared through a same-start oracle (tests/test_ppo_e2e_gpu.py)
        out[f"relu_margin.{step}"] = np.float64(relu_margin(model, mb["observations"]))
      if "min_relu_margin" in cfg:
        assert out[f"relu_margin.{step}"] >= cfg["min_relu_margin"], (name, step, out[f"relu_margin.{step}"])
      losses.append(alg.step(data).item())
      out[f"lr.{step}"] = np.float32(lr.get_tensor().item())
      for pname, p in model.named_parameters():
        for key, val in gi.summarize_tensor(p.detach().numpy()).items():
          out[f"param{step}.{pname}.{key}"] = val
    out["losses"] = np.asarray(losses, np.float32)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "losses", losses, "lr", [out[f"lr.{s}"] for s in range(cfg["nsteps"])],
          "grad_norm0", out["grad_norm0"])


def gen_minibatch_order():
  out = {}
  for tag, (n, epochs, nmb) in dict(even=(1024, 3, 4), remainder=(1030, 2, 4)).items():
    class R:
      def __init__(self):
        self.env = self.policy = None
        self.horizon = self.nsteps = self.step_count = 0
        self.nenvs = 1

      def is_exhausted(self):
        return False

      def run(self, obs=None):
        yield dict(observations=np.arange(n)[:, None], index=np.arange(n),
                   state=dict(latest_observations=None))
    np.random.seed(1234)
    it = derl.IterateWithMinibatches(R(), num_epochs=epochs, num_minibatches=nmb)
    for i, mb in enumerate(it.run()):
      out[f"{tag}.{i}"] = mb["index"]
  np.savez_compressed(os.path.join(HERE, "minibatch_order.npz"), **out)
  print("minibatch_order.npz", len(out))


class CountingEnv:
  """Deterministic batched env: obs[t] = t broadcast, reward = action parity, reset every 5."""
  def __init__(self, nenvs):
    self.nenvs = nenvs
    self.t = 0
    self.unwrapped = self

  def reset(self):
    self.t = 0
    return np.zeros((self.nenvs, 3), np.float32)

  def step(self, actions):
    self.t += 1
    obs = np.full((self.nenvs, 3), self.t, np.float32)
    rew = (np.asarray(actions) % 2).astype(np.float64)
    done = np.full(self.nenvs, self.t % 5 == 0)
    return obs, rew, done, [{} for _ in range(self.nenvs)]


class CountingPolicy:
  def __init__(self):
    self.calls = 0

  def is_recurrent(self):
    return False

  def act(self, inputs, state=None, update_state=True, training=False):
    self.calls += 1
    n = inputs.shape[0]
    return dict(actions=np.arange(n) + self.calls, log_prob=np.full(n, -0.5, np.float32),
                values=np.full((n, 1), float(self.calls), np.float32))


def gen_runner_contract():
  env, policy = CountingEnv(4), CountingPolicy()
  runner = derl.EnvRunner(env, policy, horizon=6, nsteps=48)
  out = {}
  for i, inter in enumerate(runner.run()):
    out[f"{i}.keys"] = np.array(list(inter.keys()))
    out[f"{i}.step_count"] = np.int64(runner.step_count)
    for key in ("observations", "actions", "log_prob", "values", "rewards", "resets",
                "next_observations"):
      out[f"{i}.{key}"] = np.asarray(inter[key])
    out[f"{i}.latest_observations"] = inter["state"]["latest_observations"]
  out["niters"] = np.int64(i + 1)
  out["len"] = np.int64(len(runner))
  np.savez_compressed(os.path.join(HERE, "runner_contract.npz"), **out)
  print("runner_contract.npz iters", i + 1)


def gen_anneal():
  out = {}
  for tag, (start, nsteps, counts) in dict(
      atari=(2.5e-4, 10e6, [0, 1, 1024, 32768, 65536]),
      mujoco=(3e-4, 1e6, [0, 2048, 131072, 999999, 1000000, 1000500])).items():
    lr = derl.LinearAnneal(start, nsteps, name="lr")
    vals = []
    for c in counts:
      lr.step_to(c)
      vals.append(lr.get_tensor().item())
    out[f"{tag}.counts"] = np.asarray(counts, np.int64)
    out[f"{tag}.values"] = np.asarray(vals, np.float32)
  np.savez_compressed(os.path.join(HERE, "anneal.npz"), **out)
  print("anneal.npz", {k: v for k, v in out.items()})


def copy_upstream_fixtures():
  """The reference's own data fixtures that still pin this path."""
  td = os.path.join(REFERENCE_ROOT, "testdata")
  dst = os.path.join(HERE, "upstream")
  os.makedirs(dst, exist_ok=True)
  np.save(os.path.join(dst, "dqn-base-outputs.npy"),
          np.load(os.path.join(td, "models/dqn-base-outputs.npy")))
  with np.load(os.path.join(td, "ppo/pybullet/interactions.npz"), allow_pickle=True) as d:
    keep = {k: d[k] for k in d.files if d[k].dtype != object}
    keep["latest_observations"] = d["state"].item()["latest_observations"]
  np.savez_compressed(os.path.join(dst, "ppo_pybullet_interactions.npz"), **keep)
  with np.load(os.path.join(td, "ppo/pybullet/grads.npz")) as d:
    np.savez_compressed(os.path.join(dst, "ppo_pybullet_grads.npz"), **{k: d[k] for k in d.files})
  np.save(os.path.join(dst, "ppo_pybullet_losses.npy"),
          np.load(os.path.join(td, "ppo/pybullet/losses.npy")))
  with np.load(os.path.join(td, "a2c/atari/interactions.npz"), allow_pickle=True) as d:
    keep = {k: d[k] for k in d.files if d[k].dtype != object and k != "next_observations"}
    keep["latest_observations"] = d["state"].item()["latest_observations"]
  np.savez_compressed(os.path.join(dst, "a2c_atari_interactions.npz"), **keep)
  np.save(os.path.join(dst, "a2c_atari_losses.npy"),
          np.load(os.path.join(td, "a2c/atari/losses.npy")))
  print("upstream fixtures copied:", sorted(os.listdir(dst)))


if __name__ == "__main__":
  if len(sys.argv) > 1:  # python generate.py <step case> ...: only those fixtures
    gen_steps(only=sys.argv[1:])
    sys.exit(0)
  gen_gae()
  gen_act()
  gen_steps()
  gen_minibatch_order()
  gen_runner_contract()
  gen_anneal()
  copy_upstream_fixtures()
