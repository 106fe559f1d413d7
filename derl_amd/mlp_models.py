"""MLP policies backed by the device MLP engine: ``MuJoCoModel`` with derl's constructor and
``state_dict`` names (derl/models.py:224-271) and ``MLPCategoricalModel`` (vector observation,
Discrete actions: BASELINE config 1 / SURVEY.md G7, which the reference's make_model cannot
build)."""
import numpy as np
import torch
from torch import nn

from . import _lib
from . import ops
from .mlp_engine import MlpEngine
from .models import GatheredRows, orthogonal_init
from .policies import DeviceCategorical, numpy_like_input


class MLP(nn.Sequential):
  """Parameter container with the reference's layer indices 0, 2, 4 (models.py:224-237)."""
  def __init__(self, in_features, out_features, hidden_features=(64, 64)):
    if tuple(hidden_features) != (64, 64):
      raise NotImplementedError("the device MLP engine implements hidden_features=(64, 64)")
    dims = (in_features, *hidden_features, out_features)
    layers = []
    for nin, nout in zip(dims[:-1], dims[1:]):
      layers += [nn.Linear(nin, nout), nn.Tanh()]
    layers.pop()
    super().__init__(*layers)


class DeviceNormal:
  """``act(training=True)["distribution"]`` of the Gaussian policy (API compatibility;
  torch elementwise ops, off the hot path)."""
  def __init__(self, head, logstd):
    self.mean = head[:, :logstd.numel()]
    self.stddev = torch.exp(logstd)[None].expand_as(self.mean)

  def log_prob(self, actions):
    actions = torch.as_tensor(actions, device=self.mean.device, dtype=torch.float32)
    var = self.stddev ** 2
    return (-((actions - self.mean) ** 2) / (2 * var) - self.stddev.log()
            - 0.9189385332046727).sum(-1)

  def entropy(self):
    return (0.5 + 0.9189385332046727 + self.stddev.log()).sum(-1)


class _MlpActorCritic(nn.Module):
  """Shared engine plumbing of the two MLP model kinds."""
  gaussian = False

  def _build(self, observation_dim, policy_out, init_fn, max_batch, device):
    self.observation_dim, self.policy_out = int(observation_dim), int(policy_out)
    self.module_list = nn.ModuleList([MLP(observation_dim, policy_out), MLP(observation_dim, 1)])
    self.init_fn = init_fn
    if self.init_fn is not None:
      self.apply(self.init_fn)
    self.logstd = nn.Parameter(torch.zeros(policy_out)) if self.gaussian else None
    self.engine = MlpEngine(observation_dim, policy_out, self.gaussian, max_batch, device)
    self._adopt_engine_storage()
    self._anchor = torch.zeros((), device=self.engine.device, requires_grad=True)
    self._loss_partials = None
    self._dlogstd = None

  def _logstd_data(self):
    """The logstd parameter's storage without autograd bookkeeping (a view of the flat buffer)."""
    return self.logstd.data

  def _adopt_engine_storage(self):
    pviews = self.engine.named_views(self.engine.params)
    gviews = self.engine.named_views(self.engine.grads)
    with torch.no_grad():
      for name, param in self.named_parameters():
        view = pviews[name]
        view.copy_(param.detach())
        param.data = view
        param.grad = gviews[name]
    self.engine.watch(list(self.parameters()))
    self.engine.mark_dirty()

  def state_dict(self, *args, **kwargs):
    from .policies import refuse_mid_epoch  # pylint: disable=import-outside-toplevel
    refuse_mid_epoch(self, "model.state_dict")
    return super().state_dict(*args, **kwargs)

  def load_state_dict(self, state_dict, strict=True):
    result = super().load_state_dict(state_dict, strict)
    self.engine.mark_dirty()
    return result

  def to(self, *args, **kwargs):
    target = args[0] if args else kwargs.get("device")
    if target is not None and not isinstance(target, torch.dtype):
      if torch.device(target).type not in ("cuda", "cpu"):
        raise ValueError(f"cannot move an MI355X engine model to {target}")
    return self

  def reserve(self, max_batch):
    self.engine.reserve(max_batch)

  def prepare(self, observations):
    """Any array -> contiguous float32 device tensor (collocate_inputs(), models.py:72-91:
    inputs are cast to the model's dtype and device)."""
    if isinstance(observations, GatheredRows):
      observations = observations.materialize()
    if isinstance(observations, np.ndarray):
      observations = torch.from_numpy(np.ascontiguousarray(observations))
    return observations.to(device=self.engine.device, dtype=torch.float32).contiguous()

  def head(self, observations):
    return self.engine.forward(observations)

  # ---- derl-facing forward ----------------------------------------------------------
  def forward(self, observations):
    observations = self.prepare(observations)
    squeeze = observations.ndim == 1
    if squeeze:
      observations = observations[None]
    head = self.head(observations)
    P = self.policy_out
    outs = [head[:, :P].clone()]
    if self.gaussian:
      outs.append(torch.exp(self.logstd.detach())[None].repeat_interleave(head.shape[0], 0))
    outs.append(head[:, P:P + 1].clone())
    if squeeze:
      outs = [o[0] for o in outs]
    return tuple(outs)

  # ---- policy hooks -----------------------------------------------------------------
  def _act_kernel(self, head, policy, out=None):
    raise NotImplementedError

  def policy_act(self, policy, inputs, training):
    P = self.policy_out
    if training:
      observations = self.prepare(inputs["observations"])
      head = self.head(observations)
      dist = (DeviceNormal(head, self.logstd.detach()) if self.gaussian
              else DeviceCategorical(head, P))
      return {"distribution": dist, "values": head[:, P:P + 1]}
    to_numpy = numpy_like_input(inputs)
    observations = self.prepare(inputs)
    squeeze = observations.ndim == 1
    if squeeze:
      observations = observations[None]
    head = self.head(observations)
    actions, log_prob, values = self._act_kernel(head, policy)
    policy.act_counter += 1
    values = values[:, None]
    if squeeze:
      actions, log_prob, values = actions[0], log_prob[0], values[0]
    if to_numpy:
      return {"actions": actions.cpu().numpy(), "log_prob": log_prob.cpu().numpy(),
              "values": values.cpu().numpy()}
    return {"actions": actions, "log_prob": log_prob, "values": values}

  def policy_act_into(self, policy, observations, actions_out, log_prob_out, values_out):
    head = self.head(self.prepare(observations))
    self._act_kernel(head, policy, out=(actions_out, log_prob_out, values_out.view(-1)))
    policy.act_counter += 1

  def policy_rollout_into(self, policy, env, buffers, horizon):
    """All `horizon` steps against the MuJoCo-shaped synthetic device env in one native launch (Gaussian policy,
    observations up to 64 wide); False = take the per-step loop."""
    from .env.synthetic import SyntheticMuJoCoEnv  # pylint: disable=import-outside-toplevel
    from .env.summarize import DeviceSummarize  # pylint: disable=import-outside-toplevel
    if isinstance(env, DeviceSummarize):  # statistics are taken from the buffers afterwards
      env = env.env
    obs = buffers["obs"]
    if (not self.gaussian or not isinstance(env, SyntheticMuJoCoEnv) or obs.dtype != torch.float32
        or obs.shape[-1] > 64 or buffers["actions"].dtype != torch.float32):
      return False
    if getattr(self, "_rollout_unsupported", False):
      return False
    # the kernel takes raw pointers: what it assumes about them is checked here
    nenvs = int(env.nenvs)
    if obs.ndim != 3 or obs.shape[0] < horizon + 1 or obs.shape[1] != nenvs or obs.shape[2] != self.engine.obs_dim:
      raise ValueError(f"rollout buffers: observations {tuple(obs.shape)} do not fit horizon {horizon}, "
                       f"{nenvs} envs x {self.engine.obs_dim} components")
    for key in ("obs", "actions", "log_prob", "values", "rewards", "resets"):
      t = buffers[key]
      if not t.is_cuda or not t.is_contiguous() or (key != "obs" and (t.shape[0] < horizon or t.shape[1] != nenvs)):
        raise ValueError(f"rollout buffers: '{key}' must be a contiguous device array of (>= {horizon}, {nenvs}, ...)")
    try:
      self.engine.rollout_synth(buffers, horizon, nenvs, policy.seed, policy.act_counter, env.seed, env.counter,
                                env.p_reset)
    except _lib.NativeError as error:
      # ONLY "this configuration has no one-launch rollout" (DX_ENOSUP: the layer-by-layer route, DX_MLP_UNFUSED=1)
      # selects the per-step loop; a bad argument or a failed launch is an error, not a route
      if error.status != _lib.DX_ENOSUP:
        raise
      self._rollout_unsupported = True
      return False
    policy.act_counter += horizon
    env.counter += horizon
    return True

  def loss_forward_backward(self, policy, data, mode, cliprange, value_loss_coef, entropy_coef,
                            global_batch, actions, old_log_prob, advantages, old_values,
                            value_targets):
    del policy
    eng = self.engine
    head = self.head(self.prepare(data["observations"]))
    batch = head.shape[0]
    dhead = eng.dhead[:batch * 32].view(batch, 32)
    if self.gaussian:
      need = 40 * ((batch + 255) // 256)
      if self._loss_partials is None or self._loss_partials.numel() < need:
        self._loss_partials = torch.empty(need, dtype=torch.float64, device=eng.device)
      if self._dlogstd is None or self._dlogstd.data_ptr() != eng.grads.data_ptr() + 4 * eng.ctx.off_logstd:
        self._dlogstd = eng.named_views(eng.grads)["logstd"]  # cached view of the gradient slice
      dlogstd = self._dlogstd
      actions_f32 = actions if (actions.dtype == torch.float32 and actions.is_contiguous()) \
          else actions.to(torch.float32).contiguous()
      terms = ops.normal_loss(head, self._logstd_data(), actions_f32,
                              old_log_prob, advantages, old_values, value_targets, mode, cliprange,
                              value_loss_coef, entropy_coef, dhead, dlogstd, global_batch,
                              self._loss_partials)
    else:
      need = 8 * ((batch + 7) // 8)
      if self._loss_partials is None or self._loss_partials.numel() < need:
        self._loss_partials = torch.empty(need, dtype=torch.float64, device=eng.device)
      terms = ops.categorical_loss(head, actions.long().contiguous(), old_log_prob, advantages,
                                   old_values, value_targets, self.policy_out, mode, cliprange,
                                   value_loss_coef, entropy_coef, dhead, global_batch,
                                   self._loss_partials)

    def backward_fn(grad_output):
      del grad_output
      eng.backward(batch)

    return terms, backward_fn


class MuJoCoModel(_MlpActorCritic):
  """MuJoCo model (derl/models.py:240-271) for output_units=[action_dim, 1]: returns
  ``(mean, std, values)``; parameters ``logstd``, ``module_list.{0,1}.{0,2,4}.{weight,bias}``."""
  gaussian = True

  def __init__(self, observation_dim, output_units, init_fn=orthogonal_init, max_batch=256,
               device="cuda"):
    super().__init__()
    if (not isinstance(output_units, (list, tuple)) or len(output_units) != 2
        or output_units[1] != 1):
      raise NotImplementedError(
          "the MI355X engine implements output_units=[action_dim, 1]; got " f"{output_units}")
    self._build(observation_dim, output_units[0], init_fn, max_batch, device)

  def _act_kernel(self, head, policy, out=None):
    return ops.normal_act(head, self.logstd.detach(), None, policy.seed, policy.act_counter, out=out)


class MLPCategoricalModel(_MlpActorCritic):
  """Vector observations with Discrete actions: policy logits and value from two tanh MLPs
  (the "plain MLP otherwise" of BASELINE.json; not in the reference)."""
  gaussian = False

  def __init__(self, observation_dim, output_units, init_fn=orthogonal_init, max_batch=256,
               device="cuda"):
    super().__init__()
    if (not isinstance(output_units, (list, tuple)) or len(output_units) != 2
        or output_units[1] != 1):
      raise NotImplementedError("MLPCategoricalModel implements output_units=[A, 1]")
    self._build(observation_dim, output_units[0], init_fn, max_batch, device)

  def _act_kernel(self, head, policy, out=None):
    return ops.categorical_act(head, self.policy_out, None, policy.seed, policy.act_counter, out=out)
