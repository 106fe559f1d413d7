"""Models with derl's constructors and ``state_dict`` names (derl/models.py), backed by
device engines instead of torch.nn compute.

The modules are torch ``nn.Module`` containers only for their parameter bookkeeping:
every ``Parameter`` is a VIEW into the engine's flat fp32 parameter buffer (reference
state_dict order and layout) and every ``.grad`` a view into the flat gradient buffer, so
``parameters()``, ``state_dict()`` / ``load_state_dict()`` and torch optimizers keep
working while the arithmetic runs in the HIP kernels behind the C-ABI.
"""
import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .cnn_engine import CnnEngine, PARAM_NAMES
from .env.spaces import is_box, is_discrete


def conv2d_output_shape(height, width, kernel_size, stride):
  """derl/models.py:59-69 for padding 0, dilation 1."""
  return (height - kernel_size) // stride + 1, (width - kernel_size) // stride + 1


def orthogonal_init(layer):
  """Orthogonal weights, zero biases (derl/models.py:135-138)."""
  if hasattr(layer, "weight"):
    nn.init.orthogonal_(layer.weight)
  if hasattr(layer, "bias"):
    nn.init.zeros_(layer.bias)


class _HeadFunction(torch.autograd.Function):
  """Differentiable (logits, values) = model(observations): forward and backward both run
  in the engine.  Gradients are written into the flat gradient buffer the parameters'
  ``.grad`` views alias (overwritten, not accumulated: derl zero_grads before every
  backward, alg/common.py:69-70)."""

  @staticmethod
  def forward(ctx, anchor, model, observations, sample_idx):
    head = model.engine.forward(observations, sample_idx)
    ctx.model, ctx.observations, ctx.sample_idx = model, observations, sample_idx
    A = model.engine.num_actions
    return head[:, :A].clone(), head[:, A:A + 1].clone()

  @staticmethod
  def backward(ctx, dlogits, dvalues):
    eng = ctx.model.engine
    eng._ensure_backward()
    B, A = dlogits.shape
    dhead = eng.dhead[:B * 32].view(B, 32)
    dhead.zero_()
    dhead[:, :A] = dlogits
    dhead[:, A:A + 1] = dvalues
    eng.backward(ctx.observations, ctx.sample_idx)
    return None, None, None, None


class NatureCNNBase(nn.Sequential):
  """Parameter container with the reference's module names (derl/models.py:94-115)."""
  def __init__(self, input_shape=(84, 84, 4)):
    super().__init__()
    height, width, in_channels = input_shape
    convs = [nn.Conv2d(in_channels, 32, 8, 4), nn.Conv2d(32, 64, 4, 2), nn.Conv2d(64, 64, 3, 1)]
    for i, (conv, k, s) in enumerate(zip(convs, (8, 4, 3), (4, 2, 1))):
      height, width = conv2d_output_shape(height, width, k, s)
      self.add_module(f"conv-{i}", conv)
      self.add_module(f"relu-{i}", nn.ReLU())
    self.add_module("flatten", nn.Flatten())
    self.add_module("linear", nn.Linear(height * width * 64, 512))


class NatureCNNModel(nn.Module):
  """Nature-DQN actor-critic model (derl/models.py:166-214) for output_units=[A, 1].

  ``model(observations)`` returns ``[logits (B, A), values (B, 1)]`` as device tensors
  (differentiable through the engine); unbatched ``(84, 84, 4)`` input gives ``(A,)`` and
  ``(1,)`` like the reference's ``broadcast_inputs`` (models.py:141-163).  Observations may
  be uint8 or float32, NumPy or torch, host or device.
  """
  def __init__(self, output_units, input_shape=(84, 84, 4), init_fn=orthogonal_init,
               max_batch=256, device="cuda"):
    super().__init__()
    if (not isinstance(output_units, (list, tuple)) or len(output_units) != 2
        or output_units[1] != 1):
      raise NotImplementedError(
          "the MI355X engine implements the actor-critic head layout output_units=[A, 1] "
          f"(policy logits + value); got {output_units}")
    self.output_units = list(output_units)
    self.input_shape = tuple(input_shape)
    # same construction order / RNG consumption as the reference, so seeded models coincide
    self.base = NatureCNNBase(self.input_shape)
    self.output_layers = nn.ModuleList([nn.Linear(512, n) for n in self.output_units])
    self.init_fn = init_fn
    if self.init_fn:
      self.apply(self.init_fn)
    self.engine = CnnEngine(self.output_units[0], self.input_shape, max_batch, device)
    self._adopt_engine_storage()
    self._anchor = torch.zeros((), device=self.engine.device, requires_grad=True)

  def _layers(self):
    base = dict(self.base.named_children())
    return [base["conv-0"], base["conv-1"], base["conv-2"], base["linear"],
            self.output_layers[0], self.output_layers[1]]

  def _adopt_engine_storage(self):
    """Moves the initial values into the flat buffer and re-points every Parameter (and
    its .grad) at its view."""
    pviews = self.engine.named_views(self.engine.params)
    gviews = self.engine.named_views(self.engine.grads)
    with torch.no_grad():
      for name, layer in zip(PARAM_NAMES, self._layers()):
        for kind in ("weight", "bias"):
          param = getattr(layer, kind)
          view = pviews[f"{name}.{kind}"]
          view.copy_(param.detach())
          param.data = view
          param.grad = gviews[f"{name}.{kind}"]
    self.engine.watch(list(self.parameters()))
    self.engine.mark_dirty()

  def reserve(self, max_batch):
    """Grows the activation workspaces (parameters and optimizer state are kept)."""
    self.engine.reserve(max_batch)

  def state_dict(self, *args, **kwargs):
    from .policies import refuse_mid_epoch  # pylint: disable=import-outside-toplevel
    refuse_mid_epoch(self, "model.state_dict")
    return super().state_dict(*args, **kwargs)

  def load_state_dict(self, state_dict, strict=True):
    result = super().load_state_dict(state_dict, strict)
    self.engine.mark_dirty()
    return result

  def to(self, *args, **kwargs):
    """The parameters live on the GPU; ``.to("cpu")`` (which the reference's tests call,
    SURVEY.md G8) is accepted and ignored, anything else is refused."""
    target = args[0] if args else kwargs.get("device")
    if target is not None and torch.device(target).type not in ("cuda", "cpu"):
      raise ValueError(f"cannot move an MI355X engine model to {target}")
    return self

  def prepare(self, observations):
    """NumPy / host input -> contiguous device tensor (uint8 or float32)."""
    if isinstance(observations, np.ndarray):
      observations = torch.from_numpy(np.ascontiguousarray(observations))
    if observations.dtype not in (torch.uint8, torch.float32):
      observations = observations.to(torch.float32)
    if not observations.is_cuda:
      observations = observations.to(self.engine.device, non_blocking=True)
    return observations.contiguous()

  def forward(self, observations, sample_idx=None):
    observations = self.prepare(observations)
    squeeze = observations.ndim == 3
    if squeeze:
      observations = observations[None]
    batch = sample_idx.numel() if sample_idx is not None else observations.shape[0]
    self.reserve(batch)
    logits, values = _HeadFunction.apply(self._anchor, self, observations, sample_idx)
    if squeeze:
      return [logits[0], values[0]]
    return [logits, values]

  def head(self, observations, sample_idx=None):
    """Non-differentiable fast path: the padded (B, 32) head output (a view of the
    engine's buffer, valid until the next forward)."""
    batch = sample_idx.numel() if sample_idx is not None else observations.shape[0]
    self.reserve(batch)
    return self.engine.forward(observations, sample_idx)


def _cnn_policy_act(model, policy, inputs, training):
  """ActorCriticPolicy.act for the categorical CNN model (derl/policies.py:51-80)."""
  from .policies import DeviceCategorical, numpy_like_input  # pylint: disable=import-outside-toplevel
  A = model.engine.num_actions
  if training:
    observations = inputs["observations"]
    sample_idx = None
    if isinstance(observations, GatheredRows):
      observations, sample_idx = observations.base, observations.index
    observations = model.prepare(observations)
    head = model.head(observations, sample_idx)
    return {"distribution": DeviceCategorical(head, A), "values": head[:, A:A + 1]}
  to_numpy = numpy_like_input(inputs)
  observations = model.prepare(inputs)
  squeeze = observations.ndim == 3
  if squeeze:
    observations = observations[None]
  batch = observations.shape[0]
  model.reserve(batch)
  dev = model.engine.device
  actions = torch.empty(batch, dtype=torch.int64, device=dev)
  log_prob = torch.empty(batch, dtype=torch.float32, device=dev)
  values = torch.empty(batch, dtype=torch.float32, device=dev)
  model.engine.act(observations, actions, log_prob, values, None, policy.seed, policy.act_counter)
  policy.act_counter += 1
  values = values[:, None]
  if squeeze:
    actions, log_prob, values = actions[0], log_prob[0], values[0]
  if to_numpy:
    return {"actions": actions.cpu().numpy(), "log_prob": log_prob.cpu().numpy(),
            "values": values.cpu().numpy()}
  return {"actions": actions, "log_prob": log_prob, "values": values}


def _cnn_policy_act_into(model, policy, observations, actions_out, log_prob_out, values_out):
  model.reserve(observations.shape[0])
  model.engine.act(observations, actions_out, log_prob_out, values_out.view(-1), None, policy.seed,
                   policy.act_counter)
  policy.act_counter += 1


def _cnn_loss_forward_backward(model, policy, data, mode, cliprange, value_loss_coef,
                               entropy_coef, global_batch, actions, old_log_prob, advantages,
                               old_values, value_targets):
  """Minibatch forward + fused PPO/A2C loss (derl/alg/ppo.py:100-108).  Returns the
  float32[8] loss terms on the device and the closure that runs the model backward
  (derl/alg/common.py:70) from the head gradient the loss kernel wrote.  A ``gradient=``
  argument to ``loss.backward`` other than 1 is not applied (derl never passes one)."""
  del policy
  eng = model.engine
  observations, sample_idx = data["observations"], None
  if isinstance(observations, GatheredRows):
    observations, sample_idx = observations.base, observations.index
  observations = model.prepare(observations)
  if actions.dtype != torch.int64:
    actions = actions.long()
  fused = eng.fused_heads()
  if fused:
    # conv stack + linear layer, then heads + loss + the heads' backward in ONE launch
    # (dx_cnn_heads_loss_f32): the same kernels the native epoch enqueues
    batch = sample_idx.numel() if sample_idx is not None else observations.shape[0]
    model.reserve(batch)
    eng._ensure_backward()
    eng.forward_trunk(observations, sample_idx)
  else:
    head = model.head(observations, sample_idx)
    eng._ensure_backward()
    batch = head.shape[0]
    dhead = eng.dhead[:batch * 32].view(batch, 32)
  need = 8 * ((batch + 7) // 8)
  if model._loss_partials is None or model._loss_partials.numel() < need:
    model._loss_partials = torch.empty(need, dtype=torch.float64, device=eng.device)
  if fused:
    terms = torch.empty(8, dtype=torch.float32, device=eng.device)
    eng.heads_loss(batch, actions, old_log_prob, advantages, old_values, value_targets, mode, cliprange,
                   value_loss_coef, entropy_coef, global_batch, model._loss_partials, terms)
  else:
    terms = ops.categorical_loss(head, actions, old_log_prob, advantages, old_values, value_targets,
                                 eng.num_actions, mode, cliprange, value_loss_coef, entropy_coef,
                                 dhead, global_batch, model._loss_partials)

  def backward_fn(grad_output, on_part=None):
    """``on_part(k)`` is called when half k of the gradient buffer is final (see
    _FlatOptimizer.reduce_part): the tail's all-reduce overlaps the conv layers' backward."""
    del grad_output
    if on_part is None:
      eng.backward(observations, sample_idx, part=3 if fused else None)
      return
    eng.backward(observations, sample_idx, part=2 if fused else 0)
    on_part(0)
    eng.backward(observations, sample_idx, part=1)
    on_part(1)

  return terms, backward_fn


def _cnn_policy_rollout_into(model, policy, env, buffers, horizon):
  """All `horizon` steps of the synthetic device env in one native call."""
  from .env.synthetic import SyntheticAtariEnv  # pylint: disable=import-outside-toplevel
  from .env.summarize import DeviceSummarize  # pylint: disable=import-outside-toplevel
  if isinstance(env, DeviceSummarize):  # statistics are taken from the buffers afterwards
    env = env.env
  if not isinstance(env, SyntheticAtariEnv) or buffers["obs"].dtype != torch.uint8:
    return False
  nenvs = env.nenvs
  model.reserve(nenvs)
  model.engine.rollout_synth(buffers, horizon, nenvs, policy.seed, policy.act_counter, env.seed,
                             env.counter, 0.1, env.p_reset)
  policy.act_counter += horizon
  env.counter += horizon
  return True


NatureCNNModel.policy_act = _cnn_policy_act
NatureCNNModel.policy_act_into = _cnn_policy_act_into
NatureCNNModel.policy_rollout_into = _cnn_policy_rollout_into
NatureCNNModel.loss_forward_backward = _cnn_loss_forward_backward
NatureCNNModel._loss_partials = None


class GatheredRows:
  """Lazy ``base[index]`` along dim 0: how a minibatch refers to its frames without copying
  them (the conv loader gathers by index; derl/runners/onpolicy.py:59-62 copies)."""
  def __init__(self, base, index):
    self.base = base
    self.index = index  # int32 device tensor

  @property
  def shape(self):
    return (self.index.numel(),) + tuple(self.base.shape[1:])

  def materialize(self):
    return ops.gather_rows(self.base, self.index)


def vector_size(shape):
  if len(shape) != 1:
    raise ValueError(f"expected vector shape, got shape={shape}")
  return shape[0]


def make_model(observation_space, action_space, other_outputs=None, **kwargs):
  """Default model for the given spaces (derl/models.py:281-298).  Beyond the reference:
  a vector observation with a Discrete action space gets the MLP categorical model
  (the reference would build a NatureCNN and fail)."""
  if isinstance(other_outputs, int) or other_outputs is None:
    other_outputs = [other_outputs] if other_outputs is not None else []
  spaces = getattr(action_space, "spaces", None)
  if spaces:
    action_space = spaces[0]
  if is_discrete(action_space):
    output_units = [action_space.n, *other_outputs]
    if len(observation_space.shape) == 3:
      return NatureCNNModel(input_shape=observation_space.shape, output_units=output_units,
                            **kwargs)
    from .mlp_models import MLPCategoricalModel  # pylint: disable=import-outside-toplevel
    return MLPCategoricalModel(vector_size(observation_space.shape), output_units, **kwargs)
  if is_box(action_space):
    from .mlp_models import MuJoCoModel  # pylint: disable=import-outside-toplevel
    observation_dim = vector_size(observation_space.shape)
    action_dim = vector_size(action_space.shape)
    return MuJoCoModel(observation_dim=observation_dim, output_units=[action_dim, *other_outputs],
                       **kwargs)
  raise ValueError(f"unsupported action space {action_space}")
