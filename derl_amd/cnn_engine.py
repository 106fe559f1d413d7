"""Device state of the Nature-DQN actor-critic: flat parameter / gradient buffers in the
reference's ``state_dict`` order, packed weight mirrors, activation and slab workspaces,
and the calls into the C-ABI (dx_cnn_*).  torch only allocates memory and hands out the
stream."""
import ctypes

import torch

from . import _lib, distributed

PARAM_NAMES = ("base.conv-0", "base.conv-1", "base.conv-2", "base.linear",
               "output_layers.0", "output_layers.1")


class CnnEngine:
  """Owns every device buffer of one NatureCNN([A, 1]) model (derl/models.py:166-214)."""

  def __init__(self, num_actions, input_shape=(84, 84, 4), max_batch=256, device="cuda"):
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise _lib.NativeError("CnnEngine needs a HIP device; derl_amd has no CPU path")
    ctx = _lib.CnnCtx()
    ctx.struct_bytes = ctypes.sizeof(_lib.CnnCtx)
    ctx.in_h, ctx.in_w, ctx.in_c = (int(v) for v in input_shape)
    ctx.num_actions = int(num_actions)
    ctx.max_batch = int(max_batch)
    _lib.call("dx_cnn_init", ctypes.byref(ctx))
    self.ctx = ctx
    self.num_actions = int(num_actions)
    self.input_shape = tuple(int(v) for v in input_shape)
    f32 = dict(dtype=torch.float32, device=self.device)
    self.params = torch.zeros(ctx.param_count, **f32)
    self.grads = torch.zeros(ctx.param_count, **f32)
    self.packed = torch.zeros(ctx.packed_count, **f32)
    self._forward_buffers(ctx, f32)
    self._backward_allocated = False
    ctx.params, ctx.grads, ctx.packed = (t.data_ptr() for t in (self.params, self.grads, self.packed))
    self._packed_version = None
    self._train_packed_version = None  # see ppo_epoch: only the training kernels' mirrors are current
    self._watched = [self.params]
    self.shapes = self._param_shapes()

  def _forward_buffers(self, ctx, f32):
    self.y0 = torch.empty(ctx.y0_count, **f32)
    self.y1 = torch.empty(ctx.y1_count, **f32)
    self.y2 = torch.empty(ctx.y2_count, **f32)
    self.hid = torch.empty(ctx.hid_count, **f32)
    self.head = torch.empty(ctx.head_count, **f32)
    self.hid_slabs = torch.empty(ctx.hid_slab_count, **f32)
    for name in ("y0", "y1", "y2", "hid", "head", "hid_slabs"):
      setattr(ctx, name, getattr(self, name).data_ptr())

  def _ensure_backward(self):
    if self._backward_allocated:
      return
    ctx = self.ctx
    f32 = dict(dtype=torch.float32, device=self.device)
    for name, src in (("dy0", "y0"), ("dy1", "y1"), ("dy2", "y2"), ("dhid", "hid"), ("dhead", "head")):
      buf = torch.empty_like(getattr(self, src))
      setattr(self, name, buf)
      setattr(ctx, name, buf.data_ptr())
    self.slabs = torch.empty(ctx.slab_count, **f32)
    ctx.slabs = self.slabs.data_ptr()
    self._backward_allocated = True

  def reserve(self, max_batch):
    """Grows the activation / slab workspaces to hold ``max_batch`` samples; parameters,
    gradients and packed mirrors (and anything aliasing them) are untouched."""
    if max_batch <= self.ctx.max_batch:
      return
    self.ctx.max_batch = int(max_batch)
    _lib.call("dx_cnn_init", ctypes.byref(self.ctx))
    f32 = dict(dtype=torch.float32, device=self.device)
    self._forward_buffers(self.ctx, f32)
    if self._backward_allocated:
      self._backward_allocated = False
      self._ensure_backward()

  def _param_shapes(self):
    c = self.ctx
    A = self.num_actions
    return {
        "base.conv-0": ((32, c.in_c, 8, 8), (32,)), "base.conv-1": ((64, 32, 4, 4), (64,)),
        "base.conv-2": ((64, 64, 3, 3), (64,)), "base.linear": ((512, c.flat), (512,)),
        "output_layers.0": ((A, 512), (A,)), "output_layers.1": ((1, 512), (1,)),
    }

  def named_views(self, flat):
    """state_dict-named views (weight, bias per layer) into a flat buffer."""
    out = {}
    c = self.ctx
    for i, name in enumerate(PARAM_NAMES):
      wshape, bshape = self.shapes[name]
      wn = 1
      for d in wshape:
        wn *= d
      out[f"{name}.weight"] = flat[c.off_w[i]:c.off_w[i] + wn].view(wshape)
      out[f"{name}.bias"] = flat[c.off_b[i]:c.off_b[i] + bshape[0]].view(bshape)
    return out

  def load_state_dict(self, state):
    views = self.named_views(self.params)
    with torch.no_grad():
      for key, view in views.items():
        view.copy_(torch.as_tensor(state[key]).to(self.device, torch.float32))
    self.mark_dirty()

  def mark_dirty(self):
    self._packed_version = None
    self._train_packed_version = None

  def watch(self, tensors):
    """Registers tensors that alias the flat parameter buffer with their OWN version counter
    (an ``nn.Parameter`` whose ``.data`` was re-pointed at a view does not share the buffer's):
    ``pack`` looks at all of them, so an in-place write through ``model.parameters()`` -- a
    torch optimizer, ``nn.init``, ``dist.broadcast(p)``, ``p.copy_()`` -- refreshes the mirrors."""
    self._watched = [self.params] + [t for t in tensors if t is not self.params]

  def _version(self):
    return sum(t._version for t in self._watched)

  def pack(self, force=False):
    """Refreshes the packed mirrors if the parameters changed: torch bumps ``_version`` on
    every in-place write (of the flat buffer and of each watched Parameter); native steps
    write through raw pointers and call mark_dirty."""
    version = self._version()
    if force or self._packed_version != version:
      _lib.call("dx_cnn_pack", ctypes.byref(self.ctx), _lib.stream_ptr(self.device))
      self._packed_version = version

  def _obs_args(self, obs, sample_idx):
    if not obs.is_cuda or not obs.is_contiguous():
      raise ValueError("observations must be a contiguous GPU tensor")
    if obs.dtype not in (torch.uint8, torch.float32):
      raise ValueError(f"observations must be uint8 or float32, got {obs.dtype}")
    if tuple(obs.shape[1:]) != self.input_shape:
      raise ValueError(f"observations must be (B,{self.input_shape}), got {tuple(obs.shape)}")
    if sample_idx is not None:
      if sample_idx.dtype != torch.int32 or not sample_idx.is_cuda or not sample_idx.is_contiguous():
        raise ValueError("sample_idx must be a contiguous int32 GPU tensor")
      batch = sample_idx.numel()
    else:
      batch = obs.shape[0]
    if not 1 <= batch <= self.ctx.max_batch:
      raise ValueError(f"batch {batch} outside [1, max_batch={self.ctx.max_batch}]")
    return batch, int(obs.dtype == torch.uint8)

  def forward(self, obs, sample_idx=None):
    """Returns the padded head output (B, 32): logits in [:, :A], value in [:, A]."""
    batch, is_u8 = self._obs_args(obs, sample_idx)
    self.pack()
    _lib.call("dx_cnn_forward", ctypes.byref(self.ctx), _lib.ptr(obs), is_u8,
              _lib.ptr(sample_idx), batch, _lib.stream_ptr(self.device))
    return self.head[:batch * 32].view(batch, 32)

  def act(self, obs, actions, log_prob, values, uniforms=None, seed=0, counter=0):
    """Rollout step: forward + both heads + sampling in the fused path (dx_cnn_act); writes
    actions (B,) int64, log_prob (B,) f32, values (B,) f32 in place."""
    batch, is_u8 = self._obs_args(obs, None)
    self.pack()
    _lib.call("dx_cnn_act", ctypes.byref(self.ctx), _lib.ptr(obs), is_u8, batch,
              _lib.ptr(uniforms), int(seed), int(counter), _lib.ptr(actions), _lib.ptr(log_prob),
              _lib.ptr(values), _lib.stream_ptr(self.device))

  def rollout_synth(self, buffers, horizon, nenvs, policy_seed, policy_counter, env_seed,
                    env_counter, p_reward, p_reset):
    """Enqueues `horizon` (act, synthetic env step) pairs from one native call."""
    self.pack()
    _lib.call("dx_cnn_rollout_synth", ctypes.byref(self.ctx), _lib.ptr(buffers["obs"]),
              int(horizon), int(nenvs), _lib.ptr(buffers["actions"]), _lib.ptr(buffers["log_prob"]),
              _lib.ptr(buffers["values"]), _lib.ptr(buffers["rewards"]), _lib.ptr(buffers["resets"]),
              int(policy_seed), int(policy_counter), int(env_seed), int(env_counter),
              float(p_reward), float(p_reset), _lib.stream_ptr(self.device))

  def fused_heads(self):
    """Whether heads + loss + the heads' backward run as ONE launch (dx_cnn_heads_loss_f32): up to 18 actions where the
    linear layer + heads are one affine map of y2 (84 x 84 frames), up to 7 on the layer-by-layer route."""
    return bool(_lib.load().dx_cnn_fused_heads(ctypes.byref(self.ctx)))

  def _loss_counter(self):
    if getattr(self, "_counter", None) is None:
      self._counter = torch.zeros(4, dtype=torch.int32, device=self.device)
    return self._counter

  def forward_trunk(self, obs, sample_idx=None):
    """conv stack + linear layer of a training minibatch (ctx.hid); the heads run in ``heads_loss``."""
    batch, is_u8 = self._obs_args(obs, sample_idx)
    self.pack()
    _lib.call("dx_cnn_forward_trunk", ctypes.byref(self.ctx), _lib.ptr(obs), is_u8,
              _lib.ptr(sample_idx), batch, _lib.stream_ptr(self.device))
    return batch

  def heads_loss(self, batch, actions, old_log_prob, advantages, old_values, value_targets, mode, cliprange,
                 value_loss_coef, entropy_coef, global_batch, partials, loss_out):
    """Heads forward, categorical PPO / A2C loss, the heads' dgrad and weight-gradient slabs and the
    loss scalars in ONE launch (dx_cnn_heads_loss_f32); continue with ``backward(part=3)`` (or 2, 1)."""
    self._ensure_backward()
    _lib.call("dx_cnn_heads_loss_f32", ctypes.byref(self.ctx), _lib.ptr(actions), _lib.ptr(old_log_prob),
              _lib.ptr(advantages), _lib.ptr(old_values), _lib.ptr(value_targets), None, 0.0, None, int(batch),
              int(mode), float(cliprange if cliprange is not None else -1.0), float(value_loss_coef),
              float(entropy_coef), int(global_batch), _lib.ptr(partials), partials.numel(),
              _lib.ptr(self._loss_counter()), _lib.ptr(loss_out), _lib.stream_ptr(self.device))
    return self.head[:batch * 32].view(batch, 32)

  def backward(self, obs, sample_idx=None, part=None):
    """Consumes self.dhead (B, 32) and fills self.grads (same obs / sample_idx as forward).
    ``part`` 0 / 1 runs the two halves separately (``tail_offset`` splits the gradient buffer):
    after part 0 the gradients of the linear layer and the heads are final."""
    batch, is_u8 = self._obs_args(obs, sample_idx)
    self._ensure_backward()
    if part is None:
      _lib.call("dx_cnn_backward", ctypes.byref(self.ctx), _lib.ptr(obs), is_u8,
                _lib.ptr(sample_idx), batch, _lib.stream_ptr(self.device))
    else:
      _lib.call("dx_cnn_backward_part", ctypes.byref(self.ctx), _lib.ptr(obs), is_u8,
                _lib.ptr(sample_idx), batch, int(part), _lib.stream_ptr(self.device))
    return self.grads

  # what Trainer looks at before handing updates to ``ppo_epoch``
  native_allreduce = True      # the gradient exchange happens inside the native call
  single_native_update = True  # a lone minibatch (A2C; PPO outside an epoch) goes the same way
  native_rmsprop = True

  def ppo_epoch(self, context, loss, optimizer, first_step, record_norms=False):
    """Enqueues every minibatch update of ``context`` (runners.onpolicy.EpochContext) from one C
    call (dx_cnn_ppo_epoch): advantage normalisation, weight packing, forward on the frames the
    epoch's index selects, fused loss, backward (with the two gradient all-reduces of a sharded run
    in between), gradient norm, clip + Adam / RMSprop per minibatch.  Returns the number of
    updates; the packed mirrors are current when it returns."""
    arrays = context.shuffled
    samples, mbsize = int(context.sample_size), int(context.mbsize)
    f32 = torch.float32
    dev = self.device

    def need(key, dtype=f32):
      t = arrays.get(key)
      if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous() or t.shape[0] != samples:
        raise _lib.NativeError(f"native epoch: '{key}' is not an epoch array of {samples} rows on the device")
      if t.dtype != dtype:
        raise _lib.NativeError(f"native epoch: '{key}' must be {dtype}, got {t.dtype}")
      return t

    obs = context.lazy.get("observations")
    if obs is None:
      obs = arrays.get("observations")
    index = context.order_dev
    if index is not None:
      if index.dtype != torch.int32 or not index.is_cuda or not index.is_contiguous() or index.numel() != samples:
        raise _lib.NativeError("native epoch: the epoch's index must be a contiguous int32 GPU tensor")
    elif obs is None or obs.shape[0] != samples:
      raise _lib.NativeError("native epoch: observations do not match the epoch's sample count")
    if (not isinstance(obs, torch.Tensor) or not obs.is_cuda or not obs.is_contiguous()
        or obs.dtype not in (torch.uint8, f32) or tuple(obs.shape[1:]) != self.input_shape):
      raise _lib.NativeError(f"native epoch: observations must be a contiguous (n,{self.input_shape}) "
                             "uint8 / float32 GPU tensor")
    ppo = loss["mode"] == 0
    actions = need("actions", torch.int64)
    old_lp = need("log_prob") if ppo else None
    old_v = need("values").reshape(-1) if ppo else None
    adv = need("advantages").reshape(-1)
    vt = need("value_targets").reshape(-1)
    self.reserve(mbsize)
    self._ensure_backward()
    updates = context.num_minibatches
    normalize = context.norm_eps is not None  # NormalizeAdvantages opted in with its epsilon
    context.normalized = torch.empty(samples, dtype=f32, device=dev) if normalize else None
    context.losses = torch.empty((updates, 8), dtype=f32, device=dev)
    stats_ready = context.stats_ready if normalize else None
    sharded = distributed.sharded()
    if sharded and normalize and stats_ready is None:
      raise _lib.NativeError("native epoch: a sharded run needs the prepared global advantage statistics")
    if stats_ready is not None and (stats_ready.dtype != torch.float64 or not stats_ready.is_contiguous()
                                    or tuple(stats_ready.shape) != (updates, 3)):
      raise _lib.NativeError("native epoch: statistics must be a contiguous (minibatches, 3) float64 tensor")
    if getattr(self, "_epoch_scratch", None) is None or self._epoch_scratch.numel() < 3 * updates:
      self._epoch_scratch = torch.empty(3 * updates, dtype=torch.float64, device=dev)
    capacity = 8 * ((mbsize + 7) // 8)
    if getattr(self, "_epoch_partials", None) is None or self._epoch_partials.numel() < capacity:
      self._epoch_partials = torch.empty(capacity, dtype=torch.float64, device=dev)
    state0, state1, beta1, beta2 = optimizer.native_state()
    e = _lib.CnnEpoch()
    e.struct_bytes = ctypes.sizeof(_lib.CnnEpoch)
    e.mbsize, e.samples = mbsize, samples
    e.obs, e.obs_is_u8, e.mode = obs.data_ptr(), int(obs.dtype == torch.uint8), int(loss["mode"])
    e.index = index.data_ptr() if index is not None else None
    e.actions = actions.data_ptr()
    e.old_log_prob = old_lp.data_ptr() if ppo else None
    e.old_values = old_v.data_ptr() if ppo else None
    e.advantages, e.value_targets = adv.data_ptr(), vt.data_ptr()
    e.normalize, e.norm_eps = int(normalize), float(context.norm_eps) if normalize else 0.0
    e.stats_ready = stats_ready.data_ptr() if stats_ready is not None else None
    e.stats = self._epoch_scratch.data_ptr()
    e.adv_normalized = context.normalized.data_ptr() if normalize else None
    cliprange = loss.get("cliprange")
    e.cliprange = float(cliprange) if cliprange is not None else -1.0
    e.value_loss_coef, e.entropy_coef = float(loss["value_loss_coef"]), float(loss["entropy_coef"])
    e.world, e.allreduce = distributed.world_size(), int(sharded)
    e.optimizer, e.npartials = int(optimizer.native_kind), optimizer.partials.numel()
    e.state0 = state0.data_ptr()
    e.state1 = state1.data_ptr() if state1 is not None else None
    e.sumsq_partials = optimizer.partials.data_ptr()
    e.loss_partials, e.loss_partials_capacity = self._epoch_partials.data_ptr(), int(capacity)
    e.max_grad_norm = float(optimizer.max_grad_norm) if optimizer.max_grad_norm is not None else 0.0
    e.lr, e.beta1, e.beta2 = optimizer.current_lr(), beta1, beta2
    e.opt_eps, e.first_step = float(optimizer.eps), int(first_step)
    if record_norms:  # one pre-clip norm per minibatch for the summaries
      context.grad_norms = torch.empty(updates, dtype=f32, device=dev)
      e.grad_norm_out, e.grad_norm_stride = context.grad_norms.data_ptr(), 1
    else:
      e.grad_norm_out, e.grad_norm_stride = optimizer.grad_norm.data_ptr(), 0
    e.loss_out = context.losses.data_ptr()
    # every mirror current: 1; only the training kernels' (the previous epoch of this rollout ended with the light
    # pack): 2; neither: the call packs before its first minibatch
    version = self._version()
    e.mirrors_current = 1 if self._packed_version == version else (2 if self._train_packed_version == version else 0)
    e.more_epochs = int(bool(getattr(context, "more_epochs", False)))
    e.loss_counter = self._loss_counter().data_ptr() if self.fused_heads() else None
    keep = (obs, index, actions, old_lp, old_v, adv, vt, stats_ready)  # alive until enqueued
    _lib.call("dx_cnn_ppo_epoch", ctypes.byref(self.ctx), ctypes.byref(e), _lib.stream_ptr(dev))
    del keep
    if record_norms:
      optimizer.grad_norm.copy_(context.grad_norms[-1:])
    # every update inside the call is followed by a pack: of every mirror, or -- when another epoch of the same
    # rollout follows -- of what the training kernels read only (anything else packs first: _ensure_packed)
    # (the native steps write through raw pointers: torch's version counters do not move, the marks are set by hand)
    self._train_packed_version = self._version()
    self._packed_version = None if e.more_epochs else self._train_packed_version
    return updates

  @property
  def tail_offset(self):
    """First element of the (linear layer + heads) part of the flat parameter / gradient buffer."""
    return int(self.ctx.off_w[3])
