"""Device state of the two-net tanh MLP actor-critic (derl/models.py:224-271): flat
parameter / gradient buffers in ``state_dict`` order, padded / transposed mirrors,
activation and slab workspaces, and the calls into the C-ABI (dx_mlp_*)."""
import ctypes

import torch

from . import _lib


class MlpEngine:
  """Policy net obs_dim -> 64 -> 64 -> P and value net -> 1, optional ``logstd`` (P)."""

  def __init__(self, obs_dim, policy_out, has_logstd, max_batch=256, device="cuda"):
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise _lib.NativeError("MlpEngine needs a HIP device; derl_amd has no CPU path")
    ctx = _lib.MlpCtx()
    ctx.struct_bytes = ctypes.sizeof(_lib.MlpCtx)
    ctx.obs_dim, ctx.policy_out = int(obs_dim), int(policy_out)
    ctx.has_logstd, ctx.max_batch = int(bool(has_logstd)), int(max_batch)
    _lib.call("dx_mlp_init", ctypes.byref(ctx))
    self.ctx = ctx
    self.obs_dim, self.policy_out, self.has_logstd = int(obs_dim), int(policy_out), bool(has_logstd)
    f32 = dict(dtype=torch.float32, device=self.device)
    self.params = torch.zeros(ctx.param_count, **f32)
    self.grads = torch.zeros(ctx.param_count, **f32)
    self.packed = torch.zeros(ctx.packed_count, **f32)
    ctx.params, ctx.grads, ctx.packed = (t.data_ptr() for t in (self.params, self.grads, self.packed))
    self._allocate_workspaces()
    self._packed_version = None
    self._watched = [self.params]

  def _allocate_workspaces(self):
    ctx = self.ctx
    f32 = dict(dtype=torch.float32, device=self.device)
    self.xpad = torch.empty(ctx.x_count, **f32)
    self.h = [torch.empty(ctx.h_count, **f32) for _ in range(4)]
    self.head = torch.zeros(ctx.head_count, **f32)  # columns beyond P+1 stay zero
    self.dhead = torch.zeros(ctx.head_count, **f32)
    self.da = torch.empty(ctx.h_count, **f32)
    self.db = torch.empty(ctx.h_count, **f32)
    self.slabs = torch.empty(ctx.slab_count, **f32)
    ctx.xpad = self.xpad.data_ptr()
    ctx.h1[0], ctx.h1[1] = self.h[0].data_ptr(), self.h[1].data_ptr()
    ctx.h2[0], ctx.h2[1] = self.h[2].data_ptr(), self.h[3].data_ptr()
    for name in ("head", "dhead", "da", "db", "slabs"):
      setattr(ctx, name, getattr(self, name).data_ptr())

  def reserve(self, max_batch):
    if max_batch <= self.ctx.max_batch:
      return
    self.ctx.max_batch = int(max_batch)
    _lib.call("dx_mlp_init", ctypes.byref(self.ctx))
    self._allocate_workspaces()

  def named_views(self, flat):
    """state_dict-named views: logstd, module_list.{0,1}.{0,2,4}.{weight,bias}."""
    c = self.ctx
    out = {}
    if self.has_logstd:
      out["logstd"] = flat[c.off_logstd:c.off_logstd + self.policy_out]
    for net in range(2):
      outs = self.policy_out if net == 0 else 1
      dims = [(64, self.obs_dim), (64, 64), (outs, 64)]
      for layer, (no, ni) in enumerate(dims):
        i = 3 * net + layer
        out[f"module_list.{net}.{2 * layer}.weight"] = flat[c.off_w[i]:c.off_w[i] + no * ni].view(no, ni)
        out[f"module_list.{net}.{2 * layer}.bias"] = flat[c.off_b[i]:c.off_b[i] + no]
    return out

  def load_state_dict(self, state):
    with torch.no_grad():
      for key, view in self.named_views(self.params).items():
        view.copy_(torch.as_tensor(state[key]).to(self.device, torch.float32))
    self.mark_dirty()

  def mark_dirty(self):
    self._packed_version = None

  def watch(self, tensors):
    """See CnnEngine.watch: Parameters aliasing the flat buffer keep their own version counter."""
    self._watched = [self.params] + [t for t in tensors if t is not self.params]

  def _version(self):
    return sum(t._version for t in self._watched)

  def pack(self, force=False):
    version = self._version()
    if force or self._packed_version != version:
      _lib.call("dx_mlp_pack", ctypes.byref(self.ctx), _lib.stream_ptr(self.device))
      self._packed_version = version

  def check_health(self):
    """Raises NativeError once a persistent epoch of this engine has given up at a grid barrier
    (csrc/mlp_persist.hip): the status word is pinned host memory the library fills behind every
    persistent launch -- reading it costs nothing and needs no synchronisation.  That epoch left
    parameters, moments and gradient untouched; nothing this engine computed since is trustworthy."""
    status = getattr(self, "_persist_status_np", None)
    if status is not None and status[0] != 0:
      code = (int(status[0]) & 0x7fffffff) - 1
      raise _lib.NativeError(
          f"persistent MLP epoch gave up at grid barrier {code % 2} of minibatch {code // 2}: not every "
          "workgroup arrived within the spin limit; the epoch's parameters were NOT stepped and its losses "
          "are NaN (set MlpEngine.persistent_epochs = False to train on the launch-per-stage epoch)")

  def forward(self, obs):
    """obs (B, obs_dim) float32 on the device -> padded head (B, 32)."""
    self.check_health()
    if not obs.is_cuda or obs.dtype != torch.float32 or not obs.is_contiguous():
      raise ValueError("observations must be a contiguous float32 GPU tensor")
    if obs.ndim != 2 or obs.shape[1] != self.obs_dim:
      raise ValueError(f"observations must be (B, {self.obs_dim}), got {tuple(obs.shape)}")
    batch = obs.shape[0]
    self.reserve(batch)
    self.pack()
    _lib.call("dx_mlp_forward", ctypes.byref(self.ctx), _lib.ptr(obs), batch,
              _lib.stream_ptr(self.device))
    return self.head[:batch * 32].view(batch, 32)

  def rollout_synth(self, buffers, horizon, nenvs, policy_seed, policy_counter, env_seed, env_counter, p_reset):
    """Enqueues `horizon` (act, synthetic MuJoCo-shaped env step) pairs as ONE launch.  Raises NativeError
    (DX_ENOSUP) for a categorical MLP or observations wider than 64."""
    self.check_health()
    self.reserve(nenvs)
    self.pack()
    _lib.call("dx_mlp_rollout_synth", ctypes.byref(self.ctx), _lib.ptr(buffers["obs"]), int(horizon), int(nenvs),
              _lib.ptr(buffers["actions"]), _lib.ptr(buffers["log_prob"]), _lib.ptr(buffers["values"]),
              _lib.ptr(buffers["rewards"]), _lib.ptr(buffers["resets"]), int(policy_seed), int(policy_counter),
              int(env_seed), int(env_counter), float(p_reset), _lib.stream_ptr(self.device))

  def backward(self, batch):
    """Consumes self.dhead (B, 32) and fills self.grads (logstd is written by the loss)."""
    _lib.call("dx_mlp_backward", ctypes.byref(self.ctx), int(batch), _lib.stream_ptr(self.device))
    return self.grads

  persistent_epochs = True  # False: the launch-per-stage epoch also where the persistent one applies

  def ppo_epoch(self, context, loss, optimizer, first_step, record_norms=False):
    """Enqueues every minibatch update of ``context`` (runners.onpolicy.EpochContext): advantage
    normalisation, forward, fused loss, backward, gradient norm, clip + Adam per minibatch, all
    from one C call.  Returns the number of updates."""
    self.check_health()
    arrays = context.shuffled
    samples, mbsize = context.sample_size, context.mbsize
    f32 = torch.float32

    def need(key, dtype=f32):
      t = arrays.get(key)
      if t is None or not t.is_cuda or not t.is_contiguous() or t.shape[0] != samples:
        raise _lib.NativeError(f"native epoch: '{key}' is not an epoch array of {samples} rows on the device")
      return t if t.dtype == dtype else t.to(dtype)

    obs = need("observations")
    if obs.ndim != 2 or obs.shape[1] != self.obs_dim:
      raise _lib.NativeError(f"native epoch: observations must be (n, {self.obs_dim})")
    actions = need("actions", f32 if self.has_logstd else torch.int64)
    ppo = loss["mode"] == 0
    old_lp = need("log_prob") if ppo else None
    old_v = need("values").reshape(-1) if ppo else None
    adv = need("advantages").reshape(-1)
    vt = need("value_targets").reshape(-1)
    self.reserve(mbsize)
    updates = context.num_minibatches
    dev = self.device
    normalize = context.norm_eps is not None  # NormalizeAdvantages opted in with its epsilon
    context.normalized = torch.empty(samples, dtype=f32, device=dev) if normalize else None
    context.losses = torch.empty((updates, 8), dtype=f32, device=dev)
    if getattr(self, "_epoch_scratch", None) is None:
      self._epoch_scratch = torch.empty(3, dtype=torch.float64, device=dev)
    capacity = 40 * ((mbsize + 255) // 256) if self.has_logstd else 8 * ((mbsize + 7) // 8)
    if getattr(self, "_epoch_partials", None) is None or self._epoch_partials.numel() < capacity:
      self._epoch_partials = torch.empty(capacity, dtype=torch.float64, device=dev)
    e = _lib.MlpEpoch()
    e.struct_bytes = ctypes.sizeof(_lib.MlpEpoch)
    e.mbsize, e.samples = int(mbsize), int(samples)
    e.obs, e.actions = obs.data_ptr(), actions.data_ptr()
    e.action_is_f32, e.mode = int(self.has_logstd), int(loss["mode"])
    e.old_log_prob = old_lp.data_ptr() if ppo else None
    e.old_values = old_v.data_ptr() if ppo else None
    e.advantages, e.value_targets = adv.data_ptr(), vt.data_ptr()
    e.normalize, e.norm_eps = int(normalize), float(context.norm_eps) if normalize else 0.0
    cliprange = loss.get("cliprange")
    e.cliprange = float(cliprange) if cliprange is not None else -1.0
    e.value_loss_coef, e.entropy_coef = float(loss["value_loss_coef"]), float(loss["entropy_coef"])
    e.global_batch = 0
    e.adv_normalized = context.normalized.data_ptr() if normalize else None
    e.stats = self._epoch_scratch.data_ptr()
    e.exp_avg, e.exp_avg_sq = optimizer.exp_avg.data_ptr(), optimizer.exp_avg_sq.data_ptr()
    e.sumsq_partials, e.npartials = optimizer.partials.data_ptr(), optimizer.partials.numel()
    e.loss_partials, e.loss_partials_capacity = self._epoch_partials.data_ptr(), int(capacity)
    e.max_grad_norm = float(optimizer.max_grad_norm) if optimizer.max_grad_norm is not None else 0.0
    e.lr, e.beta1, e.beta2 = optimizer.current_lr(), float(optimizer.betas[0]), float(optimizer.betas[1])
    e.adam_eps, e.first_step = float(optimizer.eps), int(first_step)
    if record_norms:  # one pre-clip norm per minibatch for the summaries
      context.grad_norms = torch.empty(updates, dtype=f32, device=dev)
      e.grad_norm_out, e.grad_norm_stride = context.grad_norms.data_ptr(), 1
    else:
      e.grad_norm_out, e.grad_norm_stride = optimizer.grad_norm.data_ptr(), 0
    e.loss_out = context.losses.data_ptr()
    # the persistent form (one launch per epoch) where the library covers the shape
    e.persistent = 0
    if self.persistent_epochs:
      groups, nbytes = ctypes.c_int(0), ctypes.c_longlong(0)
      _lib.call("dx_mlp_persist_plan", ctypes.byref(self.ctx), int(mbsize), int(samples),
                ctypes.byref(groups), ctypes.byref(nbytes))
      if groups.value > 0:
        if getattr(self, "_persist_ws", None) is None or self._persist_ws.numel() < nbytes.value:
          self._persist_ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
        if getattr(self, "_persist_stats", None) is None or self._persist_stats.numel() < 3 * updates:
          self._persist_stats = torch.empty(3 * updates, dtype=torch.float64, device=dev)
        if getattr(self, "_persist_status", None) is None:  # one pinned word, see check_health
          self._persist_status = torch.zeros(1, dtype=torch.int32).pin_memory()
          self._persist_status_np = self._persist_status.numpy()
        e.persistent = 1
        e.workspace, e.workspace_bytes = self._persist_ws.data_ptr(), int(self._persist_ws.numel())
        e.stats_all = self._persist_stats.data_ptr()
        e.status_host = self._persist_status.data_ptr()
    keep = (obs, actions, old_lp, old_v, adv, vt)  # alive until the call has been enqueued
    _lib.call("dx_mlp_ppo_epoch", ctypes.byref(self.ctx), ctypes.byref(e), _lib.stream_ptr(dev))
    self.last_epoch_route = "persistent" if _lib.load().dx_mlp_last_route() == 1 else "per-stage"
    del keep
    if record_norms:
      optimizer.grad_norm.copy_(context.grad_norms[-1:])
    self.mark_dirty()
    return updates
