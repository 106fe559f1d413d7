"""Device state of the Nature-DQN actor-critic: flat parameter / gradient buffers in the
reference's ``state_dict`` order, packed weight mirrors, activation and slab workspaces,
and the calls into the C-ABI (dx_cnn_*).  torch only allocates memory and hands out the
stream."""
import ctypes

import torch

from . import _lib

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

  @property
  def tail_offset(self):
    """First element of the (linear layer + heads) part of the flat parameter / gradient buffer."""
    return int(self.ctx.off_w[3])
