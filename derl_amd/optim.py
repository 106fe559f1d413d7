"""Optimizers over the model's FLAT parameter buffer: one fused launch per step that reads
the gradient norm partials, clips, and applies Adam / RMSprop (dx_clip_adam_step_f32 /
dx_clip_rmsprop_step_f32).  Constructor arguments follow torch.optim's so the factories
read like the reference (derl/factory/ppo.py:78-81, derl/factory/a2c.py:68-73); ``lr`` may
be the 0-dim tensor a LinearAnneal mutates in place."""
import torch

from . import distributed, ops


def _flat(model):
  engine = getattr(model, "engine", None)
  if engine is None:
    raise TypeError("derl_amd optimizers need a model backed by a device engine")
  return engine


class _FlatOptimizer:
  def __init__(self, model, lr):
    self.model = model
    self.engine = _flat(model)
    self.lr = lr
    self.max_grad_norm = None  # set by Trainer: clipping is fused into the step
    self.step_count = 0
    self.partials = torch.zeros(ops.NORM_PARTIALS, dtype=torch.float64, device=self.engine.device)
    self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.engine.device)
    self.param_groups = [dict(params=list(model.parameters()), lr=lr)]
    self._pending = []
    if distributed.sharded():
      # replicas start from rank 0's parameters whatever each process seeded its init with
      distributed.broadcast_(self.engine.params)
      self.engine.mark_dirty()

  def current_lr(self):
    return float(self.lr)

  def zero_grad(self, set_to_none=False):
    """Gradients are overwritten by every backward; nothing to clear."""

  def reduce_part(self, part):
    """Data-parallel overlap (the one exchange step of the path): called by the
    model's backward when one half of the flat gradient buffer is final -- part 0 = linear layer
    + heads (95 % of the bytes, ready before the conv layers' backward starts), part 1 = the
    conv layers -- and starts that half's all-reduce on the communicator's stream."""
    if not distributed.sharded():
      return
    off = self.engine.tail_offset
    grads = self.engine.grads
    handle = distributed.all_reduce_sum_async(grads[off:] if part == 0 else grads[:off])
    if handle is not None:
      self._pending.append(handle)

  def reduce_and_norm(self):
    """All-reduce (when sharded; or wait for the halves started by ``reduce_part``) and the
    float64 partial sums of g^2."""
    if self._pending:
      for handle in self._pending:
        handle.wait()
      self._pending = []
    else:
      distributed.all_reduce_mean_grads(self.engine.grads)
    ops.grad_sumsq(self.engine.grads, self.partials)


class Adam(_FlatOptimizer):
  """torch.optim.Adam semantics (no weight decay / amsgrad)."""
  def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    super().__init__(model, lr)
    self.betas, self.eps = betas, eps
    self.exp_avg = torch.zeros_like(self.engine.params)
    self.exp_avg_sq = torch.zeros_like(self.engine.params)

  def step(self):
    self.step_count += 1
    ops.clip_adam_step(self.engine.params, self.engine.grads, self.exp_avg, self.exp_avg_sq,
                       self.partials, self.max_grad_norm, self.current_lr(), self.step_count,
                       self.betas[0], self.betas[1], self.eps, self.grad_norm)
    self.engine.mark_dirty()

  native_kind = 0  # dx_cnn_epoch.optimizer

  def native_state(self):
    return self.exp_avg, self.exp_avg_sq, float(self.betas[0]), float(self.betas[1])

  def native_epoch(self, loss_fn, context, record_norms=False):
    """Every minibatch update of an epoch from one native call (dx_mlp_ppo_epoch /
    dx_cnn_ppo_epoch); fills ``context.losses`` / ``context.normalized`` (and, when summaries are
    being recorded, ``context.grad_norms``) and advances the step count.  The engine leaves its
    packed mirrors in the state its kernels need."""
    updates = self.engine.ppo_epoch(context, loss_fn.epoch_arguments(), self, self.step_count + 1,
                                    record_norms=record_norms)
    self.step_count += updates

  def state_dict(self):
    return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq)


class RMSprop(_FlatOptimizer):
  """torch.optim.RMSprop semantics (no momentum, not centered)."""
  def __init__(self, model, lr=1e-2, alpha=0.99, eps=1e-8):
    super().__init__(model, lr)
    self.alpha, self.eps = alpha, eps
    self.square_avg = torch.zeros_like(self.engine.params)

  def step(self):
    self.step_count += 1
    ops.clip_rmsprop_step(self.engine.params, self.engine.grads, self.square_avg, self.partials,
                          self.max_grad_norm, self.current_lr(), self.alpha, self.eps,
                          self.grad_norm)
    self.engine.mark_dirty()

  native_kind = 1  # dx_cnn_epoch.optimizer

  def native_state(self):
    return self.square_avg, None, float(self.alpha), 0.0

  def native_epoch(self, loss_fn, context, record_norms=False):
    """See Adam.native_epoch (engines whose native call knows RMSprop: dx_cnn_ppo_epoch)."""
    if not getattr(self.engine, "native_rmsprop", False):
      raise TypeError("this engine's native epoch implements Adam only")
    updates = self.engine.ppo_epoch(context, loss_fn.epoch_arguments(), self, self.step_count + 1,
                                    record_norms=record_norms)
    self.step_count += updates

  def state_dict(self):
    return dict(step=self.step_count, square_avg=self.square_avg)
