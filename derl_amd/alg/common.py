"""Algorithm base classes (derl/alg/common.py:9-106)."""
from abc import ABC, abstractmethod

import torch

from .. import distributed, summary
from ..optim import _FlatOptimizer


def r_squared(targets, predictions):
  """Coefficient of determination 1 - MSE / Var[predictions] with the unbiased variance, the
  quantity the reference logs (alg/common.py:9-12); off the hot path (the fused loss kernels
  produce the same number among their eight outputs)."""
  residual = (predictions - targets).square().mean()
  return 1. - residual / predictions.var(unbiased=True)


def total_norm(tensors, norm_type=2):
  """p-norm of all tensors taken as ONE vector (alg/common.py:15-20); ``inf`` gives the largest
  magnitude.  Fallback for torch optimizers -- flat-buffer optimizers get the norm from
  ``dx_grad_sumsq_f32``."""
  tensors = list(tensors)
  if norm_type == float("inf"):
    return torch.stack([t.detach().abs().max() for t in tensors]).max()
  powered = torch.stack([t.detach().abs().pow(norm_type).sum() for t in tensors])
  return powered.sum().pow(1. / norm_type)


class _LossBackward(torch.autograd.Function):
  """The scalar a device loss returns: ``.backward()`` runs the engine's backward from the
  head gradient the fused loss kernel already produced."""

  @staticmethod
  def forward(ctx, anchor, value, backward_fn):
    ctx.backward_fn = backward_fn
    return value.clone()

  @staticmethod
  def backward(ctx, grad_output):
    ctx.backward_fn(grad_output)
    return None, None, None


class Loss(ABC):
  """Algorithm loss function (alg/common.py:23-45)."""
  def __init__(self, model, name=None):
    self.model = model
    if name is None:
      name = self.__class__.__name__
      name = name[:-len("Loss")] if name.endswith("Loss") else name
      name = name.lower()
    self.name = name
    self.call_count = 0

  @property
  def device(self):
    return next(self.model.parameters()).device

  def to_device(self, arr, dtype=None):
    """Host array or tensor -> contiguous device tensor (the reference's
    torch_from_numpy, alg/common.py:39-41); device tensors pass through."""
    if isinstance(arr, torch.Tensor) and arr.is_cuda and (dtype is None or arr.dtype == dtype) \
        and arr.is_contiguous():
      return arr  # minibatches of the device-resident runner: nothing to do
    if not isinstance(arr, torch.Tensor):
      arr = torch.as_tensor(arr)
    return arr.to(device=self.device, dtype=dtype).contiguous()

  torch_from_numpy = to_device

  @abstractmethod
  def __call__(self, data):
    """Computes and returns loss value on given data."""


class Trainer:
  """Performs algorithm training steps (alg/common.py:48-78): loss -> zero_grad ->
  backward -> global-norm clip -> anneals -> optimizer step.  With a derl_amd optimizer the
  gradient all-reduce (sharded batches), the norm, the clip and the update are three
  launches on the flat buffers; a torch optimizer still works through the parameter views."""
  def __init__(self, optimizer, anneals=None, max_grad_norm=None):
    self.optimizer = optimizer
    self.anneals = anneals or []
    self.max_grad_norm = max_grad_norm
    self.step_count = 0

  def preprocess_gradients(self, parameters, tag):
    grad_norm = None
    if isinstance(self.optimizer, _FlatOptimizer):
      self.optimizer.max_grad_norm = self.max_grad_norm
      self.optimizer.reduce_and_norm()
      # the pre-clip norm is formed inside the fused optimizer launch: ``step`` records it
      # afterwards (recording the buffer here would log the PREVIOUS step's norm)
      self._pending_norm_tag = tag if summary.should_record() else None
      return
    parameters = list(parameters)
    if self.max_grad_norm is not None:
      grad_norm = torch.nn.utils.clip_grad_norm_(parameters, self.max_grad_norm)
    if summary.should_record():
      if grad_norm is None:
        grad_norm = total_norm(p.grad for p in parameters if p.grad is not None)
      summary.add_scalar(tag, grad_norm, global_step=self.step_count)

  # -- native epochs ------------------------------------------------------------------------
  native_epochs = True  # False: always one host call per update

  def _native_ready(self, alg):
    """The engine, or None when this trainer / algorithm cannot hand updates to a native call:
    needs a flat optimizer with ``native_epoch``, the stock ``Alg.loss``, a loss with
    ``epoch_arguments`` and an engine with ``ppo_epoch``; a sharded run additionally the library's
    own communicator (the gloo rehearsal mode reduces from Python, update by update)."""
    if not self.native_epochs:
      return None
    engine = getattr(alg.model, "engine", None)
    if (not hasattr(self.optimizer, "native_epoch") or type(alg).loss is not Alg.loss
        or not hasattr(engine, "ppo_epoch") or not hasattr(alg.loss_fn, "epoch_arguments")):
      return None
    if distributed.sharded() and not (getattr(engine, "native_allreduce", False)
                                      and distributed.native_comm()):
      return None
    return engine

  _EPOCH_ORDER = (
      "Trainer.step: the native epoch (Trainer.native_epochs) enqueued EVERY update of this epoch "
      "when its first minibatch was stepped, so the reference's one-update-per-step semantics "
      "(derl/alg/common.py:66-78) only hold if the epoch's minibatches are stepped once each, in "
      "order, with nothing reading the policy in between: {what}.  Set "
      "trainer.native_epochs = False to train update by update.")

  def _epoch_fast_path(self, alg, data):
    """(context, k) when EVERY update of this minibatch's epoch can be (or already was) enqueued
    from one native call: the minibatch is the untouched slice of the epoch's arrays (frames: the
    epoch's index into the untouched rollout buffer) and its advantages are either the raw slice
    or what NormalizeAdvantages made of it (the transform records its epsilon in the context: the
    native epoch normalises with THAT epsilon, or not at all).

    What the native epoch cannot give is the reference's view of the parameters BETWEEN the
    minibatch steps of an epoch (all updates are applied at minibatch 0).  That difference is never
    silent: stepping an epoch's minibatches out of order or twice, handing a later minibatch of a
    consumed epoch in edited form, starting another epoch before this one's minibatches were all
    stepped, and ``policy.act`` / ``state_dict`` between two minibatch steps of a consumed epoch
    (``engine.open_epoch``) all raise RuntimeError naming ``native_epochs = False``."""
    engine = self._native_ready(alg)
    if engine is None:
      return None
    state = data.get("state")
    entry = state.get("epoch") if isinstance(state, dict) else None
    if entry is None:
      return None
    context, k = entry
    return self._epoch_fast_path_checked(alg, data, context, k, entry)

  def _epoch_fast_path_checked(self, alg, data, context, k, entry):
    if context.consumed and k != context.next_k:  # (before the comparison: a minibatch stepped twice compares unequal too)
      raise RuntimeError(self._EPOCH_ORDER.format(
          what=f"minibatch {k} was stepped where minibatch {context.next_k} of the epoch is due"))
    result = self._epoch_compare(alg, data, context, k, entry)
    if result is None and context.consumed:
      raise RuntimeError(self._EPOCH_ORDER.format(
          what=f"minibatch {k} of an epoch whose updates are already applied arrived edited (a transform "
               "replaced one of its arrays), so stepping it would apply an update twice"))
    return result

  def _epoch_compare(self, alg, data, context, k, entry):
    if (context.consumed and getattr(data, "epoch", None) is entry and not data.touched
        and context.verified is data.__class__):
      # a LazyMinibatch straight from the iterator whose only outside contact was NormalizeAdvantages'
      # lazy slice of the natively normalised advantages: nothing to compare, nothing was cut
      return context, k
    start = k * context.mbsize

    def same(mine, whole):
      # mine is whole[start:...]?  (pointer arithmetic: slicing `whole` would build a tensor per check)
      if not (isinstance(mine, torch.Tensor) and isinstance(whole, torch.Tensor)):
        return False
      row = whole.stride(0) * whole.element_size() if whole.dim() else 0
      return mine.data_ptr() == whole.data_ptr() + start * row

    # still the epoch's own slices?  (a transform that replaced any of them -- or a pipeline whose
    # advantages are neither the raw slice nor NormalizeAdvantages' output -- trains step by step)
    keys = ["value_targets", "actions"]
    if getattr(alg.loss_fn, "mode", 0) == 0:
      keys += ["log_prob", "values"]
    for key in keys:
      if not same(data.get(key), context.shuffled.get(key)):
        return None
    observations = data.get("observations")
    if "observations" in context.shuffled:
      if not same(observations, context.shuffled["observations"]):
        return None
    else:  # frames referenced by index: the same rollout buffer, the epoch's own index slice
      base = context.lazy.get("observations")
      if (base is None or context.order_dev is None or getattr(observations, "base", None) is not base
          or not same(getattr(observations, "index", None), context.order_dev)):
        return None
    mine = data.get("advantages")
    if context.norm_eps is None:
      if not same(mine, context.shuffled.get("advantages")):
        return None
    elif context.consumed:
      if not same(mine, context.normalized):
        return None
    elif not (isinstance(mine, torch.Tensor) and context.norm_first is not None
              and mine.data_ptr() == context.norm_first.data_ptr()):
      return None
    if not context.consumed and k != 0:
      return None  # joined mid-epoch: step by step
    if k == 0 and getattr(data, "epoch", None) is entry:
      context.verified = data.__class__  # minibatch 0 passed the full comparison: later ones take the short cut
    return context, k

  def _single_update_context(self, alg, data):
    """A one-minibatch "epoch" for engines that take a single update natively (the CNN engine:
    PPO minibatches outside an IterateWithMinibatches epoch, every A2C update): the arrays the
    loss would upload anyway, checked like the loss checks them."""
    engine = self._native_ready(alg)
    if engine is None or not getattr(engine, "single_native_update", False):
      return None
    arrays = alg.loss_fn.native_update_arrays(data)
    if arrays is None:
      return None
    from ..runners.onpolicy import EpochContext  # pylint: disable=import-outside-toplevel
    observations, index = arrays.pop("observations"), arrays.pop("index")
    batch = arrays["actions"].shape[0]
    return EpochContext(arrays, batch, batch, order_dev=index, lazy={"observations": observations})

  def _step_epoch(self, alg, context, k):
    engine = alg.model.engine
    open_context = getattr(engine, "open_epoch", None)
    if open_context is not None and open_context is not context:
      engine.open_epoch = None
      raise RuntimeError(self._EPOCH_ORDER.format(
          what=f"a new epoch was stepped while minibatches {open_context.next_k}.."
               f"{open_context.num_minibatches - 1} of the previous one were never stepped (their updates "
               "are applied nevertheless)"))
    if k != context.next_k:
      raise RuntimeError(self._EPOCH_ORDER.format(
          what=f"minibatch {k} was stepped where minibatch {context.next_k} of the epoch is due"))
    recording = summary.should_record()
    for anneal in self.anneals:
      if recording:
        anneal.summarize(alg.runner.step_count)
      anneal.step_to(alg.runner.step_count)
    if not context.consumed:
      self.optimizer.max_grad_norm = self.max_grad_norm
      self.optimizer.native_epoch(alg.loss_fn, context, record_norms=recording)
      context.consumed = True
    if context.loss_rows is None:  # one unbind per epoch instead of two tensor views per update
      context.loss_rows = context.losses.unbind(0)
      context.loss_scalars = context.losses[:, 0].unbind(0)
    alg.loss_fn.last_terms = context.loss_rows[k]
    if recording:
      alg.loss_fn._summaries(context.loss_rows[k])  # pylint: disable=protected-access
      if context.grad_norms is not None:
        summary.add_scalar(f"{alg.name}/grad_norm", context.grad_norms[k].clone(),
                           global_step=self.step_count)
    alg.loss_fn.call_count += 1
    self.step_count += 1
    context.next_k = k + 1
    # between now and the epoch's last minibatch the parameters are AHEAD of what the reference
    # would show: the policy / model refuse to be read meanwhile (see _epoch_fast_path)
    engine.open_epoch = context if context.next_k < context.num_minibatches else None
    return context.loss_scalars[k]

  def step(self, alg, data):
    fast = self._epoch_fast_path(alg, data)
    if fast is not None:
      return self._step_epoch(alg, *fast)
    single = self._single_update_context(alg, data)
    if single is not None:
      return self._step_epoch(alg, single, 0)
    native = None
    if isinstance(self.optimizer, _FlatOptimizer) and type(alg).loss is Alg.loss:
      native = getattr(alg.loss_fn, "evaluate_native", None)
    if native is not None:
      # same launches as loss -> backward below, without building a one-node autograd graph
      # per update (the head gradient already sits in the engine when the loss returns)
      loss, backward_fn = native(data)
      self.optimizer.zero_grad()
      if distributed.sharded() and hasattr(alg.model.engine, "tail_offset"):
        backward_fn(None, on_part=self.optimizer.reduce_part)  # all-reduce overlapped with backward
      else:
        backward_fn(None)
    else:
      loss = alg.loss(data)
      self.optimizer.zero_grad()
      loss.backward()
    self.preprocess_gradients(alg.model.parameters(), f"{alg.name}/grad_norm")
    for anneal in self.anneals:
      if summary.should_record():
        anneal.summarize(alg.runner.step_count)
      anneal.step_to(alg.runner.step_count)
    self.optimizer.step()
    if getattr(self, "_pending_norm_tag", None) is not None:
      # this step's pre-clip gradient norm (the reference logs it before optimizer.step(),
      # alg/common.py:61-64: same value, same global_step); a copy, the buffer is reused
      summary.add_scalar(self._pending_norm_tag, self.optimizer.grad_norm[0].clone(),
                         global_step=self.step_count)
      self._pending_norm_tag = None
    if not isinstance(self.optimizer, _FlatOptimizer):
      engine = getattr(alg.model, "engine", None)
      if engine is not None:
        engine.mark_dirty()
    self.step_count += 1
    return loss


class Alg:
  """Generic learning algorithm specified by its loss function (alg/common.py:81-106)."""
  def __init__(self, runner, trainer, loss_fn, name=None):
    self.runner = runner
    self.model = self.runner.policy.model
    self.trainer = trainer
    self.loss_fn = loss_fn
    if name is None:
      name = self.__class__.__name__.lower()
    self.name = name

  def loss(self, data):
    return self.loss_fn(data)

  def step(self, data):
    return self.trainer.step(self, data)

  def learn(self):
    try:
      from tqdm import tqdm  # pylint: disable=import-outside-toplevel
    except ImportError:
      tqdm = None
    if tqdm is None:
      for data in self.runner.run():
        self.step(data)
      return
    with tqdm(total=len(self.runner)) as pbar:
      for data in self.runner.run():
        pbar.update(self.runner.step_count - pbar.n)
        self.step(data)
