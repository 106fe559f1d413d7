"""On-policy runner wrappers (derl/runners/onpolicy.py:11-82)."""
import ctypes
import math

import numpy as np
import torch

from .. import _lib

from .env_runner import EnvRunner, RunnerWrapper
from .summary import PeriodicSummaries
from .trajectory_transforms import GAE, MergeTimeBatch, NormalizeAdvantages
from .. import ops
from ..models import GatheredRows

# arrays whose rows are at least this big are referenced by index instead of copied
LAZY_ROW_BYTES = 4096


class EpochContext:
  """What the minibatches of one epoch share: the epoch's permuted small arrays (minibatch k is
  rows [k * mbsize, (k + 1) * mbsize) of each) -- enough for a trainer to enqueue EVERY update of
  the epoch from one native call when its first minibatch is stepped (Trainer.step,
  dx_mlp_ppo_epoch), after which ``losses`` (one row of loss terms per minibatch) and
  ``normalized`` (every minibatch's normalised advantages) are filled in."""
  STATE_KEY = "epoch"

  def __init__(self, shuffled, sample_size, mbsize, order_dev=None, lazy=None):
    self.shuffled, self.sample_size, self.mbsize = shuffled, sample_size, mbsize
    self.num_minibatches = -(-sample_size // mbsize)
    # arrays with big rows (frames) are not permuted: minibatch k reads rows
    # order_dev[k * mbsize : (k + 1) * mbsize] of lazy[key] (GatheredRows); order_dev None = identity
    self.order_dev, self.lazy = order_dev, lazy or {}
    self.consumed = False
    self.next_k = 0         # the minibatch Trainer.step hands out next (the epoch is stepped in order, once)
    self.losses = None      # (minibatches, 8) loss terms, filled by the native epoch
    self.grad_norms = None  # (minibatches,) pre-clip gradient norms when summaries are recorded
    self.stats_ready = None  # (minibatches, 3) global advantage statistics of a sharded run
    self.verified = None     # set by the trainer once minibatch 0 passed its pointer comparison
    self.loss_rows = self.loss_scalars = None  # per-minibatch views of ``losses``
    # Written by NormalizeAdvantages when it normalises the epoch's FIRST minibatch: its epsilon
    # and the tensor it produced.  That is how the transform opts in to the native epoch: the
    # trainer normalises the remaining minibatches the same way only if these are set and the
    # minibatch it is handed still carries that very tensor; a pipeline without the transform
    # trains natively on the raw advantages, anything else goes update by update.
    self.norm_eps = None
    self.norm_first = None
    self.normalized = None  # every minibatch's normalised advantages, filled by the native epoch
    # another epoch over the same rollout follows this one (IterateWithMinibatches): a native epoch may then leave
    # the mirrors only the rollout reads stale at its end (dx_cnn_epoch.more_epochs)
    self.more_epochs = False


class LazyMinibatch(dict):
  """A minibatch dict whose values are cut out of the epoch's arrays on first access.

  ``IterateWithMinibatches`` yields 320 of these per rollout at BASELINE config 3; building seven
  tensor views each cost more host time than the GPU spends on the update.  Reads behave like a
  plain dict (a pending value is produced by its thunk the first time it is looked at; iteration,
  ``items()`` ... produce them all).  ``touched`` records any assignment or deletion from outside
  -- a transform that edits the minibatch -- which sends the trainer down its checked path;
  ``lazy_set`` is for transforms that supply a derived value without forcing it."""
  __slots__ = ("_pending", "touched", "rows", "epoch")

  def __init__(self, pending, rows=None, epoch=None):
    super().__init__()
    self._pending = pending
    self.touched = False
    self.rows = rows    # (start, stop) inside the epoch's arrays
    self.epoch = epoch  # (EpochContext, minibatch number)

  def _force(self, key):
    thunk = self._pending.pop(key, None)
    if thunk is not None:
      dict.__setitem__(self, key, thunk())

  def _force_all(self):
    for key in list(self._pending):
      self._force(key)

  def lazy_set(self, key, thunk):
    dict.pop(self, key, None)
    self._pending[key] = thunk

  def __getitem__(self, key):
    if key in self._pending:
      self._force(key)
    return dict.__getitem__(self, key)

  def get(self, key, default=None):
    if key in self._pending:
      self._force(key)
    return dict.get(self, key, default)

  def __contains__(self, key):
    return key in self._pending or dict.__contains__(self, key)

  def __setitem__(self, key, value):
    self._pending.pop(key, None)
    self.touched = True
    dict.__setitem__(self, key, value)

  def __delitem__(self, key):
    self.touched = True
    if self._pending.pop(key, None) is None:
      dict.__delitem__(self, key)

  def pop(self, key, *default):
    self.touched = True
    self._force(key)
    return dict.pop(self, key, *default)

  def setdefault(self, key, default=None):
    if key not in self:
      self[key] = default
    return self[key]

  def update(self, *args, **kwargs):
    for key, value in dict(*args, **kwargs).items():
      self[key] = value

  def __iter__(self):
    self._force_all()
    return dict.__iter__(self)

  def __len__(self):
    return dict.__len__(self) + len(self._pending)

  def keys(self):
    self._force_all()
    return dict.keys(self)

  def values(self):
    self._force_all()
    return dict.values(self)

  def items(self):
    self._force_all()
    return dict.items(self)

  def copy(self):
    self._force_all()
    return dict(self)

  def __repr__(self):
    self._force_all()
    return dict.__repr__(self)


class TransformInteractions(RunnerWrapper):
  """Transforms interactions by applying a list of callables (onpolicy.py:11-30).  Lists
  from the generic runner are stacked (``np.asarray`` / ``torch.stack``); the
  device-resident runner already yields whole ``(T, N, ...)`` buffers."""
  def __init__(self, runner, transforms=None, asarray=True):
    super().__init__(runner)
    self.transforms = transforms or []
    self.asarray = asarray

  def run(self, obs=None):
    for interactions in self.runner.run(obs=obs):
      if self.asarray and not isinstance(interactions, LazyMinibatch):  # (a minibatch holds no per-step lists)
        for key, val in interactions.items():
          if key == "state" or not isinstance(val, list):
            continue
          try:
            if val and isinstance(val[0], torch.Tensor):
              interactions[key] = torch.stack(val)
            else:
              interactions[key] = np.asarray(val)
          except ValueError:
            raise ValueError(
                f"cannot convert value under key '{key}' to np.ndarray")
      for transform in self.transforms:
        transform(interactions)
      yield interactions


class IterateWithMinibatches(RunnerWrapper):
  """Iterates over interactions with minibatches for a given number of epochs
  (onpolicy.py:33-62).

  The reference shuffles every array in place before each epoch (the shuffles compose) and
  then copies minibatch slices.  Here the composed permutation is kept as an index vector:
  small per-sample arrays are gathered on the device, arrays with big rows (frames) are
  yielded as ``GatheredRows(base, index)`` and gathered inside the conv loader -- same
  samples in the same order (``np.random.permutation`` stream as the reference), no copies
  of the frame buffers."""
  def __init__(self, runner, num_epochs=3, num_minibatches=4, shuffle_before_epoch=True, prepare=None):
    super().__init__(runner)
    self.num_epochs = num_epochs
    self.num_minibatches = num_minibatches
    self.shuffle_before_epoch = shuffle_before_epoch
    # prepare(interactions, orders_dev, mbsize) -> None or f(epoch, k) -> dict merged into the
    # minibatch's "state": work over ALL minibatches of a rollout once their order is known
    # (NormalizeAdvantages.prepare: one all-reduce per rollout instead of one per minibatch)
    self.prepare = prepare
    # three pinned staging buffers used in turn (the current rollout's, the one being drawn ahead, and
    # one whose upload may still be in flight): a buffer is rewritten only after the upload that last
    # read it has completed
    self._pinned, self._pinned_event, self._pinned_turn = [None] * 3, [None] * 3, 0
    self._worker = None  # draws the next rollout's permutations ahead (see _prefetch_allowed)

  @staticmethod
  def _gather_epoch(interactions, order_dev):
    """Device arrays with small rows, permuted for a whole epoch by ONE gather launch (per 16
    arrays): the epoch's minibatches are then contiguous slices, exactly the reference's
    shuffle-in-place-then-slice (onpolicy.py:44-62) without touching the rollout buffers."""
    if order_dev is None:
      return {}
    small = [key for key, val in interactions.items()
             if key != "state" and isinstance(val, torch.Tensor) and val.is_cuda
             and val.element_size() * math.prod(val.shape[1:]) < LAZY_ROW_BYTES]
    gathered = ops.gather_rows_multi([interactions[key].contiguous() for key in small], order_dev)
    return dict(zip(small, gathered))

  @staticmethod
  def _select_all(interactions, shuffled, start, stop, order_dev, order):
    """One minibatch of every array: slices of the epoch's permuted small arrays, lazy index
    references for arrays with big rows (frames), fancy indexing for host arrays -- each produced
    when it is first looked at (LazyMinibatch)."""
    pending = {}
    plain = {}
    for key, val in interactions.items():
      if key in shuffled:
        pending[key] = lambda a=shuffled[key]: a[start:stop]
      elif key == "state":
        plain[key] = val
      elif isinstance(val, torch.Tensor) and val.is_cuda:
        pending[key] = lambda v=val: GatheredRows(v, order_dev[start:stop])
      elif isinstance(val, torch.Tensor):
        # (host arrays: the slice of `order` is taken NOW -- it may be a view of a pinned staging buffer that
        # the permutation worker rewrites two rollouts later, and a minibatch may be kept that long)
        pending[key] = lambda v=val, rows=order[start:stop].astype(np.int64): v[torch.from_numpy(rows)]
      elif isinstance(val, np.ndarray):
        pending[key] = lambda v=val, rows=order[start:stop].copy(): v[rows]
      else:
        plain[key] = val
    out = LazyMinibatch(pending, rows=(start, stop))
    for key, val in plain.items():
      dict.__setitem__(out, key, val)
    return out

  def _draw_host(self, sample_size, pinned=False, state=None):
    """The composed permutations of all epochs as an (epochs, samples) int32 array: the reference's
    per-epoch ``np.random.permutation`` draws (nothing else consumes np.random between them, so the
    stream is identical), made by the native library from NumPy's own generator state
    (dx_host_compose_permutations: bit-for-bit the NumPy result, and the call does not hold the
    GIL, so a worker thread can really run beside the training loop).  ``pinned``: into the next
    pinned staging buffer (a torch tensor; its previous upload is waited for first).

    ``state`` = None: drawn from (and advancing) the global ``np.random`` generator.  Otherwise a
    ``np.random.get_state()`` snapshot to draw from WITHOUT touching the global generator (a draw
    made ahead on the worker thread): the generator state after the draw is returned as the fourth
    element and ``run`` installs it only if the global generator still is at the snapshot."""
    shape = (self.num_epochs, sample_size)
    if pinned:
      turn = self._pinned_turn
      self._pinned_turn = (turn + 1) % len(self._pinned)
      if self._pinned[turn] is None or tuple(self._pinned[turn].shape) != shape:
        self._pinned[turn] = torch.empty(shape, dtype=torch.int32).pin_memory()
      elif self._pinned_event[turn] is not None:
        self._pinned_event[turn].synchronize()  # the upload that last read this buffer is done
      staging = self._pinned[turn]
      out = staging.numpy()
    else:
      staging, turn = None, None
      out = np.empty(shape, np.int32)
    try:
      lib = _lib.load()
    except _lib.NativeError:
      lib = None  # host-only use without the built library: NumPy's own calls (same stream, same result)
    after = None
    if lib is not None:
      name, key, pos, has_gauss, cached = np.random.get_state() if state is None else state
      key = np.ascontiguousarray(key, dtype=np.uint32).copy()
      position = ctypes.c_int(int(pos))
      _lib.check(lib.dx_host_compose_permutations(
          key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(position), int(sample_size), int(self.num_epochs),
          int(bool(self.shuffle_before_epoch)), out.ctypes.data_as(ctypes.c_void_p)), "dx_host_compose_permutations")
      after = (name, key, position.value, has_gauss, cached)
      if state is None:
        np.random.set_state(after)
    else:
      source = np.random
      if state is not None:
        source = np.random.RandomState()
        source.set_state(state)
      order = np.arange(sample_size)
      for epoch in range(self.num_epochs):
        if self.shuffle_before_epoch:
          order = order[source.permutation(sample_size)]
        out[epoch] = order
      if state is not None:
        after = source.get_state()
    return out, staging, turn, after

  @staticmethod
  def _same_generator_state(a, b):
    return (a[0] == b[0] and int(a[2]) == int(b[2]) and int(a[3]) == int(b[3]) and float(a[4]) == float(b[4])
            and np.array_equal(a[1], b[1]))

  def _draw_orders(self, sample_size, device, drawn=None):
    """``_draw_host`` (unless the permutations were drawn ahead) and, for device data, their upload
    with ONE pinned non-blocking copy: a pageable H2D copy per epoch would drain the stream."""
    if drawn is None:
      drawn = self._draw_host(sample_size, pinned=device is not None)
    orders, staging, turn = drawn[:3]
    orders_dev = None
    if device is not None:
      if staging is None:  # drawn without a staging buffer (cannot happen on the prefetch path)
        staging = torch.from_numpy(orders).pin_memory()
      orders_dev = staging.to(device, non_blocking=True)
      if turn is not None:
        self._pinned_event[turn] = torch.cuda.Event()
        self._pinned_event[turn].record(torch.cuda.current_stream(device))
    return sample_size, device, orders, orders_dev

  def _prefetch_allowed(self):
    """The permutations of the NEXT rollout may be drawn ahead -- on a worker thread, while this
    thread enqueues the current rollout's updates (numpy's permutation releases the GIL; at config
    3's 131,072 samples x 10 epochs the draws are 11 ms of a 25 ms iteration) -- only if that
    cannot reorder the global np.random stream against the reference's: the runner is
    device-resident AND the env declares ``host_rng_free`` (the built-in synthetic envs; a host
    env behind HostEnvBridge / ParallelEnvBatch may draw from np.random inside ``step``, and then
    the permutations must be drawn after the rollout, like the reference does)."""
    base = getattr(self.runner, "unwrapped", self.runner)
    check = getattr(base, "_device_resident", None)
    exhausted = getattr(base, "is_exhausted", None)
    if check is None or not check() or exhausted is None or exhausted():
      return False
    return bool(getattr(getattr(base, "env", None), "host_rng_free", False))

  def run(self, obs=None):
    inner = self.runner.run(obs=obs)
    # (what it was drawn for, the np.random snapshot it was drawn from, future of the NEXT rollout's
    # permutations).  The worker never touches the global generator: a draw made ahead counts only
    # if np.random still is at that snapshot when the rollout it was made for arrives (then the
    # generator is moved past the draw, as if it had happened now); if anything else drew from or
    # reseeded np.random meanwhile, the draw is dropped and made now -- the reference's order
    # (onpolicy.py:44-62: the permutations are drawn after the rollout).
    ahead = None
    while True:
      try:
        interactions = next(inner)
      except StopIteration:
        if ahead is not None:
          ahead[2].result()  # never leave a draw running behind the caller's back (its result is dropped)
        return
      sample_size = interactions["observations"].shape[0]
      device = None
      for val in interactions.values():
        if isinstance(val, torch.Tensor) and val.is_cuda:
          device = val.device
          break
      drawn = None
      if ahead is not None:
        drawn_for, snapshot, future = ahead
        drawn = future.result()
        ahead = None
        if (drawn_for != (sample_size, device is not None) or drawn[3] is None
            or not self._same_generator_state(np.random.get_state(), snapshot)):
          drawn = None
        else:
          np.random.set_state(drawn[3])
      _, _, orders, orders_dev = self._draw_orders(sample_size, device, drawn)
      if self._prefetch_allowed():
        if self._worker is None:
          from concurrent.futures import ThreadPoolExecutor  # pylint: disable=import-outside-toplevel
          self._worker = ThreadPoolExecutor(1)
        snapshot = np.random.get_state()
        ahead = ((sample_size, device is not None), snapshot,
                 self._worker.submit(self._draw_host, sample_size, device is not None, snapshot))
      mbsize = sample_size // self.num_minibatches
      extras = None
      if self.prepare is not None and orders_dev is not None:
        extras = self.prepare(interactions, orders_dev, mbsize)
      for epoch, order in enumerate(orders):
        order_dev = orders_dev[epoch] if orders_dev is not None else None
        shuffled = self._gather_epoch(interactions, order_dev)
        context = None
        if shuffled:
          lazy = {key: val for key, val in interactions.items()
                  if key != "state" and key not in shuffled and isinstance(val, torch.Tensor) and val.is_cuda}
          context = EpochContext(shuffled, sample_size, mbsize, order_dev, lazy)
          context.more_epochs = epoch + 1 < len(orders)
          if extras is not None:
            context.stats_ready = getattr(extras, "epoch_stats", lambda _: None)(epoch)
        for start in range(0, sample_size, mbsize):
          stop = min(start + mbsize, sample_size)
          minibatch = self._select_all(interactions, shuffled, start, stop, order_dev, order)
          state = dict.get(minibatch, "state")
          if extras is not None:
            state = dict(state or {}, **extras(epoch, start // mbsize))
          if context is not None:
            minibatch.epoch = (context, start // mbsize)
            state = dict(state or {}, **{EpochContext.STATE_KEY: minibatch.epoch})
          if state is not None:
            dict.__setitem__(minibatch, "state", state)  # (the iterator's own entry: not an outside edit)
          yield minibatch


def ppo_runner_wrap(runner, gamma=0.99, lambda_=0.95, num_epochs=3, num_minibatches=4):
  """Wraps given runner for PPO training (onpolicy.py:65-75)."""
  env, policy = runner.env, runner.policy
  transforms = [GAE(policy, gamma=gamma, lambda_=lambda_, normalize=False)]
  if not policy.is_recurrent() and getattr(env.unwrapped, "nenvs", None):
    transforms.append(MergeTimeBatch())
  runner = TransformInteractions(runner, transforms)
  normalize = NormalizeAdvantages()
  runner = IterateWithMinibatches(runner, num_epochs, num_minibatches, prepare=normalize.prepare)
  runner = TransformInteractions(runner, [normalize])
  return runner


def make_ppo_runner(env, policy, horizon, nsteps, nlogs=1e5, **wrap_kwargs):
  """Creates and wraps env runner for PPO training (onpolicy.py:78-82)."""
  runner = EnvRunner(env, policy, horizon, nsteps)
  runner = PeriodicSummaries.make_with_nlogs(runner, nlogs)
  return ppo_runner_wrap(runner, **wrap_kwargs)
