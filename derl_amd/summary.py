"""Summary facade with derl's call surface (derl/summary.py:13-64): ``should_record``,
``start_recording`` / ``stop_recording`` / ``set_recording``, ``set_writer`` / ``make_writer``
and ``add_scalar``.  tensorboard is optional: without a writer, scalars are kept in
``last_scalars`` (tag -> (value, global_step)) so callers and tests can still read them.
Values may be 0-dim device tensors; they are only converted when a writer consumes them."""

_state = {"record": False, "writer": None}
last_scalars = {}


def should_record():
  return _state["record"]


def start_recording():
  _state["record"] = True


def stop_recording():
  _state["record"] = False


def set_recording(flag):
  _state["record"] = bool(flag)


def set_writer(writer):
  _state["writer"] = writer


def make_writer(*args, **kwargs):
  """Creates a tensorboard SummaryWriter when tensorboard is installed; otherwise keeps
  the in-memory store only (summary.py:41-43)."""
  try:
    from torch.utils.tensorboard import SummaryWriter  # pylint: disable=import-outside-toplevel
  except Exception:  # tensorboard missing
    set_writer(None)
    return None
  writer = SummaryWriter(*args, **kwargs)
  set_writer(writer)
  return writer


def add_scalar(tag, value, global_step=None, **kwargs):
  last_scalars[tag] = (value, global_step)
  writer = _state["writer"]
  if writer is not None:
    writer.add_scalar(tag, float(value), global_step=global_step, **kwargs)
