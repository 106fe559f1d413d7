"""Command-line argument tables for the `derl` launcher and the factories.

Public call surface = derl/scripts/parsers.py (same function names, keyword arguments, return
types and error TYPES: ``ValueError`` for contradictory keyword arguments or a missing preset,
argparse's usage error / ``SystemExit`` for a bad command line).  The implementation is table
driven: every flag is an ``(option string, add_argument kwargs)`` pair, parsers are built by
installing such pairs, and the preset family of an env id is resolved by one helper."""
import argparse
import pathlib

from ..env import is_atari_id, is_mujoco_id

# preset families, in the order they are offered on the command line
_FAMILIES = (("atari", is_atari_id), ("mujoco", is_mujoco_id))


def _flag(name, value):
  """One table row.  A dict value is taken verbatim as ``add_argument`` keywords (how the
  factories declare store_true switches and typed-but-unset flags); anything else becomes a
  flag typed like its default."""
  spec = dict(value) if isinstance(value, dict) else dict(type=type(value), default=value)
  return f"--{name}", spec


def _install(parser, rows):
  for option, spec in rows:
    parser.add_argument(option, **spec)
  return parser


def _common_rows(with_env_id, with_logdir, nlogs):
  rows = []
  if with_env_id:
    rows.append(("--env-id", dict(required=True)))
  if with_logdir:
    rows += [("--logdir", dict(required=True)), _flag("nlogs", float(nlogs))]
  return rows


def _check_logging_request(call_log_args, logdir):
  if call_log_args and not logdir:
    raise ValueError("call_log_args=True needs the --logdir flag (logdir=True) to know where to write")


def get_simple_parser(add_env_id=True, add_logdir=True, nlogs=1e5):
  """Parser holding only ``--env-id`` and ``--logdir`` / ``--nlogs``."""
  return _install(argparse.ArgumentParser(), _common_rows(add_env_id, add_logdir, nlogs))


def get_defaults_parser(defaults, base_parser=None):
  """Installs one flag per entry of ``defaults`` on ``base_parser`` (a new parser if None)."""
  parser = argparse.ArgumentParser() if base_parser is None else base_parser
  return _install(parser, (_flag(name, value) for name, value in defaults.items()))


def get_parser(defaults, add_env_id=True, add_logdir=True, nlogs=1e5):
  """The common flags followed by the preset's flags."""
  return get_defaults_parser(defaults, get_simple_parser(add_env_id, add_logdir, nlogs))


def log_args(args, logdir=None):
  """Records the namespace as ``<logdir>/args.txt``, one ``name: value`` line per entry, and
  hands the namespace back."""
  target = pathlib.Path(args.logdir if logdir is None else logdir)
  target.mkdir(parents=True, exist_ok=True)
  lines = [f"{name}: {value}\n" for name, value in vars(args).items()]
  (target / "args.txt").write_text("".join(lines))
  return args


def get_args_from_defaults(defaults, env_id=True, logdir=True, nlogs=1e5, call_log_args=None):
  """Parses ``sys.argv`` against one preset; the arguments are logged when a logdir flag
  exists unless ``call_log_args`` says otherwise."""
  _check_logging_request(call_log_args, logdir)
  parsed = get_parser(defaults, env_id, logdir, nlogs).parse_args()
  wants_log = bool(logdir) if call_log_args is None else bool(call_log_args)
  return log_args(parsed) if wants_log else parsed


def _preset_family(parsed, leftover, presets):
  """Name of the preset family for ``parsed.env_id``: recognised ids pick their own; any
  other id must say ``--defaults <family>`` (consumed from ``leftover``; argparse usage error
  otherwise).  Returns (family, parsed, leftover)."""
  for family, recognises in _FAMILIES:
    if recognises(parsed.env_id):
      return family, parsed, leftover
  chooser = argparse.ArgumentParser()
  offered = sorted(presets)
  chooser.add_argument("--defaults", choices=offered)
  parsed, leftover = chooser.parse_known_args(leftover, parsed)
  if parsed.defaults is None:
    chooser.error(f"env id {parsed.env_id!r} belongs to no known preset family; "
                  f"pick one with --defaults, one of {offered}")
  return parsed.defaults, parsed, leftover


def get_args(atari_defaults=None, mujoco_defaults=None, args=None, logdir=True, nlogs=1e5,
             call_log_args=True):
  """Parses ``args`` (``sys.argv`` if None) in two passes: the common flags first -- the env id
  decides between the atari and the mujoco preset -- then the chosen preset's flags."""
  presets = dict(atari=atari_defaults, mujoco=mujoco_defaults)
  if all(table is None for table in presets.values()):
    raise ValueError("at least one of atari_defaults / mujoco_defaults has to be given")
  _check_logging_request(call_log_args, logdir)
  parsed, leftover = get_simple_parser(add_logdir=logdir, nlogs=nlogs).parse_known_args(args)
  family, parsed, leftover = _preset_family(parsed, leftover, presets)
  table = presets[family]
  if table is None:
    raise ValueError(f"env id {parsed.env_id!r} needs the {family} preset, which this "
                     "algorithm does not define")
  parsed = get_defaults_parser(table).parse_args(leftover, parsed)
  return log_args(parsed) if call_log_args else parsed
