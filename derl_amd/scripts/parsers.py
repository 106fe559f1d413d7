"""Argument-parsing helpers with the reference's call surface (derl/scripts/parsers.py)."""
import argparse
import os

from ..env import is_atari_id, is_mujoco_id


def get_simple_parser(add_env_id=True, add_logdir=True, nlogs=1e5):
  parser = argparse.ArgumentParser()
  if add_env_id:
    parser.add_argument("--env-id", required=True)
  if add_logdir:
    parser.add_argument("--logdir", required=True)
    parser.add_argument("--nlogs", type=float, default=nlogs)
  return parser


def get_defaults_parser(defaults, base_parser=None):
  """Adds a dictionary of defaults to a parser: dict values are add_argument kwargs, any
  other value gives ``--key`` with that value's type and default (parsers.py:21-30)."""
  if base_parser is None:
    base_parser = argparse.ArgumentParser()
  for key, val in defaults.items():
    if isinstance(val, dict):
      base_parser.add_argument(f"--{key}", **val)
    else:
      base_parser.add_argument(f"--{key}", type=type(val), default=val)
  return base_parser


def get_parser(defaults, add_env_id=True, add_logdir=True, nlogs=1e5):
  return get_defaults_parser(defaults, get_simple_parser(add_env_id, add_logdir, nlogs))


def log_args(args, logdir=None):
  """Writes the namespace to ``logdir/args.txt`` (parsers.py:39-48)."""
  if logdir is None:
    logdir = args.logdir
  os.makedirs(logdir, exist_ok=True)
  with open(os.path.join(logdir, "args.txt"), "w") as argsfile:
    for key, val in vars(args).items():
      argsfile.write(f"{key}: {val}\n")
  return args


def get_args_from_defaults(defaults, env_id=True, logdir=True, nlogs=1e5, call_log_args=None):
  if call_log_args and not logdir:
    raise ValueError("logdir must be True when call_log_args is True")
  args = get_parser(defaults, env_id, logdir, nlogs).parse_args()
  if call_log_args or call_log_args is None and logdir:
    log_args(args)
  return args


def get_args(atari_defaults=None, mujoco_defaults=None, args=None, logdir=True, nlogs=1e5,
             call_log_args=True):
  """Arguments from the defaults chosen by the env id; other envs need ``--defaults``
  (parsers.py:63-101)."""
  if atari_defaults is None and mujoco_defaults is None:
    raise ValueError("atari_defaults and mujoco_defaults cannot both be None")
  if call_log_args and not logdir:
    raise ValueError("logdir must be True when call_log_args is True")
  env_type_defaults = dict(atari=atari_defaults, mujoco=mujoco_defaults)
  namespace, unknown_args = get_simple_parser(add_logdir=logdir, nlogs=nlogs).parse_known_args(args)
  if is_atari_id(namespace.env_id):
    env_type = "atari"
  elif is_mujoco_id(namespace.env_id):
    env_type = "mujoco"
  else:
    defaults_parser = argparse.ArgumentParser()
    choices = set(env_type_defaults)
    defaults_parser.add_argument("--defaults", choices=choices)
    namespace, unknown_args = defaults_parser.parse_known_args(unknown_args, namespace)
    if namespace.defaults is None:
      defaults_parser.error(
          f"{namespace.env_id} is neither an atari nor mujoco env, "
          f"please specify which defaults to choose by using --defaults {choices}")
    env_type = namespace.defaults
  defaults = env_type_defaults[env_type]
  if defaults is None:
    raise ValueError(f"cannot run env {namespace.env_id} because {env_type} defaults are "
                     f"not specified; does this algorithm support {env_type} envs?")
  namespace = get_parser(defaults, add_env_id=False, add_logdir=False).parse_args(
      unknown_args, namespace)
  if call_log_args:
    log_args(namespace)
  return namespace
