"""Command-line plumbing (argument parsers shared by the `derl` launcher and the factories)."""
from .parsers import get_args, get_args_from_defaults, get_defaults_parser, get_parser, get_simple_parser, log_args
