from .parsers import (get_simple_parser, get_defaults_parser, get_parser, log_args,
                      get_args_from_defaults, get_args)
