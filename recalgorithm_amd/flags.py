"""tf.app.flags look-alike (algorithm/DeepFM/deepfm.py:14-41): DEFINE_* + a lazily parsed
FLAGS object, so `python deepfm.py --embedding_dim=16 --batch_norm=False` works unchanged."""
from __future__ import annotations

import argparse
import sys


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "t", "yes", "y"):
        return True
    if v.lower() in ("0", "false", "f", "no", "n"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v}")


class _Flags:
    def __init__(self):
        object.__setattr__(self, "_parser", argparse.ArgumentParser(allow_abbrev=False))
        object.__setattr__(self, "_values", None)
        object.__setattr__(self, "_overrides", {})

    def _define(self, name, default, help, type_):
        try:
            if type_ is bool:
                self._parser.add_argument(f"--{name}", default=default, type=_str2bool, nargs="?",
                                          const=True, help=help)
            else:
                self._parser.add_argument(f"--{name}", default=default, type=type_, help=help)
        except argparse.ArgumentError:
            pass  # re-definition by a second model script in the same process

    def _parse(self, argv=None):
        ns, rest = self._parser.parse_known_args(sys.argv[1:] if argv is None else argv)
        object.__setattr__(self, "_values", vars(ns))
        return rest

    def __getattr__(self, name):
        if name in self._overrides:
            return self._overrides[name]
        if self._values is None:
            self._parse()
        try:
            return self._values[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self._overrides[name] = value


FLAGS = _Flags()


def DEFINE_string(name, default, help=""):
    FLAGS._define(name, default, help, str)


def DEFINE_integer(name, default, help=""):
    FLAGS._define(name, default, help, int)


def DEFINE_float(name, default, help=""):
    FLAGS._define(name, default, help, float)


def DEFINE_boolean(name, default, help=""):
    FLAGS._define(name, default, help, bool)


DEFINE_bool = DEFINE_boolean


def DEFINE_enum(name, default, enum_values, help=""):
    """tf.app.flags.DEFINE_enum (fibinet.py:45): a string flag restricted to `enum_values`."""
    try:
        FLAGS._parser.add_argument(f"--{name}", default=default, type=str, choices=list(enum_values), help=help)
    except argparse.ArgumentError:
        pass


def run(main, argv=None):
    """tf.app.run(main)."""
    rest = FLAGS._parse(argv)
    sys.exit(main([sys.argv[0]] + rest))
