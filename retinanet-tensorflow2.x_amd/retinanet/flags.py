"""A small stand-in for the absl.flags surface the reference's entry points use (absl is not a dependency of this
build): `--name=value`, `--name value`, boolean `--name` / `--noname` / `--name=true|false|1|0`, unknown flags are an
error, required flags are checked — the accepted command lines are the reference's (retinanet/__main__.py:15-69,
retinanet/export.py:19-102)."""
from __future__ import annotations

import sys


class FlagError(SystemExit):
    pass


class FlagSet:
    def __init__(self):
        self._defs = {}
        self._values = {}

    def define(self, kind, name, default=None, help="", required=False, enum_values=None):
        self._defs[name] = dict(kind=kind, default=default, help=help, required=required, enum=enum_values)
        self._values[name] = default

    def DEFINE_string(self, name, default=None, help="", required=False):
        self.define("string", name, default, help, required)

    def DEFINE_integer(self, name, default=None, help="", required=False):
        self.define("integer", name, default, help, required)

    def DEFINE_boolean(self, name, default=False, help="", required=False):
        self.define("boolean", name, bool(default), help, required)

    def DEFINE_enum(self, name, default=None, enum_values=(), help="", required=False):
        self.define("enum", name, default, help, required, list(enum_values))

    def _convert(self, name, raw):
        d = self._defs[name]
        if d["kind"] == "integer":
            return int(raw)
        if d["kind"] == "boolean":
            if str(raw).lower() in ("1", "true", "t", "yes", "y"):
                return True
            if str(raw).lower() in ("0", "false", "f", "no", "n"):
                return False
            raise FlagError(f"flag --{name}: not a boolean: {raw!r}")
        if d["kind"] == "enum" and raw not in d["enum"]:
            raise FlagError(f"flag --{name}={raw}: value should be one of {d['enum']}")
        return raw

    def parse(self, argv=None):
        argv = list(sys.argv[1:] if argv is None else argv)
        i, seen = 0, set()
        while i < len(argv):
            arg = argv[i]
            i += 1
            if not arg.startswith("-"):
                raise FlagError(f"unexpected positional argument {arg!r}")
            body = arg.lstrip("-")
            name, eq, val = body.partition("=")
            if name not in self._defs and name.startswith("no") and name[2:] in self._defs \
                    and self._defs[name[2:]]["kind"] == "boolean" and not eq:
                self._values[name[2:]] = False
                seen.add(name[2:])
                continue
            if name not in self._defs:
                raise FlagError(f"Unknown command line flag '{name}'")
            if self._defs[name]["kind"] == "boolean" and not eq:
                self._values[name] = True
            else:
                if not eq:
                    if i >= len(argv):
                        raise FlagError(f"flag --{name} needs a value")
                    val = argv[i]
                    i += 1
                self._values[name] = self._convert(name, val)
            seen.add(name)
        missing = [n for n, d in self._defs.items() if d["required"] and self._values.get(n) is None]
        if missing:
            raise FlagError("Flag --{} must have a value other than None.".format(missing[0]))
        return self

    def __getattr__(self, name):
        values = self.__dict__.get("_values", {})
        if name in values:
            return values[name]
        raise AttributeError(name)
