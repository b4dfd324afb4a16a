"""Importable alias of the `xiangqi-alphazero_amd/` package directory (a hyphen is not a valid
identifier).  `import xiangqi_alphazero_amd` / `from xiangqi_alphazero_amd import engine` work as
if the directory were named with an underscore."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "xiangqi-alphazero_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _os, _f
