"""Importable alias of the `doppel-speller_amd/` package directory (a hyphen cannot appear in a module name).

`import doppel_speller_amd` executes doppel-speller_amd/__init__.py with this package's name, so
`doppel_speller_amd.match_maker` etc. resolve to the files under `doppel-speller_amd/`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "doppel-speller_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _handle:
    exec(compile(_handle.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _handle
