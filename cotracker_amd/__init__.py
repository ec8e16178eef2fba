"""Importable alias for the ``co-tracker_amd/`` package directory.

The layout contract names the package directory ``co-tracker_amd`` (with a
hyphen), which Python cannot import by name.  This shim makes
``import cotracker_amd`` resolve every submodule from that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "co-tracker_amd")
__path__.insert(0, _real)  # submodules (cotracker_amd.predictor, ...) come from co-tracker_amd/
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
