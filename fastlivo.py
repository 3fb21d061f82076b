"""Loader for the `fast-livo_amd/` package.

The package directory name is fixed by the project layout and contains a hyphen, so it cannot be
imported by name.  `import fastlivo` registers it as the importable package `fast_livo_amd`.
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "fast-livo_amd")


def _load():
    if "fast_livo_amd" in sys.modules:
        return sys.modules["fast_livo_amd"]
    spec = importlib.util.spec_from_file_location(
        "fast_livo_amd", os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fast_livo_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = _load()
