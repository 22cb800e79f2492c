"""Importable alias of the ``3dahv_amd`` package (whose name is not a Python identifier)."""
import importlib as _importlib
import sys as _sys

_pkg = _importlib.import_module("3dahv_amd")
_sys.modules[__name__] = _pkg
