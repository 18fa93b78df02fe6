"""Importable alias of the ``adaptive-proximal-algorithms_amd`` package (whose directory
name contains hyphens): ``import adaprox_amd as apx``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("adaptive-proximal-algorithms_amd")
sys.modules[__name__] = _pkg
