"""Import alias: ``import tds_amd`` == package directory ``tiny-differentiable-simulator_amd``
(whose name, mandated by the project layout, is not a valid Python identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("tiny-differentiable-simulator_amd")
sys.modules[__name__] = _pkg
