"""Importable alias: the package directory is named `physics-based-ray-tracing_amd` (not a valid
Python identifier), so `import pbrt_amd` loads it through importlib and stands in for it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("physics-based-ray-tracing_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("pbrt_amd", _pkg)
