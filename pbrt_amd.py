"""Importable alias: the package directory is named `physics-based-ray-tracing_amd` (not a valid
Python identifier), so `import pbrt_amd` loads it through importlib and stands in for it.  Submodules
are aliased too (`pbrt_amd.plugins` IS `physics-based-ray-tracing_amd.plugins`, one module object), so
classes keep one identity whichever name they were imported through."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "physics-based-ray-tracing_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith("pbrt_amd."):
            return importlib.util.spec_from_loader(fullname, self, origin=_REAL + fullname[len("pbrt_amd"):])
        return None

    def create_module(self, spec):
        return importlib.import_module(spec.origin)  # the real module object, registered under both names

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
sys.modules.setdefault("pbrt_amd", _pkg)
