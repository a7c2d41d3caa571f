"""Import shim: exposes the package directory `shineon-virtual-tryon_amd/` (not a valid Python
identifier) as the importable package `shineon_virtual_tryon_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shineon-virtual-tryon_amd")
_spec = importlib.util.spec_from_file_location(
    "shineon_virtual_tryon_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["shineon_virtual_tryon_amd"] = _mod
_spec.loader.exec_module(_mod)
