"""Import shim: exposes the directory `k-wave-fluid-cuda_amd/` (not a valid Python identifier) as package `kwave_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "k-wave-fluid-cuda_amd")
_spec = importlib.util.spec_from_file_location(
    "kwave_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kwave_amd"] = _mod
_spec.loader.exec_module(_mod)
