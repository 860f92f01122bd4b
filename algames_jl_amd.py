"""Import shim: the package directory is `algames.jl_amd/` (not a valid Python identifier), so this
module loads it under the importable name `algames_jl_amd`."""
import importlib.util
import os
import sys

_pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "algames.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "algames_jl_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["algames_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
