"""Import shim: makes the package that lives in `temporalgps.jl_amd/` importable as `temporalgps_jl_amd`
(a directory name with a dot cannot be named in an `import` statement)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "temporalgps.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
