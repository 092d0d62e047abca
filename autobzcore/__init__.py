"""Import shim: the product package lives in the directory `autobzcore.jl_amd/` (a name Python's
import statement cannot spell), so `import autobzcore.jl_amd` is wired up here."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autobzcore.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _name, os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = sys.modules[_name]
