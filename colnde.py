"""Import alias: `import colnde` loads the package directory `climateparameterizations.jl_amd/`
(whose name, fixed by the repo layout, is not a valid Python identifier)."""
import importlib.util as _u
import os as _os
import sys as _sys

_path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "climateparameterizations.jl_amd")
_spec = _u.spec_from_file_location("colnde", _os.path.join(_path, "__init__.py"),
                                   submodule_search_locations=[_path])
_mod = _u.module_from_spec(_spec)
_sys.modules["colnde"] = _mod
_spec.loader.exec_module(_mod)
