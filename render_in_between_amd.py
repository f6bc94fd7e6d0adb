"""Import shim: the package directory is ``render-in-between_amd/`` (the name
the repo layout mandates), which is not a valid Python identifier.  Importing
``render_in_between_amd`` executes this file, which loads the real package
from the hyphenated directory under the importable name and replaces itself
in ``sys.modules``."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "render-in-between_amd")
_spec = _u.spec_from_file_location(
    "render_in_between_amd", _os.path.join(_dir, "__init__.py"),
    submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["render_in_between_amd"] = _mod
_spec.loader.exec_module(_mod)
