"""Import shim: the package lives in the directory ``loco-edit_amd/`` (the name
the project brief fixes), which is not a legal Python identifier.  Importing
``loco_edit_amd`` loads that directory as a regular package under this name.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "loco-edit_amd")
_spec = importlib.util.spec_from_file_location(
    "loco_edit_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["loco_edit_amd"] = _mod
_spec.loader.exec_module(_mod)
