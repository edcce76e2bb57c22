"""Import shim: the product package lives in the directory ``linux-fg_amd/`` (a name Python
cannot import directly because of the hyphen).  ``import linux_fg_amd`` loads that directory
as the package ``linux_fg_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linux-fg_amd")
_spec = importlib.util.spec_from_file_location(
    "linux_fg_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["linux_fg_amd"] = _mod
_spec.loader.exec_module(_mod)
