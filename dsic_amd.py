"""Import alias: ``import dsic_amd`` loads the package that lives in
``domain-specific-image-compression_amd/`` (a directory name Python cannot
import directly because of the hyphens)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "domain-specific-image-compression_amd")
_spec = importlib.util.spec_from_file_location(
    "dsic_amd", os.path.join(_dir, "__init__.py"),
    submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dsic_amd"] = _mod
_spec.loader.exec_module(_mod)
