"""Import shim: exposes the package directory ``video-subtitle-remover_amd/`` (named after the
project, not a Python identifier) as the importable package ``vsr_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-subtitle-remover_amd")
_spec = importlib.util.spec_from_file_location("vsr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vsr_amd"] = _mod
_spec.loader.exec_module(_mod)
