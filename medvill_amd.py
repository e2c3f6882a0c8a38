"""Import shim: the package directory is named `multi-modality-self-supervision_amd`
(not a valid Python identifier), so `import medvill_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-modality-self-supervision_amd")
_spec = importlib.util.spec_from_file_location("medvill_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["medvill_amd"] = _mod
_spec.loader.exec_module(_mod)
