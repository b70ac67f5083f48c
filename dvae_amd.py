"""Import shim: the package directory `disentangle-vae-for-vc_amd/` is not a valid Python identifier,
so `import dvae_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "disentangle-vae-for-vc_amd")
_spec = importlib.util.spec_from_file_location("dvae_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dvae_amd"] = _mod
_spec.loader.exec_module(_mod)
