"""Import shim: the package directory is named ``self-paced-contrastive-learning_amd`` (not a Python identifier),
so ``import spcl_amd`` loads it under this alias."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "self-paced-contrastive-learning_amd")
_spec = importlib.util.spec_from_file_location("spcl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["spcl_amd"] = _mod
_spec.loader.exec_module(_mod)
