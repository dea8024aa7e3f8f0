"""Import alias: ``import evc_amd`` loads the package that lives in the directory
``extreme-video-compression-with-prediction-using-pre-trainded-diffusion-models-_amd/``
(the mandated directory name is not a valid Python identifier).  The package is registered under the
single name ``evc_amd`` so there is exactly one copy of every submodule."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "extreme-video-compression-with-prediction-using-pre-trainded-diffusion-models-_amd")
_spec = importlib.util.spec_from_file_location("evc_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["evc_amd"] = _mod
_spec.loader.exec_module(_mod)
