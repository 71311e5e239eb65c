"""Re-export of the drop-in classes for a reference checkout that keeps its own `models` package for the
out-of-scope ordering models."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    name = "vsrcap_dropin_models"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_here, "models", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(_here, "models")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_m = _load()
ControllableCaptioningModel = _m.ControllableCaptioningModel
_CaptioningModel = _m._CaptioningModel
