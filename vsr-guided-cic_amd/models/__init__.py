"""Drop-in package for the reference's `models` (models/__init__.py:1-4): put vsr-guided-cic_amd/ on sys.path and
`from models import ControllableCaptioningModel` / `from models import SinkhornNet, S_SSP` (coco_scripts/train.py:6,
coco_scripts/eval_coco.py:6,10) resolve here, so the reference's scripts stay the callers they are.  The two ordering
models are the inference side only (generate / forward as eval_coco.py calls them); their training scripts are out of scope.
"""
from .CaptioningModel import CaptioningModel as _CaptioningModel
from .controllable_captioning import ControllableCaptioningModel, set_default_compute_dtype
from .sinkhorn_network import SinkhornNet
from .sort_model import S_SSP

__all__ = ["ControllableCaptioningModel", "_CaptioningModel", "SinkhornNet", "S_SSP", "set_default_compute_dtype"]
