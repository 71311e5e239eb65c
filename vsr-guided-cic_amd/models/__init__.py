"""Drop-in package for the reference's `models` (models/__init__.py:1-4): put vsr-guided-cic_amd/ on
sys.path and `from models import ControllableCaptioningModel` resolves here, so coco_scripts/train.py and
coco_scripts/eval_coco.py stay the callers they are.  S_SSP / SinkhornNet (the ordering models that run
BEFORE this hot path) are out of scope and are not provided.
"""
from .CaptioningModel import CaptioningModel as _CaptioningModel
from .controllable_captioning import ControllableCaptioningModel

__all__ = ["ControllableCaptioningModel", "_CaptioningModel"]
