"""Plugin registry, same shape as the reference's ``model/__init__.py`` (:3-4): one class per method,
looked up by attribute name from ``eval.py:21`` (``getattr(import_module("model"), config["model_name"])``).
The reference's line 5 (``from .unigeo_cam import ...``) imports a file that is not in its tree and is
not reproduced."""
from .depthcrafter import DepthCrafter
from .stablenormal import StableNormal

__all__ = ["DepthCrafter", "StableNormal"]
