from .registry import MONO
from .net import Baseline
from . import modules

__all__ = ["MONO", "Baseline", "modules"]
