from .registry import MONO
from .net import Baseline
from . import modules
from .losses import IoULoss, BDLoss

__all__ = ["MONO", "Baseline", "modules", "IoULoss", "BDLoss"]
