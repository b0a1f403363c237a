from .registry import MONO
from .net import Baseline
from . import modules
from .losses import IoULoss, SoftDiceLoss, TverskyLoss, FocalLoss, BDLoss

__all__ = ["MONO", "Baseline", "modules", "IoULoss", "SoftDiceLoss", "TverskyLoss", "FocalLoss", "BDLoss"]
