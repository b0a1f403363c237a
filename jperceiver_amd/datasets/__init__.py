from .sampler import DistributedGroupSampler, DistributedSampler, GroupSampler
from .loader import DeviceLoader, collate
from .preprocess import DevicePreprocessor, ColorJitterParams, pil_resample_tables

__all__ = ["DistributedGroupSampler", "DistributedSampler", "GroupSampler", "DeviceLoader", "collate", "DevicePreprocessor",
           "ColorJitterParams", "pil_resample_tables"]
