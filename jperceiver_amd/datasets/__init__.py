from .sampler import DistributedGroupSampler, DistributedSampler, GroupSampler
from .loader import DeviceLoader, collate, build_dataloader
from .preprocess import DevicePreprocessor, ColorJitterParams, pil_resample_tables

__all__ = ["DistributedGroupSampler", "DistributedSampler", "GroupSampler", "DeviceLoader", "collate", "build_dataloader", "DevicePreprocessor",
           "ColorJitterParams", "pil_resample_tables"]
