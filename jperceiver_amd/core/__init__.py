from .dist_utils import DistOptimizerHook, allreduce_grads

__all__ = ["DistOptimizerHook", "allreduce_grads"]
