"""Process-group bring-up with the reference's interface (mono/apis/env.py:17-38): one process per GPU,
env-var rendezvous from the launcher, `rank % num_gpus` device binding, backend "nccl" (= RCCL over
xGMI on ROCm).  CPU runs (tests) pass backend="gloo"."""
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def init_dist(launcher="pytorch", backend="nccl", **kwargs):
    if launcher != "pytorch":
        raise ValueError("Invalid launcher type: {}".format(launcher))
    rank = int(os.environ["RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this platform
    if backend == "nccl":
        num_gpus = torch.cuda.device_count()
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank % max(1, num_gpus))))
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def get_root_logger(log_level="INFO"):
    """mono/apis/env.py: the process-wide logger at `log_level` on rank 0, ERROR elsewhere (one line per event, not one per rank)."""
    import logging
    logger = logging.getLogger()
    if not logger.hasHandlers():
        logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=log_level)
    rank, _ = get_dist_info()
    logger.setLevel("ERROR" if rank != 0 else log_level)
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    from .. import ops
    ops.manual_seed(seed)
