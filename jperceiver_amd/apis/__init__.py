from .trainer import (batch_processor, build_optimizer, change_input_variable, Runner, DataParallelShell,
                      StepLrUpdaterHook, train_mono)
from .env import init_dist, get_dist_info, set_random_seed, get_root_logger
from .checkpoint import save_checkpoint, load_checkpoint, weights_to_cpu
from .inference import evaluate_depth, pose_between, chain_poses, odometry, pose_nets_from_checkpoint

__all__ = ["batch_processor", "train_mono", "build_optimizer", "change_input_variable", "Runner", "DataParallelShell", "init_dist",
           "get_dist_info", "set_random_seed", "get_root_logger", "StepLrUpdaterHook", "save_checkpoint", "load_checkpoint", "weights_to_cpu"]
