from .trainer import batch_processor, build_optimizer, change_input_variable, Runner, DataParallelShell
from .env import init_dist, get_dist_info, set_random_seed

__all__ = ["batch_processor", "build_optimizer", "change_input_variable", "Runner", "DataParallelShell", "init_dist",
           "get_dist_info", "set_random_seed"]
