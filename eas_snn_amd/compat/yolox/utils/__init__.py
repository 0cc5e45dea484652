from .allreduce_norm import all_reduce_norm, get_async_norm_states
from .boxes import postprocess
from .checkpoint import load_ckpt, save_checkpoint
from .dist import (all_gather, ensure_process_group, gather, get_local_rank, get_local_size, get_num_devices, get_rank, get_world_size,
                   is_main_process, synchronize, time_synchronized, wait_for_the_master, wait_process_group_idle)
from .ema import ModelEMA, is_parallel
from .logger import setup_logger
from .lr_scheduler import LRScheduler
from .model_utils import adjust_status, fuse_model, get_model_info
from .setup_env import configure_module, configure_nccl, configure_omp
from .util import warp_decay
