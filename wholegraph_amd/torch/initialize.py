"""Library + torch.distributed bootstrap. Mirrors reference
``python/pylibwholegraph/pylibwholegraph/torch/initialize.py:22-83`` (same functions, same env-variable
defaults); the process-group backend is "nccl" (= RCCL on ROCm) when a GPU is visible."""
import os

import torch

from .. import binding as wmb
from .comm import set_world_info, get_global_communicator, get_local_node_communicator, reset_communicators
from .utils import str_to_wmb_wholememory_log_level


def init(world_rank, world_size, local_rank, local_size, wm_log_level="info"):
    wmb.check(wmb.lib().wholememory_init(0, str_to_wmb_wholememory_log_level(wm_log_level)))
    set_world_info(world_rank, world_size, local_rank, local_size)


def init_torch_env(world_rank, world_size, local_rank, local_size, wm_log_level="info"):
    os.environ["RANK"] = str(world_rank)
    os.environ["WORLD_SIZE"] = str(world_size)
    if "MASTER_ADDR" not in os.environ:
        if world_rank == 0:
            print("[WARNING] MASTER_ADDR not set, resetting to localhost")
        os.environ["MASTER_ADDR"] = "localhost"
    if "MASTER_PORT" not in os.environ:
        if world_rank == 0:
            print("[WARNING] MASTER_PORT not set, resetting to 12335")
        os.environ["MASTER_PORT"] = "12335"
    wmb.check(wmb.lib().wholememory_init(0, str_to_wmb_wholememory_log_level(wm_log_level)))
    torch.set_num_threads(1)
    torch.cuda.set_device(local_rank)
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend="nccl", init_method="env://")
    set_world_info(world_rank, world_size, local_rank, local_size)


def init_torch_env_and_create_wm_comm(world_rank, world_size, local_rank, local_size, distributed_backend_type="nccl",
                                      wm_log_level="info"):
    init_torch_env(world_rank, world_size, local_rank, local_size, wm_log_level)
    global_comm = get_global_communicator(distributed_backend_type)
    local_comm = get_local_node_communicator()
    return global_comm, local_comm


def finalize():
    wmb.lib().wholememory_finalize()
    reset_communicators()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
