"""Library + torch.distributed bootstrap.

Same four entry points as reference ``python/pylibwholegraph/pylibwholegraph/torch/initialize.py:22-83``
(``init``, ``init_torch_env``, ``init_torch_env_and_create_wm_comm``, ``finalize``) with the same arguments. The process
group is created on "nccl" (= RCCL on ROCm); rendezvous defaults to 127.0.0.1:12335 when the launcher did not set one.
"""
import os

import torch

from .. import binding as wmb
from . import comm as _comm
from .utils import str_to_wmb_wholememory_log_level

_RENDEZVOUS_DEFAULTS = (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "12335"))


def _start_library(wm_log_level, ranks):
    wmb.check(wmb.lib().wholememory_init(0, str_to_wmb_wholememory_log_level(wm_log_level)))
    _comm.set_world_info(*ranks)


def init(world_rank, world_size, local_rank, local_size, wm_log_level="info"):
    """Library only: the caller has already set up torch.distributed (or runs a single process)."""
    _start_library(wm_log_level, (world_rank, world_size, local_rank, local_size))


def init_torch_env(world_rank, world_size, local_rank, local_size, wm_log_level="info"):
    """Library + the torch process group of this rank (one process per GPU, device = local_rank)."""
    os.environ.update(RANK=str(world_rank), WORLD_SIZE=str(world_size))
    for name, default in _RENDEZVOUS_DEFAULTS:
        if name not in os.environ:
            if world_rank == 0:
                print("[WARNING] %s not set, using %s" % (name, default))
            os.environ[name] = default
    torch.set_num_threads(1)
    torch.cuda.set_device(local_rank)
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend="nccl", init_method="env://")
    _start_library(wm_log_level, (world_rank, world_size, local_rank, local_size))


def init_torch_env_and_create_wm_comm(world_rank, world_size, local_rank, local_size, distributed_backend_type="nccl",
                                      wm_log_level="info"):
    """-> (communicator over every rank, communicator over the ranks of this node)"""
    init_torch_env(world_rank, world_size, local_rank, local_size, wm_log_level)
    return _comm.get_global_communicator(distributed_backend_type), _comm.get_local_node_communicator()


def finalize():
    wmb.lib().wholememory_finalize()
    _comm.reset_communicators()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
