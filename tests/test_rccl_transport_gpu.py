"""The RCCL transport itself (csrc/communicator.cpp:rccl_provider — ncclCommInitRank, barrier allreduce, the host
all-gather behind the counts exchange, grouped ncclSend/ncclRecv all-to-all-v, ncclCommSplit) under the DISTRIBUTED ops,
bit-exact against the oracle. Reference: cpp/src/wholememory/nccl_comms.cpp:82-86,383-437, communicator.cpp:703-752.

RCCL refuses two ranks on one device and the test boxes have one GPU, so the one-GPU variants run ONE rank with
  WM_FORCE_RCCL=1         a real RCCL communicator of size 1 instead of "no transport"
  WM_EXCHANGE_SELF=1      the rank's own segment goes bucket -> counts -> all-to-all-v like a peer's
  WM_RCCL_SELF_SENDRECV=1 ...and travels as an ncclSend/ncclRecv pair inside the group, not a device-to-device copy
so every RCCL call of the provider executes. With >= 2 GPUs visible the same scenarios also run as real multi-GPU jobs."""
import pytest
import torch

from test_distributed_cpu import run_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sendrecv,chunks", [("1", "1"), ("1", "3"), ("0", "1")])
def test_rccl_provider_world1_loopback(wm_lib, sendrecv, chunks):
    run_world(1, "hip-rccl", {"WM_FORCE_RCCL": "1", "WM_EXCHANGE_SELF": "1", "WM_RCCL_SELF_SENDRECV": sendrecv,
                              "WM_EXCHANGE_CHUNKS": chunks})


def test_rccl_provider_world1_plain(wm_lib):
    """forced RCCL communicator, default routing: own rows served locally, only barrier / counts all-gather hit RCCL"""
    run_world(1, "hip-rccl", {"WM_FORCE_RCCL": "1"})


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_multi_gpu(wm_lib, world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (this box has %d)" % (world, torch.cuda.device_count()))
    run_world(world, "hip-rccl", {})
