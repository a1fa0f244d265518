"""WholeMemory communicators on top of torch.distributed bootstrap.

Mirrors reference ``python/pylibwholegraph/pylibwholegraph/torch/comm.py`` (:68-130 class, :133-168
create_group_communicator, :171-290 getters): the 128-byte RCCL unique id is created on each group root
and broadcast through torch.distributed; ``wholememory_create_communicator`` then builds the library's
own RCCL communicator (backend "nccl" of torch.distributed IS RCCL on ROCm).

When torch.distributed runs on a non-GPU backend (gloo — the CPU test-suite) the communicator is built
over ``wholememory_create_communicator_ext`` with collectives implemented by torch.distributed itself.
"""
import ctypes as C

import torch
import torch.distributed as dist

from .. import binding as wmb
from .utils import (
    str_to_wmb_wholememory_distributed_backend_type,
    wholememory_distributed_backend_type_to_str,
    str_to_wmb_wholememory_memory_type,
    str_to_wmb_wholememory_location,
)

class _JobTopology(object):
    """What this process knows about the job (ranks / ranks per node, told by init()) and the communicators already built
    for the three standard scopes. A scope that coincides with another one (a single-node job: node == world; a single-GPU
    job: device == node == world) shares that communicator instead of building a second one."""

    def __init__(self):
        self.world_rank, self.world_size, self.local_rank, self.local_size = 0, 1, 0, 1
        self.world = {}        # distributed backend name -> communicator over every rank
        self.node = None       # ranks of this node
        self.device = None     # this rank alone
        self.mnnvl = None

    def scope_sizes(self):
        return {"world": self.world_size, "node": self.local_size, "device": 1}


_job = _JobTopology()


def reset_communicators():
    global _job
    _job = _JobTopology()


def set_world_info(world_rank, world_size, local_rank, local_size):
    _job.world_rank, _job.world_size, _job.local_rank, _job.local_size = world_rank, world_size, local_rank, local_size


class WholeMemoryCommunicator(object):
    """Use create_group_communicator / get_global_communicator / ... instead of constructing directly."""

    def __init__(self, wmb_comm, keepalive=None):
        super().__init__()
        self.wmb_comm = wmb_comm  # c_void_p (wholememory_comm_t)
        self._keepalive = keepalive

    def get_rank(self):
        v = C.c_int()
        wmb.check(wmb.lib().wholememory_communicator_get_rank(C.byref(v), self.wmb_comm))
        return v.value

    def get_size(self):
        v = C.c_int()
        wmb.check(wmb.lib().wholememory_communicator_get_size(C.byref(v), self.wmb_comm))
        return v.value

    def get_clique_info(self):
        info = (C.c_int * 6)()
        wmb.check(wmb.lib().wholememory_communicator_get_clique_info(info, self.wmb_comm))
        return tuple(info)

    def barrier(self):
        wmb.check(wmb.lib().wholememory_communicator_barrier(self.wmb_comm))

    def support_type_location(self, memory_type, memory_location):
        rc = wmb.lib().wholememory_communicator_support_type_location(
            self.wmb_comm, str_to_wmb_wholememory_memory_type(memory_type),
            str_to_wmb_wholememory_location(memory_location))
        return rc == wmb.WHOLEMEMORY_SUCCESS

    def destroy(self):
        destroy_communicator(self)

    def transport(self):
        """(name, ranks) of the collective transport under this communicator: ("rccl", N) on MI355X boxes, ("external", -1)
        over torch.distributed/gloo, ("none", 0) for a single rank. Extension (wholegraph_amd_ext.h)."""
        name, ranks = C.c_char_p(), C.c_int()
        wmb.check(wmb.lib().wholememory_ext_communicator_transport(self.wmb_comm, C.byref(name), C.byref(ranks)))
        return name.value.decode(), ranks.value

    @property
    def distributed_backend(self):
        return wholememory_distributed_backend_type_to_str(
            wmb.lib().wholememory_communicator_get_distributed_backend(self.wmb_comm))

    @distributed_backend.setter
    def distributed_backend(self, value):
        wmb.check(wmb.lib().wholememory_communicator_set_distributed_backend(
            self.wmb_comm, str_to_wmb_wholememory_distributed_backend_type(value)))


class _TorchDistCollectives(object):
    """wm_ext_collectives_t implemented with torch.distributed (any backend) on a process group.

    "Device" buffers are whatever the installed device backend calls device memory: host memory
    under the CPU test backend, HBM (staged through pinned host copies) otherwise.
    """

    def __init__(self, group, rank, size, device_is_host):
        self.group, self.rank, self.size, self.device_is_host = group, rank, size, device_is_host
        self._barrier = wmb.BARRIER_FN(self._do_barrier)
        self._allgather = wmb.ALLGATHER_HOST_FN(self._do_allgather)
        self._alltoallv = wmb.ALLTOALLV_FN(self._do_alltoallv)
        self.table = wmb.ExtCollectives(None, self._barrier, self._allgather, self._alltoallv)

    @staticmethod
    def _host_view(ptr, nbytes):
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8)
        return torch.frombuffer((C.c_char * nbytes).from_address(int(ptr)), dtype=torch.uint8, count=nbytes)

    def _do_barrier(self, _ctx):
        try:
            dist.barrier(group=self.group)
            return 0
        except Exception as e:  # pragma: no cover
            print("ext barrier failed:", e)
            return 1

    def _do_allgather(self, _ctx, send, recv, nbytes):
        try:
            src = self._host_view(send, nbytes).clone()
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.size)]
            dist.all_gather(outs, src, group=self.group)
            dst = self._host_view(recv, nbytes * self.size)
            for r in range(self.size):
                dst[r * nbytes:(r + 1) * nbytes] = outs[r]
            return 0
        except Exception as e:  # pragma: no cover
            print("ext allgather failed:", e)
            return 1

    def _do_alltoallv(self, _ctx, send, sbytes, sdisp, recv, rbytes, rdisp, stream):
        try:
            W = self.size
            if not self.device_is_host:
                torch.cuda.synchronize()
            sends = []
            for r in range(W):
                n = int(sbytes[r])
                if self.device_is_host:
                    sends.append(self._host_view(int(send or 0) + int(sdisp[r]), n).clone())
                else:
                    from .wholegraph_env import torch_tensor_from_pointer
                    sends.append(torch_tensor_from_pointer(int(send or 0) + int(sdisp[r]), [n], torch.int8, [1], True).cpu()
                                 .view(torch.uint8) if n else torch.empty(0, dtype=torch.uint8))
            recvs = [torch.empty(int(rbytes[r]), dtype=torch.uint8) for r in range(W)]
            # pairwise exchange with point-to-point ops (works on every torch.distributed backend)
            reqs = []
            for r in range(W):
                if r == self.rank:
                    recvs[r].copy_(sends[r])
                    continue
                if recvs[r].numel():
                    reqs.append(dist.irecv(recvs[r], src=dist.get_global_rank(self.group, r) if self.group else r,
                                           group=self.group))
                if sends[r].numel():
                    reqs.append(dist.isend(sends[r], dst=dist.get_global_rank(self.group, r) if self.group else r,
                                           group=self.group))
            for q in reqs:
                q.wait()
            for r in range(W):
                n = int(rbytes[r])
                if n == 0:
                    continue
                if self.device_is_host:
                    self._host_view(int(recv or 0) + int(rdisp[r]), n).copy_(recvs[r])
                else:
                    from .wholegraph_env import torch_tensor_from_pointer
                    torch_tensor_from_pointer(int(recv or 0) + int(rdisp[r]), [n], torch.int8, [1], True).copy_(
                        recvs[r].view(torch.int8))
            if not self.device_is_host:
                torch.cuda.synchronize()
            return 0
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            print("ext alltoallv failed:", e)
            return 1


def _use_rccl_transport():
    return dist.is_initialized() and dist.get_backend() == "nccl" and torch.cuda.is_available()


def create_group_communicator(group_size=-1, comm_stride=1):
    """Partition the world into groups of `group_size` ranks taken with stride `comm_stride`
    (24 ranks, group_size 4, comm_stride 2 -> [0,2,4,6], [1,3,5,7], [8,10,12,14], ...) and return
    this rank's communicator. reference comm.py:133-168."""
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    world_rank = dist.get_rank() if dist.is_initialized() else 0
    if group_size == -1:
        group_size = world_size
    # The world is cut into blocks of group_size * comm_stride consecutive ranks; inside a block, the ranks congruent modulo
    # comm_stride form one group (block b, lane l: ranks b * block + l + k * comm_stride, k = 0 .. group_size - 1).
    block = group_size * comm_stride
    assert world_size % block == 0, "group_size * comm_stride must divide the world size"
    n_blocks = world_size // block
    my_block, within_block = divmod(world_rank, block)
    my_pos, my_lane = divmod(within_block, comm_stride)      # position inside my group, which of the block's groups
    L = wmb.lib()
    comm = C.c_void_p()
    import os
    if group_size == 1:
        # a one-rank group needs no bootstrap traffic. WM_FORCE_RCCL=1 still gives it a real RCCL communicator of size 1
        # (the library reads the switch): the unique id is then made right here instead of being broadcast.
        uid = wmb.UniqueId()
        if os.environ.get("WM_FORCE_RCCL") == "1":
            wmb.check(L.wholememory_create_unique_id(C.byref(uid)))
        wmb.check(L.wholememory_create_communicator(C.byref(comm), uid, 0, 1))
        return WholeMemoryCommunicator(comm)
    if not _use_rccl_transport():
        # host-framework collectives (gloo etc.): one torch process group per wholememory group
        my_group, my_ranks = None, None
        for b in range(n_blocks):
            for lane in range(comm_stride):
                ranks = [b * block + lane + k * comm_stride for k in range(group_size)]
                g = dist.new_group(ranks=ranks) if world_size != group_size else None
                if b == my_block and lane == my_lane:
                    my_group, my_ranks = g, ranks
        device_is_host = L.wholememory_ext_backend_name() != b"hip-gfx950"
        coll = _TorchDistCollectives(my_group, my_pos, group_size, device_is_host)
        wmb.check(L.wholememory_create_communicator_ext(C.byref(comm), my_pos, group_size, C.byref(coll.table)))
        return WholeMemoryCommunicator(comm, keepalive=coll)
    my_uid = wmb.UniqueId()
    for b in range(n_blocks):
        for lane in range(comm_stride):
            root = b * block + lane
            tmp = wmb.UniqueId()
            if world_rank == root:
                wmb.check(L.wholememory_create_unique_id(C.byref(tmp)))
            uid_t = torch.frombuffer(bytearray(C.string_at(C.byref(tmp), wmb.UNIQUE_ID_BYTES)), dtype=torch.uint8).cuda()
            dist.broadcast(uid_t, root)
            if b == my_block and lane == my_lane:
                raw = bytes(uid_t.cpu().numpy().tobytes())
                C.memmove(C.byref(my_uid), raw, wmb.UNIQUE_ID_BYTES)
    wmb.check(L.wholememory_create_communicator(C.byref(comm), my_uid, my_pos, group_size))
    return WholeMemoryCommunicator(comm)


def split_communicator(comm, color, key=0):
    if not isinstance(color, int) or not isinstance(key, int):
        raise TypeError("color and key must be int")
    if color < 0:
        return None
    new_comm = C.c_void_p()
    wmb.check(wmb.lib().wholememory_split_communicator(C.byref(new_comm), comm.wmb_comm, color, key))
    return WholeMemoryCommunicator(new_comm)


def destroy_communicator(wm_comm):
    if wm_comm is not None and wm_comm.wmb_comm is not None:
        wmb.check(wmb.lib().wholememory_destroy_communicator(wm_comm.wmb_comm))
        wm_comm.wmb_comm = None


def comm_set_distributed_backend(wm_comm, distributed_backend):
    wm_comm.distributed_backend = distributed_backend


def get_global_communicator(distributed_backend="nccl"):
    """Communicator over every rank of the job (one per distributed backend name; reference comm.py:197-218)."""
    comm = _job.world.get(distributed_backend)
    if comm is None:
        comm = create_group_communicator()
        comm_set_distributed_backend(comm, distributed_backend)
        _job.world[distributed_backend] = comm
        if distributed_backend == "nccl":           # the node / device scopes only ever use this backend
            sizes = _job.scope_sizes()
            if _job.node is None and sizes["node"] == sizes["world"]:
                _job.node = comm
            if _job.device is None and sizes["world"] == 1:
                _job.device = comm
    return comm


def get_local_node_communicator():
    """Communicator over the ranks of this node."""
    if _job.node is None:
        sizes = _job.scope_sizes()
        _job.node = create_group_communicator(sizes["node"])
        if sizes["node"] == sizes["world"]:
            assert "nccl" not in _job.world
            _job.world["nccl"] = _job.node
        if sizes["node"] == 1:
            assert _job.device is None
            _job.device = _job.node
    return _job.node


def get_local_device_communicator():
    """Communicator of this rank alone."""
    if _job.device is None:
        sizes = _job.scope_sizes()
        _job.device = create_group_communicator(1)
        if sizes["node"] == 1:
            assert _job.node is None
            _job.node = _job.device
        if sizes["world"] == 1:
            assert "nccl" not in _job.world
            _job.world["nccl"] = _job.device
    return _job.device


def get_local_mnnvl_communicator():
    """Ranks of this rank's multi-node NVLink clique — there is no such domain on MI355X nodes: the clique query answers
    "not in a clique" and this raises like the reference does on such a system."""
    if _job.mnnvl is None:
        world = get_global_communicator()
        in_clique, _, _, _, clique_id, _ = world.get_clique_info()
        if not in_clique:
            raise RuntimeError("the gpu does not belong to any mnnvl domain,can not create local_mnnvl_communicator")
        _job.mnnvl = split_communicator(world, clique_id)
    return _job.mnnvl
