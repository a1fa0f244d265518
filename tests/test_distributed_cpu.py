"""world_size > 1 on CPU: the product's distributed orchestration (bucket -> counts all-to-all -> ids
all-to-all-v -> owner gather -> rows all-to-all-v -> reorder; scatter mirror; gradient apply with dedup +
optimizer) runs in N processes over torch.distributed/gloo with the device seam served by the CPU test
backend, and is compared bit-exactly with the oracle's multi-rank simulation. This is what makes the
8-GPU path correct by construction: the SAME C++ host code runs there with HIP kernels + RCCL underneath."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, mode, extra_env):
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), port, mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in o, "rank %d failed:\n%s" % (
            r, "\n=====\n".join(x[-2500:] for x in outs))


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", ["1", "4"])
@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_on_one_gpu_hip_kernels(wm_lib, world, chunks):
    """N processes sharing cuda:0: the real HIP kernels + the real multi-rank orchestration, hipIpc-mapped
    CHUNKED shards across processes, collectives over gloo (RCCL refuses two ranks on one device, and the
    test boxes have a single GPU)."""
    run_world(world, "hip", {"WM_EXCHANGE_CHUNKS": chunks})


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_on_one_gpu_owner_tables_from_device_arrays(wm_lib, world):
    """the same scenarios with WM_ROWS_OWNERS_BY_VALUE=0: the row kernels resolve a chunked row's owner from the gref's
    DEVICE arrays (what a hand-built gref gets) — the multiply-high that replaced the reference's 64-bit divide
    (device_reference.cuh:47) for equal chunks, the search over rank offsets for custom partitions — instead of from the
    owner tables passed by value."""
    run_world(world, "hip", {"WM_EXCHANGE_CHUNKS": "1", "WM_ROWS_OWNERS_BY_VALUE": "0"})


@pytest.mark.parametrize("world,chunks", [(1, "1"), (2, "1"), (2, "3"), (3, "1"), (3, "3"), (8, "4")])
def test_distributed_paths_over_gloo(wm_lib, world, chunks):
    # always through make: the test backend shares struct layouts with csrc/backend.hpp and must be rebuilt when that
    # header changes (a no-op when it is up to date)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "test_backend"], stdout=subprocess.DEVNULL)
    run_world(world, "cpu", {"WHOLEGRAPH_AMD_TESTING": "1", "HIP_VISIBLE_DEVICES": "", "WM_EXCHANGE_CHUNKS": chunks})


@pytest.mark.parametrize("world,chunks", [(2, "1"), (3, "2")])
def test_loopback_self_segment_over_gloo(wm_lib, world, chunks):
    """WM_EXCHANGE_SELF=1: a rank's own segment goes through the transport like a peer's (the mode the one-GPU RCCL tests
    rely on, tests/test_rccl_transport_gpu.py) — results must not change."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "test_backend"], stdout=subprocess.DEVNULL)
    run_world(world, "cpu", {"WHOLEGRAPH_AMD_TESTING": "1", "HIP_VISIBLE_DEVICES": "", "WM_EXCHANGE_CHUNKS": chunks,
                             "WM_EXCHANGE_SELF": "1"})


@pytest.mark.parametrize("world,local,chunks", [(4, 2, "1"), (6, 3, "2"), (6, 2, "1"), (4, 1, "1"), (3, 3, "1")])
def test_hierarchy_gather_over_gloo(wm_lib, world, local, chunks):
    """HIERARCHY tables on a pretended `world / local` nodes x `local` ranks layout: ids relayed inside the node, distinct
    rows fetched along the rails, results bit-identical to the oracle's DISTRIBUTED gather."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "test_backend"], stdout=subprocess.DEVNULL)
    run_world(world, "cpu-hier", {"WHOLEGRAPH_AMD_TESTING": "1", "HIP_VISIBLE_DEVICES": "", "WM_EXCHANGE_CHUNKS": chunks,
                                  "WM_LOCAL_SIZE": str(local)})


@pytest.mark.gpu
@pytest.mark.parametrize("world,local", [(4, 2), (3, 3)])
def test_hierarchy_gather_hip_kernels(wm_lib, world, local):
    run_world(world, "hip-hier", {"WM_LOCAL_SIZE": str(local)})


@pytest.mark.parametrize("world,local,seed", [(2, 0, 1), (3, 0, 2), (4, 2, 3), (5, 0, 4), (6, 3, 5), (6, 2, 6)])
def test_random_shapes_and_partitions_over_gloo(wm_lib, world, local, seed):
    """Random table shapes / dtype pairs / id dtypes / custom row partitions through the multi-rank gather + scatter
    scenario (DISTRIBUTED, and HIERARCHY on a pretended node layout), bit-exact vs the oracle's multi-rank simulation."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "test_backend"], stdout=subprocess.DEVNULL)
    env = {"WHOLEGRAPH_AMD_TESTING": "1", "HIP_VISIBLE_DEVICES": "", "FUZZ_SEED": str(seed), "FUZZ_CASES": "12",
           "WM_EXCHANGE_CHUNKS": str(1 + seed % 3)}
    if local:
        env["WM_LOCAL_SIZE"] = str(local)
    run_world(world, "cpu-fuzz", env)


@pytest.mark.gpu
@pytest.mark.parametrize("world,local,seed", [(3, 0, 11), (4, 2, 12)])
def test_random_shapes_and_partitions_hip_kernels(wm_lib, world, local, seed):
    env = {"FUZZ_SEED": str(seed), "FUZZ_CASES": "10"}
    if local:
        env["WM_LOCAL_SIZE"] = str(local)
    run_world(world, "hip-fuzz", env)
