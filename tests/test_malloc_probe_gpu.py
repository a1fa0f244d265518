"""Placement probe (csrc/kernels/probe.hip) and the candidate choice in wholememory_malloc (csrc/memory_handle.cpp:alloc_local;
OPT-IN since round 4: WM_MALLOC_PROBE=auto is the self-calibrating search, =K forces K candidates, unset / 1 = one plain
allocation like the reference): the best of the probed candidate allocations is kept for a device shard; the table it backs must behave like any other (gather / scatter parity
with the closed form of the reference's own tests, wholememory_gather_tests.cu:288-528), and the probe itself must report a
positive time for every kind without touching memory outside [ptr, ptr + bytes)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
import os
rows, dim = int(os.environ.get("PROBE_TEST_ROWS", "65536")), 64      # default: 16 MiB shard, above the lowered threshold
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
t = emb.get_embedding_tensor()
g = torch.Generator(device="cuda").manual_seed(7)
idx = torch.randperm(20000, device="cuda", generator=g) * (rows // 20000)      # distinct rows spread over the shard
src = (idx.to(torch.float32)[:, None] + torch.arange(dim, device="cuda", dtype=torch.float32)[None, :]).contiguous()
t.scatter(src, idx)
out = emb.gather(idx)
assert torch.equal(out, src), "gather after scatter differs"
loc, first = t.get_local_tensor()
print("OK", int(first))
""" % ROOT


def test_malloc_probe_keeps_a_working_table(wm_lib):
    env = dict(os.environ, WM_MALLOC_PROBE="3", WM_MALLOC_PROBE_MIN_BYTES=str(1 << 20), WM_MALLOC_PROBE_VERBOSE="1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 0" in r.stdout
    lines = [l for l in r.stderr.splitlines() if "malloc probe: candidate" in l]
    assert len(lines) == 3, r.stderr            # three candidates were allocated and timed
    for l in lines:
        assert float(l.split(":")[-1].split()[0]) > 0


@pytest.mark.parametrize("rel,lo,hi", [("1000", 2, 2), ("1e-9", 2, 4)])
def test_auto_probe_stops_when_two_candidates_agree(wm_lib, rel, lo, hi):
    """WM_MALLOC_PROBE=auto: candidates are added until two of them are within WM_MALLOC_PROBE_REL of the best seen (a huge
    tolerance: the second one always agrees; a tiny one: the search runs to its cap of 4) — no absolute threshold."""
    env = dict(os.environ, WM_MALLOC_PROBE="auto", WM_MALLOC_PROBE_MIN_BYTES=str(1 << 20), WM_MALLOC_PROBE_VERBOSE="1",
               WM_MALLOC_PROBE_REL=rel)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 0" in r.stdout
    lines = [l for l in r.stderr.splitlines() if "malloc probe: candidate" in l]
    assert lo <= len(lines) <= hi, r.stderr


def test_default_is_one_plain_allocation(wm_lib):
    """no WM_MALLOC_PROBE in the environment: the reference's behaviour, one allocation, no probe"""
    env = dict(os.environ, WM_MALLOC_PROBE_MIN_BYTES=str(1 << 20), WM_MALLOC_PROBE_VERBOSE="1")
    env.pop("WM_MALLOC_PROBE", None)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK 0" in r.stdout, r.stdout + r.stderr
    assert "malloc probe: candidate" not in r.stderr


def test_three_processes_probe_one_device_at_once(wm_lib):
    """three processes allocate probed tables on the same GPU at the same time: the per-device file lock makes them probe one
    after the other, none runs the others (or itself) out of memory, every table works"""
    rows = (2 << 30) // 256      # 2 GiB shards
    env = dict(os.environ, WM_MALLOC_PROBE="auto", WM_MALLOC_PROBE_VERBOSE="1", PROBE_TEST_ROWS=str(rows))
    procs = [subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(3)]
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0 and "OK 0" in out, out + err
        assert len([l for l in err.splitlines() if "malloc probe: candidate" in l]) >= 2, err


def test_candidates_that_do_not_fit_end_the_search_quietly(wm_lib):
    """WM_MALLOC_PROBE=8 on a 40 GB shard: the losers alive together are capped at a quarter of the memory that is free after
    the first allocation (≈ 62 GB on an empty 288 GB device: ONE loser) — the search ends there after two candidates, the
    better one is kept, and nothing is left behind for the ops that follow. (A 70 GB shard is not probed at all: no second
    candidate fits the cap.)"""
    free_b, _ = torch.cuda.mem_get_info()
    rows = 40 * (1 << 30) // (64 * 4)
    if free_b < 3 * rows * 256:
        pytest.skip("needs most of an empty device")
    env = dict(os.environ, WM_MALLOC_PROBE="8", WM_MALLOC_PROBE_VERBOSE="1", PROBE_TEST_ROWS=str(rows))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 0" in r.stdout
    lines = [l for l in r.stderr.splitlines() if "malloc probe: candidate" in l]
    assert 2 <= len(lines) < 8, r.stderr


def test_probe_switched_off(wm_lib):
    env = dict(os.environ, WM_MALLOC_PROBE="1", WM_MALLOC_PROBE_MIN_BYTES=str(1 << 20), WM_MALLOC_PROBE_VERBOSE="1")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK 0" in r.stdout, r.stdout + r.stderr
    assert "malloc probe: candidate" not in r.stderr


def test_probe_kinds_and_bounds(wm_lib):
    from wholegraph_amd import binding as wmb
    L = wmb.lib()
    n = 8 << 20
    guard = 4096
    buf = torch.full((n + 2 * guard,), 7, dtype=torch.uint8, device="cuda")
    ptr = buf.data_ptr() + guard
    for kind in (1, 2):                          # the non-destructive kinds leave every byte as it was
        ms = ctypes.c_float(0)
        wmb.check(L.wholememory_ext_probe_memory(ctypes.c_void_p(ptr), ctypes.c_size_t(n), kind, 2, ctypes.byref(ms)))
        assert ms.value > 0
        assert bool((buf == 7).all())
    ms = ctypes.c_float(0)
    wmb.check(L.wholememory_ext_probe_memory(ctypes.c_void_p(ptr), ctypes.c_size_t(n), 0, 2, ctypes.byref(ms)))
    assert ms.value > 0
    assert bool((buf[:guard] == 7).all()) and bool((buf[-guard:] == 7).all())   # kind 0 writes zeros inside the range only
    assert int((buf[guard:-guard] == 0).sum()) > 0
    with pytest.raises(Exception):
        wmb.check(L.wholememory_ext_probe_memory(ctypes.c_void_p(ptr), ctypes.c_size_t(n), 9, 1, ctypes.byref(ms)))
