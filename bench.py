#!/usr/bin/env python
"""bench.py — WholeMemory embedding gather on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: wholememory_embedding_gather of `--indices`
random int64 ids (default 10 M) into an fp32 [n, 128] output, table resident in HBM.
  N = 1 : config C2 — CHUNKED, 100 M x 128 fp32 (51.2 GB) on one GPU (1 B x 128 = 512 GB does not fit 288 GB)
  N > 1 : config C3 — DISTRIBUTED, 125 M rows per GPU (1 B x 128 at N = 8), every rank gathers its own 10 M
          ids (weak scaling, like the reference bench where every rank gathers `gather_size`), ids and rows
          exchanged by RCCL all-to-all-v over xGMI.
`value` follows the reference bench convention (cpp/bench/wholememory_ops/gather_scatter_bench.cu:364):
output bytes / time / 1e9, aggregated over all ranks; `mlookups_per_s` and the algorithmic
(idx + row read + row write = 1032 B / lookup) rate are reported beside it.
Launch: python bench.py [--gpus N]  (N > 1 without RANK in the environment: the script starts N ranks itself, one per
visible GPU, like the reference bench forks one process per device, gather_scatter_bench.cu:257-284)
   or   python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
The timed region is EXACTLY --steps steps between two barriers; `stability` is a separate leg of per-step HIP-event times
(min / median / p95 over >= 200 steps) run after it, so one noisy neighbour cannot hide in a 36 ms window.
Table and output buffer (gather; source buffer for scatter / SGD gradient apply): plain single allocations, made once before the
timed region as in the reference bench. (`--table-candidates T --out-candidates O` is round 2's side experiment: T x O candidate
pairs probed, the fastest kept, every probe in the line as `placement.probe_ms`. Round 3 launches the row kernels in order —
DESIGN.md section 3.1 — which removed the dependence on the buffers' physical placement, and the default is 1 x 1.)
"""
import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL, hipIpc across ranks): must be set before the HIP runtime loads
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A/B of compile-time variants of the library (scripts/build_variant.sh NAME "-D..."): WHOLEGRAPH_AMD_VARIANT=NAME runs this
# file against experiments/variants/NAME/wholegraph_amd instead of the product package. Never set by the driver.
VARIANT = os.environ.get("WHOLEGRAPH_AMD_VARIANT")
if VARIANT:
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", VARIANT))

import numpy as np
import torch


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--stability-steps", type=int, default=200,
                   help="per-step HIP-event timing leg after the timed region (0 = skip)")
    p.add_argument("--rows", type=int, default=0, help="table rows per GPU (default: 100M at N=1, 125M at N>1)")
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--indices", type=int, default=10_000_000)
    p.add_argument("--dist", choices=["uniform", "zipf", "zipf_clustered", "sequential", "paged", "paged64k"], default="uniform",
                   help="diagnostics, not workloads: sequential = ids 0, 1, 2 ... (the row kernel as a plain streaming copy); paged = "
                        "the uniform ids grouped by the 2 MiB page of their row (random inside a page: the DRAM side stays random, "
                        "the TLB side becomes sequential); paged64k = the same with 64 KiB groups")
    p.add_argument("--memory-type", default="", help="override: continuous|chunked|distributed")
    p.add_argument("--location", default="cuda", help="cuda|cpu (HOST-located table, config C1)")
    p.add_argument("--op", choices=["gather", "scatter", "grad_apply", "sample_gather"], default="gather",
                   help="side measurements; the contract metric is gather. sample_gather = BASELINE config 5: 2-hop "
                        "neighbour sample + append_unique + feature gather on an ogbn-papers100M-shaped synthetic graph")
    p.add_argument("--nodes", type=int, default=111_059_956, help="sample_gather: graph nodes (papers100M: 111,059,956)")
    p.add_argument("--avg-degree", type=int, default=29, help="sample_gather: mean out-degree (papers100M, both directions: 29)")
    p.add_argument("--seeds", type=int, default=1024, help="sample_gather: seed nodes per rank per step")
    p.add_argument("--fanouts", default="30,30", help="sample_gather: fan-out per hop, seeds outwards")
    p.add_argument("--one-graph", action="store_true", help="sample_gather: do not re-time the step on the other --col-dist afterwards")
    p.add_argument("--col-dist", choices=["uniform", "powerlaw"], default="powerlaw",
                   help="sample_gather: how the synthetic graph's neighbour ids are drawn. uniform (every line before round 6): "
                        "no node is a hub. powerlaw (the default since round 6; the other one is re-timed in the same process and "
                        "reported under other_graph): node of popularity rank k with probability ~ k^-s (--col-exponent s < 1, ranks "
                        "hashed over the id range): hubs as in a citation graph — s = 0.8 puts the top node into ~0.5 %% of all edges")
    p.add_argument("--col-exponent", type=float, default=0.8, help="sample_gather --col-dist powerlaw: the exponent s, 0 < s < 1")
    p.add_argument("--c5-flow", choices=["deferred", "reference"], default="deferred",
                   help="sample_gather: deferred = sampling chain and feature gather queued back to back, one host round trip after "
                        "both (extension); reference = sample, wait for the counts, gather (the reference's call sequence)")
    p.add_argument("--optimizer", default="sgd")
    p.add_argument("--dtype", choices=["f32", "f16", "bf16"], default="f32",
                   help="table dtype (side measurements; the contract metric is f32)")
    p.add_argument("--cache-ratio", type=float, default=0.0,
                   help="side measurement: give the embedding a device row cache of this ratio (HOST tables)")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="torch.distributed backend at N > 1: nccl = RCCL over xGMI (the measured configuration); gloo = "
                        "host collectives, which also lets several ranks share one GPU (bring-up of this script only)")
    p.add_argument("--out-candidates", type=int, default=1,
                   help="side experiment (round 2): allocate this many output buffers (and --table-candidates tables), probe every "
                        "pair with a few launches and keep the fastest. Default 1 = plain single allocations, what every caller "
                        "gets: since round 3 the row kernels are launched in order (DESIGN.md section 3.1) and no longer depend on "
                        "the physical placement of the buffers. Every probe is in the line.")
    p.add_argument("--table-candidates", type=int, default=1, help="see --out-candidates")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-check", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def make_indices(n, total_rows, dist, seed):
    rng = np.random.default_rng(seed)
    if dist == "sequential":
        return (np.arange(n, dtype=np.int64) + (seed % 7) * n) % total_rows
    if dist == "uniform":
        return rng.integers(0, total_rows, n, dtype=np.int64)
    if dist in ("paged", "paged64k"):   # (512-byte rows: 4096 rows per 2 MiB page, 128 per 64 KiB)
        ids = rng.integers(0, total_rows, n, dtype=np.int64)
        return ids[np.argsort(ids >> (12 if dist == "paged" else 7), kind="stable")]
    # Zipf(s = 1.05) popularity rank k; hashed to a row so hot rows spread over owners,
    # or clustered (idx = k: every hot row on rank 0 = worst-case link skew)
    k = rng.zipf(1.05, n).astype(np.uint64)
    if dist == "zipf":
        return ((k * np.uint64(2654435761)) % np.uint64(total_rows)).astype(np.int64)
    return (k % np.uint64(total_rows)).astype(np.int64)


def fill_table(local, row_start):
    """value(row r, any col) = float(r & 0xFFFFFF): the reference tests' closed form
    (cpp/tests/wholememory_ops/embedding_test_utils.cu:197-238), exactly representable."""
    rows = local.shape[0]
    chunk = 4 << 20
    for s in range(0, rows, chunk):
        e = min(rows, s + chunk)
        r = torch.arange(row_start + s, row_start + e, device="cuda", dtype=torch.int64) & 0xFFFFFF
        local[s:e] = r.to(torch.float32).unsqueeze(1).to(local.dtype).to(local.device)
    torch.cuda.synchronize()


def usable_cores():
    """cores this process may really use: affinity mask, capped by a cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_gather_rate(rows, dim, n, seconds, threads):
    """passes of the oracle's OpenMP gather of n random int64 ids over a host rows x dim fp32 table for ~seconds"""
    import oracle
    oracle.set_num_threads(threads)
    table = np.empty((rows, dim), dtype=np.float32)
    table[:] = (np.arange(rows, dtype=np.int64) & 0xFFFFFF).astype(np.float32)[:, None]
    tab = oracle.ShardedTable([table], np.array([0, rows], dtype=np.uint64), dim)
    idx = np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)
    out = np.empty((n, dim), dtype=np.float32)
    oracle.gather(tab, idx, out)  # warm
    t0, passes = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        oracle.gather(tab, idx, out)
        passes += 1
    dt = time.perf_counter() - t0
    return passes * n / dt, passes, dt


def cpu_baseline(dim, seconds):
    """The oracle's gather (OpenMP over indices) on a host-resident table on this box's host cores:
    a bounded sample of the same workload shape (random 512 B rows, int64 ids)."""
    import oracle
    oracle.set_num_threads(usable_cores())
    rows, n = 8_000_000, 2_000_000
    table = np.empty((rows, dim), dtype=np.float32)
    table[:] = (np.arange(rows, dtype=np.int64) & 0xFFFFFF).astype(np.float32)[:, None]
    tab = oracle.ShardedTable([table], np.array([0, rows], dtype=np.uint64), dim)
    idx = np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)
    out = np.empty((n, dim), dtype=np.float32)
    oracle.gather(tab, idx, out)  # warm
    t0, passes = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        oracle.gather(tab, idx, out)
        passes += 1
    dt = time.perf_counter() - t0
    lookups = passes * n / dt
    threads = oracle.num_threads()
    res = {"value": round(lookups * dim * 4 / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
           "mlookups_per_s": round(lookups / 1e6, 2),
           "sample": "oracle gather (C, OpenMP, %d threads): %d passes of %d random int64 ids over a host "
                     "%dx%d fp32 table in %.1f s" % (threads, passes, n, rows, dim, dt)}
    # the two other host numbers SURVEY 8(d) asks for, a few seconds each: the same loop on ONE core, and what a user of
    # the reference gets from get_global_tensor(host_view=True)[idx] (torch.index_select on a CPU tensor)
    oracle.set_num_threads(1)
    t0, passes = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(3.0, seconds):
        oracle.gather(tab, idx, out)
        passes += 1
    res["single_thread_GBps"] = round(passes * n * dim * 4 / (time.perf_counter() - t0) / 1e9, 3)
    oracle.set_num_threads(threads)
    tt, ti = torch.from_numpy(table), torch.from_numpy(idx)
    to = torch.from_numpy(out)
    torch.index_select(tt, 0, ti, out=to)
    t0, passes = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(3.0, seconds):
        torch.index_select(tt, 0, ti, out=to)
        passes += 1
    res["torch_index_select_GBps"] = round(passes * n * dim * 4 / (time.perf_counter() - t0) / 1e9, 3)
    res["torch_threads"] = torch.get_num_threads()
    del table, tab, tt, to, out
    # BASELINE config C1 at its full shape (10 M x 64 fp32 host table, 1 M random ids): the reference's own CPU-runnable
    # case; the GPU number for the same shape (HOST-located table read over PCIe) is reported next to it
    lk, passes, dt = cpu_gather_rate(10_000_000, 64, 1_000_000, min(5.0, seconds), threads)
    res["c1_shape"] = {"value": round(lk * 64 * 4 / 1e9, 3), "unit": "GB/s", "mlookups_per_s": round(lk / 1e6, 2),
                       "cores": threads, "kind": "port",
                       "sample": "oracle gather, %d passes of 1000000 random int64 ids over a host 10000000x64 fp32 table "
                                 "in %.1f s" % (passes, dt)}
    return res


def gpu_c1_host(wgth, comm):
    """config C1 on the GPU path: HOST-located 10 M x 64 fp32 WholeMemory table (pinned, read by the GPU over PCIe /
    Infinity Fabric), 1 M random int64 ids -> device output. Side number for the cpu_baseline c1_shape line."""
    rows, dim, n = 10_000_000, 64, 1_000_000
    emb = wgth.create_embedding(comm, "chunked", "cpu", torch.float32, [rows, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    local[:] = (torch.arange(rows, dtype=torch.int64) & 0xFFFFFF).to(torch.float32).unsqueeze(1)
    idx = torch.from_numpy(np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)).cuda()
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for _ in range(3):
        emb.gather(idx, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out[:, 0], (idx & 0xFFFFFF).to(torch.float32))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from wholegraph_amd import binding as _wmb
    sorted_before = _wmb.lib().wholememory_ext_host_sorted_gathers()
    e0.record()
    for _ in range(20):
        emb.gather(idx, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    took_sorted = _wmb.lib().wholememory_ext_host_sorted_gathers() - sorted_before
    wgth.destroy_embedding(emb)
    return {"ms_per_gather": round(ms, 4), "value": round(n * dim * 4 / (ms * 1e-3) / 1e9, 2), "unit": "GB/s",
            "mlookups_per_s": round(n / (ms * 1e-3) / 1e6, 1), "bound": "pcie",
            "route": "ids sorted by row first (reference gather_op.cpp:116-120), sort inside the timed calls" if took_sorted == 20
                     else "ids as they come",
            "workload": "C1 HOST chunked 10000000x64 fp32 table in pinned host memory, 1000000 uniform int64 ids, output in HBM"}


def gpu_c1_host_cached(wgth, comm, ratio=0.1):
    """config C1 with the device row cache (reference embedding.cpp:564-892): the same HOST-located
    10 M x 64 fp32 table behind a `local_device` cache of `ratio` of its rows (1 M rows = 256 MB of HBM), gathers of 1 M int64
    ids with cache adjustment on — steady state after a warm-up, for uniform ids (a 10 % cache serves ~10 % of them: the cache
    cannot help, and its bookkeeping is in the time) and for Zipf(1.05) ids (the hot rows live in HBM). Same GB/s convention
    as gpu_c1_host and cpu_baseline.c1_shape."""
    rows, dim, n = 10_000_000, 64, 1_000_000
    policy = wgth.create_builtin_cache_policy("local_device", "chunked", "cpu", "readonly", ratio)
    emb = wgth.create_embedding(comm, "chunked", "cpu", torch.float32, [rows, dim], cache_policy=policy)
    local, _ = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    local[:] = (torch.arange(rows, dtype=torch.int64) & 0xFFFFFF).to(torch.float32).unsqueeze(1)
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    res = {"cache_rows": int(rows * ratio), "cache_ratio": ratio,
           "workload": "C1 HOST chunked 10000000x64 fp32 table in pinned host memory behind a local_device row cache, 1000000 int64 "
                       "ids per gather, cache adjusted on every gather, output in HBM"}
    from wholegraph_amd import binding as _wmb
    import ctypes
    for dist in ("uniform", "zipf"):
        rng = np.random.default_rng(43)
        # a fresh batch per step (a cache that sees the same batch again and again is not a measurement)
        batches = [torch.from_numpy(make_indices(n, rows, dist, 100 + k)).cuda() for k in range(8)]
        for k in range(16):    # warm-up: the cache fills and settles
            emb.gather(batches[k % 8], out=out)
        torch.cuda.synchronize()
        assert torch.equal(out[:, 0], (batches[7] & 0xFFFFFF).to(torch.float32))
        info0 = [ctypes.c_int64(0) for _ in range(5)]
        _wmb.check(_wmb.lib().wholememory_ext_embedding_cache_info(emb.wmb_embedding, *[ctypes.byref(x) for x in info0]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(16):
            emb.gather(batches[k % 8], out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 16
        info1 = [ctypes.c_int64(0) for _ in range(5)]
        _wmb.check(_wmb.lib().wholememory_ext_embedding_cache_info(emb.wmb_embedding, *[ctypes.byref(x) for x in info1]))
        hits, lookups = info1[3].value - info0[3].value, info1[4].value - info0[4].value
        res[dist] = {"ms_per_gather": round(ms, 4), "value": round(n * dim * 4 / (ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                     "mlookups_per_s": round(n / (ms * 1e-3) / 1e6, 1),
                     "hit_rate": round(hits / max(lookups, 1), 4)}
        del rng
    wgth.destroy_embedding(emb)
    return res


def run_sample_gather(a, wgth, comm, world, rank, launched, barrier):
    """BASELINE config 5 on synthetic data of the ogbn-papers100M shape: CSR (int64 row_ptr, int32 col) and a [nodes, 128] fp32
    feature table in WholeMemory (CHUNKED on one GPU, DISTRIBUTED over several); one step = unweighted 2-hop sample from
    `--seeds` seed nodes (GraphStructure.multilayer_sample_without_replacement: one-hop sample + append_unique per hop) +
    gather of the features of every node of the sampled sub-graph."""
    mt = a.memory_type or ("chunked" if world == 1 else "distributed")
    nodes, avg = a.nodes, a.avg_degree
    fanouts = [int(x) for x in a.fanouts.split(",")]
    gen = torch.Generator(device="cuda").manual_seed(1)             # the same degrees on every rank
    row = torch.zeros(nodes + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(torch.randint(0, 2 * avg + 1, (nodes,), device="cuda", generator=gen), 0, out=row[1:])
    edges = int(row[-1])
    wrow = wgth.create_wholememory_tensor(comm, mt, "cuda", [nodes + 1], torch.int64, [1])
    wcol = wgth.create_wholememory_tensor(comm, mt, "cuda", [edges], torch.int32, [1])
    lrow, rstart = wrow.get_local_tensor()
    lrow.copy_(row[rstart:rstart + lrow.shape[0]])
    del row
    lcol, _ = wcol.get_local_tensor()
    gen2 = torch.Generator(device="cuda").manual_seed(100 + rank)

    def fill_cols(col_dist):
        for s0 in range(0, lcol.shape[0], 1 << 28):
            e0 = min(lcol.shape[0], s0 + (1 << 28))
            if col_dist == "powerlaw":
                # inverse CDF of the truncated power law: rank = nodes * u^(1 / (1 - s)); ranks hashed over the id range
                u = torch.rand(e0 - s0, device="cuda", generator=gen2, dtype=torch.float64)
                rank_k = (u.pow_(1.0 / (1.0 - a.col_exponent)) * nodes).to(torch.int64).clamp_(0, nodes - 1)
                lcol[s0:e0] = ((rank_k * 2654435761) % nodes).to(torch.int32)
                del u, rank_k
            else:
                lcol[s0:e0] = torch.randint(0, nodes, (e0 - s0,), device="cuda", generator=gen2, dtype=torch.int32)
    fill_cols(a.col_dist)
    feat = wgth.create_embedding(comm, mt, "cuda", torch.float32, [nodes, a.dim])
    lfeat, fstart = feat.get_embedding_tensor().get_local_tensor()
    fill_table(lfeat, fstart)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    seeds = torch.randint(0, nodes, (a.seeds,), device="cuda", generator=gen2, dtype=torch.int32)
    stat = {}

    # --c5-flow deferred (the default): the sampling chain is QUEUED (GraphStructure.multilayer_sample_begin), the feature gather
    # is queued right behind it on the outermost frontier at its upper-bound size (entries behind the sampled nodes are -1 and
    # skipped), and the step's one host round trip — reading the counts — comes after both. Same outputs as the reference
    # flow (sample, wait, size the outputs, gather), which --c5-flow reference runs.
    def step():
        if a.c5_flow == "deferred":
            h = g.multilayer_sample_begin(seeds, fanouts)
            xp = feat.gather(h.padded_frontier)
            tg, ei, rp, ci = h.result()
            x = xp[:tg[0].numel()]
        else:
            tg, ei, rp, ci = g.multilayer_sample_without_replacement(seeds, fanouts)
            x = feat.gather(tg[0])
        stat["nodes"], stat["edges"] = tg[0].numel(), sum(int(c.numel()) for c in ci)
        stat["frontiers"] = [int(t.numel()) for t in tg]
        return x, tg[0]

    torch.cuda.reset_peak_memory_stats()
    resident = torch.cuda.memory_allocated()
    for _ in range(max(a.warmup, 2)):
        x, ids = step()
    barrier()
    if not a.no_check:
        assert torch.equal(x[:, 0], (ids.long() & 0xFFFFFF).to(torch.float32)), "gathered features differ from the closed form"
    # nothing of the warm-up stays referenced: a live output (15 GB at 65536 seeds) in the middle of the caching allocator's
    # blocks makes the next steps' buffers miss their cached blocks, and fresh hipMallocs land inside the timed region
    del x, ids
    step()   # one step in the timed loop's own pattern (outputs dropped at once): the allocator settles before the clock starts
    # The sizes of a step's outputs and scratch buffers vary by ~0.1 % from step to step (random sampler seeds), and the
    # caching allocator answers a request a few MB above every cached block with a fresh hipMalloc — ~370 ms for the 15 GB
    # feature output of a 65536-seed step. One block with 5 % headroom over everything a step ever held at once is put into
    # the cache here; `device_allocs_in_timed_region` in the line says how many fresh allocations still happened (0 expected).
    transient = torch.cuda.max_memory_allocated() - resident
    headroom = torch.empty(int(transient * 1.05) + (64 << 20), dtype=torch.uint8, device="cuda")
    del headroom
    barrier()
    # a full collection of the interpreter's garbage collector walks every object torch's import left behind (~34 ms here,
    # experiments/c5_step_windows.py: one window of 25 steps at 1.66 ms per step, flat 0.29 ms with the collector off); a step
    # allocates enough Python containers to trigger one every few hundred steps. The objects alive now are moved out of the
    # collector's reach for the timed region; the young generations keep being collected.
    gc.collect()
    gc.freeze()
    allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    wall = time.perf_counter() - t0
    gc.unfreeze()
    stat["device_allocs_in_timed_region"] = torch.cuda.memory_stats().get("num_device_alloc", 0) - allocs0
    dt = torch.tensor([wall], device="cuda" if a.backend == "nccl" else "cpu", dtype=torch.float64)
    if launched:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    wall = float(dt.item())
    per = []
    for _ in range(a.stability_steps):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t1) * 1e3)
    ms = wall / a.steps * 1e3
    row_bytes = a.dim * 4
    # algorithmic HBM bytes of one step: per sampled centre one (row_ptr[c], row_ptr[c+1]) pair + its sampled col entries and
    # outputs; append_unique sorts (id, position) pairs of targets + neighbours (3 radix passes, 12 B in + out each);
    # the feature gather moves id + row in + row out per sub-graph node
    centres = sum(stat["frontiers"][1:])
    algo = centres * (16 + 8) + stat["edges"] * (4 + 4 + 4 + 4) + (centres + stat["edges"]) * 12 * 2 * 3 + \
        stat["nodes"] * (8 + 2 * row_bytes)
    res = {
        "metric": "sample_gather_GBps_out (feature bytes of the sampled sub-graph per second; BASELINE config 5)",
        "value": round(stat["nodes"] * row_bytes * world / (ms * 1e-3) / 1e9, 2), "unit": "GB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "sampled_edges_per_s": round(stat["edges"] * world / (ms * 1e-3), 0),
        "subgraph_nodes_per_step": stat["nodes"], "sampled_edges_per_step": stat["edges"], "frontier_sizes": stat["frontiers"],
        "device_allocs_in_timed_region": stat["device_allocs_in_timed_region"],   # fresh hipMallocs by the caching allocator: 0 in a steady state
        "config": {"workload": "C5 %s graph %d nodes / %d edges (int32 col) + %dx%d fp32 features, %d-hop %s unweighted sample "
                               "from %d seeds per rank + append_unique + feature gather" % (
                                   mt, nodes, edges, nodes, a.dim, len(fanouts), fanouts, a.seeds),
                   "memory_type": mt, "seeds_per_rank": a.seeds, "fanouts": fanouts,
                   "neighbour_ids": "uniform" if a.col_dist == "uniform" else "power law, exponent %g" % a.col_exponent},
        "roofline": {"bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(algo / (ms * 1e-3) / 8e12, 4), "traffic": None,
                     "algorithmic_bytes_per_step": algo,
                     "limited_by": "dependent-load latency (row_ptr -> col -> hash table -> features): a step is %d kernels back to "
                                   "back (5 per hop: offsets scan, sampler, table insert, ranking scan, emit; + the feature "
                                   "gather) and ONE host round trip, which in the deferred flow comes after the gather is "
                                   "queued (rocprofv3 timeline: experiments/trace_c5.sh); far from the HBM roofline by "
                                   "construction, larger seed batches move it up (see --seeds)" % (5 * len(fanouts) + 1)},
        "flow": a.c5_flow,
    }
    if per:
        per = np.array(per)
        res["stability"] = {"steps": len(per), "min_ms": round(float(per.min()), 4), "median_ms": round(float(np.median(per)), 4),
                            "p95_ms": round(float(np.percentile(per, 95)), 4), "max_ms": round(float(per.max()), 4),
                            "note": "per-step host times (synchronised), separate from the timed region"}
    if not a.one_graph:
        # the OTHER synthetic graph in the same process (round-5 review: C5 is papers100M, a graph with hubs — the default — and
        # every line before round 6 was measured on the hub-free one): neighbour ids redrawn in place, same degrees and seeds
        other = "uniform" if a.col_dist == "powerlaw" else "powerlaw"
        fill_cols(other)
        for _ in range(3):
            step()
        barrier()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        ms2 = (time.perf_counter() - t2) / a.steps * 1e3
        res["other_graph"] = {"neighbour_ids": "uniform" if other == "uniform" else "power law, exponent %g" % a.col_exponent,
                              "ms_per_step": round(ms2, 4), "subgraph_nodes_per_step": stat["nodes"],
                              "sampled_edges_per_step": stat["edges"],
                              "value": round(stat["nodes"] * row_bytes * world / (ms2 * 1e-3) / 1e9, 2), "unit": "GB/s",
                              "note": "same process, same degrees and seeds, rank 0's clock (not reduced over ranks)"}
    wgth.destroy_embedding(feat)
    wgth.destroy_wholememory_tensor(wrow)
    wgth.destroy_wholememory_tensor(wcol)
    return res


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(a):
    """`python bench.py --gpus N` with no launcher: start N ranks (one per visible GPU over RCCL; with --backend gloo the
    ranks may share a GPU) and pass rank 0's single JSON line through."""
    n = a.gpus
    have = torch.cuda.device_count()
    if a.backend == "nccl" and have < n:
        sys.stderr.write("bench.py: --gpus %d asked but only %d GPU(s) are visible; RCCL needs one device per rank "
                         "(use --backend gloo to share a device for bring-up)\n" % (n, have))
        sys.exit(2)
    port = str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WM_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1:
        self_launch(a)
    # stdout carries exactly one line, the JSON record: native libraries (RCCL prints "Librccl path : ..." on load)
    # write to fd 1 directly, so fd 1 points at stderr until the record is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # under torch.distributed.run (any N)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        torch.distributed.init_process_group(backend=a.backend, init_method="env://")
        wgth.init(rank, world, local_rank, world, "warn")
        comm = wgth.get_global_communicator()
    else:
        wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
        comm = wgth.create_group_communicator(1)
    assert wmb.lib().wholememory_ext_backend_name() == b"hip-gfx950"
    transport, transport_ranks = comm.transport()
    if world > 1 and a.backend == "nccl":
        assert (transport, transport_ranks) == ("rccl", world), "expected %d RCCL ranks, have %s" % (world, (transport, transport_ranks))

    if a.op == "sample_gather":
        def barrier0():
            torch.cuda.synchronize()
            if launched:
                torch.distributed.barrier()
            torch.cuda.synchronize()
        res = run_sample_gather(a, wgth, comm, world, rank, launched, barrier0)
        res["rccl_ranks"], res["transport"] = (transport_ranks if transport == "rccl" else 0), transport
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(res) + "\n").encode())
        if launched:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    rows_per_gpu = a.rows or (100_000_000 if world == 1 else 125_000_000)
    total_rows = rows_per_gpu * world
    mt = a.memory_type or ("chunked" if world == 1 else "distributed")
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    es = 4 if a.dtype == "f32" else 2
    policy = None
    if a.cache_ratio > 0:
        policy = wgth.create_wholememory_cache_policy(comm, memory_type=mt, memory_location="cuda",
                                                      access_type="readwrite" if a.op == "grad_apply" else "readonly",
                                                      ratio=a.cache_ratio)
    emb = wgth.create_embedding(comm, mt, a.location, tdt, [total_rows, a.dim], cache_policy=policy)
    local, start = emb.get_embedding_tensor().get_local_tensor(host_view=(a.location == "cpu"))
    fill_table(local, start)
    idx_np = make_indices(a.indices, total_rows, a.dist, 42 + rank)
    idx = torch.from_numpy(idx_np).cuda()

    def barrier():
        torch.cuda.synchronize()
        if launched:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # the output buffer is allocated once, as in the reference bench (gather_scatter_bench.cu:322-343):
    # a fresh 5 GB hipMalloc inside the timed region would cost ~140 ms and is not part of the op
    out = torch.empty((a.indices, a.dim), dtype=tdt, device="cuda")
    # Round 2 side experiment, off by default (--table-candidates / --out-candidates > 1): with the persistent launch shape of
    # rounds 1-2 the gather level followed the physical placement of the (table, output) pair (profiles/r02_placement_study.txt),
    # so a few candidate tables (unfilled) and output buffers could be allocated, every pair probed and the fastest kept — all
    # before the timed region, every probe in the line (`placement.probe_ms[table][output]`). The in-order launch shape of round 3
    # made the level independent of the placement (profiles/r03_placement_*), and the contract line runs on plain allocations.
    placement = None
    sgd_apply = a.op == "grad_apply" and a.optimizer == "sgd" and a.dtype == "f32"   # (stateful optimizers: 2-3 x the table per candidate)
    if (a.op in ("gather", "scatter") or sgd_apply) and world == 1 and (a.out_candidates > 1 or a.table_candidates > 1):
        def probe_op(tb, c):   # the op about to be timed, on one candidate pair (scatter / gradient apply: the buffer is the source)
            if a.op == "gather":
                tb.gather(idx, out=c)
            elif a.op == "scatter":
                tb.get_embedding_tensor().scatter(c, idx)
            else:
                tb.add_gradients(idx, c)
                tb.need_apply = True
                tb.apply_gradients(0.01)
        tables, cands = [emb], [out]
        if a.location == "cuda" and policy is None:
            try:
                for _ in range(a.table_candidates - 1):
                    tables.append(wgth.create_embedding(comm, mt, a.location, tdt, [total_rows, a.dim]))
            except Exception:   # no room for another 51 GB candidate: choose among those that fit
                pass
        try:
            for _ in range(a.out_candidates - 1):
                cands.append(torch.empty((a.indices, a.dim), dtype=tdt, device="cuda"))
        except torch.cuda.OutOfMemoryError:
            pass
        if sgd_apply:
            for tb in tables:
                wgth.create_wholememory_optimizer(tb, a.optimizer, {})
        grid = []
        for tb in tables:
            row = []
            for c in cands:
                for _ in range(2):
                    probe_op(tb, c)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
                for _ in range(6):
                    probe_op(tb, c)
                p1.record()
                torch.cuda.synchronize()
                row.append(p0.elapsed_time(p1) / 6)
            grid.append(row)
        flat = int(np.argmin(np.array(grid)))
        ti, oi = flat // len(cands), flat % len(cands)
        for k, tb in enumerate(tables):
            if k != ti:
                wgth.destroy_embedding(tb)
        if ti != 0:   # the kept table is one of the unfilled candidates
            emb = tables[ti]
            local, start = emb.get_embedding_tensor().get_local_tensor()
            fill_table(local, start)
        out = cands[oi]
        del tables, cands, c, tb
        torch.cuda.synchronize()
        placement = {"probe_ms": [[round(x, 4) for x in row] for row in grid], "picked": {"table": ti, "output": oi},
                     "note": "6-launch probes of the same op for every (candidate table, candidate output / source buffer) pair; "
                             "[0][0] is what single allocations give; the timed region below runs on the picked pair"}
    # where the table landed in HBM: the fixed placement probe on this rank's shard, outside the timed
    # region and non-destructive (kind 1 = random 512-byte rows read, kind 2 = read and written back), ms per GiB of rows
    table_probe = None
    if a.location == "cuda" and policy is None:
        try:
            import ctypes
            shard, _ = emb.get_embedding_tensor().get_local_tensor()
            vals = []
            for kind in (1, 2):
                ms = ctypes.c_float(0)
                wmb.check(wmb.lib().wholememory_ext_probe_memory(ctypes.c_void_p(shard.data_ptr()),
                                                                 ctypes.c_size_t(shard.numel() * shard.element_size()), kind, 3,
                                                                 ctypes.byref(ms)))
                vals.append(round(ms.value, 4))
            table_probe = {"read_ms_per_GiB": vals[0], "read_write_back_ms_per_GiB": vals[1],
                           "malloc_candidates": os.environ.get("WM_MALLOC_PROBE", "off (one plain allocation, the library's default since round 4; WM_MALLOC_PROBE=auto opts in)"),
                           "note": "random-row probe of the table's allocation; a write-side value near 0.36 serves random row WRITES "
                                   "(scatter / gradient apply) best, 0.41-0.43 up to 20 % slower; the gather's random READS do "
                                   "not follow it (profiles/r04_six_fresh_processes_probe_off_vs_default.txt)"}
            del shard
        except Exception as e:   # a probe that fails must not take the bench line with it
            table_probe = {"error": str(e)[:200]}
    opt = None
    if a.op == "grad_apply":
        if not (sgd_apply and placement is not None):   # (the placement probes gave every candidate its optimizer)
            opt = wgth.create_wholememory_optimizer(emb, a.optimizer, {})
        out.normal_()

    def step():
        if a.op == "gather":
            emb.gather(idx, out=out)
        elif a.op == "scatter":
            emb.get_embedding_tensor().scatter(out, idx)
        else:
            emb.add_gradients(idx, out)
            emb.need_apply = True
            emb.apply_gradients(0.01)

    if a.op != "gather" or a.dtype != "f32":
        a.no_check = True
    # correctness first (one untimed step), in a scope of its own: a temporary that stays referenced would sit in the middle of
    # one of the caching allocator's multi-GB blocks, the op's scratch buffers would then no longer fit their cached blocks,
    # and fresh hipMallocs (~100 ms each) would land inside the timed region — seen as 10.2 instead of 6.2 ms per step on the
    # exchange route, in some runs only
    def check_once():
        step()
        torch.cuda.synchronize()
        exp = (idx & 0xFFFFFF).to(torch.float32)
        ok = bool(torch.equal(out[:, 0], exp) and torch.equal(out[:, a.dim - 1], exp) and
                  torch.equal(out[::1009].sum(1), exp[::1009] * a.dim))
        assert ok, "gathered rows differ from the closed-form table"
    if not a.no_check and out is not None:
        check_once()
    for _ in range(max(a.warmup, 1)):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    gc.collect()
    gc.freeze()    # (see run_sample_gather: no full collection of the interpreter's long-lived objects inside the timed region)
    allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    bytes0, comb0 = wmb.lib().wholememory_ext_alltoallv_bytes(), wmb.lib().wholememory_ext_combined_gradient_calls()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    gc.unfreeze()
    # the library's own counters over the timed region (rank 0's): bytes handed to the all-to-all-v for other ranks, and how many
    # gradient steps took the route that combines a sender's duplicate rows before they travel
    a2a_bytes_per_step = (wmb.lib().wholememory_ext_alltoallv_bytes() - bytes0) / a.steps
    combined_steps = wmb.lib().wholememory_ext_combined_gradient_calls() - comb0
    fresh_allocs = torch.cuda.memory_stats().get("num_device_alloc", 0) - allocs0   # hipMallocs by the caching allocator: 0 in a steady state
    dt = torch.tensor([t1 - t0], device="cuda" if a.backend == "nccl" else "cpu", dtype=torch.float64)
    if launched:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    wall = float(dt.item())
    dev_ms = ev0.elapsed_time(ev1) / a.steps  # HIP events on the stream the kernels were launched on
    rows_kernel = wmb.lib().wholememory_ext_last_rows_kernel().decode()   # what the HIP runtime calls the kernel just run

    # Side legs, outside the timed region. They must never cost the contract line its measurement: an exception in one of
    # them is recorded in the line (`side_errors`) instead of ending the run — all ranks run the same code on the same shapes,
    # so they fail or pass together.
    side_errors = {}

    def guarded(name, leg):
        try:
            return leg()
        except Exception as ex:  # noqa
            side_errors[name] = repr(ex)[:300]
            return None

    def stability_leg():   # per-step spread: one HIP event pair per step (every rank runs it: at N > 1 the step is a collective)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.stability_steps + 1)]
        evs[0].record()
        for i in range(a.stability_steps):
            step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(a.stability_steps)])
        barrier()
        return {"steps": a.stability_steps, "min_ms": round(float(per.min()), 4),
                "median_ms": round(float(np.median(per)), 4), "p95_ms": round(float(np.percentile(per, 95)), 4),
                "max_ms": round(float(per.max()), 4),
                "note": "per-step HIP-event times on rank 0, separate from the timed region"}

    def copy_leg():
        # the practical ceiling of THIS kernel in THIS process: the same gather with ids 0, 1, 2 ... (a plain copy through the
        # same launch shape: no random reads, no translation misses) — roofline.vs_copy = random-id rate / this rate
        seq = torch.arange(a.indices, device="cuda", dtype=idx.dtype)
        for _ in range(3):
            emb.gather(seq, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            emb.gather(seq, out=out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    def zipf_leg():
        # N > 1: BASELINE config C3 proper is the Zipf-skewed batch: the same step with Zipf(1.05) ids (hot rows hashed over the
        # owners), with the library's automatic request de-duplication and with it forced off — on real links this is the
        # evidence for what the de-duplication saves
        zidx = torch.from_numpy(make_indices(a.indices, total_rows, "zipf", 4242 + rank)).cuda()
        res = {"index_distribution": "zipf(1.05), hashed", "steps": 20}
        try:
            for label, env_val in (("dedup_auto", None), ("dedup_off", "0")):
                if env_val is None:
                    os.environ.pop("WM_GATHER_DEDUP", None)
                else:
                    os.environ["WM_GATHER_DEDUP"] = env_val
                wmb.reload_knobs()   # the library reads its knobs once
                for _ in range(3):
                    emb.gather(zidx, out=out)
                barrier()
                tz = time.perf_counter()
                for _ in range(20):
                    emb.gather(zidx, out=out)
                barrier()
                dz = torch.tensor([time.perf_counter() - tz], device="cuda" if a.backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(dz, op=torch.distributed.ReduceOp.MAX)
                res[label + "_ms_per_step"] = round(float(dz.item()) / 20 * 1e3, 4)
        finally:
            os.environ.pop("WM_GATHER_DEDUP", None)
            wmb.reload_knobs()
        res["dedup_auto_value_GBps"] = round(a.indices * world * a.dim * es / (res["dedup_auto_ms_per_step"] * 1e-3) / 1e9, 2)
        return res

    def local_kernel_leg():
        # N > 1: the step is link-bound (see `exchange`); the HBM roofline object then describes the dominant HBM kernel on its
        # own — the owner-side row gather — timed on this rank's local shard outside the step loop, same ids folded into it
        import ctypes as C
        from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
        lidx = idx % rows_per_gpu
        wt, wi, wo = wrap_torch_tensor(local), wrap_torch_tensor(lidx), wrap_torch_tensor(out)
        call = lambda: wmb.check(wmb.lib().wholememory_gather(wt.handle, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                                              C.c_void_p(get_stream()), -1))
        for _ in range(3):
            call()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(10):
            call()
        k1.record()
        torch.cuda.synchronize()
        kms = k0.elapsed_time(k1) / 10
        kbytes = a.indices * (8 + 2 * a.dim * es)
        return {"bound": "hbm", "achieved": round(kbytes / (kms * 1e-3) / 1e9, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(kbytes / (kms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
                "kernel": wmb.lib().wholememory_ext_last_rows_kernel().decode(), "kernel_ms_hip_events": round(kms, 4),
                "algorithmic_bytes_per_launch": kbytes,
                "scope": "owner-side row gather on rank 0's local shard, timed outside the step loop; "
                         "the step itself is link-bound (see exchange)"}

    stability = guarded("stability", stability_leg) if a.stability_steps > 0 else None
    c3_zipf = None
    if world > 1 and a.op == "gather" and a.dist == "uniform" and mt == "distributed":
        c3_zipf = guarded("c3_zipf", zipf_leg)
    local_kernel_roofline = None
    if world > 1 and a.op == "gather" and a.dtype == "f32" and rank == 0:
        local_kernel_roofline = guarded("local_kernel_roofline", local_kernel_leg)
    if launched:
        torch.distributed.barrier()

    if rank == 0:
        lookups = a.indices * world * a.steps / wall
        out_bytes = a.dim * es
        algo_bytes = 8 + a.dim * es + a.dim * es
        res = {
            "metric": "%s_GBps_out (%s row bytes/s, reference gather_scatter_bench convention)" % (
                a.op, {"gather": "gathered output", "scatter": "scattered input", "grad_apply": "gradient"}[a.op]),
            "value": round(lookups * out_bytes / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "rccl_ranks": transport_ranks if transport == "rccl" else 0, "transport": transport,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "mlookups_per_s": round(lookups / 1e6, 1),
            "algorithmic_GBps": round(lookups * algo_bytes / 1e9, 2),
            "device_allocs_in_timed_region": fresh_allocs,
            "placement": placement,   # None: plain single allocations (the default)
            "table_probe": table_probe,
            "launch_shape": "persistent (WM_ROWS_INORDER=0)" if os.environ.get("WM_ROWS_INORDER", "1") == "0" else "in-order",
            "library_variant": VARIANT or "product",
            "config": {"workload": ("C2 chunked 1-GPU %dx%d %s table, %d %s int64 ids" if world == 1 else
                                    "C3 distributed %dx%d %s table, %d %s int64 ids per rank, RCCL alltoallv")
                                   % (total_rows, a.dim, {"f32": "fp32", "f16": "fp16", "bf16": "bf16"}[a.dtype],
                                      a.indices, a.dist),
                       "memory_type": mt, "rows_per_gpu": rows_per_gpu, "indices_per_rank": a.indices,
                       "index_distribution": a.dist},
        }
        if world == 1 and a.op == "gather" and a.dtype == "f32":
            # dominant kernel = the one row kernel launch a step consists of at N = 1 (its name, as the HIP runtime reports it,
            # is in roofline.kernel: rows_batch_kernel<long, true, 32, false, 0> for 512-byte rows)
            achieved = a.indices * algo_bytes / (dev_ms * 1e-3) / 1e9
            # HBM traffic comes from PMC counters, which need their own rocprofv3 passes (scripts/collect_traffic.py runs this
            # very command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`); the figure is per launch of the same
            # kernel on the same workload, and the bench line says where it was measured
            traffic, traffic_source = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and a.indices == 10_000_000 and a.dim == 128 and a.dist == "uniform":
                try:
                    rec = json.load(open(pmc))
                    traffic = rec.get("gather_hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_traffic.json@%s (%s)" % (rec.get("commit", "r01"), rec.get("collected", "round 1"))
                except Exception:
                    traffic = None
            # what each time is: step_ms_hip_events = HIP events around the timed loop / steps (one kernel launch per step, so
            # launch gaps are inside it: an upper bound of the kernel's duration, and what `achieved` is computed from);
            # kernel_ms_rocprof = that kernel's average duration in the rocprofv3 kernel trace of this command committed under
            # profiles/ (None until a collection of this round exists)
            # One box per number (round 6): the committed profile's kernel time is cited as THIS line's kernel_ms_rocprof only when
            # the collection it comes from ran at this run's speed (its own ms_per_step within 2 % of this run's) and the kernel
            # time does not exceed this run's step (a kernel cannot take longer than the step that contains it); otherwise it
            # is reported as what it is — another run's figures, under profile_* — and kernel_ms_rocprof stays null.
            kernel_ms_rocprof, kernel_ms_source = None, None
            profile = None
            kms = os.path.join(ROOT, "profiles", "kernel_ms.json")
            if os.path.exists(kms) and a.indices == 10_000_000 and a.dim == 128 and a.dist == "uniform":
                try:
                    rec = json.load(open(kms))
                    p_kernel, p_step = rec.get("average_ms"), rec.get("collection_ms_per_step")
                    source = "profiles/%s@%s (%d launches)" % (rec.get("file"), rec.get("commit"), rec.get("launches", 0))
                    this_step = wall / a.steps * 1e3
                    same_speed = (p_kernel is not None and p_step is not None and abs(p_step - this_step) <= 0.02 * this_step
                                  and p_kernel <= this_step)
                    if same_speed:
                        kernel_ms_rocprof, kernel_ms_source = p_kernel, source
                    profile = {"profile_kernel_ms": p_kernel, "profile_step_ms": p_step,
                               "profile_frac": round(a.indices * algo_bytes / (p_kernel * 1e-3) / 8e12, 4) if p_kernel else None,
                               "profile_source": source,
                               "profile_note": "rocprofv3 kernel trace of this command in ANOTHER run (its own step time beside "
                                               "it); cited as kernel_ms_rocprof only when that run was within 2 % of this one"}
                except Exception:
                    kernel_ms_rocprof, profile = None, None
            copy_ms = guarded("copy_leg", copy_leg)
            res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source,
                               "kernel": rows_kernel, "step_ms_hip_events": round(dev_ms, 4),
                               "kernel_ms_rocprof": kernel_ms_rocprof, "kernel_ms_source": kernel_ms_source,
                               "algorithmic_bytes_per_launch": a.indices * algo_bytes}
            if profile is not None:
                res["roofline"].update(profile)
            if copy_ms is not None:
                # the same launch as a sequential copy in the same process: the ceiling random 512-byte reads are measured against
                res["roofline"]["copy_ms_sequential_ids"] = round(copy_ms, 4)
                res["roofline"]["copy_frac"] = round(a.indices * algo_bytes / (copy_ms * 1e-3) / 8e12, 4)
                res["roofline"]["vs_copy"] = round(copy_ms / dev_ms, 4)
            if stability is not None:
                res["roofline"]["frac_at_median_step"] = round(a.indices * algo_bytes / (stability["median_ms"] * 1e-3) / 8e12, 4)
            if not a.no_cpu_baseline:
                res["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(a.dim, a.cpu_seconds))
                res["gpu_c1_host"] = guarded("gpu_c1_host", lambda: gpu_c1_host(wgth, comm))
                res["gpu_c1_host_cached"] = guarded("gpu_c1_host_cached", lambda: gpu_c1_host_cached(wgth, comm))
        if world == 1 and a.op == "scatter":
            achieved = a.indices * algo_bytes / (dev_ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(achieved / 8000.0, 4), "traffic": None, "kernel": rows_kernel,
                               "step_ms_hip_events": round(dev_ms, 4), "algorithmic_bytes_per_launch": a.indices * algo_bytes}
        if world == 1 and a.op == "grad_apply" and a.optimizer == "sgd":
            # whole call: ids + gradient rows read once, every DISTINCT table row read and written once (the duplicates' sum and
            # the SGD statement are fused into that one pass); sort and run detection are overhead, not algorithmic bytes
            n_unique = guarded("n_unique", lambda: int(torch.unique(idx).numel()))
            if n_unique is not None:
                call_bytes = a.indices * (8 + a.dim * es) + n_unique * 2 * a.dim * es
                achieved = call_bytes / (dev_ms * 1e-3) / 1e9
                res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(achieved / 8000.0, 4), "traffic": None,
                                   "scope": "whole wholememory_embedding_gather_gradient_apply call (id sort + run detection + "
                                            "fused duplicate-sum / SGD kernel), HIP events around the timed loop",
                                   "call_ms": round(dev_ms, 4), "distinct_rows": n_unique,
                                   "algorithmic_bytes_per_call": call_bytes}
        if world > 1 and a.op == "gather" and a.dtype == "f32":
            res["roofline"] = local_kernel_roofline
        if world > 1 and a.op == "gather" and mt == "distributed":
            # the step is bound by the point-to-point xGMI links: every rank pulls (W-1)/W of its rows from peers,
            # one link per peer. Uniform ids -> n/W rows + ids per ordered pair per step.
            pair_bytes = a.indices / world * (out_bytes + 8) if a.dist == "uniform" else None
            res["exchange"] = {"bound": "xgmi", "link_peak_GBps_per_direction": 76.8,
                               "bytes_per_ordered_pair_per_step": pair_bytes,
                               "achieved_GBps_per_link_direction":
                                   round(pair_bytes / (wall / a.steps) / 1e9, 2) if pair_bytes else None,
                               # what the links alone allow: every ordered pair moves its bytes over its own link direction
                               "predicted_link_bound_ms_per_step": round(pair_bytes / 76.8e9 * 1e3, 4) if pair_bytes else None,
                               "predicted_link_bound_value_GBps":
                                   round(a.indices * world * out_bytes / (pair_bytes / 76.8e9) / 1e9, 1) if pair_bytes else None,
                               "note": "rows all-to-all-v over RCCL grouped send/recv, pipelined in row chunks with the "
                                       "owner-side gather and the reorder-on-receive kernels"}
            if a.backend != "nccl":
                res["exchange"]["note"] = "BRING-UP RUN: collectives over torch.distributed/%s, not RCCL" % a.backend
        if a2a_bytes_per_step > 0:
            res.setdefault("exchange", {})["alltoallv_bytes_per_step"] = a2a_bytes_per_step
        if a.op == "grad_apply" and (world > 1 or a2a_bytes_per_step > 0):
            res["grad_route"] = {"combined_steps": combined_steps, "of": a.steps,
                                 "note": "steps whose duplicate gradient rows were folded per sender before the exchange (fold free "
                                         "of the reference's order + enough duplicates; WM_GRAD_COMBINE=0|1 forces)"}
        if stability is not None:
            res["stability"] = stability
        if c3_zipf is not None:
            res["c3_zipf"] = c3_zipf

        if side_errors:
            res["side_errors"] = side_errors
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    wgth.destroy_embedding(emb)
    if launched:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
