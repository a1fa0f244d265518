#!/usr/bin/env python
"""Side measurement (not the contract bench): one-hop neighbour sampling + append_unique on a synthetic
power-law-free CSR graph resident in HBM. Usage: python experiments/sample_bench.py [nodes] [avg_degree] [centers] [fanout]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    avg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n_center = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    fanout = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
    comm = wgth.create_group_communicator(1)
    g = torch.Generator(device="cuda").manual_seed(1)
    deg = torch.randint(0, 2 * avg + 1, (nodes,), device="cuda", generator=g)
    row = torch.zeros(nodes + 1, dtype=torch.int64, device="cuda")
    row[1:] = torch.cumsum(deg, 0)
    edges = int(row[-1])
    wrow = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [nodes + 1], torch.int64, [1])
    wcol = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [edges], torch.int32, [1])
    wrow.get_local_tensor()[0].copy_(row)
    lc = wcol.get_local_tensor()[0]
    step = 1 << 28
    for s in range(0, edges, step):
        e = min(edges, s + step)
        lc[s:e] = torch.randint(0, nodes, (e - s,), device="cuda", generator=g, dtype=torch.int32)
    gs = wgth.GraphStructure()
    gs.set_csr_graph(wrow, wcol)
    centers = torch.randint(0, nodes, (n_center,), device="cuda", generator=g, dtype=torch.int32)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    ms, (off, ids, lid) = timed(lambda: gs.unweighted_sample_without_replacement_one_hop(
        centers, fanout, random_seed=7, need_center_local_output=True))
    print("graph %d nodes %d edges; %d centers fanout %d -> %d samples: sample %.3f ms (%.1f M samples/s)" % (
        nodes, edges, n_center, fanout, ids.numel(), ms, ids.numel() / ms / 1e3))
    ms2, (uniq, mapping) = timed(lambda: wgth.graph_ops.append_unique(centers, ids, True))
    print("append_unique %d + %d -> %d unique: %.3f ms" % (n_center, ids.numel(), uniq.numel(), ms2))
    ms3, _ = timed(lambda: gs.multilayer_sample_without_replacement(centers[:1024], [fanout] * 3), reps=10)
    print("3-hop [%d]*3 from 1024 seeds: %.3f ms" % (fanout, ms3))
    if os.environ.get("C5", "0") == "1":
        # BASELINE config 5 shape: 2-hop sample from a batch of 1024 seeds + gather of the 128-wide fp32 features of
        # every node of the sampled sub-graph (features in a second WholeMemory table)
        feat = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [nodes, 128])

        def step():
            tg, ei, rp, ci = gs.multilayer_sample_without_replacement(centers[:1024], [fanout, fanout])
            return feat.gather(tg[0]), tg[0].numel()
        ms5, (x, n_nodes) = timed(step, reps=20)
        print("C5 step (2-hop [%d,%d] from 1024 seeds + feature gather of %d nodes x 128 fp32): %.3f ms" % (
            fanout, fanout, n_nodes, ms5))
        for bs in (8192, 65536):
            def step_b():
                tg, ei, rp, ci = gs.multilayer_sample_without_replacement(centers[:bs], [fanout, fanout])
                return feat.gather(tg[0]), tg[0].numel()
            msb, (x, n_nodes) = timed(step_b, reps=10)
            print("  batch %6d seeds: %d nodes, %.3f ms" % (bs, n_nodes, msb))
    if os.environ.get("CPU_BASELINE", "1") == "1":
        import oracle
        rp, cl = row.cpu().numpy(), lc.cpu().numpy()
        cc = centers[:100000].cpu().numpy()
        t0 = time.perf_counter()
        oracle.sample_unweighted(rp, cl, cc, fanout, 7)
        dt = time.perf_counter() - t0
        print("oracle (1 thread) 100k centers: %.1f ms (%.2f M samples/s)" % (dt * 1e3, 100000 * min(fanout, avg) / dt / 1e6))


if __name__ == "__main__":
    main()
