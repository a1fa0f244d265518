#!/usr/bin/env python
"""append_unique's table insert: look first (WM_AU_DIRECT_CAS=0, rounds 3-4), compare-and-swap without the look (=2), and the
shipped kernel for mini-batch hops (=1): the workgroup's keys merged in LDS, then the compare-and-swap without the look — when ids
REPEAT: without the look every repeated id is one more atomic on the same address. One process, settings interleaved;
a hop-2-sized call (31 744 targets, 952 320 neighbours) with the neighbour ids drawn uniformly, Zipf(s) hashed over the nodes,
and all equal."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.graph_ops as gops
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
nodes, nt, nn = 111_059_956, 31744, int(os.environ.get("AU_NN", "952320"))
rng = np.random.default_rng(3)
def zipf(s):
    k = rng.zipf(s, nn).astype(np.uint64)
    return ((k * np.uint64(2654435761)) % np.uint64(nodes)).astype(np.int32)
cases = {"uniform": rng.integers(0, nodes, nn).astype(np.int32), "zipf 1.05": zipf(1.05), "zipf 1.3": zipf(1.3), "zipf 2.0": zipf(2.0),
         "one id": np.full(nn, 12345, np.int32)}
SETTINGS = [("round4", {"WM_AU_MERGE": "0", "WM_AU_DIRECT_CAS": "0"}), ("direct", {"WM_AU_MERGE": "0", "WM_AU_DIRECT_CAS": "2"}),
            ("merged+direct", {"WM_AU_DIRECT_CAS": "1"}), ("merged+look", {"WM_AU_DIRECT_CAS": "0"}), ("default", {})]
targets = torch.from_numpy(rng.permutation(nodes)[:nt].astype(np.int32)).cuda()
for name, nb in cases.items():
    vals, cnt = np.unique(nb, return_counts=True)
    d = torch.from_numpy(nb).cuda()
    out = []
    for name_s, env in SETTINGS * 2:
        for k in ("WM_AU_MERGE", "WM_AU_DIRECT_CAS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        wmb.reload_knobs()
        for _ in range(3): gops.append_unique(targets, d, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): gops.append_unique(targets, d, True)
        torch.cuda.synchronize()
        out.append("%s %.1f" % (name_s, (time.perf_counter() - t0) / 20 * 1e6))
    print("%-10s unique %7d  hottest id x %7d   whole append_unique call, us:  %s" % (name, len(vals), cnt.max(), "   ".join(out)), flush=True)
