#!/usr/bin/env python
"""Random CSR graphs, fan-outs, seeds and dtypes: unweighted and weighted one-hop sampling, append_unique and add_self_loop
against the CPU oracle, bit for bit. usage: fuzz_sample.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.graph_ops as gops
from wholegraph_amd import binding as wmb

torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
comm = wgth.create_group_communicator(1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def wm_array(arr, mt, loc):
    t = wgth.create_wholememory_tensor(comm, mt, loc, [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
    t.get_local_tensor(host_view=(loc == "cpu"))[0].copy_(torch.from_numpy(arr))
    return t


for case in range(cases):
    n_nodes = int(rng.integers(2, 4000))
    max_deg = int(rng.choice([0, 1, 5, 40, 200]))
    deg = rng.integers(0, max_deg + 1, n_nodes)
    for _ in range(int(rng.integers(0, 4))):
        deg[rng.integers(n_nodes)] = int(rng.choice([1023, 1024, 1025, 3000, 9000]))
    row_ptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    col_dt = np.int32 if rng.random() < 0.5 else np.int64
    ctr_dt = np.int32 if rng.random() < 0.5 else np.int64
    col = rng.integers(0, n_nodes, int(row_ptr[-1])).astype(col_dt)
    if len(col) == 0:
        col = np.zeros(1, col_dt)
    mt, loc = [("continuous", "cuda"), ("chunked", "cuda"), ("chunked", "cpu"), ("distributed", "cuda")][rng.integers(4)]
    m = int(rng.choice([-1, 1, 2, 7, 30, 31, 32, 33, 64, 100, 500, 1024, 1025, 2000]))
    seed = int(rng.integers(0, 2 ** 62))
    centers = rng.integers(0, n_nodes, int(rng.choice([1, 10, 700, 5000]))).astype(ctr_dt)
    desc = "case %d: %s/%s nodes %d max_deg %d m %d centers %d col %s ctr %s" % (
        case, mt, loc, n_nodes, max_deg, m, len(centers), np.dtype(col_dt).name, np.dtype(ctr_dt).name)
    ok = True
    try:
        wrow, wcol = wm_array(row_ptr, mt, loc), wm_array(col, mt, loc)
        g = wgth.GraphStructure()
        g.set_csr_graph(wrow, wcol)
        off, ids, lid, egid = g.unweighted_sample_without_replacement_one_hop(
            torch.from_numpy(centers).cuda(), m, random_seed=seed, need_center_local_output=True, need_edge_output=True)
        o_off, o_ids, o_lid, o_egid = oracle.sample_unweighted(row_ptr, col, centers, m, seed)
        ok = ok and np.array_equal(off.cpu().numpy(), o_off) and np.array_equal(ids.cpu().numpy(), o_ids) and \
            np.array_equal(lid.cpu().numpy(), o_lid) and np.array_equal(egid.cpu().numpy(), o_egid)
        if mt != "distributed" and m > 0:   # weighted: mapped graphs only, as in the reference
            w = (rng.random(len(col)) + 0.01).astype(np.float32)
            ww = wm_array(w, mt, loc)
            g.set_edge_attribute("weight", ww)
            woff, wids, wlid, wegid = g.weighted_sample_without_replacement_one_hop(
                "weight", torch.from_numpy(centers).cuda(), m, random_seed=seed, need_center_local_output=True,
                need_edge_output=True)
            q_off, q_ids, q_lid, q_egid = oracle.sample_weighted(row_ptr, col, w, centers, m, seed)
            ok = ok and np.array_equal(woff.cpu().numpy(), q_off) and np.array_equal(wids.cpu().numpy(), q_ids) and \
                np.array_equal(wegid.cpu().numpy(), q_egid)
            wgth.destroy_wholememory_tensor(ww)
        # append_unique on (centers, sampled ids) when the dtypes agree
        if col_dt == ctr_dt and len(o_ids) > 0:
            targets = np.unique(centers)   # the op expects distinct targets
            uniq, mapping = gops.append_unique(torch.from_numpy(targets).cuda(), ids, need_neighbor_raw_to_unique=True)
            u_uniq, u_map = oracle.append_unique(targets, o_ids)
            ok = ok and np.array_equal(uniq.cpu().numpy(), u_uniq) and np.array_equal(mapping.cpu().numpy(), u_map)
        wgth.destroy_wholememory_tensor(wrow)
        wgth.destroy_wholememory_tensor(wcol)
    except Exception as ex:  # noqa
        ok = False
        print("ERROR", repr(ex)[:400], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", desc, flush=True)
print("cases %d, failures %d" % (cases, bad))
