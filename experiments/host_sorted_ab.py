"""C1 (HOST table 10 M x 64 fp32, 1 M uniform ids) through wholememory_gather: the sorted-ids route (WM_HOST_SORTED_GATHER, the
library default for HOST tables with rows <= 512 B) against the plain route, and how many low id bits the sort may ignore
(WM_HOST_SORTED_LOW_BIT) — one process per setting (the switches are read once), interleaved rounds.
usage: host_sorted_ab.py            (driver: spawns the settings)      host_sorted_ab.py --one   (one setting, prints ms)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one():
    import numpy as np, torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    torch.cuda.set_device(0)
    wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
    comm = wgth.create_group_communicator(1)
    res = []
    for rows, dim, n in [(10_000_000, 64, 1_000_000), (10_000_000, 128, 1_000_000), (10_000_000, 64, 100_000), (40_000_000, 32, 4_000_000)]:
        emb = wgth.create_embedding(comm, "chunked", "cpu", torch.float32, [rows, dim])
        local, _ = emb.get_embedding_tensor().get_local_tensor(host_view=True)
        local.fill_(1.0)
        idx = torch.from_numpy(np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)).cuda()
        out = torch.empty((n, dim), device="cuda")
        for _ in range(3):
            emb.gather(idx, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                emb.gather(idx, out=out)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 100)
        res.append("%dx%d/%d: %.3f ms %.1f GB/s" % (rows, dim, n, best, n * dim * 4 / best / 1e6))
        wgth.destroy_embedding(emb)
    print(" | ".join(res) + " | sorted gathers %d" % wmb.lib().wholememory_ext_host_sorted_gathers(), flush=True)

if "--one" in sys.argv:
    one()
else:
    settings = [("plain", {"WM_HOST_SORTED_GATHER": "0"}), ("sorted", {}), ("sorted, low bit 4", {"WM_HOST_SORTED_LOW_BIT": "4"}),
                ("sorted, low bit 8", {"WM_HOST_SORTED_LOW_BIT": "8"}), ("sorted, low bit 12", {"WM_HOST_SORTED_LOW_BIT": "12"})]
    for rnd in range(2):
        for name, env in settings:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, **env), capture_output=True, timeout=600)
            print("round %d %-20s %s" % (rnd, name, (p.stdout.decode().strip().splitlines() or [p.stderr.decode()[-300:]])[-1]), flush=True)
