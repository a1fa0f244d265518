#!/bin/bash
# contiguous shards as the default: the whole GPU suite (hipIpc export of such blocks, 2 / 3 processes on one GPU), allocation
# time, six fresh processes per op
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import time, torch, ctypes, sys
sys.path.insert(0, '.')
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
for rows in (100_000_000, 125_000_000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, 128])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    wgth.destroy_embedding(emb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("create %d x 128 fp32 (%.1f GB): %.3f s, destroy %.3f s" % (rows, rows * 512 / 1e9, t1 - t0, t2 - t1), flush=True)
PY
O=gpurun_out/r04_six_fresh_processes_contiguous.txt
: > $O
for op in gather scatter grad_apply; do
  for i in 1 2 3 4 5 6; do
    timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('%-10s process %s  ms_per_step %.4f  frac_of_8TBps %s' % ('$op', '$i', d['ms_per_step'], r.get('frac')))
" >> $O
  done
done
cat $O
