#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_embedding_cache_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_optim_gpu.py tests/test_host_sorted_gather_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 1099511627776 ""; do
WM_SORT_RADIX_MIN=$v python - <<'PY'
import sys, json, os
sys.path.insert(0, ".")
import torch, bench
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
r = bench.gpu_c1_host_cached(wgth, comm)
print("WM_SORT_RADIX_MIN=%s" % os.environ.get("WM_SORT_RADIX_MIN", "(default)"), json.dumps({k: r[k] for k in ("uniform", "zipf")}))
PY
done
