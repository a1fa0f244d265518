#!/bin/bash
# tree fold of long duplicate runs (Zipf batch, WM_GRAD_FOLD=tree): 4 rows per thread in flight (product) vs 8 (variant depth8)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" depth8; do
  rm -rf /tmp/zt
  WM_GRAD_FOLD=tree WHOLEGRAPH_AMD_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zt -- python $R/bench.py --op grad_apply --dist zipf --no-cpu-baseline --steps 30 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-product} zipf tree: whole call ms_per_step', d['ms_per_step'], end='  ')"
  python3 - $(find /tmp/zt -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n=r['Name'].replace('wm::(anonymous namespace)::','')
    if 'step_tile' in n or 'tree_' in n: print('%s %.1f us'%(n[5:20],float(r['AverageNs'])/1e3), end='  ')
print()
PY
done; done
cd $R && timeout 900 python -m pytest tests/test_exchange_optim_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python experiments/fuzz_optim.py 300 99 2>&1 | tail -1
