#!/bin/bash
# step_tile_kernel: second row of a duplicated id prefetched with the batch (product) vs not (variant nopre =
# -DWM_TILE_DUP1_PREFETCH=0), Zipf and uniform batches, kernel time under rocprofv3 + whole call; processes alternate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "" nopre; do for dist in zipf uniform; do
  rm -rf /tmp/zt
  EXTRA="A=1"; [ $dist = zipf ] && EXTRA="WM_GRAD_FOLD=tree"
  env $EXTRA WHOLEGRAPH_AMD_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zt -- python $R/bench.py --op grad_apply --dist $dist --no-cpu-baseline --steps 30 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-product} $dist: whole call ms_per_step', d['ms_per_step'], end='  ')"
  python3 - $(find /tmp/zt -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    n=r['Name'].replace('wm::(anonymous namespace)::','')
    if 'step_tile' in n or 'tree_fold' in n: print('%s %.1f us'%(n[5:20],float(r['AverageNs'])/1e3), end='  ')
print()
PY
done; done; done
