#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "" ku4; do
  rm -rf /tmp/zt
  WHOLEGRAPH_AMD_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zt -- python $R/bench.py --op grad_apply --optimizer adam --no-cpu-baseline --steps 20 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-product} adam uniform: whole call', d['ms_per_step'], end='  ')"
  python3 - $(find /tmp/zt -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    n=r['Name'].replace('wm::(anonymous namespace)::','')
    if 'step_tile' in n: print('%s %.1f us'%(n[5:45],float(r['AverageNs'])/1e3), end='  ')
print()
PY
done; done
