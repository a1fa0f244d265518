#!/bin/bash
# the contract workload (C2 gather) and its two side ops in SIX fresh processes each, back to back, default settings (plain
# allocations, automatic placement choice in wholememory_malloc): what a user gets without choosing anything
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_six_fresh_processes.txt
: > $O
for op in gather scatter grad_apply; do
  for i in 1 2 3 4 5 6; do
    timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('%-10s process %s  ms_per_step %.4f  frac_of_8TBps %s  kernel %s  write-side probe %s' % ('$op', '$i', d['ms_per_step'], r.get('frac'), (r.get('kernel') or '')[:60], (d.get('table_probe') or {}).get('read_write_back_ms_per_GiB')))
" >> $O
  done
done
cat $O
