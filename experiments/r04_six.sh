#!/bin/bash
# six fresh processes per op, placement probe OFF (WM_MALLOC_PROBE=1: plain single allocations) and at its default
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_six_fresh_processes.txt
: > $O
for probe in 1 default; do
for op in gather scatter grad_apply; do
  for i in 1 2 3 4 5 6; do
    if [ $probe = 1 ]; then export WM_MALLOC_PROBE=1; else unset WM_MALLOC_PROBE; fi
    timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('probe=%-7s %-10s process %s  ms_per_step %.4f  frac_of_8TBps %s  write-side probe %s' % ('$probe', '$op', '$i', d['ms_per_step'], r.get('frac'), (d.get('table_probe') or {}).get('read_write_back_ms_per_GiB')))
" >> $O
  done
done
done
cat $O
