#!/bin/bash
# six fresh processes per op on the end-of-round tree, defaults only (plain single allocations), plus six with WM_MALLOC_PROBE=auto for the write side
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_six_fresh_processes_final_tree.txt
: > $O
for probe in default auto; do
for op in gather scatter grad_apply; do
  [ $probe = auto ] && [ $op = gather ] && continue
  for i in 1 2 3 4 5 6; do
    if [ $probe = auto ]; then export WM_MALLOC_PROBE=auto; else unset WM_MALLOC_PROBE; fi
    timeout 600 python bench.py --op $op --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('probe=%-7s %-10s process %s  ms_per_step %.4f  frac_of_8TBps %s' % ('$probe', '$op', '$i', d['ms_per_step'], r.get('frac')))
" >> $O
  done
done
done
cat $O
