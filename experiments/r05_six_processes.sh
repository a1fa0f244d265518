#!/bin/bash
# gather / scatter / gradient apply in SIX fresh processes each: defaults (plain allocation) and WM_MALLOC_PROBE=auto
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
O=gpurun_out/r05/six_fresh_processes.txt
: > $O
for probe in default auto; do
  for op in gather scatter grad_apply; do
    [ "$probe" = "auto" ] && [ "$op" = "gather" ] && continue
    for i in 1 2 3 4 5 6; do
      if [ "$probe" = "auto" ]; then export WM_MALLOC_PROBE=auto; else unset WM_MALLOC_PROBE; fi
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
