#!/bin/bash
# gradient apply in six fresh processes: plain allocation and WM_MALLOC_PROBE=auto (final tree of round 5)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
O=gpurun_out/r05/six_grad.txt
: > $O
for probe in default auto; do
  for i in 1 2 3 4 5 6; do
    if [ "$probe" = "auto" ]; then export WM_MALLOC_PROBE=auto; else unset WM_MALLOC_PROBE; fi
    timeout 600 python bench.py --op grad_apply --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('probe=%-7s grad_apply process %s  ms_per_step %.4f  frac_of_8TBps %s' % ('$probe', '$i', d['ms_per_step'], r.get('frac')))
" >> $O
  done
done
cat $O
