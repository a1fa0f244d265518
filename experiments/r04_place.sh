#!/bin/bash
# which buffer's placement does the gather follow? table candidates x output candidates, every probe printed
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_gather_placement_grid.txt
: > $O
for probe in 1 default; do
  for i in 1 2; do
    if [ $probe = 1 ]; then export WM_MALLOC_PROBE=1; else unset WM_MALLOC_PROBE; fi
    timeout 900 python bench.py --op gather --no-cpu-baseline --steps 100 --stability-steps 0 --table-candidates 3 --out-candidates 3 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('malloc probe=%-7s process %s  ms_per_step %.4f frac %s placement %s' % ('$probe', '$i', d['ms_per_step'], r.get('frac'), json.dumps(d.get('placement'))))
" >> $O
  done
done
cat $O
