#!/bin/bash
# same box, same session: gradient apply with the split sort (default) and with rocPRIM's sort (WM_DEDUP_SPLIT=0), alternating;
# the table's write-side probe figure beside each line (placement class of the table)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
O=gpurun_out/r05/split_ab.txt
: > $O
for i in 1 2 3 4 5 6; do
  for v in default "WM_DEDUP_SPLIT=0" "WM_MALLOC_PROBE=auto"; do
    if [ "$v" = "default" ]; then e=""; else e="$v"; fi
    env $e timeout 600 python bench.py --op grad_apply --no-cpu-baseline --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
tp = d.get('table_probe') or {}
print('%-22s process %s  ms_per_step %.4f  frac %s  table probe: read %s write-back %s ms/GiB' % ('$v', '$i', d['ms_per_step'], r.get('frac'), tp.get('read_ms_per_GiB'), tp.get('read_write_back_ms_per_GiB')))
" >> $O
  done
done
cat $O
