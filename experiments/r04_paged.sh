#!/bin/bash
# is the random-read penalty of the gather the TLB or the DRAM side? the same uniform ids grouped by 2 MiB page / 64 KiB block
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_gather_paged_ids.txt
: > $O
for rep in 1 2; do for d in uniform paged paged64k sequential; do
  timeout 600 python bench.py --dist $d --no-cpu-baseline --no-check --steps 100 --stability-steps 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('gather %-10s ids: ms_per_step %.4f frac %s' % ('$d', d['ms_per_step'], r.get('frac')))" >> $O
done; done
cat $O
