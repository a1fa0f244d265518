#!/bin/bash
# DISTRIBUTED table on one rank: what the owner-side machinery costs around the row kernel (bucketing, de-dup, reorder)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dist in uniform zipf; do
rm -rf /tmp/d1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/d1 -- python $R/bench.py --memory-type distributed --dist $dist --no-cpu-baseline --steps 20 --stability-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('distributed, 1 rank, $dist ids: ms_per_step', d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"
python3 - $(find /tmp/d1 -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('   %-90s calls %4s avg %8.1f us' % (r['Name'].replace('wm::(anonymous namespace)::','')[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
