#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "WM_SCAN_ITEMS=16" "WM_SCAN_ITEMS=4" "WM_SCAN_ITEMS=1" "X=1" "WM_SCAN_ITEMS=4" "X=1"; do
  r=$(env $v timeout 300 python bench.py --op sample_gather 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))")
  echo "$v  ms_per_step, median: $r"
done
