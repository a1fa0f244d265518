#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_ops_gpu.py tests/test_c5_flow_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in "WM_SCAN_STATIC=0" "WM_SCAN_STATIC=1" "WM_SCAN_STATIC=0" "WM_SCAN_STATIC=1"; do
  r=$(env $v timeout 300 python bench.py --op sample_gather 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('stability',{}).get('median_ms'))")
  echo "$v  ms_per_step, median: $r"
done
