#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_graph_ops_gpu.py tests/test_c5_flow_gpu.py -m gpu -x -q 2>&1 | tail -3
for flow in reference deferred; do for items in default 1 4 16; do
 if [ $items = default ]; then unset WM_SCAN_ITEMS; else export WM_SCAN_ITEMS=$items; fi
 timeout 600 python bench.py --op sample_gather --steps 200 --stability-steps 0 --no-cpu-baseline --c5-flow $flow 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 flow $flow scan items $items: ms_per_step', d['ms_per_step'])"
done; done
unset WM_SCAN_ITEMS
bash experiments/trace_c5.sh > gpurun_out/r04_c5_timeline.txt 2>&1; cat gpurun_out/r04_c5_timeline.txt
