#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_c5_flow_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for flow in reference deferred; do
 timeout 600 python bench.py --op sample_gather --steps 200 --stability-steps 0 --no-cpu-baseline --c5-flow $flow 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 flow $flow: ms_per_step', d['ms_per_step'])"
done; done
bash experiments/trace_c5.sh > gpurun_out/r04_c5_timeline_deferred.txt 2>&1; cat gpurun_out/r04_c5_timeline_deferred.txt
