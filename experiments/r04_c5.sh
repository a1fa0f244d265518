#!/bin/bash
# round 4: C5 chain with the single-launch scans and the side fill: tests, the bench leg, the timeline of one step
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_graph_ops_gpu.py tests/test_c5_flow_gpu.py tests/test_random_parity_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --op sample_gather --steps 200 --stability-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5 ms_per_step', d['ms_per_step'], json.dumps({k: d[k] for k in d if k in ('value','unit','metric')}))"
done
bash experiments/trace_c5.sh > gpurun_out/r04_c5_timeline.txt 2>&1; cat gpurun_out/r04_c5_timeline.txt
