#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/c5
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_graph_ops_gpu.py tests/test_c5_flow_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -6
for seeds in 1024 8192 65536; do python bench.py --op sample_gather --seeds $seeds --steps 50 --stability-steps 0 > $OUT/sample_gather_s$seeds.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/sample_gather_s$seeds.json')); print('seeds $seeds:', r['ms_per_step'], 'ms', r['roofline']['frac'], r['device_allocs_in_timed_region'])"; done
