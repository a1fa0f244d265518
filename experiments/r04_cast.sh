#!/bin/bash
cd $GRAFT_REPO_ROOT
python experiments/cast_sweep.py 2>&1 | grep -E "gather|scatter" | cut -c1-170 > gpurun_out/r04_cast_sweep_final.txt
cat gpurun_out/r04_cast_sweep_final.txt
timeout 1200 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_host_sorted_gather_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python experiments/fuzz_rows.py 1500 4242 2>&1 | tail -1
