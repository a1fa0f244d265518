#!/bin/bash
mkdir -p gpurun_out/r03
WM_ROWS_STAGED_ALIGNED=1 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
DIM_SWEEP_SETTINGS=default,aligned=1 timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_staged_aligned.csv 20 36 48 100 136 200 300 400 600 1000 2>&1 | grep -i "gather\|scatter" | tail -60
