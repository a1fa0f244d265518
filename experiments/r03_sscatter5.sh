#!/bin/bash
mkdir -p gpurun_out/r03
WM_ROWS_STAGED_MINROW=16 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
DIM_SWEEP_SETTINGS=default,minrow=16,minrow=1M timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_staged_small_rows.csv 9 13 17 25 27 30 33 36 41 44 50 52 60 65 2>&1 | grep -i "gather\|scatter" | cut -c1-175 | tail -90
