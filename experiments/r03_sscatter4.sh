#!/bin/bash
mkdir -p gpurun_out/r03
WM_ROWS_FLAT=1 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
DIM_SWEEP_SETTINGS=default,flat=1 timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_flat_forced.csv 20 33 36 41 48 50 65 72 80 96 160 200 240 2>&1 | grep -i "gather\|scatter" | tail -60
