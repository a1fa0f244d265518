#!/bin/bash
# round 3: LDS-staged scatter / staged kernels in order / 10 KiB chunks against the flat-stream kernels on ragged rows; parity first
mkdir -p gpurun_out/r03
WM_ROWS_STAGED_MAXROW=5120 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -4
WM_ROWS_STAGED_MAXROW=5120 WM_ROWS_INORDER=0 timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -2
DIM_SWEEP_SETTINGS=default,inorder=0,staged=0,sscatter=0,maxrow=5120,maxrow=5120+inorder=0 timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_staged_scatter.csv 100 129 130 150 250 258 301 513 602 1030 2>&1 | grep -i "gather\|scatter" | tail -130
