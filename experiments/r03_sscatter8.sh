#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -3
DIM_SWEEP_SETTINGS=default,minrow=1M timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_pow2_small.csv 4 8 16 24 28 32 64 2>&1 | grep -i "gather\|scatter" | cut -c1-175 | tail -40
