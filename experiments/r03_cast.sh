#!/bin/bash
# round 3: launch shape of the converting / generic small-row kernels (A/B), then the row-kernel parity tests on the new tiles
mkdir -p gpurun_out/r03
timeout 900 python experiments/cast_sweep.py --ab > gpurun_out/r03/cast_sweep_ab.txt 2>&1
tail -60 gpurun_out/r03/cast_sweep_ab.txt
timeout 1500 python -m pytest tests/test_gather_scatter_gpu.py tests/test_golden_fixtures_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -5
