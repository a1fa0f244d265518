#!/bin/bash
# round 3: the new defaults (staged scatter, staged kernels in order, rows up to 5120 B) — whole GPU suite, then the recorded sweep
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
DIM_SWEEP_SETTINGS=default,inorder=0,staged=0,sscatter=0 timeout 1500 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_staged_defaults.csv 100 129 130 150 200 250 258 300 301 400 513 602 1000 1030 2>&1 | grep -i "gather\|scatter" > gpurun_out/r03/dim_sweep_staged_defaults.txt
grep -c . gpurun_out/r03/dim_sweep_staged_defaults.txt
