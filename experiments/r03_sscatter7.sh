#!/bin/bash
# the final rule: whole GPU suite, then one recorded sweep of every shape class against the pre-staging kernels (minrow=1M switches
# the new small-row routes off, staged=0 / sscatter=0 the staged kernels altogether)
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
DIM_SWEEP_SETTINGS=default,staged=0,sscatter=0 timeout 1500 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_final_rule.csv 8 9 13 16 20 25 32 33 36 50 52 64 65 100 128 129 130 200 250 256 258 300 301 512 513 602 1000 1024 1030 > gpurun_out/r03/dim_sweep_final_rule.txt 2>&1
tail -3 gpurun_out/r03/dim_sweep_final_rule.txt | cut -c1-160
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-300
python bench.py --op scatter --no-cpu-baseline 2>/dev/null | cut -c1-300
