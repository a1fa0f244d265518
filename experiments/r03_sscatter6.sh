#!/bin/bash
mkdir -p gpurun_out/r03
DIM_SWEEP_SETTINGS=default,aligned=1 timeout 1200 python experiments/dim_sweep.py --ab --csv=gpurun_out/r03/dim_sweep_staged_aligned_small.csv 12 20 28 36 44 52 60 68 76 2>&1 | grep -i "gather" | cut -c1-175 | tail -40
