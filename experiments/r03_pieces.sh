#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/pieces
mkdir -p $OUT
cd $R
DIM_SWEEP_SETTINGS="default,lds=6.6k,lds=10k,lds=20k" timeout 900 python experiments/dim_sweep.py --ab --csv=$OUT/dim_sweep_occupancy.csv 64 128 256 1024 2>&1 | cut -c1-210 | tee $OUT/dim_sweep_occupancy.txt
