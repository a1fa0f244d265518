#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/grad
mkdir -p $OUT
cd $R
( AB_BATCH=8 timeout 300 python experiments/grad_inorder_ab.py sgd uniform 128 f32
  AB_BATCH=8 timeout 300 python experiments/grad_inorder_ab.py sgd zipf 128 f32
  AB_BATCH=4 timeout 300 python experiments/grad_inorder_ab.py adam uniform 128 f32
  AB_BATCH=4 timeout 300 python experiments/grad_inorder_ab.py sgd uniform 256 f16
  AB_BATCH=16 timeout 300 python experiments/grad_inorder_ab.py sgd uniform 64 f32 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/grad_inorder_ab.txt
DIM_SWEEP_SETTINGS=default,inorder=0 timeout 600 python experiments/dim_sweep.py --ab --csv=$OUT/dim_sweep_rule.csv 129 300 2>&1 | cut -c1-200 | tee $OUT/dim_sweep_rule.txt
