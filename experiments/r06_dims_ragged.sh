#!/bin/bash
# round 6: the RAGGED instantiation of step_tile_kernel (dim % 4 != 0) against the routes it replaces, whole gradient-apply calls
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06_dim_sweep_ragged.txt; : > $OUT
for dim in 513 129 127 301 130 602; do
  for mode in default off on; do
    case $mode in default) unset WM_STEP_RAGGED;; off) export WM_STEP_RAGGED=0;; on) export WM_STEP_RAGGED=1;; esac
    for rep in 1 2; do
      r=$(WM_MALLOC_PROBE=auto timeout 300 python bench.py --op grad_apply --dim $dim --rows 20000000 --indices 10000000 --no-cpu-baseline --stability-steps 0 --steps 20 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], (r.get('roofline') or {}).get('frac'))")
      echo "dim $dim WM_STEP_RAGGED=$mode grad_apply rep $rep: ms_per_step, frac = $r" | tee -a $OUT
    done
  done
done
