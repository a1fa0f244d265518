#!/bin/bash
# round 6 (review item 6): gradient apply / scatter on rows that are not whole 16-byte pieces or whole 128-byte lines, incl. the
# reference's own gradient-apply test dims (127, 129, 392). Default stride (WM_EMBEDDING_ROW_ALIGN=auto), placement probe on
# (takes the table's placement class out of the comparison). 602 floats now take step_tile_kernel's 8-byte-piece instantiation.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06_dim_sweep_tile8.txt; : > $OUT
for dim in 300 602 513 1000 127 129 392 130; do
  for op in grad_apply scatter; do
    for rep in 1 2; do
      r=$(WM_MALLOC_PROBE=auto timeout 300 python bench.py --op $op --dim $dim --rows 20000000 --indices 10000000 --no-cpu-baseline --stability-steps 0 --steps 20 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], (r.get('roofline') or {}).get('frac'), (r.get('roofline') or {}).get('kernel','')[:60])")
      echo "dim $dim align auto $op rep $rep: ms_per_step, frac, kernel = $r" | tee -a $OUT
    done
  done
done
