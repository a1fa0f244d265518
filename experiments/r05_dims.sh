#!/bin/bash
# scatter / gradient apply on rows that are not whole lines: the default stride (auto: 128-byte lines when cheap) vs the
# reference's 16-byte padding. WM_MALLOC_PROBE=auto takes the placement class of the table out of the comparison.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for dim in 300 602 513 1000; do
  for al in auto 16; do
    for op in scatter grad_apply; do
      for rep in 1 2; do
      r=$(WM_MALLOC_PROBE=auto WM_EMBEDDING_ROW_ALIGN=$al timeout 300 python bench.py --op $op --dim $dim --rows 20000000 --indices 10000000 --no-cpu-baseline --stability-steps 0 --steps 20 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], (r.get('roofline') or {}).get('frac'))")
      echo "dim $dim align $al $op rep $rep: ms_per_step, frac = $r"
      done
    done
  done
done
